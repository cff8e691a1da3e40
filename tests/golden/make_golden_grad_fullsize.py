#!/usr/bin/env python
"""Golden for the FULL-SIZE backward: the REFERENCE's own ``SqueezeDetWithLoss`` (imported read-only from
/root/reference/src), run on the CPU in float64 AND in float32 on the first NIMG images of the bs=20 1248x384 benchmark batch (same
synthetic weights / images / ground truth as bench.py and tests/test_headline_gpu.py, dropout off), ``loss.mean().backward()``
(src/engine/trainer.py:43-47).  Stored per parameter tensor: the float64 gradient's L2 norm, 256 sampled entries (seeded indices), and
the relative L2 deviation of the reference's float32 run from its float64 run -- the measured noise floor of fp32 through the ReLU /
max-pool masks that the GPU comparison's per-tensor bar is derived from.  Build container only; output = data.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_grad_fullsize.py        (about three minutes on 8 cores)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import squeezedet_pytorch_amd as sqd  # noqa: E402
from squeezedet_pytorch_amd import synthetic  # noqa: E402
from make_golden import import_reference  # noqa: E402

NIMG = 4
SIZE = (384, 1248)
NSAMP = 256


def run(ref_model, cfg, sd, x, gt, dtype):
    m = ref_model.SqueezeDetWithLoss(cfg).train()
    m.load_state_dict(sd, strict=True)
    m = m.to(dtype)
    loss, _ = m({'image': x.to(dtype), 'gt': gt.to(dtype)})
    loss.mean().backward()
    return loss.detach().double().numpy(), {k: p.grad.detach().double() for k, p in m.named_parameters()}


def main():
    ref_model, _, _, _ = import_reference()
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = sqd.make_cfg(arch='squeezedet', input_size=SIZE, device='cpu', dropout_prob=0.0)
    sd = synthetic.make_state_dict('squeezedet', seed=1234)
    x = synthetic.make_images(20, SIZE, seed=0)[:NIMG]
    gt = synthetic.make_gt(20, cfg.anchors, SIZE, seed=1)[:NIMG]
    t0 = time.time()
    loss64, g64 = run(ref_model, cfg, sd, x, gt, torch.float64)
    t1 = time.time()
    loss32, g32 = run(ref_model, cfg, sd, x, gt, torch.float32)
    print(f'float64 {t1 - t0:.0f} s, float32 {time.time() - t1:.0f} s')
    rs = np.random.RandomState(77)
    out = {'names': np.array(list(g64.keys())), 'nimg': np.array(NIMG), 'loss64': loss64, 'loss32': loss32}
    norms, dev32, idxs, vals = [], [], [], []
    for k, g in g64.items():
        flat = g.reshape(-1).numpy()
        n = flat.size
        idx = np.sort(rs.choice(n, size=min(NSAMP, n), replace=False)).astype(np.int64)
        pad = np.full(NSAMP, -1, np.int64); pad[:idx.size] = idx
        v = np.zeros(NSAMP, np.float64); v[:idx.size] = flat[idx]
        norms.append(float(np.linalg.norm(flat)))
        dev32.append(float(np.linalg.norm(g32[k].reshape(-1).numpy() - flat) / max(np.linalg.norm(flat), 1e-300)))
        idxs.append(pad); vals.append(v)
        print(f'{k:40s} |g| {norms[-1]:.6e}   fp32 vs fp64 rel-L2 {dev32[-1]:.2e}')
    out.update(grad_norm64=np.array(norms), fp32_rel_l2=np.array(dev32), sample_idx=np.stack(idxs), sample_val64=np.stack(vals))
    np.savez_compressed(os.path.join(HERE, 'grad_fullsize.npz'), **out)
    print('wrote grad_fullsize.npz', os.path.getsize(os.path.join(HERE, 'grad_fullsize.npz')), 'bytes')


if __name__ == '__main__':
    main()
