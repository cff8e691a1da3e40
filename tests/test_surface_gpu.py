"""GPU tier, the module / launcher surface around the hot path: sub-modules callable on their own like the reference's
(src/model/squeezedet.py:17-23: ``Fire.forward``; ``features[i](x)`` of the nn.Sequential at :33-49), the packed-weight
cache contract, and ``bench.py`` starting its own ranks (gloo rehearsal of N = 2 on one GPU, RCCL in a one-rank group)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


@pytest.mark.parametrize("arch", ["squeezedet", "squeezedetplus"])
def test_submodules_callable_layer_by_layer(arch):
    """Walk ``model.base.features`` one module at a time, as the reference's nn.Sequential allows, feeding each module the
    ORACLE's input for that layer: every stand-alone launch (bare stem conv, ReLU slot, pools, Fire modules, ConvDet)
    matches the oracle's per-layer capture; the whole Sequential and the fused plan agree with the walk."""
    from squeezedet_pytorch_amd.model import SqueezeDet
    size = (64, 96)
    cfg = sqd.make_cfg(arch=arch, input_size=size, device='cuda')
    m = SqueezeDet(cfg)
    sd = synthetic.make_state_dict(arch, seed=1234)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    x = synthetic.make_images(2, size, seed=3)
    cap = {}
    with torch.no_grad():
        ref_pred = oracle.backbone_forward(x, sd, arch, capture=cap)
    prev = x
    feats = m.base.features
    with torch.no_grad():
        for i, mod in enumerate(feats):
            out = mod(prev.cuda())
            ref = cap[f'features.{i}']
            assert tuple(out.shape) == tuple(ref.shape), (i, out.shape, ref.shape)
            err = (out.cpu() - ref).abs().max().item()
            assert err <= TOL * max(1.0, float(ref.abs().max())), (i, type(mod).__name__, err)
            prev = ref
        y = m.base.convdet(prev.cuda())
        assert (y.cpu() - cap['convdet']).abs().max().item() <= TOL
        # a Fire's own children, one by one (squeeze -> ReLU -> expand3x3): the pieces compose
        fire_idx = next(i for i, mod in enumerate(feats) if type(mod).__name__ == 'Fire')
        fin = cap[f'features.{fire_idx - 1}'].cuda()
        f = feats[fire_idx]
        s = torch.relu(f.squeeze(fin))
        e3 = torch.relu(f.expand3x3(s))
        whole = f(fin)
        e1 = f.expand1x1.out_channels
        assert (whole[:, e1:] - e3).abs().max().item() <= 1e-5
        # the whole Sequential, then dropout-free ConvDet, equals the fused plan
        full = m.base.convdet(feats(x.cuda()))
        pred = m.base(x.cuda())
    np.testing.assert_allclose(full.permute(0, 2, 3, 1).reshape(2, -1, 8).cpu().numpy(), pred.cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(pred.cpu().numpy(), ref_pred.numpy(), atol=TOL)
    with pytest.raises(RuntimeError):
        feats[fire_idx](x)                                  # CPU tensor: no fallback


def test_plan_cache_contract_and_invalidate():
    """Optimizer-style in-place updates refresh the packed weights by themselves; a write through ``.data`` does not move
    the version counter -- ``invalidate_plans()`` is the documented way to make it visible."""
    from squeezedet_pytorch_amd.model import SqueezeDet
    size = (64, 96)
    cfg = sqd.make_cfg(input_size=size, device='cuda')
    m = SqueezeDet(cfg)
    sd = synthetic.make_state_dict('squeezedet', seed=1234)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    x = synthetic.make_images(1, size, seed=3)
    with torch.no_grad():
        m.base(x.cuda())
        w = m.base.features[6].expand3x3.weight
        w.mul_(1.5)                                          # in-place op: version counter moves
        sd2 = {k: v.clone() for k, v in sd.items()}
        sd2['base.features.6.expand3x3.weight'] *= 1.5
        np.testing.assert_allclose(m.base(x.cuda()).cpu().numpy(), oracle.backbone_forward(x, sd2).numpy(), atol=TOL)
        w.data.mul_(2.0)                                     # invisible to the cache ...
        sd2['base.features.6.expand3x3.weight'] *= 2.0
        m.base.invalidate_plans()                            # ... until told
        np.testing.assert_allclose(m.base(x.cuda()).cpu().numpy(), oracle.backbone_forward(x, sd2).numpy(), atol=TOL)


def _run_bench(*args, timeout=900):
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), capture_output=True, text=True,
                       timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    """``python bench.py --gpus 2`` with no launcher: two fresh rank processes (gloo rehearsal, both on this box's one GPU),
    n_gpus counted by an all-reduce, ONE line with the training result inside."""
    line = _run_bench('--gpus', '2', '--backend', 'gloo', '--steps', '3', '--warmup', '2', '--no-cpu-baseline')
    assert line['n_gpus'] == 2 and line['config']['global_batch'] == 40
    assert line['train']['value'] > 0 and 'bucket' in line['train']['workload']
    assert line['value'] > 0 and line['scaling'] == 'weak'


def test_bench_rccl_path_single_rank():
    """The RCCL code path on one GPU: process group 'nccl' with device_id, the bucketed in-place all-reduce of slices of the
    flat gradient buffer on the side stream, barrier + max-over-ranks timing -- in a one-rank group."""
    line = _run_bench('--gpus', '1', '--force-dist', '--mode', 'train', '--steps', '3', '--warmup', '2')
    assert line['n_gpus'] == 1 and 'RCCL' in line['config']['workload'] and line['value'] > 0


def test_bench_gloo_world4_training_rehearsal():
    """More ranks than two without an 8-GPU node: four gloo ranks share this box's one GPU (the box admits at most six GPU
    processes) and run the real HIP backward with the bucketed, count-weighted exchange; the line counts four ranks and a
    global batch of 16."""
    line = _run_bench('--gpus', '4', '--backend', 'gloo', '--mode', 'train', '--steps', '2', '--warmup', '1', '--batch', '4',
                      '--no-cpu-baseline')
    assert line['n_gpus'] == 4 and line['config']['global_batch'] == 16 and line['value'] > 0
    assert 'bucket' in line['config']['workload']


def test_bench_rccl_training_step_is_captured():
    """With an RCCL process group the training step -- collectives included -- is timed as a hipGraph replay (one-rank group
    on this box; the N > 1 run uses the same code path, every rank agreeing through an all-reduce that the capture succeeded)."""
    line = _run_bench('--gpus', '1', '--force-dist', '--mode', 'train', '--steps', '3', '--warmup', '2')
    assert line['timed_with'] == 'hipGraph replay', line['timed_with']


def test_bench_default_line_has_pipeline_and_cpu_legs():
    """The default line carries the end-to-end leg (pinned uint8 -> H2D -> preprocess -> net -> detect -> D2H, never `value`) and
    the four CPU-baseline legs of BASELINE.md section 4."""
    line = _run_bench('--steps', '10', '--warmup', '3', '--mode', 'infer')
    pl = line['pipeline']
    assert pl['value'] > 0 and pl['results_on_host_ok'] and not pl['degraded']
    assert 20 * 375 * 1242 * 3 <= pl['h2d_bytes_per_step'] <= 1.05 * 20 * 375 * 1242 * 3      # (header + 4 KB-rounded image slots)
    dl = line['detect_dataset']
    assert dl['value'] > 0 and dl['results'] == dl['batches'] * 20 and not dl['degraded'] and not line['degraded']
    assert line['value'] >= pl['value'] * 0.5              # same kernels: the end-to-end rate is of the same order as the resident one
    cb = line['cpu_baseline']
    assert cb['kind'] == 'port' and cb['threads'] >= 1 and cb['cores'] >= 1
    legs = cb['legs']
    assert {k.split('_')[0] for k in legs} == {'bs1', 'bs20'} and len(legs) == 4
    assert all(v['value'] > 0 for v in legs.values())
    assert line['parity']['ok']


@pytest.mark.parametrize("flags", [{}, {'fuse_fire_bridge': False}, {'fuse_fire_bridge': False, 'fuse_expand_wino': False},
                                   {'fuse_fire_bridge': False, 'fuse_expand': False, 'fuse_expand_wino': False},
                                   {'fuse_fire_bridge': False, 'fuse_pool_squeeze': True}, {'use_winograd': False},
                                   {'fuse_stem_squeeze': False}, {'fuse_stem_squeeze': False, 'fuse_fire_bridge': False}])
def test_inference_launch_plan_equals_real_launches(flags):
    """plan.inference_launch_plan (host only; what tests/test_profiles.py checks profiles/traffic.json against) lists exactly
    the launches the model issues -- kernel name and shape tag, in order -- for the default flags and for non-default ones."""
    from squeezedet_pytorch_amd import ops, plan
    from squeezedet_pytorch_amd.detector import Detector
    from squeezedet_pytorch_amd.model import SqueezeDet
    cfg = sqd.make_cfg(device='cuda')
    m = SqueezeDet(cfg)
    m.load_state_dict(synthetic.make_state_dict('squeezedet', seed=1234))
    det = Detector(m, cfg)
    for k, v in flags.items():
        assert hasattr(m.base, k), k
        setattr(m.base, k, v)
    x = synthetic.make_images(20, (384, 1248), seed=0).cuda()
    det.detect_device(x)                                       # packs happen here, outside the bracketed pass
    timer = ops.KernelTimer()
    ops.set_timer(timer)
    try:
        det.detect_device(x)
    finally:
        ops.set_timer(None)
    torch.cuda.synchronize()
    got = [(r[0], r[1]) for r in timer.records]
    want = plan.inference_launch_plan('squeezedet', 20, (384, 1248), **flags)
    assert got == want, [(a, b) for a, b in zip(got, want) if a != b][:4]


@pytest.mark.parametrize("arch,bs", [("squeezedet", 20), ("squeezedetplus", 4)])
def test_training_launch_plan_equals_real_launches(arch, bs):
    """plan.training_launch_plan against the bracketed launches of one training iteration (forward with saved activations,
    loss forward / backward, backbone backward)."""
    from squeezedet_pytorch_amd import ops, plan
    from squeezedet_pytorch_amd.model import SqueezeDetWithLoss
    cfg = sqd.make_cfg(arch=arch, device='cuda')
    m = SqueezeDetWithLoss(cfg)
    m.load_state_dict(synthetic.make_state_dict(arch, seed=1234))
    m = m.cuda().train()
    batch = {'image': synthetic.make_images(bs, (384, 1248), seed=0).cuda(),
             'gt': synthetic.make_gt(bs, cfg.anchors, (384, 1248), seed=1).cuda()}

    def step():
        loss, _ = m(batch)
        m.zero_grad()
        loss.mean().backward()
    step()
    timer = ops.KernelTimer()
    ops.set_timer(timer)
    try:
        step()
    finally:
        ops.set_timer(None)
    torch.cuda.synchronize()
    got = [(r[0], r[1]) for r in timer.records]
    want = plan.training_launch_plan(arch, bs, (384, 1248))
    assert got == want, [(i, a, b) for i, (a, b) in enumerate(zip(got, want)) if a != b][:4] + [len(got), len(want)]
