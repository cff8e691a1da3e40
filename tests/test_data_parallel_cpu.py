"""CPU tier, N>1 path: world_size-2 gloo processes exercise the gradient all-reduce / sharding logic
of trainer.py (the HIP kernels themselves cannot run here; per-shard gradients come from the oracle).
Checks the reference semantic: loss.mean() over the GLOBAL batch (src/engine/trainer.py:43)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle
    import squeezedet_pytorch_amd as sqd
    from squeezedet_pytorch_amd import synthetic
    from squeezedet_pytorch_amd.trainer import allreduce_gradients, shard_sizes
    size = (64, 96)
    cfg = sqd.make_cfg(input_size=size, device='cpu')
    sd = synthetic.make_state_dict('squeezedet', seed=1234)
    B = 4
    x = synthetic.make_images(B, size, seed=3)
    gt = synthetic.make_gt(B, cfg.anchors, size, seed=2, min_boxes=2, max_boxes=3)
    sizes = shard_sizes(B, world)
    lo = sum(sizes[:rank]); hi = lo + sizes[rank]
    # local step exactly as Trainer.run_epoch does it: local mean, backward, all-reduce SUM, /W
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pred = oracle.backbone_forward(x[lo:hi], params)
    loss_vec, _ = oracle.multitask_loss(pred, gt[lo:hi], cfg.anchors, size)
    loss_vec.mean().backward()
    plist = list(params.values())
    flat = allreduce_gradients(plist, world)
    assert flat.numel() == 2082120
    torch.save({k: v.grad for k, v in params.items()}, os.path.join(out_dir, f'g{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_gloo_world2_gradient_allreduce_equals_global_mean(tmp_path):
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    import oracle
    import squeezedet_pytorch_amd as sqd
    from squeezedet_pytorch_amd import synthetic
    size = (64, 96)
    cfg = sqd.make_cfg(input_size=size, device='cpu')
    sd = synthetic.make_state_dict('squeezedet', seed=1234)
    x = synthetic.make_images(4, size, seed=3)
    gt = synthetic.make_gt(4, cfg.anchors, size, seed=2, min_boxes=2, max_boxes=3)
    _, _, grads, total, _, _ = oracle.train_step_reference(sd, None, x, gt, cfg.anchors, size)
    g0 = torch.load(os.path.join(tmp_path, 'g0.pt'))
    g1 = torch.load(os.path.join(tmp_path, 'g1.pt'))
    for k in grads:
        assert torch.equal(g0[k], g1[k]), f'{k}: ranks disagree after all-reduce'
        ref = grads[k]
        assert (g0[k] - ref).abs().max().item() <= 1e-4 * max(float(ref.abs().max()), 1e-3), k


def _exchange_worker(rank, world, port, out_dir, B=5):
    """Unequal shards (B = 5 on 2 ranks: 3 + 2) through trainer.GradientExchange exactly as backward.py drives it:
    begin -> ready(tail bucket) -> ready(middle) -> ready(head) -> finish; then the same gradients as ONE bucket."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2 if world <= 2 else 1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle
    import squeezedet_pytorch_amd as sqd
    from squeezedet_pytorch_amd import synthetic
    from squeezedet_pytorch_amd.trainer import GradientExchange, allreduce_gradients, shard_sizes
    size = (64, 96)
    cfg = sqd.make_cfg(input_size=size, device='cpu')
    sd = synthetic.make_state_dict('squeezedet', seed=1234)
    x = synthetic.make_images(B, size, seed=3)
    gt = synthetic.make_gt(B, cfg.anchors, size, seed=2, min_boxes=2, max_boxes=3)
    sizes = shard_sizes(B, world)
    lo = sum(sizes[:rank]); hi = lo + sizes[rank]
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pred = oracle.backbone_forward(x[lo:hi], params)
    loss_vec, _ = oracle.multitask_loss(pred, gt[lo:hi], cfg.anchors, size)
    loss_vec.mean().backward()                                     # local shard mean, as the Trainer does
    names = list(params)
    total = sum(p.numel() for p in params.values())
    local = torch.cat([params[k].grad.reshape(-1) for k in names])
    offs, o = {}, 0
    for k in names:
        offs[k] = o; o += params[k].numel()
    results = {}
    for label, cuts in (('three', [offs['base.features.9.squeeze.weight'], offs['base.features.6.squeeze.weight'], 0]), ('one', [0])):
        flat = torch.empty(total + 1)
        flat[:total] = local
        ex = GradientExchange()
        ex.begin(flat, total, sizes[rank])
        top = total
        for c in cuts:                                              # tail first, like the backward walk
            ex.scale_slice(c, top)                                  # (the HIP backward weights a stage's gradients where it produces them)
            ex.ready(c, top); top = c
        ex.finish()
        results[label] = flat[:total].clone()
        assert len(ex.buckets_last_step) == len(cuts) and ex.buckets_last_step[0][1] == total + 1
    if world == 2:
        assert torch.equal(results['three'], results['one'])        # bucketed == single bucket, bitwise (a + b is order-free)
    else:
        # with more ranks the collective's reduction order depends on where an element sits in its bucket (ring / halving-
        # doubling chunks): the two bucketings agree to fp32 rounding of a sum of `world` terms, not bitwise
        tol = 4e-6 * float(results['one'].abs().max())
        assert float((results['three'] - results['one']).abs().max()) <= tol
    # the stand-alone helper with the same weighting
    plist = list(params.values())
    allreduce_gradients(plist, world, local_batch=sizes[rank])
    torch.save({'flat': results['three'], 'names': names, 'helper': torch.cat([p.grad.reshape(-1) for p in plist]), 'local': local,
                'n': sizes[rank]},
               os.path.join(out_dir, f'e{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_gloo_world2_bucketed_exchange_unequal_shards(tmp_path):
    """B = 5 over 2 ranks (3 + 2 images): the bucketed, count-weighted exchange reproduces the gradient of the mean over
    the GLOBAL batch (src/engine/trainer.py:43 on the gathered vector), identically on both ranks."""
    world, port = 2, 31500 + (os.getpid() % 2000)
    mp.spawn(_exchange_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    import oracle
    import squeezedet_pytorch_amd as sqd
    from squeezedet_pytorch_amd import synthetic
    size = (64, 96)
    cfg = sqd.make_cfg(input_size=size, device='cpu')
    sd = synthetic.make_state_dict('squeezedet', seed=1234)
    x = synthetic.make_images(5, size, seed=3)
    gt = synthetic.make_gt(5, cfg.anchors, size, seed=2, min_boxes=2, max_boxes=3)
    _, _, grads, _, _, _ = oracle.train_step_reference(sd, None, x, gt, cfg.anchors, size)
    e0 = torch.load(os.path.join(tmp_path, 'e0.pt'))
    e1 = torch.load(os.path.join(tmp_path, 'e1.pt'))
    assert torch.equal(e0['flat'], e1['flat'])
    ref = torch.cat([grads[k].reshape(-1) for k in e0['names']])
    scale = float(ref.abs().max())
    assert (e0['flat'] - ref).abs().max().item() <= 1e-4 * scale
    assert (e0['helper'] - ref).abs().max().item() <= 1e-4 * scale


@pytest.mark.timeout(900)
def test_gloo_world8_bucketed_exchange_config4_shape(tmp_path):
    """BASELINE config 4's shape of job (8 ranks, one shard each) rehearsed on the CPU: B = 11 over 8 ranks = shards of
    2,2,2,1,1,1,1,1 images through the same begin / ready x3 / finish sequence the HIP backward drives, count slot riding in
    the first bucket.  Every rank ends with the identical buffer, equal to the gradient of the mean over the global batch
    (src/engine/trainer.py:43, src/utils/data_parallel.py:93-113 scatter by chunk_sizes)."""
    from squeezedet_pytorch_amd.trainer import shard_sizes
    world, B, port = 8, 11, 39500 + (os.getpid() % 2000)
    assert shard_sizes(B, world) == [2, 2, 2, 1, 1, 1, 1, 1]
    mp.spawn(_exchange_worker, args=(world, port, str(tmp_path), B), nprocs=world, join=True)
    import oracle
    import squeezedet_pytorch_amd as sqd
    from squeezedet_pytorch_amd import synthetic
    size = (64, 96)
    cfg = sqd.make_cfg(input_size=size, device='cpu')
    sd = synthetic.make_state_dict('squeezedet', seed=1234)
    x = synthetic.make_images(B, size, seed=3)
    gt = synthetic.make_gt(B, cfg.anchors, size, seed=2, min_boxes=2, max_boxes=3)
    _, _, grads, _, _, _ = oracle.train_step_reference(sd, None, x, gt, cfg.anchors, size)
    e = [torch.load(os.path.join(tmp_path, f'e{r}.pt')) for r in range(world)]
    for r in range(1, world):
        assert torch.equal(e[0]['flat'], e[r]['flat']), r
    # (1) the exchange itself, tight: the count-weighted mean of the ranks' local gradients, recomputed here in float64
    want = sum(e[r]['local'].double() * e[r]['n'] for r in range(world)) / B
    scale = float(want.abs().max())
    assert (e[0]['flat'].double() - want).abs().max().item() <= 2e-6 * scale
    assert (e[0]['helper'].double() - want).abs().max().item() <= 2e-6 * scale
    # (2) the semantic: that IS the gradient of the mean over the global batch.  Loose bar: the one-process oracle run on 11 images
    # and the per-shard runs on 1-2 images take different CPU convolution paths, and an activation within rounding of zero flips
    # its ReLU mask between them (DESIGN.md "Backward parity and ReLU flips")
    ref = torch.cat([grads[k].reshape(-1) for k in e[0]['names']])
    assert (e[0]['flat'] - ref).abs().max().item() <= 2e-3 * scale


def test_bench_refuses_more_gpus_than_visible():
    """``bench.py --gpus N`` without a launcher starts the ranks itself; with fewer than N GPUs visible (none here) it must
    exit non-zero instead of printing a one-GPU number."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1'],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '"metric"' not in r.stdout


def test_shard_sizes():
    from squeezedet_pytorch_amd.trainer import shard_sizes
    assert shard_sizes(160, 8) == [20] * 8
    assert shard_sizes(20, 1) == [20]
    assert shard_sizes(22, 4) == [6, 6, 5, 5] and sum(shard_sizes(23, 8)) == 23


def test_allreduce_flat_view_fast_path():
    """Gradients that already are consecutive views of one flat buffer (what the HIP backward emits) are all-reduced in
    place: the returned bucket aliases them, values unchanged at world size 1."""
    import torch
    from squeezedet_pytorch_amd.trainer import allreduce_gradients
    ps = [torch.nn.Parameter(torch.zeros(3, 2)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2, 2, 1, 1))]
    flat = torch.arange(6 + 5 + 4, dtype=torch.float32)
    off = 0
    for p in ps:
        p.grad = flat[off:off + p.numel()].view_as(p); off += p.numel()
    out = allreduce_gradients(ps, world=1)
    assert out.data_ptr() == flat.data_ptr() and out.numel() == flat.numel()
    assert torch.equal(out, torch.arange(15, dtype=torch.float32))
    # not consecutive -> gather / scatter path, same values
    ps[1].grad = torch.full((5,), 7.0)
    out2 = allreduce_gradients(ps, world=1)
    assert out2.data_ptr() != flat.data_ptr() and torch.equal(out2[6:11], torch.full((5,), 7.0))


def test_fused_optimizer_refuses_cpu_parameters():
    """trainer.FusedClipSGD is a HIP launch over device pointers: CPU parameters are an error, not a silent fallback."""
    import pytest as _pytest
    import torch as _torch
    from squeezedet_pytorch_amd.trainer import FusedClipSGD
    with _pytest.raises(ValueError):
        FusedClipSGD([_torch.nn.Parameter(_torch.zeros(4))], lr=0.1)
