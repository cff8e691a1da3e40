#!/usr/bin/env python
"""Benchmark of the SqueezeDet hot path on MI355X (contract: see the task brief / DESIGN.md).

A *step* is one pass of the inference hot path over one batch of 20 synthetic 1248x384 images that
are already resident in HBM: stem -> pools -> 10 Fire modules -> ConvDet (HIP kernels) -> fused
decode / top-64 / class-wise NMS / threshold kernel.  ``value`` = images/sec over all ranks.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode infer|train] [--no-cpu-baseline]

For N > 1 launch with ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``:
one process per GPU, batches sharded by rank, no data-path collective for inference (replicas), an
RCCL gradient all-reduce per step for training.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD (spec)
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FWD_GFLOP_PER_IMAGE = {'squeezedet': 10.566, 'squeezedetplus': 83.386}   # SURVEY.md 8d / BASELINE.md 3 (2*MAC, convs only)


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC pass (profiles/traffic.json, written by
    scratch/traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled for gfx950), or None."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            t = json.load(f)
        return int(t[kernel]['hbm_bytes_per_launch'])
    except (OSError, KeyError, ValueError):
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=0, help='images per GPU per step (default 20; 16 for squeezedetplus)')
    ap.add_argument('--mode', default='infer', choices=['infer', 'train'])
    ap.add_argument('--arch', default='squeezedet')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)')
    ap.add_argument('--no-graph', action='store_true', help='time eager launches instead of hipGraph replays')
    return ap.parse_args()


def cpu_baseline(cfg, sd, batch, seconds_budget=25.0):
    """The oracle (CPU restatement of the reference, kind='port') timed on the host cores on a
    bounded sample of the same workload: whole inference path for `batch` images."""
    import oracle
    from squeezedet_pytorch_amd import synthetic
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    x = synthetic.make_images(batch, cfg.input_size, seed=0)

    def one():
        with torch.no_grad():
            pred = oracle.backbone_forward(x, sd, cfg.arch)
            ids, sc, bx = oracle.inference_head(pred, cfg.anchors, cfg.input_size, cfg.num_classes)
        for b in range(batch):
            oracle.filter_detections(ids[b].numpy(), sc[b].numpy(), bx[b].numpy(), cfg.keep_top_k, cfg.nms_thresh,
                                     cfg.score_thresh, cfg.num_classes)
    t0 = time.time(); one(); warm = time.time() - t0
    n, t_acc = 0, 0.0
    while n < 3 or (t_acc + warm < seconds_budget and n < 10):
        t0 = time.time(); one(); t_acc += time.time() - t0; n += 1
        if t_acc + warm > seconds_budget:
            break
    return {'value': round(batch * n / t_acc, 2), 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'sample': f'{n} timed passes (1 warm-up) of the oracle CPU path (torch CPU fp32 backbone + numpy decode/top-k/NMS) '
                      f'on the same bs={batch} 1248x384 synthetic batch, {threads} threads of {cores} host CPUs'}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback in the product path)')
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.backend == 'nccl':
        raise SystemExit(f'LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible')
    dev_index = local_rank % max(ndev, 1)      # == local_rank except in a gloo rehearsal on fewer GPUs
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import squeezedet_pytorch_amd as sqd
    from squeezedet_pytorch_amd import ops, synthetic
    from squeezedet_pytorch_amd.detector import Detector
    from squeezedet_pytorch_amd.model import SqueezeDet

    cfg = sqd.make_cfg(arch=args.arch, device=dev)
    sd = synthetic.make_state_dict(args.arch, seed=1234)
    B = args.batch if args.batch > 0 else (16 if args.arch == 'squeezedetplus' else 20)
    x = synthetic.make_images(B, cfg.input_size, seed=rank).to(dev)

    if args.mode == 'train':
        from squeezedet_pytorch_amd.trainer import make_train_step
        step, describe = make_train_step(cfg, sd, x, rank, world, dist)
    else:
        model = SqueezeDet(cfg)
        model.load_state_dict(sd)
        det = Detector(model, cfg)
        out_bufs = ops._det_buffers(B, cfg.keep_top_k, dev, cfg.num_anchors)

        def step():
            return det.detect_device(x, out=out_bufs)
        where = '1 MI355X' if world == 1 else f'each of {world} MI355X (independent replicas, no data-path collective)'
        describe = (f'SqueezeDet KITTI 1248x384 bs={B} inference on {where} (Fire+ConvDet HIP kernels, fused NMS)' if args.arch == 'squeezedet'
                    else f'SqueezeDet+ wider Fire modules at 1248x384 bs={B} inference on {where}')

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (eager), then capture one step into a hipGraph for the timed region ----
    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    graph = None
    # inference: always a hipGraph of the step.  training: the whole step (fwd, loss, bwd, clip, SGD, weight re-pack) is
    # captured too on one GPU; with an RCCL all-reduce in the step (world > 1) it stays eager
    if not args.no_graph and (args.mode == 'infer' or world == 1):
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()                                   # allocations of the capture stream's pool
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            graph.replay(); graph.replay()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            if args.mode == 'infer':
                raise
            print(f'[bench] training step not captured ({type(e).__name__}: {e}); timing eager launches', file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    run = graph.replay if graph is not None else step

    # ---- timed region: EXACTLY K steps, barrier + synchronize on both sides ----
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    elapsed = time.perf_counter() - t0

    # ---- per-kernel HIP-event pass, inside the same process right after the timed region: every kernel of
    # `nprof` eager steps is bracketed on its launch stream.  Two un-bracketed steps are enqueued first so the
    # host runs ahead of the GPU and no bracket absorbs a launch gap (brackets are only exact when the GPU is
    # the bottleneck; checked against rocprofv3 --kernel-trace, profiles/). ----
    nprof = 5
    timer = ops.KernelTimer()
    run(); run(); run()
    ops.set_timer(timer)
    for _ in range(nprof):
        step()
    ops.set_timer(None)
    torch.cuda.synchronize()
    summ = timer.summary(nsteps=nprof)               # per-step totals from per-shape medians
    dominant = max(summ.items(), key=lambda kv: kv[1]['ms'])[0] if summ else None

    if dist is not None:
        t = torch.tensor([elapsed], device=dev if args.backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_images = B * world * args.steps
        value = total_images / elapsed
        roof = None
        if dominant is not None:
            d = summ[dominant]
            avg_s = d['ms'] / d['launches'] / 1e3
            flops_per_launch = d['flops'] / d['launches']
            bytes_per_launch = d['bytes'] / d['launches']
            ai = flops_per_launch / max(bytes_per_launch, 1.0)
            if ai > PEAK_FP32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9):    # ridge point 19.7 flop/B
                ach = flops_per_launch / avg_s / 1e12
                roof = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(ach / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': None}
            else:
                ach = bytes_per_launch / avg_s / 1e9
                roof = {'bound': 'hbm', 'achieved': round(ach, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                        'frac': round(ach / PEAK_HBM_GBS, 4), 'traffic': None}
            roof['traffic'] = measured_traffic(dominant)
            if dominant.startswith('conv_wino'):
                # Winograd F(2x2,3x3): `achieved` counts the multiply-adds the MFMA pipe executes; the same launch expressed in
                # direct-form 3x3 flops (what the implicit-GEMM kernel would have to execute) is 2.25x that
                roof['direct_form_equivalent_tflops'] = round(ach * 2.25, 2)
            roof.update({'kernel': dominant, 'launches_per_step': int(round(d['launches'])),
                         'avg_launch_us': round(avg_s * 1e6, 2),
                         'algorithmic_per_launch': {'gflop': round(flops_per_launch / 1e9, 3), 'mbytes': round(bytes_per_launch / 1e6, 3)},
                         'share_of_step': round(d['ms'] / (elapsed / args.steps * 1e3), 3),
                         'how': f'HIP events around every launch of {nprof} eager steps enqueued behind the timed region; median per kernel shape'})
        gf = FWD_GFLOP_PER_IMAGE[args.arch]
        whole = {'tflops': round(value * gf / 1e3, 2),
                 'frac_of_fp32_mfma_peak': round(value / world * gf / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4),
                 'note': 'direct-form flops of the network; layers run by the Winograd kernel execute 2.25x fewer'} if args.mode == 'infer' else None
        kernels = {k: {'ms_per_step': round(v['ms'], 4),
                       'launches_per_step': int(round(v['launches'])),
                       'tflops': round(v['flops'] / (v['ms'] / 1e3) / 1e12, 2) if v['ms'] > 0 else 0,
                       'gbs': round(v['bytes'] / (v['ms'] / 1e3) / 1e9, 1) if v['ms'] > 0 else 0}
                   for k, v in sorted(summ.items(), key=lambda kv: -kv[1]['ms'])}
        # the two layer families BASELINE.json's north_star asks for: HBM rate of the memory-bound 1x1 squeeze layers,
        # matrix-core rate of the 3x3 expand / ConvDet layers (forward launches of this step; per-shape medians)
        import re as _re
        fam = {'squeeze_1x1': [0.0, 0.0, 0.0, 0, 0.0], 'expand3x3_convdet': [0.0, 0.0, 0.0, 0, 0.0]}       # ms, direct flops, bytes, launches, executed flops
        for kname, v in summ.items():
            for tag, t in v['tags'].items():
                m = _re.match(r'(\d+)tap C(\d+) N(\d+) ', tag)
                if not m or len(t) < 4:
                    continue
                taps, Cc, Nn = int(m.group(1)), int(m.group(2)), int(m.group(3))
                key = 'squeeze_1x1' if (taps == 1 and Nn < Cc) else ('expand3x3_convdet' if taps == 9 else None)
                if key is None:
                    continue
                direct = t[2] * (2.25 if kname.startswith('conv_wino') else 1.0)                  # direct-form flops of the layer
                f = fam[key]; f[0] += t[1]; f[1] += direct; f[2] += t[3]; f[3] += int(round(t[0])); f[4] += t[2]
        layer_families = {
            'squeeze_1x1': {'launches_per_step': fam['squeeze_1x1'][3], 'ms_per_step': round(fam['squeeze_1x1'][0], 4),
                            'achieved_gbs': round(fam['squeeze_1x1'][2] / max(fam['squeeze_1x1'][0], 1e-9) / 1e6, 1),
                            'frac_of_hbm_peak': round(fam['squeeze_1x1'][2] / max(fam['squeeze_1x1'][0], 1e-9) / 1e6 / PEAK_HBM_GBS, 4),
                            'note': 'algorithmic bytes (input + output windows + weights) / HIP-event time; includes the squeeze data gradients in training mode'},
            'expand3x3_convdet': {'launches_per_step': fam['expand3x3_convdet'][3], 'ms_per_step': round(fam['expand3x3_convdet'][0], 4),
                                  'executed_tflops': round(fam['expand3x3_convdet'][4] / max(fam['expand3x3_convdet'][0], 1e-9) / 1e9, 2),
                                  'frac_of_fp32_mfma_peak': round(fam['expand3x3_convdet'][4] / max(fam['expand3x3_convdet'][0], 1e-9) / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4),
                                  'direct_form_tflops': round(fam['expand3x3_convdet'][1] / max(fam['expand3x3_convdet'][0], 1e-9) / 1e9, 2),
                                  'frac_of_fp32_mfma_peak_direct_form': round(fam['expand3x3_convdet'][1] / max(fam['expand3x3_convdet'][0], 1e-9) / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4),
                                  'note': 'executed = multiply-adds the matrix cores perform (Winograd launches: direct form / 2.25); direct_form = the 3x3 convolution flops'},
        }
        cpu = None
        if not args.no_cpu_baseline and args.mode == 'infer' and world == 1:      # rank 0 at N = 1 only (bounded sample)
            cpu = cpu_baseline(cfg, sd, B)
        line = {
            'metric': f'images/sec {"SqueezeDet" if args.arch == "squeezedet" else "SqueezeDet+"} 1248x384 bs={B} ' + ('inference' if args.mode == 'infer' else 'training'),
            'value': round(value, 1), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': describe, 'arch': args.arch, 'images_per_gpu_per_step': B, 'global_batch': B * world,
                       'input': '3x384x1248 fp32 NCHW, HBM resident', 'weights': 'synthetic Kaiming-scale, seed 1234',
                       'parallelism': f'replicas x{world}' if args.mode == 'infer' else f'dp{world}'},
            'roofline': roof, 'cpu_baseline': cpu, 'whole_network': whole, 'timed_with': 'hipGraph replay' if graph is not None else 'eager launches',
            'layer_families': layer_families,
            'kernels_event_profile': kernels,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
