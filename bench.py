#!/usr/bin/env python
"""Benchmark of the SqueezeDet hot path on MI355X (contract: see the task brief / DESIGN.md section 5).

A *step* is one pass of the hot path over one batch of 20 synthetic 1248x384 images that are already resident in HBM.
  inference step: stem -> pools -> 10 Fire modules -> ConvDet (HIP kernels) -> fused decode / top-64 / class-wise NMS
  training step : forward + multi-task loss + backward + (RCCL gradient all-reduce) + clip_grad_norm_(5) + SGD
``value`` = inference images/sec over all ranks (BASELINE.json's target is stated on inference); the default run also
times the training step and reports it in the same JSON line under ``"train"`` (the metric is "infer+train").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode both|infer|train] [--no-cpu-baseline]

N > 1: one process per GPU over RCCL.  Either the launcher starts the ranks (``python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N``: RANK / LOCAL_RANK / WORLD_SIZE in the environment) or -- WORLD_SIZE unset --
this script starts N fresh rank processes itself BEFORE anything touches the GPU and relays rank 0's line.  Batches
are sharded by rank; inference has no data-path collective (replicas), training all-reduces the gradient buckets.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD (spec)
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FWD_GFLOP_PER_IMAGE = {'squeezedet': 10.566, 'squeezedetplus': 83.386}   # SURVEY.md 8d / BASELINE.md 3 (2*MAC, convs only)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100, help='timed steps (default 100: a >= 160 ms window; the driver passes its own K)')
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=0, help='images per GPU per step (default 20; 16 for squeezedetplus)')
    ap.add_argument('--mode', default='both', choices=['both', 'infer', 'train'],
                    help='both (default): value = inference, training reported under "train"; infer / train: that step only')
    ap.add_argument('--arch', default='squeezedet')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU oracle leg (baseline timing + parity check)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)')
    ap.add_argument('--no-graph', action='store_true', help='time eager launches instead of hipGraph replays')
    ap.add_argument('--layers', action='store_true', help='add per-layer (kernel, shape) event times to the line')
    ap.add_argument('--force-dist', action='store_true', help='initialise torch.distributed even at N=1 (exercises the RCCL path on one GPU)')
    ap.add_argument('--inflight', type=int, default=2,
                    help='inference steps in flight: 2 = two captured copies of the step (own activations and result buffers) replayed '
                         'alternately on two streams, so the serial tail of a step (every kernel\'s last round, the detect launch) '
                         'overlaps the head of the next; 1 = back-to-back replays on one stream (reported either way)')
    ap.add_argument('--no-pipeline', action='store_true', help='skip the end-to-end leg (pinned uint8 -> H2D -> preprocess -> net -> detect -> D2H)')
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves (fresh processes; this parent never touches the GPU)
# ------------------------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    import torch                                     # import only: device_count() does not initialise the GPU on this image
    ndev = torch.cuda.device_count()
    if args.backend == 'nccl' and ndev < args.gpus:
        print(f'[bench] --gpus {args.gpus} but only {ndev} GPU(s) visible', file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if p.returncode != 0 or line is None:
        print(f'[bench] rank processes failed (exit {p.returncode})', file=sys.stderr)
        return p.returncode or 1
    if json.loads(line).get('n_gpus') != args.gpus:
        print(f'[bench] {json.loads(line).get("n_gpus")} ranks joined, expected {args.gpus}', file=sys.stderr)
        return 1
    print(line)
    return 0


# ------------------------------------------------------------------------------------------------------------------
# CPU leg (rank 0, N = 1 only): the oracle as reported baseline AND as the checker of the step that was just timed
# ------------------------------------------------------------------------------------------------------------------
def physical_cores():
    """(physical cores, logical CPUs) of the host: distinct thread-sibling sets in sysfs (falls back to the logical count)."""
    import glob
    logical = os.cpu_count() or 1
    sib = set()
    for f in glob.glob('/sys/devices/system/cpu/cpu[0-9]*/topology/thread_siblings_list'):
        try:
            with open(f) as fh:
                sib.add(fh.read().strip())
        except OSError:
            pass
    return (len(sib) if sib else logical), logical


def cpu_baseline_and_parity(cfg, sd, batch, hip_pred, hip_det, train_probe, seconds_budget=25.0):
    """The oracle (CPU restatement of the reference, kind='port') on the host cores, on a bounded sample of the same
    workload: the whole inference path (backbone + decode + top-64 / class-wise NMS / threshold) at
    bs=`batch` and bs=1, with all usable threads and with 1 thread.  Its first pass (bs=`batch`, all threads) doubles as the
    parity check of the timed step: ``hip_pred`` [B,A,8] / ``hip_det`` (count, class_ids, scores, boxes, anchor_idx) are the
    HIP outputs for the same batch; ``train_probe`` = (gt, hip eval-mode loss vector before the first optimizer step) or None."""
    import numpy as np
    import torch
    import oracle
    from squeezedet_pytorch_amd import synthetic
    cores, logical = physical_cores()
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = logical
    try:                                               # a container's CPU quota (cgroup v2): quota / period CPUs
        with open('/sys/fs/cgroup/cpu.max') as fh:
            q, per = fh.read().split()[:2]
        if q != 'max':
            usable = max(1, min(usable, -(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    threads = max(1, min(usable, cores, 64))          # one thread per physical core this process may run on, at most 64
    torch.set_num_threads(threads)
    x = synthetic.make_images(batch, cfg.input_size, seed=0)
    keep = {}

    def one(xs=x):
        with torch.no_grad():
            pred = oracle.backbone_forward(xs, sd, cfg.arch)
            ids, sc, bx = oracle.inference_head(pred, cfg.anchors, cfg.input_size, cfg.num_classes)
        for b in range(xs.shape[0]):
            oracle.filter_detections(ids[b].numpy(), sc[b].numpy(), bx[b].numpy(), cfg.keep_top_k, cfg.nms_thresh,
                                     cfg.score_thresh, cfg.num_classes)
        keep['pred'] = pred
    t_start = time.time()
    t0 = time.time(); one(); warm = time.time() - t0
    parity = None
    if hip_pred is not None:
        # (1) backbone: every image of the timed batch vs the oracle; (2) the fused detect kernel given the HIP pred
        # must keep exactly the anchors the oracle filter keeps on that same pred (index-exact), (3) training: the
        # eval-mode loss vector of the initial weights vs the oracle loss on the oracle pred
        ref = keep['pred']
        max_err = float((hip_pred - ref).abs().max())
        ids, sc, bx = oracle.inference_head(hip_pred, cfg.anchors, cfg.input_size, cfg.num_classes)
        cnt, _cls, _sc, _bx, idx = hip_det
        exact, ndet = True, 0
        for b in range(batch):
            d = oracle.filter_detections(ids[b].numpy(), sc[b].numpy(), bx[b].numpy(), cfg.keep_top_k, cfg.nms_thresh,
                                         cfg.score_thresh, cfg.num_classes)
            n = int(cnt[b])
            want = np.zeros(0, np.int64) if d is None else np.asarray(d['anchor_idx'])
            exact = exact and n == len(want) and bool(np.array_equal(np.asarray(idx[b, :n]), want))
            ndet += n
        parity = {'pred_max_abs_err': max_err, 'pred_tol': 1e-4, 'index_exact': exact, 'detections': ndet,
                  'images_checked': batch, 'ok': bool(max_err <= 1e-4 and exact),
                  'what': 'pred of the timed batch vs oracle.backbone_forward; kept anchor indices of the fused detect kernel vs '
                          'oracle.filter_detections on the same pred'}
        if train_probe is not None:
            gt, hip_loss = train_probe
            with torch.no_grad():
                lo, _ = oracle.multitask_loss(ref, gt, cfg.anchors, cfg.input_size, cfg.num_classes)
            rel = float(((hip_loss - lo).abs() / lo.abs().clamp_min(1e-12)).max())
            parity['train_loss_max_rel_err'] = rel
            parity['ok'] = bool(parity['ok'] and rel <= 1e-4)

    def timed(xs, nthreads, share, min_n=2, max_n=10):
        """>= min_n passes over ``xs`` within ``share`` seconds (at most max_n); returns (images/sec, passes, seconds)."""
        torch.set_num_threads(nthreads)
        t0 = time.time(); one(xs); first = time.time() - t0        # warm-up at this shape / thread count
        n, acc = 0, 0.0
        while n < min_n or (acc + first < share and n < max_n):
            t0 = time.time(); one(xs); acc += time.time() - t0; n += 1
            if acc + first > share and n >= min_n:
                break
        return xs.shape[0] * n / acc, n, acc + first
    # budget split: the headline leg (bs=batch, all threads) gets what the warm pass leaves of ~45 %; the other three share the rest
    left = max(seconds_budget - (time.time() - t_start), 6.0)
    n, t_acc = 0, 0.0
    while n < 2 or (t_acc < 0.45 * left and n < 10):
        t0 = time.time(); one(); t_acc += time.time() - t0; n += 1
    legs = {f'bs{batch}_threads{threads}': {'value': round(batch * n / t_acc, 2), 'passes': n, 'images_per_pass': batch}}
    left = max(seconds_budget - (time.time() - t_start), 4.0)
    v, k, _ = timed(x[:1], threads, 0.15 * left)
    legs[f'bs1_threads{threads}'] = {'value': round(v, 2), 'passes': k, 'images_per_pass': 1}
    v, k, _ = timed(x[:1], 1, 0.3 * left)
    legs['bs1_threads1'] = {'value': round(v, 2), 'passes': k, 'images_per_pass': 1}
    # bs=batch on ONE thread would take minutes: a 2-image sample of the same batch (the per-image cost of a 1-thread pass does
    # not depend on the batch size: no parallelism to amortise)
    sample = min(2, batch)
    v, k, _ = timed(x[:sample], 1, 0.45 * left, min_n=1, max_n=3)
    legs[f'bs{batch}_threads1'] = {'value': round(v, 2), 'passes': k, 'images_per_pass': sample,
                                   'note': f'{sample}-image sample of the bs={batch} batch'}
    torch.set_num_threads(threads)
    cpu = {'value': legs[f'bs{batch}_threads{threads}']['value'], 'unit': 'images/sec', 'cores': cores, 'threads': threads,
           'logical_cpus': logical, 'kind': 'port', 'legs': legs,
           'sample': f'{n} timed passes (1 warm-up) of the oracle CPU path (torch CPU fp32 backbone + numpy decode/top-k/NMS) '
                     f'on the same bs={batch} 1248x384 synthetic batch, {threads} threads on a host with {cores} physical cores / '
                     f'{logical} logical CPUs ({usable} usable by this process); legs: bs={batch} and bs=1, all threads and 1 thread '
                     f'(BASELINE.md section 4), {time.time() - t_start:.0f} s of CPU work in total'}
    return cpu, parity


# ------------------------------------------------------------------------------------------------------------------
# roofline of the dominant kernel from the per-launch HIP-event pass
# ------------------------------------------------------------------------------------------------------------------
def measured_traffic(kernel, launches_per_step, mode='infer'):
    """HBM-side bytes per launch of `kernel` from the committed PMC pass (profiles/traffic.json, written by
    scratch/traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled for gfx950; inference entries at
    the top level, the training step's under "train") -- only if that pass profiled the SAME launch set (launches of
    this kernel per step), else None."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            t = json.load(f)
        if mode == 'train':
            t = t['train']
        e = t[kernel]
        per_step = e.get('launches_per_step')
        if per_step is not None and int(per_step) != int(launches_per_step):
            return None
        return int(e['hbm_bytes_per_launch'])
    except (OSError, KeyError, ValueError):
        return None


def roofline_of(summ, ms_per_step, nprof, mode='infer'):
    if not summ:
        return None, None
    dominant = max(summ.items(), key=lambda kv: kv[1]['ms'])[0]
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
    roof['traffic'] = measured_traffic(dominant, int(round(d['launches'])), mode)
    if roof['traffic'] is not None and bytes_per_launch > 0:
        roof['traffic_over_algorithmic'] = round(roof['traffic'] / bytes_per_launch, 3)
    if dominant.startswith('conv_wino') or dominant.startswith('conv_wgrad_wino'):
        # Winograd F(2x2,3x3): `achieved` counts the multiply-adds the MFMA pipe executes; the same launch expressed in
        # direct-form 3x3 flops (what the implicit-GEMM kernel would have to execute) is 2.25x that
        roof['direct_form_equivalent_tflops'] = round(ach * 2.25, 2)
    # the same figure per launch shape of the dominant kernel: the aggregate above averages shapes whose grids fill the chip differently
    # (e.g. C96 -> N384 runs 3.5 rounds of workgroups, C48 -> N192 1.76) -- [launches per step, us per launch, fraction of the same peak]
    per_shape = {}
    for tag, (n_l, ms_l, fl_l, by_l) in sorted(d['tags'].items(), key=lambda kv: -kv[1][1]):
        if n_l <= 0 or ms_l <= 0:
            continue
        frac_l = (fl_l / (ms_l / 1e3) / 1e12 / PEAK_FP32_MFMA_TFLOPS) if roof['bound'] == 'mfma' else (by_l / (ms_l / 1e3) / 1e9 / PEAK_HBM_GBS)
        per_shape[tag] = {'launches_per_step': round(n_l, 2), 'avg_launch_us': round(ms_l / n_l * 1e3, 2), 'frac': round(frac_l, 4)}
    roof['per_shape'] = per_shape
    roof.update({'kernel': dominant, 'launches_per_step': int(round(d['launches'])),
                 'avg_launch_us': round(avg_s * 1e6, 2),
                 'algorithmic_per_launch': {'gflop': round(flops_per_launch / 1e9, 3), 'mbytes': round(bytes_per_launch / 1e6, 3)},
                 'share_of_step': round(d['ms'] / ms_per_step, 3),
                 'how': f'HIP events around every launch of {nprof} eager steps enqueued behind the timed region; median per kernel shape'})
    kernels = {k: {'ms_per_step': round(v['ms'], 4),
                   'launches_per_step': int(round(v['launches'])),
                   'tflops': round(v['flops'] / (v['ms'] / 1e3) / 1e12, 2) if v['ms'] > 0 else 0,
                   'gbs': round(v['bytes'] / (v['ms'] / 1e3) / 1e9, 1) if v['ms'] > 0 else 0}
               for k, v in sorted(summ.items(), key=lambda kv: -kv[1]['ms'])}
    return roof, kernels


def layer_families_of(summ):
    """The two layer families BASELINE.json's north_star asks for: HBM rate of the memory-bound 1x1 squeeze layers,
    matrix-core rate of the 3x3 expand / ConvDet layers (forward launches of this step; per-shape medians)."""
    fam = {'squeeze_1x1': [0.0, 0.0, 0.0, 0, 0.0], 'expand3x3_convdet': [0.0, 0.0, 0.0, 0, 0.0]}       # ms, direct flops, bytes, launches, executed flops
    for kname, v in summ.items():
        for tag, t in v['tags'].items():
            m = re.match(r'(\d+)tap C(\d+) N(\d+) ', tag)
            if not m or len(t) < 4:
                continue
            taps, Cc, Nn = int(m.group(1)), int(m.group(2)), int(m.group(3))
            key = 'squeeze_1x1' if (taps == 1 and Nn < Cc) else ('expand3x3_convdet' if taps == 9 else None)
            if key is None:
                continue
            direct = t[2] * (2.25 if kname.startswith('conv_wino') else 1.0)                  # direct-form flops of the layer
            f = fam[key]; f[0] += t[1]; f[1] += direct; f[2] += t[3]; f[3] += int(round(t[0])); f[4] += t[2]
    s, e = fam['squeeze_1x1'], fam['expand3x3_convdet']
    return {
        'squeeze_1x1': {'launches_per_step': s[3], 'ms_per_step': round(s[0], 4),
                        'achieved_gbs': round(s[2] / max(s[0], 1e-9) / 1e6, 1),
                        'frac_of_hbm_peak': round(s[2] / max(s[0], 1e-9) / 1e6 / PEAK_HBM_GBS, 4),
                        'note': 'algorithmic bytes (input + output windows + weights) / HIP-event time; includes the squeeze data gradients in training mode'},
        'expand3x3_convdet': {'launches_per_step': e[3], 'ms_per_step': round(e[0], 4),
                              'executed_tflops': round(e[4] / max(e[0], 1e-9) / 1e9, 2),
                              'frac_of_fp32_mfma_peak': round(e[4] / max(e[0], 1e-9) / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4),
                              'direct_form_tflops': round(e[1] / max(e[0], 1e-9) / 1e9, 2),
                              'frac_of_fp32_mfma_peak_direct_form': round(e[1] / max(e[0], 1e-9) / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4),
                              'note': 'executed = multiply-adds the matrix cores perform (Winograd launches: direct form / 2.25); direct_form = the 3x3 convolution flops'},
    }


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))                  # nothing above this line touches the GPU

    import numpy as np  # noqa: F401
    import torch

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
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
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        # the number of ranks that actually joined: an all-reduce of ones (not the environment variable)
        ones = torch.ones(1, device=dev if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(ones)
        joined = int(ones.item())
        if joined != args.gpus:
            raise SystemExit(f'{joined} ranks joined, --gpus {args.gpus}')
    else:
        joined = 1

    import squeezedet_pytorch_amd as sqd
    from squeezedet_pytorch_amd import ops, synthetic
    from squeezedet_pytorch_amd.detector import Detector
    from squeezedet_pytorch_amd.model import SqueezeDet

    cfg = sqd.make_cfg(arch=args.arch, device=dev)
    sd = synthetic.make_state_dict(args.arch, seed=1234)
    B = args.batch if args.batch > 0 else (16 if args.arch == 'squeezedetplus' else 20)
    x = synthetic.make_images(B, cfg.input_size, seed=rank).to(dev)
    net = 'SqueezeDet' if args.arch == 'squeezedet' else 'SqueezeDet+'

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        t = torch.tensor([seconds], device=dev if args.backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def measure(step, capture):
        """W warm-up steps, optional hipGraph capture of one step, then EXACTLY K timed steps bracketed by barrier +
        synchronize; a second identical window right behind it shows whether the first ran at steady clocks.
        Returns (seconds for K steps (max over ranks), seconds of the repeat window, 'hipGraph replay' | 'eager launches', run)."""
        for _ in range(max(1, args.warmup)):
            step()
        torch.cuda.synchronize()
        graph = None
        if capture and not args.no_graph:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    step()                                   # allocations of the capture stream's pool
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    # with a process group alive other threads of this process (the collective library's watchdog) may touch
                    # the runtime while this thread captures: only this thread's calls belong to the capture
                    mode = {'capture_error_mode': 'thread_local'} if dist is not None else {}
                    with torch.cuda.graph(graph, stream=side, **mode):
                        step()
                torch.cuda.current_stream().wait_stream(side)
                graph.replay(); graph.replay()
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001 -- a failed capture must not cost the measurement: time eager launches
                print(f'[bench] step not captured ({type(e).__name__}: {e}); timing eager launches', file=sys.stderr)
                graph = None
                torch.cuda.synchronize()
            if dist is not None and capture:
                # every rank replays or every rank launches eagerly: a rank whose capture failed would otherwise issue its
                # collectives from another code path than its peers
                ok = torch.tensor([1.0 if graph is not None else 0.0], device=dev if args.backend == 'nccl' else 'cpu')
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if float(ok.item()) < 1.0:
                    graph = None
        run = graph.replay if graph is not None else step
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        barrier()
        elapsed = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        barrier()
        repeat = time.perf_counter() - t0
        return max_over_ranks(elapsed), max_over_ranks(repeat), ('hipGraph replay' if graph is not None else 'eager launches'), run

    def event_profile(step, run, nprof=5):
        """Per-kernel HIP-event pass inside the same process right behind the timed region: every kernel of `nprof` eager
        steps is bracketed on its launch stream.  Un-bracketed steps are enqueued first so the host runs ahead of the GPU
        and no bracket absorbs a launch gap (brackets are only exact when the GPU is the bottleneck; checked against
        rocprofv3 --kernel-trace, profiles/)."""
        timer = ops.KernelTimer()
        run(); run(); run()
        ops.set_timer(timer)
        for _ in range(nprof):
            step()
        ops.set_timer(None)
        torch.cuda.synchronize()
        return timer.summary(nsteps=nprof), nprof

    result = {}
    hip_pred = hip_det = train_probe = None

    # ---------------- end-to-end legs (reported beside the main line, never `value`) ----------------
    def pipeline_leg(det):
        """What the reference's only published figure measures (README 117 FPS on V100: ``detect_dataset``,
        src/engine/detector.py:52-85 = data loading + network + per-image NMS + D2H), minus disk and JPEG decode, through the
        package's own executor (``Detector.stream`` = lanes.DetectStream): per batch B KITTI-sized (375x1242) uint8 HWC images that
        sit in a pinned staging buffer (``Staging.view``: a decoder would write them there) -> ONE H2D copy on the copy stream ->
        ``preprocess_kernel`` + the lane's captured step (backbone -> fused detect) -> ONE D2H copy of the packed results into
        pinned memory -> unpacked on the host.  Timed like the main leg (barrier + synchronize both sides, K batches)."""
        import numpy as np
        H0, W0 = 375, 1242
        ex = det.stream(lanes=max(1, args.inflight), graph=not args.no_graph)
        rs = np.random.RandomState(7 + rank)
        pix = [rs.randint(0, 256, (H0, W0, 3), dtype=np.uint8) for _ in range(4)]

        def one(fill):
            st = ex.stage(B)
            for b in range(B):
                v = st.view(b, H0, W0)
                if fill:
                    v[:] = pix[b % len(pix)]
            ex.submit(st)
            while ex.pending() > 2 * len(ex._lanes) - 1:
                ex.fetch()
        for i in range(max(60, args.warmup, 3 * len(ex._host))):          # fills every staging buffer of the ring, captures every lane, warms the clocks
            one(fill=i < len(ex._host))
        last = ex.drain()[-1][1]
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one(False)
        ex.drain()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        ms = elapsed / args.steps * 1e3
        total = ex._host[0]['hdr'] + B * ex._host[0]['slot']
        # the upload alone (same pinned buffer, same copy stream, nothing else running): is the leg bound by the host link?
        ncopy = max(4, min(args.steps, 20))
        dst = torch.empty(total, dtype=torch.uint8, device=dev)
        with torch.cuda.stream(ex._copy):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(ex._copy)
            for i in range(ncopy):
                dst.copy_(ex._host[i % len(ex._host)]['t'][:total], non_blocking=True)
            e1.record(ex._copy)
        torch.cuda.synchronize()
        h2d_alone_ms = e0.elapsed_time(e1) / ncopy
        how = (f'lanes.DetectStream, {len(ex._lanes)} lanes: preprocess_kernel (eager) + hipGraph replay of net + detect per lane, copies on their own streams'
               if ex.replayed_batches and not ex.degraded else 'lanes.DetectStream, eager launches')
        return {'value': round(B * joined * args.steps / elapsed, 1), 'unit': 'images/sec', 'ms_per_step': round(ms, 4),
                'h2d_bytes_per_step': total, 'd2h_bytes_per_step': int(ops.det_packed_layout(B, cfg.keep_top_k)[1]),
                'pcie_h2d_gbs_implied': round(total / (ms / 1e3) / 1e9, 2),
                'h2d_alone': {'ms_per_batch': round(h2d_alone_ms, 4), 'gbs': round(total / (h2d_alone_ms / 1e3) / 1e9, 2),
                              'note': 'the same pinned uint8 batch uploaded back to back with nothing else running: the floor the '
                                      'host link sets for ms_per_step of this leg'},
                'timed_with': how, 'degraded': bool(ex.degraded or (not args.no_graph and not ex.replayed_batches)),
                'results_on_host_ok': bool(int(last.count.sum()) > 0),
                'what': f'Detector.stream(): per batch pinned uint8 {B}x{H0}x{W0}x3 (+ header) -> one H2D copy -> preprocess_kernel -> captured lane step '
                        f'(backbone -> fused detect) -> one D2H copy of (count, class_ids, scores, boxes, anchor_idx) '
                        f'-> host arrays; {len(ex._lanes)} lanes, results fetched late; excludes disk read, JPEG decode and the packing '
                        f'copy (pixels are produced in place in the pinned buffer); NOT part of `value`'}

    def dataset_leg(det):
        """``Detector.detect_dataset`` itself (the reference's published driver, src/engine/detector.py:52-85) on synthetic uint8
        "files" held in memory: ``cfg.num_workers`` loader threads hand out KITTI-sized arrays and pack them into the pinned staging
        buffers, the main thread runs the lanes and builds the per-image result dicts.  Includes everything but disk and decode."""
        import contextlib
        import io
        import numpy as np
        H0, W0 = 375, 1242
        rs = np.random.RandomState(11 + rank)
        pix = [rs.randint(0, 256, (H0, W0, 3), dtype=np.uint8) for _ in range(8)]

        class InMemory:
            rgb_mean = rgb_std = None

            def __init__(self, n):
                self.n = n

            def __len__(self):
                return self.n

            def load_image(self, i):
                return pix[i % len(pix)], f'{i:06d}'
        nb = max(8, min(args.steps, 60))
        old = (cfg.batch_size, getattr(cfg, 'num_workers', 4), getattr(cfg, 'print_interval', 10))
        cfg.batch_size, cfg.num_workers, cfg.print_interval = B, min(8, max(1, (os.cpu_count() or 2) // 2)), 1 << 30
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                det.detect_dataset(InMemory(40 * B))                      # warm: captures the lanes at this batch size, clocks up
                barrier()
                t0 = time.perf_counter()
                res = det.detect_dataset(InMemory(nb * B))
                barrier()
            elapsed = max_over_ranks(time.perf_counter() - t0)
        finally:
            workers = cfg.num_workers
            cfg.batch_size, cfg.num_workers, cfg.print_interval = old
        ex = det.stream()
        return {'value': round(nb * B * joined / elapsed, 1), 'unit': 'images/sec', 'ms_per_batch': round(elapsed / nb * 1e3, 4),
                'batches': nb, 'loader_threads': workers, 'results': len(res), 'detections': int(sum(len(r.get('scores', ())) for r in res)),
                'degraded': bool(ex.degraded),
                'what': f'Detector.detect_dataset over {nb * B} in-memory uint8 {H0}x{W0} images, bs={B}: loader threads pack into pinned '
                        f'staging, lanes replay captured steps, per-image result dicts built on the host; excludes disk + JPEG decode'}

    # ---------------- inference ----------------
    def bench_infer():
        nonlocal hip_pred, hip_det
        model = SqueezeDet(cfg)
        model.load_state_dict(sd)
        det = Detector(model, cfg)
        out_bufs = ops._det_buffers(B, cfg.keep_top_k, dev, cfg.num_anchors)

        eager_steps = [0]

        def infer_step():
            eager_steps[0] += 1
            return det.detect_device(x, out=out_bufs)
        elapsed, repeat, how, run = measure(infer_step, capture=True)
        summ, nprof = event_profile(infer_step, run)
        serial_ms = elapsed / args.steps * 1e3
        inflight = 1
        degraded = (how != 'hipGraph replay' and not args.no_graph)          # the serial step fell back to eager launches
        if args.inflight >= 2 and how == 'hipGraph replay':
            # Steps in flight, through the package's own executor (Detector.stream() = lanes.DetectStream, the mode detect_dataset
            # runs in): the K timed steps are the same K passes over the resident batch, but step i + 1 goes to another lane than
            # step i (its own stream, captured graph, activations and packed result buffer), so it starts while step i drains -- the
            # last round of every persistent kernel and the 160-workgroup detect launch leave most of the chip idle.  Every step's
            # compact results are copied to pinned memory and unpacked on the host inside the timed region.
            try:
                ex = det.stream(lanes=args.inflight)

                def run_lanes(n):
                    last = None
                    for _ in range(n):
                        ex.submit_device(x)
                        while ex.pending() > 2 * args.inflight - 1:
                            last = ex.fetch()[1]
                    for _t, last in ex.drain():
                        pass
                    return last
                # eager first use, capture, then enough replays to bring the clocks back up: building the executor probes the streams'
                # hardware queues with idle kernels (tens of ms of a nearly idle GPU), and the first ~10 ms after idle run at a lower clock
                run_lanes(max(args.warmup, 3 * args.inflight + 2, 60))
                barrier()
                t0 = time.perf_counter()
                last = run_lanes(args.steps)
                barrier()
                e2 = max_over_ranks(time.perf_counter() - t0)
                t0 = time.perf_counter()
                run_lanes(args.steps)
                barrier()
                r2 = max_over_ranks(time.perf_counter() - t0)
                want = [t.cpu().numpy() for t in out_bufs[:5]]
                got = (last.count, last.class_ids, last.scores, last.boxes, last.anchor_idx)
                if not all((a == b).all() for a, b in zip(got, want)):
                    raise RuntimeError('the lanes of the in-flight run disagree with the serial step')
                if ex.degraded or ex.captures < args.inflight:
                    raise RuntimeError('a lane did not capture its step')
                elapsed, repeat, inflight = e2, r2, args.inflight
                how = (f'Detector.stream(): lanes.DetectStream, {inflight} lanes, one captured hipGraph per lane, packed results copied to '
                       f'pinned memory and unpacked per step')
            except Exception as e:  # noqa: BLE001 -- the serial measurement stands, flagged
                print(f'[bench] in-flight run failed ({type(e).__name__}: {e}); reporting the serial replays', file=sys.stderr)
                torch.cuda.synchronize()
                degraded = True
        ms = elapsed / args.steps * 1e3
        roof, kernels = roofline_of(summ, serial_ms, nprof)      # (kernel shares are of the serial step: in flight they overlap)
        value = B * joined * args.steps / elapsed
        gf = FWD_GFLOP_PER_IMAGE[args.arch]
        where = '1 MI355X' if joined == 1 else f'each of {joined} MI355X (independent replicas, no data-path collective)'
        result['infer'] = {
            'value': round(value, 1), 'ms_per_step': round(ms, 4), 'timed_with': how, 'degraded': bool(degraded),
            'repeat_window_ms_per_step': round(repeat / args.steps * 1e3, 4),
            'steps_in_flight': inflight, 'serial_ms_per_step': round(serial_ms, 4),      # (back-to-back replays on ONE stream)
            'value_serial': round(B * joined / (serial_ms / 1e3), 1),                    # the bs=B step rate (what rounds 1-3 reported as value)
            f'value_inflight{inflight}': round(value, 1),
            'eager_steps_launched': eager_steps[0],      # (profiling scripts divide launch counts by this; with --no-graph = every step)
            'workload': (f'SqueezeDet KITTI 1248x384 bs={B} inference on {where} (Fire+ConvDet HIP kernels, fused NMS)' if args.arch == 'squeezedet'
                         else f'SqueezeDet+ wider Fire modules at 1248x384 bs={B} inference on {where}'),
            'roofline': roof, 'kernels_event_profile': kernels, 'layer_families': layer_families_of(summ),
            'layers': {f'{k} | {tag}': round(t[1] * 1e3, 1) for k, v in summ.items() for tag, t in v['tags'].items()},      # us per step
            'whole_network': {'direct_form_tflops': round(value * gf / 1e3, 2),
                              'direct_form_tflops_over_fp32_mfma_peak': round(value / joined * gf / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4),
                              'note': 'NOT a utilisation: direct-form flops of the network / step time / peak; the layers run by the Winograd '
                                      'kernels execute 2.25x fewer multiply-adds (the achieved fraction of the matrix pipe is roofline.frac)'},
        }
        if not args.no_pipeline and args.arch == 'squeezedet':
            try:
                result['infer']['pipeline'] = pipeline_leg(det)
            except Exception as e:  # noqa: BLE001 -- the end-to-end leg is a reported extra: its failure must not cost the main line
                print(f'[bench] pipeline leg failed ({type(e).__name__}: {e})', file=sys.stderr)
                result['infer']['pipeline'] = None
                torch.cuda.synchronize()
            try:
                result['infer']['detect_dataset'] = dataset_leg(det)
            except Exception as e:  # noqa: BLE001
                print(f'[bench] detect_dataset leg failed ({type(e).__name__}: {e})', file=sys.stderr)
                result['infer']['detect_dataset'] = None
                torch.cuda.synchronize()
        if rank == 0 and joined == 1 and not args.no_cpu_baseline:
            # outputs of the step that was timed, for the parity check of the CPU leg: the detections the last replay left
            # in out_bufs, and pred from one more (bitwise identical: tests/test_headline_gpu.py) eager backbone pass
            run()
            torch.cuda.synchronize()
            with torch.no_grad():
                hip_pred = model.base(x).cpu()
            hip_det = tuple(t.cpu().numpy() for t in out_bufs[:5])
        # release the captured step (graph exec, its memory pool) before the training half: nothing after this needs it
        det.__dict__.pop('_streams', None)
        del model, det, run, infer_step
        import gc
        gc.collect()
        torch.cuda.empty_cache()

    # ---------------- training ----------------
    def bench_train():
        nonlocal train_probe
        from squeezedet_pytorch_amd.trainer import make_train_step
        step_parts = {}
        step, describe, probe = make_train_step(cfg, sd, x, rank, joined, dist, force_exchange=args.force_dist, parts=step_parts,
                                                fused_optimizer=True)
        if rank == 0 and joined == 1 and not args.no_cpu_baseline and args.mode == 'both':
            train_probe = probe()                    # (gt, eval-mode loss of the initial weights), before any optimizer step
        # the whole step (fwd, loss, bwd, gradient exchange, clip, SGD, weight re-pack) replays as a hipGraph: RCCL collectives
        # issued through torch.distributed are capturable (the bucketed all-reduces on the side stream fork from and join the
        # capturing stream); a gloo group copies through the host and stays eager
        eager_steps = [0]
        inner = step

        def step():
            eager_steps[0] += 1
            return inner()
        wanted_graph = (dist is None or args.backend == 'nccl') and not args.no_graph
        try:
            elapsed, repeat, how, run = measure(step, capture=(dist is None or args.backend == 'nccl'))
        except Exception as e:  # noqa: BLE001
            print(f'[bench] training step not captured ({type(e).__name__}: {e}); timing eager launches', file=sys.stderr)
            torch.cuda.synchronize()
            args.no_graph = True
            elapsed, repeat, how, run = measure(step, capture=False)
        summ, nprof = event_profile(step, run, nprof=3)
        ms = elapsed / args.steps * 1e3
        roof, kernels = roofline_of(summ, ms, nprof, mode='train')
        result['train'] = {
            'value': round(B * joined * args.steps / elapsed, 1), 'unit': 'images/sec', 'ms_per_step': round(ms, 4), 'timed_with': how,
            'degraded': bool(wanted_graph and how != 'hipGraph replay'),       # a capture that failed: eager launches were timed instead
            'repeat_window_ms_per_step': round(repeat / args.steps * 1e3, 4),
            'eager_steps_launched': eager_steps[0],
            'workload': describe, 'roofline': roof, 'kernels_event_profile': kernels, 'layer_families': layer_families_of(summ),
            'layers': {f'{k} | {tag}': round(t[1] * 1e3, 1) for k, v in summ.items() for tag, t in v['tags'].items()},
        }

    # Order of the two halves: inference, then training, with or without a process group.  (Round 2 ran training first when ranks
    # communicate, after a two-ranks-on-one-GPU gloo rehearsal showed eager training steps behind the graph-replay phase at
    # 95..1500 ms.  Round 3 isolated the trigger -- DESIGN.md section 6: torch's own clip_grad_norm_ + torch.optim.SGD launches in
    # that situation; with the fused optimizer step this bench uses, both orders run at 16-20 ms there and inference-first is
    # the faster one -- so the special order is gone.)
    for half in ('infer', 'train'):
        if half == 'infer' and args.mode in ('both', 'infer'):
            bench_infer()
        if half == 'train' and args.mode in ('both', 'train'):
            bench_train()

    if rank == 0:
        cpu = parity = None
        if joined == 1 and not args.no_cpu_baseline and args.mode != 'train':
            cpu, parity = cpu_baseline_and_parity(cfg, sd, B, hip_pred, hip_det, train_probe)
        result.pop('_keep', None)
        layer_detail = {m: r.pop('layers', None) for m, r in result.items()}
        head = result['infer'] if 'infer' in result else result['train']
        what = 'inference' if 'infer' in result else 'training'
        line = {
            'metric': f'images/sec {net} 1248x384 bs={B} {what}' + (' (+ training under "train")' if args.mode == 'both' else ''),
            'value': head['value'], 'unit': 'images/sec', 'n_gpus': joined, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': head['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': head['workload'], 'arch': args.arch, 'images_per_gpu_per_step': B, 'global_batch': B * joined,
                       'input': '3x384x1248 fp32 NCHW, HBM resident', 'weights': 'synthetic Kaiming-scale, seed 1234',
                       'parallelism': (f'replicas x{joined}' if what == 'inference' else f'dp{joined}'),
                       **({'steps_in_flight': head['steps_in_flight']} if 'steps_in_flight' in head else {})},
            'roofline': head['roofline'], 'cpu_baseline': cpu, 'parity': parity,
            'whole_network': head.get('whole_network'), 'timed_with': head['timed_with'],
            'repeat_window_ms_per_step': head['repeat_window_ms_per_step'],
            **({k: head[k] for k in head if k in ('steps_in_flight', 'serial_ms_per_step', 'value_serial', 'degraded') or k.startswith('value_inflight')}),
            'layer_families': head['layer_families'], 'kernels_event_profile': head['kernels_event_profile'],
        }
        if 'eager_steps_launched' in head:
            line['eager_steps_launched'] = head['eager_steps_launched']
        if head.get('pipeline') is not None:
            line['pipeline'] = head['pipeline']
        if head.get('detect_dataset') is not None:
            line['detect_dataset'] = head['detect_dataset']
        if args.layers:
            line['layers'] = layer_detail
        if args.mode == 'both':
            line['train'] = result['train']
        print(json.dumps(line))
        if parity is not None and not parity['ok']:
            print(f'[bench] PARITY FAILED: {parity}', file=sys.stderr)
            if dist is not None:
                dist.destroy_process_group()
            sys.exit(3)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
