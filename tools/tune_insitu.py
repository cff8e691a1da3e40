#!/usr/bin/env python
"""In-situ refinement of tuning.json: the per-layer table of tools/tune_conv.py is measured with every layer in a loop of
its own, which is not how the layer runs inside the network (cold L2 / Infinity Cache state, neighbours' tails, clocks):
round 2 found configurations that win by 7-12 % in isolation and lose 1 % in the step.  This tool times the WHOLE
inference step (hipGraph replays of backbone + detect, the thing bench.py reports) and walks the layers one at a time
(coordinate descent), trying for each the best few configurations of the isolated measurement, and keeps a change only
if the step gets faster by more than the noise floor.  Run on the GPU box:

    python tools/tune_insitu.py [--arch squeezedet] [--batch 20] [--top 6] [--out gpurun_out/tuning_insitu.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import squeezedet_pytorch_amd as sqd  # noqa: E402
from squeezedet_pytorch_amd import ops, plan as planmod, synthetic  # noqa: E402
from squeezedet_pytorch_amd.detector import Detector  # noqa: E402
from squeezedet_pytorch_amd.model import SqueezeDet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='squeezedet')
    ap.add_argument('--batch', type=int, default=0)
    ap.add_argument('--top', type=int, default=6, help='candidates per layer (fastest of the isolated measurement)')
    ap.add_argument('--ws', action='store_true', help='also try every applicable weight-stationary 1x1 configuration (conv_ws, caps 0..2) on the 1x1 rows')
    ap.add_argument('--mode', default='infer', choices=('infer', 'train'),
                    help="train: time the TRAINING step (fwd + loss + bwd + clip + SGD, hipGraph) and walk the rows only it consults "
                         "(data-gradient orientations, measured in isolation WITHOUT their ReLU-mask / accumulate epilogues, and the forward "
                         "rows the inference step no longer uses)")
    ap.add_argument('--splits', action='store_true', help='(train) also walk the split-K factors of the weight-gradient kernels (G: rows)')
    ap.add_argument('--rows', type=int, default=1, help='(train) 0 = skip the conv-configuration rows')
    ap.add_argument('--replays', type=int, default=40)
    ap.add_argument('--passes', type=int, default=1)
    ap.add_argument('--gain', type=float, default=0.0015, help='relative step-time gain a change must show (noise floor)')
    ap.add_argument('--table', default=os.path.join(ROOT, 'squeezedet-pytorch_amd', 'tuning.json'))
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'tuning_insitu.json'))
    args = ap.parse_args()
    B = args.batch or (16 if args.arch == 'squeezedetplus' else 20)
    full = json.load(open(args.table))
    cfg = sqd.make_cfg(arch=args.arch, device='cuda')
    model = SqueezeDet(cfg)
    model.load_state_dict(synthetic.make_state_dict(args.arch, seed=1234))
    det = Detector(model, cfg)
    x = synthetic.make_images(B, cfg.input_size, seed=0).cuda()
    bufs = ops._det_buffers(B, cfg.keep_top_k, x.device, cfg.num_anchors)
    tab = ops._tuning()                          # live {key: cfg} table the choosers read; edited in place
    train_keys = []
    if args.mode == 'train':
        from squeezedet_pytorch_amd.model import SqueezeDetWithLoss
        tmodel = SqueezeDetWithLoss(cfg)
        tmodel.load_state_dict(synthetic.make_state_dict(args.arch, seed=1234))
        tmodel = tmodel.cuda().train()
        params = [p for p in tmodel.parameters() if p.requires_grad]
        opt = torch.optim.SGD(params, lr=cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
        batch = {'image': x, 'gt': synthetic.make_gt(B, cfg.anchors, cfg.input_size, cfg.num_classes, seed=1).cuda()}

        def train_step():
            loss, _ = tmodel(batch)
            loss = loss.mean()
            opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, cfg.grad_norm)
            opt.step()

        # the rows the step consults: record the choosers' exact hits during one eager step
        real_choose, real_wino = ops.choose_cfg, ops.choose_wino_cfg

        def rec_choose(taps, C, N, npix, *a, **kw):
            k = f'{taps}:{C}:{N}:{npix}'
            if k in full and k not in train_keys:
                train_keys.append(k)
            return real_choose(taps, C, N, npix, *a, **kw)

        def rec_wino(C, N, npix, *a, **kw):
            k = f'W:{C}:{N}:{npix}'
            if k in full and k not in train_keys:
                train_keys.append(k)
            return real_wino(C, N, npix, *a, **kw)
        wgrad_keys = {}                              # 'G:taps:N:C:npix' -> split the heuristic (or the table) gives today
        real_split = ops.wgrad_split

        def rec_split(N, C, taps, Bq, H, W, wino=None, **kw):
            S, stride = real_split(N, C, taps, Bq, H, W, wino, **kw)
            wgrad_keys.setdefault(f'G:{taps}:{N}:{C}:{Bq * H * W}', S)
            return S, stride
        ops.choose_cfg, ops.choose_wino_cfg, ops.wgrad_split = rec_choose, rec_wino, rec_split
        train_step()
        torch.cuda.synchronize()
        ops.choose_cfg, ops.choose_wino_cfg, ops.wgrad_split = real_choose, real_wino, real_split

    def run_step():
        if args.mode == 'train':
            train_step()
        else:
            det.detect_device(x, out=bufs)

    def step_ms():
        """Median hipGraph replay time of the whole step with the current table."""
        (tmodel.base if args.mode == 'train' else model.base).invalidate_plans()
        if args.mode == 'train':
            tmodel.base._wgrad_batches.clear()          # (slab workspaces are sized by the splits)
        for _ in range(2):
            run_step()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run_step()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                run_step()
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        times = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.replays):
                g.replay()
            e1.record(); torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / args.replays)
        del g
        return sorted(times)[1]

    # the table keys this workload actually consults, in launch order
    keys = []
    for name, tag in planmod.inference_launch_plan(args.arch, B, cfg.input_size):
        import re
        m = re.match(r'(\d)tap C(\d+) N(\d+) (\d+)x(\d+)', tag)
        if not m:
            continue
        taps, C, N, h, w = (int(v) for v in m.groups())
        k = f'W:{C}:{N}:{B * h * w}' if taps == 9 else f'{taps}:{C}:{N}:{B * h * w}'
        if k in full and k not in keys:
            keys.append(k)
    if args.mode == 'train':
        keys = [k for k in train_keys if k not in keys]          # leave the inference step's rows alone
    base = step_ms()
    print(f'start: {base:.4f} ms/step ({B / base * 1e3:.0f} img/s), {len(keys)} table rows in play', flush=True)
    changed = {}
    for p in range(args.passes):
        for k in (keys if args.rows else []):
            cands = sorted(full[k].get('all', {}).items(), key=lambda kv: kv[1])[:args.top]
            cands = [int(c) for c, _ in cands if int(c) != tab[k]]
            if args.ws and k.startswith('1:'):
                Ck, Nk = int(k.split(':')[1]), int(k.split(':')[2])
                ct = ops.cfg_table()
                for c in ct:
                    bn = ct[c][3]
                    if ops._CFG_DMA[c] >= 3 and ops.conv_cfg_ok(c, Ck) and not (-(-Nk // bn) * bn > 2 * Nk and bn > 16):
                        cands += [c + 1000 * cap for cap in (0, 1, 2) if c + 1000 * cap != tab[k] and c + 1000 * cap not in cands]
            if k.startswith('W:'):
                cands = [c for c in cands if c % 1000 < 4] + [c for c in cands if c % 1000 >= 4][:1]
            best_c, best_t = tab[k], base
            keep = tab[k]
            for c in cands:
                tab[k] = c
                try:
                    t = step_ms()
                except Exception as e:  # noqa: BLE001
                    print('  skip', k, c, e)
                    continue
                if t < best_t * (1.0 - args.gain):
                    best_c, best_t = c, t
            tab[k] = best_c
            print(f'  .. {k}: {len(cands)} candidates, best {best_c} ({best_t:.4f} ms vs {base:.4f})', flush=True)
            if best_c != keep:
                t2 = step_ms()                              # confirm against a fresh measurement of the incumbent
                tab[k] = keep
                t1 = step_ms()
                if t2 < t1 * (1.0 - args.gain / 2):
                    tab[k] = best_c
                    changed[k] = (keep, best_c, t1, t2)
                    base = t2
                    print(f'  {k:24s} cfg {keep} -> {best_c}: step {t1:.4f} -> {t2:.4f} ms', flush=True)
                else:
                    base = t1
            else:
                base = min(base, best_t) if best_t else base
        print(f'pass {p}: {base:.4f} ms/step ({B / base * 1e3:.0f} img/s)', flush=True)
    # ---- split-K factors of the weight-gradient kernels (training mode): each layer's split tried inside the step ----
    if args.mode == 'train' and args.splits:
        for k, s0 in wgrad_keys.items():
            cands = sorted({max(1, int(round(s0 * f))) for f in (0.5, 0.67, 0.8, 1.25, 1.5, 2.0)} - {s0})
            best_c, best_t = s0, base
            for c in cands:
                tab[k] = c
                t = step_ms()
                if t < best_t * (1.0 - args.gain):
                    best_c, best_t = c, t
            tab.pop(k, None)
            print(f'  .. {k}: split {s0}, best {best_c} ({best_t:.4f} ms vs {base:.4f})', flush=True)
            if best_c != s0:
                tab[k] = best_c
                t2 = step_ms()
                tab.pop(k, None)
                t1 = step_ms()
                if t2 < t1 * (1.0 - args.gain / 2):
                    tab[k] = best_c
                    full[k] = {'cfg': best_c, 'us': round(t2 * 1e3, 1), 'separate_us': round(t1 * 1e3, 1),
                               'note': f'split-K slabs of this weight gradient (heuristic {s0}); WHOLE training step (us) with / without, tools/tune_insitu.py --mode train --splits'}
                    base = t2
                    print(f'  {k:24s} split {s0} -> {best_c}: step {t1:.4f} -> {t2:.4f} ms', flush=True)
                else:
                    base = t1
    for k, (old, new, t1, t2) in changed.items():
        full[k]['cfg'] = new
        full[k]['insitu'] = {'from': old, 'step_ms_before': round(t1, 4), 'step_ms_after': round(t2, 4)}
        if k.startswith('W:'):
            full[k]['direct_us'] = max(full[k].get('direct_us', 0), full[k]['us'] + 1.0)    # keep the Winograd row selected
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(full, open(args.out, 'w'), indent=1, sort_keys=True)
    print(f'final: {step_ms():.4f} ms/step; {len(changed)} rows changed; wrote {args.out}')


if __name__ == '__main__':
    main()
