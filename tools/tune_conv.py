#!/usr/bin/env python
"""Measure every compiled conv tile configuration on every conv shape of the network (forward and
data-gradient orientation) on the current GPU and write the fastest per shape to
squeezedet-pytorch_amd/tuning.json (consulted by ops.choose_cfg).  Run on the GPU box:

    python tools/tune_conv.py [--arch squeezedet] [--batch 20] [--size 384 1248]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import squeezedet_pytorch_amd as sqd  # noqa: E402
from squeezedet_pytorch_amd import ops  # noqa: E402
from squeezedet_pytorch_amd.synthetic import layer_table, convdet_in_channels  # noqa: E402


def shapes(arch, B, H, W):
    """[(taps, C, N, h, w)] of every conv launched by forward + backward at this input size."""
    out = set()
    layers = layer_table(arch)
    h, w = ops.stem_out_size(H, W, layers[0][3])
    for l in layers[2:]:
        if l[0] == 'pool':
            h, w = ops.pool_out_size(h, w)
            continue
        _, cin, s, e1, e3 = l
        out.add((1, cin, s, h, w)); out.add((1, s, e1, h, w)); out.add((9, s, e3, h, w))
        out.add((1, s, cin, h, w)); out.add((1, e1, s, h, w)); out.add((9, e3, s, h, w))      # dgrads
    ncd = 9 * 8
    c = convdet_in_channels(arch)
    out.add((9, c, ncd, h, w)); out.add((9, ncd, c, h, w))
    return sorted(out)


def time_cfg(taps, C, N, B, h, w, cid, reps=10):
    k = 3 if taps == 9 else 1
    wt = torch.randn(N, C, k, k, device='cuda') * 0.05
    bias = torch.randn(N, device='cuda')
    plan = ops.ConvPlan(wt, bias, cid)
    x = torch.randn(B, h, w, C, device='cuda')
    y = torch.empty(B, h, w, N, device='cuda')
    for _ in range(2):
        ops.conv(x, 0, plan, y, 0, relu=True)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv(x, 0, plan, y, 0, relu=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def time_wino(C, N, B, h, w, cid, reps=10):
    wt = torch.randn(N, C, 3, 3, device='cuda') * 0.05
    bias = torch.randn(N, device='cuda')
    plan = ops.WinoPlan(wt, bias, cid)
    x = torch.randn(B, h, w, C, device='cuda')
    y = torch.empty(B, h, w, N, device='cuda')
    for _ in range(2):
        ops.conv_wino(x, 0, plan, y, 0, relu=True)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv_wino(x, 0, plan, y, 0, relu=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def fused_shapes(arch, B, H, W):
    """[(C, E, h, w)] of every Fire expand pair (inference forward, fused launch)."""
    out = set()
    layers = layer_table(arch)
    h, w = ops.stem_out_size(H, W, layers[0][3])
    for l in layers[2:]:
        if l[0] == 'pool':
            h, w = ops.pool_out_size(h, w)
            continue
        _, cin, s, e1, e3 = l
        if e1 == e3 and e1 % 16 == 0:
            out.add((s, e1, h, w))
    return sorted(out)


def time_fused(C, E, B, h, w, cid, reps=10):
    w1 = torch.randn(E, C, 1, 1, device='cuda') * 0.05; b1 = torch.randn(E, device='cuda')
    w3 = torch.randn(E, C, 3, 3, device='cuda') * 0.05; b3 = torch.randn(E, device='cuda')
    plan = ops.FusedExpandPlan(w1, b1, w3, b3, cid)
    x = torch.randn(B, h, w, C, device='cuda')
    y = torch.empty(B, h, w, 2 * E, device='cuda')
    for _ in range(2):
        ops.fire_expand(x, 0, plan, y, 0)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.fire_expand(x, 0, plan, y, 0)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='squeezedet')
    ap.add_argument('--batch', type=int, default=20)
    ap.add_argument('--size', type=int, nargs=2, default=[384, 1248])
    ap.add_argument('--out', default=os.path.join(ROOT, 'squeezedet-pytorch_amd', 'tuning.json'))
    ap.add_argument('--all-wino', action='store_true', help='also time the deep-prefetch Winograd ids 4..7 (isolated timing flatters them: profiles/r02f_ab_table_pipe.log)')
    ap.add_argument('--only', default='', help="'wino': keep the direct-kernel entries of the existing table, re-measure only the Winograd (W:) and fused-expand (F:) keys")
    args = ap.parse_args()
    B = args.batch
    table = {}
    if os.path.exists(args.out):
        table = json.load(open(args.out))
    tab = ops.cfg_table()
    print('configs:', {c: tab[c] + (ops.cfg_is_dma(c),) for c in tab})
    for taps, C, N, h, w in ([] if args.only == 'wino' else shapes(args.arch, B, *args.size)):
        res = []
        for cid, (t, kc, px, bn) in tab.items():
            if t != taps:
                continue
            if -(-N // bn) * bn > 2 * N and bn > 16:        # more than 2x channel padding: skip
                continue
            if not ops.conv_cfg_ok(cid, C):
                continue
            try:
                res.append((time_cfg(taps, C, N, B, h, w, cid), cid))
            except Exception as e:  # noqa: BLE001
                print('skip', (taps, C, N, h, w), cid, e)
        res.sort()
        # second pass: cap the persistent grid at k workgroups per CU (cfg + 1000 * k) for the three fastest tilings
        for _, cid in list(res[:3]):
            for cap in (1, 2, 3):
                try:
                    res.append((time_cfg(taps, C, N, B, h, w, cid + 1000 * cap), cid + 1000 * cap))
                except Exception as e:  # noqa: BLE001
                    print('skip', (taps, C, N, h, w), cid + 1000 * cap, e)
        res.sort()
        best_us, best = res[0]
        gf = 2.0 * B * h * w * N * C * taps / 1e9
        key = f'{taps}:{C}:{N}:{B * h * w}'
        table[key] = {'cfg': best, 'us': round(best_us, 1), 'tflops': round(gf / best_us * 1e-3 * 1e3 / 1e3 * 1e3, 1) if False else round(gf / (best_us * 1e-6) / 1e3, 1),
                      'all': {str(c): round(u, 1) for u, c in res}}
        print(f'{key:24s} best cfg {best:4d} {tab[best % 1000]}  {best_us:8.1f} us  {gf / (best_us * 1e-6) / 1e3:6.1f} TF/s   '
              + ' '.join(f'{c}:{u:.0f}' for u, c in res[:5]), flush=True)
    # Winograd F(2x2,3x3) form of every 3x3 shape (key W:C:N:npix), compared with the best direct configuration above
    for taps, C, N, h, w in shapes(args.arch, B, *args.size):
        if taps != 9 or C % 8:
            continue
        res = []
        for cid, (bn, wv) in ops.wino_cfgs().items():
            if -(-N // bn) * bn > 2 * N and bn > 16:
                continue
            if not ops.wino_cfg_ok(cid, C) or (4 <= cid < 8 and not args.all_wino):
                continue
            for cap in (0, 1):
                try:
                    res.append((time_wino(C, N, B, h, w, cid + 1000 * cap), cid + 1000 * cap))
                except Exception as e:  # noqa: BLE001
                    print('skip wino', (C, N, h, w), cid, e)
        if not res:
            continue
        res.sort()
        best_us, best = res[0]
        npix = B * h * w
        direct = table.get(f'9:{C}:{N}:{npix}', {}).get('us', 0)
        gf = 2.0 * npix * N * C * 9 / 1e9
        table[f'W:{C}:{N}:{npix}'] = {'cfg': best, 'us': round(best_us, 1), 'tflops_effective': round(gf / (best_us * 1e-6) / 1e3, 1),
                                     'direct_us': direct, 'all': {str(c): round(u, 1) for u, c in res}}
        print(f'W:{C}:{N}:{npix:<14d} best wino cfg {best:4d} {best_us:8.1f} us  {gf / (best_us * 1e-6) / 1e3:6.1f} eff TF/s  (direct {direct:.1f} us)  '
              + ' '.join(f'{c}:{u:.0f}' for u, c in res[:6]), flush=True)
    # fused Fire expand (key F:C:E:npix): compared with the sum of the two separate launches of the same layer
    for C, E, h, w in fused_shapes(args.arch, B, *args.size):
        res = []
        for cid in ops.fused_expand_cfgs(E):
            for cap in (0, 2, 3):
                try:
                    res.append((time_fused(C, E, B, h, w, cid + 1000 * cap), cid + 1000 * cap))
                except Exception as e:  # noqa: BLE001
                    print('skip fused', (C, E, h, w), cid, e)
        if not res:
            continue
        res.sort()
        best_us, best = res[0]
        npix = B * h * w
        d3 = table.get(f'9:{C}:{E}:{npix}', {}).get('us', 0)
        w3 = table.get(f'W:{C}:{E}:{npix}', {}).get('us', 0)
        sep = table.get(f'1:{C}:{E}:{npix}', {}).get('us', 0) + (min(d3, w3) if (d3 and w3) else (d3 or w3))
        gf = 2.0 * npix * E * C * 10 / 1e9
        table[f'F:{C}:{E}:{npix}'] = {'cfg': best, 'us': round(best_us, 1), 'tflops': round(gf / (best_us * 1e-6) / 1e3, 1),
                                     'separate_us': round(sep, 1), 'all': {str(c): round(u, 1) for u, c in res}}
        print(f'F:{C}:{E}:{npix:<14d} best cfg {best:4d} {tab[best % 1000]}  {best_us:8.1f} us  {gf / (best_us * 1e-6) / 1e3:6.1f} TF/s  (separate {sep:.1f} us)  '
              + ' '.join(f'{c}:{u:.0f}' for u, c in res[:6]), flush=True)
    json.dump(table, open(args.out, 'w'), indent=1, sort_keys=True)
    print('wrote', args.out)


if __name__ == '__main__':
    main()
