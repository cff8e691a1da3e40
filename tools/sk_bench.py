"""Isolated timing of the balanced Winograd kernel (cfg 16) against the unit kernel (cfg 2) on the 24x78 bs=20 layer shapes, over
schedule variants (tools replace ``plans.wino_sk_params`` for the sweep).  usage: python tools/sk_bench.py [variant ...]   variant = ks,hb,minseg"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from squeezedet_pytorch_amd import ops, plans

B = int(os.environ.get('BATCH', 20))
ITERS = int(os.environ.get('ITERS', 30))
shapes = [(768, 72, 24, 78), (72, 768, 24, 78), (96, 384, 24, 78), (384, 96, 24, 78), (48, 192, 24, 78), (64, 256, 24, 78)]
if os.environ.get('SHAPES'):
    shapes = [tuple(int(v) for v in s.split('x')) for s in os.environ['SHAPES'].split(',')]
variants = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(0, 1000, 2), (4, 1000, 2)]


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS * 1e3


print('shape'.ljust(22) + 'cfg2'.rjust(9) + ''.join(f'{v}'.rjust(16) for v in variants), flush=True)
for (C, N, H, W) in shapes:
    torch.manual_seed(0)
    x = torch.randn(B, H, W, C, device='cuda').relu_()
    w = torch.randn(N, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(N, device='cuda') * 0.1
    y = torch.empty(B, H, W, N, device='cuda')
    p2 = ops.WinoPlan(w, b, 2)
    line = f'C{C}->N{N} {H}x{W}'.ljust(22) + f'{timeit(lambda: ops.conv_wino(x, 0, p2, y, 0, relu=True)):9.1f}'
    ref = y.clone()
    p16 = ops.WinoPlan(w, b, ops.WINO_SK_CFG)
    for (ks, hb, ms) in variants:
        plans.wino_sk_params = (lambda N=None, C=None, _v=(ms, hb, ks): _v); plans._SK_SCHEDULES.clear()
        t = timeit(lambda: ops.conv_wino(x, 0, p16, y, 0, relu=True))
        err = (y - ref).abs().max().item()
        line += f'{t:12.1f}{" " if err < 1e-3 else "!"}   '
    print(line, flush=True)
