"""Fused Winograd Fire expand vs expand1x1 + Winograd expand3x3 (table configurations), per Fire shape at bs=20 (isolated)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from squeezedet_pytorch_amd import ops
B = int(os.environ.get('BATCH', 20)); ITERS = 30
def timeit(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS * 1e3
for (C, E, H, W) in [(16, 64, 96, 312)]:
    torch.manual_seed(0)
    x = torch.randn(B, H, W, C, device='cuda').relu_()
    w1 = torch.randn(E, C, 1, 1, device='cuda') * 0.1; b1 = torch.randn(E, device='cuda') * 0.1
    w3 = torch.randn(E, C, 3, 3, device='cuda') * 0.05; b3 = torch.randn(E, device='cuda') * 0.1
    npix = B * H * W
    y = torch.empty(B, H, W, 2 * E, device='cuda')
    p1 = ops.ConvPlan(w1, b1, ops.choose_cfg(1, C, E, npix)); p3 = ops.WinoPlan(w3, b3, ops.choose_wino_cfg(C, E, npix))
    t1 = timeit(lambda: ops.conv(x, 0, p1, y, 0, relu=True)); t3 = timeit(lambda: ops.conv_wino(x, 0, p3, y, E, relu=True))
    line = f'C{C} E{E} {H}x{W}: separate {t1:.1f} + {t3:.1f} = {t1 + t3:.1f} us |'
    for cid in (12, 6, 1006, 10, 1010, 8, 4):
        if not ops.fire_wino_cfg_ok(cid, C, E, E): continue
        fp = ops.FireWinoPlan(w1, b1, w3, b3, cid)
        line += f' x{cid} {timeit(lambda: ops.fire_wino(x, 0, fp, y, 0, E)):.1f}'
    print(line, flush=True)
