"""Winograd vs direct 3x3 kernels on the SqueezeDet layer shapes (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from squeezedet_pytorch_amd import ops

def timeit(fn, iters=None):
    iters = iters or ITERS
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

B = int(os.environ.get('BATCH', 20))
shapes = [(96, 384, 24, 78), (768, 72, 24, 78), (64, 256, 24, 78), (48, 192, 24, 78), (32, 128, 48, 156), (16, 64, 96, 312)]
if os.environ.get('SHAPE'):
    shapes = [shapes[int(os.environ['SHAPE'])]]
ITERS = int(os.environ.get('ITERS', 30))
WC = [int(v) for v in os.environ['WCFG'].split(',')] if os.environ.get('WCFG') else None
for (C, N, H, W) in shapes:
    torch.manual_seed(0)
    x = torch.randn(B, H, W, C, device='cuda').relu_()
    w = torch.randn(N, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(N, device='cuda') * 0.1
    npix = B * H * W
    cid = ops.choose_cfg(9, C, N, npix)
    plan = ops.ConvPlan(w, b, cid)
    y0 = torch.empty(B, H, W, N, device='cuda')
    t_direct = timeit(lambda: ops.conv(x, 0, plan, y0, 0, relu=True))
    gf = 2.0 * npix * N * C * 9 / 1e6      # flop per us = TF/s x 1e-6 ... (TF/s = gf / us)
    line = f'C{C:4d} N{N:4d} {H}x{W}: direct cfg {cid} {t_direct:7.1f} us {gf / t_direct:6.1f} TF/s |'
    for wc in (WC if WC is not None else ops.wino_cfgs()):
        for cap in ([0] if len(sys.argv) < 2 else [0, 1]):
            wp = ops.WinoPlan(w, b, wc + 1000 * cap)
            y1 = torch.empty(B, H, W, N, device='cuda')
            t = timeit(lambda: ops.conv_wino(x, 0, wp, y1, 0, relu=True))
            err = (y1 - y0).abs().max().item()
            line += f' w{wc}{"c" + str(cap) if cap else ""} {t:7.1f} us ({gf / t:5.1f} eff TF/s, err {err:.1e})'
    print(line, flush=True)
