"""Weight-stationary barrier-free 1x1 kernel (conv_ws) vs the table's configuration, every 1x1 layer shape at bs=20 (isolated)."""
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
shapes = [(64, 16, 96, 312), (16, 64, 96, 312), (128, 16, 96, 312), (128, 32, 48, 156), (32, 128, 48, 156), (256, 32, 48, 156),
          (256, 48, 24, 78), (48, 192, 24, 78), (384, 48, 24, 78), (384, 64, 24, 78), (64, 256, 24, 78), (512, 64, 24, 78),
          (512, 96, 24, 78), (96, 384, 24, 78), (768, 96, 24, 78)]
tab = ops.cfg_table()
ws = [c for c in tab if ops._CFG_DMA[c] >= 3]
print('shape'.ljust(22) + 'table'.rjust(14) + ''.join(f'{ops.cfg_kernel_name(c)[7:]}'.rjust(9) for c in ws) + '   best(+cap)')
for (C, N, H, W) in shapes:
    torch.manual_seed(0)
    x = torch.randn(B, H, W, C, device='cuda').relu_()
    w = torch.randn(N, C, 1, 1, device='cuda') * (2.0 / C) ** 0.5
    b = torch.randn(N, device='cuda') * 0.1
    y = torch.empty(B, H, W, N, device='cuda')
    c0 = ops.choose_cfg(1, C, N, B * H * W)
    p0 = ops.ConvPlan(w, b, c0)
    t0 = timeit(lambda: ops.conv(x, 0, p0, y, 0, relu=True)); ref = y.clone()
    line = f'C{C}->N{N} {H}x{W}'.ljust(22) + f'{c0}:{t0:.1f}'.rjust(14)
    best = (t0, c0)
    for c in ws:
        bn = tab[c][3]
        if not ops.conv_cfg_ok(c, C) or (-(-N // bn) * bn > 2 * N and bn > 16):
            line += '        -'; continue
        for cap in (0, 1, 2):
            if cap and cap != 1 and False: continue
            p = ops.ConvPlan(w, b, c + 1000 * cap)
            t = timeit(lambda: ops.conv(x, 0, p, y, 0, relu=True))
            ok = torch.equal(y, ref) or (y - ref).abs().max().item() < 1e-4
            if cap == 0: line += f'{t:8.1f}{" " if ok else "!"}'
            if ok and t < best[0]: best = (t, c + 1000 * cap)
    print(line + f'   {best[1]}:{best[0]:.1f}', flush=True)
