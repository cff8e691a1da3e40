"""Time Winograd configurations against each other on the bs=20 layer shapes, forward and data-gradient orientation, and
check that every configuration reproduces the first one's output bit for bit (same arithmetic, different staging)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from squeezedet_pytorch_amd import ops
B = int(os.environ.get('BATCH', 20))
ITERS = int(os.environ.get('ITERS', 30))
shapes = [(16, 64, 96, 312), (32, 128, 48, 156), (48, 192, 24, 78), (64, 256, 24, 78), (96, 384, 24, 78), (768, 72, 24, 78),
          (64, 16, 96, 312), (128, 32, 48, 156), (192, 48, 24, 78), (256, 64, 24, 78), (384, 96, 24, 78), (72, 768, 24, 78)]
if os.environ.get('ARCH') == 'plus':
    shapes = [(96, 64, 96, 312), (192, 128, 96, 312), (192, 128, 48, 156), (288, 192, 48, 156), (384, 256, 48, 156), (384, 256, 24, 78), (512, 72, 24, 78)]
cfgs = [int(c) for c in os.environ.get('WCFGS', '2,10,1002,1010,8,3,11').split(',')]

def timeit(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS * 1e3

print('shape'.ljust(22) + ''.join(f'cfg{c}'.rjust(10) for c in cfgs) + '   table', flush=True)
for (C, N, H, W) in shapes:
    torch.manual_seed(0)
    x = torch.randn(B, H, W, C, device='cuda').relu_()
    w = torch.randn(N, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(N, device='cuda') * 0.1
    ref = None
    line = f'C{C}->N{N} {H}x{W}'.ljust(22)
    for c in cfgs:
        if not ops.wino_cfg_ok(c, C):
            line += '        - '
            continue
        plan = ops.WinoPlan(w, b, c)
        y = torch.empty(B, H, W, N, device='cuda')
        t = timeit(lambda: ops.conv_wino(x, 0, plan, y, 0, relu=True))
        if ref is None: ref = y.clone()
        ok = torch.equal(y, ref)
        line += f'{t:9.1f}{" " if ok else "!"}'
    line += f'   {ops.choose_wino_cfg(C, N, B * H * W)}'
    print(line, flush=True)
