"""Winograd vs direct 3x3 weight-gradient kernels on the SqueezeDet layer shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from squeezedet_pytorch_amd import ops
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B = int(os.environ.get('BATCH', 20))
for (C, N, H, W) in [(16, 64, 96, 312), (32, 128, 48, 156), (48, 192, 24, 78), (64, 256, 24, 78), (96, 384, 24, 78), (768, 72, 24, 78)]:
    x = torch.randn(B, H, W, C, device='cuda'); dy = torch.randn(B, H, W, 2 * N, device='cuda')
    ops.tiles._TARGET_WGS_WINO = int(os.environ.get('WTARGET', 512))
    td = timeit(lambda: ops.conv_wgrad(dy, N, N, x, 0, C, 9, wino=False))
    tw = timeit(lambda: ops.conv_wgrad(dy, N, N, x, 0, C, 9, wino=True))
    a = ops.conv_wgrad(dy, N, N, x, 0, C, 9, wino=False); b = ops.conv_wgrad(dy, N, N, x, 0, C, 9, wino=True)
    gf = 2.0 * B * H * W * N * C * 9 / 1e6
    S, _ = ops.wgrad_split(N, C, 9, B, H, W)
    print(f'C{C:3d} N{N:4d} {H}x{W} S={S}: direct {td:7.1f} us ({gf / td:6.1f} TF/s)  winograd {tw:7.1f} us ({gf / tw:6.1f} eff TF/s)  '
          f'relerr {(a[0] - b[0]).abs().max().item() / a[0].abs().max().item():.1e} bias {(a[1] - b[1]).abs().max().item() / a[1].abs().max().item():.1e}', flush=True)
