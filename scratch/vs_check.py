"""conv_wino_vs (cfg 17, csrc/conv_wino_vs.hip) against conv_wino<2,4> (bit for bit) and fp32 conv2d, then timed at the ConvDet shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from squeezedet_pytorch_amd import ops, tiles

def run(C, N, B, H, W, relu, pitch_extra=0):
    torch.manual_seed(C * 7 + N + H)
    x = torch.randn(B, H, W, C + pitch_extra, device='cuda')
    w = torch.randn(N, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(N, device='cuda') * 0.1
    xo = pitch_extra // 2 // 4 * 4
    p2 = ops.WinoPlan(w, b, 2); p17 = ops.WinoPlan(w, b, tiles.WINO_VS_CFG)
    y2 = torch.full((B, H, W, N + 8), 7.0, device='cuda'); y17 = torch.full((B, H, W, N + 8), 7.0, device='cuda')
    ops.conv_wino(x, xo, p2, y2, 4, relu=relu)
    ops.conv_wino(x, xo, p17, y17, 4, relu=relu)
    torch.cuda.synchronize()
    ref = F.conv2d(x[..., xo:xo + C].permute(0, 3, 1, 2).cpu(), w.cpu(), b.cpu(), padding=1)
    if relu: ref = ref.relu()
    ref = ref.permute(0, 2, 3, 1)
    err = (y17[..., 4:4 + N].cpu() - ref).abs().max().item()
    same = torch.equal(y2, y17)
    untouched = bool((y17[..., :4] == 7).all() and (y17[..., 4 + N:] == 7).all())
    print(f'C{C} N{N} B{B} {H}x{W} relu={relu}: bitwise==cfg2 {same}  max err vs conv2d {err:.2e}  window untouched {untouched}', flush=True)
    return same and err < 1e-4 * max(1.0, ref.abs().max().item()) and untouched

ok = True
for case in [(768, 72, 2, 24, 78, False), (512, 72, 1, 24, 78, False), (16, 72, 3, 5, 17, True), (8, 80, 1, 3, 3, True), (24, 20, 2, 7, 35, False),
             (64, 48, 5, 2, 2, True), (40, 4, 1, 9, 33, False), (96, 16, 2, 11, 50, True), (768, 72, 1, 1, 1, False), (32, 72, 7, 4, 16, True)]:
    ok = run(*case) and ok
ok = run(48, 72, 2, 6, 20, True, pitch_extra=16) and ok
print('ALL OK' if ok else 'FAILED', flush=True)
if not ok:
    sys.exit(1)

def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (C, N, B) in [(768, 72, 20), (512, 72, 16), (768, 72, 8), (768, 72, 40)]:
    torch.manual_seed(0)
    x = torch.randn(B, 24, 78, C, device='cuda').relu_()
    w = torch.randn(N, C, 3, 3, device='cuda') * 0.01; b = torch.randn(N, device='cuda')
    y = torch.empty(B, 24, 78, N, device='cuda')
    line = f'C{C}->N{N} bs={B}:'
    for cfg in (2, 1002, 3, tiles.WINO_SK_CFG, tiles.WINO_VS_CFG):
        plan = ops.WinoPlan(w, b, cfg)
        t = timeit(lambda: ops.conv_wino(x, 0, plan, y, 0))
        line += f'  cfg{cfg} {t:7.1f} us'
    print(line, flush=True)
