import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from squeezedet_pytorch_amd import ops, tiles
def run(C, N, B, H, W, pe, seed):
    torch.manual_seed(seed)
    x = torch.randn(B, H, W, C + pe, device='cuda')
    w = torch.randn(N, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(N, device='cuda') * 0.1
    xo = pe // 2 // 4 * 4
    p2 = ops.WinoPlan(w, b, 2); p17 = ops.WinoPlan(w, b, tiles.WINO_VS_CFG)
    y2 = torch.zeros(B, H, W, N, device='cuda'); y17 = torch.zeros(B, H, W, N, device='cuda')
    ops.conv_wino(x, xo, p2, y2, 0, relu=False)
    for rep in range(3):
        y17.zero_()
        ops.conv_wino(x, xo, p17, y17, 0, relu=False)
        torch.cuda.synchronize()
        bad = (y2 != y17)
        if bad.any():
            idx = bad.nonzero()
            bs, hs, ws, cs = [sorted(set(idx[:, k].tolist())) for k in range(4)]
            print(f'  C{C} N{N} B{B} {H}x{W} pe{pe} rep{rep}: {int(bad.sum())} bad; b {bs} rows {hs} cols {ws[:6]}..{ws[-3:]} ch {cs[:4]}..{cs[-3:]} maxdiff {(y2-y17).abs().max().item():.3e}')
        else:
            print(f'  C{C} N{N} B{B} {H}x{W} pe{pe} rep{rep}: ok')
for args in [(48, 72, 2, 6, 20, 16, 1), (48, 72, 2, 6, 20, 0, 1), (48, 72, 2, 6, 20, 16, 2), (48, 72, 2, 8, 32, 16, 1), (16, 72, 2, 6, 20, 16, 1), (96, 72, 2, 6, 20, 16, 1), (48, 72, 1, 4, 16, 16, 1), (768, 72, 2, 24, 78, 16, 1)]:
    run(*args)
