"""Correctness probe of the deep-prefetch Winograd kernel (cfg 4..7) vs the base kernel (bitwise) on a few shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from squeezedet_pytorch_amd import ops
for (B, C, N, H, W) in [(2, 16, 64, 12, 20), (20, 16, 64, 96, 312), (20, 96, 384, 24, 78), (20, 64, 256, 24, 78), (20, 768, 72, 24, 78)]:
    torch.manual_seed(0)
    x = torch.randn(B, H, W, C, device='cuda'); w = torch.randn(N, C, 3, 3, device='cuda') * 0.1; b = torch.randn(N, device='cuda')
    outs = {}
    for c in (2, 6, 3, 7, 0, 4, 1006):
        y = torch.full((B, H, W, N), float('nan'), device='cuda')
        ops.conv_wino(x, 0, ops.WinoPlan(w, b, c), y, 0, relu=False)
        torch.cuda.synchronize()
        outs[c] = y
    msg = f'B{B} C{C} N{N} {H}x{W}:'
    for base, pipe in ((2, 6), (2, 1006), (0, 4)):
        d = (outs[pipe] - outs[base]).abs()
        bad = (d > 0) | torch.isnan(outs[pipe])
        msg += f'  cfg{pipe}: {int(bad.sum())} bad of {bad.numel()}'
        if bad.any():
            idx = bad.nonzero()
            i0 = tuple(idx[0].tolist())
            msg += f' (first {list(i0)} got {outs[pipe][i0].item():.4f} want {outs[base][i0].item():.4f}; x there {x[i0[0], i0[1], i0[2], :4].tolist()}, rows {sorted(set(idx[:,1].tolist()))[:8]}, cols {sorted(set(idx[:,2].tolist()))[:20]}, ch {sorted(set(idx[:,3].tolist()))[:6]}..)'
    print(msg[:300], flush=True)
    for base, pipe in ((2, 6),):
        bad = ((outs[pipe] - outs[base]).abs() > 0) | torch.isnan(outs[pipe])
        idx = bad.nonzero().cpu()
        if len(idx):
            import collections
            yy, xx, ch = idx[:, 1] % 4, idx[:, 2] % 16, idx[:, 3]
            print('   in-group row histogram', dict(collections.Counter(yy.tolist())), 'in-group col histogram', dict(sorted(collections.Counter(xx.tolist()).items())))
            print('   channel%32 histogram', dict(sorted(collections.Counter((ch % 32).tolist()).items())))
            grp = collections.Counter(zip(idx[:, 0].tolist(), (idx[:, 1] // 4).tolist(), (idx[:, 2] // 16).tolist(), (ch // 32).tolist()))
            print('   bad (image, row group, col group, slice) -> count:', dict(list(grp.items())[:12]), 'n groups', len(grp))
