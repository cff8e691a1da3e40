import os, sys
sys.path.insert(0, '.')
import torch
from squeezedet_pytorch_amd import ops
torch.manual_seed(0)
w = torch.randn(64, 3, 3, 3, device='cuda') * 0.2; b = torch.randn(64, device='cuda') * 0.1
x = torch.randn(1, 3, 64, 128, device='cuda')
os.environ['SQD_STEM_WAVE'] = '0'
am0 = torch.full((1, 16, 32, 64), 77, dtype=torch.uint8, device='cuda'); y0 = ops.stem_pool(x, w, b, argmax=am0)
os.environ['SQD_STEM_WAVE'] = '2'
am2 = torch.full((1, 16, 32, 64), 77, dtype=torch.uint8, device='cuda'); y2 = ops.stem_pool(x, w, b, argmax=am2)
d = (y0 != y2)[0]
print('pooled mismatches', int(d.sum()), 'by channel%16', d.sum((0, 1)).view(4, 16).sum(0).tolist())
print('by channel//16', d.sum((0, 1)).view(4, 16).sum(1).tolist())
print('by row', d.sum((1, 2)).tolist())
print('by col', d.sum((0, 2)).tolist())
dc = (am0 != am2)[0]
print('code mismatches', int(dc.sum()), 'by col', dc.sum((0, 2)).tolist())
print('by row', dc.sum((1, 2)).tolist())
idx = d.nonzero()[:8]
for r, c, n in idx.tolist(): print(r, c, n, float(y0[0, r, c, n]), float(y2[0, r, c, n]), int(am0[0, r, c, n]), int(am2[0, r, c, n]))
