"""Training-forward stem: stem_wave_kernel<2,2,ARGMAX> (SQD_STEM_WAVE=2) vs the workgroup kernel (0): pooled values and codes equal; time."""
import os, sys
sys.path.insert(0, '.')
import torch
from squeezedet_pytorch_amd import ops
def run(v, fn):
    os.environ['SQD_STEM_WAVE'] = v
    return fn()
torch.manual_seed(0)
w = torch.randn(64, 3, 3, 3, device='cuda') * 0.2; b = torch.randn(64, device='cuda') * 0.1
ok = True
for (B, H, W) in ((2, 64, 96), (1, 52, 68), (2, 12, 16), (1, 9, 8), (1, 130, 1000), (3, 384, 1248)):
    x = torch.randn(B, 3, H, W, device='cuda')
    def f():
        y = ops.stem_pool(x, w, b)
        am = torch.full(tuple(y.shape), 77, dtype=torch.uint8, device='cuda')
        y2 = ops.stem_pool(x, w, b, argmax=am)
        return y, y2, am
    y0, y0t, am0 = run('0', f)
    y2, y2t, am2 = run('2', f)
    same = torch.equal(y0t, y2t) and torch.equal(am0, am2) and torch.equal(y2, y2t)
    ok = ok and same
    print(B, H, W, 'pooled equal', torch.equal(y0t, y2t), 'codes equal', torch.equal(am0, am2), 'mismatches', int((am0 != am2).sum()), 'inference == training values', torch.equal(y2, y2t))
x = torch.randn(20, 3, 384, 1248, device='cuda')
am = torch.empty(20, 96, 312, 64, dtype=torch.uint8, device='cuda')
for v in ('0', '2', '5'):
    def t():
        for _ in range(3): ops.stem_pool(x, w, b, argmax=am)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ops.stem_pool(x, w, b, argmax=am)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 30 * 1e3
    print('variant', v, f'{run(v, t):.1f} us')
sys.exit(0 if ok else 1)
