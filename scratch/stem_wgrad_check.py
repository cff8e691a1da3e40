"""Gather stem weight gradient (stem_wgrad_gather_kernel) vs the dense kernel (SQD_STEM_WGRAD_GATHER=0) vs CPU autograd; timing."""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch, torch.nn.functional as F
from squeezedet_pytorch_amd import ops

def run(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

def nhwc(t): return t.permute(0, 2, 3, 1).contiguous()
torch.manual_seed(0)
for (B, H, W) in ((2, 64, 96), (1, 52, 68), (2, 12, 16), (1, 9, 8), (1, 130, 1000)):
    x = torch.randn(B, 3, H, W); w = (torch.randn(64, 3, 3, 3) * 0.2).requires_grad_(True); b = (torch.randn(64) * 0.1).requires_grad_(True)
    y = F.max_pool2d(F.relu(F.conv2d(x, w, b, stride=2, padding=1)), 3, 2, ceil_mode=True)
    dy = torch.randn_like(y); y.backward(dy)
    am = torch.empty(*nhwc(y.detach()).shape, dtype=torch.uint8, device='cuda')
    pooled = ops.stem_pool(x.cuda(), w.detach().cuda(), b.detach().cuda(), argmax=am)
    res = {}
    for name, env in (('dense', {'SQD_STEM_WGRAD_GATHER': '0'}), ('gather2', {'SQD_STEM_WGRAD_GATHER': '1', 'SQD_STEM_GATHER_OCC': '2'}),
                      ('gather3', {'SQD_STEM_WGRAD_GATHER': '1', 'SQD_STEM_GATHER_OCC': '3'})):
        dw, db = run(env, lambda: ops.stem_wgrad_pooled(nhwc(dy).cuda(), None, am, x.cuda(), 64, 3))
        dwp, dbp = run(env, lambda: ops.stem_wgrad_pooled(nhwc(dy).cuda(), pooled, am, x.cuda(), 64, 3))
        res[name] = (dw, db)
        ew = (dw.cpu() - w.grad).abs().max().item() / max(1.0, float(w.grad.abs().max()))
        eb = (db.cpu() - b.grad).abs().max().item() / max(1.0, float(b.grad.abs().max()))
        print(f'{B}x{H}x{W} {name}: dW err {ew:.2e} db err {eb:.2e}  pooled-given == codes-only: {torch.equal(dw, dwp) and torch.equal(db, dbp)}')
    print('   gather2 == gather3:', torch.equal(res['gather2'][0], res['gather3'][0]))

x = torch.randn(20, 3, 384, 1248, device='cuda'); w = torch.randn(64, 3, 3, 3, device='cuda') * 0.2; b = torch.randn(64, device='cuda') * 0.1
am = torch.empty(20, 96, 312, 64, dtype=torch.uint8, device='cuda')
pooled = ops.stem_pool(x, w, b, argmax=am)
dp = torch.randn_like(pooled)
out = (torch.empty(64, 3, 3, 3, device='cuda'), torch.empty(64, device='cuda'))
ref = None
for name, env in (('dense', {'SQD_STEM_WGRAD_GATHER': '0'}), ('gather2', {'SQD_STEM_WGRAD_GATHER': '1', 'SQD_STEM_GATHER_OCC': '2'}),
                  ('gather3', {'SQD_STEM_WGRAD_GATHER': '1', 'SQD_STEM_GATHER_OCC': '3'})):
    def t():
        for _ in range(3): ops.stem_wgrad_pooled(dp, None, am, x, 64, 3, out=out)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.stem_wgrad_pooled(dp, None, am, x, 64, 3, out=out)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    us = run(env, t)
    dw = out[0].clone()
    if ref is None: ref = dw
    print(f'KITTI bs=20 {name}: {us:.1f} us (incl. slab reduction)  max |dW - dense| / max|dW| = {(dw - ref).abs().max().item() / ref.abs().max().item():.2e}')
