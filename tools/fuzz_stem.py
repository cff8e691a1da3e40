"""Randomised parity sweep of the kernels added in round 3's second session against torch (CPU fp32):
  * stem_pool (inference: the wave kernels 2 / 3 / 4 and the workgroup kernel 0; training: arg-max codes) on random image sizes
    (heights from 5, widths that are multiples of 4 from 8: maps smaller than a tile, partial tiles, all four borders),
  * stem_wgrad_pooled (the gather kernel) vs autograd,
  * stem_pool_squeeze (stem + the first Fire's squeeze),
usage: fuzz_stem.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, torch.nn.functional as F
from squeezedet_pytorch_amd import ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous()
def fail(what, **kw):
    print('MISMATCH', what, kw); sys.exit(1)
t0 = time.time(); n = {'stem': 0, 'stem_argmax': 0, 'gather': 0, 'stem_sq': 0}; t_say = t0; worst = 0.0
w = (torch.randn(64, 3, 3, 3) * 0.25); b = torch.randn(64) * 0.2
ws = torch.randn(16, 64, 1, 1) * 0.2; bs = torch.randn(16) * 0.1
while time.time() - t0 < budget:
    if time.time() - t_say > 30:
        print(f'  .. {n} after {time.time() - t0:.0f} s', flush=True); t_say = time.time()
    B = int(rs.randint(1, 4)); H = int(rs.randint(5, 140)); W = 4 * int(rs.randint(2, 90))
    x = torch.randn(B, 3, H, W) * float(rs.choice([0.5, 1.0, 3.0]))
    if rs.rand() < 0.3:
        w = torch.randn(64, 3, 3, 3) * 0.25; b = torch.randn(64) * float(rs.choice([0.0, 0.2, 1.0]))
    xr = x.clone().requires_grad_(False); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    conv = F.relu(F.conv2d(xr, wr, br, stride=2, padding=1))
    if conv.shape[2] < 3 or conv.shape[3] < 3:
        continue
    ref = F.max_pool2d(conv, 3, 2, ceil_mode=True)
    refn = nhwc(ref.detach())
    tol = 2e-5 * max(1.0, refn.abs().max().item()) + 1e-5
    xg, wg, bg = x.cuda(), w.cuda(), b.cuda()
    os.environ['SQD_STEM_WAVE'] = '0'
    y0 = ops.stem_pool(xg, wg, bg)
    for v in ('2', '3', '4'):
        os.environ['SQD_STEM_WAVE'] = v
        y = ops.stem_pool(xg, wg, bg)
        err = (y.cpu() - refn).abs().max().item()
        if not (err <= tol and torch.equal(y, y0)):
            fail('stem_pool', variant=v, B=B, H=H, W=W, err=err, tol=tol, equal=torch.equal(y, y0))
        worst = max(worst, err / tol); n['stem'] += 1
    # training forward: values and codes equal to the workgroup kernel's, bit for bit
    os.environ['SQD_STEM_WAVE'] = '0'
    am0 = torch.full(tuple(y0.shape), 77, dtype=torch.uint8, device='cuda'); yt0 = ops.stem_pool(xg, wg, bg, argmax=am0)
    os.environ['SQD_STEM_WAVE'] = '2'
    am = torch.full(tuple(y0.shape), 77, dtype=torch.uint8, device='cuda'); yt = ops.stem_pool(xg, wg, bg, argmax=am)
    if not (torch.equal(yt, yt0) and torch.equal(am, am0) and torch.equal(yt, y0)):
        fail('stem_pool argmax', B=B, H=H, W=W, values=torch.equal(yt, yt0), codes=int((am != am0).sum()))
    n['stem_argmax'] += 1
    # weight gradient through the codes (gather kernel) vs autograd
    dy = torch.randn_like(ref)
    ref.backward(dy)
    # (vs the dense kernel on the SAME codes at rounding level; vs autograd only loosely: where two window elements differ by less than
    #  the convolution's rounding noise the reference's own arg-max may sit elsewhere, which moves one dy * patch term -- both
    #  kernels then differ from autograd by the same amount)
    os.environ['SQD_STEM_WGRAD_GATHER'] = '0'
    dwd, dbd = ops.stem_wgrad_pooled(nhwc(dy).cuda(), None, am, xg, 64, 3)
    os.environ['SQD_STEM_WGRAD_GATHER'] = '1'
    dw, db = ops.stem_wgrad_pooled(nhwc(dy).cuda(), None, am, xg, 64, 3)
    gmax = max(1.0, float(wr.grad.abs().max()))
    ed = (dw - dwd).abs().max().item() / gmax
    ew = (dw.cpu() - wr.grad).abs().max().item() / gmax
    eb = (db.cpu() - br.grad).abs().max().item() / max(1.0, float(br.grad.abs().max()))
    flips = ew > 2e-4 and (dwd.cpu() - wr.grad).abs().max().item() / gmax > 2e-4       # the dense kernel disagrees with autograd the same way
    # (a pooled value within rounding of zero: the ReLU is active in one summation order and not in the other, and one dy term enters
    #  or leaves the bias gradient -- again in both kernels alike, since they read the same codes)
    bmax = max(1.0, float(br.grad.abs().max()))
    edb = (db - dbd).abs().max().item() / bmax
    flips_b = eb > 2e-4 and (dbd.cpu() - br.grad).abs().max().item() / bmax > 2e-4
    if not (ed <= 1e-5 and edb <= 1e-5 and (eb <= 2e-4 or flips_b) and (ew <= 2e-4 or flips)):
        fail('stem_wgrad gather', B=B, H=H, W=W, ed=ed, edb=edb, ew=ew, eb=eb, flips=flips, flips_b=flips_b)
    n['gather'] += 1
    # stem + first squeeze
    sq_ref = nhwc(F.relu(F.conv2d(ref.detach(), ws, bs)))
    ysq = ops.stem_pool_squeeze(xg, wg, bg, ws.cuda(), bs.cuda())
    err = (ysq.cpu() - sq_ref).abs().max().item()
    tol2 = 2e-5 * max(1.0, sq_ref.abs().max().item()) + 1e-5
    if not err <= tol2:
        fail('stem_pool_squeeze', B=B, H=H, W=W, err=err, tol=tol2)
    n['stem_sq'] += 1
os.environ.pop('SQD_STEM_WAVE', None); os.environ.pop('SQD_STEM_WGRAD_GATHER', None)
print(f'{n} cases ok in {time.time() - t0:.0f} s; worst stem error {worst:.2f} of the tolerance')
