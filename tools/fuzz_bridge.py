"""Randomised parity sweep of the STORING forms of the fused launches (round 4: the training forward of the first Fires):
  * fire_bridge(save=): squeeze output bit-equal to the inference bridge, stored expand output bit-equal to the plain fused expand of
    the same kernel family (fire_wino cfg 12), both within tolerance of torch's fp32 modules;
  * fire_pool_bridge(save=, codes=): squeeze output bit-equal to the inference bridge; pooled tensor and arg-max / ReLU codes bit-equal to
    ops.maxpool(relu_codes=True) on that expand output (clipped windows, 1..many segments, carried rows, partial channel blocks);
  * stem_pool_squeeze(argmax=): pooled tensor + codes bit-equal to stem_pool(argmax=), squeeze output vs torch;
  * FireBridgePlan refresh in place == a fresh packing.
usage: fuzz_bridge.py [seconds] [seed]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, torch.nn.functional as F
from squeezedet_pytorch_amd import ops, plans

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous()
def fail(what, **kw):
    print('MISMATCH', what, kw); sys.exit(1)
t0 = time.time(); n = {'bridge': 0, 'poolbridge': 0, 'stem': 0, 'refresh': 0}; t_say = t0; worst = 0.0
while time.time() - t0 < budget:
    if time.time() - t_say > 30:
        print(f'  .. {n} after {time.time() - t0:.0f} s', flush=True); t_say = time.time()
    C = int(rs.choice([8, 16])); E1 = 16 * int(rs.randint(1, 5)); E3 = 4 * int(rs.randint(1, 17)); S = 4 * int(rs.randint(1, 9))
    B = int(rs.randint(1, 4)); H = int(rs.randint(3, 40)); W = int(rs.randint(3, 70))
    sc = float(rs.choice([0.3, 1.0, 3.0]))
    x = F.relu(torch.randn(B, C, H, W) * sc)
    w1 = torch.randn(E1, C, 1, 1) * (2.0 / C) ** 0.5; b1 = torch.randn(E1) * float(rs.choice([0.0, 0.1, 1.0]))
    w3 = torch.randn(E3, C, 3, 3) * (2.0 / (9 * C)) ** 0.5; b3 = torch.randn(E3) * float(rs.choice([0.0, 0.1, 1.0]))
    ws = torch.randn(S, E1 + E3, 1, 1) * (2.0 / (E1 + E3)) ** 0.5; bs = torch.randn(S) * 0.1
    mid = torch.cat([F.relu(F.conv2d(x, w1, b1)), F.relu(F.conv2d(x, w3, b3, padding=1))], 1)
    ref_sq = nhwc(F.relu(F.conv2d(mid, ws, bs)))
    midn = nhwc(mid)
    tol = 2e-5 * max(1.0, midn.abs().max().item(), ref_sq.abs().max().item()) + 1e-5
    xg = nhwc(x).cuda()
    g = [t.cuda() for t in (w1, b1, w3, b3, ws, bs)]
    # the expand output of this kernel family (plain fused expand)
    if not ops.fire_wino_cfg_ok(12, C, E1, E3):
        continue
    out = torch.empty(B, H, W, E1 + E3, device='cuda')
    ops.fire_wino(xg, 0, ops.FireWinoPlan(g[0], g[1], g[2], g[3], 12), out, 0, E1)
    if ops.fire_bridge_cfg_ok(12, C, E3, E1, S):
        plan = ops.FireBridgePlan(*g, 12)
        pad = 4 * int(rs.randint(0, 3))
        y = torch.full((B, H, W, S + pad), -7.0, device='cuda'); yi = torch.full((B, H, W, S + pad), -7.0, device='cuda')
        sv = torch.full((B, H, W, E1 + E3 + pad), -5.0, device='cuda')
        swap = bool(rs.rand() < 0.5)                               # expand3x3 window in front of the expand1x1 window
        c1, c3 = (E3 + pad, pad) if swap else (pad, pad + E1)
        c1 = min(c1, E3 + pad) if swap else c1
        sv = torch.full((B, H, W, E1 + E3 + pad), -5.0, device='cuda')
        ops.fire_bridge(xg, 0, plan, y, pad, save=sv, save_coff1=c1, save_coff3=c3)
        ops.fire_bridge(xg, 0, plan, yi, pad)
        torch.cuda.synchronize()
        e = (y[..., pad:pad + S].cpu() - ref_sq).abs().max().item()
        ok = (torch.equal(y, yi) and e <= tol and torch.equal(sv[..., c1:c1 + E1], out[..., :E1]) and torch.equal(sv[..., c3:c3 + E3], out[..., E1:])
              and bool((sv[..., :pad] == -5.0).all()))
        if not ok:
            fail('fire_bridge save', C=C, E1=E1, E3=E3, S=S, B=B, H=H, W=W, pad=pad, swap=swap, err=e, tol=tol)
        worst = max(worst, e / tol); n['bridge'] += 1
        # refresh in place
        g2 = [torch.randn_like(t) for t in g]
        fresh = ops.FireBridgePlan(*g2, 12)
        plans.refresh_bridge_plans([(plan, *g2)])
        torch.cuda.synchronize()
        if not all(torch.equal(getattr(plan, k), getattr(fresh, k)) for k in ('w', 'sq_ops', 'bias_tab', 'sq_bias')):
            fail('bridge plan refresh', C=C, E1=E1, E3=E3, S=S)
        n['refresh'] += 1
    if ops.fire_pool_bridge_ok(C, E3, E1, S):
        plan = ops.FireBridgePlan(*g, 12, pooled=True)
        Hp, Wp = ops.pool_out_size(H, W)
        nseg = int(rs.randint(1, 8))
        y = torch.full((B, Hp, Wp, S), -7.0, device='cuda'); yi = torch.full((B, Hp, Wp, S), -7.0, device='cuda')
        sv = torch.full((B, Hp, Wp, E1 + E3), -5.0, device='cuda'); cd = torch.full((B, Hp, Wp, E1 + E3), 99, dtype=torch.uint8, device='cuda')
        ops.fire_pool_bridge(xg, 0, plan, y, 0, nseg=nseg, save=sv, codes=cd)
        ops.fire_pool_bridge(xg, 0, plan, yi, 0, nseg=nseg)
        am = torch.empty(B, Hp, Wp, E1 + E3, dtype=torch.uint8, device='cuda')
        pooled = ops.maxpool(out, argmax=am, relu_codes=True)
        torch.cuda.synchronize()
        if not (torch.equal(y, yi) and torch.equal(sv, pooled) and torch.equal(cd, am)):
            fail('fire_pool_bridge save', C=C, E1=E1, E3=E3, S=S, B=B, H=H, W=W, nseg=nseg, sq=torch.equal(y, yi),
                 pooled=int((sv != pooled).sum()), codes=int((cd != am).sum()))
        n['poolbridge'] += 1
    # stem + squeeze, training form
    Hs = int(rs.randint(5, 120)); Wsz = 4 * int(rs.randint(2, 80))
    img = torch.randn(B, 3, Hs, Wsz) * sc
    wst = torch.randn(64, 3, 3, 3) * 0.25; bst = torch.randn(64) * float(rs.choice([0.0, 0.2, 1.0]))
    wq = torch.randn(16, 64, 1, 1) * 0.2; bq = torch.randn(16) * 0.1
    conv = F.relu(F.conv2d(img, wst, bst, stride=2, padding=1))
    if conv.shape[2] >= 3 and conv.shape[3] >= 3:
        pl = F.max_pool2d(conv, 3, 2, ceil_mode=True)
        rsq = nhwc(F.relu(F.conv2d(pl, wq, bq)))
        shp = tuple(nhwc(pl).shape)
        am0 = torch.full(shp, 77, dtype=torch.uint8, device='cuda'); am1 = torch.full(shp, 78, dtype=torch.uint8, device='cuda')
        p0 = ops.stem_pool(img.cuda(), wst.cuda(), bst.cuda(), argmax=am0)
        ysq, p1 = ops.stem_pool_squeeze(img.cuda(), wst.cuda(), bst.cuda(), wq.cuda(), bq.cuda(), argmax=am1)
        torch.cuda.synchronize()
        e = (ysq.cpu() - rsq).abs().max().item(); t2 = 2e-5 * max(1.0, rsq.abs().max().item(), pl.abs().max().item()) + 1e-5
        if not (torch.equal(p0, p1) and torch.equal(am0, am1) and e <= t2):
            fail('stem_pool_squeeze train', B=B, H=Hs, W=Wsz, err=e, tol=t2, pooled=torch.equal(p0, p1), codes=int((am0 != am1).sum()))
        n['stem'] += 1
print(f'OK {n} in {time.time() - t0:.0f} s, worst err/tol {worst:.3f}')
