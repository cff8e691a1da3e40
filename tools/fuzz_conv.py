"""Randomised parity sweep of sqd_conv_fwd / sqd_fire_expand_fwd / sqd_conv_wgrad against torch (CPU fp32):
every compiled tile configuration (and the Winograd family on the 3x3 cases) x random shapes (tiny maps, widths off the 16-pixel grid, partial K chunks,
channel windows inside wider buffers), with the epilogue options the backward uses.  usage: fuzz_conv.py [seconds]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, torch.nn.functional as F
from squeezedet_pytorch_amd import ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
tab = ops.cfg_table()
t0 = time.time(); n_ok = 0; worst = 0.0; t_say = t0
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous()
while time.time() - t0 < budget:
    if time.time() - t_say > 30:
        print(f'  .. {n_ok} cases ok after {time.time() - t0:.0f} s', flush=True); t_say = time.time()
    cid = int(rs.choice(list(tab)))
    taps, kc, px, bn = tab[cid]
    B = int(rs.randint(1, 4)); H = int(rs.choice([1, 2, 3, 5, 8, 9, 16, 17, 24, 31])); W = int(rs.choice([1, 3, 7, 15, 16, 17, 33, 47, 78]))
    C = int(rs.choice([4, 8, 12, 16, 20, 32, 40, 48, 64, 96])); N = int(rs.choice([4, 8, 16, 24, 32, 48, 64, 72, 80, 96]))
    k = 3 if taps == 9 else 1
    x = torch.randn(B, C, H, W); w = torch.randn(N, C, k, k) * (1.0 / (C * taps)) ** 0.5; b = torch.randn(N) * 0.1
    xp, xo = C + 4 * int(rs.randint(0, 3)), 0
    xo = 4 * int(rs.randint(0, (xp - C) // 4 + 1))
    yp = N + 4 * int(rs.randint(0, 3)); yo = 4 * int(rs.randint(0, (yp - N) // 4 + 1))
    xb = torch.randn(B, H, W, xp); xb[..., xo:xo + C] = nhwc(x)
    mode = int(rs.randint(0, 4))
    relu = bool(rs.randint(0, 2))
    ref = F.conv2d(x, w, b, padding=k // 2)
    y0 = torch.randn(B, H, W, yp)
    yb = y0.clone().cuda()
    kw = {}
    refn = nhwc(ref)
    dma = ops.cfg_is_dma(cid)
    if mode == 1:                      # accumulate
        kw['accumulate'] = True; refn = refn + y0[..., yo:yo + N]
    if mode == 2:                      # ReLU-backward mask with the output's geometry
        m = torch.randn(B, H, W, yp); kw['ymask'] = m.cuda(); kw['ymask_coff'] = yo
        refn = torch.where(m[..., yo:yo + N] > 0, refn, torch.zeros(()))
    if mode == 3:                      # dropout scale with a different geometry (slow epilogue path)
        m = torch.rand(B, H, W, N + 4); kw['ymul'] = m.cuda(); kw['ymul_coff'] = 4
        refn = refn * m[..., 4:4 + N]
    if relu: refn = torch.relu(refn)
    cap = int(rs.choice([0, 0, 1, 2]))
    plan = ops.ConvPlan(w.cuda(), b.cuda(), cid + 1000 * cap)
    ops.conv(xb.cuda(), xo, plan, yb, yo, relu=relu, **kw)
    out = yb.cpu()
    err = (out[..., yo:yo + N] - refn).abs().max().item()
    tol = 2e-5 * max(1.0, refn.abs().max().item()) + 1e-5
    untouched = torch.equal(out[..., :yo], y0[..., :yo]) and torch.equal(out[..., yo + N:], y0[..., yo + N:])
    if not (err <= tol and untouched):
        print('MISMATCH', dict(cid=cid, cfg=tab[cid], dma=dma, B=B, H=H, W=W, C=C, N=N, xp=xp, xo=xo, yp=yp, yo=yo, mode=mode, relu=relu, cap=cap, err=err, tol=tol, untouched=untouched))
        sys.exit(1)
    worst = max(worst, err / tol); n_ok += 1
    # fused expand on the same input when the shape allows
    if taps == 9 and N % 16 == 0 and (cid in ops.fused_expand_cfgs(N)):
        w1 = torch.randn(N, C, 1, 1) * (1.0 / C) ** 0.5; b1 = torch.randn(N) * 0.1
        want = torch.cat([torch.relu(F.conv2d(x, w1, b1)), torch.relu(ref)], 1)
        yf = torch.randn(B, H, W, 2 * N + 8).cuda(); keep = yf.clone()
        ops.fire_expand(xb.cuda(), xo, ops.FusedExpandPlan(w1.cuda(), b1.cuda(), w.cuda(), b.cuda(), cid), yf, 4)
        e2 = (yf[..., 4:4 + 2 * N].cpu() - nhwc(want)).abs().max().item()
        if not (e2 <= tol and torch.equal(yf[..., :4], keep[..., :4]) and torch.equal(yf[..., 4 + 2 * N:], keep[..., 4 + 2 * N:])):
            print('FUSED MISMATCH', dict(cid=cid, B=B, H=H, W=W, C=C, E=N, err=e2, tol=tol)); sys.exit(1)
        n_ok += 1
    # Winograd form of the same 3x3 layer (C % 8 == 0), same windows; the epilogue options need y's own geometry
    if taps == 9 and C % 8 == 0:
        wc = int(rs.choice([c for c in ops.wino_cfgs() if ops.wino_cfg_ok(c, C) and ops.wino_cfg_ok(c, N) and (c != ops.WINO_VS_CFG or N <= 80)]))
        vs = wc == ops.WINO_VS_CFG                              # the V-shared kernel: N <= 80, plain epilogue, no workgroup cap
        wc += 0 if vs else 1000 * int(rs.choice([0, 0, 1]))
        wmode = 0 if vs else int(rs.randint(0, 4)); wrelu = bool(rs.randint(0, 2))
        yw = y0.clone().cuda(); wkw = {}; wref = nhwc(ref)
        if wmode in (1, 3):
            wkw['accumulate'] = True; wref = wref + y0[..., yo:yo + N]
        if wmode in (2, 3):
            mm = torch.rand(B, H, W, yp) + 0.5; mk = torch.randn(B, H, W, yp)
            wkw['ymul'] = mm.cuda(); wkw['ymask'] = mk.cuda()
            wref = torch.where(mk[..., yo:yo + N] > 0, wref * mm[..., yo:yo + N], torch.zeros(()))
        if wrelu: wref = torch.relu(wref)
        ops.conv_wino(xb.cuda(), xo, ops.WinoPlan(w.cuda(), b.cuda(), wc), yw, yo, relu=wrelu, **wkw)
        outw = yw.cpu()
        e5 = (outw[..., yo:yo + N] - wref).abs().max().item()
        tolw = 2e-5 * max(1.0, wref.abs().max().item()) + 1e-5
        if not (e5 <= tolw and torch.equal(outw[..., :yo], y0[..., :yo]) and torch.equal(outw[..., yo + N:], y0[..., yo + N:])):
            print('WINOGRAD MISMATCH', dict(wc=wc, B=B, H=H, W=W, C=C, N=N, xp=xp, xo=xo, yp=yp, yo=yo, mode=wmode, relu=wrelu, err=e5, tol=tolw)); sys.exit(1)
        worst = max(worst, e5 / tolw); n_ok += 1
        if N % 8 == 0 and rs.rand() < 0.5 and (not vs or C <= 80):             # data-gradient packing: dX = conv_transpose(dY, w)
            dyw = torch.randn(B, N, H, W)
            refdx = nhwc(F.conv_transpose2d(dyw, w, None, padding=1))
            dxw = torch.full((B, H, W, C), float('nan')).cuda()
            ops.conv_wino(nhwc(dyw).cuda(), 0, ops.WinoPlan(w.cuda(), None, wc, dgrad=True), dxw, 0)
            e6 = (dxw.cpu() - refdx).abs().max().item()
            if not e6 <= 2e-5 * max(1.0, refdx.abs().max().item()) + 1e-5:
                print('WINOGRAD DGRAD MISMATCH', dict(wc=wc, B=B, H=H, W=W, C=C, N=N, err=e6)); sys.exit(1)
            n_ok += 1
    # weight gradient of the same layer
    if rs.rand() < 0.3:
        dy = torch.randn(B, N, H, W)
        xg = x.clone().requires_grad_(True); wg = w.clone().requires_grad_(True); bg = b.clone().requires_grad_(True)
        F.conv2d(xg, wg, bg, padding=k // 2).backward(dy)
        dyb = torch.randn(B, H, W, yp); dyb[..., yo:yo + N] = nhwc(dy)
        dw, db = ops.conv_wgrad(dyb.cuda(), yo, N, xb.cuda(), xo, C, taps)
        sc = max(1.0, wg.grad.abs().max().item())
        e3 = (dw.cpu() - wg.grad).abs().max().item(); e4 = (db.cpu() - bg.grad).abs().max().item()
        if not (e3 <= 1e-4 * sc and e4 <= 1e-4 * max(1.0, bg.grad.abs().max().item())):
            print('WGRAD MISMATCH', dict(taps=taps, B=B, H=H, W=W, C=C, N=N, e3=e3, e4=e4, sc=sc)); sys.exit(1)
        n_ok += 1
# Fused squeeze backward (sqd_squeeze_bwd: weight-gradient slabs + data gradient + ReLU mask in one launch) and the pool whose
# arg-max codes carry the ReLU mask: random shapes (partial pixel blocks, channel counts off the 64-channel tile)
ts = time.time()
nsq = 0
while time.time() - ts < max(2.0, 0.1 * budget):
    B = int(rs.randint(1, 4)); H = int(rs.choice([1, 3, 5, 8, 13, 24, 31])); W = int(rs.choice([1, 3, 7, 16, 17, 33, 47, 78]))
    C = int(rs.choice([4, 16, 48, 64, 68, 128, 200, 256, 384])); N = int(rs.choice([4, 16, 24, 32, 48, 64, 72, 96]))
    mask = bool(rs.randint(0, 2))
    pre = torch.randn(B, C, H, W, requires_grad=True)
    xin = torch.relu(pre) if mask else pre
    w = (torch.randn(N, C, 1, 1) * (1.0 / C) ** 0.5).requires_grad_(True); b = (torch.randn(N) * 0.1).requires_grad_(True)
    dy = torch.randn(B, N, H, W)
    F.conv2d(xin, w, b).backward(dy)
    S, stride = ops.wgrad_split(N, C, 1, B, H, W, fused_dgrad=True)
    slab = torch.full((S * stride,), float('nan')).cuda(); dx = torch.full((B, H, W, C), float('nan')).cuda()
    ops.squeeze_bwd(nhwc(dy).cuda(), nhwc(xin.detach()).cuda(), w.detach().cuda().contiguous(), slab, dx, relu_mask=mask)
    red = slab.view(S, stride).double().sum(0).cpu()
    e1 = (red[:N * C].view(N, C, 1, 1) - w.grad.double()).abs().max().item(); e2 = (red[N * C:] - b.grad.double()).abs().max().item()
    e3 = (dx.cpu() - nhwc(pre.grad)).abs().max().item()
    if not (e1 <= 1e-4 * max(1.0, w.grad.abs().max().item()) and e2 <= 1e-4 * max(1.0, b.grad.abs().max().item())
            and e3 <= 2e-5 * max(1.0, pre.grad.abs().max().item()) + 1e-5):
        print('SQUEEZE_BWD MISMATCH', dict(B=B, H=H, W=W, C=C, N=N, mask=mask, e1=e1, e2=e2, e3=e3)); sys.exit(1)
    n_ok += 1; nsq += 1
    if H >= 3 and W >= 3 and C % 4 == 0:
        prep = torch.randn(B, C, H, W, requires_grad=True)
        pooled = F.max_pool2d(torch.relu(prep), 3, 2, ceil_mode=True)
        dyp = torch.randn(*pooled.shape); pooled.backward(dyp)
        xg = nhwc(torch.relu(prep.detach())).cuda()
        am = torch.empty(*nhwc(pooled.detach()).shape, dtype=torch.uint8).cuda()
        yg = ops.maxpool(xg, argmax=am, relu_codes=True)
        dxp = ops.maxpool_bwd(nhwc(dyp).cuda(), am, (H, W))
        if not (torch.equal(yg.cpu(), nhwc(pooled.detach())) and (dxp.cpu() - nhwc(prep.grad)).abs().max().item() <= 1e-6):
            print('POOL RELU-CODES MISMATCH', dict(B=B, H=H, W=W, C=C)); sys.exit(1)
        n_ok += 1
print(f'  .. {nsq} fused squeeze-backward cases ok', flush=True)
# Grouped Winograd weight gradient (sqd_conv_wgrad_wino_group: several 3x3 layers of one pixel grid in one launch, same splits for all):
# random groups of 1..6 layers of one tile form, channel windows inside wider buffers, ragged grids; every member's slab sum against autograd
tg = time.time()
ngrp = 0
while time.time() - tg < max(2.0, 0.08 * budget):
    B = int(rs.randint(1, 4)); H = int(rs.choice([1, 3, 4, 5, 8, 13, 24, 31])); W = int(rs.choice([1, 3, 15, 16, 17, 33, 47, 78]))
    tc = int(rs.choice([1, 2]))
    nl = int(rs.randint(1, 7))
    ngroups = B * -(-H // 4) * -(-W // 16)
    S = int(rs.randint(1, min(ngroups, 9) + 1))
    items, refs = [], []
    for _ in range(nl):
        C = int(rs.choice([4, 16, 20, 48, 80]) if tc == 1 else rs.choice([32, 36, 64, 96, 128]))
        N = int(rs.choice([64, 128, 192]))
        assert ops._wino_wgrad_tc(N, C) == tc
        xin = torch.randn(B, C, H, W)
        w = (torch.randn(N, C, 3, 3) * (1.0 / (9 * C)) ** 0.5).requires_grad_(True); b = (torch.randn(N) * 0.1).requires_grad_(True)
        dy = torch.randn(B, N, H, W)
        F.conv2d(xin, w, b, padding=1).backward(dy)
        dyp = N + 4 * int(rs.randint(0, 3)); dyo = 4 * int(rs.randint(0, (dyp - N) // 4 + 1))
        xp = C + 4 * int(rs.randint(0, 3)); xo = 4 * int(rs.randint(0, (xp - C) // 4 + 1))
        dyb = torch.randn(B, H, W, dyp); dyb[..., dyo:dyo + N] = nhwc(dy)
        xb = torch.randn(B, H, W, xp); xb[..., xo:xo + C] = nhwc(xin)
        slab = torch.full((S * (N * 9 * C + N),), float('nan')).cuda()
        items.append((dyb.cuda(), dyo, N, xb.cuda(), xo, C, slab)); refs.append((w.grad, b.grad))
    ops.conv_wgrad_wino_group(items, S, tc)
    for (dyb, dyo, N, xb, xo, C, slab), (gw, gb) in zip(items, refs):
        red = slab.view(S, N * 9 * C + N).double().sum(0).cpu()
        dw = red[:N * 9 * C].view(N, 3, 3, C).permute(0, 3, 1, 2)
        e1 = (dw - gw.double()).abs().max().item(); e2 = (red[N * 9 * C:] - gb.double()).abs().max().item()
        if not (e1 <= 1e-4 * max(1.0, gw.abs().max().item()) and e2 <= 1e-4 * max(1.0, gb.abs().max().item())):
            print('WGRAD GROUP MISMATCH', dict(B=B, H=H, W=W, C=C, N=N, S=S, tc=tc, nl=nl, e1=e1, e2=e2)); sys.exit(1)
        worst = max(worst, e1 / (1e-4 * max(1.0, gw.abs().max().item())))
    n_ok += 1; ngrp += 1
print(f'  .. {ngrp} grouped weight-gradient launches ok', flush=True)
# Fire bridges (one launch for expand pair + next squeeze, optionally through the max pool): random small-C Fire shapes
tb = time.time()
nb = 0
while time.time() - tb < max(2.0, 0.15 * budget):
    B = int(rs.randint(1, 4)); H = int(rs.choice([3, 4, 5, 8, 9, 13, 16, 23, 31])); W = int(rs.choice([3, 7, 14, 15, 16, 17, 29, 33, 47, 61]))
    C = int(rs.choice([8, 16])); E1 = int(rs.choice([16, 32, 48, 64])); E3 = int(rs.choice([8, 16, 24, 40, 64])); S = int(rs.choice([4, 12, 16, 24, 32]))
    x = torch.relu(torch.randn(B, C, H, W))
    w1 = torch.randn(E1, C, 1, 1) / C ** 0.5; b1 = torch.randn(E1) * 0.1
    w3 = torch.randn(E3, C, 3, 3) / (9 * C) ** 0.5; b3 = torch.randn(E3) * 0.1
    ws = torch.randn(S, E1 + E3, 1, 1) / (E1 + E3) ** 0.5; bs = torch.randn(S) * 0.1
    mid = torch.cat([torch.relu(F.conv2d(x, w1, b1)), torch.relu(F.conv2d(x, w3, b3, padding=1))], 1)
    pooled = bool(rs.randint(0, 2))
    args = [t.cuda() for t in (w1, b1, w3, b3, ws, bs)]
    if pooled:
        if not ops.fire_pool_bridge_ok(C, E3, E1, S):
            continue
        ref = nhwc(torch.relu(F.conv2d(F.max_pool2d(mid, 3, 2, ceil_mode=True), ws, bs)))
        nseg = int(rs.randint(1, 8))
        y = torch.full((B, ref.shape[1], ref.shape[2], S + 8), -3.0).cuda()
        ops.fire_pool_bridge(nhwc(x).cuda(), 0, ops.FireBridgePlan(*args, 12, pooled=True), y, 4, nseg=nseg)
        what = dict(kind='pool bridge', nseg=nseg)
    else:
        cands = [c for c in ops.FIRE_BRIDGE_CFGS if ops.fire_bridge_cfg_ok(c, C, E3, E1, S)]
        if not cands:
            continue
        cid = int(rs.choice(cands))
        ref = nhwc(torch.relu(F.conv2d(mid, ws, bs)))
        y = torch.full((B, H, W, S + 8), -3.0).cuda()
        ops.fire_bridge(nhwc(x).cuda(), 0, ops.FireBridgePlan(*args, cid), y, 4)
        what = dict(kind='bridge', cfg=cid)
    out = y.cpu()
    e7 = (out[..., 4:4 + S] - ref).abs().max().item()
    tol7 = 2e-5 * max(1.0, ref.abs().max().item()) + 1e-5
    if not (e7 <= tol7 and bool((out[..., :4] == -3.0).all()) and bool((out[..., 4 + S:] == -3.0).all())):
        print('BRIDGE MISMATCH', dict(B=B, H=H, W=W, C=C, E1=E1, E3=E3, S=S, err=e7, tol=tol7, **what)); sys.exit(1)
    worst = max(worst, e7 / tol7); nb += 1
n_ok += nb
print(f'fuzz ok: {n_ok} cases ({nb} bridge launches) in {time.time() - t0:.0f} s, worst err/tol {worst:.2f}')
