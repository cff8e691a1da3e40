"""GPU tier, kernel level: each HIP kernel against a plain fp32 CPU reference of the same op
(torch.nn.functional on CPU = the arithmetic the reference delegates to), through the C-ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _ops():
    from squeezedet_pytorch_amd import ops
    return ops


def _rand(*shape, seed=0, scale=1.0):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32))


def _nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def _tol(ref):
    # fp32 summation-order noise scales with the magnitude of the sums
    return TOL * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("taps,C,N,B,H,W", [
    (1, 64, 16, 2, 12, 20), (1, 128, 32, 1, 9, 13), (1, 256, 48, 2, 7, 11), (1, 512, 64, 1, 6, 10),
    (1, 768, 96, 1, 5, 9), (1, 16, 64, 2, 12, 20), (1, 48, 192, 1, 7, 11), (1, 96, 384, 1, 6, 7),
    (9, 16, 64, 2, 12, 20), (9, 32, 128, 1, 9, 33), (9, 48, 192, 1, 24, 78), (9, 96, 384, 1, 5, 17),
    (9, 768, 72, 1, 6, 18), (9, 16, 64, 1, 3, 3), (1, 100, 20, 1, 4, 5),
])
def test_conv_fwd_all_cfgs(taps, C, N, B, H, W):
    ops = _ops()
    k = 3 if taps == 9 else 1
    x = _rand(B, C, H, W, seed=1)
    w = _rand(N, C, k, k, seed=2, scale=(2.0 / (C * taps)) ** 0.5)
    b = _rand(N, seed=3, scale=0.1)
    ref = _nhwc(F.relu(F.conv2d(x, w, b, padding=k // 2)))
    xg = _nhwc(x).cuda()
    cfgs = [cid for cid, (t, kc, px, bn) in ops.cfg_table().items() if t == taps and ops.conv_cfg_ok(cid, C)]
    assert cfgs
    for cid in cfgs:
        plan = ops.ConvPlan(w.cuda(), b.cuda(), cid)
        y = torch.full((B, H, W, N), float('nan'), device='cuda')
        ops.conv(xg, 0, plan, y, 0, relu=True)
        err = (y.cpu() - ref).abs().max().item()
        assert err <= _tol(ref), f'cfg {cid}: max err {err}'
    # automatic choice is one of them
    assert ops.choose_cfg(taps, C, N, B * H * W) % 1000 in cfgs          # (+ 1000 * k = workgroup cap)


def test_conv_channel_windows_accumulate_mask():
    ops = _ops()
    B, H, W = 2, 10, 19
    C, N = 32, 64
    xfull = _rand(B, 48, H, W, seed=4)                 # read channels [8, 40)
    w = _rand(N, C, 3, 3, seed=5, scale=0.1)
    mask_src = _rand(B, 48, H, W, seed=6)
    x = xfull[:, 8:40]
    xm = x * (mask_src[:, 8:40] > 0)
    ref = F.conv2d(xm, w, None, padding=1)
    y0 = _rand(B, 96, H, W, seed=7)                    # write channels [16, 80) of a 96-wide buffer, accumulate
    exp = y0.clone()
    exp[:, 16:80] += ref
    plan = ops.ConvPlan(w.cuda(), None, ops.choose_cfg(9, C, N, B * H * W, staged=True))   # xmask: register-staged family
    y = _nhwc(y0).cuda()
    ops.conv(_nhwc(xfull).cuda(), 8, plan, y, 16, relu=False, accumulate=True, xmask=_nhwc(mask_src).cuda(), xmask_coff=8)
    got = y.cpu().permute(0, 3, 1, 2)
    assert (got - exp).abs().max().item() <= _tol(exp)
    # untouched channels are bit-identical
    assert torch.equal(got[:, :16], y0[:, :16]) and torch.equal(got[:, 80:], y0[:, 80:])


def test_conv_dgrad_weights_equal_autograd():
    ops = _ops()
    B, C, N, H, W = 1, 16, 64, 9, 21
    x = _rand(B, C, H, W, seed=8).requires_grad_(True)
    w = _rand(N, C, 3, 3, seed=9, scale=0.1)
    y = F.conv2d(x, w, None, padding=1)
    dy = _rand(B, N, H, W, seed=10)
    y.backward(dy)
    plan = ops.ConvPlan(ops.dgrad_weight(w.cuda()), None, ops.choose_cfg(9, N, C, B * H * W))
    dx = torch.empty(B, H, W, C, device='cuda')
    ops.conv(_nhwc(dy).cuda(), 0, plan, dx, 0)
    ref = _nhwc(x.grad)
    assert (dx.cpu() - ref).abs().max().item() <= _tol(ref)


@pytest.mark.parametrize("k,N,H,W", [(3, 64, 64, 96), (3, 64, 33, 47), (7, 96, 64, 96), (7, 96, 30, 50)])
def test_stem(k, N, H, W):
    ops = _ops()
    x = _rand(2, 3, H, W, seed=11)
    w = _rand(N, 3, k, k, seed=12, scale=(2.0 / (3 * k * k)) ** 0.5)
    b = _rand(N, seed=13, scale=0.1)
    ref = _nhwc(F.relu(F.conv2d(x, w, b, stride=2, padding=1 if k == 3 else 3)))
    y = ops.stem_conv_relu(x.cuda(), w.cuda(), b.cuda())
    assert tuple(y.shape) == tuple(ref.shape)
    assert (y.cpu() - ref).abs().max().item() <= _tol(ref)


@pytest.mark.parametrize("C,H,W", [(64, 32, 48), (128, 17, 23), (256, 12, 39), (64, 3, 3), (8, 4, 4)])
def test_maxpool_fwd_bwd(C, H, W):
    ops = _ops()
    x = _rand(2, C, H, W, seed=14).requires_grad_(True)
    ref = F.max_pool2d(x, 3, 2, ceil_mode=True)
    dy = _rand(*ref.shape, seed=15)
    ref.backward(dy)
    xg = _nhwc(x.detach()).cuda()
    am = torch.empty(*_nhwc(ref.detach()).shape, dtype=torch.uint8, device='cuda')
    y = ops.maxpool(xg, argmax=am)
    assert torch.equal(y.cpu(), _nhwc(ref.detach()))            # max is exact
    dx = ops.maxpool_bwd(_nhwc(dy).cuda(), am, (H, W))
    assert (dx.cpu() - _nhwc(x.grad)).abs().max().item() <= 1e-6


def test_maxpool_bwd_batches_beyond_one_grid():
    """The backward kernel's grid carries (image, row pair) in blockIdx.y (<= 65535 per launch): ops.maxpool_bwd slices larger batches
    into launches of whole images instead of refusing them (B * ceil(H / 2) = 40000 * 2 here); bit-equal to the per-slice result of
    autograd."""
    ops = _ops()
    B, C, H, W = 40000, 4, 4, 5
    x = _rand(B, C, H, W, seed=18).requires_grad_(True)
    ref = F.max_pool2d(x, 3, 2, ceil_mode=True)
    dy = _rand(*ref.shape, seed=19)
    ref.backward(dy)
    xg = _nhwc(x.detach()).cuda()
    am = torch.empty(*_nhwc(ref.detach()).shape, dtype=torch.uint8, device='cuda')
    half = B // 2
    ops.maxpool(xg[:half], argmax=am[:half]); ops.maxpool(xg[half:], argmax=am[half:])
    dx = ops.maxpool_bwd(_nhwc(dy).cuda(), am, (H, W))
    assert (dx.cpu() - _nhwc(x.grad)).abs().max().item() <= 1e-6


@pytest.mark.parametrize("C,H,W", [(64, 32, 48), (128, 17, 23), (8, 4, 4), (128, 96, 312)])
def test_maxpool_relu_codes_carry_the_mask(C, H, W):
    """Training path: the pool input is a ReLU output.  ``maxpool(relu_codes=True)`` writes code 15 where the pooled value is not
    > 0; ``maxpool_bwd`` on those codes WITHOUT a mask tensor equals the old path (plain codes + relu_src re-read) bit for bit,
    and equals autograd through relu -> max_pool2d."""
    ops = _ops()
    pre = _rand(2, C, H, W, seed=16).requires_grad_(True)
    ref = F.max_pool2d(F.relu(pre), 3, 2, ceil_mode=True)
    dy = _rand(*ref.shape, seed=17)
    ref.backward(dy)
    xg = _nhwc(F.relu(pre.detach())).cuda()
    shape = _nhwc(ref.detach()).shape
    am_plain = torch.empty(*shape, dtype=torch.uint8, device='cuda')
    am_relu = torch.empty(*shape, dtype=torch.uint8, device='cuda')
    y0 = ops.maxpool(xg, argmax=am_plain)
    y1 = ops.maxpool(xg, argmax=am_relu, relu_codes=True)
    assert torch.equal(y0, y1) and torch.equal(y0.cpu(), _nhwc(ref.detach()))
    zero = y0 <= 0
    assert bool((am_relu[zero] == 15).all()) and torch.equal(am_relu[~zero], am_plain[~zero]) and int(am_plain.max()) <= 8
    assert bool(zero.any())                                   # the case is exercised
    dyg = _nhwc(dy).cuda()
    old = ops.maxpool_bwd(dyg, am_plain, (H, W), relu_src=xg)
    new = ops.maxpool_bwd(dyg, am_relu, (H, W))
    assert torch.equal(old, new)
    assert (new.cpu() - _nhwc(pre.grad)).abs().max().item() <= 1e-6


@pytest.mark.parametrize("k,N,H,W", [(3, 64, 64, 96), (3, 64, 50, 70), (3, 64, 37, 45), (7, 96, 64, 96), (7, 96, 30, 50)])
def test_fused_stem_pool(k, N, H, W):
    ops = _ops()
    x = _rand(2, 3, H, W, seed=21)
    w = _rand(N, 3, k, k, seed=22, scale=(2.0 / (3 * k * k)) ** 0.5)
    b = _rand(N, seed=23, scale=0.1)
    conv = F.relu(F.conv2d(x, w, b, stride=2, padding=1 if k == 3 else 3))
    ref, ridx = F.max_pool2d(conv, 3, 2, ceil_mode=True, return_indices=True)
    am = torch.empty(*_nhwc(ref).shape, dtype=torch.uint8, device='cuda')
    y = ops.stem_pool(x.cuda(), w.cuda(), b.cuda(), argmax=am)
    assert tuple(y.shape) == tuple(_nhwc(ref).shape)
    assert (y.cpu() - _nhwc(ref)).abs().max().item() <= _tol(ref)
    # argmax consistent with the separate kernels (same window code 0..8, 15 where the pooled value is 0: the ReLU mask)
    y2 = ops.stem_conv_relu(x.cuda(), w.cuda(), b.cuda())
    am2 = torch.empty_like(am)
    p2 = ops.maxpool(y2, argmax=am2, relu_codes=True)
    assert torch.equal(p2, y)
    assert (am == am2).float().mean().item() > 0.999
    assert bool((am[y <= 0] == 15).all()) and int(am[y > 0].max()) <= 8
    # inference instantiation (no argmax, v_max3 pooling): same bits
    assert torch.equal(ops.stem_pool(x.cuda(), w.cuda(), b.cuda()), y)


@pytest.mark.parametrize("variant", ["2", "3", "4"])
@pytest.mark.parametrize("H,W", [(64, 96), (50, 68), (37, 44), (12, 16), (9, 8), (130, 250 * 4)])
def test_stem_wave_kernels(variant, H, W, monkeypatch):
    """The wave-autonomous inference stem (stem_wave_kernel<PH, CB>, selected by SQD_STEM_WAVE) == features[0..2] of the reference
    (src/model/squeezedet.py:34-36) and, bit for bit, the workgroup kernel it replaces: widths that are a multiple of 4 (its 16-byte
    row DMA), maps smaller than one tile, borders on all four sides, partial last tiles."""
    ops = _ops()
    x = _rand(2, 3, H, W, seed=24)
    w = _rand(64, 3, 3, 3, seed=25, scale=(2.0 / 27) ** 0.5)
    b = _rand(64, seed=26, scale=0.1)
    ref = F.max_pool2d(F.relu(F.conv2d(x, w, b, stride=2, padding=1)), 3, 2, ceil_mode=True)
    monkeypatch.setenv('SQD_STEM_WAVE', '0')
    y_old = ops.stem_pool(x.cuda(), w.cuda(), b.cuda())
    monkeypatch.setenv('SQD_STEM_WAVE', variant)
    y = ops.stem_pool(x.cuda(), w.cuda(), b.cuda())
    assert tuple(y.shape) == tuple(_nhwc(ref).shape)
    assert (y.cpu() - _nhwc(ref)).abs().max().item() <= _tol(ref)
    assert torch.equal(y, y_old)


@pytest.mark.parametrize("H,W", [(64, 96), (50, 68), (37, 44), (12, 16), (9, 8), (384, 1248)])
def test_stem_pool_squeeze_one_launch(H, W):
    """features[0..2] + the first Fire's squeeze in one launch (ops.stem_pool_squeeze, inference) == the reference's modules in fp32
    (src/model/squeezedet.py:34-37, :17-18) and == the two launches it replaces within summation-order noise."""
    ops = _ops()
    B = 1 if H == 384 else 2
    x = _rand(B, 3, H, W, seed=27)
    w = _rand(64, 3, 3, 3, seed=28, scale=(2.0 / 27) ** 0.5); b = _rand(64, seed=29, scale=0.1)
    ws = _rand(16, 64, 1, 1, seed=30, scale=(2.0 / 64) ** 0.5); bs = _rand(16, seed=31, scale=0.1)
    pooled = F.max_pool2d(F.relu(F.conv2d(x, w, b, stride=2, padding=1)), 3, 2, ceil_mode=True)
    ref = _nhwc(F.relu(F.conv2d(pooled, ws, bs)))
    assert ops.stem_pool_squeeze_ok(x.shape, w.shape, 16)
    y = ops.stem_pool_squeeze(x.cuda(), w.cuda(), b.cuda(), ws.cuda(), bs.cuda())
    assert tuple(y.shape) == tuple(ref.shape)
    assert (y.cpu() - ref).abs().max().item() <= _tol(ref)
    two = torch.empty_like(y)
    p = ops.stem_pool(x.cuda(), w.cuda(), b.cuda())
    ops.conv(p, 0, ops.ConvPlan(ws.cuda(), bs.cuda(), ops.choose_cfg(1, 64, 16, p.shape[0] * p.shape[1] * p.shape[2])), two, 0, relu=True)
    assert (y - two).abs().max().item() <= 1e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(y, ops.stem_pool_squeeze(x.cuda(), w.cuda(), b.cuda(), ws.cuda(), bs.cuda()))       # run-to-run
    # widths the 16-byte row DMA cannot take are refused on the host (the model then keeps the two launches)
    assert not ops.stem_pool_squeeze_ok((2, 3, 50, 70), w.shape, 16) and not ops.stem_pool_squeeze_ok(x.shape, w.shape, 32)


@pytest.mark.parametrize("H,W", [(64, 96), (50, 68), (37, 44), (12, 16), (9, 8), (384, 1248)])
def test_stem_pool_squeeze_training_form(H, W):
    """Training form of the stem + squeeze launch (sqd_stem_pool_squeeze_train_fwd): the pooled tensor and the arg-max / ReLU codes
    are BIT-identical to the stem launch the training forward used before (ops.stem_pool(argmax=)), and the squeeze output equals
    the reference's modules in fp32 (src/model/squeezedet.py:34-37, :17-18)."""
    ops = _ops()
    B = 1 if H == 384 else 2
    x = _rand(B, 3, H, W, seed=27)
    w = _rand(64, 3, 3, 3, seed=28, scale=(2.0 / 27) ** 0.5); b = _rand(64, seed=29, scale=0.1)
    ws = _rand(16, 64, 1, 1, seed=30, scale=(2.0 / 64) ** 0.5); bs = _rand(16, seed=31, scale=0.1)
    pooled = F.max_pool2d(F.relu(F.conv2d(x, w, b, stride=2, padding=1)), 3, 2, ceil_mode=True)
    ref = _nhwc(F.relu(F.conv2d(pooled, ws, bs)))
    Hp, Wp = pooled.shape[2], pooled.shape[3]
    am0 = torch.full((B, Hp, Wp, 64), 77, dtype=torch.uint8, device='cuda')
    p0 = ops.stem_pool(x.cuda(), w.cuda(), b.cuda(), argmax=am0)
    am1 = torch.full((B, Hp, Wp, 64), 78, dtype=torch.uint8, device='cuda')
    y, p1 = ops.stem_pool_squeeze(x.cuda(), w.cuda(), b.cuda(), ws.cuda(), bs.cuda(), argmax=am1)
    torch.cuda.synchronize()
    assert torch.equal(p0, p1) and torch.equal(am0, am1)
    assert tuple(y.shape) == tuple(ref.shape)
    assert (y.cpu() - ref).abs().max().item() <= _tol(ref)
    with pytest.raises(ValueError):
        ops.stem_pool_squeeze(x.cuda(), w.cuda(), b.cuda(), ws.cuda(), bs.cuda(), argmax=am1[:, :-1])


def test_fused_stem_pool_kitti_size_both_paths_agree():
    ops = _ops()
    x = _rand(3, 3, 384, 1248, seed=31).cuda()
    w = _rand(64, 3, 3, 3, seed=32, scale=(2.0 / 27) ** 0.5).cuda()
    b = _rand(64, seed=33, scale=0.1).cuda()
    y_inf = ops.stem_pool(x, w, b)
    am = torch.empty(*y_inf.shape, dtype=torch.uint8, device='cuda')
    y_tr = ops.stem_pool(x, w, b, argmax=am)
    assert torch.equal(y_inf, y_tr)
    ref = F.max_pool2d(F.relu(F.conv2d(x[:1].cpu(), w.cpu(), b.cpu(), stride=2, padding=1)), 3, 2, ceil_mode=True)
    assert (y_inf[:1].cpu() - _nhwc(ref)).abs().max().item() <= _tol(ref)
    assert int(am[y_tr > 0].max()) <= 8 and bool((am[y_tr <= 0] == 15).all())


@pytest.mark.parametrize("C,E,H,W", [(16, 64, 40, 70), (32, 128, 24, 78), (48, 192, 13, 29), (96, 384, 24, 78), (64, 256, 9, 17)])
def test_fused_fire_expand_equals_separate_kernels(C, E, H, W):
    """fire_expand (one launch) == expand1x1 + expand3x3 launches, bit for bit, for every usable tile configuration."""
    ops = _ops()
    x = F.relu(_rand(2, C, H, W, seed=41)).cuda()
    xn = x.permute(0, 2, 3, 1).contiguous()
    w1 = _rand(E, C, 1, 1, seed=42, scale=(2.0 / C) ** 0.5).cuda(); b1 = _rand(E, seed=43, scale=0.1).cuda()
    w3 = _rand(E, C, 3, 3, seed=44, scale=(2.0 / (9 * C)) ** 0.5).cuda(); b3 = _rand(E, seed=45, scale=0.1).cuda()
    npix = 2 * H * W
    ref = torch.empty(2, H, W, 2 * E + 8, device='cuda').fill_(7.0)          # wider pitch + channel offset 4: window writes
    ops.conv(xn, 0, ops.ConvPlan(w1, b1, ops.choose_cfg(1, C, E, npix)), ref, 4, relu=True)
    ops.conv(xn, 0, ops.ConvPlan(w3, b3, ops.choose_cfg(9, C, E, npix)), ref, 4 + E, relu=True)
    want = torch.cat([F.relu(F.conv2d(x.cpu(), w1.cpu(), b1.cpu())), F.relu(F.conv2d(x.cpu(), w3.cpu(), b3.cpu(), padding=1))], 1)
    assert (ref[..., 4:4 + 2 * E].cpu() - _nhwc(want)).abs().max().item() <= _tol(want)
    cfgs = ops.fused_expand_cfgs(E)
    assert cfgs
    for cid in cfgs + [cfgs[0] + 2000]:
        out = torch.empty_like(ref).fill_(7.0)
        ops.fire_expand(xn, 0, ops.FusedExpandPlan(w1, b1, w3, b3, cid), out, 4)
        assert torch.equal(out, ref), f'cfg {cid}'
    with pytest.raises(ValueError):
        ops.FusedExpandPlan(w1, b1, w3, b3, ops.choose_cfg(1, C, E, npix))


@pytest.mark.parametrize("C,N,H,W", [(128, 32, 96, 312), (256, 48, 48, 156), (128, 16, 7, 9), (64, 96, 12, 13), (32, 20, 3, 3)])
def test_fused_pool_squeeze_vs_separate(C, N, H, W):
    """pool_squeeze == maxpool kernel followed by the 1x1 conv kernel (pooled values are exact; the 1x1 sums in the same
    ascending-k order), also against torch."""
    ops = _ops()
    B = 2
    x = F.relu(_rand(B, C, H, W, seed=51)).cuda()
    xn = x.permute(0, 2, 3, 1).contiguous()
    w = _rand(N, C, 1, 1, seed=52, scale=(2.0 / C) ** 0.5).cuda(); b = _rand(N, seed=53, scale=0.1).cuda()
    assert ops.pool_squeeze_ok(C, N)
    Ho, Wo = ops.pool_out_size(H, W)
    out = torch.empty(B, Ho, Wo, N + 8, device='cuda').fill_(3.0)
    ops.pool_squeeze(xn, 0, C, ops.ConvPlan(w, b, ops.POOL_SQUEEZE_CFG), out, 4)
    pooled = ops.maxpool(xn)
    sep = torch.empty_like(out).fill_(3.0)
    ops.conv(pooled, 0, ops.ConvPlan(w, b, ops.POOL_SQUEEZE_CFG), sep, 4, relu=True)
    assert torch.equal(out, sep)
    ref = F.relu(F.conv2d(F.max_pool2d(x.cpu(), 3, 2, ceil_mode=True), w.cpu(), b.cpu()))
    assert (out[..., 4:4 + N].cpu() - _nhwc(ref)).abs().max().item() <= _tol(ref)
    assert not ops.pool_squeeze_ok(256, 192)


@pytest.mark.parametrize("C,N,B,H,W", [
    (16, 64, 2, 12, 20), (32, 128, 1, 9, 33), (48, 192, 1, 24, 78), (96, 384, 1, 5, 17), (768, 72, 1, 6, 18),
    (16, 64, 1, 3, 3), (8, 16, 3, 4, 16), (64, 32, 7, 2, 2), (24, 20, 2, 7, 35),
])
def test_conv_winograd_all_cfgs(C, N, B, H, W):
    """Winograd F(2x2,3x3) kernel == fp32 conv2d of the reference (within the same 1e-4 bound as the direct kernel),
    every configuration, odd sizes (rows % 4, columns % 16, channel slices % 32 != 0), partial last super-group."""
    ops = _ops()
    x = _rand(B, C, H, W, seed=11)
    w = _rand(N, C, 3, 3, seed=12, scale=(2.0 / (C * 9)) ** 0.5)
    b = _rand(N, seed=13, scale=0.1)
    ref = _nhwc(F.relu(F.conv2d(x, w, b, padding=1)))
    xg = _nhwc(x).cuda()
    for cid in ops.wino_cfgs():
        if not ops.wino_cfg_ok(cid, C, N):
            continue
        plan = ops.WinoPlan(w.cuda(), b.cuda(), cid)
        y = torch.full((B, H, W, N), float('nan'), device='cuda')
        ops.conv_wino(xg, 0, plan, y, 0, relu=True)
        err = (y.cpu() - ref).abs().max().item()
        assert err <= _tol(ref), f'wino cfg {cid}: max err {err}'


def test_conv_winograd_windows_no_relu_dgrad():
    """Channel windows of wider buffers (bytes outside the window untouched), no bias / no ReLU, and the data-gradient
    packing (== conv_transpose of the forward weight)."""
    ops = _ops()
    B, H, W, C, N = 2, 10, 19, 32, 64
    xfull = _rand(B, 48, H, W, seed=14)
    w = _rand(N, C, 3, 3, seed=15, scale=0.1)
    ref = F.conv2d(xfull[:, 8:40], w, None, padding=1)
    y0 = _rand(B, 96, H, W, seed=16)
    exp = y0.clone(); exp[:, 16:80] = ref
    for cid in ops.wino_cfgs():
        if not ops.wino_cfg_ok(cid, C, N):
            continue
        y = _nhwc(y0).cuda()
        ops.conv_wino(_nhwc(xfull).cuda(), 8, ops.WinoPlan(w.cuda(), None, cid), y, 16, relu=False)
        got = y.cpu().permute(0, 3, 1, 2)
        assert (got - exp).abs().max().item() <= _tol(exp), f'wino cfg {cid}'
    dy = _rand(B, N, H, W, seed=17)
    ref_dx = F.conv_transpose2d(dy, w, None, padding=1)
    plan = ops.WinoPlan(w.cuda(), None, 0, dgrad=True)
    assert (plan.C, plan.N) == (N, C)
    dx = torch.full((B, H, W, C), float('nan'), device='cuda')
    ops.conv_wino(_nhwc(dy).cuda(), 0, plan, dx, 0)
    assert (dx.cpu() - _nhwc(ref_dx)).abs().max().item() <= _tol(ref_dx)


def test_conv_winograd_dgrad_epilogue():
    """The epilogue options the backward uses: accumulate into y, dropout scale (ymul) and ReLU-backward mask (ymask) read
    through y's own channel window -- same semantics as sqd_conv_fwd."""
    ops = _ops()
    B, H, W, C, N = 2, 9, 21, 64, 32
    dy = _rand(B, C, H, W, seed=21)
    w = _rand(C, N, 3, 3, seed=22, scale=0.1)               # forward weight of a conv N -> C; its dgrad maps C -> N
    ref = F.conv_transpose2d(dy, w, None, padding=1)
    y0 = _rand(B, N, H, W, seed=23)
    mul = _rand(B, N, H, W, seed=24).abs() + 0.5
    mask = _rand(B, N, H, W, seed=25)
    exp = (y0 + ref) * mul * (mask > 0)
    for cid in ops.wino_cfgs():
        if not ops.wino_cfg_ok(cid, C, N) or cid == ops.WINO_VS_CFG:       # (the V-shared kernel has the plain epilogue only)
            continue
        plan = ops.WinoPlan(w.cuda(), None, cid, dgrad=True)
        y = _nhwc(y0).cuda()
        ops.conv_wino(_nhwc(dy).cuda(), 0, plan, y, 0, accumulate=True, ymul=_nhwc(mul).cuda(), ymask=_nhwc(mask).cuda())
        assert (y.cpu() - _nhwc(exp)).abs().max().item() <= _tol(exp), f'wino cfg {cid}'
    with pytest.raises(ValueError):
        ops.conv_wino(_nhwc(dy).cuda(), 0, plan, y, 0, ymask=torch.zeros(B, H, W, N + 4, device='cuda'))


def test_wgrad_batched_reduce_is_bitwise_the_per_layer_reduce():
    """ops.WgradBatch (partial slabs of several layers + ONE reduction into a flat buffer) == per-layer conv_wgrad,
    bit for bit, for 1x1 and 3x3 layers of different sizes."""
    ops = _ops()
    B, H, W = 2, 9, 21
    layers = [('a', 32, 16, 9), ('b', 16, 64, 1), ('c', 72, 24, 9), ('d', 48, 48, 1)]       # (key, N, C, taps)
    off, entries, slots = 0, [], {}
    for key, N, C, taps in layers:
        slots[key] = (off, off + N * C * taps)
        entries.append((key, N, C, taps, B, H, W, off, off + N * C * taps))
        off += N * C * taps + N
    wb = ops.WgradBatch(entries, torch.device('cuda'))
    flat = torch.full((off,), float('nan'), device='cuda')
    want = {}
    for i, (key, N, C, taps) in enumerate(layers):
        dy = _rand(B, H, W, N + 8, seed=40 + i).cuda(); x = _rand(B, H, W, C + 4, seed=50 + i).cuda()
        want[key] = ops.conv_wgrad(dy, 4, N, x, 4, C, taps)
        assert ops.conv_wgrad(dy, 4, N, x, 4, C, taps, slab=wb.slab(key)) is None
    wb.reduce(flat)
    for key, N, C, taps in layers:
        wo, bo = slots[key]
        k = 3 if taps == 9 else 1
        assert torch.equal(flat[wo:bo].view(N, C, k, k), want[key][0]), key
        assert torch.equal(flat[bo:bo + N], want[key][1]), key
    with pytest.raises(ValueError):
        ops.conv_wgrad(dy, 4, N, x, 4, C, taps, slab=wb.slab('a'))                          # slab of another layer


def test_winograd_batched_repack_matches_single_pack():
    ops = _ops()
    ws = [_rand(64, 16, 3, 3, seed=60).cuda(), _rand(72, 24, 3, 3, seed=61).cuda()]
    plans = [ops.WinoPlan(ws[0], None, 0), ops.WinoPlan(ws[1], None, 2, dgrad=True)]
    ref = [p.w.clone() for p in plans]
    for p in plans:
        p.w.fill_(float('nan'))
    ops.repack_wino_batched(list(zip(plans, ws)), [False, True])
    for p, r in zip(plans, ref):
        assert torch.equal(p.w, r)


@pytest.mark.parametrize("C,N,B,H,W", [
    (16, 64, 2, 12, 20), (32, 128, 1, 9, 33), (48, 192, 1, 24, 78), (96, 64, 1, 5, 17), (64, 256, 2, 8, 16), (24, 64, 3, 3, 3),
    (768, 72, 1, 6, 18), (40, 72, 2, 7, 19), (16, 48, 1, 4, 16),
])
def test_conv_wgrad_winograd(C, N, B, H, W):
    """Winograd weight gradient == autograd of fp32 conv2d (same bound as the direct kernel), channel windows, odd sizes,
    partial last input-channel block; and agrees with the direct kernel."""
    ops = _ops()
    x = _rand(B, C, H, W, seed=71).requires_grad_(False)
    dy = _rand(B, N, H, W, seed=72)
    w = torch.zeros(N, C, 3, 3, requires_grad=True); b = torch.zeros(N, requires_grad=True)
    F.conv2d(x, w, b, padding=1).backward(dy)
    dyb = _rand(B, H, W, N + 8, seed=73); dyb[..., 4:4 + N] = _nhwc(dy)
    xb = _rand(B, H, W, C + 4, seed=74); xb[..., 4:4 + C] = _nhwc(x)
    dw, db = ops.conv_wgrad(dyb.cuda(), 4, N, xb.cuda(), 4, C, 9, wino=True)
    dw0, db0 = ops.conv_wgrad(dyb.cuda(), 4, N, xb.cuda(), 4, C, 9, wino=False)
    sc = max(1.0, w.grad.abs().max().item())
    assert (dw.cpu() - w.grad).abs().max().item() <= 1e-4 * sc
    assert (db.cpu() - b.grad).abs().max().item() <= 1e-4 * max(1.0, b.grad.abs().max().item())
    assert (dw - dw0).abs().max().item() <= 1e-4 * sc and (db - db0).abs().max().item() <= 1e-4 * sc
    dw2, db2 = ops.conv_wgrad(dyb.cuda(), 4, N, xb.cuda(), 4, C, 9, wino=True)        # fixed-order sums: bitwise reproducible
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("C,E1,E3,S,B,H,W", [
    (16, 64, 64, 16, 2, 12, 20), (16, 64, 64, 16, 6, 96, 312), (8, 32, 40, 12, 1, 5, 17), (16, 96, 48, 16, 2, 11, 23), (8, 32, 32, 32, 2, 10, 21),
    (16, 144, 32, 24, 3, 13, 50), (16, 48, 96, 32, 2, 7, 40), (32, 128, 128, 32, 2, 24, 78), (48, 192, 192, 32, 1, 9, 20),
])
def test_fire_bridge_one_launch(C, E1, E3, S, B, H, W):
    """Fire k's expand pair + concat + Fire k+1's squeeze in ONE launch (sqd_fire_bridge_fwd) == the three fp32 convolutions of
    the reference (src/model/squeezedet.py:18-22 applied twice), every configuration that can run the shape (C <= 16 register-resident
    form, U-resident, streamed), odd sizes, partial slices (E1 not a multiple of 128, E3 not of 32, S not of 16), bytes outside
    the output window untouched."""
    ops = _ops()
    x = F.relu(_rand(B, C, H, W, seed=51))
    w1 = _rand(E1, C, 1, 1, seed=52, scale=(2.0 / C) ** 0.5); b1 = _rand(E1, seed=53, scale=0.1)
    w3 = _rand(E3, C, 3, 3, seed=54, scale=(2.0 / (C * 9)) ** 0.5); b3 = _rand(E3, seed=55, scale=0.1)
    ws = _rand(S, E1 + E3, 1, 1, seed=56, scale=(2.0 / (E1 + E3)) ** 0.5); bs = _rand(S, seed=57, scale=0.1)
    mid = torch.cat([F.relu(F.conv2d(x, w1, b1)), F.relu(F.conv2d(x, w3, b3, padding=1))], 1)
    ref = _nhwc(F.relu(F.conv2d(mid, ws, bs)))
    xg = _nhwc(x).cuda()
    ran = 0
    for cid in ops.FIRE_BRIDGE_CFGS:
        if not ops.fire_bridge_cfg_ok(cid, C, E3, E1, S):
            with pytest.raises(ValueError):
                ops.FireBridgePlan(w1.cuda(), b1.cuda(), w3.cuda(), b3.cuda(), ws.cuda(), bs.cuda(), cid)
            continue
        plan = ops.FireBridgePlan(w1.cuda(), b1.cuda(), w3.cuda(), b3.cuda(), ws.cuda(), bs.cuda(), cid)
        y = torch.full((B, H, W, S + 8), -7.0, device='cuda')
        ops.fire_bridge(xg, 0, plan, y, 4)
        torch.cuda.synchronize()
        yc = y.cpu()
        assert (yc[..., 4:4 + S] - ref).abs().max().item() <= _tol(ref), f'cfg {cid}'
        assert bool((yc[..., :4] == -7.0).all()) and bool((yc[..., 4 + S:] == -7.0).all())
        ran += 1
    if ran == 0:
        pytest.skip('no bridge configuration fits this shape (the streamed-U general form is retired: measured slower than separate launches)')


@pytest.mark.parametrize("C,E1,E3,S,B,H,W", [
    (16, 64, 64, 16, 2, 12, 20), (16, 64, 64, 16, 6, 96, 312), (8, 32, 40, 12, 1, 5, 17), (8, 96, 48, 16, 2, 11, 23), (8, 32, 32, 32, 2, 10, 21),
    (16, 128, 32, 24, 3, 13, 50), (16, 48, 32, 32, 2, 7, 40), (8, 80, 20, 28, 2, 9, 33), (16, 64, 64, 16, 1, 4, 16), (16, 64, 64, 16, 1, 3, 15),
])
def test_fire_bridge_storing_form(C, E1, E3, S, B, H, W):
    """Training form of the Fire -> Fire bridge (sqd_fire_bridge_save_fwd): the squeeze output AND the concatenated expand output of
    ONE launch == the reference's modules in fp32 (src/model/squeezedet.py:18-22 twice); the stored expand output is bit-identical to
    the plain fused-expand launch of the same kernel family (sqd_fire_wino_fwd cfg 12), the squeeze output bit-identical to the
    inference bridge; odd sizes, partial channel blocks, bytes outside the three windows untouched."""
    ops = _ops()
    x = F.relu(_rand(B, C, H, W, seed=71))
    w1 = _rand(E1, C, 1, 1, seed=72, scale=(2.0 / C) ** 0.5); b1 = _rand(E1, seed=73, scale=0.1)
    w3 = _rand(E3, C, 3, 3, seed=74, scale=(2.0 / (C * 9)) ** 0.5); b3 = _rand(E3, seed=75, scale=0.1)
    ws = _rand(S, E1 + E3, 1, 1, seed=76, scale=(2.0 / (E1 + E3)) ** 0.5); bs = _rand(S, seed=77, scale=0.1)
    mid = torch.cat([F.relu(F.conv2d(x, w1, b1)), F.relu(F.conv2d(x, w3, b3, padding=1))], 1)
    ref = _nhwc(F.relu(F.conv2d(mid, ws, bs)))
    mid = _nhwc(mid)
    xg = _nhwc(x).cuda()
    assert ops.fire_bridge_cfg_ok(12, C, E3, E1, S)
    plan = ops.FireBridgePlan(w1.cuda(), b1.cuda(), w3.cuda(), b3.cuda(), ws.cuda(), bs.cuda(), 12)
    y = torch.full((B, H, W, S + 8), -7.0, device='cuda')
    sv = torch.full((B, H, W, E1 + E3 + 12), -5.0, device='cuda')
    ops.fire_bridge(xg, 0, plan, y, 4, save=sv, save_coff1=4, save_coff3=8 + E1)
    y_inf = torch.full((B, H, W, S + 8), -7.0, device='cuda')
    ops.fire_bridge(xg, 0, plan, y_inf, 4)
    torch.cuda.synchronize()
    yc, sc = y.cpu(), sv.cpu()
    assert (yc[..., 4:4 + S] - ref).abs().max().item() <= _tol(ref)
    assert torch.equal(yc, y_inf.cpu())
    assert (sc[..., 4:4 + E1] - mid[..., :E1]).abs().max().item() <= _tol(mid)
    assert (sc[..., 8 + E1:8 + E1 + E3] - mid[..., E1:]).abs().max().item() <= _tol(mid)
    assert bool((sc[..., :4] == -5.0).all()) and bool((sc[..., 4 + E1:8 + E1] == -5.0).all()) and bool((sc[..., 8 + E1 + E3:] == -5.0).all())
    with pytest.raises(ValueError):
        ops.fire_bridge(xg, 0, plan, y, 4, save=sv, save_coff1=4, save_coff3=4 + E1 - 4)          # overlapping windows
    with pytest.raises(ValueError):
        ops.fire_bridge(xg, 0, plan, y, 4, save=sv[:, :, :-1], save_coff1=0)


@pytest.mark.parametrize("C,E1,E3,S,cfg,pooled", [(16, 64, 64, 16, 12, False), (16, 64, 64, 32, 12, True), (8, 32, 40, 12, 12, False),
                                                 (8, 80, 20, 28, 12, False), (16, 64, 64, 16, 10, False)])
def test_fire_bridge_plan_refresh_in_place(C, E1, E3, S, cfg, pooled):
    """After an optimizer step the bridges' operands are rewritten IN PLACE by two batched launches (plans.refresh_bridge_plans: the
    Winograd transform of the expand3x3 part + one scaled gather for everything else): bit-identical to a plan built from scratch on
    the new parameter values, same buffers."""
    from squeezedet_pytorch_amd import plans
    ops = _ops()
    mk = lambda seed: [t.cuda() for t in (_rand(E1, C, 1, 1, seed=seed), _rand(E1, seed=seed + 1), _rand(E3, C, 3, 3, seed=seed + 2),
                                          _rand(E3, seed=seed + 3), _rand(S, E1 + E3, 1, 1, seed=seed + 4), _rand(S, seed=seed + 5))]
    old, new = mk(100), mk(200)
    plan = ops.FireBridgePlan(*old, cfg, pooled=pooled)
    ptrs = (plan.w.data_ptr(), plan.sq_ops.data_ptr(), plan.bias_tab.data_ptr(), plan.sq_bias.data_ptr())
    fresh = ops.FireBridgePlan(*new, cfg, pooled=pooled)
    assert not torch.equal(plan.w, fresh.w)
    plans.refresh_bridge_plans([(plan, *new)])
    torch.cuda.synchronize()
    for name in ('w', 'sq_ops', 'bias_tab', 'sq_bias'):
        assert torch.equal(getattr(plan, name), getattr(fresh, name)), name
    assert ptrs == (plan.w.data_ptr(), plan.sq_ops.data_ptr(), plan.bias_tab.data_ptr(), plan.sq_bias.data_ptr())


@pytest.mark.parametrize("C,E1,E3,S,B,H,W,nseg", [
    (16, 64, 64, 32, 1, 8, 32, 1), (16, 64, 64, 32, 1, 8, 32, 2), (16, 64, 64, 32, 2, 9, 37, 1), (8, 32, 40, 12, 1, 5, 17, 1),
    (16, 48, 64, 16, 2, 12, 30, 3), (16, 64, 32, 24, 3, 13, 50, 2), (8, 16, 16, 32, 1, 24, 14, 6), (16, 64, 64, 32, 3, 96, 312, 4),
    (8, 16, 8, 4, 1, 3, 3, 1), (16, 64, 64, 32, 1, 31, 45, 5), (16, 64, 64, 32, 2, 7, 100, 9),
    (16, 64, 64, 32, 4, 96, 312, 24),      # 2208 wave tasks > 2048 wave slots: waves walk on to a second task (the bs >= 32 rows do)
])
def test_fire_pool_bridge_one_launch(C, E1, E3, S, B, H, W, nseg):
    """Fire k's expand pair + concat + MaxPool2d(3, 2, ceil_mode=True) + Fire k+1's squeeze in ONE launch (sqd_fire_pool_bridge_fwd)
    == the reference's four modules in fp32 (src/model/squeezedet.py:18-22, 47-52): odd and even map sizes (windows clipped at the
    right / bottom edge), maps smaller than a group, one to many segments per strip (the carried pooled row crosses segment
    and image ends), partial channel blocks, bytes outside the output window untouched."""
    ops = _ops()
    x = F.relu(_rand(B, C, H, W, seed=61))
    w1 = _rand(E1, C, 1, 1, seed=62, scale=(2.0 / C) ** 0.5); b1 = _rand(E1, seed=63, scale=0.1)
    w3 = _rand(E3, C, 3, 3, seed=64, scale=(2.0 / (C * 9)) ** 0.5); b3 = _rand(E3, seed=65, scale=0.1)
    ws = _rand(S, E1 + E3, 1, 1, seed=66, scale=(2.0 / (E1 + E3)) ** 0.5); bs = _rand(S, seed=67, scale=0.1)
    mid = torch.cat([F.relu(F.conv2d(x, w1, b1)), F.relu(F.conv2d(x, w3, b3, padding=1))], 1)
    ref = _nhwc(F.relu(F.conv2d(F.max_pool2d(mid, 3, 2, ceil_mode=True), ws, bs)))
    assert ops.fire_pool_bridge_ok(C, E3, E1, S)
    plan = ops.FireBridgePlan(w1.cuda(), b1.cuda(), w3.cuda(), b3.cuda(), ws.cuda(), bs.cuda(), 12, pooled=True)
    Hp, Wp = ops.pool_out_size(H, W)
    assert tuple(ref.shape[:3]) == (B, Hp, Wp)
    y = torch.full((B, Hp, Wp, S + 8), -7.0, device='cuda')
    ops.fire_pool_bridge(_nhwc(x).cuda(), 0, plan, y, 4, nseg=nseg)
    torch.cuda.synchronize()
    yc = y.cpu()
    assert (yc[..., 4:4 + S] - ref).abs().max().item() <= _tol(ref)
    assert bool((yc[..., :4] == -7.0).all()) and bool((yc[..., 4 + S:] == -7.0).all())
    with pytest.raises(ValueError):
        ops.fire_bridge(_nhwc(x).cuda(), 0, plan, torch.empty(B, H, W, S, device='cuda'), 0)      # a pooled plan is not a plain bridge plan
    assert not ops.fire_pool_bridge_ok(32, E3, E1, S) and not ops.fire_pool_bridge_ok(C, E3, 128, S)


@pytest.mark.parametrize("C,E1,E3,S,B,H,W,nseg", [
    (16, 64, 64, 32, 1, 8, 32, 1), (16, 64, 64, 32, 1, 8, 32, 2), (16, 64, 64, 32, 2, 9, 37, 1), (8, 32, 40, 12, 1, 5, 17, 1),
    (16, 48, 64, 16, 2, 12, 30, 3), (16, 64, 32, 24, 3, 13, 50, 2), (8, 16, 16, 32, 1, 24, 14, 6), (16, 64, 64, 32, 3, 96, 312, 4),
    (8, 16, 8, 4, 1, 3, 3, 1), (16, 64, 64, 32, 1, 31, 45, 5), (16, 64, 64, 32, 2, 7, 100, 9), (16, 64, 64, 32, 4, 96, 312, 24),
])
def test_fire_pool_bridge_storing_form(C, E1, E3, S, B, H, W, nseg):
    """Training form of the Fire -> pool -> Fire bridge (sqd_fire_pool_bridge_save_fwd).  The squeeze output is bit-identical to the
    inference bridge; the stored pooled tensor and the arg-max / ReLU codes are BIT-identical to the max-pool kernel of the unfused
    training forward (ops.maxpool(relu_codes=True)) applied to the expand output of the same kernel family (ops.fire_wino cfg 12: the
    same matrix-core summation order), i.e. first window position holding the maximum, 15 where it is not > 0, clipped windows at
    the right / bottom edge, segments and carried rows included; the pooled values also equal the reference's modules in fp32
    (src/model/squeezedet.py:18-22, 47-52); bytes outside the windows untouched."""
    ops = _ops()
    x = F.relu(_rand(B, C, H, W, seed=61))
    w1 = _rand(E1, C, 1, 1, seed=62, scale=(2.0 / C) ** 0.5); b1 = _rand(E1, seed=63, scale=0.1)
    w3 = _rand(E3, C, 3, 3, seed=64, scale=(2.0 / (C * 9)) ** 0.5); b3 = _rand(E3, seed=65, scale=0.1)
    ws = _rand(S, E1 + E3, 1, 1, seed=66, scale=(2.0 / (E1 + E3)) ** 0.5); bs = _rand(S, seed=67, scale=0.1)
    mid = torch.cat([F.relu(F.conv2d(x, w1, b1)), F.relu(F.conv2d(x, w3, b3, padding=1))], 1)
    pref = _nhwc(F.max_pool2d(mid, 3, 2, ceil_mode=True))
    plan = ops.FireBridgePlan(w1.cuda(), b1.cuda(), w3.cuda(), b3.cuda(), ws.cuda(), bs.cuda(), 12, pooled=True)
    Hp, Wp = ops.pool_out_size(H, W)
    xg = _nhwc(x).cuda()
    y = torch.full((B, Hp, Wp, S + 8), -7.0, device='cuda')
    sv = torch.full((B, Hp, Wp, E1 + E3 + 8), -5.0, device='cuda')
    cd = torch.full((B, Hp, Wp, E1 + E3 + 8), 99, dtype=torch.uint8, device='cuda')
    ops.fire_pool_bridge(xg, 0, plan, y, 4, nseg=nseg, save=sv, codes=cd, save_coff1=4, save_coff3=4 + E1)
    y_inf = torch.full((B, Hp, Wp, S + 8), -7.0, device='cuda')
    ops.fire_pool_bridge(xg, 0, plan, y_inf, 4, nseg=nseg)
    # the unfused training forward on the same expand arithmetic
    out = torch.empty(B, H, W, E1 + E3, device='cuda')
    ops.fire_wino(xg, 0, ops.FireWinoPlan(w1.cuda(), b1.cuda(), w3.cuda(), b3.cuda(), 12), out, 0, E1)
    am = torch.empty(B, Hp, Wp, E1 + E3, dtype=torch.uint8, device='cuda')
    pooled = ops.maxpool(out, argmax=am, relu_codes=True)
    torch.cuda.synchronize()
    assert torch.equal(y, y_inf)
    assert torch.equal(sv[..., 4:4 + E1 + E3], pooled)
    assert torch.equal(cd[..., 4:4 + E1 + E3], am)
    assert (sv[..., 4:4 + E1 + E3].cpu() - pref).abs().max().item() <= _tol(pref)
    assert bool((sv[..., :4] == -5.0).all()) and bool((sv[..., 4 + E1 + E3:] == -5.0).all())
    assert bool((cd[..., :4] == 99).all()) and bool((cd[..., 4 + E1 + E3:] == 99).all())
    with pytest.raises(ValueError):
        ops.fire_pool_bridge(xg, 0, plan, y, 4, nseg=nseg, save=sv)                                   # save without codes


def test_fire_pool_bridge_codes_on_exact_ties():
    """Every window position equal (zero weights, the outputs are the biases): the code is the FIRST position (0) where the bias is
    positive and 15 where it is not -- the tie rule of MaxPool2d's CPU scan that the backward relies on."""
    ops = _ops()
    C, E1, E3, S, B, H, W = 16, 64, 64, 32, 2, 13, 31
    x = F.relu(_rand(B, C, H, W, seed=81))
    z1, z3 = torch.zeros(E1, C, 1, 1), torch.zeros(E3, C, 3, 3)
    b1 = _rand(E1, seed=82); b3 = _rand(E3, seed=83)
    ws = _rand(S, E1 + E3, 1, 1, seed=84); bs = _rand(S, seed=85)
    plan = ops.FireBridgePlan(z1.cuda(), b1.cuda(), z3.cuda(), b3.cuda(), ws.cuda(), bs.cuda(), 12, pooled=True)
    Hp, Wp = ops.pool_out_size(H, W)
    y = torch.empty(B, Hp, Wp, S, device='cuda')
    sv = torch.empty(B, Hp, Wp, E1 + E3, device='cuda')
    cd = torch.empty(B, Hp, Wp, E1 + E3, dtype=torch.uint8, device='cuda')
    ops.fire_pool_bridge(_nhwc(x).cuda(), 0, plan, y, 0, nseg=2, save=sv, codes=cd)
    torch.cuda.synchronize()
    bias = torch.cat([b1, b3])
    want = torch.where(bias > 0, torch.zeros(E1 + E3, dtype=torch.uint8), torch.full((E1 + E3,), 15, dtype=torch.uint8))
    assert torch.equal(cd.cpu(), want.view(1, 1, 1, -1).expand(B, Hp, Wp, -1))
    assert torch.equal(sv.cpu(), F.relu(bias).view(1, 1, 1, -1).expand(B, Hp, Wp, -1))


@pytest.mark.parametrize("C,E1,E3,B,H,W", [
    (16, 64, 64, 2, 12, 20), (32, 128, 128, 1, 9, 33), (48, 192, 192, 1, 24, 78), (64, 256, 256, 2, 6, 18), (96, 384, 384, 1, 5, 17),
    (16, 64, 64, 6, 96, 312), (96, 384, 384, 20, 24, 78), (8, 16, 20, 3, 4, 16), (24, 48, 40, 2, 7, 35), (16, 96, 48, 2, 11, 23), (8, 32, 64, 1, 5, 50),
])
def test_fire_expand_winograd_one_launch(C, E1, E3, B, H, W):
    """Fire's expand pair in ONE Winograd launch (expand1x1 as the four inner transform positions): both halves of the concat
    == the two fp32 convolutions of the reference (src/model/squeezedet.py:18-22), every usable configuration, odd sizes,
    partial expand1x1 slices (E1 not a multiple of 128), bytes outside the two windows untouched; and bit-identical to the
    separate Winograd launch for the expand3x3 half (same arithmetic)."""
    ops = _ops()
    x = F.relu(_rand(B, C, H, W, seed=41))
    w1 = _rand(E1, C, 1, 1, seed=42, scale=(2.0 / C) ** 0.5); b1 = _rand(E1, seed=43, scale=0.1)
    w3 = _rand(E3, C, 3, 3, seed=44, scale=(2.0 / (C * 9)) ** 0.5); b3 = _rand(E3, seed=45, scale=0.1)
    ref1 = _nhwc(F.relu(F.conv2d(x, w1, b1)))
    ref3 = _nhwc(F.relu(F.conv2d(x, w3, b3, padding=1)))
    xg = _nhwc(x).cuda()
    ran = 0
    for cid in ops.FIRE_WINO_CFGS + (1008, 1010):
        if not ops.fire_wino_cfg_ok(cid, C, E1, E3):
            continue
        plan = ops.FireWinoPlan(w1.cuda(), b1.cuda(), w3.cuda(), b3.cuda(), cid)
        y = torch.full((B, H, W, E1 + E3 + 8), -7.0, device='cuda')
        ops.fire_wino(xg, 0, plan, y, 4, 4 + E1)
        torch.cuda.synchronize()
        yc = y.cpu()
        assert (yc[..., 4:4 + E1] - ref1).abs().max().item() <= _tol(ref1), f'cfg {cid} expand1x1'
        assert (yc[..., 4 + E1:4 + E1 + E3] - ref3).abs().max().item() <= _tol(ref3), f'cfg {cid} expand3x3'
        assert bool((yc[..., :4] == -7.0).all()) and bool((yc[..., 4 + E1 + E3:] == -7.0).all())
        if C % 8 == 0 and cid < 12:
            y3 = torch.empty(B, H, W, E3, device='cuda')
            ops.conv_wino(xg, 0, ops.WinoPlan(w3.cuda(), b3.cuda(), cid), y3, 0, relu=True)
            assert torch.equal(y3.cpu(), yc[..., 4 + E1:4 + E1 + E3])
        ran += 1
    if ran == 0:
        pytest.skip('no one-launch Winograd Fire configuration fits this squeeze width (the streamed-U ids are retired)')
