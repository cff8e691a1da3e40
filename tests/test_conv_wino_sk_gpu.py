"""GPU tier: the balanced (stream-K) Winograd convolution, csrc/conv_wino_sk.hip (sqd_conv_wino_sk_fwd) -- the 3x3 layers of the
reference (Fire expand3x3 src/model/squeezedet.py:14,20-22; ConvDet :73-75,83; their data gradients) -- against fp32 conv2d on the
CPU at the 1e-4 bound of the other kernels, through the C ABI.  The grid size is forced (SQD_SK_GRID) so that units are cut into
1, 2..3 and many parts: every grid must give the same result up to the summation order of cut units, and the same grid the same bits."""
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


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _tol(ref):
    return TOL * max(1.0, float(ref.abs().max()))


def _counters_clean(ops, ngroups, N, C):
    from squeezedet_pytorch_amd import plans
    sk = plans.wino_sk_schedule(ngroups, N, C, torch.device('cuda', torch.cuda.current_device()))
    torch.cuda.synchronize()
    return int(sk.cnt.abs().sum()) == 0, sk


@pytest.mark.parametrize('grid', [0, 2, 7, 61])
@pytest.mark.parametrize('C,N,B,H,W', [
    (768, 72, 2, 24, 78),       # ConvDet: both workgroup classes, every unit cut (many parts on the full grid)
    (96, 384, 1, 24, 78),       # fire13/14 expand3x3
    (48, 192, 2, 9, 33),        # partial groups on both axes, short K
    (72, 768, 1, 6, 18),        # ConvDet's data-gradient orientation (9 chunks, 24 slices)
    (16, 80, 3, 5, 17),         # N = 80: the 16-channel class with all 16 channels real
    (8, 16, 1, 3, 3), (24, 20, 2, 7, 35), (64, 48, 7, 2, 2),
])
def test_conv_wino_sk_matches_conv2d(C, N, B, H, W, grid, monkeypatch):
    ops = _ops()
    if grid:
        monkeypatch.setenv('SQD_SK_GRID', str(grid))
    x = _rand(B, C, H, W, seed=31)
    w = _rand(N, C, 3, 3, seed=32, scale=(2.0 / (C * 9)) ** 0.5)
    b = _rand(N, seed=33, scale=0.1)
    ref = _nhwc(F.relu(F.conv2d(x, w, b, padding=1)))
    xg = _nhwc(x).cuda()
    plan = ops.WinoPlan(w.cuda(), b.cuda(), ops.WINO_SK_CFG)
    y = torch.full((B, H, W, N + 8), float('nan'), device='cuda')
    ops.conv_wino(xg, 0, plan, y, 4, relu=True)
    ok, sk = _counters_clean(ops, B * -(-H // 4) * -(-W // 16), N, C)
    assert ok, 'arrival counters must return to zero'
    got = y.cpu()
    assert torch.isnan(got[..., :4]).all() and torch.isnan(got[..., 4 + N:]).all(), 'bytes outside the channel window were written'
    err = (got[..., 4:4 + N] - ref).abs().max().item()
    assert err <= _tol(ref), f'max err {err} (grid {sk.G}, {sk.nslabs} slabs)'
    # bitwise reproducible: the parts of a cut unit are added in part order whatever the arrival order
    for _ in range(3):
        y2 = torch.full_like(y, float('nan'))
        ops.conv_wino(xg, 0, plan, y2, 4, relu=True)
        assert torch.equal(y2[..., 4:4 + N], y[..., 4:4 + N])


def test_conv_wino_sk_dgrad_epilogue():
    """The epilogue options of the backward: accumulate into y, per-element multiplier (ymul), constant factor (yscale: the dropout
    scale once the keep mask is folded into the ReLU mask) and ReLU-backward mask (ymask), on the data-gradient packing."""
    ops = _ops()
    B, H, W, C, N = 2, 9, 21, 64, 96
    dy = _rand(B, C, H, W, seed=21)
    w = _rand(C, N, 3, 3, seed=22, scale=0.1)               # forward weight of a conv N -> C; its dgrad maps C -> N
    ref = F.conv_transpose2d(dy, w, None, padding=1)
    y0 = _rand(B, N, H, W, seed=23)
    mul = _rand(B, N, H, W, seed=24).abs() + 0.5
    mask = _rand(B, N, H, W, seed=25)
    plan = ops.WinoPlan(w.cuda(), None, ops.WINO_SK_CFG, dgrad=True)
    exp = (y0 + ref) * mul * 2.0 * (mask > 0)
    y = _nhwc(y0).cuda()
    ops.conv_wino(_nhwc(dy).cuda(), 0, plan, y, 0, accumulate=True, ymul=_nhwc(mul).cuda(), ymask=_nhwc(mask).cuda(), yscale=2.0)
    assert (y.cpu() - _nhwc(exp)).abs().max().item() <= _tol(exp)
    exp2 = ref * 1.5 * (mask > 0)
    y = torch.full((B, H, W, N), float('nan'), device='cuda')
    ops.conv_wino(_nhwc(dy).cuda(), 0, plan, y, 0, ymask=_nhwc(mask).cuda(), yscale=1.5)
    assert (y.cpu() - _nhwc(exp2)).abs().max().item() <= _tol(exp2)
    with pytest.raises(ValueError):
        ops.conv_wino(_nhwc(dy).cuda(), 0, ops.WinoPlan(w.cuda(), None, 2, dgrad=True), y, 0, yscale=1.5)   # only the balanced kernel scales


@pytest.mark.parametrize('C,N', [(768, 72), (96, 384), (48, 192), (64, 256)])
def test_conv_wino_sk_headline_shapes_vs_unit_kernel(C, N):
    """bs=20 24x78 (the shapes bench.py times): the balanced kernel against conv_wino<2,4> on the same operands -- equal to fp32
    rounding (whole units are the same arithmetic bit for bit; cut units add their K ranges in a different order) -- and against
    fp32 conv2d on the CPU for the first and last image."""
    ops = _ops()
    B, H, W = 20, 24, 78
    x = _rand(B, C, H, W, seed=41)
    w = _rand(N, C, 3, 3, seed=42, scale=(2.0 / (C * 9)) ** 0.5)
    b = _rand(N, seed=43, scale=0.1)
    xg = _nhwc(x).cuda()
    y_sk = torch.full((B, H, W, N), float('nan'), device='cuda')
    y_un = torch.full((B, H, W, N), float('nan'), device='cuda')
    ops.conv_wino(xg, 0, ops.WinoPlan(w.cuda(), b.cuda(), ops.WINO_SK_CFG), y_sk, 0, relu=False)
    ops.conv_wino(xg, 0, ops.WinoPlan(w.cuda(), b.cuda(), 2), y_un, 0, relu=False)
    ref = _nhwc(F.conv2d(x[[0, B - 1]], w, b, padding=1))
    tol = _tol(ref)
    assert (y_sk.cpu()[[0, B - 1]] - ref).abs().max().item() <= tol
    assert (y_sk - y_un).abs().max().item() <= tol
    ok, sk = _counters_clean(ops, B * 6 * 5, N, C)
    assert ok and sk.G == 2 * torch.cuda.get_device_properties(0).multi_processor_count
