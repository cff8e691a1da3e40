"""GPU tier: the V-shared Winograd kernel for narrow outputs, csrc/conv_wino_vs.hip (sqd_conv_wino_vs_fwd; ConvDet, reference
Conv2d(768 -> 72, 3, padding 1), src/model/squeezedet.py:73-75,83) -- bit for bit against conv_wino_kernel<2,4> (same transforms, same
k order) and within the 1e-4 bound of fp32 conv2d, through the C ABI: workgroups that cut groups, single-wave group slots, idle
waves in the last workgroup, image borders, channel windows inside wider buffers, every N that is a multiple of 4 up to 80."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(C, N, B, H, W, relu, pitch_extra=0, seed=0):
    from squeezedet_pytorch_amd import ops
    rs = np.random.RandomState(seed + C + N)
    x = torch.from_numpy(rs.standard_normal((B, H, W, C + pitch_extra)).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.standard_normal((N, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).astype(np.float32)).cuda()
    b = torch.from_numpy((rs.standard_normal(N) * 0.1).astype(np.float32)).cuda()
    xo = pitch_extra // 2 // 4 * 4
    p2, p17 = ops.WinoPlan(w, b, 2), ops.WinoPlan(w, b, ops.WINO_VS_CFG)
    assert p17.Npad == 80
    y2 = torch.full((B, H, W, N + 8), 7.0, device='cuda')
    y17 = torch.full((B, H, W, N + 8), 7.0, device='cuda')
    ops.conv_wino(x, xo, p2, y2, 4, relu=relu)
    for _ in range(3):                                    # (a race would show as run-to-run differences)
        y17.fill_(7.0)
        ops.conv_wino(x, xo, p17, y17, 4, relu=relu)
        torch.cuda.synchronize()
        assert torch.equal(y2, y17)
    ref = F.conv2d(x[..., xo:xo + C].permute(0, 3, 1, 2).cpu(), w.cpu(), b.cpu(), padding=1)
    ref = (ref.relu() if relu else ref).permute(0, 2, 3, 1)
    assert (y17[..., 4:4 + N].cpu() - ref).abs().max().item() <= 1e-4 * max(1.0, float(ref.abs().max()))
    assert bool((y17[..., :4] == 7).all()) and bool((y17[..., 4 + N:] == 7).all())


@pytest.mark.parametrize('C,N,B,H,W,relu', [
    (768, 72, 2, 24, 78, False),      # ConvDet: 60 groups = 300 units = 25 whole workgroups
    (512, 72, 1, 24, 78, False),      # squeezedetplus' ConvDet
    (16, 72, 3, 5, 17, True),         # partial groups on both axes
    (8, 80, 1, 3, 3, True),           # one chunk, one group, all five blocks real
    (24, 20, 2, 7, 35, False),        # N = 20: block 1 partial, blocks 2..4 idle
    (64, 48, 5, 2, 2, True),          # maps smaller than a group
    (40, 4, 1, 9, 33, False),         # N = 4
    (96, 16, 2, 11, 50, True),
    (768, 72, 1, 1, 1, False),        # a single pixel
    (32, 72, 7, 4, 16, True),         # 35 units: idle wave in the last workgroup
    (48, 72, 2, 8, 32, False),        # 8 groups: the last one is a single-wave slot of workgroup 2 (it transforms every chunk)
])
def test_conv_wino_vs_equals_conv_wino_bitwise(C, N, B, H, W, relu):
    _case(C, N, B, H, W, relu)


@pytest.mark.parametrize('pitch_extra', [16, 40])
def test_conv_wino_vs_channel_window_inside_wider_input(pitch_extra):
    _case(48, 72, 2, 6, 20, True, pitch_extra=pitch_extra)
    _case(768, 72, 2, 24, 78, False, pitch_extra=pitch_extra, seed=3)


def test_conv_wino_vs_refuses_what_it_cannot_run():
    from squeezedet_pytorch_amd import ops, _native as nat
    w = torch.randn(96, 16, 3, 3, device='cuda'); b = torch.zeros(96, device='cuda')
    x = torch.randn(1, 4, 16, 16, device='cuda'); y = torch.empty(1, 4, 16, 96, device='cuda')
    with pytest.raises(ValueError):
        ops.conv_wino(x, 0, ops.WinoPlan(w, b, ops.WINO_VS_CFG), y, 0)                    # N = 96 > 80
    w = torch.randn(72, 16, 3, 3, device='cuda'); b = torch.zeros(72, device='cuda')
    plan = ops.WinoPlan(w, b, ops.WINO_VS_CFG)
    y = torch.empty(1, 4, 16, 72, device='cuda')
    with pytest.raises(ValueError):
        ops.conv_wino(x, 0, plan, y, 0, accumulate=True)                                  # plain epilogue only
    null = nat.c_p(0)
    assert nat.lib().sqd_conv_wino_vs_fwd(null, null, null, null, 1, 4, 16, 16, 16, 0, 72, 80, 72, 0, 0, null) == 1
    assert nat.lib().sqd_conv_wino_vs_fwd(nat.ptr(x), nat.ptr(plan.w), null, nat.ptr(y), 1, 4, 16, 16, 16, 0, 72, 96, 72, 0, 0, null) == 1   # Npad != 80


def test_table_sends_convdet_to_the_v_shared_kernel_only_where_its_grid_fills_the_chip():
    from squeezedet_pytorch_amd import ops, tiles
    assert ops.choose_wino_cfg(768, 72, 20 * 24 * 78) == tiles.WINO_VS_CFG                # measured row
    assert ops.choose_wino_cfg(512, 72, 16 * 24 * 78) == tiles.WINO_VS_CFG
    assert ops.choose_wino_cfg(768, 72, 8 * 24 * 78) != tiles.WINO_VS_CFG                 # 100 workgroups on 256 CUs: the slice kernel
    assert ops.choose_wino_cfg(768, 72, 40 * 24 * 78) == tiles.WINO_VS_CFG                # two full rounds
    assert tiles.wino_vs_fills_chip(20 * 24 * 78) and not tiles.wino_vs_fills_chip(24 * 24 * 78)
