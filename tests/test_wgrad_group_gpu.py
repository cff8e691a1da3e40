"""GPU tier: the grouped Winograd weight-gradient launch (csrc/wino_wgrad.hip ``wino_wgrad_group_kernel``; reference: autograd of the
Fire expand3x3 convolutions, src/model/squeezedet.py:14,20-22, as triggered by loss.backward(), src/engine/trainer.py:47) writes, for every
member layer, bit for bit the slabs of that layer's own launch with the same number of splits -- and the reduced gradients agree with a
float64 convolution gradient."""
import numpy as np
import pytest
import torch

from squeezedet_pytorch_amd import ops, tiles

pytestmark = pytest.mark.gpu


def _layer(B, H, W, C, N, seed, pitch_extra=0, coff=0):
    g = torch.Generator().manual_seed(seed)
    dy = torch.randn(B, H, W, coff + N + pitch_extra, generator=g).cuda()
    x = torch.randn(B, H, W, C, generator=g).cuda()
    return dy, coff, N, x, 0, C


@pytest.mark.parametrize("shapes,S", [
    ([(64, 256), (64, 256), (96, 384), (96, 384)], 9),         # fire11..14 of SqueezeDet (tile form 64 x 32)
    ([(48, 192), (48, 192)], 28),                              # fire9 / fire10 (64 x 16)
    ([(32, 128), (32, 128)], 5),
    ([(16, 64), (16, 64), (16, 64)], 7),
    ([(64, 256), (32, 64)], 3),                                # unequal members
])
def test_group_slabs_are_bitwise_the_single_launch_slabs(shapes, S):
    B, H, W = 3, 22, 45                                        # ragged: 22 = 5.5 row groups, 45 = 2.8 column groups
    tc = tiles._wino_wgrad_tc(shapes[0][1], shapes[0][0])
    items, want = [], []
    for k, (C, N) in enumerate(shapes):
        assert tiles._wino_wgrad_tc(N, C) == tc
        dy, dyc, N, x, xc, C = _layer(B, H, W, C, N, seed=10 + k, pitch_extra=8 if k % 2 else 0, coff=4 * k)
        stride = N * 9 * C + N
        slab = torch.full((S * stride,), float('nan'), device='cuda')
        items.append((dy, dyc, N, x, xc, C, slab))
        ref = torch.full((S * stride,), float('nan'), device='cuda')
        rc = ops.nat.lib().sqd_conv_wgrad_wino(ops.nat.ptr(dy), ops.nat.ptr(x), ops.nat.ptr(ref), None, None, B, H, W, N, dy.shape[3], dyc,
                                               C, x.shape[3], xc, S, tc, ops.nat.stream_handle(dy.device))
        ops.nat.check(rc, 'sqd_conv_wgrad_wino')
        want.append(ref)
    ops.conv_wgrad_wino_group(items, S, tc)
    torch.cuda.synchronize()
    for (dy, dyc, N, x, xc, C, slab), ref in zip(items, want):
        assert not torch.isnan(slab).any()
        assert torch.equal(slab, ref), (C, N)
        # and the sum of the slabs is the convolution's weight gradient (float64 reference; 3x3, pad 1)
        stride = N * 9 * C + N
        tot = slab.view(S, stride).double().sum(0).cpu()
        dw = tot[:N * 9 * C].view(N, 3, 3, C).permute(0, 3, 1, 2)
        xd = x[..., xc:xc + C].double().cpu().permute(0, 3, 1, 2)
        dyd = dy[..., dyc:dyc + N].double().cpu().permute(0, 3, 1, 2)
        ref_dw = torch.nn.grad.conv2d_weight(xd, (N, C, 3, 3), dyd, padding=1)
        scale = float(ref_dw.abs().max())
        assert float((dw - ref_dw).abs().max()) <= 1e-4 * max(1.0, scale)
        np.testing.assert_allclose(tot[N * 9 * C:].numpy(), dyd.sum((0, 2, 3)).numpy(), rtol=0, atol=1e-4 * max(1.0, float(dyd.sum((0, 2, 3)).abs().max())))


def test_group_rejects_mixed_tile_forms_and_bad_layers():
    B, H, W = 1, 8, 16
    dy, dyc, N, x, xc, C = _layer(B, H, W, 64, 64, 1)
    slab = torch.empty(2 * (N * 9 * C + N), device='cuda')
    with pytest.raises(ValueError):
        ops.conv_wgrad_wino_group([(dy, dyc, N, x, xc, C, slab)], 2, 1)          # C = 64 wants the 32-channel form
    import ctypes
    rows = (ctypes.c_longlong * 9)(dy.data_ptr(), x.data_ptr(), slab.data_ptr(), 72, 72, 0, 64, 64, 0)
    rc = ops.nat.lib().sqd_conv_wgrad_wino_group(ctypes.cast(rows, ctypes.c_void_p), 1, B, H, W, 2, 2, None)
    assert rc == 2, rc                                                            # SQD_ERR_UNSUPPORTED: N = 72 (ConvDet's 5-block form) has its own launch
    rc = ops.nat.lib().sqd_conv_wgrad_wino_group(ctypes.cast(rows, ctypes.c_void_p), 7, B, H, W, 2, 2, None)
    assert rc == 1, rc                                                            # SQD_ERR_ARG: more layers than a launch carries


@pytest.mark.parametrize("shapes,S", [
    ([(64, 256), (64, 256), (96, 384), (96, 384)], 16),        # fire11..14 expand1x1 of SqueezeDet (64-channel in-tiles)
    ([(48, 192), (48, 192)], 28),                              # fire9 / fire10 (48-channel in-tiles)
    ([(256, 128), (300, 64)], 3),                              # 128-channel in-tiles, a partial last tile
    ([(16, 64), (12, 128), (16, 100)], 5),                     # 16-channel in-tiles, N off the 64 grid
])
def test_1x1_group_slabs_are_bitwise_the_single_launch_slabs(shapes, S):
    """``sqd_conv_wgrad_group`` (the wide expand1x1 layers of a stage in one launch) against ``sqd_conv_wgrad`` per layer with the same
    number of splits, and the slab sums against a float64 GEMM."""
    B, H, W = 3, 22, 45
    items, want = [], []
    for k, (C, N) in enumerate(shapes):
        dy, dyc, N, x, xc, C = _layer(B, H, W, C, N, seed=40 + k, pitch_extra=8 if k % 2 else 0, coff=4 * k)
        stride = N * C + N
        slab = torch.full((S * stride,), float('nan'), device='cuda')
        items.append((dy, dyc, N, x, xc, C, slab))
        ref = torch.full((S * stride,), float('nan'), device='cuda')
        rc = ops.nat.lib().sqd_conv_wgrad(ops.nat.ptr(dy), ops.nat.ptr(x), ops.nat.ptr(ref), None, None, B, H, W, N, dy.shape[3], dyc,
                                          C, x.shape[3], xc, 1, S, ops.nat.stream_handle(dy.device))
        ops.nat.check(rc, 'sqd_conv_wgrad')
        want.append(ref)
    ops.conv_wgrad_group(items, S)
    torch.cuda.synchronize()
    for (dy, dyc, N, x, xc, C, slab), ref in zip(items, want):
        assert not torch.isnan(slab).any()
        assert torch.equal(slab, ref), (C, N)
        tot = slab.view(S, N * C + N).double().sum(0).cpu()
        dyd = dy[..., dyc:dyc + N].double().cpu().reshape(-1, N); xd = x[..., xc:xc + C].double().cpu().reshape(-1, C)
        ref_dw = dyd.t() @ xd
        assert float((tot[:N * C].view(N, C) - ref_dw).abs().max()) <= 1e-4 * max(1.0, float(ref_dw.abs().max()))
        assert float((tot[N * C:] - dyd.sum(0)).abs().max()) <= 1e-4 * max(1.0, float(dyd.sum(0).abs().max()))


def test_1x1_group_rejects_other_tile_forms():
    import ctypes
    B, H, W = 1, 8, 16
    dy, dyc, N, x, xc, C = _layer(B, H, W, 64, 96, 1)                            # N = 96 runs the 6-tile form: its own launch
    slab = torch.empty(2 * (N * C + N), device='cuda')
    rows = (ctypes.c_longlong * 9)(dy.data_ptr(), x.data_ptr(), slab.data_ptr(), N, N, 0, C, C, 0)
    assert ops.nat.lib().sqd_conv_wgrad_group(ctypes.cast(rows, ctypes.c_void_p), 1, B, H, W, 2, None) == 2
    dy2, _, N2, x2, _, C2 = _layer(B, H, W, 48, 128, 2)
    slab2 = torch.empty(2 * (N2 * C2 + N2), device='cuda')
    dy1, _, N1, x1, _, C1 = _layer(B, H, W, 64, 128, 3)
    slab1 = torch.empty(2 * (N1 * C1 + N1), device='cuda')
    rows = (ctypes.c_longlong * 18)(dy1.data_ptr(), x1.data_ptr(), slab1.data_ptr(), N1, N1, 0, C1, C1, 0,
                                    dy2.data_ptr(), x2.data_ptr(), slab2.data_ptr(), N2, N2, 0, C2, C2, 0)
    assert ops.nat.lib().sqd_conv_wgrad_group(ctypes.cast(rows, ctypes.c_void_p), 2, B, H, W, 2, None) == 2      # 64- and 48-channel in-tiles do not mix
    assert ops.nat.lib().sqd_conv_wgrad_group(ctypes.cast(rows, ctypes.c_void_p), 1, B, H, W, 100, None) == 2    # more splits than pixel blocks
