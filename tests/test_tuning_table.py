"""Host logic around the measured tile table (no GPU needed: the configuration list comes from host-side entry points of
the C-ABI library)."""
import json
import os

import pytest

from squeezedet_pytorch_amd import ops
from squeezedet_pytorch_amd.synthetic import layer_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def table():
    with open(os.path.join(ROOT, "squeezedet-pytorch_amd", "tuning.json")) as f:
        return json.load(f)


def test_every_tuned_entry_names_a_compiled_configuration(table):
    tab = ops.cfg_table()
    assert len(tab) >= 70
    for key, v in table.items():
        cid = int(v["cfg"])
        if key.startswith("W:"):                                                  # Winograd family: its own configuration list
            _, C, N, npix = key.split(":")
            assert cid % 1000 in ops.wino_cfgs() and cid // 1000 <= 8 and int(C) % 8 == 0, key
            assert v["us"] > 0 and "direct_us" in v
            continue
        if key.startswith("G:"):                                                  # split-K slabs of a weight gradient
            assert cid >= 1 and v["us"] > 0, key
            continue
        if key.startswith(("Y:", "Z:")):                                          # Fire bridges: Y = configuration id, Z = segments per strip
            _, C, E1, E3, Nsq, npix = key.split(":")
            if key.startswith("Y:"):
                assert ops.fire_bridge_cfg_ok(cid, int(C), int(E3), int(E1), int(Nsq)), key
            else:
                assert cid >= 1 and ops.fire_pool_bridge_ok(int(C), int(E3), int(E1), int(Nsq)), key
            assert v["us"] > 0 and v["separate_us"] > v["us"], key                 # only kept where the one launch wins
            continue
        assert 0 <= cid % 1000 < len(tab) and cid // 1000 <= 8, key
        taps = tab[cid % 1000][0]
        if key.startswith("F:"):
            _, C, E, npix = key.split(":")
            assert cid % 1000 in ops.fused_expand_cfgs(int(E)), key          # 3x3 LDS-DMA tiling, even group count, slice | 2E
            assert v["us"] > 0 and "separate_us" in v
        else:
            t, C, N, npix = (int(x) for x in key.split(":"))
            assert taps == t, key
            assert v["us"] > 0 and v["tflops"] > 0


def test_headline_workload_is_fully_tuned(table):
    """Every conv of SqueezeDet at bs=20, 1248x384 (forward and data-gradient orientation) has a measured entry."""
    B, h, w = 20, 192, 624                 # after the stride-2 stem
    layers = layer_table("squeezedet")
    missing = []
    for l in layers[2:]:
        if l[0] == "pool":
            h, w = ops.pool_out_size(h, w)
            continue
        _, cin, s, e1, e3 = l
        npix = B * h * w
        for taps, C, N in ((1, cin, s), (1, s, e1), (9, s, e3), (1, s, cin), (1, e1, s), (9, e3, s)):
            if f"{taps}:{C}:{N}:{npix}" not in table:
                missing.append((taps, C, N, npix))
    assert not missing, missing
    assert f"9:768:72:{20 * 24 * 78}" in table and f"9:72:768:{20 * 24 * 78}" in table


def test_choose_cfg_fallbacks():
    tab = ops.cfg_table()
    # exact hit
    exact = ops.choose_cfg(9, 96, 384, 20 * 24 * 78)
    assert tab[exact % 1000][0] == 9
    # same layer at bs=16: nearest measured shape, without the workgroup cap that was measured for the other grid size
    near = ops.choose_cfg(9, 96, 384, 16 * 24 * 78)
    assert near < 1000 and tab[near][0] == 9 and ops.cfg_is_dma(near)
    # unknown layer: heuristic over the LDS-DMA 4-wave family; staged=True asks for the register-staged family
    h = ops.choose_cfg(9, 20, 40, 5000)
    assert ops.cfg_is_dma(h) and tab[h][0] == 9
    st = ops.choose_cfg(9, 20, 40, 5000, staged=True)
    assert not ops.cfg_is_dma(st) and tab[st][0] == 9
    assert ops.cfg_kernel_name(h).startswith("conv_dma<9,") and ops.cfg_kernel_name(st).startswith("conv_igemm<9,")
    # fused expand: measured slower on this shape -> None (two separate launches); unknown half-width not a multiple of 16 -> None
    assert ops.choose_fused_cfg(96, 384, 20 * 24 * 78) is None
    assert ops.choose_fused_cfg(16, 24, 1000) is None
    # every measured F: entry: fused where it beat the two separate launches (expand1x1 + the faster of direct / Winograd 3x3)
    with open(os.path.join(ROOT, "squeezedet-pytorch_amd", "tuning.json")) as fh:
        for key, v in json.load(fh).items():
            if key.startswith("F:"):
                _, C, E, npix = key.split(":")
                f = ops.choose_fused_cfg(int(C), int(E), int(npix))
                if v["separate_us"] and v["us"] >= v["separate_us"]:
                    assert f is None, key
                else:
                    assert f is not None and f % 1000 in ops.fused_expand_cfgs(int(E)), key
    assert ops.pool_squeeze_ok(128, 32) and ops.pool_squeeze_ok(256, 48) and not ops.pool_squeeze_ok(256, 192)


def test_winograd_choice_follows_the_measured_comparison(table):
    """W: entries win only where they were measured faster than the best direct configuration of the same shape."""
    for key, v in table.items():
        if not key.startswith("W:"):
            continue
        _, C, N, npix = key.split(":")
        got = ops.choose_wino_cfg(int(C), int(N), int(npix))
        if v["direct_us"] and v["us"] >= v["direct_us"]:
            assert got is None, key
        else:
            assert got == int(v["cfg"]), key
    assert ops.choose_wino_cfg(20, 64, 1000) is None                              # C % 8 != 0: direct kernel only
    assert ops.choose_wino_cfg(24, 40, 1234) is None                              # never measured: direct kernel


def test_wgrad_split_plans_one_resident_round_for_winograd_layers():
    """Host logic of the weight-gradient split (no GPU): every 3x3 layer of both architectures takes the Winograd kernel,
    its S x channel-block grid is one resident round (<= 512 workgroups), S never exceeds the number of 4x16-pixel groups,
    and the slab stride matches the C-ABI contract (N*taps*C + N)."""
    for arch, B in (("squeezedet", 20), ("squeezedetplus", 16), ("squeezedet", 1)):
        layers = layer_table(arch)
        h, w = ops.stem_out_size(384, 1248, layers[0][3])
        shapes = []
        for l in layers[2:]:
            if l[0] == "pool":
                h, w = ops.pool_out_size(h, w)
                continue
            _, cin, s, e1, e3 = l
            shapes += [(e3, s, 9, h, w), (e1, s, 1, h, w), (s, cin, 1, h, w)]
        shapes.append((72, layers[-1][3] + layers[-1][4], 9, h, w))                    # ConvDet
        for N, C, taps, hh, ww in shapes:
            S, stride = ops.wgrad_split(N, C, taps, B, hh, ww)
            assert stride == N * taps * C + N and S >= 1
            ngroups = B * -(-hh // 4) * -(-ww // 16)
            if taps == 9:
                assert ops.wgrad_uses_wino(N, C, taps, B, hh, ww), (arch, N, C)
                tc = ops._wino_wgrad_tc(N, C)
                blocks = -(-C // 16) if N % 64 else (N // 64) * -(-C // (16 * tc))
                assert S <= ngroups and S * blocks <= 512, (arch, N, C, S, blocks)
                assert ops.wgrad_split(N, C, taps, B, hh, ww, wino=False)[1] == stride  # same slab layout either way
            else:
                assert not ops.wgrad_uses_wino(N, C, taps, B, hh, ww)
