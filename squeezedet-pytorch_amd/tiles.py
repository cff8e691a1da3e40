"""Tile / configuration choice for the HIP kernels: the compiled configuration tables, the measured per-shape table
(``tuning.json``, written on an MI355X by tools/tune_conv.py and tools/tune_insitu.py), the heuristics behind unmeasured
shapes, the feasibility predicates of every kernel family (LDS budgets, channel granularity) and the split-K factors of the
weight-gradient kernels.  Pure host logic: nothing here launches a kernel."""
from __future__ import annotations

import math  # noqa: F401

from . import _native as nat


_CFG_TABLE = None


def cfg_table():
    """{cfg_id: (taps, kc, tile_px, bn)} from the compiled library."""
    global _CFG_TABLE
    if _CFG_TABLE is None:
        rows = nat.conv_cfgs()
        _CFG_TABLE = {i: (t, k, px, bn) for i, t, k, px, bn, _ in rows}
        _CFG_DMA.update({i: int(d) for i, _, _, _, _, d in rows})
    return _CFG_TABLE


_CFG_DMA = {}


def cfg_is_dma(cfg_id):
    cfg_table()
    return _CFG_DMA[cfg_id % 1000] != 0


def cfg_kernel_name(cfg_id):
    """Canonical kernel name of a configuration: conv_igemm<TAPS,KC,MT,NT> or conv_dma<TAPS,KC,MT,NT,WAVES>."""
    cfg_id %= 1000                            # + 1000 * k = workgroups-per-CU cap (see sqd_conv_fwd)
    taps, kc, px, bn = cfg_table()[cfg_id]
    d = _CFG_DMA[cfg_id]
    if d >= 3:                                # weight-stationary, barrier-free 1x1: conv_ws<NT,WAVES>
        return f'conv_ws<{bn // 16},{8 if d == 4 else 4}>'
    waves = 8 if d == 2 else 4
    mt = px // (16 * waves)
    return f'conv_dma<{taps},{kc},{mt},{bn // 16},{waves}>' if d else f'conv_igemm<{taps},{kc},{mt},{bn // 16}>'


def conv_cfg_ok(cfg_id, C):
    """Whether tile configuration ``cfg_id`` can run a layer with ``C`` input channels: the weight-stationary 1x1 family
    (conv_ws) keeps the slice's whole weight matrix in LDS next to at least a 3-stage activation ring per wave."""
    c = cfg_id % 1000
    cfg_table()
    d = _CFG_DMA[c]
    if d < 3:
        return True
    bn = cfg_table()[c][3]
    wv = 8 if d == 4 else 4
    nthr = wv * 64
    wslots = -(-(-(-C // 32) * 8 * bn) // nthr) * nthr
    return wslots * 16 + 3 * wv * 2048 <= 160 * 1024


_TUNING = None


def _tuning():
    """Measured per-shape table written by tools/tune_conv.py on an MI355X ({} if absent)."""
    global _TUNING
    if _TUNING is None:
        import json
        import os
        path = os.environ.get('SQD_TUNING_JSON') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuning.json')     # (override: A/B of tables)
        try:
            with open(path) as f:
                # F: (fused expand) and W: (Winograd) entries carry the time of the alternative they were measured against;
                # -1 = the alternative was faster
                def _pick(k, v):
                    if k.startswith('F:') and v.get('separate_us', 0) and v['us'] >= v['separate_us']:
                        return -1
                    if k.startswith('W:') and v.get('direct_us', 0) and v['us'] >= v['direct_us']:
                        return -1
                    return int(v['cfg'])
                _TUNING = {k: _pick(k, v) for k, v in json.load(f).items()}
        except (OSError, ValueError, KeyError):
            _TUNING = {}
    return _TUNING


def _nearest_tuned(prefix, npix):
    """Measured configuration of the same (taps, C, N) layer at the pixel count closest (in ratio) to ``npix``, if the
    table has one within a factor of 4: other batch sizes / resolutions then run the LDS-DMA tilings chosen on hardware
    instead of the generic heuristic below."""
    import math
    best = None
    for k, v in _tuning().items():
        if k.startswith(prefix) and v is not None and v >= 0:
            d = abs(math.log(max(int(k[len(prefix):]), 1) / max(npix, 1)))
            if d <= math.log(4.0) and (best is None or d < best[0]):
                best = (d, v)
    return None if best is None else best[1] % 1000             # drop the workgroup cap: it was measured for that grid size


def choose_cfg(taps, C, N, npix, staged=False):
    """Tile configuration for a conv layer: the measured table if it has this shape, else a heuristic
    (least channel padding, 128-pixel tiles when that still yields >= 4 workgroups per CU).  ``staged=True`` asks for a
    register-staged tiling (the only family that supports the input-side ``xmask``)."""
    tab = cfg_table()
    hit = None if staged else _tuning().get(f'{taps}:{C}:{N}:{npix}')
    if hit is None and not staged:
        hit = _nearest_tuned(f'{taps}:{C}:{N}:', npix)          # same layer at another batch size / resolution
    if hit is not None and hit % 1000 in tab and tab[hit % 1000][0] == taps:
        return hit
    want_kc = 16 if (taps == 9 or C <= 128 and C % 32 != 0 or C < 64) else 32
    best = None
    for cid, (t, kc, px, bn) in tab.items():
        if t != taps or _CFG_DMA.get(cid, 0) != (0 if staged else 1):     # default: LDS-DMA 4-wave tilings (fastest family measured)
            continue
        slices = -(-N // bn)
        pad = slices * bn / N
        tiles = -(-npix // px) * slices
        cost = pad                                   # wasted MFMA work
        cost += 0.15 * (kc != want_kc)
        cost += 0.02 * slices                        # each slice re-reads the activation tile
        if tiles < 1024 and px > 64:
            cost += 0.25                             # too few workgroups for 256 CUs
        if px == 64 and tiles >= 4096:
            cost += 0.05
        if best is None or cost < best[0]:
            best = (cost, cid)
    if best is None:
        raise RuntimeError(f'no conv configuration for taps={taps}')
    return best[1]


def fused_expand_cfgs(E):
    """3x3 LDS-DMA configurations usable by the fused Fire expand for half-width E: even number of 16-channel groups
    per slice, slice width dividing 2E."""
    tab = cfg_table()
    return [c for c, (t, kc, px, bn) in tab.items() if t == 9 and _CFG_DMA.get(c, 0) and (bn // 16) % 2 == 0 and (2 * E) % bn == 0]


def choose_fused_cfg(C, E, npix):
    """Configuration for ``fire_expand``: measured table key ``F:C:E:npix`` if present, else 64-channel slices."""
    ok = fused_expand_cfgs(E)
    if not ok:
        return None
    hit = _tuning().get(f'F:{C}:{E}:{npix}')
    if hit is not None:
        return hit if (hit >= 0 and hit % 1000 in ok) else None      # -1: measured slower than the two separate launches
    if any(k.startswith(f'F:{C}:{E}:') for k in _tuning()):
        near = _nearest_tuned(f'F:{C}:{E}:', npix)                   # only entries where fusing won are >= 0
        return near if (near is not None and near in ok) else None
    tab = cfg_table()
    pref = [c for c in ok if tab[c][3] == 64 and _CFG_DMA[c] == 1 and tab[c][2] == 64]
    return (pref or ok)[0]


WINO_SK_CFG = 16            # the balanced (stream-K) Winograd kernel, csrc/conv_wino_sk.hip (32-channel packing, four waves)
WINO_VS_CFG = 17            # the V-shared kernel for N <= 80 (ConvDet), csrc/conv_wino_vs.hip (80-channel packing, twelve waves)


def wino_cfgs():
    """{cfg_id: (slice width, waves per workgroup)} of the Winograd F(2x2,3x3) kernel family."""
    import ctypes
    out = {WINO_SK_CFG: (32, 4), WINO_VS_CFG: (80, 12)}
    for i in range(nat.lib().sqd_wino_num_cfgs()):
        if 4 <= i <= 7:             # retired ids (deep-prefetch, streamed U): the library answers "unsupported"
            continue
        bn, wv = ctypes.c_int(), ctypes.c_int()
        nat.check(nat.lib().sqd_wino_cfg_info(i, ctypes.byref(bn), ctypes.byref(wv)), 'sqd_wino_cfg_info')
        out[i] = (bn.value, wv.value)
    return out


def wino_kernel_name(cfg_id):
    """Name of a Winograd configuration as bench.py / the profiles print it: conv_wino<NT,WAVES> (ids 0..3),
    conv_wino_us<..> (8..11: U-stationary, barrier-free; ids 4..7, the deep-prefetch streamed-U forms, are retired)."""
    c = cfg_id % 1000
    if c == WINO_SK_CFG:
        return 'conv_wino_sk'
    if c == WINO_VS_CFG:
        return 'conv_wino_vs'
    bn, wv = wino_cfgs()[c]
    return f'conv_wino{("", "_dp", "_us")[c // 4]}<{bn // 16},{wv}>'


def wino_cfg_ok(cfg_id, C, N=None):
    """Whether Winograd configuration ``cfg_id`` can run a layer with ``C`` input (and, when given, ``N`` output) channels: ids 8..11
    (U-stationary kernel) keep the slice's whole transformed weight set in LDS next to the patch ring; the V-shared kernel runs
    N <= 80 (with the plain bias / ReLU epilogue)."""
    c = cfg_id % 1000
    if c == WINO_VS_CFG:
        return N is None or N <= 80
    if c < 8 or c == WINO_SK_CFG:
        return True
    bn, wv = wino_cfgs()[c]
    return (2 * wv * 256 * 4 + (C // 8) * 32 * bn * 4) * 4 <= 160 * 1024


def choose_wino_cfg(C, N, npix):
    """Winograd configuration for a 3x3 layer if the measured table (key ``W:C:N:npix``) says it beats the direct kernel,
    else None (unmeasured shapes run the direct kernel)."""
    if C % 8:
        return None
    hit = _tuning().get(f'W:{C}:{N}:{npix}')
    if hit is None:
        hit = _nearest_tuned(f'W:{C}:{N}:', npix) if any(k.startswith(f'W:{C}:{N}:') for k in _tuning()) else None
        if hit is not None and hit % 1000 == WINO_VS_CFG and not wino_vs_fills_chip(npix):
            hit = 2                 # a neighbouring batch size measured the V-shared kernel; here its one-workgroup-per-CU grid would not fill
    return hit if (hit is not None and hit >= 0) else None


def wino_vs_fills_chip(npix, cus=256):
    """The V-shared kernel (WINO_VS_CFG) runs ONE twelve-unit workgroup per CU: its grid of ceil(5 * groups / 12) workgroups (groups of
    64 pixels, estimated from the pixel count) should fill >= 85 % of its rounds -- 250 workgroups at bs=20, 24x78; at bs=8 (100
    workgroups) conv_wino<2,4> measured 128 against 174 us."""
    nwg = -(-5 * -(-npix // 64) // 12)
    return nwg / (cus * -(-nwg // cus)) >= 0.85


FIRE_WINO_CFGS = (8, 10, 12)  # 32-channel-slice ids of the U-stationary Winograd family; 12: the small-C form


def fire_wino_cfg_ok(cfg_id, C, E1=None, E3=None):
    """Whether ``fire_wino`` can run configuration ``cfg_id`` on a Fire with squeeze width C (and, for the small-C form 12 whose
    LDS plan holds every channel pass's U, expand widths E1 / E3)."""
    if cfg_id % 1000 == 12:
        if C % 8 or C > 16 or E1 is None or E3 is None or E3 > 64 or E1 > 128:      # (the kernel enumerates at most 4 + 2 channel passes)
            return False
        P3, P1 = -(-E3 // 32), -(-E1 // 128)
        lds = 4 * (2 * 8 * 256 * 4 + 2 * P3 * (C // 8) * 2048 + 2 * P1 * (C // 8) * (1024 if E1 <= 64 else 2048) + (2 * P3 + 2 * P1) * 64)
        return lds <= 160 * 1024
    return cfg_id % 1000 in FIRE_WINO_CFGS and wino_cfg_ok(cfg_id, C)


def choose_fire_wino_cfg(C, E1, E3, npix):
    """Configuration for the fused Winograd Fire expand (key ``X:C:E3:npix`` of the measured table) or None: only where the
    table says the one launch beats expand1x1 + Winograd expand3x3 inside the step."""
    if C % 8 or E1 % 16 or E3 % 4:
        return None
    hit = _tuning().get(f'X:{C}:{E3}:{npix}')
    return hit if (hit is not None and hit >= 0 and fire_wino_cfg_ok(hit, C, E1, E3)) else None


def fire_wino_kernel_name(cfg_id):
    return 'fire_wino16' if cfg_id % 1000 == 12 else wino_kernel_name(cfg_id).replace('conv_wino', 'fire_wino')


FIRE_BRIDGE_CFGS = (10, 12)


def fire_bridge_lds_bytes(cfg_id, C, N3, N1, Nsq):
    P3, P1 = -(-N3 // 32), -(-N1 // 128)
    if cfg_id % 1000 == 12:         # 16-wide passes, eight waves, U resident (expand1x1 stages halved when N1 <= 64)
        return 4 * (2 * 8 * 256 * 4 + 2 * P3 * (C // 8) * 2048 + 2 * P1 * (C // 8) * (1024 if N1 <= 64 else 2048)
                    + (2 * P3 + 8 * P1) * 4 * -(-Nsq // 16) * 64 + (2 * P3 + 2 * P1) * 64)
    P = P3 + P1
    ustages = P * (C // 8) if cfg_id % 1000 == 10 else 3
    return 4 * (2 * 4 * 256 * 4 + ustages * 1024 * 4 + (2 * P3 + 8 * P1) * 4 * -(-Nsq // 16) * 64 + P * 128)


def fire_bridge_cfg_ok(cfg_id, C, N3, N1, Nsq):
    """Whether ``fire_bridge`` can run this Fire pair: 8 | C, 16 | N1, 4 | N3, Nsq <= 32, and the LDS plan fits one CU."""
    return (cfg_id % 1000 in FIRE_BRIDGE_CFGS and C % 8 == 0 and N1 % 16 == 0 and N3 % 4 == 0 and Nsq % 4 == 0 and Nsq <= 32
            and (cfg_id % 1000 != 12 or (C <= 16 and N3 <= 64 and N1 <= 128)) and fire_bridge_lds_bytes(cfg_id, C, N3, N1, Nsq) <= 160 * 1024)


def choose_fire_bridge_cfg(C, N1, N3, Nsq, npix):
    """Tuned bridge configuration of a Fire pair (tuning.json row 'Y:C:N1:N3:Nsq:npix'), or None."""
    hit = _tuning().get(f'Y:{C}:{N1}:{N3}:{Nsq}:{npix}')
    return hit if (hit is not None and hit >= 0 and fire_bridge_cfg_ok(hit, C, N3, N1, Nsq)) else None


def fire_pool_bridge_ok(C, N3, N1, Nsq):
    """Whether ``fire_pool_bridge`` can run a Fire (squeeze width C, expands N1 + N3) -> pool -> squeeze (Nsq) chain."""
    if C % 8 or C > 16 or N1 % 16 or N3 % 4 or N1 > 64 or N3 > 64 or Nsq % 4 or Nsq > 32:
        return False
    P3, P1 = -(-N3 // 32), -(-N1 // 128)
    lds = 4 * (2 * 8 * 224 * 4 + 2 * P3 * (C // 8) * 2048 + 2 * P1 * (C // 8) * 1024 + (2 * P3 + 4 * P1) * 4 * -(-Nsq // 16) * 64
               + (2 * P3 + 2 * P1) * 64 + -(-Nsq // 16) * 16)
    return lds <= 160 * 1024


def choose_fire_pool_bridge(C, N1, N3, Nsq, npix):
    """Segments per column strip for the Fire -> pool -> Fire bridge (tuning.json row 'Z:C:N1:N3:Nsq:npix', cfg = segments) or None."""
    hit = _tuning().get(f'Z:{C}:{N1}:{N3}:{Nsq}:{npix}')
    return hit if (hit is not None and hit >= 1 and fire_pool_bridge_ok(C, N3, N1, Nsq)) else None


POOL_SQUEEZE_CFG = 28        # 1x1 tiling with KC = 32, 16-channel slices: its packed weights are [C/4][ceil16(N)][4]


def pool_squeeze_ok(C, N):
    """Whether ``pool_squeeze`` can run a (C -> N) squeeze behind a pool: KC | C, N <= 96, weights + one 128-channel
    activation chunk fit the LDS with room for two workgroups per CU."""
    return C % 32 == 0 and N % 4 == 0 and N <= 96 and (C // 4) * (-(-N // 16) * 16) * 16 + 32 * 1024 <= 80 * 1024


def stem_out_size(h, w, ksize):
    pad = 1 if ksize == 3 else 3
    return (h + 2 * pad - ksize) // 2 + 1, (w + 2 * pad - ksize) // 2 + 1


def pool_out_size(h, w):
    return (h - 3 + 1) // 2 + 1, (w - 3 + 1) // 2 + 1


# ---------------------------------------------------------------------------------------------
# training-side ops
# ---------------------------------------------------------------------------------------------
_TARGET_WGS = 1536          # workgroups a 3x3 weight-gradient launch aims for (measured sweep 768 / 1536 / 3072)


_TARGET_WGS_SQBWD = 512     # fused squeeze backward (ops.squeeze_bwd): two ~60 KB workgroups per CU, one resident round
_TARGET_WGS_1X1 = 512       # 1x1: fewer, longer pixel streams (less slab traffic per MFMA; round 2, inside the training step: 512 6.26 ms, 1024 6.29, 384 6.40)


WINO_WGRAD = True          # 3x3 weight gradients: Winograd kernel where it applies (N % 64 == 0)


_TARGET_WGS_WINO = 512      # Winograd wgrad: one resident round (two 4-wave workgroups per CU), every workgroup the same work


def wgrad_uses_wino(N, C, taps, B, H, W, wino=None):
    """Whether conv_wgrad runs the Winograd F(2x2,3x3) kernel for this layer (3x3, N % 64 == 0; ``wino`` overrides the
    module default WINO_WGRAD)."""
    return bool(WINO_WGRAD if wino is None else wino) and taps == 9 and (N % 64 == 0 or N <= 80) and N % 4 == 0 and C % 4 == 0


def _wino_wgrad_tc(N, C):
    """Input-channel blocks of 16 per workgroup of the Winograd wgrad kernel: 2 (32 channels) unless that would leave the
    last block half empty (C = 16, 48, ...: measured 50 vs 62 us on C48 -> N192) or the 5-block ConvDet variant runs (its 32-channel form
    was measured and removed: 228 -> 218 us for the kernel, but twice the splits and slab bytes cost the reduction more than that)."""
    if N % 64:
        return 1                # the 5-block (N <= 80: ConvDet) variant: 16 input channels per workgroup
    return 1 if (C % 32 == 16 or C < 32) else 2


WINO_WGRAD_GROUP = True    # the expand3x3 weight gradients of one backward stage share a launch (ops.conv_wgrad_wino_group;
                           # per model: SqueezeDetBase.group_wgrad)
WINO_WGRAD_GROUP_MAX = 6   # csrc/wino_wgrad.hip WW_MAX_GROUP


def wino_wgrad_blocks(N, C):
    """(out-channel, in-channel) blocks of dU a Winograd weight-gradient launch cuts a layer into (64 x 16|32; N <= 80: all of N x 16)."""
    tc = _wino_wgrad_tc(N, C)
    return -(-C // (16 * tc)) if N % 64 else (N // 64) * -(-C // (16 * tc))


def wino_wgrad_groups(layers, wino=None, enabled=None):
    """Which 3x3 weight gradients share a launch.  ``layers``: [(key, N, C, B, H, W)] in launch (backward) order.  Returns
    {key: (group id, S, tc, member keys)} for the layers that run grouped: Winograd form, N % 64 == 0, same pixel grid, same tile form, at
    least two of them (at most WINO_WGRAD_GROUP_MAX per launch).  S = the splits EVERY member is cut into: one resident round of
    workgroups over the whole group (or the measured row 'GW:tc:blocks:npix' of the table)."""
    if not (WINO_WGRAD_GROUP if enabled is None else enabled):
        return {}
    buckets = {}
    for key, N, C, B, H, W in layers:
        if wgrad_uses_wino(N, C, 9, B, H, W, wino) and N % 64 == 0:
            buckets.setdefault((B, H, W, _wino_wgrad_tc(N, C)), []).append((key, N, C))
    out, gid = {}, 0
    for (B, H, W, tc), members in buckets.items():
        for lo in range(0, len(members), WINO_WGRAD_GROUP_MAX):
            part = members[lo:lo + WINO_WGRAD_GROUP_MAX]
            if len(part) < 2:
                continue
            blocks = sum(wino_wgrad_blocks(N, C) for _k, N, C in part)
            ngroups = B * -(-H // 4) * -(-W // 16)
            tw = _TARGET_WGS_WINO
            S = max(1, min(ngroups, tw // blocks if blocks <= tw else 1))
            tuned = _tuning().get(f'GW:{tc}:{blocks}:{B * H * W}')
            if tuned is not None and tuned >= 1:
                S = max(1, min(ngroups, int(tuned)))
            keys = tuple(k for k, _n, _c in part)
            for k in keys:
                out[k] = (gid, S, tc, keys)
            gid += 1
    return out


def _wgrad1x1_tc(C):
    """In-channel tiles of 16 per workgroup of the direct 1x1 weight-gradient kernel for N >= 64 layers (csrc/wgrad.hip sqd_conv_wgrad)."""
    return 8 if C >= 256 else (4 if C >= 64 else -(-C // 16))


def wgrad1x1_groups(layers, enabled=None):
    """Which 1x1 weight gradients that do NOT run in the fused squeeze backward share a launch (``ops.conv_wgrad_group``).  ``layers``:
    [(key, N, C, B, H, W)] in launch (backward) order.  Returns {key: (group id, S, tc, member keys)}: same pixel grid, 64-out-channel
    tile form (N >= 64, not 64 < N <= 96), same in-channel tile, at least two members (at most WINO_WGRAD_GROUP_MAX per launch).
    S = the splits every member is cut into: one resident round of workgroups over the whole group ('G1:tc:groups:npix' overrides)."""
    if not (WINO_WGRAD_GROUP if enabled is None else enabled):
        return {}
    buckets = {}
    for key, N, C, B, H, W in layers:
        if N >= 64 and not (64 < N <= 96) and N % 4 == 0 and C % 4 == 0:
            buckets.setdefault((B, H, W, _wgrad1x1_tc(C)), []).append((key, N, C))
    out, gid = {}, 1000
    for (B, H, W, tc), members in buckets.items():
        for lo in range(0, len(members), WINO_WGRAD_GROUP_MAX):
            part = members[lo:lo + WINO_WGRAD_GROUP_MAX]
            if len(part) < 2:
                continue
            groups = sum(-(-N // 64) * -(-C // (16 * tc)) for _k, N, C in part)
            pb = 64 if tc == 1 else 32                     # pixels per block of the tile form (TH * 16)
            nblocks = -(-(B * H * W) // pb)
            S = max(1, min(nblocks, _TARGET_WGS_1X1 // groups if groups <= _TARGET_WGS_1X1 else 1))
            tuned = _tuning().get(f'G1:{tc}:{groups}:{B * H * W}')
            if tuned is not None and tuned >= 1:
                S = max(1, min(nblocks, int(tuned)))
            keys = tuple(k for k, _n, _c in part)
            for k in keys:
                out[k] = (gid, S, tc, keys)
            gid += 1
    return out


def wgrad_split(N, C, taps, B, H, W, wino=None, fused_dgrad=False, group_S=None):
    """(S, slab stride): number of split-K partial slabs the weight-gradient kernel writes for this layer, floats per slab.
    ``group_S``: the layer runs inside a grouped launch (``wino_wgrad_groups``) with that many splits.
    ``fused_dgrad``: the layer runs ``ops.squeeze_bwd`` (weight + data gradient in one launch: all N in one group, 64-channel in-tiles).
    S comes from the workgroup targets below unless the measured table has a row 'G:taps:N:C:npix' (tools/tune_insitu.py --mode
    train: the split of each layer tried inside the training step)."""
    tuned = _tuning().get(f'G:{taps}:{N}:{C}:{B * H * W}')
    if group_S is not None:
        if fused_dgrad or not (taps == 1 or wgrad_uses_wino(N, C, taps, B, H, W, wino)):
            raise ValueError('grouped weight gradient: Winograd 3x3 layers or plain 1x1 layers only')
        return int(group_S), N * taps * C + N
    if wgrad_uses_wino(N, C, taps, B, H, W, wino):
        ngroups = B * -(-H // 4) * -(-W // 16)                    # 4x16-pixel groups = the K axis of the 16 position GEMMs
        # (out-channel, in-channel) blocks of dU per workgroup: 64 x 16|32, or all of N <= 80 x 16 (ConvDet)
        blocks = -(-C // (16 * _wino_wgrad_tc(N, C))) if N % 64 else (N // 64) * -(-C // (16 * _wino_wgrad_tc(N, C)))
        tw = _TARGET_WGS_WINO
        S = max(1, min(ngroups, tw // blocks if blocks <= tw else 1))
        if tuned is not None and tuned >= 1:
            S = max(1, min(ngroups, int(tuned)))
        return S, N * taps * C + N
    if fused_dgrad:
        if taps != 1 or N > 128:
            raise ValueError('fused 1x1 backward: 1x1 layers with N <= 128')
        nblocks = -(-(B * H * W) // 32)
        groups = -(-C // 64)
        target = _TARGET_WGS_SQBWD
        S = max(1, min(nblocks, target // groups, 1024))
        if tuned is not None and tuned >= 1:
            S = max(1, min(nblocks, int(tuned), 1024))
        return S, N * C + N
    tn = 4 if N >= 64 else -(-N // 16)
    if taps == 9:
        if 64 < N <= 80:
            tn, tc = 5, 1
        elif tn == 4:
            tc = 2 if C % 32 == 0 else 1
        elif tn in (1, 2):
            tc = 2
        else:
            tc = 1
        nblocks = B * -(-H // 4) * -(-W // 16)
    else:
        tc = 4 if C >= 64 else -(-C // 16)
        if 64 < N <= 96:                        # (mirrors csrc/wgrad.hip sqd_conv_wgrad: 128-channel in-tiles, N = 96 as one 6-tile group)
            tn = 6
        if C >= 256:
            tc = 8
        nblocks = -(-(B * H * W) // 128)
    groups = -(-N // (tn * 16)) * -(-C // (tc * 16))
    S = max(1, min(nblocks, (_TARGET_WGS if taps == 9 else _TARGET_WGS_1X1) // groups, 256))
    if tuned is not None and tuned >= 1:
        S = max(1, min(nblocks, int(tuned), 256))
    return S, N * taps * C + N
