"""Thin, shape-checked Python wrappers over the C-ABI kernels (NHWC fp32 device tensors in,
NHWC fp32 device tensors out).  Every wrapper validates operand shapes against what the kernel
and its grid assume *before* launching -- an out-of-bounds access on the GPU can take the node down.

All launches go to the caller's current HIP stream; nothing synchronises.
"""
from __future__ import annotations

import math

import torch

from . import _native as nat

_CFG_TABLE = None


class KernelTimer:
    """Optional per-launch HIP-event bracket (bench.py / profiling only).  ``select`` limits the
    bracketing to kernels whose name is in the set (None = all).  Events are recorded on the stream
    the kernels are launched on (torch's current stream)."""

    def __init__(self, select=None):
        self.select = select
        self.records = []          # (name, tag, flops, bytes, start_event, end_event)

    def wants(self, name):
        return self.select is None or name in self.select

    def summary(self, nsteps=1):
        """{name: dict(launches, ms, flops, bytes, tags)} per step -- call after a synchronize.  ``nsteps`` = number
        of identical steps that were bracketed: per (kernel, shape) the MEDIAN launch time over all its samples is
        used, so a bracket that absorbed a host stall (GPU idle between the start marker and the launch) cannot
        distort the totals."""
        groups = {}
        for name, tag, fl, by, e0, e1 in self.records:
            g = groups.setdefault((name, tag), dict(ms=[], flops=fl, bytes=by))
            g['ms'].append(e0.elapsed_time(e1))
        out = {}
        for (name, tag), g in groups.items():
            ms = sorted(g['ms'])
            med = ms[len(ms) // 2]
            per_step = len(ms) / float(nsteps)             # launches of this shape per step
            d = out.setdefault(name, dict(launches=0.0, ms=0.0, flops=0.0, bytes=0.0, tags={}))
            d['launches'] += per_step; d['ms'] += med * per_step
            d['flops'] += g['flops'] * per_step; d['bytes'] += g['bytes'] * per_step
            d['tags'][tag] = [per_step, med * per_step, g['flops'] * per_step, g['bytes'] * per_step]
        return out


_timer = None


def set_timer(t):
    global _timer
    _timer = t


class _Bracket:
    __slots__ = ('rec',)

    def __init__(self, name, tag, flops, nbytes):
        t = _timer
        self.rec = None
        if t is not None and t.wants(name):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            self.rec = (name, tag, flops, nbytes, e0, e1)
            # a start marker that directly follows a kernel is time-stamped while that kernel still runs
            # (measured: +40..80 us vs rocprofv3); a preceding fence marker makes it wait for the stream
            torch.cuda.Event(enable_timing=True).record()
            e0.record()

    def done(self):
        if self.rec is not None:
            self.rec[5].record()
            _timer.records.append(self.rec)


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


class ConvPlan:
    """Packed weights ([C/KC][TAPS][Npad][KC], zero padded) + bias for one conv in one direction.
    ``dgrad=True`` packs the data-gradient orientation of the same OIHW parameter (in/out channels
    swapped, taps flipped).  Packing is one HIP kernel launch on the current stream."""
    __slots__ = ('cfg_id', 'taps', 'kc', 'bn', 'C', 'N', 'Npad', 'w', 'bias')

    def __init__(self, w_oihw, bias, cfg_id, dgrad=False):
        taps_cfg, kc, _px, bn = cfg_table()[cfg_id % 1000]
        No, Ci, kh, kw = w_oihw.shape
        taps = kh * kw
        if taps != taps_cfg or kh != kw or taps not in (1, 9):
            raise ValueError(f'weight {tuple(w_oihw.shape)} does not fit conv cfg {cfg_id} (taps={taps_cfg})')
        N, C = (Ci, No) if dgrad else (No, Ci)
        if C % 4 or N % 4:
            raise ValueError('channel counts must be multiples of 4')
        if not w_oihw.is_cuda or w_oihw.dtype != torch.float32:
            raise ValueError('weights must be fp32 CUDA tensors')
        self.cfg_id, self.taps, self.kc, self.bn, self.C, self.N = cfg_id, taps, kc, bn, C, N
        self.Npad = -(-N // bn) * bn
        nchunks = -(-C // kc)
        src = w_oihw.detach().contiguous()
        self.w = torch.empty(nchunks, taps, self.Npad, kc, device=src.device, dtype=torch.float32)
        rc = nat.lib().sqd_pack_conv_weight(nat.ptr(src), nat.ptr(self.w), No, Ci, taps, kc, self.Npad, int(dgrad),
                                            nat.stream_handle(src.device))
        nat.check(rc, 'sqd_pack_conv_weight')
        self.bias = None if (bias is None or dgrad) else bias.detach().contiguous()


_PACK_TABLES = {}


def repack_batched(plans_and_weights, is_dgrad):
    """Refresh many packed weight copies with ONE kernel launch.  plans_and_weights: [(ConvPlan, weight)],
    is_dgrad: parallel list of bools."""
    if not plans_and_weights:
        return
    rows = []
    for (plan, w), dg in zip(plans_and_weights, is_dgrad):
        if not w.is_contiguous():
            raise ValueError('repack_batched: parameters must be contiguous')
        rows.append([w.data_ptr(), plan.w.data_ptr(), w.shape[0], w.shape[1], plan.taps, plan.kc, plan.Npad, plan.w.shape[0],
                     int(dg), plan.w.numel()])
    dev = plans_and_weights[0][1].device
    # the descriptor table only holds pointers and shapes: after the first optimizer step it is the same every step, so the
    # device copy is cached (no host-to-device copy per step; also what makes the training step hipGraph-capturable)
    key = (str(dev), tuple(tuple(r) for r in rows))
    table = _PACK_TABLES.get(key)
    if table is None:
        if len(_PACK_TABLES) > 16:
            _PACK_TABLES.clear()
        table = torch.tensor(rows, dtype=torch.int64).to(dev)
        _PACK_TABLES[key] = table
    rc = nat.lib().sqd_pack_conv_weights_batched(nat.ptr(table), len(rows), 16, nat.stream_handle(dev))
    nat.check(rc, 'sqd_pack_conv_weights_batched')
    return table


def dgrad_weight(w_oihw):
    """Weights of the convolution that computes dX from dY: swap in/out channels, flip taps."""
    return w_oihw.permute(1, 0, 2, 3).flip(2, 3).contiguous()


def _check_nhwc(t, name):
    if t.dim() != 4 or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f'{name} must be a contiguous fp32 CUDA tensor [B,H,W,C], got {tuple(t.shape)} {t.dtype} {t.device}')


def conv(x, x_coff, plan, y, y_coff, relu=False, accumulate=False, xmask=None, xmask_coff=0, ymask=None, ymask_coff=0,
         ymul=None, ymul_coff=0):
    """y[..., y_coff:y_coff+N] (=|+=) conv(x[..., x_coff:x_coff+C] [* (xmask>0)]) (+bias) (ReLU)."""
    _check_nhwc(x, 'x'); _check_nhwc(y, 'y')
    B, H, W, xp = x.shape
    if tuple(y.shape[:3]) != (B, H, W):
        raise ValueError(f'conv: x {tuple(x.shape)} and y {tuple(y.shape)} disagree on B,H,W')
    yp = y.shape[3]
    if x_coff < 0 or x_coff + plan.C > xp or y_coff < 0 or y_coff + plan.N > yp:
        raise ValueError('conv: channel window out of range')
    mp = 0
    if xmask is not None:
        if cfg_is_dma(plan.cfg_id):
            raise ValueError('conv: xmask needs a register-staged configuration')
        _check_nhwc(xmask, 'xmask')
        if tuple(xmask.shape[:3]) != (B, H, W) or xmask_coff + plan.C > xmask.shape[3]:
            raise ValueError('conv: xmask geometry mismatch')
        mp = xmask.shape[3]
    ymp = ylp = 0
    if ymask is not None:
        _check_nhwc(ymask, 'ymask')
        if tuple(ymask.shape[:3]) != (B, H, W) or ymask_coff + plan.N > ymask.shape[3]:
            raise ValueError('conv: ymask geometry mismatch')
        ymp = ymask.shape[3]
    if ymul is not None:
        _check_nhwc(ymul, 'ymul')
        if tuple(ymul.shape[:3]) != (B, H, W) or ymul_coff + plan.N > ymul.shape[3]:
            raise ValueError('conv: ymul geometry mismatch')
        ylp = ymul.shape[3]
    if B * H * W * max(xp, yp) >= 2 ** 40:
        raise ValueError('conv: tensor too large')
    br = None
    if _timer is not None:
        npix = B * H * W
        br = _Bracket(cfg_kernel_name(plan.cfg_id),
                      f'{plan.taps}tap C{plan.C} N{plan.N} {H}x{W}', 2.0 * npix * plan.N * plan.C * plan.taps,
                      4.0 * (npix * (plan.C + plan.N) + plan.N * plan.C * plan.taps))
    rc = nat.lib().sqd_conv_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias), nat.ptr(y), nat.ptr(xmask),
                                nat.ptr(ymask), nat.ptr(ymul), B, H, W, plan.C, xp, x_coff, plan.N, plan.Npad, yp, y_coff,
                                int(relu), int(accumulate), mp, xmask_coff, ymp, ymask_coff, ylp, ymul_coff,
                                plan.cfg_id, nat.stream_handle(x.device))
    nat.check(rc, 'sqd_conv_fwd')
    if br is not None:
        br.done()
    return y


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


class FusedExpandPlan(object):
    """Packed weights of one Fire's expand pair for ``fire_expand``: the 2E output channels in alternating 16-channel
    groups (expand1x1 group as a centre-tap-only 3x3, then the expand3x3 group), packed like any 3x3 conv."""

    def __init__(self, w1, b1, w3, b3, cfg_id):
        E, C = w1.shape[0], w1.shape[1]
        if tuple(w1.shape) != (E, C, 1, 1) or tuple(w3.shape) != (E, C, 3, 3) or E % 16:
            raise ValueError(f'fused expand: need expand1x1 [E,C,1,1] and expand3x3 [E,C,3,3] with E % 16 == 0, got {tuple(w1.shape)}, {tuple(w3.shape)}')
        if cfg_id % 1000 not in fused_expand_cfgs(E):
            raise ValueError(f'conv cfg {cfg_id} cannot run the fused expand with E={E}')
        w1 = w1.detach(); w3 = w3.detach()
        wf = torch.zeros(E // 16, 2, 16, C, 3, 3, device=w3.device, dtype=torch.float32)
        wf[:, 0, :, :, 1, 1] = w1.reshape(E // 16, 16, C)
        wf[:, 1] = w3.reshape(E // 16, 16, C, 3, 3)
        bf = torch.stack([b1.detach().reshape(E // 16, 16), b3.detach().reshape(E // 16, 16)], 1).reshape(-1)
        self.plan = ConvPlan(wf.reshape(2 * E, C, 3, 3), bf, cfg_id)
        self.E, self.C, self.cfg_id = E, C, cfg_id


def fire_expand(x, x_coff, fplan, y, y_coff):
    """y[..., y_coff:y_coff+E] = relu(expand1x1(x)), y[..., y_coff+E:y_coff+2E] = relu(expand3x3(x)), one launch."""
    _check_nhwc(x, 'x'); _check_nhwc(y, 'y')
    B, H, W, xp = x.shape
    if tuple(y.shape[:3]) != (B, H, W):
        raise ValueError('fire_expand: x and y disagree on B,H,W')
    p = fplan.plan
    if x_coff + fplan.C > xp or y_coff + 2 * fplan.E > y.shape[3]:
        raise ValueError('fire_expand: channel window out of range')
    br = _Bracket(cfg_kernel_name(fplan.cfg_id).replace('conv_dma', 'fire_expand'), f'expand C{fplan.C} E{fplan.E} {H}x{W}',
                  2.0 * B * H * W * fplan.E * fplan.C * 10, 4.0 * B * H * W * (fplan.C + 2 * fplan.E)) if _timer is not None else None
    rc = nat.lib().sqd_fire_expand_fwd(nat.ptr(x), nat.ptr(p.w), nat.ptr(p.bias), nat.ptr(y), B, H, W, fplan.C, xp, x_coff, fplan.E,
                                       p.Npad, y.shape[3], y_coff, fplan.cfg_id, nat.stream_handle(x.device))
    nat.check(rc, 'sqd_fire_expand_fwd')
    if br is not None:
        br.done()
    return y


def wino_cfgs():
    """{cfg_id: (slice width, waves per workgroup)} of the Winograd F(2x2,3x3) kernel family."""
    import ctypes
    out = {}
    for i in range(nat.lib().sqd_wino_num_cfgs()):
        bn, wv = ctypes.c_int(), ctypes.c_int()
        nat.check(nat.lib().sqd_wino_cfg_info(i, ctypes.byref(bn), ctypes.byref(wv)), 'sqd_wino_cfg_info')
        out[i] = (bn.value, wv.value)
    return out


def wino_kernel_name(cfg_id):
    """Name of a Winograd configuration as bench.py / the profiles print it: conv_wino<NT,WAVES> (ids 0..3),
    conv_wino_dp<..> (4..7: deep-prefetch staging), conv_wino_us<..> (8..11: U-stationary, barrier-free)."""
    c = cfg_id % 1000
    bn, wv = wino_cfgs()[c]
    return f'conv_wino{("", "_dp", "_us")[c // 4]}<{bn // 16},{wv}>'


def wino_cfg_ok(cfg_id, C):
    """Whether Winograd configuration ``cfg_id`` can run a layer with ``C`` input channels: ids 8..11 (U-stationary kernel)
    keep the slice's whole transformed weight set in LDS next to the patch ring."""
    c = cfg_id % 1000
    if c < 8:
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
    return hit if (hit is not None and hit >= 0) else None


class WinoPlan:
    """Transformed weights U = G g G^T ([C/8][16][Npad][8]) + bias of one 3x3 conv for ``conv_wino``."""
    __slots__ = ('cfg_id', 'C', 'N', 'Npad', 'bn', 'w', 'bias')

    def __init__(self, w_oihw, bias, cfg_id, dgrad=False):
        No, Ci, kh, kw = w_oihw.shape
        if (kh, kw) != (3, 3):
            raise ValueError(f'Winograd plan needs a 3x3 weight, got {tuple(w_oihw.shape)}')
        if not w_oihw.is_cuda or w_oihw.dtype != torch.float32:
            raise ValueError('weights must be fp32 CUDA tensors')
        N, C = (Ci, No) if dgrad else (No, Ci)
        if C % 8 or N % 4:
            raise ValueError('Winograd conv: C must be a multiple of 8 and N of 4')
        bn = wino_cfgs()[cfg_id % 1000][0]
        self.cfg_id, self.C, self.N, self.bn = cfg_id, C, N, bn
        self.Npad = -(-N // bn) * bn
        src = w_oihw.detach().contiguous()
        self.w = torch.empty(C // 8, 16, self.Npad, 8, device=src.device, dtype=torch.float32)
        nat.check(nat.lib().sqd_pack_wino_weight(nat.ptr(src), nat.ptr(self.w), No, Ci, self.Npad, int(dgrad),
                                                 nat.stream_handle(src.device)), 'sqd_pack_wino_weight')
        self.bias = None if (bias is None or dgrad) else bias.detach().contiguous()

    def repack(self, w_oihw, bias, dgrad=False):
        """Re-transform into the same buffer after the parameter changed (pointer-stable: hipGraph replays stay valid)."""
        src = w_oihw.detach()
        if not src.is_contiguous():
            raise ValueError('WinoPlan.repack: parameters must be contiguous')
        nat.check(nat.lib().sqd_pack_wino_weight(nat.ptr(src), nat.ptr(self.w), src.shape[0], src.shape[1], self.Npad, int(dgrad),
                                                 nat.stream_handle(src.device)), 'sqd_pack_wino_weight')
        if self.bias is not None:
            self.bias = bias.detach()


def repack_wino_batched(plans_and_weights, is_dgrad):
    """Re-transform many WinoPlans with ONE kernel launch (pointer-stable).  plans_and_weights: [(WinoPlan, weight)]."""
    if not plans_and_weights:
        return None
    rows = []
    for (plan, w), dg in zip(plans_and_weights, is_dgrad):
        if not w.is_contiguous():
            raise ValueError('repack_wino_batched: parameters must be contiguous')
        rows.append([w.data_ptr(), plan.w.data_ptr(), w.shape[0], w.shape[1], plan.Npad, int(dg), (plan.C // 8) * plan.Npad * 8])
    dev = plans_and_weights[0][1].device
    key = ('wino', str(dev), tuple(tuple(r) for r in rows))
    table = _PACK_TABLES.get(key)
    if table is None:
        if len(_PACK_TABLES) > 16:
            _PACK_TABLES.clear()
        table = torch.tensor(rows, dtype=torch.int64).to(dev)
        _PACK_TABLES[key] = table
    nat.check(nat.lib().sqd_pack_wino_weights_batched(nat.ptr(table), len(rows), 16, nat.stream_handle(dev)), 'sqd_pack_wino_weights_batched')
    return table


def conv_wino(x, x_coff, plan, y, y_coff, relu=False, accumulate=False, ymask=None, ymul=None):
    """y[..., y_coff:y_coff+N] (=|+=) conv3x3(x[..., x_coff:x_coff+C]) (+bias) (* ymul) (zero where ymask <= 0) (ReLU),
    Winograd F(2x2,3x3) kernel.  ``ymask`` / ``ymul`` are read through y's own channel window (same shape as y)."""
    _check_nhwc(x, 'x'); _check_nhwc(y, 'y')
    B, H, W, xp = x.shape
    if tuple(y.shape[:3]) != (B, H, W):
        raise ValueError(f'conv_wino: x {tuple(x.shape)} and y {tuple(y.shape)} disagree on B,H,W')
    yp = y.shape[3]
    if x_coff < 0 or x_coff + plan.C > xp or y_coff < 0 or y_coff + plan.N > yp:
        raise ValueError('conv_wino: channel window out of range')
    if B * H * W * max(xp, yp) >= 2 ** 40:
        raise ValueError('conv_wino: tensor too large')
    br = None
    if _timer is not None:
        npix = B * H * W
        # flops = what the MFMA pipe executes (16 element-wise GEMMs per 2x2 tile = direct form / 2.25): the roofline
        # fraction of this kernel is against that; bench.py also quotes the direct-form equivalent
        br = _Bracket(wino_kernel_name(plan.cfg_id), f'9tap C{plan.C} N{plan.N} {H}x{W}', 2.0 * npix * plan.N * plan.C * 4,
                      4.0 * (npix * (plan.C + plan.N) + plan.N * plan.C * 16))
    for t, nm in ((ymask, 'ymask'), (ymul, 'ymul')):
        if t is not None:
            _check_nhwc(t, nm)
            if tuple(t.shape) != tuple(y.shape):
                raise ValueError(f'conv_wino: {nm} must have the shape of y')
    rc = nat.lib().sqd_conv_wino_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias), nat.ptr(y), nat.ptr(ymask), nat.ptr(ymul),
                                     B, H, W, plan.C, xp, x_coff, plan.N, plan.Npad, yp, y_coff, int(relu), int(accumulate),
                                     plan.cfg_id, nat.stream_handle(x.device))
    nat.check(rc, 'sqd_conv_wino_fwd')
    if br is not None:
        br.done()
    return y


FIRE_WINO_CFGS = (4, 6, 8, 10, 12)  # 32-channel-slice ids of the deep-prefetch / U-stationary Winograd family; 12: the small-C form


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


class FireWinoPlan:
    """Transformed weights of a Fire's expand pair for ``fire_wino``: expand3x3's U followed by expand1x1's four inner
    positions as virtual channels (csrc/conv_wino.hip, sqd_pack_wino_fire)."""
    __slots__ = ('cfg_id', 'C', 'N3', 'N1', 'Npad', 'w', 'b3', 'b1')

    def __init__(self, w1, b1, w3, b3, cfg_id):
        N3, C = w3.shape[0], w3.shape[1]
        N1 = w1.shape[0]
        if tuple(w3.shape) != (N3, C, 3, 3) or tuple(w1.shape) != (N1, C, 1, 1) or C % 8 or N1 % 16 or N3 % 4:
            raise ValueError(f'fire_wino: need expand3x3 [N3,C,3,3] and expand1x1 [N1,C,1,1], C % 8 == 0, got {tuple(w3.shape)}, {tuple(w1.shape)}')
        if not fire_wino_cfg_ok(cfg_id, C, N1, N3):
            raise ValueError(f'fire_wino: configuration {cfg_id} cannot run C={C} E={N1}+{N3}')
        self.cfg_id, self.C, self.N3, self.N1 = cfg_id, C, N3, N1
        self.Npad = -(-N3 // 32) * 32 + -(-N1 // 128) * 32
        self.w = torch.empty(C // 8, 16, self.Npad, 8, device=w3.device, dtype=torch.float32)
        nat.check(nat.lib().sqd_pack_wino_fire(nat.ptr(w3.detach().contiguous()), nat.ptr(w1.detach().contiguous()), nat.ptr(self.w),
                                               N3, N1, C, self.Npad, nat.stream_handle(w3.device)), 'sqd_pack_wino_fire')
        self.b3 = None if b3 is None else b3.detach().contiguous()
        self.b1 = None if b1 is None else b1.detach().contiguous()


def fire_wino(x, x_coff, plan, y, y_coff1, y_coff3):
    """y[..., y_coff1:+N1] = relu(expand1x1(x)), y[..., y_coff3:+N3] = relu(expand3x3(x)) in ONE Winograd launch."""
    _check_nhwc(x, 'x'); _check_nhwc(y, 'y')
    B, H, W, xp = x.shape
    if tuple(y.shape[:3]) != (B, H, W):
        raise ValueError('fire_wino: x and y disagree on B,H,W')
    yp = y.shape[3]
    if x_coff < 0 or x_coff + plan.C > xp or min(y_coff1, y_coff3) < 0 or y_coff1 + plan.N1 > yp or y_coff3 + plan.N3 > yp:
        raise ValueError('fire_wino: channel window out of range')
    if not (y_coff1 + plan.N1 <= y_coff3 or y_coff3 + plan.N3 <= y_coff1):
        raise ValueError('fire_wino: output windows overlap')
    br = None
    if _timer is not None:
        npix = B * H * W
        br = _Bracket(fire_wino_kernel_name(plan.cfg_id), f'fire C{plan.C} E{plan.N1}+{plan.N3} {H}x{W}',
                      2.0 * npix * plan.C * (4 * plan.N3 + plan.N1), 4.0 * (npix * (plan.C + plan.N1 + plan.N3) + plan.C * (16 * plan.N3 + plan.N1)))
    rc = nat.lib().sqd_fire_wino_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.b3), nat.ptr(plan.b1), nat.ptr(y), B, H, W, plan.C, xp, x_coff,
                                     plan.N3, y_coff3, plan.N1, y_coff1, plan.Npad, yp, plan.cfg_id, nat.stream_handle(x.device))
    nat.check(rc, 'sqd_fire_wino_fwd')
    if br is not None:
        br.done()
    return y


FIRE_BRIDGE_CFGS = (6, 10, 12)


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


class FireBridgePlan:
    """Operands of ``fire_bridge``: the Fire's expand pair transformed as in FireWinoPlan, the per-pass bias table, and the next
    Fire's squeeze weights laid out as MFMA A operands (include/sqd_hip.h, sqd_fire_bridge_fwd)."""
    __slots__ = ('cfg_id', 'C', 'N3', 'N1', 'Npad', 'Nsq', 'w', 'bias_tab', 'sq_ops', 'sq_bias', 'pooled')

    def __init__(self, w1, b1, w3, b3, wsq, bsq, cfg_id, pooled=False):
        N3, C = w3.shape[0], w3.shape[1]
        N1, Nsq = w1.shape[0], wsq.shape[0]
        if tuple(w3.shape) != (N3, C, 3, 3) or tuple(w1.shape) != (N1, C, 1, 1) or tuple(wsq.shape) != (Nsq, N1 + N3, 1, 1):
            raise ValueError(f'fire_bridge: need expand3x3 [N3,C,3,3], expand1x1 [N1,C,1,1] and the next squeeze [Nsq,N1+N3,1,1], got '
                             f'{tuple(w3.shape)}, {tuple(w1.shape)}, {tuple(wsq.shape)}')
        if pooled:
            if not fire_pool_bridge_ok(C, N3, N1, Nsq):
                raise ValueError(f'fire_pool_bridge: cannot run C={C} E={N1}+{N3} -> {Nsq}')
            cfg_id = 12                              # (the operand layout of the 16-wide-pass form)
        elif not fire_bridge_cfg_ok(cfg_id, C, N3, N1, Nsq):
            raise ValueError(f'fire_bridge: configuration {cfg_id} cannot run C={C} E={N1}+{N3} -> {Nsq}')
        self.pooled = pooled
        dev = w3.device
        self.cfg_id, self.C, self.N3, self.N1, self.Nsq = cfg_id, C, N3, N1, Nsq
        P3, P1 = -(-N3 // 32), -(-N1 // 128)
        self.Npad = 32 * (P3 + P1)
        self.w = torch.empty(C // 8, 16, self.Npad, 8, device=dev, dtype=torch.float32)
        nat.check(nat.lib().sqd_pack_wino_fire(nat.ptr(w3.detach().contiguous()), nat.ptr(w1.detach().contiguous()), nat.ptr(self.w),
                                               N3, N1, C, self.Npad, nat.stream_handle(dev)), 'sqd_pack_wino_fire')
        # cat channel of every 16-channel block, in pass order: expand3x3 slices (cat offset N1), then expand1x1 slices
        narrow = cfg_id % 1000 == 12                 # 16-wide passes: 1 block per expand3x3 pass, 4 per expand1x1 pass
        if narrow:
            rb = 2 if pooled else 4               # (the pooled form keeps only the blocks that exist when N1 <= 64)
            base = [N1 + 16 * p for p in range(2 * P3)] + [128 * (s1 >> 1) + (2 * r + (s1 & 1)) * 16 for s1 in range(2 * P1) for r in range(rb)]
        else:
            base = [N1 + 32 * s + 16 * j for s in range(P3) for j in range(2)] + [128 * s + 16 * blk for s in range(P1) for blk in range(8)]
        limit = [N1 + N3] * (2 * P3) + [N1] * (len(base) - 2 * P3)
        nblk, nq = len(base), -(-Nsq // 16)
        ch = torch.tensor(base, device=dev).view(nblk, 1) + torch.arange(16, device=dev).view(1, 16)          # [blk][c16]
        ok = ch < torch.tensor(limit, device=dev).view(nblk, 1)
        chs = torch.where(ok, ch, torch.zeros_like(ch))
        wz = torch.zeros(16 * nq, N1 + N3, device=dev, dtype=torch.float32)
        wz[:Nsq] = wsq.detach().reshape(Nsq, N1 + N3)
        g = wz[:, chs.reshape(-1)].view(nq, 16, nblk, 4, 4) * ok.view(1, 1, nblk, 4, 4)                        # [q][lr][blk][g][t]
        self.sq_ops = g.permute(2, 4, 0, 3, 1).contiguous()                                                   # [blk][t][q][g][lr]
        bcat = torch.cat([torch.zeros(N1, device=dev) if b1 is None else b1.detach().float(),
                          torch.zeros(N3, device=dev) if b3 is None else b3.detach().float()])
        bvals = bcat[chs.reshape(-1)].view(nblk, 16) * ok
        if narrow:
            bt = torch.zeros(2 * P3 + 2 * P1, 4, 16, device=dev, dtype=torch.float32)
            bt[:2 * P3, 0] = bvals[:2 * P3]
            rb = 2 if pooled else 4
            bt[2 * P3:, :rb] = bvals[2 * P3:].view(2 * P1, rb, 16)
        else:
            bt = torch.zeros(P3 + P1, 8, 16, device=dev, dtype=torch.float32)
            bt[:P3, :2] = bvals[:2 * P3].view(P3, 2, 16)
            bt[P3:] = bvals[2 * P3:].view(P1, 8, 16)
        self.bias_tab = bt.contiguous()
        self.sq_bias = (torch.zeros(Nsq, device=dev) if bsq is None else bsq.detach().float()).contiguous()


def fire_bridge(x, x_coff, plan, y, y_coff):
    """y[..., y_coff:+Nsq] = relu(squeeze'(cat(relu(expand1x1(x)), relu(expand3x3(x))))) in ONE launch (inference)."""
    _check_nhwc(x, 'x'); _check_nhwc(y, 'y')
    B, H, W, xp = x.shape
    if tuple(y.shape[:3]) != (B, H, W) or plan.pooled:
        raise ValueError('fire_bridge: x and y disagree on B,H,W (or the plan is a pooled one)')
    yp = y.shape[3]
    if x_coff < 0 or x_coff + plan.C > xp or y_coff < 0 or y_coff + plan.Nsq > yp:
        raise ValueError('fire_bridge: channel window out of range')
    br = None
    if _timer is not None:
        npix = B * H * W
        br = _Bracket('fire_bridge', f'fire C{plan.C} E{plan.N1}+{plan.N3} -> S{plan.Nsq} {H}x{W}',
                      2.0 * npix * (plan.C * (4 * plan.N3 + plan.N1) + (plan.N1 + plan.N3) * plan.Nsq),
                      4.0 * (npix * (plan.C + plan.Nsq) + plan.C * (16 * plan.N3 + plan.N1) + (plan.N1 + plan.N3) * plan.Nsq))
    rc = nat.lib().sqd_fire_bridge_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias_tab), nat.ptr(plan.sq_ops), nat.ptr(plan.sq_bias),
                                       nat.ptr(y), B, H, W, plan.C, xp, x_coff, plan.N3, plan.N1, plan.Npad, plan.Nsq, yp, y_coff,
                                       plan.cfg_id, nat.stream_handle(x.device))
    nat.check(rc, 'sqd_fire_bridge_fwd')
    if br is not None:
        br.done()
    return y


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


def fire_pool_bridge(x, x_coff, plan, y, y_coff, nseg=4):
    """y[..., y_coff:+Nsq] = relu(squeeze'(maxpool3x3s2_ceil(cat(relu(expand1x1(x)), relu(expand3x3(x)))))) in ONE launch
    (inference); y is [B, Hp, Wp, .] with (Hp, Wp) = pool_out_size(H, W).  ``plan``: FireBridgePlan(..., pooled=True)."""
    _check_nhwc(x, 'x'); _check_nhwc(y, 'y')
    B, H, W, xp = x.shape
    Hp, Wp = pool_out_size(H, W)
    if tuple(y.shape[:3]) != (B, Hp, Wp) or not plan.pooled:
        raise ValueError('fire_pool_bridge: y must be [B, Hp, Wp, .] of the pooled map and the plan a pooled one')
    yp = y.shape[3]
    if x_coff < 0 or x_coff + plan.C > xp or y_coff < 0 or y_coff + plan.Nsq > yp:
        raise ValueError('fire_pool_bridge: channel window out of range')
    br = None
    if _timer is not None:
        npix = B * H * W
        br = _Bracket('fire_pool_bridge', f'fire C{plan.C} E{plan.N1}+{plan.N3} -> pool -> S{plan.Nsq} {H}x{W}',
                      2.0 * (npix * plan.C * (4 * plan.N3 + plan.N1) + B * Hp * Wp * (plan.N1 + plan.N3) * plan.Nsq),
                      4.0 * (npix * plan.C + B * Hp * Wp * plan.Nsq + plan.C * (16 * plan.N3 + plan.N1) + (plan.N1 + plan.N3) * plan.Nsq))
    rc = nat.lib().sqd_fire_pool_bridge_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias_tab), nat.ptr(plan.sq_ops), nat.ptr(plan.sq_bias),
                                            nat.ptr(y), B, H, W, plan.C, xp, x_coff, plan.N3, plan.N1, plan.Npad, plan.Nsq, Hp, Wp, yp, y_coff,
                                            int(nseg), nat.stream_handle(x.device))
    nat.check(rc, 'sqd_fire_pool_bridge_fwd')
    if br is not None:
        br.done()
    return y


POOL_SQUEEZE_CFG = 28        # 1x1 tiling with KC = 32, 16-channel slices: its packed weights are [C/4][ceil16(N)][4]


def pool_squeeze_ok(C, N):
    """Whether ``pool_squeeze`` can run a (C -> N) squeeze behind a pool: KC | C, N <= 96, weights + one 128-channel
    activation chunk fit the LDS with room for two workgroups per CU."""
    return C % 32 == 0 and N % 4 == 0 and N <= 96 and (C // 4) * (-(-N // 16) * 16) * 16 + 32 * 1024 <= 80 * 1024


def pool_squeeze(x, x_coff, C, plan, y, y_coff):
    """y[..., y_coff:y_coff+N] = relu(squeeze1x1(maxpool3x3s2_ceil(x[..., x_coff:x_coff+C]))) without materialising the
    pooled tensor (inference).  ``plan``: ConvPlan of the squeeze packed for POOL_SQUEEZE_CFG."""
    _check_nhwc(x, 'x'); _check_nhwc(y, 'y')
    B, H, W, xp = x.shape
    Ho, Wo = pool_out_size(H, W)
    if tuple(y.shape[:3]) != (B, Ho, Wo) or plan.taps != 1 or plan.C != C or plan.kc != 32 or plan.Npad != -(-plan.N // 16) * 16:
        raise ValueError('pool_squeeze: geometry / plan mismatch')
    if x_coff + C > xp or y_coff + plan.N > y.shape[3] or not pool_squeeze_ok(C, plan.N):
        raise ValueError('pool_squeeze: unsupported channel configuration')
    br = _Bracket('pool_squeeze', f'pool+squeeze C{C} N{plan.N} {H}x{W}', 2.0 * B * Ho * Wo * plan.N * C,
                  4.0 * B * (H * W * C + Ho * Wo * plan.N)) if _timer is not None else None
    rc = nat.lib().sqd_pool_squeeze_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias), nat.ptr(y), B, H, W, C, xp, x_coff, plan.N, plan.Npad,
                                        y.shape[3], y_coff, nat.stream_handle(x.device))
    nat.check(rc, 'sqd_pool_squeeze_fwd')
    if br is not None:
        br.done()
    return y


def stem_out_size(h, w, ksize):
    pad = 1 if ksize == 3 else 3
    return (h + 2 * pad - ksize) // 2 + 1, (w + 2 * pad - ksize) // 2 + 1


def stem_conv_relu(image, weight, bias, out=None, relu=True):
    """image NCHW [B,3,H,W]; weight OIHW [N,3,k,k] (the checkpoint tensor as is); -> NHWC [B,Ho,Wo,N]
    (``relu=False``: the bare convolution)."""
    if image.dim() != 4 or image.shape[1] != 3 or image.dtype != torch.float32 or not image.is_cuda:
        raise ValueError(f'stem: image must be fp32 CUDA NCHW with 3 channels, got {tuple(image.shape)}')
    image = image.contiguous()
    N, ci, k, k2 = weight.shape
    if ci != 3 or k != k2 or (k, N) not in ((3, 64), (7, 96)):
        raise ValueError(f'stem: unsupported weight {tuple(weight.shape)}')
    B, _, H, W = image.shape
    Ho, Wo = stem_out_size(H, W, k)
    if out is None:
        out = torch.empty(B, Ho, Wo, N, device=image.device, dtype=torch.float32)
    elif tuple(out.shape) != (B, Ho, Wo, N):
        raise ValueError('stem: bad out shape')
    w = weight.detach().contiguous()
    b = None if bias is None else bias.detach().contiguous()
    br = _Bracket(f'stem_conv<{k}>', f'stem {H}x{W}', 2.0 * B * Ho * Wo * N * 3 * k * k,
                  4.0 * (B * 3 * H * W + B * Ho * Wo * N)) if _timer is not None else None
    rc = nat.lib().sqd_stem_conv_fwd(nat.ptr(image), nat.ptr(w), nat.ptr(b), nat.ptr(out), B, H, W, N, k, int(bool(relu)),
                                     nat.stream_handle(image.device))
    nat.check(rc, 'sqd_stem_conv_fwd')
    if br is not None:
        br.done()
    return out


def stem_pool(image, weight, bias, argmax=None):
    """Fused conv(3->N,k,s2)+ReLU+MaxPool(3,2,ceil): NCHW image -> NHWC pooled features [B,Hp,Wp,N]."""
    if image.dim() != 4 or image.shape[1] != 3 or image.dtype != torch.float32 or not image.is_cuda:
        raise ValueError(f'stem_pool: image must be fp32 CUDA NCHW with 3 channels, got {tuple(image.shape)}')
    image = image.contiguous()
    N, ci, k, k2 = weight.shape
    if ci != 3 or k != k2 or (k, N) not in ((3, 64), (7, 96)):
        raise ValueError(f'stem_pool: unsupported weight {tuple(weight.shape)}')
    B, _, H, W = image.shape
    Ho, Wo = stem_out_size(H, W, k)
    if Ho < 3 or Wo < 3:
        raise ValueError('stem_pool: input too small')
    Hp, Wp = pool_out_size(Ho, Wo)
    out = torch.empty(B, Hp, Wp, N, device=image.device, dtype=torch.float32)
    if argmax is not None and (tuple(argmax.shape) != (B, Hp, Wp, N) or argmax.dtype != torch.uint8):
        raise ValueError('stem_pool: bad argmax tensor')
    w = weight.detach().contiguous()
    b = None if bias is None else bias.detach().contiguous()
    br = _Bracket(f'stem_pool<{k}>', f'stem+pool {H}x{W}', 2.0 * B * Ho * Wo * N * 3 * k * k,
                  4.0 * (B * 3 * H * W + B * Hp * Wp * N)) if _timer is not None else None
    rc = nat.lib().sqd_stem_conv_relu_pool_fwd(nat.ptr(image), nat.ptr(w), nat.ptr(b), nat.ptr(out), nat.ptr(argmax), B, H, W, N, k,
                                               nat.stream_handle(image.device))
    nat.check(rc, 'sqd_stem_conv_relu_pool_fwd')
    if br is not None:
        br.done()
    return out


def pool_out_size(h, w):
    return (h - 3 + 1) // 2 + 1, (w - 3 + 1) // 2 + 1


def maxpool(x, out=None, argmax=None):
    """MaxPool2d(3, 2, ceil_mode=True) on NHWC; ``argmax`` (uint8, same shape as out) is filled if given."""
    _check_nhwc(x, 'x')
    B, H, W, C = x.shape
    if H < 3 or W < 3 or C % 4:
        raise ValueError('maxpool: need H,W >= 3 and C % 4 == 0')
    Ho, Wo = pool_out_size(H, W)
    if out is None:
        out = torch.empty(B, Ho, Wo, C, device=x.device, dtype=torch.float32)
    if tuple(out.shape) != (B, Ho, Wo, C):
        raise ValueError('maxpool: bad out shape')
    if argmax is not None and (tuple(argmax.shape) != (B, Ho, Wo, C) or argmax.dtype != torch.uint8):
        raise ValueError('maxpool: bad argmax tensor')
    br = _Bracket('maxpool_fwd', f'pool C{C} {H}x{W}', 0.0, 4.0 * B * C * (H * W + Ho * Wo)) if _timer is not None else None
    rc = nat.lib().sqd_maxpool3x3s2_ceil_fwd(nat.ptr(x), nat.ptr(out), nat.ptr(argmax), B, H, W, C, nat.stream_handle(x.device))
    nat.check(rc, 'sqd_maxpool3x3s2_ceil_fwd')
    if br is not None:
        br.done()
    return out


def maxpool_bwd(dy, argmax, in_hw, out=None, relu_src=None):
    _check_nhwc(dy, 'dy')
    B, Ho, Wo, C = dy.shape
    H, W = in_hw
    if pool_out_size(H, W) != (Ho, Wo) or tuple(argmax.shape) != (B, Ho, Wo, C) or argmax.dtype != torch.uint8:
        raise ValueError('maxpool_bwd: geometry mismatch')
    if out is None:
        out = torch.empty(B, H, W, C, device=dy.device, dtype=torch.float32)
    if relu_src is not None and (tuple(relu_src.shape) != (B, H, W, C) or not relu_src.is_contiguous()):
        raise ValueError('maxpool_bwd: relu_src must match the pool input')
    br = _Bracket('maxpool_bwd', f'poolbwd C{C} {H}x{W}', 0.0, 4.0 * B * C * (H * W * 2 + Ho * Wo * 1.25)) if _timer is not None else None
    rc = nat.lib().sqd_maxpool3x3s2_ceil_bwd(nat.ptr(dy), nat.ptr(argmax), nat.ptr(out), nat.ptr(relu_src), B, H, W, C,
                                             nat.stream_handle(dy.device))
    if br is not None:
        br.done()
    nat.check(rc, 'sqd_maxpool3x3s2_ceil_bwd')
    return out


def decode(pred, anchors, input_size, num_classes):
    """pred [B,A,C+5], anchors [A,4] fp32 -> class_ids int64 [B,A], scores [B,A], boxes [B,A,4]."""
    if pred.dim() != 3 or pred.shape[2] != num_classes + 5 or pred.dtype != torch.float32 or not pred.is_cuda:
        raise ValueError(f'decode: bad pred {tuple(pred.shape)}')
    pred = pred.contiguous()
    B, A, _ = pred.shape
    if tuple(anchors.shape) != (A, 4) or anchors.dtype != torch.float32 or anchors.device != pred.device:
        raise ValueError('decode: anchors must be fp32 [A,4] on the same device')
    ids = torch.empty(B, A, device=pred.device, dtype=torch.int64)
    scores = torch.empty(B, A, device=pred.device, dtype=torch.float32)
    boxes = torch.empty(B, A, 4, device=pred.device, dtype=torch.float32)
    rc = nat.lib().sqd_decode_fwd(nat.ptr(pred), nat.ptr(anchors.contiguous()), nat.ptr(ids), nat.ptr(scores), nat.ptr(boxes),
                                  B, A, num_classes, int(input_size[0]), int(input_size[1]), nat.stream_handle(pred.device))
    nat.check(rc, 'sqd_decode_fwd')
    return ids, scores, boxes


def resolve(pred, anchors, input_size, num_classes, log_softmax=False):
    """The reference's PredictionResolver outputs: (probs [B,A,C], log_probs [B,A,C] | None, scores [B,A,1],
    deltas [B,A,4], boxes [B,A,4])."""
    if pred.dim() != 3 or pred.shape[2] != num_classes + 5 or pred.dtype != torch.float32 or not pred.is_cuda:
        raise ValueError(f'resolve: bad pred {tuple(pred.shape)}')
    pred = pred.contiguous()
    B, A, _ = pred.shape
    if tuple(anchors.shape) != (A, 4) or anchors.dtype != torch.float32 or anchors.device != pred.device:
        raise ValueError('resolve: anchors must be fp32 [A,4] on the same device')
    dev = pred.device
    probs = torch.empty(B, A, num_classes, device=dev, dtype=torch.float32)
    logp = torch.empty(B, A, num_classes, device=dev, dtype=torch.float32) if log_softmax else None
    scores = torch.empty(B, A, 1, device=dev, dtype=torch.float32)
    deltas = torch.empty(B, A, 4, device=dev, dtype=torch.float32)
    boxes = torch.empty(B, A, 4, device=dev, dtype=torch.float32)
    rc = nat.lib().sqd_resolve_fwd(nat.ptr(pred), nat.ptr(anchors.contiguous()), nat.ptr(probs), nat.ptr(logp), nat.ptr(scores),
                                   nat.ptr(deltas), nat.ptr(boxes), B, A, num_classes, int(input_size[0]), int(input_size[1]),
                                   nat.stream_handle(dev))
    nat.check(rc, 'sqd_resolve_fwd')
    return probs, logp, scores, deltas, boxes


def _det_buffers(B, K, device, A=None):
    """(count, class_ids, scores, boxes, anchor_idx[, keys workspace]) for the fused detection kernels."""
    bufs = (torch.zeros(B, device=device, dtype=torch.int32), torch.zeros(B, K, device=device, dtype=torch.int64),
            torch.zeros(B, K, device=device, dtype=torch.float32), torch.zeros(B, K, 4, device=device, dtype=torch.float32),
            torch.zeros(B, K, device=device, dtype=torch.int32))
    if A is not None:
        bufs = bufs + (_det_workspace(B, A, device),)
    return bufs


def _det_workspace(B, A, device):
    """Placeholder for the ABI's ``keys_ws`` argument (the one-launch detect kernel keeps its keys in LDS)."""
    return torch.zeros(4, device=device, dtype=torch.int32)


def detect(pred, anchors, input_size, num_classes, keep_top_k=64, nms_thresh=0.4, score_thresh=0.3, scales=None, out=None):
    """Fused decode + top-k + class-wise NMS + threshold for a batch.
    Returns (count int32 [B], class_ids int64 [B,K], scores [B,K], boxes [B,K,4], anchor_idx int32 [B,K])."""
    if pred.dim() != 3 or pred.shape[2] != num_classes + 5 or pred.dtype != torch.float32 or not pred.is_cuda:
        raise ValueError(f'detect: bad pred {tuple(pred.shape)}')
    pred = pred.contiguous()
    B, A, _ = pred.shape
    if tuple(anchors.shape) != (A, 4) or anchors.dtype != torch.float32 or anchors.device != pred.device:
        raise ValueError('detect: anchors must be fp32 [A,4] on the same device')
    if scales is not None and (tuple(scales.shape) != (B, 2) or scales.dtype != torch.float32 or scales.device != pred.device):
        raise ValueError('detect: scales must be fp32 [B,2] (sy, sx)')
    bufs = out if out is not None else _det_buffers(B, keep_top_k, pred.device, A)
    if len(bufs) == 5:
        bufs = tuple(bufs) + (_det_workspace(B, A, pred.device),)
    cnt, cls, sc, bx, idx, keys = bufs
    if keys.dtype != torch.int32 or keys.device != pred.device:
        raise ValueError('detect: workspace must be an int32 tensor on the same device')
    br = _Bracket('detect', f'detect A{A}', 0.0, 4.0 * B * A * (num_classes + 5)) if _timer is not None else None
    rc = nat.lib().sqd_detect_fwd(nat.ptr(pred), nat.ptr(anchors.contiguous()), nat.ptr(scales), nat.ptr(keys), nat.ptr(cnt), nat.ptr(cls),
                                  nat.ptr(sc), nat.ptr(bx), nat.ptr(idx), B, A, num_classes, int(input_size[0]),
                                  int(input_size[1]), int(keep_top_k), float(nms_thresh), float(score_thresh),
                                  nat.stream_handle(pred.device))
    nat.check(rc, 'sqd_detect_fwd')
    if br is not None:
        br.done()
    return bufs[:5]


def filter_dense(class_ids, scores, boxes, num_classes, keep_top_k=64, nms_thresh=0.4, score_thresh=0.3):
    """``Detector.filter`` on already decoded dense tensors ([B,A] / [B,A,4])."""
    if scores.dim() != 2 or class_ids.shape != scores.shape or tuple(boxes.shape) != tuple(scores.shape) + (4,):
        raise ValueError('filter: shape mismatch')
    if class_ids.dtype != torch.int64 or scores.dtype != torch.float32 or boxes.dtype != torch.float32 or not scores.is_cuda:
        raise ValueError('filter: dtype/device mismatch')
    B, A = scores.shape
    bufs = _det_buffers(B, keep_top_k, scores.device, A)
    cnt, cls, sc, bx, idx, keys = bufs
    rc = nat.lib().sqd_filter_fwd(nat.ptr(class_ids.contiguous()), nat.ptr(scores.contiguous()), nat.ptr(boxes.contiguous()),
                                  nat.ptr(keys), nat.ptr(cnt), nat.ptr(cls), nat.ptr(sc), nat.ptr(bx), nat.ptr(idx), B, A, num_classes,
                                  int(keep_top_k), float(nms_thresh), float(score_thresh), nat.stream_handle(scores.device))
    nat.check(rc, 'sqd_filter_fwd')
    return bufs[:5]


# ---------------------------------------------------------------------------------------------
# training-side ops
# ---------------------------------------------------------------------------------------------
_TARGET_WGS = 1536          # workgroups a 3x3 weight-gradient launch aims for (measured sweep 768 / 1536 / 3072)
_TARGET_WGS_1X1 = 512       # 1x1: fewer, longer pixel streams (less slab traffic per MFMA; round 2, inside the training step: 512 6.26 ms, 1024 6.29, 384 6.40)


WINO_WGRAD = True          # 3x3 weight gradients: Winograd kernel where it applies (N % 64 == 0)
_TARGET_WGS_WINO = 512      # Winograd wgrad: one resident round (two 4-wave workgroups per CU), every workgroup the same work


def wgrad_uses_wino(N, C, taps, B, H, W, wino=None):
    """Whether conv_wgrad runs the Winograd F(2x2,3x3) kernel for this layer (3x3, N % 64 == 0; ``wino`` overrides the
    module default WINO_WGRAD)."""
    return bool(WINO_WGRAD if wino is None else wino) and taps == 9 and (N % 64 == 0 or N <= 80) and N % 4 == 0 and C % 4 == 0


def _wino_wgrad_tc(N, C):
    """Input-channel blocks of 16 per workgroup of the Winograd wgrad kernel: 2 (32 channels) unless that would leave the
    last block half empty (C = 16, 48, ...: measured 50 vs 62 us on C48 -> N192) or the 5-block ConvDet variant runs."""
    return 1 if (N % 64 or C % 32 == 16 or C < 32) else 2


def wgrad_split(N, C, taps, B, H, W, wino=None):
    """(S, slab stride): number of split-K partial slabs the weight-gradient kernel writes for this layer, floats per slab.
    S comes from the workgroup targets below unless the measured table has a row 'G:taps:N:C:npix' (tools/tune_insitu.py --mode
    train: the split of each layer tried inside the training step)."""
    tuned = _tuning().get(f'G:{taps}:{N}:{C}:{B * H * W}')
    if wgrad_uses_wino(N, C, taps, B, H, W, wino):
        ngroups = B * -(-H // 4) * -(-W // 16)                    # 4x16-pixel groups = the K axis of the 16 position GEMMs
        # (out-channel, in-channel) blocks of dU per workgroup: 64 x 16|32, or all of N <= 80 x 16 (ConvDet)
        blocks = -(-C // 16) if N % 64 else (N // 64) * -(-C // (16 * _wino_wgrad_tc(N, C)))
        S = max(1, min(ngroups, _TARGET_WGS_WINO // blocks if blocks <= _TARGET_WGS_WINO else 1))
        if tuned is not None and tuned >= 1:
            S = max(1, min(ngroups, int(tuned)))
        return S, N * taps * C + N
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
        nblocks = -(-(B * H * W) // 128)
    groups = -(-N // (tn * 16)) * -(-C // (tc * 16))
    S = max(1, min(nblocks, (_TARGET_WGS if taps == 9 else _TARGET_WGS_1X1) // groups, 256))
    if tuned is not None and tuned >= 1:
        S = max(1, min(nblocks, int(tuned), 256))
    return S, N * taps * C + N


def conv_wgrad(dy, dy_coff, N, x, x_coff, C, taps, slab=None, wino=None):
    """(dW OIHW [N,C,k,k], db [N]) from dy[..., dy_coff:dy_coff+N] (already ReLU-masked) and
    x[..., x_coff:x_coff+C].  With ``slab`` (a workspace view of S * stride floats, see ``WgradBatch``) only the partial
    slabs are written and None is returned: the caller reduces all layers with one launch."""
    _check_nhwc(dy, 'dy'); _check_nhwc(x, 'x')
    B, H, W, dyp = dy.shape
    if tuple(x.shape[:3]) != (B, H, W):
        raise ValueError('wgrad: dy and x disagree on B,H,W')
    xp = x.shape[3]
    if dy_coff + N > dyp or x_coff + C > xp or N % 4 or C % 4 or taps not in (1, 9):
        raise ValueError('wgrad: channel window out of range')
    k = 3 if taps == 9 else 1
    use_wino = wgrad_uses_wino(N, C, taps, B, H, W, wino)
    S, stride = wgrad_split(N, C, taps, B, H, W, wino)
    deferred = slab is not None
    if deferred:
        if slab.numel() != S * stride or not slab.is_contiguous() or slab.dtype != torch.float32:
            raise ValueError('wgrad: slab workspace does not match this layer')
        dw = db = None
    else:
        slab = torch.empty(S * stride, device=dy.device, dtype=torch.float32)
        dw = torch.empty(N, C, k, k, device=dy.device, dtype=torch.float32)
        db = torch.empty(N, device=dy.device, dtype=torch.float32)
    if use_wino:
        # executed multiply-adds = direct form / 2.25 (16 position GEMMs per 2x2 tile)
        br = _Bracket('conv_wgrad_wino', f'wgrad 9tap C{C} N{N} {H}x{W}', 2.0 * B * H * W * N * C * 4,
                      4.0 * (B * H * W * (C + N) + 2 * S * stride)) if _timer is not None else None
        rc = nat.lib().sqd_conv_wgrad_wino(nat.ptr(dy), nat.ptr(x), nat.ptr(slab), nat.ptr(dw), nat.ptr(db), B, H, W, N, dyp, dy_coff,
                                           C, xp, x_coff, S, _wino_wgrad_tc(N, C), nat.stream_handle(dy.device))
        nat.check(rc, 'sqd_conv_wgrad_wino')
    else:
        br = _Bracket(f'conv_wgrad<{taps}>', f'wgrad {taps}tap C{C} N{N} {H}x{W}', 2.0 * B * H * W * N * C * taps,
                      4.0 * (B * H * W * (C + N) + 2 * S * stride)) if _timer is not None else None
        rc = nat.lib().sqd_conv_wgrad(nat.ptr(dy), nat.ptr(x), nat.ptr(slab), nat.ptr(dw), nat.ptr(db), B, H, W, N, dyp, dy_coff,
                                      C, xp, x_coff, taps, S, nat.stream_handle(dy.device))
        nat.check(rc, 'sqd_conv_wgrad')
    if br is not None:
        br.done()
    return None if deferred else (dw, db)


_WGR_OUT = 64               # outputs per workgroup of the slab-reduction kernels (csrc/wgrad.hip WGR_OUT)


class WgradBatch:
    """Workspace + descriptor table for reducing the partial slabs of many conv weight gradients with ONE launch into a
    flat gradient buffer.  ``entries``: [(key, N, C, taps, B, H, W, dw_offset, db_offset)] (offsets in floats into the flat
    buffer).  Slab workspace and table are allocated once and reused every step (pointer-stable)."""

    def __init__(self, entries, device):
        rows, self.slabs, off, blk = [], {}, 0, 0
        self.row_blocks = [0]                  # first workgroup of every record (+ the total at the end)
        for key, N, C, taps, B, H, W, dw_off, db_off in entries:
            S, stride = wgrad_split(N, C, taps, B, H, W)
            rows.append([off, dw_off, db_off, S, stride, N, C, taps, blk])
            self.slabs[key] = (off, S * stride)
            off += S * stride
            blk += -(-stride // _WGR_OUT)
            self.row_blocks.append(blk)
        self.total_blocks = blk
        self.row_of = {key: i for i, (key, *_rest) in enumerate(entries)}
        self.workspace = torch.empty(off, device=device, dtype=torch.float32)
        self.table = torch.tensor(rows, dtype=torch.int64).to(device)
        self.nrows = len(rows)
        self.bytes = 4.0 * off

    def slab(self, key):
        off, n = self.slabs[key]
        return self.workspace[off:off + n]

    def reduce(self, grad_flat, row_lo=0, row_hi=None):
        """Reduce the slabs of records [row_lo, row_hi) (default: all) into ``grad_flat``."""
        row_hi = self.nrows if row_hi is None else row_hi
        if not (0 <= row_lo < row_hi <= self.nrows):
            raise ValueError('WgradBatch.reduce: bad record range')
        nrec = row_hi - row_lo
        b0, b1 = self.row_blocks[row_lo], self.row_blocks[row_hi]
        br = _Bracket('wgrad_reduce_batched', f'{nrec} layers', 0.0, self.bytes * (b1 - b0) / max(self.total_blocks, 1)) if _timer is not None else None
        if nrec == self.nrows:
            rc = nat.lib().sqd_wgrad_reduce_batched(nat.ptr(self.table), self.nrows, self.total_blocks, nat.ptr(self.workspace),
                                                    nat.ptr(grad_flat), nat.stream_handle(grad_flat.device))
        else:
            rc = nat.lib().sqd_wgrad_reduce_batched_range(nat.c_p(self.table.data_ptr() + row_lo * 9 * 8), nrec, b0, b1 - b0,
                                                          nat.ptr(self.workspace), nat.ptr(grad_flat), nat.stream_handle(grad_flat.device))
        nat.check(rc, 'sqd_wgrad_reduce_batched')
        if br is not None:
            br.done()


def _check_stem_out(dw, db, N, ksize):
    if tuple(dw.shape) != (N, 3, ksize, ksize) or tuple(db.shape) != (N,) or not dw.is_contiguous() or not db.is_contiguous() \
            or dw.dtype != torch.float32 or db.dtype != torch.float32:
        raise ValueError('stem wgrad: out=(dw, db) must be contiguous fp32 [N,3,k,k] / [N]')


def stem_wgrad(dy, image, N, ksize, out=None):
    """(dW [N,3,k,k], db [N]) of the stem from dy NHWC [B,Ho,Wo,N] (ReLU-masked) and the NCHW image."""
    _check_nhwc(dy, 'dy')
    B, Ho, Wo, n = dy.shape
    if n != N or image.dim() != 4 or image.shape[0] != B or image.shape[1] != 3 or not image.is_contiguous():
        raise ValueError('stem_wgrad: geometry mismatch')
    H, W = image.shape[2], image.shape[3]
    if stem_out_size(H, W, ksize) != (Ho, Wo) or (ksize, N) not in ((3, 64), (7, 96)):
        raise ValueError('stem_wgrad: unsupported geometry')
    nblocks = B * -(-Ho // 8) * -(-Wo // 16)
    S = max(1, min(nblocks, 1024))
    K = 3 * ksize * ksize
    slab = torch.empty(S * (N * K + N), device=dy.device, dtype=torch.float32)
    dw, db = out if out is not None else (torch.empty(N, 3, ksize, ksize, device=dy.device, dtype=torch.float32),
                                          torch.empty(N, device=dy.device, dtype=torch.float32))
    _check_stem_out(dw, db, N, ksize)
    br = _Bracket(f'stem_wgrad<{ksize}>', f'stem wgrad {H}x{W}', 2.0 * B * Ho * Wo * N * K,
                  4.0 * (B * Ho * Wo * N + B * 3 * H * W)) if _timer is not None else None
    rc = nat.lib().sqd_stem_wgrad(nat.ptr(dy), nat.ptr(image), nat.ptr(slab), nat.ptr(dw), nat.ptr(db), B, H, W, N, ksize, S,
                                  nat.stream_handle(dy.device))
    nat.check(rc, 'sqd_stem_wgrad')
    if br is not None:
        br.done()
    return dw, db


def stem_wgrad_pooled(dpool, pooled, argmax, image, N, ksize, out=None):
    """Stem (dW, db) when the forward ran fused (stem_pool with argmax): ReLU + max-pool backward folded in."""
    _check_nhwc(dpool, 'dpool'); _check_nhwc(pooled, 'pooled')
    B, Hp, Wp, n = dpool.shape
    if n != N or tuple(pooled.shape) != (B, Hp, Wp, N) or tuple(argmax.shape) != (B, Hp, Wp, N) or argmax.dtype != torch.uint8 \
            or not argmax.is_contiguous():
        raise ValueError('stem_wgrad_pooled: dpool / pooled / argmax geometry mismatch')
    if image.dim() != 4 or image.shape[0] != B or image.shape[1] != 3 or not image.is_contiguous():
        raise ValueError('stem_wgrad_pooled: bad image')
    H, W = image.shape[2], image.shape[3]
    Ho, Wo = stem_out_size(H, W, ksize)
    if pool_out_size(Ho, Wo) != (Hp, Wp) or (ksize, N) not in ((3, 64), (7, 96)):
        raise ValueError('stem_wgrad_pooled: unsupported geometry')
    nblocks = B * -(-Ho // 8) * -(-Wo // 16)
    S = max(1, min(nblocks, 1024))
    K = 3 * ksize * ksize
    slab = torch.empty(S * (N * K + N), device=dpool.device, dtype=torch.float32)
    dw, db = out if out is not None else (torch.empty(N, 3, ksize, ksize, device=dpool.device, dtype=torch.float32),
                                          torch.empty(N, device=dpool.device, dtype=torch.float32))
    _check_stem_out(dw, db, N, ksize)
    br = _Bracket(f'stem_wgrad_pooled<{ksize}>', f'stem wgrad (pooled) {H}x{W}', 2.0 * B * Ho * Wo * N * K,
                  4.0 * (B * Hp * Wp * N * 2.25 + B * 3 * H * W)) if _timer is not None else None
    rc = nat.lib().sqd_stem_wgrad_pooled(nat.ptr(dpool), nat.ptr(pooled), nat.ptr(argmax), nat.ptr(image), nat.ptr(slab), nat.ptr(dw),
                                         nat.ptr(db), B, H, W, N, ksize, S, nat.stream_handle(dpool.device))
    nat.check(rc, 'sqd_stem_wgrad_pooled')
    if br is not None:
        br.done()
    return dw, db


def _check_loss_args(pred, gt, anchors, num_classes):
    if pred.dim() != 3 or pred.shape[2] != num_classes + 5 or pred.dtype != torch.float32 or not pred.is_cuda:
        raise ValueError(f'loss: bad pred {tuple(pred.shape)}')
    B, A, _ = pred.shape
    if tuple(gt.shape) != (B, A, num_classes + 9) or gt.dtype != torch.float32 or gt.device != pred.device:
        raise ValueError(f'loss: gt must be fp32 [B,A,C+9] on the same device, got {tuple(gt.shape)}')
    if tuple(anchors.shape) != (A, 4) or anchors.dtype != torch.float32 or anchors.device != pred.device:
        raise ValueError('loss: anchors must be fp32 [A,4] on the same device')
    return B, A


def loss_fwd(pred, gt, anchors, input_size, num_classes, weights):
    """-> (losses [4,B] = class, score, bbox, total; nobj [B])."""
    B, A = _check_loss_args(pred, gt, anchors, num_classes)
    pred, gt, anchors = pred.contiguous(), gt.contiguous(), anchors.contiguous()
    ws = torch.empty(B * 16 * 5, device=pred.device, dtype=torch.float32)
    losses = torch.empty(4, B, device=pred.device, dtype=torch.float32)
    nobj = torch.empty(B, device=pred.device, dtype=torch.float32)
    br = _Bracket('loss_fwd', f'loss A{A}', 0.0, 4.0 * B * A * (2 * num_classes + 14)) if _timer is not None else None
    rc = nat.lib().sqd_loss_fwd(nat.ptr(pred), nat.ptr(gt), nat.ptr(anchors), nat.ptr(ws), nat.ptr(losses), nat.ptr(nobj), B, A,
                                num_classes, int(input_size[0]), int(input_size[1]), *[float(w) for w in weights],
                                nat.stream_handle(pred.device))
    nat.check(rc, 'sqd_loss_fwd')
    if br is not None:
        br.done()
    return losses, nobj


def loss_bwd(pred, gt, anchors, nobj, coef, input_size, num_classes, weights):
    """coef [3,B] -> dpred [B,A,C+5]."""
    B, A = _check_loss_args(pred, gt, anchors, num_classes)
    if tuple(coef.shape) != (3, B) or tuple(nobj.shape) != (B,):
        raise ValueError('loss_bwd: coef must be [3,B], nobj [B]')
    pred, gt, anchors, coef = pred.contiguous(), gt.contiguous(), anchors.contiguous(), coef.contiguous().float()
    dpred = torch.empty_like(pred)
    br = _Bracket('loss_bwd', f'lossbwd A{A}', 0.0, 4.0 * B * A * (3 * num_classes + 19)) if _timer is not None else None
    rc = nat.lib().sqd_loss_bwd(nat.ptr(pred), nat.ptr(gt), nat.ptr(anchors), nat.ptr(nobj), nat.ptr(coef), nat.ptr(dpred), B, A,
                                num_classes, int(input_size[0]), int(input_size[1]), *[float(w) for w in weights],
                                nat.stream_handle(pred.device))
    nat.check(rc, 'sqd_loss_bwd')
    if br is not None:
        br.done()
    return dpred


def encode_gt(boxes, class_ids, box_offsets, anchors64, num_classes, dense=True, parallel=True):
    """On-device GT encoding (compute_deltas + prepare_annotations, src/utils/boxes.py:84-135,
    src/datasets/base.py:61-76).  boxes [total,4] fp32 xyxy, class_ids [total] i32, box_offsets [B+1] i32,
    anchors64 [A,4] float64 -- all on the GPU.  -> (gt [B,A,C+9] or None, anchor_idx [total] i32, deltas [total,4])."""
    if boxes.dtype != torch.float32 or class_ids.dtype != torch.int32 or box_offsets.dtype != torch.int32:
        raise ValueError('encode_gt: boxes must be float32, class_ids / box_offsets int32')
    if anchors64.dtype != torch.float64 or anchors64.dim() != 2 or anchors64.shape[1] != 4:
        raise ValueError('encode_gt: anchors must be float64 [A,4] (the reference matches anchors in float64)')
    if boxes.dim() != 2 or boxes.shape[1] != 4 or class_ids.shape[0] != boxes.shape[0] or box_offsets.dim() != 1:
        raise ValueError('encode_gt: boxes [total,4], class_ids [total], box_offsets [B+1]')
    B, A, total = box_offsets.shape[0] - 1, anchors64.shape[0], boxes.shape[0]
    if B < 1:
        raise ValueError('encode_gt: empty batch')
    boxes, class_ids, box_offsets, anchors64 = boxes.contiguous(), class_ids.contiguous(), box_offsets.contiguous(), anchors64.contiguous()
    dev = boxes.device
    gt = torch.empty(B, A, num_classes + 9, device=dev, dtype=torch.float32) if dense else None
    idx = torch.empty(max(total, 1), device=dev, dtype=torch.int32)
    deltas = torch.empty(max(total, 1), 4, device=dev, dtype=torch.float32)
    br = _Bracket('encode_gt', f'gt A{A}', 0.0, 4.0 * B * A * (num_classes + 9)) if _timer is not None else None
    ws = torch.empty(max(total, 1) * 2, device=dev, dtype=torch.float64)       # 16 bytes per box: first-choice candidates
    rc = nat.lib().sqd_encode_gt_fwd(nat.ptr(boxes) if total else None, nat.ptr(class_ids) if total else nat.ptr(idx),
                                     nat.ptr(box_offsets), nat.ptr(anchors64), nat.ptr(gt) if dense else None, nat.ptr(idx),
                                     nat.ptr(deltas), nat.ptr(ws) if (total and parallel) else None, int(total), B, A,
                                     int(num_classes), nat.stream_handle(dev))
    nat.check(rc, 'sqd_encode_gt_fwd')
    if br is not None:
        br.done()
    return gt, idx[:total], deltas[:total]
