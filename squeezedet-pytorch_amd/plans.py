"""Operand plans of the HIP kernels: packed / transformed weight copies (``ConvPlan``, ``WinoPlan``, ``FusedExpandPlan``,
``FireWinoPlan``, ``FireBridgePlan``), their batched one-launch refresh after an optimizer step, and the split-K slab
workspace of the weight gradients (``WgradBatch``).  Built once per (parameter, configuration), pointer-stable afterwards so
hipGraph replays stay valid."""
from __future__ import annotations

import torch

from . import _native as nat
from . import timing
from .timing import _Bracket
from .tiles import (cfg_table, wino_cfgs, fused_expand_cfgs, fire_wino_cfg_ok, fire_bridge_cfg_ok, fire_pool_bridge_ok,
                    wgrad_split)


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
    rc = nat.lib().sqd_pack_conv_weights_batched(nat.ptr(table), len(rows), 96, nat.stream_handle(dev))
    nat.check(rc, 'sqd_pack_conv_weights_batched')
    return table


def dgrad_weight(w_oihw):
    """Weights of the convolution that computes dX from dY: swap in/out channels, flip taps."""
    return w_oihw.permute(1, 0, 2, 3).flip(2, 3).contiguous()


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


_SK_SCHEDULES = {}


def wino_sk_params(N=None, C=None):
    """(minseg, h_bias_pm, ksplit) of the balanced Winograd schedule for a C -> N layer: shortest part of a cut unit in stages,
    per-mille correction of the 16-channel class's share of the grid, and how units are cut: ksplit = k >= 1 cuts every unit at the
    same K boundaries into k parts dealt round-robin (the workgroups of a round walk the same K chunks together: the transformed
    weights they stage are shared through the L2), 0 = contiguous runs of exactly equal length.  Measured on the 24x78 bs=20 shapes
    (profiles/r04b_sk_schedule_variants.log): a long reduction (ConvDet, 96 chunks) is fastest cut in four (215 us against 241 for
    the unit kernel and 270 for contiguous runs, whose workgroups sit at 512 different K positions); short reductions are fastest
    uncut (k = 1), where the round-robin deal alone beats the unit kernel's grid rounding on many-slice layers (C72 -> N768: 177
    against 231 us).  (Sweeps replace this function: tools/sk_bench.py.)"""
    ks = 1 if (C is None or C // 8 < 48) else 4
    return 2, 1000, ks


def wino_sk_grid():
    """Workgroups of a balanced Winograd launch: every resident slot of the device (``SQD_SK_GRID`` overrides it: tests force
    many / few parts per unit with it; any grid gives the same results up to the summation order of cut units)."""
    import os
    g = int(os.environ.get('SQD_SK_GRID', 0))
    return g if g > 0 else int(nat.lib().sqd_wino_sk_grid())


def wino_sk_host_schedule(ngroups, N, C, G, minseg=2, h_bias_pm=1000, ksplit=0):
    """sqd_wino_sk_schedule as numpy arrays: (seg_off int32 [G + 1], segs int32 [nsegs, 8], nslabs).  Pure host code."""
    import ctypes
    import numpy as np
    nchunks, nslices = C // 8, -(-N // 32)
    max_segs = G * (2 + 2 * (-(-(ngroups * nslices * nchunks) // (4 * G * nchunks)) + 1)) + 16 + (ngroups + 4) * nslices * max(ksplit, 1)
    seg_off = np.zeros(G + 1, dtype=np.int32)
    segs = np.zeros((max_segs, 8), dtype=np.int32)
    ns, nslabs = ctypes.c_int(), ctypes.c_int()
    rc = nat.lib().sqd_wino_sk_schedule(int(ngroups), int(N), int(C), int(G), int(minseg), int(h_bias_pm), int(ksplit),
                                        seg_off.ctypes.data_as(ctypes.c_void_p),
                                        segs.ctypes.data_as(ctypes.c_void_p), max_segs, ctypes.byref(ns), ctypes.byref(nslabs))
    nat.check(rc, 'sqd_wino_sk_schedule')
    return seg_off, segs[:ns.value].copy(), nslabs.value


class WinoSkSchedule:
    """Device copy of the balanced schedule of one (pixel geometry, N, C) layer shape, shared by every layer of that shape, + the
    partial-slab workspaces and arrival counters of its launches (zeroed once; every launch leaves its counters zero).  A workspace
    belongs to ONE stream: launches on a stream are ordered, launches on different streams (two inference lanes, two user threads)
    may overlap and must not share slabs or tickets, so ``workspace()`` hands out one set per launch stream."""

    def __init__(self, ngroups, N, C, device):
        self.G = wino_sk_grid()
        minseg, hb, ks = wino_sk_params(N, C)
        seg_off, segs, nslabs = wino_sk_host_schedule(ngroups, N, C, self.G, minseg, hb, ks)
        self.device = torch.device(device)
        self.seg_off = torch.from_numpy(seg_off).to(device)
        self.segs = torch.from_numpy(segs).contiguous().to(device)
        self.nslabs = nslabs
        self._ws = {}

    def workspace(self):
        """(slab workspace, arrival counters) of the stream the launch is about to be enqueued on."""
        key = torch.cuda.current_stream(self.device).cuda_stream
        hit = self._ws.get(key)
        if hit is None:
            if len(self._ws) > 16:
                self._ws.clear()
            hit = self._ws[key] = (torch.empty(max(self.nslabs, 1) * 8192, device=self.device, dtype=torch.float32),
                                   torch.zeros(max(self.nslabs, 1) * 4, device=self.device, dtype=torch.int32))
        return hit

    # (single-stream callers and the tests read these)
    @property
    def ws(self):
        return self.workspace()[0]

    @property
    def cnt(self):
        return self.workspace()[1]


def wino_sk_schedule(ngroups, N, C, device):
    key = (int(ngroups), int(N), int(C), str(device), wino_sk_grid()) + wino_sk_params(N, C)
    hit = _SK_SCHEDULES.get(key)
    if hit is None:
        if len(_SK_SCHEDULES) > 64:
            _SK_SCHEDULES.clear()
        hit = _SK_SCHEDULES[key] = WinoSkSchedule(ngroups, N, C, device)
    return hit


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
    nat.check(nat.lib().sqd_pack_wino_weights_batched(nat.ptr(table), len(rows), 96, nat.stream_handle(dev)), 'sqd_pack_wino_weights_batched')
    return table


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


class FireBridgePlan:
    """Operands of ``fire_bridge``: the Fire's expand pair transformed as in FireWinoPlan, the per-pass bias table, and the next
    Fire's squeeze weights laid out as MFMA A operands (include/sqd_hip.h, sqd_fire_bridge_fwd).  The buffers are allocated once;
    ``refresh_bridge_plans`` rewrites them in place after the parameters changed (training: every step)."""
    __slots__ = ('cfg_id', 'C', 'N3', 'N1', 'Npad', 'Nsq', 'w', 'aux', 'bias_tab', 'sq_ops', 'sq_bias', 'pooled', 'map_u', 'map_aux')

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
        self.cfg_id, self.C, self.N3, self.N1, self.Nsq = cfg_id, C, N3, N1, Nsq
        P3, P1 = -(-N3 // 32), -(-N1 // 128)
        self.Npad = 32 * (P3 + P1)
        self.map_u = self.map_aux = None
        u, so, bt, sb = self._build(w1, b1, w3, b3, wsq, bsq)
        self.w = u
        self.aux = torch.cat([so.reshape(-1), bt.reshape(-1), sb.reshape(-1)]).contiguous()       # one buffer: one gather record
        n0, n1 = so.numel(), so.numel() + bt.numel()
        self.sq_ops = self.aux[:n0].view(so.shape)
        self.bias_tab = self.aux[n0:n1].view(bt.shape)
        self.sq_bias = self.aux[n1:].view(sb.shape)
        if b1 is not None and b3 is not None and bsq is not None:
            self.make_maps()           # (here, not at the first refresh: that one may run inside a hipGraph capture, where host-built tensors are illegal)

    def _build(self, w1, b1, w3, b3, wsq, bsq):
        """(u, sq_ops, bias_tab, sq_bias) of these parameter values."""
        C, N3, N1, Nsq, cfg_id, pooled = self.C, self.N3, self.N1, self.Nsq, self.cfg_id, self.pooled
        P3, P1 = -(-N3 // 32), -(-N1 // 128)
        dev = w3.device
        u = torch.empty(C // 8, 16, self.Npad, 8, device=dev, dtype=torch.float32)
        nat.check(nat.lib().sqd_pack_wino_fire(nat.ptr(w3.detach().contiguous()), nat.ptr(w1.detach().contiguous()), nat.ptr(u),
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
        sq_ops = g.permute(2, 4, 0, 3, 1).contiguous()                                                        # [blk][t][q][g][lr]
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
        sq_bias = (torch.zeros(Nsq, device=dev) if bsq is None else bsq.detach().float()).contiguous()
        return u, sq_ops, bt.contiguous(), sq_bias

    def make_maps(self):
        """Index maps of the in-place refresh (sqd_gather_pack_batched), derived ONCE by pushing coded parameter values through
        ``_build``: every operand element besides the expand3x3 transform is a copy (or +-0.25 times a copy) of one parameter element
        at a position that depends on the shapes only.  Sources: 0 = next squeeze weight / expand1x1 weight (u), 1 = expand1x1 bias,
        2 = expand3x3 bias, 3 = next squeeze bias."""
        if self.map_aux is not None:
            return
        C, N3, N1, Nsq = self.C, self.N3, self.N1, self.Nsq
        dev = self.w.device
        assert max(N1 * C, Nsq * (N1 + N3)) < (1 << 22)
        code = lambda n, base=0: torch.arange(1 + base, n + 1 + base, device=dev, dtype=torch.float32)
        w1c = code(N1 * C).view(N1, C, 1, 1)
        wsc = code(Nsq * (N1 + N3)).view(Nsq, N1 + N3, 1, 1)
        z3 = torch.zeros(N3, C, 3, 3, device=dev)
        u, so, bt, sb = self._build(w1c, code(N1), z3, code(N3, N1), wsc, code(Nsq))
        # u: only the virtual (expand1x1) channels belong to the gather; the expand3x3 part is the batched Winograd transform's
        uq = torch.round(u * 4.0).to(torch.int64)                                       # +-(element + 1), 0 where padded
        mu = torch.where(uq == 0, torch.full_like(uq, -1), (uq.abs() - 1) | torch.where(uq > 0, 1 << 28, 2 << 28))
        virt = torch.zeros(self.Npad // 16, dtype=torch.bool, device=dev)
        virt[-(-N3 // 32) * 2:] = True
        mu = torch.where(virt.view(1, 1, -1, 1, 1, 1, 1), mu.view(C // 8, 8, self.Npad // 16, 4, 16, 2, 2), torch.full_like(mu, -2).view(C // 8, 8, self.Npad // 16, 4, 16, 2, 2))
        self.map_u = mu.reshape(-1).to(torch.int32).contiguous()
        so_i = torch.round(so).to(torch.int64)
        m_so = torch.where(so_i == 0, torch.full_like(so_i, -1), so_i - 1)              # source 0
        bt_i = torch.round(bt).to(torch.int64)
        m_bt = torch.where(bt_i == 0, torch.full_like(bt_i, -1),
                           torch.where(bt_i <= N1, (bt_i - 1) | (1 << 26), (bt_i - 1 - N1) | (2 << 26)))
        m_sb = (torch.round(sb).to(torch.int64) - 1) | (3 << 26)
        self.map_aux = torch.cat([m_so.reshape(-1), m_bt.reshape(-1), m_sb.reshape(-1)]).to(torch.int32).contiguous()
        assert self.map_aux.numel() == self.aux.numel() and self.map_u.numel() == self.w.numel()


def refresh_bridge_plans(items):
    """In-place refresh of Fire-bridge operands after their parameters changed: ``items`` = [(FireBridgePlan, w1, b1, w3, b3, wsq, bsq)].
    Two launches whatever the number of bridges: the batched Winograd transform (expand3x3 parts) and the batched gather (the rest)."""
    if not items:
        return None
    wrows, grows = [], []
    for plan, w1, b1, w3, b3, wsq, bsq in items:
        plan.make_maps()
        for t in (w1, b1, w3, b3, wsq, bsq):
            if t is None or not t.is_contiguous() or t.dtype != torch.float32:
                raise ValueError('refresh_bridge_plans: parameters must be contiguous fp32 tensors (biases included)')
        wrows.append([w3.data_ptr(), plan.w.data_ptr(), plan.N3, plan.C, plan.Npad, 0, (plan.C // 8) * plan.Npad * 8])
        grows.append([plan.w.data_ptr(), plan.map_u.data_ptr(), plan.w.numel(), w1.data_ptr(), 0, 0, 0])
        grows.append([plan.aux.data_ptr(), plan.map_aux.data_ptr(), plan.aux.numel(), wsq.data_ptr(), b1.data_ptr(), b3.data_ptr(), bsq.data_ptr()])
    dev = items[0][0].w.device
    key = ('bridge', str(dev), tuple(tuple(r) for r in wrows), tuple(tuple(r) for r in grows))
    tables = _PACK_TABLES.get(key)
    if tables is None:
        if len(_PACK_TABLES) > 16:
            _PACK_TABLES.clear()
        tables = (torch.tensor(wrows, dtype=torch.int64).to(dev), torch.tensor(grows, dtype=torch.int64).to(dev))
        _PACK_TABLES[key] = tables
    s = nat.stream_handle(dev)
    nat.check(nat.lib().sqd_pack_wino_weights_batched(nat.ptr(tables[0]), len(wrows), 16, s), 'sqd_pack_wino_weights_batched')
    nat.check(nat.lib().sqd_gather_pack_batched(nat.ptr(tables[1]), len(grows), 16, s), 'sqd_gather_pack_batched')
    return tables


_WGR_OUT = 64               # outputs per workgroup of the slab-reduction kernels (csrc/wgrad.hip WGR_OUT)


class WgradBatch:
    """Workspace + descriptor table for reducing the partial slabs of many conv weight gradients with ONE launch into a
    flat gradient buffer.  ``entries``: [(key, N, C, taps, B, H, W, dw_offset, db_offset[, fused[, group]])] (offsets in floats into the
    flat buffer; ``fused``: slabs written by ``ops.squeeze_bwd``; ``group``: a ``tiles.wino_wgrad_groups`` value).  Slab workspace and table are allocated once and reused every step (pointer-stable)."""

    def __init__(self, entries, device):
        rows, self.slabs, off, blk = [], {}, 0, 0
        self.row_blocks = [0]                  # first workgroup of every record (+ the total at the end)
        self.fused, self.group_of, self.splits = {}, {}, {}
        for key, N, C, taps, B, H, W, dw_off, db_off, *flags in entries:
            fused = bool(flags and flags[0])       # the layer's slabs come from ops.squeeze_bwd (its own split)
            self.fused[key] = fused
            grp = flags[1] if len(flags) > 1 else None      # (group id, S, tc, member keys): the layer runs in a grouped launch
            if grp is not None:
                self.group_of[key] = grp
            S, stride = wgrad_split(N, C, taps, B, H, W, fused_dgrad=fused, group_S=None if grp is None else grp[1])
            self.splits[key] = S
            rows.append([off, dw_off, db_off, S, stride, N, C, taps, blk])
            self.slabs[key] = (off, S * stride)
            off += S * stride
            blk += -(-stride // _WGR_OUT)
            self.row_blocks.append(blk)
        self.total_blocks = blk
        self.row_of = {e[0]: i for i, e in enumerate(entries)}
        self.workspace = torch.empty(off, device=device, dtype=torch.float32)
        self.table = torch.tensor(rows, dtype=torch.int64).to(device)
        self.nrows = len(rows)
        self.bytes = 4.0 * off

    def slab(self, key):
        off, n = self.slabs[key]
        return self.workspace[off:off + n]

    def reduce(self, grad_flat, row_lo=0, row_hi=None, scale=1.0):
        """Reduce the slabs of records [row_lo, row_hi) (default: all) into ``grad_flat``, times ``scale`` (the data-parallel exchange's
        image-count weight; 1 leaves the sums bit for bit)."""
        row_hi = self.nrows if row_hi is None else row_hi
        if not (0 <= row_lo < row_hi <= self.nrows):
            raise ValueError('WgradBatch.reduce: bad record range')
        nrec = row_hi - row_lo
        b0, b1 = self.row_blocks[row_lo], self.row_blocks[row_hi]
        br = _Bracket('wgrad_reduce_batched', f'{nrec} layers', 0.0, self.bytes * (b1 - b0) / max(self.total_blocks, 1)) if timing._timer is not None else None
        if nrec == self.nrows:
            rc = nat.lib().sqd_wgrad_reduce_batched(nat.ptr(self.table), self.nrows, self.total_blocks, nat.ptr(self.workspace),
                                                    nat.ptr(grad_flat), float(scale), nat.stream_handle(grad_flat.device))
        else:
            rc = nat.lib().sqd_wgrad_reduce_batched_range(nat.c_p(self.table.data_ptr() + row_lo * 9 * 8), nrec, b0, b1 - b0,
                                                          nat.ptr(self.workspace), nat.ptr(grad_flat), float(scale), nat.stream_handle(grad_flat.device))
        nat.check(rc, 'sqd_wgrad_reduce_batched')
        if br is not None:
            br.done()
