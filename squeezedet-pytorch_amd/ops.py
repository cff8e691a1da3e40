"""Thin, shape-checked Python wrappers over the C-ABI kernels (NHWC fp32 device tensors in,
NHWC fp32 device tensors out).  Every wrapper validates operand shapes against what the kernel
and its grid assume *before* launching -- an out-of-bounds access on the GPU can take the node down.

All launches go to the caller's current HIP stream; nothing synchronises.

Layout of the host side: ``tiles`` (which configuration runs a layer: measured table, heuristics, feasibility),
``plans`` (packed / transformed operand copies, slab workspaces), ``timing`` (optional HIP-event brackets) and this
module (marshalling: one function per C-ABI entry point).  Their public names are re-exported here, so callers keep
writing ``ops.choose_cfg`` / ``ops.ConvPlan`` / ``ops.KernelTimer``.
"""
from __future__ import annotations

import ctypes
import torch

from . import _native as nat
from . import timing
from .timing import KernelTimer, set_timer, _Bracket  # noqa: F401
from . import tiles
from .tiles import *  # noqa: F401,F403
from .tiles import (_tuning, _nearest_tuned, _CFG_DMA, _wino_wgrad_tc)  # noqa: F401
from . import plans
from .plans import (ConvPlan, repack_batched, dgrad_weight, FusedExpandPlan, WinoPlan, repack_wino_batched, FireWinoPlan,  # noqa: F401
                    FireBridgePlan, WgradBatch, wino_sk_schedule, wino_sk_host_schedule)


def _check_nhwc(t, name):
    if t.dim() != 4 or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f'{name} must be a contiguous fp32 CUDA tensor [B,H,W,C], got {tuple(t.shape)} {t.dtype} {t.device}')


class DropState:
    """Device state {seed, step} of the counter-based dropout in front of ConvDet (csrc/sqd_common.h; reference nn.Dropout,
    src/model/squeezedet.py:71-72,81-82) + its rate.  ``keep16`` = round((1 - p) * 65536), ``scale`` = 1 / (1 - p)."""
    __slots__ = ('state', 'p', 'keep16', 'scale')

    def __init__(self, p, seed, device, step=0):
        if not 0.0 <= p < 1.0:
            raise ValueError(f'dropout probability must be in [0, 1), got {p}')
        self.p = float(p)
        self.keep16 = int(round((1.0 - self.p) * 65536))
        self.scale = 1.0 / (1.0 - self.p)
        self.state = torch.tensor([int(seed) & 0x7fffffffffffffff, int(step)], dtype=torch.int64, device=device)

    def get(self):
        """(seed, step) as Python ints (one device-to-host copy: checkpoints, tests)."""
        s = self.state.cpu()
        return int(s[0]), int(s[1])

    def set(self, seed, step):
        self.state.copy_(torch.tensor([int(seed) & 0x7fffffffffffffff, int(step)], dtype=torch.int64))


def dropout_mask(drop, shape):
    """The scaled keep mask of ``drop``'s CURRENT (seed, step) for an NHWC tensor of ``shape`` (numel % 4 == 0), from the stand-alone
    kernel: element e = flat index.  Does not advance the step."""
    n = 1
    for d in shape:
        n *= int(d)
    if n % 4:
        raise ValueError('dropout_mask: number of elements must be a multiple of 4')
    m = torch.empty(shape, device=drop.state.device, dtype=torch.float32)
    br = _Bracket('dropout_mask', f'{n} elements', 0.0, 4.0 * n) if timing._timer is not None else None
    nat.check(nat.lib().sqd_dropout_mask_fwd(nat.ptr(drop.state), drop.keep16, float(drop.scale), nat.ptr(m), n // 4,
                                              nat.stream_handle(m.device)), 'sqd_dropout_mask_fwd')
    if br is not None:
        br.done()
    return m


def dropout_advance(drop):
    """step += 1 on the device (one forward consumed its mask) when no kernel of the forward carried the advance."""
    nat.check(nat.lib().sqd_dropout_advance(nat.ptr(drop.state), nat.stream_handle(drop.state.device)), 'sqd_dropout_advance')


def dropout_mask_reference(seed, step, keep16, scale, n):
    """Host restatement (numpy, uint32 arithmetic) of csrc/sqd_common.h's sqd_drop_mul4 for elements [0, n): what every kernel that
    applies the dropout -- fused epilogue or stand-alone -- must reproduce bit for bit."""
    import numpy as np

    def mix(x):
        x = x.astype(np.uint32)
        x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d); x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b); x ^= x >> np.uint32(16)
        return x
    u = lambda v: np.array([v & 0xffffffff], dtype=np.uint32)
    with np.errstate(over='ignore'):
        s0, s1, t0, t1 = u(seed), u(seed >> 32), u(step), u(step >> 32)
        k0 = mix(s0 ^ mix(t0 + np.uint32(0x9e3779b9)))
        k1 = mix(s1 ^ mix(t0 ^ np.uint32(0x85ebca6b)) ^ (t1 * np.uint32(0xc2b2ae35)))
        e4 = np.arange((n + 3) // 4, dtype=np.uint64)
        lo = (e4 & np.uint64(0xffffffff)).astype(np.uint32) ^ ((e4 >> np.uint64(32)).astype(np.uint32) * np.uint32(0x9e3779b9))
        h1 = mix(lo ^ k0)
        h2 = mix(h1 ^ k1)
    f = np.stack([h1 & np.uint32(0xffff), h1 >> np.uint32(16), h2 & np.uint32(0xffff), h2 >> np.uint32(16)], 1).reshape(-1)[:n]
    return np.where(f < np.uint32(keep16), np.float32(scale), np.float32(0.0)).astype(np.float32)


def conv_drop_cfg(C, N, npix):
    """A weight-stationary 1x1 configuration (the family whose epilogue carries the fused dropout) for a C -> N layer: the table's
    choice if it is one, else the widest slice whose weights fit the LDS; None if there is none."""
    hit = choose_cfg(1, C, N, npix)
    if _CFG_DMA.get(hit % 1000, 0) >= 3 and conv_cfg_ok(hit, C):
        return hit
    best = None
    for cid, (taps, kc, px, bn) in cfg_table().items():
        if taps == 1 and _CFG_DMA.get(cid, 0) >= 3 and conv_cfg_ok(cid, C):
            pad = -(-N // bn) * bn / N
            key = (pad, -bn, -(8 if _CFG_DMA[cid] == 4 else 4))
            if best is None or key < best[0]:
                best = (key, cid)
    return None if best is None else best[1]


def conv(x, x_coff, plan, y, y_coff, relu=False, accumulate=False, xmask=None, xmask_coff=0, ymask=None, ymask_coff=0,
         ymul=None, ymul_coff=0, drop=None):
    """y[..., y_coff:y_coff+N] (=|+=) conv(x[..., x_coff:x_coff+C] [* (xmask>0)]) (+bias) (ReLU).  ``drop`` (a DropState; forward
    only, ``plan`` on a weight-stationary 1x1 configuration): counter-based dropout of the output in the epilogue."""
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
    if timing._timer is not None:
        npix = B * H * W
        br = _Bracket(cfg_kernel_name(plan.cfg_id),
                      f'{plan.taps}tap C{plan.C} N{plan.N} {H}x{W}', 2.0 * npix * plan.N * plan.C * plan.taps,
                      4.0 * (npix * (plan.C + plan.N) + plan.N * plan.C * plan.taps))
    if drop is not None:
        if accumulate or xmask is not None or ymask is not None or ymul is not None:
            raise ValueError('conv: the fused dropout is a forward-only epilogue')
        rc = nat.lib().sqd_conv_drop_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias), nat.ptr(y), B, H, W, plan.C, xp, x_coff, plan.N,
                                         plan.Npad, yp, y_coff, int(relu), nat.ptr(drop.state), drop.keep16, float(drop.scale),
                                         plan.cfg_id, nat.stream_handle(x.device))
        nat.check(rc, 'sqd_conv_drop_fwd')
    else:
        rc = nat.lib().sqd_conv_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias), nat.ptr(y), nat.ptr(xmask),
                                    nat.ptr(ymask), nat.ptr(ymul), B, H, W, plan.C, xp, x_coff, plan.N, plan.Npad, yp, y_coff,
                                    int(relu), int(accumulate), mp, xmask_coff, ymp, ymask_coff, ylp, ymul_coff,
                                    plan.cfg_id, nat.stream_handle(x.device))
        nat.check(rc, 'sqd_conv_fwd')
    if br is not None:
        br.done()
    return y


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
                  2.0 * B * H * W * fplan.E * fplan.C * 10, 4.0 * B * H * W * (fplan.C + 2 * fplan.E)) if timing._timer is not None else None
    rc = nat.lib().sqd_fire_expand_fwd(nat.ptr(x), nat.ptr(p.w), nat.ptr(p.bias), nat.ptr(y), B, H, W, fplan.C, xp, x_coff, fplan.E,
                                       p.Npad, y.shape[3], y_coff, fplan.cfg_id, nat.stream_handle(x.device))
    nat.check(rc, 'sqd_fire_expand_fwd')
    if br is not None:
        br.done()
    return y


def conv_wino(x, x_coff, plan, y, y_coff, relu=False, accumulate=False, ymask=None, ymul=None, yscale=1.0, drop=None, drop_advance=None):
    """y[..., y_coff:y_coff+N] (=|+=) conv3x3(x[..., x_coff:x_coff+C]) (+bias) (* ymul) (* yscale) (zero where ymask <= 0) (ReLU),
    Winograd F(2x2,3x3) kernel.  ``ymask`` / ``ymul`` are read through y's own channel window (same shape as y).  ``yscale``
    (a constant factor, e.g. the dropout scale), ``drop`` (a DropState: counter-based dropout of the output) and ``drop_advance`` (a
    DropState whose step this launch advances) need the balanced kernel (``plan.cfg_id`` = tiles.WINO_SK_CFG)."""
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
    if timing._timer is not None:
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
    if plan.cfg_id % 1000 == tiles.WINO_SK_CFG:
        sk = wino_sk_schedule(B * -(-H // 4) * -(-W // 16), plan.N, plan.C, x.device)
        sk_ws, sk_cnt = sk.workspace()               # (per launch stream: concurrent lanes must not share slabs / tickets)
        rc = nat.lib().sqd_conv_wino_sk_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias), nat.ptr(y), nat.ptr(ymask), nat.ptr(ymul),
                                            float(yscale), B, H, W, plan.C, xp, x_coff, plan.N, plan.Npad, yp, y_coff, int(relu),
                                            int(accumulate), nat.ptr(sk.seg_off), nat.ptr(sk.segs), sk.G, sk.nslabs, nat.ptr(sk_ws),
                                            nat.ptr(sk_cnt), nat.ptr(drop.state) if drop is not None else None,
                                            drop.keep16 if drop is not None else 0, float(drop.scale) if drop is not None else 0.0,
                                            nat.ptr(drop_advance.state) if drop_advance is not None else None, nat.stream_handle(x.device))
        nat.check(rc, 'sqd_conv_wino_sk_fwd')
    elif plan.cfg_id % 1000 == tiles.WINO_VS_CFG:
        if yscale != 1.0 or drop is not None or drop_advance is not None or accumulate or ymask is not None or ymul is not None:
            raise ValueError('conv_wino: the V-shared kernel (cfg tiles.WINO_VS_CFG) has the plain bias / ReLU epilogue only')
        if plan.N > 80 or plan.Npad != 80:
            raise ValueError('conv_wino: the V-shared kernel (cfg tiles.WINO_VS_CFG) runs N <= 80 packed 80 wide')
        rc = nat.lib().sqd_conv_wino_vs_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias), nat.ptr(y), B, H, W, plan.C, xp, x_coff,
                                            plan.N, plan.Npad, yp, y_coff, int(relu), nat.stream_handle(x.device))
        nat.check(rc, 'sqd_conv_wino_vs_fwd')
    else:
        if yscale != 1.0 or drop is not None or drop_advance is not None:
            raise ValueError('conv_wino: yscale / drop / drop_advance need the balanced kernel (cfg tiles.WINO_SK_CFG)')
        rc = nat.lib().sqd_conv_wino_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias), nat.ptr(y), nat.ptr(ymask), nat.ptr(ymul),
                                         B, H, W, plan.C, xp, x_coff, plan.N, plan.Npad, yp, y_coff, int(relu), int(accumulate),
                                         plan.cfg_id, nat.stream_handle(x.device))
        nat.check(rc, 'sqd_conv_wino_fwd')
    if br is not None:
        br.done()
    return y


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
    if timing._timer is not None:
        npix = B * H * W
        br = _Bracket(fire_wino_kernel_name(plan.cfg_id), f'fire C{plan.C} E{plan.N1}+{plan.N3} {H}x{W}',
                      2.0 * npix * plan.C * (4 * plan.N3 + plan.N1), 4.0 * (npix * (plan.C + plan.N1 + plan.N3) + plan.C * (16 * plan.N3 + plan.N1)))
    rc = nat.lib().sqd_fire_wino_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.b3), nat.ptr(plan.b1), nat.ptr(y), B, H, W, plan.C, xp, x_coff,
                                     plan.N3, y_coff3, plan.N1, y_coff1, plan.Npad, yp, plan.cfg_id, nat.stream_handle(x.device))
    nat.check(rc, 'sqd_fire_wino_fwd')
    if br is not None:
        br.done()
    return y


def fire_bridge(x, x_coff, plan, y, y_coff, save=None, save_coff1=0, save_coff3=None):
    """y[..., y_coff:+Nsq] = relu(squeeze'(cat(relu(expand1x1(x)), relu(expand3x3(x))))) in ONE launch.  Inference: the concatenated
    expand output is never written.  Training (``save``: [B,H,W,>=N1+N3]): it is stored too -- expand1x1 at ``save_coff1``, expand3x3 at
    ``save_coff3`` (default: right behind the expand1x1 window) -- for the backward; only the small-C form (plan.cfg_id 12) has it."""
    _check_nhwc(x, 'x'); _check_nhwc(y, 'y')
    B, H, W, xp = x.shape
    if tuple(y.shape[:3]) != (B, H, W) or plan.pooled:
        raise ValueError('fire_bridge: x and y disagree on B,H,W (or the plan is a pooled one)')
    yp = y.shape[3]
    if x_coff < 0 or x_coff + plan.C > xp or y_coff < 0 or y_coff + plan.Nsq > yp:
        raise ValueError('fire_bridge: channel window out of range')
    if save is not None:
        _check_nhwc(save, 'save')
        if save_coff3 is None:
            save_coff3 = save_coff1 + plan.N1
        sp = save.shape[3]
        if tuple(save.shape[:3]) != (B, H, W) or min(save_coff1, save_coff3) < 0 or save_coff1 + plan.N1 > sp or save_coff3 + plan.N3 > sp:
            raise ValueError('fire_bridge: save must be [B,H,W,.] with both expand windows inside')
        if not (save_coff1 + plan.N1 <= save_coff3 or save_coff3 + plan.N3 <= save_coff1):
            raise ValueError('fire_bridge: the two windows of save overlap')
        if plan.cfg_id % 1000 != 12:
            raise ValueError('fire_bridge: only the small-C form (cfg 12) stores the expand output')
    br = None
    if timing._timer is not None:
        npix = B * H * W
        br = _Bracket('fire_bridge_save' if save is not None else 'fire_bridge', f'fire C{plan.C} E{plan.N1}+{plan.N3} -> S{plan.Nsq} {H}x{W}',
                      2.0 * npix * (plan.C * (4 * plan.N3 + plan.N1) + (plan.N1 + plan.N3) * plan.Nsq),
                      4.0 * (npix * (plan.C + plan.Nsq + (plan.N1 + plan.N3 if save is not None else 0))
                             + plan.C * (16 * plan.N3 + plan.N1) + (plan.N1 + plan.N3) * plan.Nsq))
    if save is not None:
        rc = nat.lib().sqd_fire_bridge_save_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias_tab), nat.ptr(plan.sq_ops), nat.ptr(plan.sq_bias),
                                                nat.ptr(y), nat.ptr(save), B, H, W, plan.C, xp, x_coff, plan.N3, plan.N1, plan.Npad, plan.Nsq,
                                                yp, y_coff, sp, save_coff3, save_coff1, nat.stream_handle(x.device))
        nat.check(rc, 'sqd_fire_bridge_save_fwd')
    else:
        rc = nat.lib().sqd_fire_bridge_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias_tab), nat.ptr(plan.sq_ops), nat.ptr(plan.sq_bias),
                                           nat.ptr(y), B, H, W, plan.C, xp, x_coff, plan.N3, plan.N1, plan.Npad, plan.Nsq, yp, y_coff,
                                           plan.cfg_id, nat.stream_handle(x.device))
        nat.check(rc, 'sqd_fire_bridge_fwd')
    if br is not None:
        br.done()
    return y


def fire_pool_bridge(x, x_coff, plan, y, y_coff, nseg=4, save=None, codes=None, save_coff1=0, save_coff3=None):
    """y[..., y_coff:+Nsq] = relu(squeeze'(maxpool3x3s2_ceil(cat(relu(expand1x1(x)), relu(expand3x3(x)))))) in ONE launch; y is
    [B, Hp, Wp, .] with (Hp, Wp) = pool_out_size(H, W).  ``plan``: FireBridgePlan(..., pooled=True).  Inference: neither the expand
    output nor the pooled tensor is written.  Training (``save`` fp32 [B,Hp,Wp,>=N1+N3] and ``codes`` uint8 of the same shape): the
    pooled tensor and the pool's arg-max / ReLU codes are stored too (everything the backward reads of this stage)."""
    _check_nhwc(x, 'x'); _check_nhwc(y, 'y')
    B, H, W, xp = x.shape
    Hp, Wp = pool_out_size(H, W)
    if tuple(y.shape[:3]) != (B, Hp, Wp) or not plan.pooled:
        raise ValueError('fire_pool_bridge: y must be [B, Hp, Wp, .] of the pooled map and the plan a pooled one')
    yp = y.shape[3]
    if x_coff < 0 or x_coff + plan.C > xp or y_coff < 0 or y_coff + plan.Nsq > yp:
        raise ValueError('fire_pool_bridge: channel window out of range')
    if (save is None) != (codes is None):
        raise ValueError('fire_pool_bridge: save and codes go together')
    if save is not None:
        _check_nhwc(save, 'save')
        if save_coff3 is None:
            save_coff3 = save_coff1 + plan.N1
        sp = save.shape[3]
        if tuple(save.shape[:3]) != (B, Hp, Wp) or min(save_coff1, save_coff3) < 0 or save_coff1 + plan.N1 > sp or save_coff3 + plan.N3 > sp:
            raise ValueError('fire_pool_bridge: save must be [B,Hp,Wp,.] with both expand windows inside')
        if not (save_coff1 + plan.N1 <= save_coff3 or save_coff3 + plan.N3 <= save_coff1):
            raise ValueError('fire_pool_bridge: the two windows of save overlap')
        if tuple(codes.shape) != tuple(save.shape) or codes.dtype != torch.uint8 or not codes.is_contiguous() or codes.device != save.device:
            raise ValueError('fire_pool_bridge: codes must be a contiguous uint8 tensor of save\'s shape')
    br = None
    if timing._timer is not None:
        npix = B * H * W
        br = _Bracket('fire_pool_bridge_save' if save is not None else 'fire_pool_bridge',
                      f'fire C{plan.C} E{plan.N1}+{plan.N3} -> pool -> S{plan.Nsq} {H}x{W}',
                      2.0 * (npix * plan.C * (4 * plan.N3 + plan.N1) + B * Hp * Wp * (plan.N1 + plan.N3) * plan.Nsq),
                      4.0 * (npix * plan.C + B * Hp * Wp * (plan.Nsq + (1.25 * (plan.N1 + plan.N3) if save is not None else 0))
                             + plan.C * (16 * plan.N3 + plan.N1) + (plan.N1 + plan.N3) * plan.Nsq))
    if save is not None:
        rc = nat.lib().sqd_fire_pool_bridge_save_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias_tab), nat.ptr(plan.sq_ops), nat.ptr(plan.sq_bias),
                                                     nat.ptr(y), nat.ptr(save), nat.ptr(codes), B, H, W, plan.C, xp, x_coff, plan.N3, plan.N1,
                                                     plan.Npad, plan.Nsq, Hp, Wp, yp, y_coff, sp, save_coff3, save_coff1, int(nseg),
                                                     nat.stream_handle(x.device))
        nat.check(rc, 'sqd_fire_pool_bridge_save_fwd')
    else:
        rc = nat.lib().sqd_fire_pool_bridge_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias_tab), nat.ptr(plan.sq_ops), nat.ptr(plan.sq_bias),
                                                nat.ptr(y), B, H, W, plan.C, xp, x_coff, plan.N3, plan.N1, plan.Npad, plan.Nsq, Hp, Wp, yp, y_coff,
                                                int(nseg), nat.stream_handle(x.device))
        nat.check(rc, 'sqd_fire_pool_bridge_fwd')
    if br is not None:
        br.done()
    return y


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
                  4.0 * B * (H * W * C + Ho * Wo * plan.N)) if timing._timer is not None else None
    rc = nat.lib().sqd_pool_squeeze_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias), nat.ptr(y), B, H, W, C, xp, x_coff, plan.N, plan.Npad,
                                        y.shape[3], y_coff, nat.stream_handle(x.device))
    nat.check(rc, 'sqd_pool_squeeze_fwd')
    if br is not None:
        br.done()
    return y


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
                  4.0 * (B * 3 * H * W + B * Ho * Wo * N)) if timing._timer is not None else None
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
                  4.0 * (B * 3 * H * W + B * Hp * Wp * N)) if timing._timer is not None else None
    rc = nat.lib().sqd_stem_conv_relu_pool_fwd(nat.ptr(image), nat.ptr(w), nat.ptr(b), nat.ptr(out), nat.ptr(argmax), B, H, W, N, k,
                                               nat.stream_handle(image.device))
    nat.check(rc, 'sqd_stem_conv_relu_pool_fwd')
    if br is not None:
        br.done()
    return out


def stem_pool_squeeze_ok(image_shape, stem_weight_shape, squeeze_width):
    """True where ops.stem_pool_squeeze has a kernel: the 3x3 / 64-channel stem, a 16-channel squeeze, image width a multiple of 4."""
    N, ci, k, k2 = stem_weight_shape
    return (k, N, squeeze_width) == (3, 64, 16) and image_shape[3] % 4 == 0 and image_shape[2] >= 5 and image_shape[3] >= 5


def stem_pool_squeeze(image, weight, bias, sq_weight, sq_bias, argmax=None):
    """conv(3->64,3,s2)+ReLU+MaxPool(3,2,ceil) AND the first Fire's squeeze (1x1, 64->16, +ReLU) in one launch
    (src/model/squeezedet.py:34-37, :17-18).  Inference (``argmax`` None): NCHW image -> NHWC squeeze output [B,Hp,Wp,16]; the pooled
    tensor is never written.  Training (``argmax``: uint8 [B,Hp,Wp,64]): -> (squeeze output, pooled tensor); the pooled tensor and
    its codes are stored for the backward as by ``stem_pool(argmax=)``."""
    if image.dim() != 4 or image.shape[1] != 3 or image.dtype != torch.float32 or not image.is_cuda:
        raise ValueError(f'stem_pool_squeeze: image must be fp32 CUDA NCHW with 3 channels, got {tuple(image.shape)}')
    image = image.contiguous()
    N, ci, k, k2 = weight.shape
    nsq = sq_weight.shape[0]
    if ci != 3 or k != k2 or tuple(sq_weight.shape[1:]) != (N, 1, 1) or not stem_pool_squeeze_ok(image.shape, weight.shape, nsq):
        raise ValueError(f'stem_pool_squeeze: unsupported geometry {tuple(image.shape)} / {tuple(weight.shape)} / {tuple(sq_weight.shape)}')
    B, _, H, W = image.shape
    Ho, Wo = stem_out_size(H, W, k)
    Hp, Wp = pool_out_size(Ho, Wo)
    if argmax is not None and (tuple(argmax.shape) != (B, Hp, Wp, N) or argmax.dtype != torch.uint8 or not argmax.is_contiguous()):
        raise ValueError('stem_pool_squeeze: bad argmax tensor')
    out = torch.empty(B, Hp, Wp, nsq, device=image.device, dtype=torch.float32)
    w = weight.detach().contiguous()
    b = None if bias is None else bias.detach().contiguous()
    ws = sq_weight.detach().contiguous()
    bs = None if sq_bias is None else sq_bias.detach().contiguous()
    br = _Bracket(f'stem_pool_sq{"_train" if argmax is not None else ""}<{k}>', f'stem+pool+squeeze {H}x{W} S{nsq}',
                  2.0 * B * (Ho * Wo * N * 3 * k * k + Hp * Wp * N * nsq),
                  4.0 * (B * 3 * H * W + B * Hp * Wp * nsq + (B * Hp * Wp * N * 1.25 if argmax is not None else 0))) if timing._timer is not None else None
    if argmax is not None:
        pooled = torch.empty(B, Hp, Wp, N, device=image.device, dtype=torch.float32)
        rc = nat.lib().sqd_stem_pool_squeeze_train_fwd(nat.ptr(image), nat.ptr(w), nat.ptr(b), nat.ptr(ws), nat.ptr(bs), nat.ptr(pooled),
                                                       nat.ptr(argmax), nat.ptr(out), B, H, W, N, k, nsq, nat.stream_handle(image.device))
        nat.check(rc, 'sqd_stem_pool_squeeze_train_fwd')
        if br is not None:
            br.done()
        return out, pooled
    rc = nat.lib().sqd_stem_pool_squeeze_fwd(nat.ptr(image), nat.ptr(w), nat.ptr(b), nat.ptr(ws), nat.ptr(bs), nat.ptr(out), B, H, W, N, k,
                                             nsq, nat.stream_handle(image.device))
    nat.check(rc, 'sqd_stem_pool_squeeze_fwd')
    if br is not None:
        br.done()
    return out


def maxpool(x, out=None, argmax=None, relu_codes=False):
    """MaxPool2d(3, 2, ceil_mode=True) on NHWC; ``argmax`` (uint8, same shape as out) is filled if given.  ``relu_codes=True``
    (x is a ReLU output, a backward follows): the codes also carry x's ReLU mask (15 = pooled value not > 0), so
    ``maxpool_bwd`` runs without ``relu_src``."""
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
    br = _Bracket('maxpool_fwd', f'pool C{C} {H}x{W}', 0.0, 4.0 * B * C * (H * W + Ho * Wo)) if timing._timer is not None else None
    if relu_codes:
        if argmax is None:
            raise ValueError('maxpool: relu_codes needs an argmax tensor')
        rc = nat.lib().sqd_maxpool3x3s2_ceil_fwd_relu(nat.ptr(x), nat.ptr(out), nat.ptr(argmax), B, H, W, C, nat.stream_handle(x.device))
    else:
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
    if not (dy.is_contiguous() and argmax.is_contiguous() and out.is_contiguous()):
        raise ValueError('maxpool_bwd: dy, argmax and out must be contiguous')
    br = _Bracket('maxpool_bwd', f'poolbwd C{C} {H}x{W}', 0.0, 4.0 * B * C * (H * W * (2 if relu_src is not None else 1) + Ho * Wo * 1.25)) if timing._timer is not None else None
    # the kernel's grid carries (image, row pair) in blockIdx.y (<= 65535): larger batches go in slices of whole images
    per = max(1, 65535 // ((H + 1) // 2))
    if (H + 1) // 2 > 65535:
        raise ValueError(f'maxpool_bwd: {H} input rows exceed the kernel\'s grid (at most {2 * 65535})')
    for b0 in range(0, B, per):
        b1 = min(B, b0 + per)
        rc = nat.lib().sqd_maxpool3x3s2_ceil_bwd(nat.ptr(dy[b0:b1]), nat.ptr(argmax[b0:b1]), nat.ptr(out[b0:b1]),
                                                 nat.ptr(relu_src[b0:b1]) if relu_src is not None else None, b1 - b0, H, W, C,
                                                 nat.stream_handle(dy.device))
        nat.check(rc, 'sqd_maxpool3x3s2_ceil_bwd')
    if br is not None:
        br.done()
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


def det_packed_layout(B, K):
    """Byte layout of the packed result buffer: [(offset, elements, dtype, shape)] for (count, class_ids, scores, boxes, anchor_idx)
    -- 16-byte aligned sections -- and the total size.  The host reads a copied buffer back through the same table (lanes.py)."""
    sizes = [(B, torch.int32), (B * K, torch.int64), (B * K, torch.float32), (B * K * 4, torch.float32), (B * K, torch.int32)]
    shapes = [(B,), (B, K), (B, K), (B, K, 4), (B, K)]
    secs, off = [], 0
    for (n, dt), shp in zip(sizes, shapes):
        off = -(-off // 16) * 16
        secs.append((off, n, dt, shp))
        off += n * torch.empty(0, dtype=dt).element_size()
    return secs, -(-off // 16) * 16


def det_buffers_packed(B, K, device, A=None):
    """The same five result tensors as views of ONE allocation (16-byte aligned sections), so that a whole batch's compact
    detections leave the GPU with a single device-to-host copy.  -> (bufs as ``_det_buffers``, flat uint8 tensor)."""
    secs, total = det_packed_layout(B, K)
    flat = torch.zeros(total, device=device, dtype=torch.uint8)
    bufs = tuple(flat[o:o + n * torch.empty(0, dtype=dt).element_size()].view(dt).view(shp) for o, n, dt, shp in secs)
    if A is not None:
        bufs = bufs + (_det_workspace(B, A, device),)
    return bufs, flat


def det_workspace_words(B, A):
    """int32 words of the detect workspace: B x ceil4(A) keys + B arrival counters (sqd_detect_shift_fwd)."""
    return B * (-(-A // 4) * 4) + B


def _det_workspace(B, A, device):
    """Workspace of the fused detect launch (``keys_ws``): the keys the eight scoring workgroups of an image hand to its last arriver
    + one arrival counter per image.  Zeroed once: every launch returns the counters to zero."""
    return torch.zeros(det_workspace_words(B, A), device=device, dtype=torch.int32)


def detect(pred, anchors, input_size, num_classes, keep_top_k=64, nms_thresh=0.4, score_thresh=0.3, scales=None, out=None, shifts=None):
    """Fused decode + top-k + class-wise NMS + threshold for a batch.
    Returns (count int32 [B], class_ids int64 [B,K], scores [B,K], boxes [B,K,4], anchor_idx int32 [B,K]).  ``scales`` [B,2] =
    (sy, sx): boxes are divided by them; ``shifts`` [B,2] = (dy, dx): added afterwards (the padding / crops terms of
    ``boxes_postprocess``, src/utils/boxes.py:149-155)."""
    if pred.dim() != 3 or pred.shape[2] != num_classes + 5 or pred.dtype != torch.float32 or not pred.is_cuda:
        raise ValueError(f'detect: bad pred {tuple(pred.shape)}')
    pred = pred.contiguous()
    B, A, _ = pred.shape
    if tuple(anchors.shape) != (A, 4) or anchors.dtype != torch.float32 or anchors.device != pred.device:
        raise ValueError('detect: anchors must be fp32 [A,4] on the same device')
    if scales is not None and (tuple(scales.shape) != (B, 2) or scales.dtype != torch.float32 or scales.device != pred.device):
        raise ValueError('detect: scales must be fp32 [B,2] (sy, sx)')
    if shifts is not None and (tuple(shifts.shape) != (B, 2) or shifts.dtype != torch.float32 or shifts.device != pred.device or not shifts.is_contiguous()):
        raise ValueError('detect: shifts must be contiguous fp32 [B,2] (dy, dx)')
    bufs = out if out is not None else _det_buffers(B, keep_top_k, pred.device, A)
    if len(bufs) == 5:
        bufs = tuple(bufs) + (_det_workspace(B, A, pred.device),)
    cnt, cls, sc, bx, idx, keys = bufs
    if keys.dtype != torch.int32 or keys.device != pred.device:
        raise ValueError('detect: workspace must be an int32 tensor on the same device')
    if keys.numel() < det_workspace_words(B, A) or not keys.is_contiguous():
        keys = None                                  # (a caller-made placeholder: one workgroup per image)
    br = _Bracket('detect', f'detect A{A}', 0.0, 4.0 * B * A * (num_classes + 5)) if timing._timer is not None else None
    rc = nat.lib().sqd_detect_shift_fwd(nat.ptr(pred), nat.ptr(anchors.contiguous()), nat.ptr(scales), nat.ptr(shifts), nat.ptr(keys),
                                        nat.ptr(cnt), nat.ptr(cls), nat.ptr(sc), nat.ptr(bx), nat.ptr(idx), B, A, num_classes,
                                        int(input_size[0]), int(input_size[1]), int(keep_top_k), float(nms_thresh), float(score_thresh),
                                        nat.stream_handle(pred.device))
    nat.check(rc, 'sqd_detect_shift_fwd')
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
                      4.0 * (B * H * W * (C + N) + 2 * S * stride)) if timing._timer is not None else None
        rc = nat.lib().sqd_conv_wgrad_wino(nat.ptr(dy), nat.ptr(x), nat.ptr(slab), nat.ptr(dw), nat.ptr(db), B, H, W, N, dyp, dy_coff,
                                           C, xp, x_coff, S, _wino_wgrad_tc(N, C), nat.stream_handle(dy.device))
        nat.check(rc, 'sqd_conv_wgrad_wino')
    else:
        br = _Bracket(f'conv_wgrad<{taps}>', f'wgrad {taps}tap C{C} N{N} {H}x{W}', 2.0 * B * H * W * N * C * taps,
                      4.0 * (B * H * W * (C + N) + 2 * S * stride)) if timing._timer is not None else None
        rc = nat.lib().sqd_conv_wgrad(nat.ptr(dy), nat.ptr(x), nat.ptr(slab), nat.ptr(dw), nat.ptr(db), B, H, W, N, dyp, dy_coff,
                                      C, xp, x_coff, taps, S, nat.stream_handle(dy.device))
        nat.check(rc, 'sqd_conv_wgrad')
    if br is not None:
        br.done()
    return None if deferred else (dw, db)


def conv_wgrad_wino_group(items, S, tc):
    """The Winograd weight-gradient slabs of several 3x3 layers of one backward stage in ONE launch (``tiles.wino_wgrad_groups``
    decides which).  ``items``: [(dy, dy_coff, N, x, x_coff, C, slab)] with dy / x NHWC on the same [B,H,W] grid; ``slab`` = the layer's
    ``WgradBatch`` view of S * (N*9*C + N) floats.  Every layer is cut into the same ``S`` splits; slabs only (the batched reduction
    follows)."""
    if not 1 <= len(items) <= WINO_WGRAD_GROUP_MAX:
        raise ValueError('grouped weight gradient: 1..%d layers' % WINO_WGRAD_GROUP_MAX)
    B, H, W = items[0][0].shape[:3]
    rows, flops, byts, tag = [], 0.0, 0.0, []
    for dy, dy_coff, N, x, x_coff, C, slab in items:
        _check_nhwc(dy, 'dy'); _check_nhwc(x, 'x')
        if tuple(dy.shape[:3]) != (B, H, W) or tuple(x.shape[:3]) != (B, H, W):
            raise ValueError('grouped weight gradient: the layers of a group share B,H,W')
        if dy_coff + N > dy.shape[3] or x_coff + C > x.shape[3] or N % 64 or C % 4 or _wino_wgrad_tc(N, C) != tc:
            raise ValueError('grouped weight gradient: channel window out of range / wrong tile form')
        if slab.numel() != S * (N * 9 * C + N) or not slab.is_contiguous() or slab.dtype != torch.float32:
            raise ValueError('grouped weight gradient: slab workspace does not match the layer')
        rows += [dy.data_ptr(), x.data_ptr(), slab.data_ptr(), N, dy.shape[3], dy_coff, C, x.shape[3], x_coff]
        flops += 2.0 * B * H * W * N * C * 4
        byts += 4.0 * (B * H * W * (C + N) + 2 * S * (N * 9 * C + N))
        tag.append(f'C{C} N{N}')
    table = (ctypes.c_longlong * len(rows))(*rows)
    br = _Bracket('conv_wgrad_wino_group', f'wgrad 9tap {" + ".join(tag)} {H}x{W}', flops, byts) if timing._timer is not None else None
    rc = nat.lib().sqd_conv_wgrad_wino_group(ctypes.cast(table, ctypes.c_void_p), len(items), B, H, W, int(S), int(tc),
                                             nat.stream_handle(items[0][0].device))
    nat.check(rc, 'sqd_conv_wgrad_wino_group')
    if br is not None:
        br.done()


def conv_wgrad_group(items, S):
    """The direct 1x1 weight-gradient slabs of several layers of one pixel grid in ONE launch (``tiles.wgrad1x1_groups`` decides which).
    ``items`` as ``conv_wgrad_wino_group``; ``slab`` = the layer's ``WgradBatch`` view of S * (N*C + N) floats."""
    if not 1 <= len(items) <= WINO_WGRAD_GROUP_MAX:
        raise ValueError('grouped weight gradient: 1..%d layers' % WINO_WGRAD_GROUP_MAX)
    B, H, W = items[0][0].shape[:3]
    rows, flops, byts, tag = [], 0.0, 0.0, []
    for dy, dy_coff, N, x, x_coff, C, slab in items:
        _check_nhwc(dy, 'dy'); _check_nhwc(x, 'x')
        if tuple(dy.shape[:3]) != (B, H, W) or tuple(x.shape[:3]) != (B, H, W):
            raise ValueError('grouped weight gradient: the layers of a group share B,H,W')
        if dy_coff + N > dy.shape[3] or x_coff + C > x.shape[3] or N % 4 or C % 4:
            raise ValueError('grouped weight gradient: channel window out of range')
        if slab.numel() != S * (N * C + N) or not slab.is_contiguous() or slab.dtype != torch.float32:
            raise ValueError('grouped weight gradient: slab workspace does not match the layer')
        rows += [dy.data_ptr(), x.data_ptr(), slab.data_ptr(), N, dy.shape[3], dy_coff, C, x.shape[3], x_coff]
        flops += 2.0 * B * H * W * N * C
        byts += 4.0 * (B * H * W * (C + N) + 2 * S * (N * C + N))
        tag.append(f'C{C} N{N}')
    table = (ctypes.c_longlong * len(rows))(*rows)
    br = _Bracket('conv_wgrad_group<1>', f'wgrad 1tap {" + ".join(tag)} {H}x{W}', flops, byts) if timing._timer is not None else None
    rc = nat.lib().sqd_conv_wgrad_group(ctypes.cast(table, ctypes.c_void_p), len(items), B, H, W, int(S), nat.stream_handle(items[0][0].device))
    nat.check(rc, 'sqd_conv_wgrad_group')
    if br is not None:
        br.done()


def squeeze_bwd_ok(N, C):
    """Whether ``squeeze_bwd`` can run a (C -> N) 1x1 layer: all N out-channels in one group."""
    return N % 4 == 0 and C % 4 == 0 and N <= 128


def squeeze_bwd(dy, x, weight, slab, dx, relu_mask, dy_coff=0, N=None):
    """A 1x1 layer's backward in ONE launch: the weight / bias gradient slabs (as ``conv_wgrad(..., slab=...)`` writes them; ``slab``
    from a ``WgradBatch`` entry flagged fused) and the data gradient dx = dy . W, zeroed where ``relu_mask`` and x <= 0.  dy [B,H,W,.]
    (already masked; channel window ``[dy_coff, dy_coff + N)``, default all of it), x / dx [B,H,W,C], weight: the layer's OIHW
    [N,C,1,1] parameter.  Fire.squeeze (relu_mask = the previous layer is a Fire) and, where N <= 128, Fire.expand1x1 (dy = the
    expand1x1 window of the Fire output's gradient, x = the squeeze output, relu_mask False)."""
    _check_nhwc(dy, 'dy'); _check_nhwc(x, 'x'); _check_nhwc(dx, 'dx')
    B, H, W, dyp = dy.shape
    N = dyp - dy_coff if N is None else int(N)
    if dy_coff < 0 or dy_coff % 4 or dy_coff + N > dyp:
        raise ValueError('squeeze_bwd: dy channel window out of range')
    C = x.shape[3]
    if tuple(x.shape[:3]) != (B, H, W) or tuple(dx.shape) != tuple(x.shape):
        raise ValueError('squeeze_bwd: dy, x and dx disagree on B,H,W / C')
    if tuple(weight.shape) != (N, C, 1, 1) or weight.dtype != torch.float32 or not weight.is_cuda or not weight.is_contiguous():
        raise ValueError(f'squeeze_bwd: weight must be a contiguous fp32 CUDA [N,C,1,1] tensor, got {tuple(weight.shape)}')
    if not squeeze_bwd_ok(N, C):
        raise ValueError(f'squeeze_bwd: unsupported layer C={C} N={N}')
    S, stride = wgrad_split(N, C, 1, B, H, W, fused_dgrad=True)
    if slab.numel() != S * stride or not slab.is_contiguous() or slab.dtype != torch.float32:
        raise ValueError('squeeze_bwd: slab workspace does not match this layer')
    npix = B * H * W
    br = _Bracket('squeeze_bwd', f'sqbwd C{C} N{N} {H}x{W}', 4.0 * npix * N * C,
                  4.0 * (npix * (2 * C + N) + N * C + 2 * S * stride)) if timing._timer is not None else None
    rc = nat.lib().sqd_squeeze_bwd(nat.ptr(dy), nat.ptr(x), nat.ptr(weight.detach()), nat.ptr(slab), nat.ptr(dx), B, H, W, N, dyp, dy_coff, C, C, 0,
                                   C, 0, int(bool(relu_mask)), S, nat.stream_handle(dy.device))
    nat.check(rc, 'sqd_squeeze_bwd')
    if br is not None:
        br.done()
    return dx


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
                  4.0 * (B * Ho * Wo * N + B * 3 * H * W)) if timing._timer is not None else None
    rc = nat.lib().sqd_stem_wgrad(nat.ptr(dy), nat.ptr(image), nat.ptr(slab), nat.ptr(dw), nat.ptr(db), B, H, W, N, ksize, S,
                                  nat.stream_handle(dy.device))
    nat.check(rc, 'sqd_stem_wgrad')
    if br is not None:
        br.done()
    return dw, db


def stem_wgrad_pooled(dpool, pooled, argmax, image, N, ksize, out=None):
    """Stem (dW, db) when the forward ran fused (stem_pool with argmax): ReLU + max-pool backward folded in.  ``pooled`` may be
    None: the forward's arg-max codes carry the ReLU mask (15 = pooled value 0)."""
    _check_nhwc(dpool, 'dpool')
    if pooled is not None:
        _check_nhwc(pooled, 'pooled')
    B, Hp, Wp, n = dpool.shape
    if n != N or (pooled is not None and tuple(pooled.shape) != (B, Hp, Wp, N)) or tuple(argmax.shape) != (B, Hp, Wp, N) \
            or argmax.dtype != torch.uint8 or not argmax.is_contiguous():
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
                  4.0 * (B * Hp * Wp * N * (2.25 if pooled is not None else 1.25) + B * 3 * H * W)) if timing._timer is not None else None
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
    br = _Bracket('loss_fwd', f'loss A{A}', 0.0, 4.0 * B * A * (2 * num_classes + 14)) if timing._timer is not None else None
    rc = nat.lib().sqd_loss_fwd(nat.ptr(pred), nat.ptr(gt), nat.ptr(anchors), nat.ptr(ws), nat.ptr(losses), nat.ptr(nobj), B, A,
                                num_classes, int(input_size[0]), int(input_size[1]), *[float(w) for w in weights],
                                nat.stream_handle(pred.device))
    nat.check(rc, 'sqd_loss_fwd')
    if br is not None:
        br.done()
    return losses, nobj


def loss_mean_fwd(pred, gt, anchors, input_size, num_classes, weights):
    """-> (losses [4,B], nobj [B], mean4 [4] = batch means of class / score / bbox / total): ``loss.mean()`` inside the loss launches."""
    B, A = _check_loss_args(pred, gt, anchors, num_classes)
    pred, gt, anchors = pred.contiguous(), gt.contiguous(), anchors.contiguous()
    ws = torch.empty(B * 16 * 5, device=pred.device, dtype=torch.float32)
    losses = torch.empty(4, B, device=pred.device, dtype=torch.float32)
    nobj = torch.empty(B, device=pred.device, dtype=torch.float32)
    mean4 = torch.empty(4, device=pred.device, dtype=torch.float32)
    br = _Bracket('loss_fwd', f'loss A{A}', 0.0, 4.0 * B * A * (2 * num_classes + 14)) if timing._timer is not None else None
    rc = nat.lib().sqd_loss_mean_fwd(nat.ptr(pred), nat.ptr(gt), nat.ptr(anchors), nat.ptr(ws), nat.ptr(losses), nat.ptr(nobj), nat.ptr(mean4),
                                     B, A, num_classes, int(input_size[0]), int(input_size[1]), *[float(w) for w in weights],
                                     nat.stream_handle(pred.device))
    nat.check(rc, 'sqd_loss_mean_fwd')
    if br is not None:
        br.done()
    return losses, nobj, mean4


def loss_mean_bwd(pred, gt, anchors, nobj, gmean, input_size, num_classes, weights):
    """gmean: device scalar (gradient arriving at mean(total)) -> dpred [B,A,C+5]."""
    B, A = _check_loss_args(pred, gt, anchors, num_classes)
    if gmean.numel() != 1 or gmean.dtype != torch.float32 or gmean.device != pred.device:
        raise ValueError('loss_mean_bwd: gmean must be one fp32 value on the same device')
    pred, gt, anchors = pred.contiguous(), gt.contiguous(), anchors.contiguous()
    dpred = torch.empty_like(pred)
    br = _Bracket('loss_bwd', f'lossbwd A{A}', 0.0, 4.0 * B * A * (3 * num_classes + 19)) if timing._timer is not None else None
    rc = nat.lib().sqd_loss_mean_bwd(nat.ptr(pred), nat.ptr(gt), nat.ptr(anchors), nat.ptr(nobj), nat.ptr(gmean.contiguous()), nat.ptr(dpred), B, A,
                                     num_classes, int(input_size[0]), int(input_size[1]), *[float(w) for w in weights],
                                     nat.stream_handle(pred.device))
    nat.check(rc, 'sqd_loss_mean_bwd')
    if br is not None:
        br.done()
    return dpred


def loss_bwd(pred, gt, anchors, nobj, coef, input_size, num_classes, weights):
    """coef [3,B] -> dpred [B,A,C+5]."""
    B, A = _check_loss_args(pred, gt, anchors, num_classes)
    if tuple(coef.shape) != (3, B) or tuple(nobj.shape) != (B,):
        raise ValueError('loss_bwd: coef must be [3,B], nobj [B]')
    pred, gt, anchors, coef = pred.contiguous(), gt.contiguous(), anchors.contiguous(), coef.contiguous().float()
    dpred = torch.empty_like(pred)
    br = _Bracket('loss_bwd', f'lossbwd A{A}', 0.0, 4.0 * B * A * (3 * num_classes + 19)) if timing._timer is not None else None
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
    br = _Bracket('encode_gt', f'gt A{A}', 0.0, 4.0 * B * A * (num_classes + 9)) if timing._timer is not None else None
    ws = torch.empty(max(total, 1) * 2, device=dev, dtype=torch.float64)       # 16 bytes per box: first-choice candidates
    rc = nat.lib().sqd_encode_gt_fwd(nat.ptr(boxes) if total else None, nat.ptr(class_ids) if total else nat.ptr(idx),
                                     nat.ptr(box_offsets), nat.ptr(anchors64), nat.ptr(gt) if dense else None, nat.ptr(idx),
                                     nat.ptr(deltas), nat.ptr(ws) if (total and parallel) else None, int(total), B, A,
                                     int(num_classes), nat.stream_handle(dev))
    nat.check(rc, 'sqd_encode_gt_fwd')
    if br is not None:
        br.done()
    return gt, idx[:total], deltas[:total]
