"""Timeline shares of the Winograd weight-gradient kernel (diagnostic build with s_memtime stamps: scratch/diag/ww_stamp.sh).
usage (GPU box, repo root): bash scratch/diag/ww_stamp.sh && python scratch/diag/run_ww_stamp.py
Prints, per layer shape of the bs=20 training step, the mean share of a wave's kernel time spent in: waiting for the group's DMA + the
workgroup barrier, issuing the next group's DMA, the V transform (LDS reads + packed adds), the dM transforms + MFMAs, the epilogue."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from squeezedet_pytorch_amd import tiles  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, 'scratch', 'diag', 'libww_stamp.so'))
c_p, c_i = ctypes.c_void_p, ctypes.c_int
lib.sqd_conv_wgrad_wino.argtypes = [c_p] * 5 + [c_i] * 11 + [c_p]
lib.sqd_conv_wgrad_wino.restype = c_i
lib.sqd_ww_set_debug.argtypes = [c_p]
lib.sqd_ww_set_debug.restype = c_i

SHAPES = [(768, 72, 24, 78), (96, 384, 24, 78), (64, 256, 24, 78), (48, 192, 24, 78), (32, 128, 48, 156), (16, 64, 96, 312)]
B = 20
dev = torch.device('cuda')
for C, N, H, W in SHAPES:
    S, stride = tiles.wgrad_split(N, C, 9, B, H, W)
    tc = tiles._wino_wgrad_tc(N, C)
    dy = torch.randn(B, H, W, N, device=dev)
    x = torch.randn(B, H, W, C, device=dev)
    slab = torch.empty(S * stride, device=dev)
    blocks = (-(-C // (16 * tc))) * (1 if N % 64 else N // 64)
    nwg = blocks * S
    dbg = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
    assert lib.sqd_ww_set_debug(dbg.data_ptr()) == 0
    st = torch.cuda.current_stream().cuda_stream
    for it in range(3):
        rc = lib.sqd_conv_wgrad_wino(dy.data_ptr(), x.data_ptr(), slab.data_ptr(), None, None, B, H, W, N, N, 0, C, C, 0, S, tc, st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(5):
        lib.sqd_conv_wgrad_wino(dy.data_ptr(), x.data_ptr(), slab.data_ptr(), None, None, B, H, W, N, N, 0, C, C, 0, S, tc, st)
    e1.record()
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(nwg, 4, 8).astype(np.float64)
    tot = d[..., 5]
    sh = [d[..., k] / tot for k in range(5)]
    t0 = d[..., 7]
    span = (t0 + tot).max() - t0.min()
    ngr = d[..., 6].mean()
    print(f'C{C}->N{N} {H}x{W}  S={S} tc={tc} workgroups={nwg} groups/wg={ngr:.1f}  kernel {e0.elapsed_time(e1) / 5 * 1e3:.1f} us (stamped build), '
          f'wave time mean {tot.mean():.0f} / max {tot.max():.0f} ticks, launch span {span:.0f} ticks, start spread {t0.max() - t0.min():.0f}')
    names = ['wait+barrier', 'DMA issue', 'V transform', 'dM + MFMA', 'epilogue']
    print('    shares of a wave\'s time: ' + ', '.join(f'{n} {100 * s.mean():.1f} %' for n, s in zip(names, sh))
          + f', unaccounted (prologue) {100 * (1 - sum(s.mean() for s in sh)):.1f} %')
    per = [d[..., k].sum() / d[..., 6].sum() for k in range(4)]
    print('    ticks per group: ' + ', '.join(f'{n} {p:.0f}' for n, p in zip(names, per)) + f';  epilogue {d[..., 4].mean():.0f} ticks per wave')
