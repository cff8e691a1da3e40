"""Per-workgroup start / end times of conv_wino_kernel<2,4> (diagnostic build: scratch/diag/wino_stamp.sh): do the two workgroups that share
a CU finish together?  usage (GPU box, repo root): bash scratch/diag/wino_stamp.sh && python scratch/diag/run_wino_stamp.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from squeezedet_pytorch_amd import ops  # noqa: E402
lib = ctypes.CDLL(os.path.join(ROOT, 'scratch', 'diag', 'libwino_stamp.so'))
c_p, c_i = ctypes.c_void_p, ctypes.c_int
lib.sqd_conv_wino_fwd.argtypes = [c_p] * 6 + [c_i] * 13 + [c_p]
lib.sqd_conv_wino_fwd.restype = c_i
lib.sqd_wino_set_debug.argtypes = [c_p]
B = 20
for C, N, H, W, cfg in [(768, 72, 24, 78, 2), (96, 384, 24, 78, 2), (48, 192, 24, 78, 2), (384, 96, 24, 78, 2)]:
    w = torch.randn(N, C, 3, 3, device='cuda') * 0.05; b = torch.randn(N, device='cuda') * 0.1
    plan = ops.WinoPlan(w, b, cfg)
    x = torch.randn(B, H, W, C, device='cuda'); y = torch.empty(B, H, W, N, device='cuda')
    dbg = torch.zeros(4096 * 8, dtype=torch.int64, device='cuda')
    lib.sqd_wino_set_debug(dbg.data_ptr())
    st = torch.cuda.current_stream().cuda_stream
    for it in range(3):
        dbg.zero_()
        rc = lib.sqd_conv_wino_fwd(x.data_ptr(), plan.w.data_ptr(), plan.bias.data_ptr(), y.data_ptr(), None, None, B, H, W, C, C, 0, N, plan.Npad, N, 0, 1, 0, cfg, st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(-1, 8)
    d = d[d[:, 3] > 0]
    t0, t1 = d[:, 2], d[:, 3]
    T0 = t0.min(); span = (t1.max() - T0) * 0.01
    cu = (d[:, 1] & 0xf) * 4096 + ((d[:, 0] >> 13) & 7) * 256 + ((d[:, 0] >> 12) & 1) * 16 + ((d[:, 0] >> 8) & 0xf)
    dur = (t1 - t0) * 0.01
    end = (t1 - T0) * 0.01
    start = (t0 - T0) * 0.01
    per_cu = {}
    for c, e, s_, n in zip(cu, end, start, d[:, 4]):
        per_cu.setdefault(int(c), []).append((e, s_, int(n)))
    first = np.array([min(v)[0] for v in per_cu.values()]); last = np.array([max(v)[0] for v in per_cu.values()])
    nper = np.array([len(v) for v in per_cu.values()])
    print(f'C{C}->N{N} {H}x{W}: {len(d)} workgroups on {len(per_cu)} CUs ({nper.min()}..{nper.max()} per CU), launch span {span:.1f} us; workgroup duration mean {dur.mean():.1f} '
          f'min {dur.min():.1f} max {dur.max():.1f} us; start spread {start.max():.1f} us; tiles per workgroup {d[:, 4].min()}..{d[:, 4].max()}')
    print(f'    per CU: first workgroup ends at {first.mean():.1f} us (mean), last at {last.mean():.1f} us (mean), gap {np.mean(last - first):.1f} us = {100 * np.mean(last - first) / span:.0f} % of the span; '
          f'end-time percentiles 10/50/90/100: {np.percentile(end, 10):.1f} / {np.percentile(end, 50):.1f} / {np.percentile(end, 90):.1f} / {end.max():.1f}')
