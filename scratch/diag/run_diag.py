import ctypes, os, sys, torch
sys.path.insert(0, '/root/repo')
here = os.path.dirname(os.path.abspath(__file__))
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import ops, _native as nat
shapes = [(9, 16, 64, 96, 312, (42,)), (9, 32, 128, 48, 156, (42, 71)), (9, 96, 384, 24, 78, (39,)), (9, 64, 256, 24, 78, (42,)), (9, 768, 72, 24, 78, (41,))]
B = 20
for v in ['base', 'nostore', 'noflush', 'noflush_nodma']:
    lib = ctypes.CDLL(os.path.join(here, f'libdiag_{v}.so'))
    lib.sqd_conv_fwd.argtypes = nat._SIGNATURES['sqd_conv_fwd']; lib.sqd_conv_fwd.restype = ctypes.c_int
    for taps, C, N, h, w, cfgs in shapes:
        for cid in cfgs:
            wt = torch.randn(N, C, 3, 3, device='cuda') * 0.05; bias = torch.randn(N, device='cuda')
            plan = ops.ConvPlan(wt, bias, cid)
            x = torch.randn(B, h, w, C, device='cuda'); y = torch.zeros(B, h, w, N, device='cuda')
            def run():
                rc = lib.sqd_conv_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias), nat.ptr(y), None, None, None, B, h, w, C, C, 0, N, plan.Npad, N, 0, 1, 0, 0, 0, 0, 0, 0, 0, cid, nat.stream_handle(x.device))
                assert rc == 0, rc
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            gf = 2.0 * B * h * w * N * C * taps / 1e9
            print(f'{v:22s} {taps}:{C}:{N} cfg {cid}: {us:7.1f} us  {gf / (us * 1e-6) / 1e3:6.1f} TF/s', flush=True)
