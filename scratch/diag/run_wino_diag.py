"""Ablation timing of conv_wino_kernel on the bs=20 SqueezeDet layer shapes (GPU box): every libwino_<mask>.so built by
build_wino_diag.sh (mask bits: 1 no MFMA, 2 no input transform, 4 no DMA in the loop, 8 no stores)."""
import ctypes, glob, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
here = os.path.dirname(os.path.abspath(__file__))
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import ops, _native as nat
B = int(os.environ.get('BATCH', 20))
shapes = [(16, 64, 96, 312), (32, 128, 48, 156), (48, 192, 24, 78), (64, 256, 24, 78), (96, 384, 24, 78), (768, 72, 24, 78)]
masks = [(int(m) if m.isdigit() else m) for m in os.environ.get('MASKS', '0,1,2,4,8,3,6,12,14').split(',')]
names = {0: 'base', 1: 'noMFMA', 2: 'noXform', 4: 'noDMA', 8: 'noStore', 3: 'noMFMA+noXform', 6: 'noXform+noDMA', 12: 'noDMA+noStore', 14: 'MFMA only'}
res = {}
for m in masks:
    lib = ctypes.CDLL(os.path.join(here, f'libwino_{m}.so'))
    lib.sqd_conv_wino_fwd.argtypes = nat._SIGNATURES['sqd_conv_wino_fwd']; lib.sqd_conv_wino_fwd.restype = ctypes.c_int
    for (C, N, h, w) in shapes:
        wc = ops.choose_wino_cfg(C, N, B * h * w)
        if wc is None: wc = 2
        if os.environ.get('WCFG'): wc = int(os.environ['WCFG'])
        wt = torch.randn(N, C, 3, 3, device='cuda') * 0.05; bias = torch.randn(N, device='cuda')
        plan = ops.WinoPlan(wt, bias, wc)
        x = torch.randn(B, h, w, C, device='cuda'); y = torch.zeros(B, h, w, N, device='cuda')
        def run():
            rc = lib.sqd_conv_wino_fwd(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias), nat.ptr(y), None, None, B, h, w, C, C, 0, N, plan.Npad, N, 0, 1, 0, wc, nat.stream_handle(x.device))
            assert rc == 0, rc
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        res[(m, C, N)] = e0.elapsed_time(e1) / 20 * 1e3
print(f'{"variant":18s}' + ''.join(f' C{C}->N{N}'.rjust(13) for (C, N, h, w) in shapes))
for m in masks:
    print(f'{names.get(m, str(m)):18s}' + ''.join(f'{res[(m, C, N)]:10.1f} us' for (C, N, h, w) in shapes), flush=True)
gf = [2.0 * B * h * w * N * C * 4 / 1e6 for (C, N, h, w) in shapes]
print(f'{"base exec TF/s":18s}' + ''.join(f'{g / res[(0, C, N)]:10.1f}   ' for g, (C, N, h, w) in zip(gf, shapes)))
print(f'{"MFMA-only floor us":18s}' + ''.join(f'{g / 157.3:10.1f}   ' for g in gf))
