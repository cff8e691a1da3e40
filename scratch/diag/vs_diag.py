"""Time the ablation builds of conv_wino_vs (scratch/diag/libvsdiag.so, vs_diag.sh) at the ConvDet shape, bs=20."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from squeezedet_pytorch_amd import ops, tiles, _native as nat
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'libvsdiag.so'))
B, H, W, C, N = 20, 24, 78, 768, 72
x = torch.randn(B, H, W, C, device='cuda').relu_()
w = torch.randn(N, C, 3, 3, device='cuda') * 0.01; b = torch.randn(N, device='cuda')
plan = ops.WinoPlan(w, b, tiles.WINO_VS_CFG)
y = torch.empty(B, H, W, N, device='cuda')
names = {11: 'warm-up', 0: 'full', 100: 'full, no s_setprio for the duty wave', 1: 'no input transform (patch still fetched)', 2: 'no patch DMA in the loop', 3: 'no transform, no patch DMA', 4: 'no stage barrier',
         8: 'U operands not loaded in the loop', -11: 'no transform, no patch DMA, no U loads'}
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for m, nm in names.items():
    f = getattr(lib, f'sqd_vs_diag_{abs(m)}')
    f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 11 + [ctypes.c_void_p]; f.restype = ctypes.c_int
    def call():
        rc = f(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias), nat.ptr(y), B, H, W, C, C, 0, N, 80, N, 0, 0, nat.stream_handle(x.device))
        assert rc == 0
    print(f'mask {m:2d} {nm:32s} {timeit(call):7.1f} us', flush=True)
