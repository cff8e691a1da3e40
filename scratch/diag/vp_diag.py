"""Time the ablation builds of conv_wino_vp (scratch/libvpdiag.so, vp_diag.sh) at C96 -> N384, bs=20."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from squeezedet_pytorch_amd import ops, tiles, _native as nat
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'libvpdiag.so'))
B, H, W, C, N = 20, 24, 78, 96, 384
x = torch.randn(B, H, W, C, device='cuda').relu_()
w = torch.randn(N, C, 3, 3, device='cuda') * 0.05; b = torch.randn(N, device='cuda')
class _P: pass
plan = _P(); plan.bias = b
plan.w = torch.empty(C // 8, 16, N, 8, device='cuda')
nat.check(nat.lib().sqd_pack_wino_weight(nat.ptr(w), nat.ptr(plan.w), N, C, N, 0, nat.stream_handle(w.device)), 'pack')
y = torch.empty(B, H, W, N, device='cuda')
names = {31: 'warm-up', 0: 'full', 1: 'no input transform', 2: 'no epilogue stores', 4: 'no epilogue (inverse transform + stores)', 8: 'no stage barrier',
         16: 'U operands not loaded in the loop', -31: 'none of them'}
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for m, nm in names.items():
    f = getattr(lib, f'sqd_vp_diag_{abs(m)}')
    f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 11 + [ctypes.c_void_p]; f.restype = ctypes.c_int
    def call():
        rc = f(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias), nat.ptr(y), B, H, W, C, C, 0, N, N, N, 0, 1, nat.stream_handle(x.device))
        assert rc == 0
    print(f'mask {m:3d} {nm:42s} {timeit(call):7.1f} us', flush=True)
