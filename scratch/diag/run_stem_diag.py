import ctypes, os, sys, torch
sys.path.insert(0, '/root/repo')
here = os.path.dirname(os.path.abspath(__file__))
from squeezedet_pytorch_amd import _native as nat
B, H, W, N = 20, 384, 1248, 64
x = torch.randn(B, 3, H, W, device='cuda'); w = torch.randn(N, 3, 3, 3, device='cuda') * 0.2; b = torch.randn(N, device='cuda') * 0.1
y = torch.empty(B, 96, 312, N, device='cuda')
for v in ['w4', 'w8']:
    lib = ctypes.CDLL(os.path.join(here, f'libstem_{v}.so'))
    f = lib.sqd_stem_conv_relu_pool_fwd; f.argtypes = nat._SIGNATURES['sqd_stem_conv_relu_pool_fwd']; f.restype = ctypes.c_int
    def run():
        rc = f(nat.ptr(x), nat.ptr(w), nat.ptr(b), nat.ptr(y), None, B, H, W, N, 3, nat.stream_handle(x.device)); assert rc == 0, rc
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    print(f'{v:14s} {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us', flush=True)
