"""Isolated time of the uint8 -> fp32 input kernels on a resident KITTI-sized batch (20 x 375x1242 -> 384x1248)."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, '.')
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import _native as nat
B, H0, W0, H, W = 20, 375, 1242, 384, 1248
rs = np.random.RandomState(0)
src = torch.from_numpy(rs.randint(0, 256, (B * H0 * W0 * 3,), dtype=np.uint8)).cuda()
off = torch.arange(B, dtype=torch.int64).cuda() * (H0 * W0 * 3)
sizes = torch.tensor([[H0, W0]] * B, dtype=torch.int32).cuda()
out = torch.empty(B, 3, H, W, device='cuda'); sc = torch.empty(B, 2, device='cuda')
mean = (ctypes.c_float * 3)(93.877, 98.801, 95.923); std = (ctypes.c_float * 3)(78.782, 80.130, 81.200)
st = nat.stream_handle(out.device)
import os
LIB = nat.lib()
if os.environ.get('PRE_LIB'):            # A/B: the round-4 kernels built alone (scratch/libpre_r04.so)
    LIB = ctypes.CDLL(os.environ['PRE_LIB'])
    for f in ('sqd_preprocess_u8_fwd', 'sqd_preprocess_u8_padcrop_fwd'):
        getattr(LIB, f).argtypes = getattr(nat.lib(), f).argtypes; getattr(LIB, f).restype = ctypes.c_int
def resize(): nat.check(LIB.sqd_preprocess_u8_fwd(nat.ptr(src), nat.ptr(off), nat.ptr(sizes), nat.ptr(out), nat.ptr(sc), mean, std, B, H, W, st), 'pre')
def padcrop(): nat.check(LIB.sqd_preprocess_u8_padcrop_fwd(nat.ptr(src), nat.ptr(off), nat.ptr(sizes), nat.ptr(out), nat.ptr(sc), None, mean, std, B, H, W, st), 'pc')
for name, fn in (('resize', resize), ('padcrop', padcrop)):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    mb = (B * H0 * W0 * 3 + B * 3 * H * W * 4) / 1e6
    print(f'{name}: {us:.1f} us per batch of {B}  ({mb:.0f} MB algorithmic -> {mb / us * 1e3:.0f} GB/s)')
