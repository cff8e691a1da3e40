"""conv_wino_vp (the experimental persistent V-shared kernel, scratch/diag/conv_wino_vp.hip) against conv_wino<2,4> (bit for bit) and fp32 conv2d, then timed at the C96 -> N384 shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch, torch.nn.functional as F
from squeezedet_pytorch_amd import ops, tiles, _native as nat
_lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libvpdiag.so'))      # scratch/diag/vp_diag.sh
_vp = _lib.sqd_vp_diag_0
_vp.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 11 + [ctypes.c_void_p]; _vp.restype = ctypes.c_int

class VpPlan:
    """Weights packed Npad = N wide for the experimental persistent kernel."""
    def __init__(self, w, b):
        N, C = w.shape[0], w.shape[1]
        self.N, self.C, self.Npad, self.bias = N, C, N, b.contiguous()
        self.w = torch.empty(C // 8, 16, N, 8, device=w.device)
        nat.check(nat.lib().sqd_pack_wino_weight(nat.ptr(w.contiguous()), nat.ptr(self.w), N, C, N, 0, nat.stream_handle(w.device)), 'pack')

def conv_vp(x, xo, plan, y, yo, relu):
    B, H, W, xp = x.shape
    rc = _vp(nat.ptr(x), nat.ptr(plan.w), nat.ptr(plan.bias), nat.ptr(y), B, H, W, plan.C, xp, xo, plan.N, plan.Npad, y.shape[3], yo, int(relu),
             nat.stream_handle(x.device))
    assert rc == 0, rc
def run(C, N, B, H, W, relu, pe=0):
    torch.manual_seed(C + N + H)
    x = torch.randn(B, H, W, C + pe, device='cuda')
    w = torch.randn(N, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(N, device='cuda') * 0.1
    xo = pe // 2 // 4 * 4
    p2 = ops.WinoPlan(w, b, 2); p18 = VpPlan(w, b)
    y2 = torch.full((B, H, W, N + 8), 7.0, device='cuda'); y18 = torch.full((B, H, W, N + 8), 7.0, device='cuda')
    ops.conv_wino(x, xo, p2, y2, 4, relu=relu)
    ok = True
    for rep in range(3):
        y18.fill_(7.0)
        conv_vp(x, xo, p18, y18, 4, relu)
        torch.cuda.synchronize()
        same = torch.equal(y2, y18)
        if not same:
            bad = (y2 != y18).nonzero()
            print('   mismatch rep', rep, int((y2 != y18).sum()), 'elements; first', bad[0].tolist(), 'b', sorted(set(bad[:, 0].tolist()))[:5], 'rows', sorted(set(bad[:, 1].tolist()))[:8], 'ch', sorted(set(bad[:, 3].tolist()))[:6])
        ok = ok and same
    ref = F.conv2d(x[..., xo:xo + C].permute(0, 3, 1, 2).cpu(), w.cpu(), b.cpu(), padding=1)
    ref = (ref.relu() if relu else ref).permute(0, 2, 3, 1)
    err = (y18[..., 4:4 + N].cpu() - ref).abs().max().item()
    print(f'C{C} N{N} B{B} {H}x{W} relu={relu} pe={pe}: bitwise==cfg2 {ok}  max err vs conv2d {err:.2e}', flush=True)
    return ok and err < 1e-4 * max(1.0, ref.abs().max().item())
ok = True
for case in [(96, 384, 2, 24, 78, True), (96, 192, 1, 24, 78, False), (104, 192, 3, 5, 17, True), (96, 384, 1, 3, 3, True), (128, 192, 7, 4, 16, False),
             (96, 384, 5, 9, 33, True), (200, 192, 1, 1, 1, False), (96, 576, 2, 6, 20, True)]:
    ok = run(*case) and ok
ok = run(96, 384, 2, 6, 20, True, pe=16) and ok
print('ALL OK' if ok else 'FAILED', flush=True)
if not ok: sys.exit(1)
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (C, N, B) in [(96, 384, 20), (128, 384, 16), (96, 384, 40), (96, 192, 20)]:
    torch.manual_seed(0)
    x = torch.randn(B, 24, 78, C, device='cuda').relu_()
    w = torch.randn(N, C, 3, 3, device='cuda') * 0.05; b = torch.randn(N, device='cuda')
    y = torch.empty(B, 24, 78, N, device='cuda')
    line = f'C{C}->N{N} bs={B}:'
    for cfg in (2, 0, tiles.WINO_SK_CFG):
        plan = ops.WinoPlan(w, b, cfg)
        t = timeit(lambda: ops.conv_wino(x, 0, plan, y, 0, relu=True))
        line += f'  cfg{cfg} {t:7.1f} us'
    vplan = VpPlan(w, b)
    line += f'  conv_wino_vp {timeit(lambda: conv_vp(x, 0, vplan, y, 0, True)):7.1f} us'
    print(line, flush=True)
