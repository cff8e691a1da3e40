"""fire_pool_bridge vs torch on a few shapes + timing at the headline shape."""
import sys; sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from squeezedet_pytorch_amd import ops
torch.manual_seed(0)
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous()
def run(B, H, W, C, E1, E3, S, nseg):
    x = torch.randn(B, C, H, W)
    w1 = torch.randn(E1, C, 1, 1) / C ** 0.5; b1 = torch.randn(E1) * 0.1
    w3 = torch.randn(E3, C, 3, 3) / (9 * C) ** 0.5; b3 = torch.randn(E3) * 0.1
    ws = torch.randn(S, E1 + E3, 1, 1) / (E1 + E3) ** 0.5; bs = torch.randn(S) * 0.1
    mid = torch.cat([F.relu(F.conv2d(x, w1, b1)), F.relu(F.conv2d(x, w3, b3, padding=1))], 1)
    ref = nhwc(F.relu(F.conv2d(F.max_pool2d(mid, 3, 2, ceil_mode=True), ws, bs)))
    plan = ops.FireBridgePlan(w1.cuda(), b1.cuda(), w3.cuda(), b3.cuda(), ws.cuda(), bs.cuda(), 12, pooled=True)
    Hp, Wp = ops.pool_out_size(H, W)
    assert tuple(ref.shape[1:3]) == (Hp, Wp), (ref.shape, Hp, Wp)
    y0 = torch.randn(B, Hp, Wp, S + 8); y = y0.clone().cuda()
    ops.fire_pool_bridge(nhwc(x).cuda(), 0, plan, y, 4, nseg=nseg)
    out = y.cpu()
    d = (out[..., 4:4 + S] - ref).abs()
    err = d.max().item()
    keep = torch.equal(out[..., :4], y0[..., :4]) and torch.equal(out[..., 4 + S:], y0[..., 4 + S:])
    ok = err < 2e-5 * max(1.0, ref.abs().max().item()) + 1e-5 and keep
    print(f'B{B} {H}x{W} C{C} E{E1}+{E3} S{S} nseg{nseg}: err {err:.2e} scale {ref.abs().max():.2f} untouched {keep} {"ok" if ok else "BAD"}', flush=True)
    if not ok:
        bad = (d.amax(dim=(0, 3)) > 1e-4)
        print('bad rows', bad.any(dim=1).nonzero().flatten().tolist()[:20], 'bad cols', bad.any(dim=0).nonzero().flatten().tolist()[:30])
    return ok
ok = True
for args in [(1, 8, 32, 16, 64, 64, 32, 1), (1, 8, 32, 16, 64, 64, 32, 2), (2, 9, 37, 16, 64, 64, 32, 1), (1, 5, 17, 8, 32, 40, 12, 1), (2, 12, 30, 16, 48, 64, 16, 3),
             (3, 13, 50, 16, 64, 32, 24, 2), (1, 24, 14, 8, 16, 16, 32, 6), (2, 96, 312, 16, 64, 64, 32, 4), (1, 3, 3, 8, 16, 8, 4, 1), (1, 31, 45, 16, 64, 64, 32, 5)]:
    ok &= run(*args)
print('ALL OK' if ok else 'FAILED')
if not ok: sys.exit(1)
B, H, W = 20, 96, 312
x = torch.randn(B, H, W, 16, device='cuda')
w1 = torch.randn(64, 16, 1, 1).cuda(); w3 = torch.randn(64, 16, 3, 3).cuda(); ws = torch.randn(32, 128, 1, 1).cuda()
b = torch.zeros(64).cuda(); bs = torch.zeros(32).cuda()
plan = ops.FireBridgePlan(w1, b, w3, b, ws, bs, 12, pooled=True)
y = torch.empty(B, 48, 156, 32, device='cuda')
for nseg in (2, 3, 4, 6, 8):
    for _ in range(3): ops.fire_pool_bridge(x, 0, plan, y, 0, nseg=nseg)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.fire_pool_bridge(x, 0, plan, y, 0, nseg=nseg)
    e1.record(); torch.cuda.synchronize()
    print(f'pool bridge nseg {nseg}: {e0.elapsed_time(e1) / 20 * 1000:.1f} us', flush=True)
