"""fire_bridge vs torch on a few shapes + timing at the headline shape.  usage: bridge_check.py"""
import sys; sys.path.insert(0, '.')
import torch, torch.nn.functional as F, time
from squeezedet_pytorch_amd import ops
torch.manual_seed(0)
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous()
def run(B, H, W, C, E1, E3, S, cfg):
    if not ops.fire_bridge_cfg_ok(cfg, C, E3, E1, S):
        print(f'(cfg {cfg} cannot run C{C} E{E1}+{E3} S{S})'); return True
    x = torch.randn(B, C, H, W)
    w1 = torch.randn(E1, C, 1, 1) / C ** 0.5; b1 = torch.randn(E1) * 0.1
    w3 = torch.randn(E3, C, 3, 3) / (9 * C) ** 0.5; b3 = torch.randn(E3) * 0.1
    ws = torch.randn(S, E1 + E3, 1, 1) / (E1 + E3) ** 0.5; bs = torch.randn(S) * 0.1
    mid = torch.cat([F.relu(F.conv2d(x, w1, b1)), F.relu(F.conv2d(x, w3, b3, padding=1))], 1)
    ref = nhwc(F.relu(F.conv2d(mid, ws, bs)))
    plan = ops.FireBridgePlan(w1.cuda(), b1.cuda(), w3.cuda(), b3.cuda(), ws.cuda(), bs.cuda(), cfg)
    y0 = torch.randn(B, H, W, S + 8); y = y0.clone().cuda()
    ops.fire_bridge(nhwc(x).cuda(), 0, plan, y, 4)
    out = y.cpu()
    err = (out[..., 4:4 + S] - ref).abs().max().item()
    keep = torch.equal(out[..., :4], y0[..., :4]) and torch.equal(out[..., 4 + S:], y0[..., 4 + S:])
    print(f'B{B} {H}x{W} C{C} E{E1}+{E3} S{S} cfg{cfg}: err {err:.2e} scale {ref.abs().max():.2f} untouched {keep}', flush=True)
    return err < 2e-5 * max(1.0, ref.abs().max().item()) + 1e-5 and keep
ok = True
for cfg in (12, 10, 6):
    ok &= run(1, 8, 32, 16, 64, 64, 16, cfg)
    ok &= run(2, 9, 37, 16, 64, 64, 16, cfg)
    ok &= run(2, 24, 78, 32, 128, 128, 32, cfg) if cfg == 6 else True
    ok &= run(1, 5, 17, 8, 32, 40, 12, cfg)
    ok &= run(2, 11, 23, 16, 96, 48, 16, cfg)
    ok &= run(3, 13, 50, 16, 144, 32, 24, cfg)
    ok &= run(2, 7, 40, 16, 48, 96, 32, cfg)
print('ALL OK' if ok else 'FAILED')
if not ok: sys.exit(1)
# timing at the headline shape: fire3 -> fire4 squeeze
B, H, W = 20, 96, 312
x = torch.randn(B, H, W, 16, device='cuda')
for (C, E, S, cfgs) in ((16, 64, 16, (12, 10, 6)),):
    w1 = torch.randn(E, C, 1, 1).cuda(); w3 = torch.randn(E, C, 3, 3).cuda(); ws = torch.randn(S, 2 * E, 1, 1).cuda()
    b = torch.zeros(E).cuda(); bs = torch.zeros(S).cuda()
    for cfg in cfgs:
        plan = ops.FireBridgePlan(w1, b, w3, b, ws, bs, cfg)
        y = torch.empty(B, H, W, S, device='cuda')
        for _ in range(3): ops.fire_bridge(x, 0, plan, y, 0)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.fire_bridge(x, 0, plan, y, 0)
        e1.record(); torch.cuda.synchronize()
        print(f'bridge C{C} E{E} S{S} cfg {cfg}: {e0.elapsed_time(e1) / 20 * 1000:.1f} us', flush=True)
