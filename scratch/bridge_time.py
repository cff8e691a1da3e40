import sys; sys.path.insert(0, '.')
import torch
from squeezedet_pytorch_amd import ops
B = 20
for (C, E, S, H, W) in ((32, 128, 32, 48, 156), (48, 192, 48, 24, 78)):
    x = torch.randn(B, H, W, C, device='cuda')
    w1 = torch.randn(E, C, 1, 1).cuda(); w3 = torch.randn(E, C, 3, 3).cuda(); ws = torch.randn(S, 2 * E, 1, 1).cuda()
    b = torch.zeros(E).cuda(); bs = torch.zeros(S).cuda()
    for cfg in (6, 1006, 10, 12):
        if not ops.fire_bridge_cfg_ok(cfg, C, E, E, S): continue
        plan = ops.FireBridgePlan(w1, b, w3, b, ws, bs, cfg)
        y = torch.empty(B, H, W, S, device='cuda')
        for _ in range(3): ops.fire_bridge(x, 0, plan, y, 0)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.fire_bridge(x, 0, plan, y, 0)
        e1.record(); torch.cuda.synchronize()
        print(f'bridge C{C} E{E} S{S} {H}x{W} cfg {cfg}: {e0.elapsed_time(e1) / 20 * 1000:.1f} us', flush=True)
