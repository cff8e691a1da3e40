"""ConvDet forward (C768 -> N72, 24x78, bs=20) on every Winograd configuration that can run it, in isolation (HIP events over 20 launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from squeezedet_pytorch_amd import ops, tiles
B, H, W = 20, 24, 78
for C, N in [(768, 72), (96, 384), (48, 192), (384, 96)]:
    w = torch.randn(N, C, 3, 3, device='cuda') * 0.05; b = torch.randn(N, device='cuda') * 0.1
    x = torch.randn(B, H, W, C, device='cuda'); y = torch.empty(B, H, W, N, device='cuda')
    row = []
    for cfg in list(range(12)) + [1002, 1003, 2003, 16]:
        if not tiles.wino_cfg_ok(cfg % 1000 if cfg != 16 else 16, C) and cfg != 16:
            continue
        try:
            plan = ops.WinoPlan(w, b, cfg)
            for _ in range(3): ops.conv_wino(x, 0, plan, y, 0, relu=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.conv_wino(x, 0, plan, y, 0, relu=True)
            e1.record(); torch.cuda.synchronize()
            row.append(f'{cfg}:{e0.elapsed_time(e1) / 20 * 1e3:.1f}')
        except Exception as e:  # noqa: BLE001
            row.append(f'{cfg}:x')
    print(f'C{C}->N{N}: ' + '  '.join(row), flush=True)
