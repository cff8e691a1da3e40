import sys; sys.path.insert(0, '.')
import torch
from squeezedet_pytorch_amd import ops
x = torch.randn(20, 3, 384, 1248, device='cuda'); w = torch.randn(64, 3, 3, 3, device='cuda') * 0.2; b = torch.randn(64, device='cuda') * 0.1
am = torch.empty(20, 96, 312, 64, dtype=torch.uint8, device='cuda')
for name, kw in (('inference', {}), ('argmax', {'argmax': am})):
    for _ in range(3): ops.stem_pool(x, w, b, **kw)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.stem_pool(x, w, b, **kw)
    e1.record(); torch.cuda.synchronize()
    print(name, f'{e0.elapsed_time(e1) / 20 * 1e3:.1f} us')
