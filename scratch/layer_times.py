"""Per-launch in-situ times of one inference step (eager, HIP events around every launch, median of N steps)."""
import sys; sys.path.insert(0, '.')
import torch
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import ops, synthetic
from squeezedet_pytorch_amd.detector import Detector
from squeezedet_pytorch_amd.model import SqueezeDet
arch = sys.argv[1] if len(sys.argv) > 1 else 'squeezedet'
B = 16 if arch == 'squeezedetplus' else 20
cfg = sqd.make_cfg(arch=arch, device='cuda')
m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict(arch, seed=1234)); det = Detector(m, cfg)
x = synthetic.make_images(B, cfg.input_size, seed=0).cuda()
bufs = ops._det_buffers(B, cfg.keep_top_k, x.device, cfg.num_anchors)
with torch.no_grad():
    for _ in range(30): det.detect_device(x, out=bufs)
    torch.cuda.synchronize()
    t = ops.KernelTimer(); ops.set_timer(t)
    for _ in range(7): det.detect_device(x, out=bufs)
    ops.set_timer(None); torch.cuda.synchronize()
seen = {}; order = []
for name, tag, fl, by, e0, e1 in t.records:
    k = (name, tag)
    if k not in seen: seen[k] = []; order.append(k)
    seen[k].append((e0.elapsed_time(e1) * 1e3, fl, by))
tot = 0
for k in order:
    v = sorted(s[0] for s in seen[k]); med = v[len(v) // 2]; fl, by = seen[k][0][1], seen[k][0][2]
    n = len(v) / 7.0
    tot += med * n
    print(f'{k[0]:28s} {k[1]:40s} x{n:.0f} {med:7.1f} us  {fl / med / 1e6:6.1f} TF/s {by / med / 1e3:7.0f} GB/s')
print(f'sum {tot:.1f} us')
