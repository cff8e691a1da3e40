import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import ops, synthetic
from squeezedet_pytorch_amd.trainer import make_train_step
mode = sys.argv[1] if len(sys.argv) > 1 else 'train'
cfg = sqd.make_cfg(device='cuda')
sd = synthetic.make_state_dict('squeezedet', seed=1234)
x = synthetic.make_images(20, cfg.input_size, seed=0).cuda()
if mode == 'train':
    step, _ = make_train_step(cfg, sd, x, 0, 1, None)
else:
    from squeezedet_pytorch_amd.model import SqueezeDet
    from squeezedet_pytorch_amd.detector import Detector
    m = SqueezeDet(cfg); m.load_state_dict(sd); det = Detector(m, cfg)
    step = lambda: det.detect_device(x)
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(2): step()
t = ops.KernelTimer(); ops.set_timer(t)
for _ in range(3): step()
ops.set_timer(None); torch.cuda.synchronize()
rows = []
for name, d in t.summary(nsteps=3).items():
    for tag, (n, ms) in d['tags'].items():
        rows.append((ms, int(round(n)), name, tag))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f'total bracketed {tot:.3f} ms/step')
filt = sys.argv[2] if len(sys.argv) > 2 else ''
for ms, n, name, tag in rows:
    if filt in name: print(f'{ms*1e3:8.1f} us x{n}  {name:26s} {tag}')
