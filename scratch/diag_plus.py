import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle, squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import synthetic
from squeezedet_pytorch_amd.model import SqueezeDetWithLoss
arch = sys.argv[1] if len(sys.argv) > 1 else 'squeezedetplus'
size = (64, 96)
cfg = sqd.make_cfg(arch=arch, input_size=size, dropout_prob=0.0)
m = SqueezeDetWithLoss(cfg); sd = synthetic.make_state_dict(arch, seed=1234); m.load_state_dict(sd); m = m.cuda().train()
x = synthetic.make_images(2, size, seed=3); gt = synthetic.make_gt(2, cfg.anchors, size, seed=2, min_boxes=2, max_boxes=3)
loss, _ = m({'image': x.cuda(), 'gt': gt.cuda()}); loss.mean().backward()
_, _, g32, _, _, _ = oracle.train_step_reference(sd, None, x, gt, cfg.anchors, size, arch=arch)
sd64 = {k: v.double() for k, v in sd.items()}
_, _, g64, _, _, _ = oracle.train_step_reference(sd64, None, x.double(), gt.double(), cfg.anchors.astype(np.float64), size, arch=arch)
print(f'{"param":40s} {"gpu_vs_f64 max":>14s} {"cpu32_vs_f64 max":>16s} {"gpu relL2":>10s} {"cpu relL2":>10s}')
for name, p in m.named_parameters():
    r = g64[name]; a = p.grad.cpu().double(); c = g32[name].double()
    sc = float(r.abs().max())
    print(f'{name:40s} {float((a-r).abs().max())/sc:14.2e} {float((c-r).abs().max())/sc:16.2e} {float((a-r).norm()/r.norm()):10.2e} {float((c-r).norm()/r.norm()):10.2e}')
