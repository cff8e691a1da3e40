import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import synthetic
from squeezedet_pytorch_amd.model import SqueezeDet
from squeezedet_pytorch_amd.detector import Detector
cfg = sqd.make_cfg()
m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict()); det = Detector(m, cfg)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
x = synthetic.make_images(B, cfg.input_size).cuda()
for _ in range(3): out = det.detect_device(x)
torch.cuda.synchronize()
t = time.time(); n = 10
for _ in range(n): out = det.detect_device(x)
torch.cuda.synchronize(); dt = (time.time() - t) / n
print(f'eager: {dt*1e3:.3f} ms/batch  {B/dt:.1f} img/s')
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): out = det.detect_device(x)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        out = det.detect_device(x)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t = time.time(); n = 20
for _ in range(n): g.replay()
torch.cuda.synchronize(); dt = (time.time() - t) / n
print(f'graph: {dt*1e3:.3f} ms/batch  {B/dt:.1f} img/s  counts {out[0].tolist()}')
