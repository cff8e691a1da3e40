import sys, os, time
sys.path.insert(0, '/root/repo')
import torch
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import ops, synthetic
from squeezedet_pytorch_amd.model import SqueezeDet
from squeezedet_pytorch_amd.detector import Detector
cfg = sqd.make_cfg(); m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict()); det = Detector(m, cfg)
x = synthetic.make_images(20, cfg.input_size).cuda()
bufs = ops._det_buffers(20, 64, 'cuda', cfg.num_anchors)
def full(): return det.detect_device(x, out=bufs)
def backbone():
    with torch.no_grad(): return det.model.base(x)
def capture(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s): fn()
    torch.cuda.current_stream().wait_stream(s)
    return g
gf, gb = capture(full), capture(backbone)
def timeit(g, n=50):
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for r in range(3):
    a, b = timeit(gf), timeit(gb)
    print(f'round {r}: full {a:.4f} ms  backbone-only {b:.4f} ms  detect = {1e3*(a-b):.1f} us')
