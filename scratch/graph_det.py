import sys, os, time
sys.path.insert(0, '/root/repo')
import torch
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import ops, synthetic
from squeezedet_pytorch_amd.model import SqueezeDet
cfg = sqd.make_cfg(); m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict()); m = m.cuda().eval()
x = synthetic.make_images(20, cfg.input_size).cuda()
with torch.no_grad(): pred = m.base(x)
anc = torch.from_numpy(cfg.anchors).float().cuda()
bufs = ops._det_buffers(20, 64, 'cuda', cfg.num_anchors)
def det(): return ops.detect(pred, anc, cfg.input_size, 3, 64, 0.4, 0.3, out=bufs)
def dec(): return ops.decode(pred, anc, cfg.input_size, 3)
def capture(fn, reps=1):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
    torch.cuda.current_stream().wait_stream(s)
    return g
def timeit(g, n=100):
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
g1, g10 = capture(det, 1), capture(det, 10)
print('graph(detect x1): %.1f us/replay ; graph(detect x10): %.1f us/replay -> %.1f us per detect' % (timeit(g1), timeit(g10), (timeit(g10)) / 10))
for _ in range(5): det()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(200): det()
torch.cuda.synchronize(); print('eager loop: %.1f us per detect' % ((time.perf_counter() - t) / 200 * 1e6))
