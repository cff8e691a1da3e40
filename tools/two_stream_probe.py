"""Experiment: do two inference steps in flight (two captured graphs with their own activations, replayed alternately on two streams)
beat back-to-back replays on one stream?  The serial tail of a step (detect: 20-160 workgroups; every kernel's last round) would then
overlap the head of the next.  usage: python tools/two_stream_probe.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import ops, synthetic
from squeezedet_pytorch_amd.detector import Detector
from squeezedet_pytorch_amd.model import SqueezeDet

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = 20
cfg = sqd.make_cfg(device='cuda')
sd = synthetic.make_state_dict('squeezedet', seed=1234)
dev = torch.device('cuda', 0)
x = synthetic.make_images(B, cfg.input_size, seed=0).to(dev)


def make(stream):
    model = SqueezeDet(cfg); model.load_state_dict(sd)
    det = Detector(model, cfg)
    out = ops._det_buffers(B, cfg.keep_top_k, dev, cfg.num_anchors)
    with torch.cuda.stream(stream):
        for _ in range(3):
            det.detect_device(x, out=out)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            det.detect_device(x, out=out)
    return g, out, det


s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
g1, o1, d1 = make(s1)
g2, o2, d2 = make(s2)
g3, o3, d3 = make(s3)
torch.cuda.synchronize()


def run_three(n):
    for i in range(n):
        s, g = ((s1, g1), (s2, g2), (s3, g3))[i % 3]
        with torch.cuda.stream(s):
            g.replay()
    torch.cuda.synchronize()


def run_one(n):
    with torch.cuda.stream(s1):
        for _ in range(n):
            g1.replay()
    torch.cuda.synchronize()


def run_two(n):
    for i in range(n):
        if i & 1:
            with torch.cuda.stream(s2):
                g2.replay()
        else:
            with torch.cuda.stream(s1):
                g1.replay()
    torch.cuda.synchronize()


for f in (run_one, run_two, run_three):
    f(21)
for rep in range(3):
    t0 = time.perf_counter(); run_one(K); t1 = time.perf_counter(); run_two(K); t2 = time.perf_counter(); run_three(K); t3 = time.perf_counter()
    print(f'one stream {1e3 * (t1 - t0) / K:.4f} ms/step   two streams {1e3 * (t2 - t1) / K:.4f}   three {1e3 * (t3 - t2) / K:.4f} ms/step', flush=True)
assert torch.equal(o1[0], o2[0]) and torch.equal(o1[4], o2[4])
