"""Experiment: one inference step over 20 images as TWO half-batches on two streams inside one captured graph (tails of one
half's persistent grids filled by the other half's workgroups?) vs the single-batch step.  usage: python scratch/split_batch.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import ops, synthetic
from squeezedet_pytorch_amd.detector import Detector
from squeezedet_pytorch_amd.model import SqueezeDet

cfg = sqd.make_cfg(device='cuda')
m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict('squeezedet', seed=1234)); det = Detector(m, cfg)
x = synthetic.make_images(20, cfg.input_size, seed=0).cuda()


def timed(run, n=100):
    for _ in range(10): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def capture(fn):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    return g


bufs = ops._det_buffers(20, cfg.keep_top_k, x.device, cfg.num_anchors)
g1 = capture(lambda: det.detect_device(x, out=bufs))
print(f'single batch of 20          : {timed(g1.replay):.4f} ms')
for split in ((10, 10), (12, 8), (14, 6), (16, 4)):
    xa, xb = x[:split[0]].contiguous(), x[split[0]:].contiguous()
    ba = ops._det_buffers(split[0], cfg.keep_top_k, x.device, cfg.num_anchors); bb = ops._det_buffers(split[1], cfg.keep_top_k, x.device, cfg.num_anchors)
    s2 = torch.cuda.Stream()

    def both():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(cur)
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            det.detect_device(xb, out=bb)
        det.detect_device(xa, out=ba)
        cur.wait_stream(s2)
    both(); torch.cuda.synchronize()
    g2 = capture(both)
    print(f'two streams, {split[0]:2d} + {split[1]:2d} images : {timed(g2.replay):.4f} ms')
    # sequential halves (no overlap) for reference
    def seq():
        det.detect_device(xa, out=ba); det.detect_device(xb, out=bb)
    g3 = capture(seq)
    print(f'  (same halves back to back : {timed(g3.replay):.4f} ms)')
