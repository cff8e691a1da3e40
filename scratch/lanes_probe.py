"""Where does a two-lane pipeline batch spend 1.85 ms?  Variants of the raw-image path of lanes.DetectStream (A/B by monkeypatching)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import synthetic, lanes as L
from squeezedet_pytorch_amd.detector import Detector
from squeezedet_pytorch_amd.model import SqueezeDet

cfg = sqd.make_cfg(); B = 20; cfg.batch_size = B
m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict()); det = Detector(m, cfg)
H0, W0 = 375, 1242
K = int(os.environ.get('K', 100))

def run(lanes, skip_upload=False, skip_d2h=False, depth=None, pre_streams=0):
    keep = [torch.cuda.Stream() for _ in range(pre_streams)]      # shifts the executor's streams in torch's pool (HW queue mapping)
    ex = L.DetectStream(det, lanes=lanes)
    if skip_upload:
        orig = torch.Tensor.copy_
    depth = depth if depth is not None else 2 * lanes - 1
    def one(fill):
        st = ex.stage(B)
        for b in range(B):
            v = st.view(b, H0, W0)
            if fill: v[:] = 127
        ex.submit(st)
        while ex.pending() > depth:
            ex.fetch()
    if skip_upload:
        ex._copy_real = ex._copy
    for i in range(12): one(i < 4)
    ex.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): one(False)
    ex.drain(); torch.cuda.synchronize()
    print('   queues distinct:', ex.queues_distinct, end=' ')
    return (time.perf_counter() - t0) / K * 1e3

for lanes in (2, 3):
    for pre in (0, 1, 2, 3, 5):
        print(f'lanes {lanes} pre_streams {pre}: {run(lanes, pre_streams=pre):.3f} ms/batch', flush=True)
print('depth sweep, 2 lanes:', [round(run(2, depth=d), 3) for d in (1, 2, 3)])
