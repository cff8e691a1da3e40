"""Time the fused detect launch alone on a bs=20 KITTI-size pred (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import ops
cfg = sqd.make_cfg()
B = int(os.environ.get('BATCH', 20))
rs = np.random.RandomState(11)
pred = torch.from_numpy((rs.standard_normal((B, 16848, 8)) * np.array([2, 2, 2, 2, .4, .4, .4, .4]) + np.array([0, 0, 0, -2, 0, 0, 0, 0])).astype(np.float32)).cuda()
anc = torch.from_numpy(cfg.anchors).float().cuda()
bufs = ops._det_buffers(B, 64, pred.device, 16848)
f = lambda: ops.detect(pred, anc, cfg.input_size, 3, 64, 0.4, 0.3, out=bufs)
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f()
e1.record(); torch.cuda.synchronize()
print(f'detect bs={B}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch, counts {bufs[0].tolist()[:6]}')
