"""Stress of the lane executor: many batches of mixed image sizes and batch sizes through Detector.detect_stream, every 16th batch checked
bit for bit against detect_images; device / pinned memory must not grow after the first pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import synthetic
from squeezedet_pytorch_amd.detector import Detector
from squeezedet_pytorch_amd.model import SqueezeDet
cfg = sqd.make_cfg(); cfg.batch_size = 20
m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict()); det = Detector(m, cfg)
rs = np.random.RandomState(1)
sizes = [(375, 1242), (370, 1224), (374, 1238), (376, 1241), (384, 1248), (360, 1200)]
pool = []
for i in range(48):
    h, w = sizes[i % len(sizes)]
    base = rs.standard_normal((-(-h // 8), -(-w // 8), 3)) * 60 + 100
    pool.append(np.clip(np.kron(base, np.ones((8, 8, 1))), 0, 255).astype(np.uint8)[:h, :w])
NB = int(os.environ.get('NB', 600))
def batches():
    for b in range(NB):
        n = 20 if b % 37 else int(rs.randint(1, 20))          # mostly full batches, now and then a ragged one
        yield [pool[(b * 7 + k) % len(pool)] for k in range(n)]
bl = list(batches())
def one_pass():
    nimg = bad = 0
    for it, res in enumerate(det.detect_stream(iter(bl))):
        nimg += len(res)
        if it % 16 == 0:
            want = det.detect_images(bl[it])
            for r, w in zip(res, want):
                same = ('boxes' in r) == ('boxes' in w) and (('boxes' not in r) or (np.array_equal(r['boxes'], w['boxes']) and np.array_equal(r['anchor_idx'], w['anchor_idx']) and np.array_equal(r['scores'], w['scores'])))
                bad += 0 if same else 1
    return nimg, bad
t0 = time.time()
nimg, bad = one_pass()
dt = time.time() - t0
mem1 = torch.cuda.memory_reserved()
ex = det.stream()
print(f'pass 1: {NB} batches, {nimg} images in {dt:.1f} s ({nimg / dt:.0f} img/s incl. the checks); mismatching images {bad}; captures {ex.captures}, eager {ex.eager_batches}, replayed {ex.replayed_batches}, degraded {ex.degraded}')
t0 = time.time()
nimg2, bad2 = one_pass()
dt2 = time.time() - t0
mem2 = torch.cuda.memory_reserved()
print(f'pass 2: {nimg2} images in {dt2:.1f} s ({nimg2 / dt2:.0f} img/s); mismatching images {bad2}; captures {ex.captures}; reserved device memory {mem1 / 2**20:.0f} MiB after pass 1, {mem2 / 2**20:.0f} MiB after pass 2')
nimg3, bad3 = one_pass()
mem3 = torch.cuda.memory_reserved()
print(f'pass 3: mismatching images {bad3}; captures {ex.captures}; reserved {mem3 / 2**20:.0f} MiB')
assert bad == 0 and bad2 == 0 and bad3 == 0 and not ex.degraded and mem3 <= mem2 + (64 << 20) and ex.captures <= 2 * 2
print('stress ok')
