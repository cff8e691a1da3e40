"""Where the host spends its time in Detector.detect_dataset (in-memory uint8 images, bs=20): cProfile of the main thread."""
import cProfile, pstats, io, contextlib, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import synthetic
from squeezedet_pytorch_amd.detector import Detector
from squeezedet_pytorch_amd.model import SqueezeDet
cfg = sqd.make_cfg(device='cuda'); cfg.batch_size, cfg.num_workers, cfg.print_interval = 20, 8, 1 << 30
m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict('squeezedet', seed=1234))
det = Detector(m, cfg)
rs = np.random.RandomState(0)
pix = [rs.randint(0, 256, (375, 1242, 3), dtype=np.uint8) for _ in range(8)]
class DS:
    rgb_mean = rgb_std = None
    def __init__(self, n): self.n = n
    def __len__(self): return self.n
    def load_image(self, i): return pix[i % 8], f'{i:06d}'
with contextlib.redirect_stdout(io.StringIO()):
    det.detect_dataset(DS(40 * 20))
    t0 = time.perf_counter(); det.detect_dataset(DS(100 * 20)); t1 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable(); det.detect_dataset(DS(100 * 20)); pr.disable()
print(f'{(t1 - t0) / 100 * 1e3:.3f} ms per batch unprofiled')
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:5000])
