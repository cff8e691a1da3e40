import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch, ctypes
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import ops, synthetic, _native as nat
from squeezedet_pytorch_amd.model import SqueezeDet
from squeezedet_pytorch_amd.detector import Detector
from squeezedet_pytorch_amd.preprocess import preprocess_batch, KITTI_RGB_MEAN, KITTI_RGB_STD
cfg = sqd.make_cfg(); m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict()); det = Detector(m, cfg)
B = 20
rs = np.random.RandomState(0)
images = [rs.randint(0, 256, (375, 1242, 3), dtype=np.uint8) for _ in range(B)]
# (1) the convenience path (host packing included)
for _ in range(3): det.detect_images(images)
torch.cuda.synchronize(); t = time.perf_counter(); n = 10
for _ in range(n): det.detect_images(images)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print(f'detect_images (host pack + H2D + preprocess + net + detect + D2H): {dt*1e3:.2f} ms/batch = {B/dt:.0f} img/s')
# (2) pre-packed pinned uint8 buffer: H2D + preprocess kernel + net + detect, double-buffered on two streams
total = sum(im.size for im in images)
pinned = torch.empty(total, dtype=torch.uint8, pin_memory=True)
off = np.cumsum([0] + [im.size for im in images[:-1]]).astype(np.int64)
for o, im in zip(off, images): pinned.numpy()[o:o + im.size] = im.reshape(-1)
d_off = torch.from_numpy(off).cuda(); d_sizes = torch.tensor([[375, 1242]] * B, dtype=torch.int32).cuda()
mean = (ctypes.c_float * 3)(*KITTI_RGB_MEAN.tolist()); std = (ctypes.c_float * 3)(*KITTI_RGB_STD.tolist())
bufs = [dict(src=torch.empty(total, dtype=torch.uint8, device='cuda'), img=torch.empty(B, 3, 384, 1248, device='cuda'),
             sc=torch.empty(B, 2, device='cuda'), out=ops._det_buffers(B, 64, 'cuda', cfg.num_anchors), stream=torch.cuda.Stream()) for _ in range(2)]
def one(i):
    b = bufs[i & 1]
    with torch.cuda.stream(b['stream']):
        b['src'].copy_(pinned, non_blocking=True)
        rc = nat.lib().sqd_preprocess_u8_fwd(nat.ptr(b['src']), nat.ptr(d_off), nat.ptr(d_sizes), nat.ptr(b['img']), nat.ptr(b['sc']), mean, std, B, 384, 1248, nat.stream_handle())
        assert rc == 0
        det.detect_device(b['img'], scales=b['sc'], out=b['out'])
for i in range(6): one(i)
torch.cuda.synchronize(); t = time.perf_counter(); n = 40
for i in range(n): one(i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print(f'pre-packed pinned uint8 -> H2D -> preprocess -> net -> detect (2 streams, eager): {dt*1e3:.2f} ms/batch = {B/dt:.0f} img/s  ({total/1e6:.1f} MB/batch over PCIe)')
# (3) preprocess kernel alone
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
b = bufs[0]; e0.record()
for _ in range(20):
    nat.lib().sqd_preprocess_u8_fwd(nat.ptr(b['src']), nat.ptr(d_off), nat.ptr(d_sizes), nat.ptr(b['img']), nat.ptr(b['sc']), mean, std, B, 384, 1248, nat.stream_handle())
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f'preprocess kernel alone: {us:.1f} us/batch ({(total + B*3*384*1248*4)/us/1e3:.0f} GB/s)')
