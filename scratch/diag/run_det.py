import ctypes, os, sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
here = os.path.dirname(os.path.abspath(__file__))
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import ops, synthetic, _native as nat
from squeezedet_pytorch_amd.model import SqueezeDet
cfg = sqd.make_cfg(); m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict()); m = m.cuda().eval()
x = synthetic.make_images(20, cfg.input_size).cuda()
with torch.no_grad(): pred = m.base(x)
lib = ctypes.CDLL(os.path.join(here, 'libdiag_det.so'))
lib.sqd_detect_fwd.argtypes = nat._SIGNATURES['sqd_detect_fwd']; lib.sqd_detect_fwd.restype = ctypes.c_int
B, A = 20, 16848
anc = torch.from_numpy(cfg.anchors).float().cuda()
cnt, cls, sc, bx, idx = ops._det_buffers(B, 64, 'cuda')
keys = torch.zeros(B * A + 64, dtype=torch.int32, device='cuda')
for _ in range(3):
    rc = lib.sqd_detect_fwd(nat.ptr(pred), nat.ptr(anc), None, nat.ptr(keys), nat.ptr(cnt), nat.ptr(cls), nat.ptr(sc), nat.ptr(bx), nat.ptr(idx), B, A, 3, 384, 1248, 64, 0.4, 0.3, nat.stream_handle())
    assert rc == 0
torch.cuda.synchronize()
st = keys[B * A:B * A + 18].cpu().numpy().view(np.uint64)
d = np.diff(st.astype(np.int64))
print('counts', cnt.tolist()[:5], 'M>0.3 for img0:', int((keys[:A] != 0).sum()))
names = ['load keys', 'compaction', 'select', 'rank', 'barrier->wave0', 'decode', 'iou rows', 'nms loop', 'compact out']
for n, v in zip(names, d[:9]): print(f'{n:16s} {v:8d} ticks')
print('total', st[8] - st[0], '(s_memtime ticks = shader cycles)')
