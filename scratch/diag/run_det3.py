import ctypes, os, sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
here = os.path.dirname(os.path.abspath(__file__))
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import ops, synthetic, _native as nat
from squeezedet_pytorch_amd.model import SqueezeDet
cfg = sqd.make_cfg(); m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict()); m = m.cuda().eval()
x = synthetic.make_images(20, cfg.input_size).cuda()
lib = ctypes.CDLL(os.path.join(here, 'libdiag_det.so'))
lib.sqd_detect_fwd.argtypes = nat._SIGNATURES['sqd_detect_fwd']; lib.sqd_detect_fwd.restype = ctypes.c_int
B, A = 20, 16848
anc = torch.from_numpy(cfg.anchors).float().cuda()
cnt, cls, sc, bx, idx = ops._det_buffers(B, 64, 'cuda')
keys = torch.zeros(B * A + 256, dtype=torch.int32, device='cuda')
def step():
    with torch.no_grad(): pred = m.base(x)
    rc = lib.sqd_detect_fwd(nat.ptr(pred), nat.ptr(anc), None, nat.ptr(keys), nat.ptr(cnt), nat.ptr(cls), nat.ptr(sc), nat.ptr(bx), nat.ptr(idx), B, A, 3, 384, 1248, 64, 0.4, 0.3, nat.stream_handle())
    assert rc == 0
for _ in range(3): step()
torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s): step()
torch.cuda.current_stream().wait_stream(s)
for _ in range(30): g.replay()
torch.cuda.synchronize()
st = keys[B * A:B * A + 160].cpu().numpy().view(np.uint64).reshape(20, 4).astype(np.int64)
d = (st[:, 1] - st[:, 0]) / 100.0; cyc = st[:, 3] - st[:, 2]
print('GRAPH replay: block dur us: min %.1f max %.1f ; cycles min %d max %d ; clock GHz %.2f..%.2f ; start spread %.1f us' % (d.min(), d.max(), cyc.min(), cyc.max(), (cyc / d / 1e3).min(), (cyc / d / 1e3).max(), (st[:,0].max()-st[:,0].min())/100.0))
