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
keys = torch.zeros(B * A + 256, dtype=torch.int32, device='cuda')
def run():
    rc = lib.sqd_detect_fwd(nat.ptr(pred), nat.ptr(anc), None, nat.ptr(keys), nat.ptr(cnt), nat.ptr(cls), nat.ptr(sc), nat.ptr(bx), nat.ptr(idx), B, A, 3, 384, 1248, 64, 0.4, 0.3, nat.stream_handle())
    assert rc == 0
for _ in range(5): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print('avg per detect call (both kernels, back-to-back): %.1f us' % (e0.elapsed_time(e1) / 20 * 1e3))
st = keys[B * A:B * A + 160].cpu().numpy().view(np.uint64).reshape(20, 4).astype(np.int64)
t0 = st[:, 0].min()
for b in range(20):
    print(f'block {b:2d}: start {(st[b,0]-t0)/100:7.2f} us  dur {(st[b,1]-st[b,0])/100:7.2f} us  cycles {st[b,3]-st[b,2]:7d}  -> {(st[b,3]-st[b,2])/max((st[b,1]-st[b,0])/100,1e-9)/1e3:.2f} GHz  M={int((keys[b*A:(b+1)*A]!=0).sum())}')

print('--- in pipeline (backbone then detect), per-block stamps of the last call ---')
evs = []
for it in range(12):
    with torch.no_grad(): pred2 = m.base(x)
    f = torch.cuda.Event(enable_timing=True); f.record()
    a = torch.cuda.Event(enable_timing=True); b_ = torch.cuda.Event(enable_timing=True)
    a.record()
    rc = lib.sqd_detect_fwd(nat.ptr(pred2), nat.ptr(anc), None, nat.ptr(keys), nat.ptr(cnt), nat.ptr(cls), nat.ptr(sc), nat.ptr(bx), nat.ptr(idx), B, A, 3, 384, 1248, 64, 0.4, 0.3, nat.stream_handle())
    b_.record(); evs.append((a, b_))
torch.cuda.synchronize()
print('detect in pipeline (events):', ['%.0f' % (a.elapsed_time(b_) * 1e3) for a, b_ in evs[2:]], 'us')
st = keys[B * A:B * A + 160].cpu().numpy().view(np.uint64).reshape(20, 4).astype(np.int64)
d = (st[:, 1] - st[:, 0]) / 100.0; cyc = st[:, 3] - st[:, 2]
print('block dur us: min %.1f max %.1f ; cycles min %d max %d ; clock GHz %.2f..%.2f' % (d.min(), d.max(), cyc.min(), cyc.max(), (cyc / d / 1e3).min(), (cyc / d / 1e3).max()))
