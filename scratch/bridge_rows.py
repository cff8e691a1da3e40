"""For other batch sizes: does the whole inference step get faster with the two bridge launches?  Injects Y:/Z: rows for the batch's
pixel count into the live table, times hipGraph replays of the step with fuse_fire_bridge on / off (nseg swept), prints rows to add."""
import sys, json; sys.path.insert(0, '.')
import torch
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import ops, synthetic
from squeezedet_pytorch_amd.detector import Detector
from squeezedet_pytorch_amd.model import SqueezeDet
cfg = sqd.make_cfg(arch='squeezedet', device='cuda')
m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict('squeezedet', seed=1234)); det = Detector(m, cfg)
tab = ops._tuning()
def step_ms(x, bufs):
    m.base.invalidate_plans()
    with torch.no_grad():
        for _ in range(3): det.detect_device(x, out=bufs)
        torch.cuda.synchronize()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            det.detect_device(x, out=bufs); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side): det.detect_device(x, out=bufs)
        torch.cuda.current_stream().wait_stream(side)
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): g.replay()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 30)
    return sorted(ts)[1]
out = {}
for B in [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16, 32, 40, 64]:
    x = synthetic.make_images(B, cfg.input_size, seed=0).cuda()
    bufs = ops._det_buffers(B, cfg.keep_top_k, x.device, cfg.num_anchors)
    npix = B * 96 * 312
    ky, kz = f'Y:16:64:64:16:{npix}', f'Z:16:64:64:32:{npix}'
    tab.pop(ky, None); tab.pop(kz, None)
    t0 = step_ms(x, bufs)
    tab[ky] = 12
    ty = step_ms(x, bufs)
    best = (None, 1e9)
    for nseg in (1, 2, 3, 4, 6, 8, 12, 24):
        tab[kz] = nseg
        t = step_ms(x, bufs)
        if t < best[1]: best = (nseg, t)
    tab[kz] = best[0]
    print(f'bs={B}: plain {t0:.4f} ms, +fire bridge {ty:.4f}, +pool bridge (nseg {best[0]}) {best[1]:.4f}  ({B / t0 * 1e3:.0f} -> {B / best[1] * 1e3:.0f} img/s)', flush=True)
    out[B] = dict(plain=t0, y=ty, z=best[1], nseg=best[0])
json.dump(out, open('gpurun_out/bridge_rows.json', 'w'))
