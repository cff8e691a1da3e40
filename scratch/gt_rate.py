"""Timing of the on-device GT encoder vs the host numpy path (bs=20, KITTI-like box counts)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import boxes as hb, ops
from squeezedet_pytorch_amd.annotations import encode_annotations, pack_annotations, anchors_f64_on
cfg = sqd.make_cfg(arch='squeezedet', device='cuda')
rs = np.random.RandomState(0)
for nb in (5, 10, 20):
    bl, cl = [], []
    for b in range(20):
        x1 = rs.uniform(0, 1100, nb); y1 = rs.uniform(0, 300, nb)
        bl.append(np.stack([x1, y1, np.minimum(x1 + rs.uniform(20, 300, nb), 1247), np.minimum(y1 + rs.uniform(20, 150, nb), 383)], 1).astype(np.float32))
        cl.append(rs.randint(0, 3, nb))
    t0 = time.perf_counter()
    for c, b in zip(cl, bl): hb.prepare_annotations(c, b, cfg.anchors, 3)
    t_host = time.perf_counter() - t0
    encode_annotations(cl, bl, cfg.anchors, 3); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): gt = encode_annotations(cl, bl, cfg.anchors, 3)
    torch.cuda.synchronize(); t_e2e = (time.perf_counter() - t0) / 20
    boxes, cls, offs = pack_annotations(cl, bl)
    d = [torch.from_numpy(x).cuda() for x in (boxes, cls, offs)]
    a64 = anchors_f64_on(cfg.anchors, 'cuda')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.encode_gt(d[0], d[1], d[2], a64, 3); torch.cuda.synchronize()
    e0.record()
    for _ in range(50): ops.encode_gt(d[0], d[1], d[2], a64, 3)
    e1.record(); torch.cuda.synchronize()
    print(f'{nb} boxes/img x 20 img: host numpy (1 core) {t_host*1e3:.1f} ms; device end-to-end incl. pack+H2D {t_e2e*1e3:.3f} ms; kernel {e0.elapsed_time(e1)/50*1e3:.1f} us', flush=True)
