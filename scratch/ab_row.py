"""A/B of a SqueezeDetBase switch inside the whole inference step (hipGraph replays).  usage: ab_flags.py <attr> [arch]"""
import sys; sys.path.insert(0, '.')
import torch
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import ops, synthetic
from squeezedet_pytorch_amd.detector import Detector
from squeezedet_pytorch_amd.model import SqueezeDet
key = sys.argv[1]; val = int(sys.argv[2]); arch = "squeezedet"
B = 16 if arch == 'squeezedetplus' else 20
cfg = sqd.make_cfg(arch=arch, device='cuda')
m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict(arch, seed=1234)); det = Detector(m, cfg)
x = synthetic.make_images(B, cfg.input_size, seed=0).cuda()
bufs = ops._det_buffers(B, cfg.keep_top_k, x.device, cfg.num_anchors)
def step_ms():
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
        for _ in range(50): g.replay()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 50)
    return sorted(ts)[1]
ref = None
for rep in range(3):
    for on in (False, True):
        tab = ops._tuning(); tab.pop(key, None)
        if on: tab[key] = val
        print(f'{key} {"on" if on else "off"}: {step_ms():.4f} ms', flush=True)
