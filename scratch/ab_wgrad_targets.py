"""Training step (hipGraph) time vs the split-K workgroup targets of the weight-gradient kernels.  usage: ab_wgrad_targets.py [k=v ...] (one config) or no args (sweep in subprocesses)"""
import sys, os, subprocess
sys.path.insert(0, '.')
if len(sys.argv) == 1:
    base = dict(_TARGET_WGS=1536, _TARGET_WGS_1X1=1024, _TARGET_WGS_WINO=512)
    cfgs = [dict(base)] + [dict(base, _TARGET_WGS_1X1=v) for v in (int(a) for a in os.environ.get('SWEEP_1X1', '256 384 512 640').split())] + [dict(base)]
    for c in cfgs:
        out = subprocess.run([sys.executable, __file__] + [f'{k}={v}' for k, v in c.items()], capture_output=True, text=True)
        print(c, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
    sys.exit(0)
import torch
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import ops, synthetic
from squeezedet_pytorch_amd.model import SqueezeDetWithLoss
for kv in sys.argv[1:]:
    k, v = kv.split("="); setattr(ops.tiles, k, int(v))
cfg = sqd.make_cfg(arch='squeezedet', device='cuda')
m = SqueezeDetWithLoss(cfg); m.load_state_dict(synthetic.make_state_dict('squeezedet', seed=1234)); m = m.cuda().train()
params = [p for p in m.parameters() if p.requires_grad]
opt = torch.optim.SGD(params, lr=cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
batch = {'image': synthetic.make_images(20, cfg.input_size, seed=0).cuda(), 'gt': synthetic.make_gt(20, cfg.anchors, cfg.input_size, cfg.num_classes, seed=1).cuda()}
def step():
    loss, _ = m(batch); loss = loss.mean(); opt.zero_grad(); loss.backward(); torch.nn.utils.clip_grad_norm_(params, cfg.grad_norm); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side): step()
torch.cuda.current_stream().wait_stream(side)
for _ in range(5): g.replay()
torch.cuda.synchronize(); ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
print(f'{sorted(ts)[1]:.4f} ms/step')
