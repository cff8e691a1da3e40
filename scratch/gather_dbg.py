import os, sys
sys.path.insert(0, '.')
import numpy as np, torch, torch.nn.functional as F
from squeezedet_pytorch_amd import ops
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous()
torch.manual_seed(0)
for (B, H, W) in ((2, 74, 288), (2, 74, 256), (1, 74, 288), (2, 66, 288), (2, 74, 320), (3, 139, 356)):
  for scale in (0.5, 1.0, 3.0):
    x = torch.randn(B, 3, H, W) * scale; w = (torch.randn(64, 3, 3, 3) * 0.25).requires_grad_(True); b = (torch.randn(64) * 0.2).requires_grad_(True)
    ref = F.max_pool2d(F.relu(F.conv2d(x, w, b, stride=2, padding=1)), 3, 2, ceil_mode=True)
    dy = torch.randn_like(ref); ref.backward(dy)
    am = torch.empty(*nhwc(ref.detach()).shape, dtype=torch.uint8, device='cuda')
    ops.stem_pool(x.cuda(), w.detach().cuda(), b.detach().cuda(), argmax=am)
    out = {}
    for name, env in (('dense', '0'), ('gather', '1')):
        os.environ['SQD_STEM_WGRAD_GATHER'] = env
        dw, db = ops.stem_wgrad_pooled(nhwc(dy).cuda(), None, am, x.cuda(), 64, 3)
        out[name] = dw.cpu()
        e = (dw.cpu() - w.grad).abs()
        idx = int(e.argmax()); n, r = divmod(idx, 27)
        print(B, H, W, scale, name, 'max err', float(e.max()), 'at n', n, 'k', r, 'ref', float(w.grad.flatten()[idx]), 'got', float(dw.cpu().flatten()[idx]), 'gradmax', float(w.grad.abs().max()))
