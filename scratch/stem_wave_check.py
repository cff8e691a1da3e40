"""A/B of the fused stem kernels: stem_wave_kernel<PH> (SQD_STEM_WAVE = 3 / 4) vs the workgroup kernel (0): bit equality + time."""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, '.')
    import torch
    from squeezedet_pytorch_amd import ops
    torch.manual_seed(0)
    x = torch.randn(20, 3, 384, 1248, device='cuda'); w = torch.randn(64, 3, 3, 3, device='cuda') * 0.2; b = torch.randn(64, device='cuda') * 0.1
    am = torch.empty(20, 96, 312, 64, dtype=torch.uint8, device='cuda')
    y_tr = ops.stem_pool(x, w, b, argmax=am)
    y = ops.stem_pool(x, w, b)
    print(sys.argv[1], 'equal to the argmax kernel:', torch.equal(y, y_tr), 'max diff', (y - y_tr).abs().max().item())
    for H, W in ((64, 96), (52, 68), (12, 16), (384, 1248), (100, 40)):
        xs = torch.randn(2, 3, H, W, device='cuda')
        ams = torch.empty(0, dtype=torch.uint8, device='cuda')
        a = ops.stem_pool(xs, w, b)
        Hp, Wp = a.shape[1], a.shape[2]
        ams = torch.empty(2, Hp, Wp, 64, dtype=torch.uint8, device='cuda')
        r = ops.stem_pool(xs, w, b, argmax=ams)
        print('  ', H, W, 'equal', torch.equal(a, r))
    for _ in range(3): ops.stem_pool(x, w, b)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.stem_pool(x, w, b)
    e1.record(); torch.cuda.synchronize()
    print(sys.argv[1], f'{e0.elapsed_time(e1) / 50 * 1e3:.1f} us')
else:
    for ph in ('0', '3', '4', '2'):
        subprocess.run([sys.executable, __file__, ph], env=dict(os.environ, SQD_STEM_WAVE=ph), check=False)
