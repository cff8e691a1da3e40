"""GPU tier: the counter-based dropout in front of ConvDet (reference: nn.Dropout(p, inplace=True), src/model/squeezedet.py:71-72,
81-82; csrc/sqd_common.h).  The keep decision of an element is a pure function of (seed, step, element index): the stand-alone mask
kernel, the fused epilogues of the last Fire's two expand launches and the host restatement must agree bit for bit; a training step
with the fused form equals the same step with that mask injected; the step counter advances once per forward."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import synthetic

pytestmark = pytest.mark.gpu


def _ops():
    from squeezedet_pytorch_amd import ops
    return ops


def _rand(*shape, seed=0, scale=1.0):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32))


@pytest.mark.parametrize('p', [0.5, 0.2, 0.75])
def test_mask_kernel_equals_host_restatement(p):
    ops = _ops()
    d = ops.DropState(p, 0x123456789abc, 'cuda', step=3)
    shape = (2, 5, 7, 24)
    m = ops.dropout_mask(d, shape).cpu().numpy().reshape(-1)
    seed, step = d.get()
    assert (seed, step) == (0x123456789abc, 3)
    ref = ops.dropout_mask_reference(seed, step, d.keep16, d.scale, m.size)
    assert np.array_equal(m, ref)
    big = ops.dropout_mask(d, (4, 24, 78, 768)).cpu().numpy().reshape(-1)
    keep = float((big > 0).mean())
    assert abs(keep - (1.0 - p)) < 2e-3 and set(np.unique(big).tolist()) <= {0.0, float(np.float32(d.scale))}
    ops.dropout_advance(d)
    assert d.get() == (seed, 4)
    m2 = ops.dropout_mask(d, shape).cpu().numpy().reshape(-1)
    assert not np.array_equal(m, m2) and np.array_equal(m2, ops.dropout_mask_reference(seed, 4, d.keep16, d.scale, m2.size))


@pytest.mark.parametrize('C,E1,E3,B,H,W', [(96, 384, 384, 2, 24, 78), (48, 192, 192, 1, 9, 21), (64, 256, 256, 3, 5, 17)])
def test_fused_epilogues_equal_the_mask(C, E1, E3, B, H, W):
    """expand1x1 (weight-stationary 1x1 kernel) and expand3x3 (balanced Winograd kernel) with the fused dropout == the same launches
    without it times the mask of their output buffer, bit for bit (multiplying by 0 or by the positive scale commutes with the ReLU)."""
    ops = _ops()
    x = torch.relu(_rand(B, H, W, C, seed=1)).cuda()
    w1 = (_rand(E1, C, 1, 1, seed=2) * (2.0 / C) ** 0.5).cuda(); b1 = (_rand(E1, seed=3) * 0.1).cuda()
    w3 = (_rand(E3, C, 3, 3, seed=4) * (2.0 / (9 * C)) ** 0.5).cuda(); b3 = (_rand(E3, seed=5) * 0.1).cuda()
    d = ops.DropState(0.5, 777, 'cuda', step=11)
    cfg1 = ops.conv_drop_cfg(C, E1, B * H * W)
    assert cfg1 is not None
    p1 = ops.ConvPlan(w1, b1, cfg1)
    p3 = ops.WinoPlan(w3, b3, ops.WINO_SK_CFG)
    plain = torch.empty(B, H, W, E1 + E3, device='cuda'); dropped = torch.full_like(plain, float('nan'))
    ops.conv(x, 0, p1, plain, 0, relu=True); ops.conv_wino(x, 0, p3, plain, E1, relu=True)
    ops.conv(x, 0, p1, dropped, 0, relu=True, drop=d); ops.conv_wino(x, 0, p3, dropped, E1, relu=True, drop=d)
    mask = ops.dropout_mask(d, tuple(plain.shape))
    assert torch.equal(dropped, plain * mask)
    assert d.get() == (777, 11), 'a launch without drop_advance must not advance the step'
    with pytest.raises(ValueError):
        ops.conv(x, 0, p1, dropped, 0, relu=True, drop=d, accumulate=True)
    with pytest.raises(ValueError):
        ops.conv_wino(x, 0, ops.WinoPlan(w3, b3, 2), dropped, E1, relu=True, drop=d)


def _model(size, p=0.5, arch='squeezedet'):
    from squeezedet_pytorch_amd.model import SqueezeDetWithLoss
    cfg = sqd.make_cfg(arch=arch, input_size=size, dropout_prob=p)
    m = SqueezeDetWithLoss(cfg)
    sd = synthetic.make_state_dict(arch, seed=1234)
    m.load_state_dict(sd, strict=True)
    return cfg, m.cuda().train(), sd


@pytest.mark.parametrize('arch', ['squeezedet', 'squeezedetplus'])
def test_training_step_with_fused_dropout_vs_oracle_with_the_same_mask(arch):
    """One training iteration with the fused dropout: the step counter advances by one, and losses / gradients equal the float64 oracle
    run with THAT step's mask (read back through the stand-alone kernel) -- and the same model with the mask injected."""
    ops = _ops()
    size = (64, 96)
    torch.manual_seed(21)
    cfg, m, sd = _model(size, arch=arch)
    x = synthetic.make_images(2, size, seed=3)
    gt = synthetic.make_gt(2, cfg.anchors, size, seed=2, min_boxes=2, max_boxes=3)
    d = m.base.drop_state(torch.device('cuda', torch.cuda.current_device()))
    seed, step0 = d.get()
    cch = 768 if arch == 'squeezedet' else 512
    mask_nhwc = ops.dropout_mask(d, (2, 4, 6, cch))                      # the mask the NEXT forward will apply
    loss, _ = m({'image': x.cuda(), 'gt': gt.cuda()})
    loss.mean().backward()
    assert d.get() == (seed, step0 + 1)
    mask = mask_nhwc.permute(0, 3, 1, 2).contiguous().cpu()
    sd64 = {k: v.double() for k, v in sd.items()}
    _, _, grads, total, loss_vec, _ = oracle.train_step_reference(sd64, None, x.double(), gt.double(), cfg.anchors.astype(np.float64), size,
                                                                  arch=arch, drop_mask=mask.double())
    np.testing.assert_allclose(loss.detach().cpu().numpy(), loss_vec.numpy(), rtol=1e-4)
    from test_training_gpu import _check_grads_flip_aware
    _check_grads_flip_aware(m.named_parameters(), grads)
    got = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    # the same step with the mask injected (tensor multiply in the expand epilogues, mask tensor read by the ConvDet data gradient)
    m.zero_grad()
    m.base._forced_drop_mask = mask
    loss2, _ = m({'image': x.cuda(), 'gt': gt.cuda()})
    loss2.mean().backward()
    assert d.get() == (seed, step0 + 1), 'an injected mask must not consume the stream'
    np.testing.assert_allclose(loss2.detach().cpu().numpy(), loss.detach().cpu().numpy(), rtol=1e-5)
    for n, p in m.named_parameters():
        ref = p.grad
        scale = float(ref.abs().max()) + 1e-12
        assert float((got[n] - ref).abs().max()) <= 2e-3 * scale, n
    # eval mode: no dropout, no advance
    m.base._forced_drop_mask = None
    m.eval()
    with torch.no_grad():
        pred = m.base(x.cuda())
    assert d.get() == (seed, step0 + 1)
    np.testing.assert_allclose(pred.cpu().numpy(), oracle.backbone_forward(x, sd, arch=arch).numpy(), atol=1e-4)


def test_masks_differ_between_steps_and_seeds_and_follow_manual_seed():
    size = (64, 96)
    torch.manual_seed(5)
    cfg, m, _ = _model(size)
    dev = torch.device('cuda', torch.cuda.current_device())
    x = synthetic.make_images(2, size, seed=3).cuda()
    m.base.use_winograd = True
    with torch.no_grad():
        a = m.base(x).clone(); b = m.base(x).clone()
    assert not torch.equal(a, b), 'two forwards drew the same mask'
    s1 = m.base.drop_state(dev).get()
    assert s1[1] == 2
    torch.manual_seed(6)                       # another seed: another stream, from step 0
    with torch.no_grad():
        c = m.base(x).clone()
    assert not torch.equal(a, c) and m.base.drop_state(dev).get()[1] == 1
    torch.manual_seed(5)                       # back to the first seed: its stream restarts, the first forward repeats
    with torch.no_grad():
        a2 = m.base(x).clone()
    assert torch.equal(a, a2)
    # the stand-alone form (fused_dropout off) draws the same masks: same outputs up to the kernels' rounding
    torch.manual_seed(6); m.base.drop_state(dev); torch.manual_seed(5)
    m.base.fused_dropout = False
    with torch.no_grad():
        a3 = m.base(x).clone()
    assert float((a3 - a).abs().max()) <= 1e-4 * max(1.0, float(a.abs().max()))
    assert m.base.drop_state(dev).get()[1] == 1


def test_restored_dropout_stream_survives_reseeding():
    """a stream restored by ``set_dropout_rng`` (checkpoint resume) used to be dropped -- silently restarted at step 0 --
    as soon as torch was re-seeded (the per-rank offset of ``attach_data_parallel`` on ranks > 0, a ``torch.manual_seed`` of the
    training script).  It now stays until ``set_dropout_rng`` replaces it; a stream DERIVED from torch's seed still follows it."""
    import squeezedet_pytorch_amd as sqd
    from squeezedet_pytorch_amd.model import SqueezeDetBase
    cfg = sqd.make_cfg(input_size=(64, 96))
    base = SqueezeDetBase(cfg).cuda()
    torch.manual_seed(11)
    d0 = base.drop_state('cuda:0')
    assert base.get_dropout_rng()[1] == 0
    torch.manual_seed(12)                                   # derived stream: follows the new seed
    assert base.drop_state('cuda:0') is not d0
    base.set_dropout_rng(0x1234567, 41, 'cuda:0')
    torch.manual_seed(13)
    torch.manual_seed(torch.initial_seed() + 3)             # what attach_data_parallel does on rank 3
    assert tuple(base.get_dropout_rng()) == (0x1234567, 41)
    kept = base.drop_state('cuda:0')
    assert tuple(kept.get()) == (0x1234567, 41)
    base.set_dropout_rng(99, 7, 'cuda:0')
    assert tuple(base.drop_state('cuda:0').get()) == (99, 7)
