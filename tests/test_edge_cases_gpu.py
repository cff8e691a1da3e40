"""GPU tier: edge cases of the hot path -- ragged input sizes, batch 1, other class counts, small top-k,
fewer anchors than K, all-below-threshold, degenerate boxes -- against the oracle."""
import numpy as np
import pytest
import torch

import oracle
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import synthetic

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rand(*shape, seed=0, scale=1.0):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32))


@pytest.mark.parametrize("size,batch", [((70, 100), 3), ((48, 48), 1), ((375 // 4 * 2, 1242 // 4), 2), ((384, 1248), 1)])
def test_forward_ragged_sizes_and_batches(size, batch):
    from squeezedet_pytorch_amd.model import SqueezeDet
    cfg = sqd.make_cfg(input_size=size)
    m = SqueezeDet(cfg)
    sd = synthetic.make_state_dict('squeezedet', seed=1234)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    x = synthetic.make_images(batch, size, seed=5)
    with torch.no_grad():
        det = m({'image': x.cuda()})
        pred = m.base(x.cuda())
        ref = oracle.backbone_forward(x, sd)
    assert tuple(pred.shape) == tuple(ref.shape) == (batch, cfg.num_anchors, 8)
    assert (pred.cpu() - ref).abs().max().item() <= TOL
    ids, sc, bx = oracle.inference_head(ref, cfg.anchors, size)
    np.testing.assert_allclose(det['scores'].cpu().numpy(), sc.numpy(), atol=TOL)
    np.testing.assert_allclose(det['boxes'].cpu().numpy(), bx.numpy(), atol=5e-3)


def test_training_ragged_size_and_batch1():
    from squeezedet_pytorch_amd.model import SqueezeDetWithLoss
    size = (70, 100)
    cfg = sqd.make_cfg(input_size=size, dropout_prob=0.0)
    m = SqueezeDetWithLoss(cfg)
    sd = synthetic.make_state_dict('squeezedet', seed=1234)
    m.load_state_dict(sd)
    m = m.cuda().train()
    x = synthetic.make_images(1, size, seed=6)
    gt = synthetic.make_gt(1, cfg.anchors, size, seed=3, min_boxes=2, max_boxes=2)
    loss, _ = m({'image': x.cuda(), 'gt': gt.cuda()})
    loss.mean().backward()
    sd64 = {k: v.double() for k, v in sd.items()}
    _, _, grads, total, loss_vec, _ = oracle.train_step_reference(sd64, None, x.double(), gt.double(), cfg.anchors.astype(np.float64), size)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), loss_vec.numpy(), rtol=1e-4)
    for name in ('base.convdet.weight', 'base.convdet.bias'):
        ref = grads[name].float()
        got = dict(m.named_parameters())[name].grad.cpu()
        assert (got - ref).abs().max().item() <= 2e-4 * max(float(ref.abs().max()), 1e-3)
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())))
    assert abs(gn - total) <= 2e-2 * total


@pytest.mark.parametrize("C", [1, 5, 16])
def test_decode_detect_loss_other_class_counts(C):
    from squeezedet_pytorch_amd import ops
    size = (64, 96)
    cfg = sqd.make_cfg(input_size=size, num_classes=C, class_names=tuple(f'c{i}' for i in range(C)))
    A = cfg.num_anchors
    pred = _rand(2, A, C + 5, seed=C) * torch.tensor([2.0] * (C + 1) + [0.4] * 4)
    anc = torch.from_numpy(cfg.anchors).float().cuda()
    ids, sc, bx = ops.decode(pred.cuda(), anc, size, C)
    ido, sco, bxo = oracle.inference_head(pred, cfg.anchors, size, C)
    np.testing.assert_allclose(sc.cpu().numpy(), sco.numpy(), atol=1e-6)
    np.testing.assert_allclose(bx.cpu().numpy(), bxo.numpy(), atol=1e-3)
    assert (ids.cpu().numpy() == ido.numpy()).mean() > 0.999
    cnt, cls, s2, b2, idx = (t.cpu().numpy() for t in ops.detect(pred.cuda(), anc, size, C, 64, 0.4, 0.3))
    for b in range(2):
        d = oracle.filter_detections(ido[b].numpy(), sco[b].numpy(), bxo[b].numpy(), 64, 0.4, 0.3, C)
        n = int(cnt[b])
        assert n == (0 if d is None else len(d['scores']))
        if d is not None:
            assert np.array_equal(idx[b, :n], d['anchor_idx']) and np.array_equal(cls[b, :n], d['class_ids'])
    # loss + gradient with C classes
    rs = np.random.RandomState(3)
    gt = np.zeros((2, A, C + 9), np.float32)
    for b in range(2):
        bxs = np.array([[5, 5, 40, 30], [50, 20, 90, 60]], np.float32)
        gt[b] = oracle.encode_gt(rs.randint(0, C, 2), bxs, cfg.anchors, C)
    gt = torch.from_numpy(gt)
    p = pred.clone().cuda().requires_grad_(True)
    losses, nobj = ops.loss_fwd(p.detach(), gt.cuda(), anc, size, C, (1., 3.75, 100., 6.))
    po = pred.clone().requires_grad_(True)
    lo, st = oracle.multitask_loss(po, gt, cfg.anchors, size, C)
    np.testing.assert_allclose(losses[3].cpu().numpy(), lo.detach().numpy(), rtol=1e-5)
    lo.sum().backward()
    coef = torch.ones(3, 2, device='cuda')
    dp = ops.loss_bwd(p.detach(), gt.cuda(), anc, nobj, coef, size, C, (1., 3.75, 100., 6.))
    np.testing.assert_allclose(dp.cpu().numpy(), po.grad.numpy(), rtol=2e-4, atol=1e-7)


def test_detect_small_topk_few_anchors_and_empty():
    from squeezedet_pytorch_amd import ops
    size = (64, 96)
    cfg = sqd.make_cfg(input_size=size)
    A = cfg.num_anchors            # 216
    pred = _rand(3, A, 8, seed=9) * torch.tensor([2., 2, 2, 2, .4, .4, .4, .4])
    pred[2, :, 3] = -20.0          # image 2: every confidence ~ 0 -> nothing above threshold
    anc = torch.from_numpy(cfg.anchors).float().cuda()
    ido, sco, bxo = oracle.inference_head(pred, cfg.anchors, size)
    for K, st in ((10, 0.3), (64, 0.3), (64, 0.05), (1, 0.0)):
        cnt, cls, s2, b2, idx = (t.cpu().numpy() for t in ops.detect(pred.cuda(), anc, size, 3, K, 0.4, st))
        for b in range(3):
            d = oracle.filter_detections(ido[b].numpy(), sco[b].numpy(), bxo[b].numpy(), K, 0.4, st, 3)
            n = int(cnt[b])
            assert n == (0 if d is None else len(d['scores'])), (K, st, b)
            if d is not None:
                assert np.array_equal(idx[b, :n], d['anchor_idx']), (K, st, b)
                np.testing.assert_allclose(s2[b, :n], d['scores'], atol=1e-6)
    # fewer anchors than K: 2x3 grid would need a 32x48 input; emulate with the dense filter on 20 anchors
    c = torch.zeros(1, 20, dtype=torch.int64); s = torch.linspace(0.9, 0.31, 20)[None]; bx = torch.zeros(1, 20, 4)
    bx[0, :, 0] = torch.arange(20) * 30; bx[0, :, 2] = bx[0, :, 0] + 10; bx[0, :, 3] = 10
    cnt, cls, s2, b2, idx = (t.cpu().numpy() for t in ops.filter_dense(c.cuda(), s.cuda(), bx.cuda(), 3))
    assert int(cnt[0]) == 20 and list(idx[0, :20]) == list(range(20))
    with pytest.raises(RuntimeError):
        ops.detect(pred.cuda(), anc, size, 3, 65, 0.4, 0.3)       # keep_top_k > 64 is unsupported, loudly


def test_cpu_input_and_missing_gpu_fail_loudly():
    from squeezedet_pytorch_amd.model import SqueezeDet
    cfg = sqd.make_cfg(input_size=(64, 96))
    m = SqueezeDet(cfg).cuda().eval()
    with pytest.raises(RuntimeError):
        m.base(torch.zeros(1, 3, 64, 96))                           # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        m.base(torch.zeros(1, 3, 64, 80).cuda())                    # anchors do not match this input size
