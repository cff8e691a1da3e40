"""GPU tier, path level: the HIP inference path (backbone -> decode -> fused detection) against the
oracle and the committed golden vectors, through the mirrored module surface."""
import os

import numpy as np
import pytest
import torch

import oracle
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import synthetic

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _model(arch, input_size, **kw):
    from squeezedet_pytorch_amd.model import SqueezeDet
    cfg = sqd.make_cfg(arch=arch, input_size=input_size, **kw)
    m = SqueezeDet(cfg)
    sd = synthetic.make_state_dict(arch, seed=1234)
    missing = m.load_state_dict(sd, strict=True)
    return cfg, m.cuda().eval(), sd


def _decode_pred():
    rs = np.random.RandomState(11)
    return torch.from_numpy((rs.standard_normal((2, 16848, 8)) * np.array([2, 2, 2, 2, .4, .4, .4, .4])).astype(np.float32))


def _class_margin(pred, cfg):
    """Gap between the best and the second-best class score of every anchor (oracle arithmetic on ``pred``)."""
    probs, _, scores, _, _ = oracle.resolve_predictions(pred, cfg.anchors, cfg.input_size, cfg.num_classes)
    top2 = torch.topk(probs * scores, 2, dim=2)[0]
    return (top2[..., 0] - top2[..., 1]).numpy()


def _assert_class_ids(got, want, margin, tol=2e-4):
    """class_ids are an argmax: bit-exact wherever the runner-up is more than ``tol`` behind (everywhere a 1e-4 difference
    in the inputs cannot change the winner); the few anchors inside the margin may flip."""
    diff = got != want
    assert not (diff & (margin > tol)).any(), f'{int((diff & (margin > tol)).sum())} class ids differ outside the tie margin'
    assert diff.sum() <= (margin <= tol).sum()


def test_state_dict_contract():
    from squeezedet_pytorch_amd.model import SqueezeDet, SqueezeDetWithLoss
    cfg = sqd.make_cfg()
    for cls in (SqueezeDet, SqueezeDetWithLoss):
        m = cls(cfg)
        sd = m.state_dict()
        assert {k: tuple(v.shape) for k, v in sd.items()} == oracle.param_shapes('squeezedet')
        assert len(sd) == 64 and sum(v.numel() for v in sd.values()) == 2082120
    bad = sqd.make_cfg()
    bad.arch = 'nope'
    with pytest.raises(ValueError, match='Invalid architecture.'):
        SqueezeDet(bad)


@pytest.mark.parametrize("arch", ["squeezedet", "squeezedetplus"])
def test_backbone_small_vs_golden_and_oracle(golden_dir, arch):
    g = np.load(os.path.join(golden_dir, "backbone_small.npz"))
    cfg, m, sd = _model(arch, (64, 96))
    x = synthetic.make_images(2, (64, 96), seed=3)
    with torch.no_grad():
        pred = m.base(x.cuda())
        det = m({'image': x.cuda()})
    np.testing.assert_allclose(pred.cpu().numpy(), g[f"{arch}_pred"], atol=TOL, rtol=0)
    np.testing.assert_allclose(det['scores'].cpu().numpy(), g[f"{arch}_scores"], atol=TOL, rtol=0)
    np.testing.assert_allclose(det['boxes'].cpu().numpy(), g[f"{arch}_boxes"], atol=5e-3, rtol=0)
    _assert_class_ids(det['class_ids'].cpu().numpy(), g[f"{arch}_class_ids"], _class_margin(torch.from_numpy(g[f"{arch}_pred"]), cfg))
    assert det['class_ids'].dtype == torch.int64


def test_kitti_forward_vs_oracle_and_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "kitti_full.npz"))
    cfg, m, sd = _model('squeezedet', (384, 1248))
    x = synthetic.make_images(2, (384, 1248), seed=0)
    with torch.no_grad():
        pred = m.base(x.cuda())
        ref = oracle.backbone_forward(x, sd)
    assert tuple(pred.shape) == (2, 16848, 8)
    err = (pred.cpu() - ref).abs().max().item()
    assert err <= TOL, err
    # the golden rows come from the reference run on the batch-1 draw of the same seed
    x1 = synthetic.make_images(1, (384, 1248), seed=0)
    with torch.no_grad():
        pred1 = m.base(x1.cuda())
    np.testing.assert_allclose(pred1[0, ::257].cpu().numpy(), g["pred_rows"], atol=TOL, rtol=0)
    top = g["top_idx"]
    np.testing.assert_allclose(pred1[0].cpu().numpy()[top], g["pred_top"], atol=TOL, rtol=0)


def test_decode_vs_golden(golden_dir):
    from squeezedet_pytorch_amd import ops
    g = np.load(os.path.join(golden_dir, "decode.npz"))
    cfg = sqd.make_cfg()
    pred = _decode_pred()
    anc = torch.from_numpy(cfg.anchors).float().cuda()
    ids, sc, bx = ops.decode(pred.cuda(), anc, cfg.input_size, 3)
    sel = g["sel"]
    np.testing.assert_allclose(sc.cpu().numpy()[:, sel], g["best"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(bx.cpu().numpy()[:, sel], g["boxes"], atol=1e-3, rtol=0)
    # identical pred on both sides: the decode kernel mirrors the oracle op for op, so the argmax may only differ where the
    # two best classes are within rounding of each other
    _assert_class_ids(ids.cpu().numpy()[:, sel], g["class_ids"], _class_margin(pred, cfg)[:, sel], tol=1e-6)


def _check_detect_against_oracle(pred, cfg, cnt, cls, sc, bx, idx, exact_scores=True):
    """Given identical pred, kept anchor indices must be bit-exact vs the oracle filter."""
    ids_o, sc_o, bx_o = oracle.inference_head(pred, cfg.anchors, cfg.input_size)
    for b in range(pred.shape[0]):
        d = oracle.filter_detections(ids_o[b].numpy(), sc_o[b].numpy(), bx_o[b].numpy(), cfg.keep_top_k,
                                     cfg.nms_thresh, cfg.score_thresh, cfg.num_classes)
        n = int(cnt[b])
        if d is None:
            assert n == 0
            continue
        assert n == len(d['scores'])
        assert np.array_equal(idx[b, :n], d['anchor_idx']), (idx[b, :n], d['anchor_idx'])
        assert np.array_equal(cls[b, :n], d['class_ids'])
        np.testing.assert_allclose(sc[b, :n], d['scores'], atol=1e-6, rtol=0)
        np.testing.assert_allclose(bx[b, :n], d['boxes'], atol=1e-3, rtol=0)


def test_fused_detect_index_exact_on_identical_pred(golden_dir):
    from squeezedet_pytorch_amd import ops
    cfg = sqd.make_cfg()
    pred = _decode_pred()
    anc = torch.from_numpy(cfg.anchors).float().cuda()
    out = ops.detect(pred.cuda(), anc, cfg.input_size, 3, 64, 0.4, 0.3)
    cnt, cls, sc, bx, idx = (t.cpu().numpy() for t in out)
    _check_detect_against_oracle(pred, cfg, cnt, cls, sc, bx, idx)
    g = np.load(os.path.join(golden_dir, "filter.npz"))
    for b in range(2):      # the reference's own Detector.filter output
        n = int(cnt[b])
        assert np.array_equal(cls[b, :n], g[f"syn{b}_class_ids"])
        np.testing.assert_allclose(sc[b, :n], g[f"syn{b}_scores"], atol=1e-6, rtol=0)
        np.testing.assert_allclose(bx[b, :n], g[f"syn{b}_boxes"], atol=1e-3, rtol=0)


def test_filter_dense_kats():
    from squeezedet_pytorch_amd import ops
    A = 200
    scores = np.linspace(0.9, 0.1, A).astype(np.float32)
    cls = (np.arange(A) % 3).astype(np.int64)
    boxes = np.zeros((A, 4), np.float32)
    boxes[:, 0] = np.arange(A) * 20; boxes[:, 2] = boxes[:, 0] + 10; boxes[:, 3] = 10
    cases = [(cls, scores, boxes),
             (np.zeros(A, np.int64), np.full(A, 0.5, np.float32), boxes),                    # all ties -> index order
             (np.zeros(A, np.int64), np.full(A, 0.3, np.float32), boxes),                    # == threshold -> none
             (cls[:40], scores[:40], boxes[:40]),                                            # fewer than K anchors
             ]
    # overlapping chain + degenerate boxes
    ob = np.array([[0, 0, 10, 10], [4.2, 0, 14.2, 10], [4.3, 20, 14.3, 30], [0, 20, 10, 30], [5, 5, 5, 5], [5, 5, 5, 5],
                   [10, 40, 4, 50], [0, 40, 10, 50]], np.float32)
    cases.append((np.zeros(8, np.int64), np.array([.9, .8, .7, .75, .6, .6, .5, .45], np.float32), ob))
    for c, s, b in cases:
        exp = oracle.filter_detections(c, s, b)
        out = ops.filter_dense(torch.from_numpy(c)[None].cuda(), torch.from_numpy(s)[None].cuda(), torch.from_numpy(b)[None].cuda(), 3)
        cnt, ocl, osc, obx, oidx = (t.cpu().numpy() for t in out)
        if exp is None:
            assert cnt[0] == 0
            continue
        n = int(cnt[0])
        assert n == len(exp['scores'])
        assert np.array_equal(oidx[0, :n], exp['anchor_idx'])
        assert np.array_equal(ocl[0, :n], exp['class_ids'])
        assert np.array_equal(osc[0, :n], exp['scores']) and np.array_equal(obx[0, :n], exp['boxes'])


def test_detector_end_to_end_kitti():
    """Full path at the KITTI size: HIP backbone + fused detect vs oracle backbone + oracle filter.
    Index-exactness end to end is only meaningful away from near-ties (fp32 summation order differs),
    so candidates whose oracle score is within 2e-4 of the top-k boundary / threshold are tolerated."""
    from squeezedet_pytorch_amd.detector import Detector
    from squeezedet_pytorch_amd.model import SqueezeDet
    cfg = sqd.make_cfg()
    m = SqueezeDet(cfg)
    sd = synthetic.make_state_dict('squeezedet', seed=1234)
    m.load_state_dict(sd)
    det = Detector(m, cfg)
    x = synthetic.make_images(2, cfg.input_size, seed=0)
    scales = np.array([384 / 375., 1248 / 1242.], np.float32)
    batch = {'image': x.cuda(), 'image_meta': {'scales': torch.from_numpy(np.stack([scales, scales])),
                                               'image_id': ['a', 'b']}}
    res = det.detect(batch)
    with torch.no_grad():
        pred_o = oracle.backbone_forward(x, sd)
        # stage A: fused kernel on the HIP pred must be index-exact vs the oracle filter on the SAME pred
        pred_h = det.model.base(x.cuda()).cpu()
    ids_h, sc_h, bx_h = oracle.inference_head(pred_h, cfg.anchors, cfg.input_size)
    ids_o, sc_o, bx_o = oracle.inference_head(pred_o, cfg.anchors, cfg.input_size)
    for b in range(2):
        d = oracle.filter_detections(ids_h[b].numpy(), sc_h[b].numpy(), bx_h[b].numpy())
        r = res[b]
        assert np.array_equal(r['anchor_idx'], d['anchor_idx'])
        assert np.array_equal(r['class_ids'], d['class_ids'])
        np.testing.assert_allclose(r['scores'], d['scores'], atol=1e-6)
        np.testing.assert_allclose(r['boxes'], oracle.boxes_postprocess(d['boxes'], scales), atol=2e-3)
        # stage B: against the oracle end to end (oracle backbone).  fp32 summation order differs between the two backbones,
        # so the outcome is only DETERMINED where no decision sits within the 1e-4 score / box tolerance of a boundary:
        # top-64 cut, threshold, rank order of overlapping same-class candidates, IoU vs 0.4.  Where it is determined the
        # kept anchors must be identical, in the same order; otherwise the two lists may differ only by fragile anchors.
        do = oracle.filter_detections(ids_o[b].numpy(), sc_o[b].numpy(), bx_o[b].numpy())
        so = sc_o[b].numpy()
        np.testing.assert_allclose(sc_h[b].numpy(), so, atol=TOL)
        order = np.argsort(-so, kind='stable')
        top = order[:65]
        eps = 2e-4
        fragile = set()
        if so[top[63]] - so[top[64]] <= eps:
            fragile |= {int(top[63]), int(top[64])}
        fragile |= {int(a) for a in top[:64] if abs(so[a] - 0.3) <= eps}
        cls_o, box_o = ids_o[b].numpy(), bx_o[b].numpy()
        for i in range(64):
            for j in range(i + 1, 64):
                ai, aj = top[i], top[j]
                if cls_o[ai] != cls_o[aj]:
                    continue
                bi, bj = box_o[ai], box_o[aj]
                w = max(0.0, min(bi[2], bj[2]) - max(bi[0], bj[0])); h = max(0.0, min(bi[3], bj[3]) - max(bi[1], bj[1]))
                inter = w * h
                union = (bi[2] - bi[0]) * (bi[3] - bi[1]) + (bj[2] - bj[0]) * (bj[3] - bj[1]) - inter
                iou = inter / union if union > 0 else 0.0
                if abs(iou - 0.4) <= 2e-3 or (iou > 0.39 and abs(so[ai] - so[aj]) <= eps):
                    fragile |= {int(ai), int(aj)}
        got, want = [int(v) for v in r['anchor_idx']], [int(v) for v in do['anchor_idx']]
        if not fragile:
            assert got == want, (got, want)
        else:
            # everything that differs must be explained by a fragile anchor (itself, or one that suppresses / frees it)
            assert len(set(got) ^ set(want)) <= 2 * len(fragile), (sorted(set(got) ^ set(want)), sorted(fragile))


def test_prediction_resolver_five_outputs_vs_golden(golden_dir):
    """PredictionResolver.forward returns the reference's own 5-tuple (src/model/squeezedet.py:109-120)."""
    from squeezedet_pytorch_amd.model import PredictionResolver
    g = np.load(os.path.join(golden_dir, "decode.npz"))
    cfg = sqd.make_cfg()
    pred = _decode_pred()
    res = PredictionResolver(cfg, log_softmax=True)
    probs, logp, scores, deltas, boxes = res(pred.cuda())
    assert tuple(scores.shape) == (2, 16848, 1) and tuple(probs.shape) == (2, 16848, 3)
    sel = g["sel"]
    np.testing.assert_allclose(probs.cpu().numpy()[:, sel], g["probs"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(logp.cpu().numpy()[:, sel], g["logp"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(scores.cpu().numpy()[:, sel], g["scores"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(boxes.cpu().numpy()[:, sel], g["boxes"], atol=1e-3, rtol=0)
    assert torch.equal(deltas.cpu(), pred[..., 4:])
    assert PredictionResolver(cfg, log_softmax=False)(pred.cuda())[1] is None


def test_eval_flow_images_to_ap(tmp_path):
    """The reference's eval flow end to end on the device path: raw uint8 images -> Detector.detect_images (GPU
    pre-processing, backbone, fused NMS) -> results.save_results (KITTI text files) -> results.evaluate (native AP).
    Ground truth = the detections themselves, so every evaluated class must come out with AP 1 (n_gt permitting)."""
    import os
    from squeezedet_pytorch_amd import results as R
    from squeezedet_pytorch_amd.detector import Detector
    from squeezedet_pytorch_amd.model import SqueezeDet
    cfg = sqd.make_cfg(arch='squeezedet', device='cuda')
    m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict('squeezedet', seed=1234))
    det = Detector(m, cfg)
    rs = np.random.RandomState(3)
    n = 6
    imgs = [rs.randint(0, 256, (375, 1242, 3), dtype=np.uint8) for _ in range(n)]
    ids = ['%06d' % i for i in range(n)]
    res = det.detect_images(imgs, image_ids=ids)
    assert len(res) == n and all(r['image_meta']['image_id'] == i for r, i in zip(res, ids))
    assert sum(len(r.get('class_ids', [])) for r in res) > 0
    R.save_results(res, str(tmp_path / 'results'))
    lab = tmp_path / 'training' / 'label_2'
    os.makedirs(lab)
    for r, i in zip(res, ids):
        with open(lab / (i + '.txt'), 'w') as f:
            for c, b in zip(r.get('class_ids', []), r.get('boxes', [])):
                # label = the detection as written to the results file (2 decimals), tall enough for every difficulty
                f.write('{} 0.00 0 0.00 {:.2f} {:.2f} {:.2f} {:.2f} 1.5 1.6 3.9 1.0 1.5 20.0 0.1\n'.format(R.KITTI_CLASS_NAMES[int(c)], *[float(v) for v in b]))
    with open(tmp_path / 'set.txt', 'w') as f:
        f.write('\n'.join(ids) + '\n')
    aps = R.evaluate(str(tmp_path / 'results'), str(lab), str(tmp_path / 'set.txt'))
    assert set(aps) == {f'{c}_{d}' for c in R.KITTI_CLASS_NAMES for d in ('easy', 'moderate', 'hard')} | {'mAP'}
    assert all(0.0 <= v <= 1.0 for v in aps.values()) and aps['mAP'] > 0.0


@pytest.fixture
def force_winograd(monkeypatch):
    """Route EVERY 3x3 convolution (forward, data gradient, weight gradient where it applies) through the Winograd kernels,
    whatever the measured table would choose for the test's small shapes."""
    from squeezedet_pytorch_amd import ops
    used = {'n': 0}

    def always(C, N, npix):
        if C % 8:
            return None
        used['n'] += 1
        return 2                                              # <2,4>: the configuration the headline workload runs
    monkeypatch.setattr(ops, 'choose_wino_cfg', always)
    monkeypatch.setattr(ops, 'WINO_WGRAD', True)
    return used


def test_golden_forward_through_winograd_kernels(golden_dir, force_winograd):
    """The reference-generated goldens (small backbone + KITTI-size rows) with all 3x3 layers forced onto the Winograd
    kernel: same 1e-4 bound as the direct path."""
    g = np.load(os.path.join(golden_dir, "backbone_small.npz"))
    cfg, m, sd = _model('squeezedet', (64, 96))
    m.base.fuse_expand = False                               # (the fused expand launch would take the 3x3 away from Winograd)
    x = synthetic.make_images(2, (64, 96), seed=3)
    with torch.no_grad():
        pred = m.base(x.cuda())
    np.testing.assert_allclose(pred.cpu().numpy(), g["squeezedet_pred"], atol=TOL, rtol=0)
    gk = np.load(os.path.join(golden_dir, "kitti_full.npz"))
    cfg, m, sd = _model('squeezedet', (384, 1248))
    m.base.fuse_expand = False
    x1 = synthetic.make_images(1, (384, 1248), seed=0)
    with torch.no_grad():
        pred1 = m.base(x1.cuda())
    np.testing.assert_allclose(pred1[0, ::257].cpu().numpy(), gk["pred_rows"], atol=TOL, rtol=0)
    np.testing.assert_allclose(pred1[0].cpu().numpy()[gk["top_idx"]], gk["pred_top"], atol=TOL, rtol=0)
    assert force_winograd['n'] >= 22                         # 10 expand3x3 + ConvDet, both models


def test_golden_forward_through_one_launch_winograd_fire(golden_dir, monkeypatch):
    """The reference-generated goldens with every Fire's expand pair forced onto the ONE-launch Winograd form (expand1x1 as the
    four inner transform positions; the shipped table has no X: rows because the separate launches measured faster, so
    the default path never takes it): same 1e-4 bound."""
    from squeezedet_pytorch_amd import ops
    used = {'n': 0}

    def always(C, E1, E3, npix):
        if C % 8 or E1 % 16:
            return None
        if not ops.fire_wino_cfg_ok(10, C):          # (larger squeezes: the U-stationary form does not fit; the streamed-U ids are retired)
            return None
        used['n'] += 1
        return 10
    monkeypatch.setattr(ops, 'choose_fire_wino_cfg', always)
    monkeypatch.setattr(ops, 'choose_wino_cfg', lambda C, N, npix: (2 if C % 8 == 0 else None))
    g = np.load(os.path.join(golden_dir, "backbone_small.npz"))
    for arch in ("squeezedet", "squeezedetplus"):
        cfg, m, sd = _model(arch, (64, 96))
        x = synthetic.make_images(2, (64, 96), seed=3)
        with torch.no_grad():
            pred = m.base(x.cuda())
        np.testing.assert_allclose(pred.cpu().numpy(), g[f"{arch}_pred"], atol=TOL, rtol=0)
    gk = np.load(os.path.join(golden_dir, "kitti_full.npz"))
    cfg, m, sd = _model('squeezedet', (384, 1248))
    x1 = synthetic.make_images(1, (384, 1248), seed=0)
    with torch.no_grad():
        pred1 = m.base(x1.cuda())
    np.testing.assert_allclose(pred1[0, ::257].cpu().numpy(), gk["pred_rows"], atol=TOL, rtol=0)
    assert used['n'] >= 12


def test_golden_forward_with_and_without_stem_squeeze(golden_dir):
    """The reference-generated goldens through the stem launch that also runs the first Fire's squeeze (default at inference,
    ``fuse_stem_squeeze``) and through the two separate launches: same 1e-4 bound, and the two paths agree to summation-order noise."""
    from squeezedet_pytorch_amd import timing
    g = np.load(os.path.join(golden_dir, "backbone_small.npz"))
    preds = {}
    for flag in (True, False):
        cfg, m, sd = _model('squeezedet', (64, 96))
        m.base.fuse_stem_squeeze = flag
        x = synthetic.make_images(2, (64, 96), seed=3)
        kt = timing.KernelTimer()
        timing.set_timer(kt)
        try:
            with torch.no_grad():
                preds[flag] = m.base(x.cuda())
        finally:
            timing.set_timer(None)
        torch.cuda.synchronize()
        names = [r[0] for r in kt.records]
        assert ('stem_pool_sq<3>' in names) == flag and ('stem_pool<3>' in names) == (not flag)
        np.testing.assert_allclose(preds[flag].cpu().numpy(), g["squeezedet_pred"], atol=TOL, rtol=0)
    assert (preds[True] - preds[False]).abs().max().item() <= 2e-5
    gk = np.load(os.path.join(golden_dir, "kitti_full.npz"))
    cfg, m, sd = _model('squeezedet', (384, 1248))
    assert m.base.fuse_stem_squeeze
    x1 = synthetic.make_images(1, (384, 1248), seed=0)
    with torch.no_grad():
        pred1 = m.base(x1.cuda())
    np.testing.assert_allclose(pred1[0, ::257].cpu().numpy(), gk["pred_rows"], atol=TOL, rtol=0)


def test_golden_forward_through_fire_bridges(golden_dir, monkeypatch):
    """The reference-generated goldens with every Fire -> Fire pair the bridge launches can take (expand pair + the next Fire's
    squeeze in one kernel, squeeze width <= 32: fire3 -> fire4 and fire6 -> fire7 of SqueezeDet; and through the max pool,
    fire4 -> pool -> fire6; none of SqueezeDet+; the shipped table takes fire3 -> fire4 and fire4 -> pool -> fire6 at the
    headline shape) forced onto them: same 1e-4 bound, and the switch ``fuse_fire_bridge = False`` gives the plain path back."""
    from squeezedet_pytorch_amd import ops
    used = {'n': 0}

    def always(C, E1, E3, Nsq, npix):
        for cid in (12, 10):
            if ops.fire_bridge_cfg_ok(cid, C, E3, E1, Nsq):
                used['n'] += 1
                return cid
        return None
    monkeypatch.setattr(ops, 'choose_fire_bridge_cfg', always)

    def always_pooled(C, E1, E3, Nsq, npix):
        if not ops.fire_pool_bridge_ok(C, E3, E1, Nsq):
            return None
        used['n'] += 1
        return 3
    monkeypatch.setattr(ops, 'choose_fire_pool_bridge', always_pooled)
    monkeypatch.setattr(ops, 'choose_wino_cfg', lambda C, N, npix: (2 if C % 8 == 0 else None))
    g = np.load(os.path.join(golden_dir, "backbone_small.npz"))
    for arch in ("squeezedet", "squeezedetplus"):
        cfg, m, sd = _model(arch, (64, 96))
        x = synthetic.make_images(2, (64, 96), seed=3)
        with torch.no_grad():
            pred = m.base(x.cuda())
            n_bridged = used['n']
            m.base.fuse_fire_bridge = False
            plain = m.base(x.cuda())
        assert used['n'] == n_bridged
        np.testing.assert_allclose(pred.cpu().numpy(), g[f"{arch}_pred"], atol=TOL, rtol=0)
        np.testing.assert_allclose(pred.cpu().numpy(), plain.cpu().numpy(), atol=2e-5, rtol=0)
    gk = np.load(os.path.join(golden_dir, "kitti_full.npz"))
    cfg, m, sd = _model('squeezedet', (384, 1248))
    x1 = synthetic.make_images(1, (384, 1248), seed=0)
    with torch.no_grad():
        pred1 = m.base(x1.cuda())
    np.testing.assert_allclose(pred1[0, ::257].cpu().numpy(), gk["pred_rows"], atol=TOL, rtol=0)
    np.testing.assert_allclose(pred1[0].cpu().numpy()[gk["top_idx"]], gk["pred_top"], atol=TOL, rtol=0)
    assert used['n'] == 4                                   # two bridges (fire3 -> fire4, fire4 -> pool -> fire6) x two SqueezeDet forwards


def test_filter_nms_boundary_gpu():
    """IoU == nms_thresh exactly (see tests/test_oracle_golden.py::test_nms_exactly_at_threshold_is_kept): the HIP filter keeps the
    box (strict >), and suppresses it with the threshold one float32 ulp lower -- same side of the boundary as the oracle."""
    from squeezedet_pytorch_amd import ops
    from test_oracle_golden import nms_boundary_case
    boxes, scores, half, below, above = nms_boundary_case()
    A = 70                                                  # more anchors than keep_top_k; the rest score below everything
    sc = np.full((1, A), 0.01, np.float32); sc[0, :2] = scores
    bx = np.zeros((1, A, 4), np.float32); bx[0, :2] = boxes
    bx[0, 2:, 0] = 100 + 10 * np.arange(A - 2); bx[0, 2:, 2] = bx[0, 2:, 0] + 5; bx[0, 2:, 3] = 5
    ids = np.zeros((1, A), np.int64)
    for thr, want in ((float(half), [0, 1]), (float(below), [0]), (float(above), [0, 1])):
        cnt, cls, s2, b2, idx = ops.filter_dense(torch.from_numpy(ids).cuda(), torch.from_numpy(sc).cuda(), torch.from_numpy(bx).cuda(),
                                                 3, 64, thr, 0.3)
        n = int(cnt[0])
        assert idx[0, :n].cpu().tolist() == want, (thr, idx[0, :n].cpu().tolist())
        d = oracle.filter_detections(ids[0], sc[0], bx[0], 64, thr, 0.3)
        assert list(d['anchor_idx']) == want


def test_detect_into_packed_result_buffer():
    """ops.det_buffers_packed: the five result tensors as views of ONE allocation (what bench.py's end-to-end leg copies back with
    a single D2H): the fused detect kernel writes the same results into them as into separately allocated tensors."""
    from squeezedet_pytorch_amd import ops
    cfg = sqd.make_cfg()
    pred = _decode_pred().cuda()
    anc = torch.from_numpy(cfg.anchors).float().cuda()
    plain = ops.detect(pred, anc, cfg.input_size, cfg.num_classes, cfg.keep_top_k, cfg.nms_thresh, cfg.score_thresh)
    bufs, flat = ops.det_buffers_packed(pred.shape[0], cfg.keep_top_k, pred.device, cfg.num_anchors)
    packed = ops.detect(pred, anc, cfg.input_size, cfg.num_classes, cfg.keep_top_k, cfg.nms_thresh, cfg.score_thresh, out=bufs)
    assert int(plain[0].sum()) > 0
    for a, b in zip(plain, packed):
        assert a.dtype == b.dtype and torch.equal(a, b)
    host = flat.cpu()                                            # one copy carries everything
    assert torch.equal(host[:4 * pred.shape[0]].view(torch.int32), plain[0].cpu()) and flat.numel() % 16 == 0
    assert all(t.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for t in bufs[:5])


@pytest.mark.parametrize("B,A,seed", [(20, 16848, 1), (3, 16848, 2), (2, 1000, 3), (5, 37, 4), (1, 24000, 5)])
def test_detect_split_scoring_equals_one_workgroup_per_image(B, A, seed):
    """The fused detect launch with its key workspace (the anchors of an image scored by eight workgroups, the image's last arriver
    selecting and suppressing; csrc/postproc.hip) == the same launch without (one workgroup per image), bit for bit, launch after
    launch (the arrival counters return to zero), for anchor counts that are / are not multiples of the split."""
    from squeezedet_pytorch_amd import ops
    rs = np.random.RandomState(seed)
    pred = torch.from_numpy((rs.standard_normal((B, A, 8)) * 1.5).astype(np.float32)).cuda()
    anchors = torch.from_numpy(np.abs(rs.standard_normal((A, 4)) * 50 + 100).astype(np.float32)).cuda()
    K = 64
    one = ops._det_buffers(B, K, pred.device) + (torch.zeros(4, device='cuda', dtype=torch.int32),)      # placeholder -> one workgroup per image
    ref = [t.clone() for t in ops.detect(pred, anchors, (384, 1248), 3, K, 0.4, 0.3, out=one)]
    bufs = ops._det_buffers(B, K, pred.device, A)
    assert bufs[5].numel() == ops.det_workspace_words(B, A)
    for it in range(3):
        for t in bufs[:5]:
            t.zero_()
        got = ops.detect(pred, anchors, (384, 1248), 3, K, 0.4, 0.3, out=bufs)
        torch.cuda.synchronize()
        assert int(bufs[5][-B:].abs().sum()) == 0, 'arrival counters must return to zero'
        cnt = got[0].cpu().numpy()
        assert np.array_equal(cnt, ref[0].cpu().numpy()) and (cnt > 0).all()
        for b in range(B):
            n = int(cnt[b])
            for g, r in zip(got[1:], ref[1:]):
                assert torch.equal(g[b, :n], r[b, :n]), (it, b)
