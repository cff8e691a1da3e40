"""GT encoding: oracle vs the reference's golden vectors on CPU; HIP encoder vs oracle on the MI355X.

Index parity with the reference is defined where the reference's pick is uniquely determined (no other free anchor
with exactly the same overlap / distance); tests/golden/make_golden_gt.py stores that flag per box.  Once a tie has
been resolved differently the taken-sets differ, so golden indices are compared on the prefix up to the first
non-unique box of each set; past it, every pick is checked to be a maximal-free-overlap anchor (true for any tie rule).
"""
import os

import numpy as np
import pytest
import torch

import oracle
import squeezedet_pytorch_amd as sqd
from squeezedet_pytorch_amd import boxes as host_boxes


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "gt_encode.npz"))


@pytest.fixture(scope="module")
def anchors():
    return sqd.make_cfg(arch="squeezedet", device="cpu").anchors


def _sets(gold):
    for s in range(int(gold["num_sets"][0])):
        yield s, gold[f"boxes{s}"], gold[f"cls{s}"], gold[f"idx{s}"], gold[f"deltas{s}"], gold[f"unique{s}"]


def _prefix(unique):
    nz = np.nonzero(~unique)[0]
    return int(nz[0]) if nz.size else unique.shape[0]


def _overlaps(anchors, box):
    ax = np.stack([anchors[:, 0] - 0.5 * (anchors[:, 2] - 1), anchors[:, 1] - 0.5 * (anchors[:, 3] - 1),
                   anchors[:, 0] + 0.5 * (anchors[:, 2] - 1), anchors[:, 1] + 0.5 * (anchors[:, 3] - 1)], 1)
    lr = np.maximum(np.minimum(ax[:, 2], box[2]) - np.maximum(ax[:, 0], box[0]), 0)
    tb = np.maximum(np.minimum(ax[:, 3], box[3]) - np.maximum(ax[:, 1], box[1]), 0)
    inter = lr * tb
    union = (ax[:, 2] - ax[:, 0]) * (ax[:, 3] - ax[:, 1]) + (box[2] - box[0]) * (box[3] - box[1]) - inter
    return inter / (union + 1e-10)


def _check_valid_assignment(anchors, bx, idx):
    """every pick is a free anchor of maximal overlap (or, with no overlap left, of minimal distance)"""
    taken = np.zeros(anchors.shape[0], bool)
    bxywh = np.stack([(bx[:, 0] + bx[:, 2]) / 2., (bx[:, 1] + bx[:, 3]) / 2., bx[:, 2] - bx[:, 0] + 1., bx[:, 3] - bx[:, 1] + 1.], 1)
    for i in range(bx.shape[0]):
        ov = _overlaps(anchors, bx[i])
        free = ~taken
        assert free[idx[i]], "anchor assigned twice"
        best = ov[free].max()
        if best > 0:
            assert ov[idx[i]] == best
        else:
            d = np.sum((bxywh[i] - anchors) ** 2, axis=1)
            assert d[idx[i]] == d[free].min()
        taken[idx[i]] = True


@pytest.mark.parametrize("ties", ["argsort", "lowest"])
def test_oracle_vs_reference_golden(gold, anchors, ties):
    for s, bx, cls, idx_ref, deltas_ref, unique in _sets(gold):
        deltas, idx = oracle.compute_deltas(bx, anchors, ties=ties)
        p = _prefix(unique)
        assert np.array_equal(idx[:p], idx_ref[:p]), f"set {s}"
        assert np.array_equal(deltas[:p], deltas_ref[:p]), f"set {s}"
        _check_valid_assignment(anchors, bx, idx)
        same = idx == idx_ref                      # wherever the same anchor was picked the deltas are the reference's bits
        assert np.array_equal(deltas[same], deltas_ref[same])


def test_host_compute_deltas_vs_reference_golden(gold, anchors):
    for s, bx, cls, idx_ref, deltas_ref, unique in _sets(gold):
        deltas, idx = host_boxes.compute_deltas(bx, anchors)
        p = _prefix(unique)
        assert np.array_equal(idx[:p], idx_ref[:p]) and np.array_equal(deltas[:p], deltas_ref[:p])
        _check_valid_assignment(anchors, bx, idx)


def test_golden_has_fully_determined_sets(gold):
    full = [s for s, _, _, _, _, u in _sets(gold) if u.all()]
    assert len(full) >= 2                          # incl. the zero-overlap (nearest-anchor fallback) set
    assert sum(int(u.sum()) for _, _, _, _, _, u in _sets(gold)) >= 80


def test_pack_annotations_checks():
    from squeezedet_pytorch_amd.annotations import pack_annotations
    b, c, o = pack_annotations([[0, 2], [], [1]], [np.array([[1, 2, 30, 40], [5, 6, 70, 80]], np.float32),
                                                  np.zeros((0, 4), np.float32), np.array([[0, 0, 9, 9]], np.float32)])
    assert b.shape == (3, 4) and c.tolist() == [0, 2, 1] and o.tolist() == [0, 2, 2, 3]
    with pytest.raises(AssertionError):
        pack_annotations([[0]], [np.array([[10, 2, 5, 40]], np.float32)])          # x1 >= x2 (boxes.py:14)
    with pytest.raises(ValueError):
        pack_annotations([[0, 1]], [np.array([[1, 2, 5, 40]], np.float32)])


# ------------------------------------------------------------------------------------------------------ GPU
def _gpu_encode(cls_list, box_list, anchors, num_classes=3):
    from squeezedet_pytorch_amd.annotations import encode_annotations
    gt, idx, deltas, offs = encode_annotations(cls_list, box_list, anchors, num_classes, device="cuda", return_sparse=True)
    torch.cuda.synchronize()
    return gt.cpu().numpy(), idx.cpu().numpy(), deltas.cpu().numpy(), offs.cpu().numpy()


def _ulp_close(a, b, ulps=1):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return np.all(np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)) <= ulps)


@pytest.mark.gpu
def test_gpu_encoder_vs_oracle_and_golden(gold, anchors):
    sets = list(_sets(gold))
    gt, idx, deltas, offs = _gpu_encode([s[2] for s in sets], [s[1] for s in sets], anchors)
    assert gt.shape == (len(sets), anchors.shape[0], 12)
    for n, (s, bx, cls, idx_ref, deltas_ref, unique) in enumerate(sets):
        lo, hi = offs[n], offs[n + 1]
        d_or, i_or = oracle.compute_deltas(bx, anchors, ties="lowest")
        assert np.array_equal(idx[lo:hi], i_or), f"set {s}: anchor indices differ from the oracle"        # bit-exact
        # dx, dy: float64 divide rounded to float32 -> exact; dw, dh go through log(): device libm vs numpy, 1 ulp
        assert np.array_equal(deltas[lo:hi, :2], d_or[:, :2])
        assert _ulp_close(deltas[lo:hi, 2:], d_or[:, 2:], 1)
        p = _prefix(unique)
        assert np.array_equal(idx[lo:lo + p], idx_ref[:p]), f"set {s}: differs from the reference where it is determined"
        assert _ulp_close(deltas[lo:lo + p], deltas_ref[:p], 1)
        # dense tensor against the oracle's prepare_annotations
        g_or = oracle.encode_gt(cls, bx, anchors, 3, ties="lowest")
        assert np.array_equal(gt[n][:, :5], g_or[:, :5]) and np.array_equal(gt[n][:, 9:], g_or[:, 9:])
        assert np.array_equal(gt[n][:, 5:7], g_or[:, 5:7]) and _ulp_close(gt[n][:, 7:9], g_or[:, 7:9], 1)
        assert int(gt[n][:, 0].sum()) == bx.shape[0]


@pytest.mark.gpu
def test_gpu_encoder_ragged_empty_and_crowded(anchors):
    rs = np.random.RandomState(3)

    def rand_boxes(n):
        cx = rs.uniform(0, 1248, n); cy = rs.uniform(0, 384, n)
        w = np.exp(rs.uniform(np.log(4), np.log(500), n)); h = np.exp(rs.uniform(np.log(4), np.log(300), n))
        b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
        b[:, [0, 2]] = np.clip(b[:, [0, 2]], 0, 1247); b[:, [1, 3]] = np.clip(b[:, [1, 3]], 0, 383)
        return b[(b[:, 2] - b[:, 0] > 0.5) & (b[:, 3] - b[:, 1] > 0.5)].astype(np.float32)

    same = np.tile(np.array([[400., 100., 520., 190.]], np.float32), (300, 1))     # 300 identical boxes: deep into the order
    box_list = [rand_boxes(200), np.zeros((0, 4), np.float32), rand_boxes(1), same, rand_boxes(37)]
    cls_list = [rs.randint(0, 3, b.shape[0]) for b in box_list]
    gt, idx, deltas, offs = _gpu_encode(cls_list, box_list, anchors)
    assert not gt[1].any()                                                          # image without boxes: all-zero gt
    for n, (bx, cls) in enumerate(zip(box_list, cls_list)):
        if bx.shape[0] == 0:
            continue
        lo, hi = offs[n], offs[n + 1]
        d_or, i_or = oracle.compute_deltas(bx, anchors, ties="lowest")
        assert np.array_equal(idx[lo:hi], i_or)
        assert len(set(idx[lo:hi].tolist())) == bx.shape[0]                         # distinct anchors
        assert _ulp_close(deltas[lo:hi], d_or, 1)
        g_or = oracle.encode_gt(cls, bx, anchors, 3, ties="lowest")
        assert np.array_equal(gt[n][:, :7], g_or[:, :7]) and np.array_equal(gt[n][:, 9:], g_or[:, 9:])


@pytest.mark.gpu
def test_gpu_encoder_more_boxes_than_anchors_and_other_shapes():
    from squeezedet_pytorch_amd import ops
    from squeezedet_pytorch_amd.annotations import anchors_f64_on, encode_annotations
    cfg = sqd.make_cfg(arch="squeezedet", input_size=(32, 48), device="cpu")       # 2x3 grid x 9 = 54 anchors
    A = cfg.anchors.shape[0]
    rs = np.random.RandomState(9)
    n = A + 5
    x1 = rs.uniform(0, 30, n); y1 = rs.uniform(0, 20, n)
    bx = np.stack([x1, y1, x1 + rs.uniform(2, 15, n), y1 + rs.uniform(2, 10, n)], 1).astype(np.float32)
    cls = rs.randint(0, 5, n).astype(np.int32)
    with pytest.raises(IndexError):
        encode_annotations([cls], [bx], cfg.anchors, 5, device="cuda")
    # through the C-ABI: the first A boxes get distinct anchors, the rest are reported unassigned (= A, like the reference's
    # num_anchors sentinel) and leave gt untouched
    dev = torch.device("cuda")
    gt, idx, deltas = ops.encode_gt(torch.from_numpy(bx).to(dev), torch.from_numpy(cls).to(dev),
                                    torch.tensor([0, n], dtype=torch.int32, device=dev), anchors_f64_on(cfg.anchors, dev), 5)
    idx = idx.cpu().numpy(); gt = gt.cpu().numpy()
    d_or, i_or = oracle.compute_deltas(bx, cfg.anchors, ties="lowest")
    assert np.array_equal(idx, i_or)
    assert sorted(idx[:A].tolist()) == list(range(A)) and np.all(idx[A:] == A)
    assert gt.shape == (1, A, 14) and int(gt[0, :, 0].sum()) == A
    # argument validation
    with pytest.raises(ValueError):
        ops.encode_gt(torch.from_numpy(bx).to(dev), torch.from_numpy(cls).to(dev), torch.tensor([0, n], dtype=torch.int32, device=dev),
                      torch.from_numpy(cfg.anchors.astype(np.float32)).to(dev), 5)


@pytest.mark.gpu
def test_gpu_encoder_parallel_first_choice_equals_serial(gold, anchors):
    """The parallel first-choice pass (+ serial conflict resolution) and the all-serial path give identical tensors, on the
    golden sets (ties, crowds, fallbacks) and on a heavily conflicting batch."""
    from squeezedet_pytorch_amd import ops
    from squeezedet_pytorch_amd.annotations import anchors_f64_on, pack_annotations
    sets = list(_sets(gold))
    same = np.tile(np.array([[400., 100., 520., 190.]], np.float32), (120, 1))
    box_list = [s[1] for s in sets] + [same]
    cls_list = [s[2] for s in sets] + [np.zeros(120, np.int32)]
    boxes, cls, offs = pack_annotations(cls_list, box_list)
    dev = torch.device("cuda")
    d = [torch.from_numpy(v).to(dev) for v in (boxes, cls, offs)]
    a64 = anchors_f64_on(anchors, dev)
    gt_p, idx_p, del_p = ops.encode_gt(d[0], d[1], d[2], a64, 3, parallel=True)
    gt_s, idx_s, del_s = ops.encode_gt(d[0], d[1], d[2], a64, 3, parallel=False)
    assert torch.equal(idx_p, idx_s) and torch.equal(del_p, del_s) and torch.equal(gt_p, gt_s)


@pytest.mark.gpu
def test_gpu_encoded_gt_feeds_the_loss(anchors):
    """loss on the device-encoded gt == loss on the host-encoded gt (same picks wherever ties do not interfere)."""
    from squeezedet_pytorch_amd import ops
    cfg = sqd.make_cfg(arch="squeezedet", device="cuda")
    rs = np.random.RandomState(5)
    box_list, cls_list = [], []
    for b in range(4):
        n = 3 + b
        x1 = rs.uniform(0, 1100, n); y1 = rs.uniform(0, 300, n)
        box_list.append(np.stack([x1, y1, np.minimum(x1 + rs.uniform(20, 300, n), 1247),
                                  np.minimum(y1 + rs.uniform(20, 150, n), 383)], 1).astype(np.float32))
        cls_list.append(rs.randint(0, 3, n))
    gt_dev, _, _, _ = _gpu_encode(cls_list, box_list, anchors)
    gt_host = np.stack([oracle.encode_gt(c, b, anchors, 3, ties="lowest") for c, b in zip(cls_list, box_list)])
    pred = torch.from_numpy(np.random.RandomState(1).randn(4, anchors.shape[0], 8).astype(np.float32)).cuda()
    a32 = torch.from_numpy(anchors.astype(np.float32)).cuda()
    w = (1.0, 3.75, 100.0, 6.0)
    l_dev, _ = ops.loss_fwd(pred, torch.from_numpy(gt_dev).cuda(), a32, cfg.input_size, 3, w)
    l_host, _ = ops.loss_fwd(pred, torch.from_numpy(gt_host).cuda(), a32, cfg.input_size, 3, w)
    torch.testing.assert_close(l_dev, l_host, rtol=1e-6, atol=1e-6)
