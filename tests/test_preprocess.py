"""Input pipeline: oracle known-answer tests on CPU, GPU kernel vs oracle on the MI355X."""
import numpy as np
import pytest
import torch

import oracle


def test_resize_identity_and_ramp_known_answers():
    rs = np.random.RandomState(0)
    img = rs.uniform(-2, 2, (7, 9, 3)).astype(np.float32)
    assert np.array_equal(oracle.resize_linear_f32(img, (7, 9)), img)                     # same size: exact copy
    # 2x upsample of a horizontal ramp: half-pixel centres -> dst x maps to src (x+0.5)/2-0.5, clamped at the border
    ramp = np.tile(np.arange(4, dtype=np.float32)[None, :, None], (2, 1, 3))
    up = oracle.resize_linear_f32(ramp, (2, 8))[0, :, 0]
    np.testing.assert_allclose(up, [0, 0.25, 0.75, 1.25, 1.75, 2.25, 2.75, 3.0], atol=1e-6)
    # 2x downsample averages pixel pairs
    down = oracle.resize_linear_f32(ramp, (2, 2))[0, :, 0]
    np.testing.assert_allclose(down, [0.5, 2.5], atol=1e-6)
    # constant image stays constant under any resize
    c = np.full((5, 6, 3), 1.25, np.float32)
    assert np.allclose(oracle.resize_linear_f32(c, (11, 13)), 1.25)


@pytest.mark.parametrize("src,dst", [((375, 1242), (384, 1248)), ((370, 1224), (384, 1248)), ((50, 61), (64, 96)), ((123, 77), (64, 96)),
                                     ((800, 2000), (384, 1248)), ((9, 5), (17, 33))])
def test_resize_agrees_with_an_independent_half_pixel_bilinear(src, dst):
    """A second witness for the restated cv2.resize(INTER_LINEAR) (OpenCV itself is absent here: parity against cv2 stays unpinned):
    torch's ``F.interpolate(mode='bilinear', align_corners=False, antialias=False)`` implements the same published rule -- source
    coordinate (x + 0.5) * scale - 0.5, border clamp, no prefilter when shrinking -- independently of the oracle's code."""
    import torch.nn.functional as F
    rs = np.random.RandomState(3)
    img = rs.uniform(-2.5, 2.5, src + (3,)).astype(np.float32)
    # The witness runs in float64.  cv2 (and the oracle) round the source coordinate to float32 BEFORE splitting it into index and
    # weight, so at x ~ 1242 the weight is only good to one float32 ulp of the coordinate (1.2e-4 pixel); on this white-noise image
    # (neighbouring pixels differ by up to 5) that is up to ~6e-4 in the output.  A wrong rule (corner-aligned coordinates, a
    # missing half-pixel offset, a prefilter) would differ by O(1).
    want = F.interpolate(torch.from_numpy(img).double().permute(2, 0, 1)[None], size=dst, mode='bilinear', align_corners=False, antialias=False)
    want = want[0].permute(1, 2, 0).numpy()
    got = oracle.resize_linear_f32(img, dst)
    np.testing.assert_allclose(got, want, atol=6e-4, rtol=0)
    wrong = F.interpolate(torch.from_numpy(img).double().permute(2, 0, 1)[None], size=dst, mode='bilinear', align_corners=True)[0].permute(1, 2, 0).numpy()
    if src != dst:
        assert np.abs(got - wrong).max() > 0.05                           # (the comparison has teeth)


def test_preprocess_image_layout_and_scales():
    img = np.random.RandomState(1).randint(0, 256, (375, 1242, 3), dtype=np.uint8)
    x, scales = oracle.preprocess_image(img, (384, 1248))
    assert x.shape == (3, 384, 1248) and x.dtype == np.float32
    np.testing.assert_allclose(scales, [384 / 375., 1248 / 1242.], rtol=1e-7)
    # whitening statistics: channel c of a constant image
    const = np.zeros((10, 10, 3), np.uint8); const[:] = (93, 98, 95)
    y, _ = oracle.preprocess_image(const, (20, 20))
    exp = (np.array([93, 98, 95], np.float32) - oracle.KITTI_RGB_MEAN) / oracle.KITTI_RGB_STD
    np.testing.assert_allclose(y[:, 3, 4], exp, rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [[(375, 1242), (370, 1224)], [(384, 1248)], [(50, 61), (123, 77), (9, 5)], [(800, 2000)]])
def test_gpu_preprocess_vs_oracle(sizes):
    from squeezedet_pytorch_amd.preprocess import preprocess_batch
    rs = np.random.RandomState(2)
    images = [rs.randint(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    out, scales, meta = preprocess_batch(images, (384, 1248))
    assert tuple(out.shape) == (len(images), 3, 384, 1248)
    for b, im in enumerate(images):
        ref, sc = oracle.preprocess_image(im, (384, 1248))
        np.testing.assert_allclose(out[b].cpu().numpy(), ref, atol=2e-5, rtol=0)
        np.testing.assert_allclose(scales[b].cpu().numpy(), sc, rtol=1e-6)
        np.testing.assert_allclose(meta['scales'][b], sc, rtol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("sizes,target", [
    ([(33, 3001), (1, 1), (7, 1), (1, 9)], (384, 1248)),        # odd byte offsets, one-pixel sources, the packed buffer ends inside a dword
    ([(40, 3000), (3, 2731)], (64, 96)),                          # a row segment larger than the LDS staging buffer: direct path
    ([(40, 2728), (41, 2729)], (64, 96)),                         # ... and just inside it
    ([(375, 1242)] * 3, (384, 1248)),
    ([(97, 300), (13, 1023)], (61, 517)),                         # target not a multiple of the 4-row / 256-column workgroup tile
])
def test_gpu_preprocess_staged_rows_vs_oracle(sizes, target):
    """The resize kernel stages each workgroup's two source-row segments in LDS (aligned dwords; the tail of the packed buffer
    byte by byte) and whitens through a per-channel table: same values as the oracle's per-pixel arithmetic at every alignment."""
    from squeezedet_pytorch_amd.preprocess import preprocess_batch
    rs = np.random.RandomState(5)
    images = [rs.randint(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    out, scales, _meta = preprocess_batch(images, target)
    for b, im in enumerate(images):
        ref, sc = oracle.preprocess_image(im, target)
        np.testing.assert_allclose(out[b].cpu().numpy(), ref, atol=2e-5, rtol=0)
        np.testing.assert_allclose(scales[b].cpu().numpy(), sc, rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("sizes,target", [
    ([(1, 1), (400, 1300), (383, 1249), (5, 2000), (390, 7)], (384, 1248)),
    ([(61, 517), (60, 516), (62, 519), (3, 3)], (61, 517)),
])
def test_gpu_padcrop_staged_rows_bit_exact_vs_oracle(sizes, target):
    from squeezedet_pytorch_amd.preprocess import preprocess_batch
    rs = np.random.RandomState(6)
    images = [rs.randint(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    out, _shifts, meta = preprocess_batch(images, target, forbid_resize=True)
    for b, im in enumerate(images):
        ref, pad, crop = oracle.crop_or_pad_image(im, target, oracle.KITTI_RGB_MEAN, oracle.KITTI_RGB_STD)
        assert np.array_equal(out[b].cpu().numpy(), ref), b
        assert meta['padding'][b].tolist() == list(pad) and meta['crops'][b].tolist() == list(crop)


@pytest.mark.gpu
def test_detect_images_end_to_end():
    """uint8 images -> GPU preprocess -> backbone -> fused detect, vs the oracle fed with the oracle's own
    pre-processed tensor (same kept anchors away from near-ties; boxes in original-image coordinates)."""
    import squeezedet_pytorch_amd as sqd
    from squeezedet_pytorch_amd import synthetic
    from squeezedet_pytorch_amd.detector import Detector
    from squeezedet_pytorch_amd.model import SqueezeDet
    cfg = sqd.make_cfg()
    m = SqueezeDet(cfg); sd = synthetic.make_state_dict(); m.load_state_dict(sd)
    det = Detector(m, cfg)
    rs = np.random.RandomState(3)
    base = (rs.standard_normal((2, 48, 156, 3)) * 60 + 100)
    images = [np.clip(np.kron(base[0], np.ones((8, 8, 1))), 0, 255).astype(np.uint8)[:375, :1242],
              np.clip(np.kron(base[1], np.ones((8, 8, 1))), 0, 255).astype(np.uint8)[:370, :1224]]
    res = det.detect_images(images, image_ids=['a', 'b'])
    assert len(res) == 2
    for b, im in enumerate(images):
        x, sc = oracle.preprocess_image(im, cfg.input_size)
        with torch.no_grad():
            pred = oracle.backbone_forward(torch.from_numpy(x)[None], sd)
        ids, s, bx = oracle.inference_head(pred, cfg.anchors, cfg.input_size)
        d = oracle.filter_detections(ids[0].numpy(), s[0].numpy(), bx[0].numpy())
        r = res[b]
        assert r['image_meta']['image_id'] == ['a', 'b'][b]
        if d is None:
            assert 'boxes' not in r
            continue
        common = set(r['anchor_idx'].tolist()) & set(d['anchor_idx'].tolist())
        assert len(common) >= len(d['anchor_idx']) - 2
        ref_boxes = {int(i): bb for i, bb in zip(d['anchor_idx'], oracle.boxes_postprocess(d['boxes'], sc))}
        for i, bb in zip(r['anchor_idx'], r['boxes']):
            if int(i) in ref_boxes:
                np.testing.assert_allclose(bb, ref_boxes[int(i)], atol=2e-2)


class _FakeKitti:
    """The three members ``Detector.detect_dataset`` / ``DataWrapper`` use of the reference's dataset classes
    (src/datasets/kitti.py:47-55 ``load_image`` -> float32 RGB HWC + id; src/datasets/base.py:43-59 ``preprocess``)."""

    def __init__(self, images, input_size, integral=True):
        self.images, self.input_size, self.integral = images, input_size, integral

    def __len__(self):
        return len(self.images)

    def load_image(self, index):
        im = self.images[index].astype(np.float32)
        return (im if self.integral else im + 0.25), f'{index:06d}'

    def preprocess(self, image, image_meta, boxes=None):
        x = (image - oracle.KITTI_RGB_MEAN.reshape(1, 1, 3)) / oracle.KITTI_RGB_STD.reshape(1, 1, 3)
        scales = np.array([self.input_size[0] / image.shape[0], self.input_size[1] / image.shape[1]], dtype=np.float32)
        image_meta = dict(image_meta, scales=scales)
        return oracle.resize_linear_f32(x.astype(np.float32), self.input_size), image_meta, boxes


def test_data_wrapper_item_layout():
    from squeezedet_pytorch_amd.detector import DataWrapper
    rs = np.random.RandomState(5)
    ds = _FakeKitti([rs.randint(0, 256, (37, 53, 3), dtype=np.uint8), rs.randint(0, 256, (41, 60, 3), dtype=np.uint8)], (48, 64))
    w = DataWrapper(ds)
    assert len(w) == 2
    it = w[1]
    assert it['image'].shape == (3, 48, 64) and it['image'].dtype == np.float32
    m = it['image_meta']
    assert m['index'] == 1 and m['image_id'] == '000001' and m['orig_size'].tolist() == [41, 60, 3] and m['scales'].shape == (2,)


@pytest.mark.gpu
def test_detect_dataset_driver(capsys):
    """``Detector.detect_dataset`` (src/engine/detector.py:52-85): batches of raw images through the GPU input pipeline == calling
    ``detect_images`` batch by batch (ragged last batch, dataset indices in ``image_meta``), the reference's two kinds of log
    lines, the empty dataset; a dataset whose pixels are not uint8-representable takes the host route (``dataset.preprocess`` +
    ``detect``) and agrees with the GPU route up to the pre-processing difference of a quarter grey level."""
    import squeezedet_pytorch_amd as sqd
    from squeezedet_pytorch_amd import synthetic
    from squeezedet_pytorch_amd.detector import Detector
    from squeezedet_pytorch_amd.model import SqueezeDet
    cfg = sqd.make_cfg()
    cfg.batch_size, cfg.num_workers, cfg.print_interval = 2, 2, 1
    m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict())
    det = Detector(m, cfg)
    rs = np.random.RandomState(3)
    base = rs.standard_normal((5, 48, 156, 3)) * 60 + 100
    sizes = [(375, 1242), (370, 1224), (375, 1242), (374, 1238), (376, 1241)]
    images = [np.clip(np.kron(base[i], np.ones((8, 8, 1))), 0, 255).astype(np.uint8)[:h, :w] for i, (h, w) in enumerate(sizes)]
    ds = _FakeKitti(images, cfg.input_size)
    res = det.detect_dataset(ds)
    out = capsys.readouterr().out
    assert out.count('eval: [') == 3 and 'Elapsed' in out and 'frames/s' in out
    assert len(res) == 5 and [r['image_meta']['index'] for r in res] == [0, 1, 2, 3, 4]
    assert [r['image_meta']['image_id'] for r in res] == [f'{i:06d}' for i in range(5)]
    want = det.detect_images(images[0:2]) + det.detect_images(images[2:4]) + det.detect_images(images[4:5])
    assert sum('boxes' in r for r in want) >= 3
    for r, w in zip(res, want):
        assert ('boxes' in r) == ('boxes' in w)
        if 'boxes' in r:
            assert np.array_equal(r['anchor_idx'], w['anchor_idx']) and np.array_equal(r['boxes'], w['boxes']) and np.array_equal(r['scores'], w['scores'])
        assert r['image_meta']['orig_size'].tolist() == w['image_meta']['orig_size'].tolist()
    assert det.detect_dataset(_FakeKitti([], cfg.input_size)) == []
    # host route: pixels + 0.25 are not uint8-representable
    res_h = det.detect_dataset(_FakeKitti(images, cfg.input_size, integral=False))
    assert len(res_h) == 5 and [r['image_meta']['index'] for r in res_h] == [0, 1, 2, 3, 4]
    for r, w in zip(res_h, want):
        if 'boxes' in r and 'boxes' in w:
            common = set(r['anchor_idx'].tolist()) & set(w['anchor_idx'].tolist())
            assert len(common) >= min(len(r['anchor_idx']), len(w['anchor_idx'])) - 3
