"""The input pipeline's ``cfg.forbid_resize`` branch (reference src/datasets/base.py:51-54 -> ``whiten`` + ``crop_or_pad``,
src/utils/image.py:9-19,91-158; ``boxes_postprocess``' padding / crops terms, src/utils/boxes.py:149-155), pinned by
tests/golden/padcrop.npz, which tests/golden/make_golden_padcrop.py produced by running the REFERENCE's own functions:
the oracle restatement on the CPU, ``preprocess_kernel``'s pad/crop form on the GPU bit for bit, and the fused detect kernel's box
shift against the reference-postprocessed boxes.  Also the ``detect_dataset`` driver with 0 / 1 loader workers (a
one-worker pool deadlocked)."""
import os
import threading

import numpy as np
import pytest
import torch

import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'padcrop.npz')
ROWS = [0, 1, 5, 191, 192, 378, 382, 383]


def _weights(shape):
    c, h, w = shape
    return ((np.arange(c).reshape(c, 1, 1) * 0.37 + 1.0) * (np.arange(h).reshape(1, h, 1) * 0.011 + 1.0)
            * (np.arange(w).reshape(1, 1, w) * 0.0013 + 1.0)).astype(np.float64)


def _image(n, h, w):
    return np.random.RandomState(100 + n).randint(0, 256, size=(h, w, 3)).astype(np.uint8)


def _check_against_golden(g, n, chw, padding, crops):
    assert np.array_equal(np.asarray(padding).astype(np.int64), g[f'padding{n}'].astype(np.int64)), (n, padding, g[f'padding{n}'])
    assert np.array_equal(np.asarray(crops).astype(np.int64), g[f'crops{n}'].astype(np.int64)), (n, crops, g[f'crops{n}'])
    assert chw.dtype == np.float32 and chw.shape == (3, 384, 1248)
    assert np.array_equal(chw[:, ::11, ::13], g[f'sample{n}'])
    assert np.array_equal(chw[:, ROWS, :], g[f'rows{n}'])
    d = chw.astype(np.float64)
    assert np.array_equal(np.array([d.sum(), (d * _weights(chw.shape)).sum(), np.abs(d).sum()]), g[f'check{n}'])


def test_oracle_crop_or_pad_equals_reference_goldens():
    g = np.load(GOLD)
    for n, (h, w) in enumerate(g['sizes']):
        chw, padding, crops = oracle.crop_or_pad_image(_image(n, h, w), tuple(g['target']), g['mean'], g['std'])
        _check_against_golden(g, n, chw, padding, crops)
        assert np.array_equal(oracle.boxes_unpad_uncrop(g[f'boxes_in{n}'], padding, crops), g[f'boxes_out{n}'])
        from squeezedet_pytorch_amd.boxes import boxes_postprocess          # the host-side general form, same goldens
        assert np.array_equal(boxes_postprocess(g[f'boxes_in{n}'].copy(), {'padding': padding, 'crops': crops}), g[f'boxes_out{n}'])


@pytest.mark.gpu
def test_padcrop_kernel_bit_exact_vs_reference_goldens():
    from squeezedet_pytorch_amd.preprocess import preprocess_batch
    g = np.load(GOLD)
    images = [_image(n, h, w) for n, (h, w) in enumerate(g['sizes'])]
    out, shifts, meta = preprocess_batch(images, tuple(g['target']), rgb_mean=g['mean'], rgb_std=g['std'], forbid_resize=True)
    out, shifts = out.cpu().numpy(), shifts.cpu().numpy()
    assert 'scales' not in meta and meta['padding'].dtype == np.int16 and meta['crops'].dtype == np.int16
    for n, (h, w) in enumerate(g['sizes']):
        _check_against_golden(g, n, out[n], meta['padding'][n], meta['crops'][n])
        ref, _, _ = oracle.crop_or_pad_image(images[n], tuple(g['target']), g['mean'], g['std'])
        assert np.array_equal(out[n], ref), f'image {n}: not bit-identical to the oracle'
        assert shifts[n].tolist() == [float(g[f'crops{n}'][0]) - float(g[f'padding{n}'][0]), float(g[f'crops{n}'][2]) - float(g[f'padding{n}'][2])]


@pytest.mark.gpu
def test_detect_shift_equals_reference_boxes_postprocess():
    """The fused detect kernel with ``shifts`` == the same launch without, post-processed by the REFERENCE's boxes_postprocess with the
    recorded padding / crops (goldens: reference outputs on given boxes; here: on the kernel's own boxes via the pinned oracle)."""
    import squeezedet_pytorch_amd as sqd
    from squeezedet_pytorch_amd import ops, synthetic
    g = np.load(GOLD)
    cfg = sqd.make_cfg()
    B = int(g['n'])
    rs = np.random.RandomState(2)
    pred = torch.from_numpy(rs.standard_normal((B, cfg.num_anchors, 8)).astype(np.float32) * 1.5).cuda()
    anchors = torch.from_numpy(np.asarray(cfg.anchors, np.float32)).cuda()
    sh = np.stack([[float(g[f'crops{n}'][0]) - float(g[f'padding{n}'][0]), float(g[f'crops{n}'][2]) - float(g[f'padding{n}'][2])] for n in range(B)])
    c0, k0, s0, b0, i0 = (t.cpu().numpy() for t in ops.detect(pred, anchors, cfg.input_size, 3))
    c1, k1, s1, b1, i1 = (t.cpu().numpy() for t in ops.detect(pred, anchors, cfg.input_size, 3, shifts=torch.from_numpy(sh.astype(np.float32)).cuda()))
    assert np.array_equal(c0, c1) and np.array_equal(i0, i1) and np.array_equal(s0, s1) and c0.min() > 0
    for n in range(B):
        want = oracle.boxes_unpad_uncrop(b0[n, :c0[n]], g[f'padding{n}'], g[f'crops{n}'])
        assert np.array_equal(b1[n, :c1[n]], want), n


class _Set:
    def __init__(self, n):
        self.n = n
        self.rgb_mean = np.array([10., 20., 30.], np.float32).reshape(1, 1, 3)
        self.rgb_std = np.array([50., 60., 70.], np.float32).reshape(1, 1, 3)

    def __len__(self):
        return self.n

    def load_image(self, i):
        return np.full((6, 8, 3), i, np.float32), f'{i:06d}'


@pytest.mark.parametrize('workers', [0, 1, 3])
def test_detect_dataset_does_not_deadlock_with_few_workers(workers, capsys):
    """Loader threads: one future per image (the round-3 form nested a per-batch task inside the pool it waited on: one worker hung)."""
    import types
    from squeezedet_pytorch_amd.detector import Detector
    det = Detector.__new__(Detector)
    det.cfg = types.SimpleNamespace(batch_size=2, print_interval=10, num_workers=workers, device='cpu')
    seen = {}

    class _Staged:                                  # the host side of lanes.Staging: loader threads put their image into slot b
        def __init__(self, n):
            self.n, self.images = n, [None] * n

        def put(self, b, im):
            self.images[b] = np.asarray(im)
            return True

    class _Result:
        def __init__(self, st, ids, tag):
            self.tag = tag
            self._out = [{'image_meta': {'image_id': iid, 'pix': int(im[0, 0, 0]), 'index': b}} for b, (im, iid) in enumerate(zip(st.images, ids))]

        def per_image(self):
            return self._out

    class _FakeStream:                              # lanes.DetectStream's interface without a GPU: results come back one batch late
        _lanes = [None, None]

        def __init__(self, rgb_mean):
            seen['mean'] = rgb_mean
            self.q = []

        def pending(self):
            return len(self.q)

        def oldest_ready(self):
            return len(self.q) > 1

        def stage(self, n):
            return _Staged(n)

        def submit(self, st, image_ids=None, tag=None):
            assert all(im is not None for im in st.images)
            self.q.append(_Result(st, image_ids, tag))

        def discard(self, st):
            pass

        def fetch(self):
            return 0, self.q.pop(0)

        def drain(self):
            out, self.q = [(0, r) for r in self.q], []
            return out
    det.stream = lambda lanes=None, graph=True, rgb_mean=None, rgb_std=None: _FakeStream(rgb_mean)
    res = []
    t = threading.Thread(target=lambda: res.extend(det.detect_dataset(_Set(5))), daemon=True)
    t.start(); t.join(30)
    assert not t.is_alive(), f'detect_dataset hung with num_workers={workers}'
    assert [r['image_meta']['index'] for r in res] == [0, 1, 2, 3, 4] and [r['image_meta']['pix'] for r in res] == [0, 1, 2, 3, 4]
    assert seen['mean'] is not None and float(np.asarray(seen['mean']).reshape(-1)[1]) == 20.0, 'the dataset\'s own whitening statistics'
    # a config without num_workers takes the reference's default of 4
    del det.cfg.num_workers
    assert len(det.detect_dataset(_Set(3))) == 3


@pytest.mark.gpu
def test_detect_images_forbid_resize_end_to_end():
    """``Detector.detect_images`` with ``cfg.forbid_resize``: uint8 upload -> pad/crop kernel -> backbone -> fused detect with the
    un-pad / un-crop shift == the reference's route restated on the host (oracle whiten + crop_or_pad -> the same backbone -> detect ->
    ``boxes_postprocess`` padding / crops terms), bit for bit; ``detect`` on a host-built batch carrying padding / crops takes the
    same fused route."""
    import squeezedet_pytorch_amd as sqd
    from squeezedet_pytorch_amd import synthetic
    from squeezedet_pytorch_amd.detector import Detector
    from squeezedet_pytorch_amd.model import SqueezeDet
    cfg = sqd.make_cfg(forbid_resize=True)
    m = SqueezeDet(cfg); m.load_state_dict(synthetic.make_state_dict())
    det = Detector(m, cfg)
    rs = np.random.RandomState(3)
    sizes = [(375, 1242), (400, 1300), (300, 1400), (384, 1248)]
    base = rs.standard_normal((len(sizes), 64, 176, 3)) * 60 + 100
    images = [np.clip(np.kron(base[i], np.ones((8, 8, 1))), 0, 255).astype(np.uint8)[:h, :w] for i, (h, w) in enumerate(sizes)]
    res = det.detect_images(images)
    host = [oracle.crop_or_pad_image(im, cfg.input_size) for im in images]
    x = torch.from_numpy(np.stack([h[0] for h in host])).cuda()
    cnt, cls, sc, bx, idx = (t.cpu().numpy() for t in det.detect_device(x))
    assert sum('boxes' in r for r in res) >= 2
    for b, r in enumerate(res):
        n = int(cnt[b])
        assert ('boxes' in r) == (n > 0)
        assert np.array_equal(r['image_meta']['padding'], host[b][1]) and np.array_equal(r['image_meta']['crops'], host[b][2])
        assert 'scales' not in r['image_meta']
        if n:
            assert np.array_equal(r['anchor_idx'], idx[b, :n]) and np.array_equal(r['scores'], sc[b, :n])
            assert np.array_equal(r['boxes'], oracle.boxes_unpad_uncrop(bx[b, :n], host[b][1], host[b][2]))
    batch = {'image': x, 'image_meta': {'padding': np.stack([h[1] for h in host]), 'crops': np.stack([h[2] for h in host]),
                                        'index': np.arange(len(images)), 'image_id': [str(i) for i in range(len(images))]}}
    res2 = det.detect(batch)
    for r, r2 in zip(res, res2):
        assert ('boxes' in r) == ('boxes' in r2)
        if 'boxes' in r:
            assert np.array_equal(r['boxes'], r2['boxes']) and np.array_equal(r['anchor_idx'], r2['anchor_idx'])
