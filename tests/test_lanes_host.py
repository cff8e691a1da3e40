"""CPU tier: the host-side logic of the lane executor (lanes.py; reference driver src/engine/detector.py:52-85) that needs no GPU -- the
staging buffer layout (header of byte offsets + sizes, fixed-size image slots), ``Staging.put`` / ``view`` (uint8 pass-through, integral
floats accepted, anything else refused, oversized images kept for the re-pack), the packed result layout that the device writes and
the host reads back, and the per-image ``image_meta`` of both pre-processing branches."""
import numpy as np
import pytest
import torch


class _Owner:
    """What Staging touches of a DetectStream: one host buffer (plain memory here, pinned on the GPU box)."""

    def __init__(self, cap, slot):
        from squeezedet_pytorch_amd import lanes
        hdr = lanes._header_bytes(cap)
        buf = np.zeros(hdr + cap * slot, np.uint8)
        self._host = [{'np': buf, 'hdr': hdr, 'slot': slot, 'cap': cap}]


def test_header_layout():
    from squeezedet_pytorch_amd import lanes
    assert lanes._header_bytes(1) == 256 and lanes._header_bytes(16) == 256 and lanes._header_bytes(17) == 512
    assert lanes._header_bytes(20) % 256 == 0 and lanes._header_bytes(20) >= 16 * 20


def test_staging_put_view_refuse_overflow():
    from squeezedet_pytorch_amd import lanes
    ow = _Owner(cap=4, slot=10 * 12 * 3)
    st = lanes.Staging(ow, 0, 3)
    rs = np.random.RandomState(0)
    im0 = rs.randint(0, 256, (10, 12, 3), dtype=np.uint8)
    assert st.put(0, im0)
    hb = ow._host[0]
    assert np.array_equal(hb['np'][hb['hdr']:hb['hdr'] + im0.size].reshape(10, 12, 3), im0)
    # integral floats are taken as their uint8 values, non-integral ones and wrong shapes are refused
    assert st.put(1, np.full((4, 5, 3), 7.0, np.float32))
    off1 = hb['hdr'] + hb['slot']
    assert bool((hb['np'][off1:off1 + 60] == 7).all()) and st.sizes[1].tolist() == [4, 5]
    assert not st.put(2, np.full((4, 5, 3), 0.5, np.float32)) and 2 in st.refused
    assert not lanes.Staging(ow, 0, 1).put(0, np.zeros((4, 5), np.uint8))
    assert not lanes.Staging(ow, 0, 1).put(0, np.zeros((0, 5, 3), np.uint8))
    # an image larger than its slot is kept by reference for the re-pack (submit grows the buffers), its size is recorded
    st2 = lanes.Staging(ow, 0, 2)
    big = rs.randint(0, 256, (11, 12, 3), dtype=np.uint8)
    assert st2.put(0, big) and st2.overflow[0] is big and st2.sizes[0].tolist() == [11, 12] and all(st2.filled[:1])
    # view: a writable window of the slot, None when the image would not fit
    st3 = lanes.Staging(ow, 0, 2)
    v = st3.view(1, 5, 6)
    assert v.shape == (5, 6, 3) and v.dtype == np.uint8
    v[:] = 9
    assert bool((hb['np'][off1:off1 + 90] == 9).all()) and st3.filled == [False, True]
    assert st3.view(0, 11, 12) is None and st3.view(0, 0, 3) is None


def test_packed_result_layout_roundtrip():
    """The five result tensors in ONE allocation: 16-byte aligned sections, and the host reads exactly what views of the flat buffer hold."""
    from squeezedet_pytorch_amd import ops
    for B, K in [(1, 64), (3, 64), (20, 64), (7, 5)]:
        secs, total = ops.det_packed_layout(B, K)
        assert total % 16 == 0 and all(o % 16 == 0 for o, *_ in secs)
        ends = [o + n * torch.empty(0, dtype=dt).element_size() for o, n, dt, _ in secs]
        assert all(e <= o2 for e, (o2, *_) in zip(ends[:-1], secs[1:])) and ends[-1] <= total
        assert [s[3] for s in secs] == [(B,), (B, K), (B, K), (B, K, 4), (B, K)]
        assert [s[2] for s in secs] == [torch.int32, torch.int64, torch.float32, torch.float32, torch.int32]


def test_batch_result_per_image_and_meta():
    from squeezedet_pytorch_amd import lanes
    cnt = np.array([2, 0, 1], np.int32)
    cls = np.arange(3 * 4, dtype=np.int64).reshape(3, 4)
    sc = np.linspace(0, 1, 12, dtype=np.float32).reshape(3, 4)
    bx = np.arange(3 * 4 * 4, dtype=np.float32).reshape(3, 4, 4)
    idx = np.arange(12, dtype=np.int32).reshape(3, 4)
    metas = [{'index': b, 'image_id': f'{b:06d}'} for b in range(3)]
    out = lanes.BatchResult((cnt, cls, sc, bx, idx), metas, tag='t').per_image()
    assert [('boxes' in r) for r in out] == [True, False, True]
    assert out[0]['boxes'].shape == (2, 4) and out[0]['anchor_idx'].dtype == np.int64 and out[2]['scores'].tolist() == [sc[2, 0]]
    assert out[1] == {'image_meta': metas[1]} and out[0]['image_meta'] is not metas[0]          # copies: the caller may edit them
    # image_meta of the two input branches (scales / padding + crops), without a device
    import types
    for forbid in (False, True):
        fake = types.SimpleNamespace(cfg=types.SimpleNamespace(input_size=(384, 1248)), forbid=forbid)
        m = lanes.DetectStream._image_meta(fake, np.array([[375, 1242], [400, 1300]], np.int32), ['a', 'b'])
        assert [x['image_id'] for x in m] == ['a', 'b'] and m[1]['orig_size'].tolist() == [400, 1300, 3]
        if forbid:
            assert m[0]['padding'].tolist() == [4, 5, 3, 3] and m[0]['crops'].tolist() == [0, 0, 0, 0]
            assert m[1]['crops'].tolist() == [8, 8, 26, 26] and 'scales' not in m[0]
        else:
            np.testing.assert_allclose(m[0]['scales'], [384 / 375, 1248 / 1242], rtol=1e-7)


def test_stream_needs_a_gpu():
    import squeezedet_pytorch_amd as sqd
    from squeezedet_pytorch_amd import lanes
    fake = type('D', (), {})()
    fake.cfg = sqd.make_cfg(device='cpu')
    with pytest.raises(RuntimeError):
        lanes.DetectStream(fake)
