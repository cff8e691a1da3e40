"""The lane executor (``lanes.DetectStream``; reference driver: ``Detector.detect_dataset``, src/engine/detector.py:52-85): batches
in flight on captured per-lane steps must give, bit for bit, what ``Detector.detect_images`` / ``detect_device`` give one batch at
a time -- full batches (eager first use, capture, replay), the ragged last batch, mixed image sizes, images that outgrow the
staging slots, the ``forbid_resize`` branch, new weights under live graphs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _detector(input_size=(384, 1248), batch_size=3, **over):
    import squeezedet_pytorch_amd as sqd
    from squeezedet_pytorch_amd import synthetic
    from squeezedet_pytorch_amd.detector import Detector
    from squeezedet_pytorch_amd.model import SqueezeDet
    cfg = sqd.make_cfg(input_size=input_size, **over)
    cfg.batch_size = batch_size
    m = SqueezeDet(cfg)
    m.load_state_dict(synthetic.make_state_dict())
    return Detector(m, cfg), cfg


def _images(n, sizes, seed=3):
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        h, w = sizes[i % len(sizes)]
        base = rs.standard_normal((-(-h // 8), -(-w // 8), 3)) * 60 + 100
        out.append(np.clip(np.kron(base, np.ones((8, 8, 1))), 0, 255).astype(np.uint8)[:h, :w])
    return out


def _same(r, w):
    assert ('boxes' in r) == ('boxes' in w)
    if 'boxes' in r:
        for k in ('anchor_idx', 'boxes', 'scores', 'class_ids'):
            assert np.array_equal(r[k], w[k]), k
            assert r[k].dtype == w[k].dtype
    assert r['image_meta']['orig_size'].tolist() == w['image_meta']['orig_size'].tolist()
    for k in ('scales', 'padding', 'crops'):
        if k in w['image_meta']:
            assert np.array_equal(np.asarray(r['image_meta'][k]), np.asarray(w['image_meta'][k])), k


def test_detect_stream_equals_detect_images_bitwise_incl_ragged_batch():
    det, cfg = _detector(batch_size=3)
    sizes = [(375, 1242), (370, 1224), (374, 1238), (376, 1241)]
    images = _images(3 * 7 + 2, sizes)                                  # 7 full batches + a ragged one of 2
    batches = [images[i:i + 3] for i in range(0, len(images), 3)]
    got = list(det.detect_stream(batches))
    ex = det.stream()
    assert len(got) == len(batches) and [len(g) for g in got] == [len(b) for b in batches]
    assert not ex.degraded and ex.captures == 2 and ex.replayed_batches == len(batches) - 1 - 2     # 2 lanes: first use eager, then capture
    assert ex.eager_batches == 3                                        # two first uses + the ragged batch
    assert ex.pending() == 0
    ndet = 0
    for b, res in zip(batches, got):
        want = det.detect_images(b)
        for r, w in zip(res, want):
            _same(r, w)
            ndet += len(w.get('scores', ()))
    assert ndet > 20
    # a second pass over the cached executor replays from the first full batch on
    got2 = list(det.detect_stream(batches[:3]))
    assert ex.captures == 2
    for res, res0 in zip(got2, got[:3]):
        for r, w in zip(res, res0):
            _same(r, w)


def test_stream_device_path_equals_detect_device():
    from squeezedet_pytorch_amd import synthetic
    det, cfg = _detector(batch_size=4)
    xs = [synthetic.make_images(4, cfg.input_size, seed=s).cuda() for s in range(2)]
    want = []
    for x in xs:
        want.append(tuple(t.cpu().numpy() for t in det.detect_device(x)))
    ex = det.stream(lanes=2)
    order = [0, 1, 0, 1, 0, 1, 0, 1, 1, 0]                              # each lane sees both tensors: 4 keys, all captured
    for i in order:
        ex.submit_device(xs[i], tag=i)
    out = ex.drain()
    assert [r.tag for _t, r in out] == order and not ex.degraded and ex.replayed_batches > 0
    for _t, r in out:
        cnt, cls, sc, bx, idx = want[r.tag]
        assert np.array_equal(r.count, cnt)
        for b in range(4):
            n = int(cnt[b])
            assert np.array_equal(r.anchor_idx[b, :n], idx[b, :n]) and np.array_equal(r.boxes[b, :n], bx[b, :n])
            assert np.array_equal(r.scores[b, :n], sc[b, :n]) and np.array_equal(r.class_ids[b, :n], cls[b, :n])
    assert int(sum(w[0].sum() for w in want)) > 10


def test_stream_keeps_a_bounded_number_of_graphs_per_lane():
    """A caller that keeps handing ``submit_device`` new tensors: every tensor seen twice on a lane is captured, but a lane keeps at
    most ``max_graphs_per_lane`` captured steps (the oldest is dropped) -- and the results stay those of ``detect_device``."""
    from squeezedet_pytorch_amd import synthetic
    det, cfg = _detector(batch_size=2)
    ex = det.stream(lanes=1)
    ex.max_graphs_per_lane = 3
    xs = [synthetic.make_images(2, cfg.input_size, seed=10 + s).cuda() for s in range(5)]
    want = [tuple(t.cpu().numpy() for t in det.detect_device(x)) for x in xs]
    order = [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 0, 0, 4]

    def check(r):
        cnt, _cls, sc, bx, idx = want[r.tag]
        assert np.array_equal(r.count, cnt)
        for b in range(2):                                           # (rows >= count are padding)
            n = int(cnt[b])
            assert np.array_equal(r.anchor_idx[b, :n], idx[b, :n]) and np.array_equal(r.boxes[b, :n], bx[b, :n]) and np.array_equal(r.scores[b, :n], sc[b, :n])
    for i in order:
        ex.submit_device(xs[i], tag=i)
        if ex.pending() > 1:
            check(ex.fetch()[1])
    for _t, r in ex.drain():
        check(r)
    lane = ex._lanes[0]
    assert sum(1 for v in lane.graphs.values() if v != 'eager') <= 3 and ex.captures >= 5 and not ex.degraded


def test_stream_slots_grow_for_larger_images_and_three_lanes():
    det, cfg = _detector(batch_size=2)
    small = _images(4, [(375, 1242)], seed=5)
    big = _images(2, [(800, 2000), (600, 1300)], seed=6)                # 4.8 MB > the 1.44 MB first-guess slot
    batches = [small[:2], big, small[2:], big, small[:2], big]
    ex = det.stream(lanes=3)
    got = list(ex.run(batches))
    assert not ex.degraded
    for b, res in zip(batches, got):
        for r, w in zip(res, det.detect_images(b)):
            _same(r, w)


def test_stream_forbid_resize_branch_and_ids():
    det, cfg = _detector(batch_size=2, forbid_resize=True)
    images = _images(6, [(375, 1242), (370, 1224), (400, 1300)], seed=8)
    batches = [(images[i:i + 2], [f'id{i}', f'id{i + 1}']) for i in range(0, 6, 2)]
    got = list(det.detect_stream(batches))
    for (b, ids), res in zip(batches, got):
        want = det.detect_images(b, image_ids=ids)
        for r, w in zip(res, want):
            _same(r, w)
            assert r['image_meta']['image_id'] == w['image_meta']['image_id']


def test_stream_sees_new_weights():
    from squeezedet_pytorch_amd import synthetic
    det, cfg = _detector(batch_size=2)
    images = _images(2, [(375, 1242)], seed=9)
    ex = det.stream()
    first = list(ex.run([images] * 5))
    assert ex.captures == 2
    det.model.load_state_dict(synthetic.make_state_dict(seed=77))
    second = list(ex.run([images] * 5))
    want = det.detect_images(images)
    for res in second:
        for r, w in zip(res, want):
            _same(r, w)
    assert ex.captures == 4                                             # the old graphs were dropped, the step re-captured
    differs = any(('boxes' in a) != ('boxes' in b) or ('boxes' in a and (len(a['scores']) != len(b['scores']) or not np.array_equal(a['scores'], b['scores'])))
                  for a, b in zip(first[0], second[0]))
    assert differs


def test_stream_refuses_what_it_cannot_run():
    det, cfg = _detector(batch_size=2)
    ex = det.stream()
    st = ex.stage(2)
    assert st.put(0, np.zeros((10, 10, 3), np.uint8))
    assert not st.put(1, np.full((10, 10, 3), 0.25, np.float32))        # not uint8-representable
    with pytest.raises(ValueError):
        ex.submit(st)
    with pytest.raises(RuntimeError):
        ex.fetch()
    with pytest.raises(ValueError):
        ex.submit_device(torch.zeros(2, 3, 384, 1248))                  # a CPU tensor
    # integral floats are accepted (a dataset that hands out float pixels)
    st = ex.stage(1)
    assert st.put(0, np.full((12, 20, 3), 7.0, np.float32))
    ex.submit(st)
    (_t, r), = ex.drain()
    assert r.count.shape == (1,)


def test_balanced_winograd_launches_on_two_streams_do_not_share_slabs():
    """the balanced Winograd kernel's slab workspace / arrival counters were shared by every launch of a shape;
    two lanes launching it concurrently interleaved tickets.  Workspaces now belong to the launch stream."""
    from squeezedet_pytorch_amd import ops, tiles
    from squeezedet_pytorch_amd.plans import wino_sk_schedule
    torch.manual_seed(0)
    B, H, W, C, N = 8, 24, 78, 72, 768
    x = torch.randn(B, H, W, C, device='cuda')
    w = torch.randn(N, C, 3, 3, device='cuda') * 0.05
    bias = torch.randn(N, device='cuda')
    plan = ops.WinoPlan(w, bias, tiles.WINO_SK_CFG)
    ref = torch.empty(B, H, W, N, device='cuda')
    ops.conv_wino(x, 0, plan, ref, 0, relu=True)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ys = [[torch.zeros_like(ref) for _ in range(6)] for _ in range(2)]
    for k in range(6):
        for s, y in ((s1, ys[0][k]), (s2, ys[1][k])):
            with torch.cuda.stream(s):
                ops.conv_wino(x, 0, plan, y, 0, relu=True)
    torch.cuda.synchronize()
    for lane in ys:
        for y in lane:
            assert torch.equal(y, ref)
    sk = wino_sk_schedule(B * -(-H // 4) * -(-W // 16), N, C, x.device)
    assert len(sk._ws) >= 3                                             # default stream + the two lanes
    for ws, cnt in sk._ws.values():
        assert int(cnt.abs().sum()) == 0


def test_stream_picker_finds_separate_hardware_queues():
    """The executor's streams must not share a hardware queue (a copy stream aliasing a lane's queue serialises upload and
    compute); the probe itself: a stream aliases itself, the chosen ones do not alias each other."""
    from squeezedet_pytorch_amd import lanes
    dev = torch.device('cuda', torch.cuda.current_device())
    s = torch.cuda.Stream()
    assert lanes.streams_alias(s, s)
    chosen, distinct = lanes.pick_streams(dev, 4)
    assert len(chosen) == 4 and len({c.cuda_stream for c in chosen}) == 4
    if distinct:
        for i in range(4):
            for j in range(4):
                if i != j:
                    assert not (lanes.streams_alias(chosen[i], chosen[j]) and lanes.streams_alias(chosen[i], chosen[j]))
    det, cfg = _detector(batch_size=2)
    assert det.stream().queues_distinct in (True, False)
