"""Host logic of the balanced (stream-K) Winograd launch, csrc/conv_wino_sk.hip: sqd_wino_sk_schedule must hand every (super-group,
slice, K chunk) stage of a 3x3 layer (reference: Fire expand3x3 src/model/squeezedet.py:14,20-22, ConvDet :73-75,83) to exactly one
workgroup, in runs of near-equal length, with consistent part numbering for the units it cuts.  No GPU needed."""
import numpy as np
import pytest

import squeezedet_pytorch_amd  # noqa: F401
from squeezedet_pytorch_amd import plans

SHAPES = [(600, 72, 768), (600, 384, 96), (600, 192, 48), (600, 256, 64), (600, 768, 72), (60, 72, 768), (7, 72, 16), (1, 8, 8),
          (600, 80, 768), (599, 48, 24), (13, 100, 40)]


@pytest.mark.parametrize('ngroups,N,C', SHAPES)
@pytest.mark.parametrize('G,minseg,ksplit', [(512, 2, 0), (512, 1, 0), (3, 2, 0), (1, 1, 0), (64, 3, 0), (512, 2, 4), (7, 2, 3), (512, 2, 1)])
def test_schedule_covers_every_stage_once(ngroups, N, C, G, minseg, ksplit):
    nch, nsl = C // 8, -(-N // 32)
    half = (N - 32 * (nsl - 1)) <= 16
    if half and nsl > 1 and G == 1:                      # two workgroup classes need two workgroups
        with pytest.raises(RuntimeError):
            plans.wino_sk_host_schedule(ngroups, N, C, G, minseg, 1000, ksplit)
        return
    seg_off, segs, nslabs = plans.wino_sk_host_schedule(ngroups, N, C, G, minseg, 1000, ksplit)
    nfull = nsl - 1 if half else nsl
    assert seg_off[0] == 0 and seg_off[G] == len(segs) and np.all(np.diff(seg_off) >= 0)
    units = {}
    for r in range(G):
        rows = segs[seg_off[r]:seg_off[r + 1]]
        assert len({int(x[4]) for x in rows}) <= 1, 'a workgroup runs one class'
        for t, n0, c0, c1, cls, npar, part, slab0 in rows:
            assert 0 <= c0 < c1 <= nch
            units.setdefault((int(cls), int(t), int(n0)), []).append((int(c0), int(c1), int(part), int(npar), int(slab0)))
    slabs = set()
    for (cls, t, n0), parts in units.items():
        parts.sort()
        assert parts[0][0] == 0 and parts[-1][1] == nch, 'unit not covered from its first to its last K chunk'
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:])), 'gap / overlap inside a unit'
        if len(parts) == 1:
            assert parts[0][2:] == (0, 1, -1)
        else:
            assert [p[2] for p in parts] == list(range(len(parts))) and all(p[3] == len(parts) for p in parts)
            assert len({p[4] for p in parts}) == 1
            s0 = parts[0][4]
            assert not (slabs & set(range(s0, s0 + len(parts)))), 'slab ranges of two units overlap'
            slabs |= set(range(s0, s0 + len(parts)))
            if ksplit == 0 and minseg > 1 and nch >= 2 * minseg:
                assert parts[0][1] - parts[0][0] >= minseg and parts[-1][1] - parts[-1][0] >= minseg, 'a cut closer than minseg to a unit edge'
        if cls == 0:
            assert n0 % 32 == 0 and n0 < 32 * nfull and t < -(-ngroups // 4)
        else:
            assert half and n0 == 32 * nfull and t < -(-ngroups // 8)
    assert slabs == set(range(nslabs))
    assert sum(1 for k in units if k[0] == 0) == -(-ngroups // 4) * nfull
    assert sum(1 for k in units if k[0] == 1) == (-(-ngroups // 8) if half else 0)


def test_headline_shapes_are_balanced():
    """bs=20 24x78: ConvDet 70-71 stages on every one of the 512 workgroups (450 whole units of 96 before), fire13/14 41-44."""
    for (N, C, lo, hi) in [(72, 768, 70, 71), (384, 96, 41, 44), (768, 72, 62, 66)]:
        seg_off, segs, _ = plans.wino_sk_host_schedule(600, N, C, 512, 2)
        st = [int(sum(x[3] - x[2] for x in segs[seg_off[r]:seg_off[r + 1]])) for r in range(512)]
        assert lo <= min(st) and max(st) <= hi, (N, C, min(st), max(st))
