#!/usr/bin/env python
"""Golden vectors for the GT encoder: the REFERENCE's ``compute_deltas``
(src/utils/boxes.py:84-135, imported read-only with the same ``cv2`` module stub as make_golden.py) on seeded
box sets -> tests/golden/gt_encode.npz.  Data only: inputs (float32 xyxy boxes, class ids), the reference's
anchor indices and deltas, and per box a flag telling whether the reference's pick was *uniquely determined*.

Why the flag: the reference walks ``np.argsort(-overlaps)`` (default introsort, unstable, SIMD-dispatched in
numpy >= 1.25), so among anchors with exactly equal overlap the winner is an accident of the sort implementation.
Exact ties are real (an anchor shape that fully contains a small box has the same IoU at every grid cell that
contains it).  The build breaks ties towards the lowest anchor index; index parity is asserted where the flag
is set, and "picked an anchor with the maximal free overlap" where it is not.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_gt.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
from make_golden import import_reference, ref_cfg  # noqa: E402


def box_sets():
    rs = np.random.RandomState(77)
    sets = []

    def clip(b):
        b = np.asarray(b, np.float64)
        b[:, [0, 2]] = np.clip(b[:, [0, 2]], 0, 1247); b[:, [1, 3]] = np.clip(b[:, [1, 3]], 0, 383)
        keep = (b[:, 2] - b[:, 0] > 1) & (b[:, 3] - b[:, 1] > 1)
        return b[keep].astype(np.float32)

    # KITTI-like: centres uniform, sizes log-uniform
    for n in (12, 25, 40):
        cx = rs.uniform(0, 1248, n); cy = rs.uniform(100, 384, n)
        w = np.exp(rs.uniform(np.log(8), np.log(400), n)); h = w * rs.uniform(0.4, 1.6, n)
        sets.append(clip(np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)))
    # crowd: near-identical boxes compete for the same anchors (walks down the overlap order)
    j = rs.uniform(-3, 3, (30, 4))
    sets.append(clip(np.array([600., 150., 760., 260.]) + j))
    # tiny boxes inside one grid cell: exact IoU ties between cells are likely
    cx = rs.uniform(50, 1200, 16); cy = rs.uniform(50, 330, 16)
    sets.append(clip(np.stack([cx - 4, cy - 5, cx + 4, cy + 5], 1)))
    # integer-aligned boxes (exact arithmetic -> exact ties)
    x0 = rs.randint(0, 70, 10) * 16.; y0 = rs.randint(0, 20, 10) * 16.
    sets.append(clip(np.stack([x0, y0, x0 + 12, y0 + 10], 1)))
    # single box
    sets.append(np.array([[100.5, 120.25, 300.75, 250.5]], np.float32))
    return sets


def main():
    _, _, ref_boxes, _ = import_reference()
    cfg = ref_cfg("squeezedet", (384, 1248))
    anchors = cfg.anchors
    axyxy = ref_boxes.xywh_to_xyxy(anchors)
    rs = np.random.RandomState(78)
    out = {}
    sets = box_sets()
    # far-away boxes: zero overlap with every anchor -> nearest-anchor fallback (boxes.py:115-121); not clipped
    far = np.array([[3000., 50., 3100., 120.], [3000., 50., 3100., 120.], [3010., 60., 3090., 130.],
                    [-900., -700., -800., -650.]], np.float32)
    sets.append(far)
    for s, bx in enumerate(sets):
        cls = rs.randint(0, 3, bx.shape[0])
        deltas, idx = ref_boxes.compute_deltas(bx.copy(), anchors)
        # uniqueness of every pick, replayed with the reference's own arithmetic and its own taken-set sequence
        taken = np.zeros(anchors.shape[0], bool)
        unique = np.zeros(bx.shape[0], bool)
        bxywh = ref_boxes.xyxy_to_xywh(bx)
        for i in range(bx.shape[0]):
            ov = ref_boxes.compute_overlaps(axyxy, bx[i])
            free = ~taken
            best = ov[free].max()
            if best > 0:
                unique[i] = np.count_nonzero(ov[free] == best) == 1
                assert ov[idx[i]] == best
            else:
                d = np.sum((bxywh[i] - anchors) ** 2, axis=1)
                unique[i] = np.count_nonzero(d[free] == d[free].min()) == 1
                assert d[idx[i]] == d[free].min()
            taken[idx[i]] = True
        out[f"boxes{s}"] = bx; out[f"cls{s}"] = cls.astype(np.int32)
        out[f"idx{s}"] = idx; out[f"deltas{s}"] = deltas; out[f"unique{s}"] = unique
        print(f"set {s}: {bx.shape[0]} boxes, {int(unique.sum())} uniquely determined")
    out["num_sets"] = np.array([len(sets)])
    np.savez_compressed(os.path.join(HERE, "gt_encode.npz"), **out)


if __name__ == "__main__":
    main()
