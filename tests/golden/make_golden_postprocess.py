#!/usr/bin/env python
"""Golden vectors for ``boxes_postprocess`` (src/utils/boxes.py:138-168): the REFERENCE function, imported read-only from
/root/reference/src (a ``cv2`` module object must exist at import time; it is never called), run on seeded random boxes
and every combination of the ``image_meta`` keys it understands.  Build container only; output = data.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_postprocess.py
"""
import itertools
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, "/root/reference/src")
from utils.boxes import boxes_postprocess  # noqa: E402


def cases():
    rs = np.random.RandomState(7)
    for n, (sc, pad, crop, flip, dsize, drift) in enumerate(itertools.product((0, 1), repeat=6)):
        boxes = rs.uniform(0, 600, (5, 4)).astype(np.float32)
        boxes[:, 2:] += boxes[:, :2]
        meta = {'orig_size': np.array([375, 1242], np.int64)}
        if sc:
            meta['scales'] = rs.uniform(0.5, 2.0, 2).astype(np.float32)
        if pad:
            meta['padding'] = rs.randint(0, 40, 4).astype(np.int64)
        if crop:
            meta['crops'] = rs.randint(0, 40, 4).astype(np.int64)
        meta['flipped'] = np.array(bool(flip))
        if dsize:
            meta['drifted_size'] = np.array([360, 1200], np.int64)
        if drift:
            meta['drifts'] = rs.randint(-25, 25, 2).astype(np.int64)
        yield n, boxes, meta


def main():
    out = {}
    for n, boxes, meta in cases():
        out[f'in{n}'] = boxes
        for k, v in meta.items():
            out[f'meta{n}_{k}'] = np.asarray(v)
        m = {k: (bool(v) if k == 'flipped' else v) for k, v in meta.items()}
        out[f'out{n}'] = boxes_postprocess(boxes.copy(), m)
    out['n'] = np.array(n + 1)
    np.savez_compressed(os.path.join(HERE, 'boxes_postprocess.npz'), **out)
    print('wrote boxes_postprocess.npz:', n + 1, 'cases')


if __name__ == '__main__':
    main()
