#!/usr/bin/env python
"""Golden vectors for the input pipeline's ``cfg.forbid_resize`` branch (src/datasets/base.py:51-54): the REFERENCE functions
``whiten`` and ``crop_or_pad`` (src/utils/image.py:9-19, 91-158) and ``boxes_postprocess`` (src/utils/boxes.py:138-168), imported
read-only from /root/reference/src (a ``cv2`` module object must exist at import time; it is never called by these functions), run on
seeded random uint8 images cast to float32 exactly as KITTI.load_image does (src/datasets/kitti.py:52).  Build container only;
output = data: per case the image size + seed (the pixels are regenerated from the seed), the padding / crops the reference recorded,
position-weighted float64 checksums and a strided sample of the (3, 384, 1248) result, and reference-postprocessed boxes.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_padcrop.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, "/root/reference/src")
from utils.image import whiten, crop_or_pad  # noqa: E402
from utils.boxes import boxes_postprocess  # noqa: E402

TARGET = (384, 1248)
MEAN = np.array([93.877, 98.801, 95.923], dtype=np.float32).reshape(1, 1, 3)      # src/datasets/kitti.py:17-18
STD = np.array([78.782, 80.130, 81.200], dtype=np.float32).reshape(1, 1, 3)
# KITTI's two common sizes (both axes padded), one larger than the target on both axes (cropped), mixed cases, odd differences,
# exactly the target, a tiny image
SIZES = [(375, 1242), (370, 1224), (400, 1300), (300, 1400), (500, 1000), (384, 1248), (383, 1249), (7, 9)]


def image_of(seed, h, w):
    return np.random.RandomState(seed).randint(0, 256, size=(h, w, 3)).astype(np.uint8)


def weights(shape):
    """Position-dependent weights of the checksum (a moved, mirrored or shifted image changes it)."""
    c, h, w = shape
    return ((np.arange(c).reshape(c, 1, 1) * 0.37 + 1.0) * (np.arange(h).reshape(1, h, 1) * 0.011 + 1.0)
            * (np.arange(w).reshape(1, 1, w) * 0.0013 + 1.0)).astype(np.float64)


def main():
    out = {'sizes': np.array(SIZES, np.int32), 'target': np.array(TARGET, np.int32), 'mean': MEAN.reshape(3), 'std': STD.reshape(3)}
    rs = np.random.RandomState(11)
    for n, (h, w) in enumerate(SIZES):
        img = image_of(100 + n, h, w).astype(np.float32)                   # KITTI.load_image: imread(...).astype(np.float32)
        meta = {'orig_size': np.array(img.shape, dtype=np.int32)}
        x, meta = whiten(img, meta, mean=MEAN, std=STD)
        x, meta, _ = crop_or_pad(x, meta, TARGET)
        assert x.shape == (TARGET[0], TARGET[1], 3) and x.dtype == np.float32, (x.shape, x.dtype)
        chw = np.ascontiguousarray(x.transpose(2, 0, 1))
        out[f'padding{n}'] = np.asarray(meta['padding']); out[f'crops{n}'] = np.asarray(meta['crops'])
        wts = weights(chw.shape)
        out[f'check{n}'] = np.array([chw.astype(np.float64).sum(), (chw.astype(np.float64) * wts).sum(), np.abs(chw.astype(np.float64)).sum()])
        out[f'sample{n}'] = chw[:, ::11, ::13].copy()
        out[f'rows{n}'] = chw[:, [0, 1, 5, 191, 192, 378, 382, 383], :].copy()    # whole rows through both borders
        boxes = rs.uniform(0, 380, (6, 4)).astype(np.float32)
        boxes[:, 2:] += boxes[:, :2]
        out[f'boxes_in{n}'] = boxes
        out[f'boxes_out{n}'] = boxes_postprocess(boxes.copy(), {k: meta[k] for k in ('padding', 'crops')})
    out['n'] = np.array(len(SIZES))
    np.savez_compressed(os.path.join(HERE, 'padcrop.npz'), **out)
    print('wrote padcrop.npz:', len(SIZES), 'cases', os.path.getsize(os.path.join(HERE, 'padcrop.npz')), 'bytes')


if __name__ == '__main__':
    main()
