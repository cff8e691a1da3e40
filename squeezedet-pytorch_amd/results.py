"""Detection results on disk and KITTI AP.

``save_results`` writes the reference's text format (``KITTI.save_results`` src/datasets/kitti.py:78-97: one file per
image under ``<results_dir>/data/<image_id>.txt``, lines ``'{cls} -1 -1 0 x1 y1 x2 y2 0 0 0 0 0 0 0 score'`` with 2 / 3
decimals, an empty file for an image without detections).  ``evaluate`` returns the same dict ``KITTI.evaluate``
(:99-124) builds from the ``stats_<class>_ap.txt`` files of the C++ evaluator -- here the AP comes from the native
restatement ``sqd_kitti_ap`` (csrc/kitti_eval.hip), and the same ``stats_*`` files are written for downstream tools.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _native as nat

KITTI_CLASS_NAMES = ('Car', 'Pedestrian', 'Cyclist')          # src/datasets/kitti.py:16
_KINDS = {'car': 0, 'pedestrian': 1, 'cyclist': 2, 'van': 3, 'person_sitting': 4, 'dontcare': 5}


def save_results(results, results_dir, class_names=KITTI_CLASS_NAMES):
    txt_dir = os.path.join(results_dir, 'data')
    os.makedirs(txt_dir, exist_ok=True)
    for res in results:
        txt_path = os.path.join(txt_dir, res['image_meta']['image_id'] + '.txt')
        with open(txt_path, 'w') as fp:
            if 'class_ids' not in res:
                continue
            for i in range(len(res['class_ids'])):
                name = class_names[int(res['class_ids'][i])].lower()
                fp.write('{} -1 -1 0 {:.2f} {:.2f} {:.2f} {:.2f} 0 0 0 0 0 0 0 {:.3f}\n'.format(
                    name, *[float(v) for v in res['boxes'][i, :]], float(res['scores'][i])))


def _records(path, nfields):
    """Whitespace-separated records of ``nfields`` tokens, read across line breaks like the evaluator's fscanf."""
    with open(path) as f:
        tok = f.read().split()
    return [tok[i:i + nfields] for i in range(0, len(tok) - nfields + 1, nfields)]


def read_label_file(path):
    """KITTI label_2 file -> (kind codes [n], boxes [n,4] f64, truncation [n], occlusion [n])."""
    rec = _records(path, 15)
    kind = np.array([_KINDS.get(r[0].lower(), 6) for r in rec], dtype=np.int32)
    box = np.array([[float(v) for v in r[4:8]] for r in rec], dtype=np.float64).reshape(-1, 4)
    trunc = np.array([float(r[1]) for r in rec], dtype=np.float64)
    occ = np.array([int(r[2]) for r in rec], dtype=np.int32)
    return kind, box, trunc, occ


def read_result_file(path):
    """Results file -> (class 0..2 or -1 [n], boxes [n,4] f64, scores [n])."""
    rec = _records(path, 16)
    cls = np.array([c if (c := _KINDS.get(r[0].lower(), -1)) in (0, 1, 2) else -1 for r in rec], dtype=np.int32)
    box = np.array([[float(v) for v in r[4:8]] for r in rec], dtype=np.float64).reshape(-1, 4)
    score = np.array([float(r[15]) for r in rec], dtype=np.float64)
    return cls, box, score


def _cat(parts, dtype, width=None):
    parts = [np.asarray(p, dtype=dtype) for p in parts]
    if not parts:
        return np.zeros((0,) if width is None else (0, width), dtype=dtype)
    out = np.concatenate(parts, 0)
    return np.ascontiguousarray(out)


def kitti_ap(gts, dets):
    """gts: list (per image) of (kind, box, trunc, occ); dets: list of (cls, box, score).
    -> (ap [3,3] class x (easy, moderate, hard), precision [3,3,41], evaluated [3] bool)."""
    n = len(gts)
    if len(dets) != n:
        raise ValueError('kitti_ap: need one ground-truth and one detection entry per image')
    gt_off = np.zeros(n + 1, np.int32); det_off = np.zeros(n + 1, np.int32)
    for i in range(n):
        gt_off[i + 1] = gt_off[i] + len(gts[i][0]); det_off[i + 1] = det_off[i] + len(dets[i][0])
    gk = _cat([g[0] for g in gts], np.int32); gb = _cat([np.reshape(g[1], (-1, 4)) for g in gts], np.float64, 4)
    gtr = _cat([g[2] for g in gts], np.float64); goc = _cat([g[3] for g in gts], np.int32)
    dc = _cat([d[0] for d in dets], np.int32); db = _cat([np.reshape(d[1], (-1, 4)) for d in dets], np.float64, 4)
    ds = _cat([d[2] for d in dets], np.float64)
    ap = np.zeros((3, 3), np.float64); prec = np.zeros((3, 3, 41), np.float64); ev = np.zeros(3, np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)           # noqa: E731
    rc = nat.lib().sqd_kitti_ap(n, p(gt_off), p(gk), p(gb), p(gtr), p(goc), p(det_off), p(dc), p(db), p(ds), p(ap), p(prec), p(ev))
    nat.check(rc, 'sqd_kitti_ap')
    return ap, prec, ev.astype(bool)


def evaluate(results_dir, label_dir, sample_set_path, class_names=KITTI_CLASS_NAMES, write_stats=True):
    """``KITTI.evaluate`` (src/datasets/kitti.py:99-124): AP per class and difficulty + ``mAP`` (mean over all nine
    entries; a class without detections contributes zeros).  ``label_dir`` is ``<data_dir>/training/label_2``."""
    with open(sample_set_path) as f:
        ids = f.read().split()
    gts = [read_label_file(os.path.join(label_dir, i + '.txt')) for i in ids]
    dets = [read_result_file(os.path.join(results_dir, 'data', i + '.txt')) for i in ids]
    ap, prec, ev = kitti_ap(gts, dets)
    aps = {}
    for c, name in enumerate(class_names):
        low = name.lower()
        ci = _KINDS.get(low, -1)
        vals = [0., 0., 0.]
        if ci in (0, 1, 2) and ev[ci]:
            # the reference parses 'AP=<6 significant digits>' back from the stats file (std::stringstream default)
            vals = [float('%g' % ap[ci, d]) for d in range(3)]
            if write_stats:
                with open(os.path.join(results_dir, 'stats_{}_ap.txt'.format(low)), 'w') as f:
                    for d in range(3):
                        f.write('AP=%g\n' % ap[ci, d])
                with open(os.path.join(results_dir, 'stats_{}_detection.txt'.format(low)), 'w') as f:
                    for d in range(3):
                        f.write(''.join('%f ' % prec[ci, d, i] for i in range(0, 41, 4)) + '\n')
        aps[name + '_easy'], aps[name + '_moderate'], aps[name + '_hard'] = vals
    aps['mAP'] = sum(aps.values()) / len(aps)
    return aps
