"""On-device GT encoding.

The reference builds the dense ``gt [A, C+9]`` tensor per image on the CPU inside DataLoader workers
(``BaseDataset.prepare_annotations`` src/datasets/base.py:61-76 -> ``compute_deltas`` src/utils/boxes.py:84-135: a Python
loop over boxes with an ``argsort`` over all 16848 anchors per box) and uploads 0.81 MB of fp32 per image.  Here only
the boxes and class ids are uploaded (a few hundred bytes) and one kernel assigns anchors, computes the regression
targets and writes the dense tensor for the whole batch.

Tie rule: among free anchors with exactly equal overlap (or distance) the LOWEST anchor index wins.  The reference
leaves such ties to ``np.argsort``'s unstable order (numpy-version and CPU dependent); exact ties are common (an
anchor shape lying inside a box has the same IoU at every grid position where it still lies inside).  The host
restatement ``boxes.compute_deltas`` keeps the reference's literal ``np.argsort`` call for anyone who needs the
same-machine behaviour.
"""
from __future__ import annotations

import hashlib

import numpy as np
import torch

from . import ops

_anchor_cache = {}


def anchors_f64_on(anchors, device):
    """float64 [A,4] device copy of ``cfg.anchors`` (cached per (content hash, device))."""
    a = np.ascontiguousarray(np.asarray(anchors, dtype=np.float64))
    key = (a.shape, hashlib.sha1(a.tobytes()).hexdigest(), str(device))
    t = _anchor_cache.get(key)
    if t is None:
        t = torch.from_numpy(a).to(device)
        _anchor_cache[key] = t
    return t


def pack_annotations(class_ids_list, boxes_list):
    """Per-image lists -> (boxes [total,4] f32, class_ids [total] i32, box_offsets [B+1] i32) numpy, with the
    reference's input checks (xyxy_to_xywh asserts x1 < x2 and y1 < y2, src/utils/boxes.py:13-15)."""
    if len(class_ids_list) != len(boxes_list) or len(boxes_list) == 0:
        raise ValueError('pack_annotations: need one class-id array and one box array per image')
    offs = np.zeros(len(boxes_list) + 1, dtype=np.int32)
    bl, cl = [], []
    for i, (c, b) in enumerate(zip(class_ids_list, boxes_list)):
        b = np.asarray(b, dtype=np.float32).reshape(-1, 4)
        c = np.asarray(c).reshape(-1)
        if c.shape[0] != b.shape[0]:
            raise ValueError(f'pack_annotations: image {i}: {c.shape[0]} class ids for {b.shape[0]} boxes')
        assert np.all(b[:, 0] < b[:, 2]) and np.all(b[:, 1] < b[:, 3]), 'boxes must satisfy x1 < x2 and y1 < y2'
        bl.append(b); cl.append(c.astype(np.int32))
        offs[i + 1] = offs[i] + b.shape[0]
    return np.concatenate(bl, 0) if bl else np.zeros((0, 4), np.float32), np.concatenate(cl, 0), offs


def encode_annotations(class_ids_list, boxes_list, anchors, num_classes, device='cuda', return_sparse=False):
    """Batch version of ``prepare_annotations``: lists (one entry per image) of class ids [n_i] and xyxy boxes
    [n_i,4] in network-input coordinates -> gt fp32 [B, A, num_classes+9] on ``device``.  With ``return_sparse``
    also returns (anchor_idx [total] i32, deltas [total,4] f32, box_offsets [B+1] i32), all on the device."""
    boxes, cls, offs = pack_annotations(class_ids_list, boxes_list)
    A = np.asarray(anchors).shape[0]
    if np.any(np.diff(offs) > A):
        raise IndexError('more boxes than anchors in one image')       # the reference indexes gt[num_anchors] here
    if cls.size and (cls.min() < 0 or cls.max() >= num_classes):
        raise IndexError('class id out of range')
    dev = torch.device(device)
    d_boxes = torch.from_numpy(boxes).to(dev, non_blocking=True)
    d_cls = torch.from_numpy(cls).to(dev, non_blocking=True)
    d_offs = torch.from_numpy(offs).to(dev, non_blocking=True)
    gt, idx, deltas = ops.encode_gt(d_boxes, d_cls, d_offs, anchors_f64_on(anchors, dev), num_classes)
    if return_sparse:
        return gt, idx, deltas, d_offs
    return gt
