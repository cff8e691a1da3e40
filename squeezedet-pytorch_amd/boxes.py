"""Host-side box utilities the hot path consumes as *inputs*: the anchor grid and the dense
GT tensor.  Mirrors the interface of the reference's ``src/utils/boxes.py`` /
``src/datasets/base.py`` (same names and argument meaning); numpy only.

Reference: ``generate_anchors`` src/utils/boxes.py:37-67, ``compute_deltas`` :84-135,
``boxes_postprocess`` :138-168, ``BaseDataset.prepare_annotations`` src/datasets/base.py:61-76.
"""
from __future__ import annotations

import numpy as np

EPSILON = 1e-10

# KITTI constants (src/datasets/kitti.py:15,26-31)
KITTI_INPUT_SIZE = (384, 1248)
KITTI_ANCHORS_SEED = np.array([[34, 30], [75, 45], [38, 90], [127, 68], [80, 174], [196, 97],
                               [194, 178], [283, 156], [381, 185]], dtype=np.float32)


def generate_anchors(grid_size, input_size, anchors_seed):
    """(grid_h*grid_w*N, 4) float64 anchors ``(cx, cy, w, h)``, ordered (y, x, k).

    Cell centres sit at ``input * (1/(2*grid) + i/grid)``; the expression is evaluated through
    ``linspace`` exactly as the reference does so the float64 values agree bit for bit."""
    anchors_seed = np.asarray(anchors_seed)
    assert anchors_seed.ndim == 2 and anchors_seed.shape[1] == 2
    grid_h, grid_w = grid_size
    in_h, in_w = input_size
    n = anchors_seed.shape[0]
    xs = in_w * (1 / (grid_w * 2) + np.linspace(0, 1, grid_w + 1)[:-1])
    ys = in_h * (1 / (grid_h * 2) + np.linspace(0, 1, grid_h + 1)[:-1])
    a = np.zeros((grid_h, grid_w, n, 4), dtype=np.float64)
    a[:, :, :, 0] = xs.reshape(1, grid_w, 1)
    a[:, :, :, 1] = ys.reshape(grid_h, 1, 1)
    a[:, :, :, 2:] = anchors_seed.reshape(1, 1, n, 2)
    return a.reshape(-1, 4)


def xyxy_to_xywh(boxes_xyxy):
    boxes_xyxy = np.asarray(boxes_xyxy)
    assert boxes_xyxy.ndim == 2
    assert np.all(boxes_xyxy[:, 0] < boxes_xyxy[:, 2]) and np.all(boxes_xyxy[:, 1] < boxes_xyxy[:, 3])
    out = np.empty_like(boxes_xyxy)
    out[:, 0] = (boxes_xyxy[:, 0] + boxes_xyxy[:, 2]) / 2.
    out[:, 1] = (boxes_xyxy[:, 1] + boxes_xyxy[:, 3]) / 2.
    out[:, 2] = boxes_xyxy[:, 2] - boxes_xyxy[:, 0] + 1.
    out[:, 3] = boxes_xyxy[:, 3] - boxes_xyxy[:, 1] + 1.
    return out


def xywh_to_xyxy(boxes_xywh):
    boxes_xywh = np.asarray(boxes_xywh)
    assert boxes_xywh.ndim == 2 and np.all(boxes_xywh > 0)
    half_w = 0.5 * (boxes_xywh[:, 2] - 1)
    half_h = 0.5 * (boxes_xywh[:, 3] - 1)
    return np.stack([boxes_xywh[:, 0] - half_w, boxes_xywh[:, 1] - half_h,
                     boxes_xywh[:, 0] + half_w, boxes_xywh[:, 1] + half_h], axis=1)


def _iou_one_to_many(boxes, box):
    w = np.maximum(np.minimum(boxes[:, 2], box[2]) - np.maximum(boxes[:, 0], box[0]), 0)
    h = np.maximum(np.minimum(boxes[:, 3], box[3]) - np.maximum(boxes[:, 1], box[1]), 0)
    inter = w * h
    union = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]) + \
            (box[2] - box[0]) * (box[3] - box[1]) - inter
    return inter / (union + EPSILON)


def compute_deltas(boxes_xyxy, anchors_xywh):
    """Assign every GT box a *distinct* anchor (best free IoU, else nearest free by squared
    (cx,cy,w,h) distance) and return (deltas float32 [n,4], anchor_indices int32 [n])."""
    boxes_xyxy = np.asarray(boxes_xyxy)
    num_anchors = anchors_xywh.shape[0]
    boxes_xywh = xyxy_to_xywh(boxes_xyxy)
    anchors_xyxy = xywh_to_xyxy(anchors_xywh)
    used = np.zeros(num_anchors, dtype=bool)
    indices = np.empty(boxes_xyxy.shape[0], dtype=np.int32)
    deltas = np.empty((boxes_xyxy.shape[0], 4), dtype=np.float32)
    for i in range(boxes_xyxy.shape[0]):
        iou = _iou_one_to_many(anchors_xyxy, boxes_xyxy[i])
        pick = -1
        for j in np.argsort(-iou):
            if iou[j] <= 0:
                break
            if not used[j]:
                pick = j
                break
        if pick < 0:
            d2 = np.sum((boxes_xywh[i] - anchors_xywh) ** 2, axis=1)
            for j in np.argsort(d2):
                if not used[j]:
                    pick = j
                    break
        used[pick] = True
        indices[i] = pick
        a = anchors_xywh[pick]
        deltas[i] = [(boxes_xywh[i, 0] - a[0]) / a[2], (boxes_xywh[i, 1] - a[1]) / a[3],
                     np.log(boxes_xywh[i, 2] / a[2]), np.log(boxes_xywh[i, 3] / a[3])]
    return deltas, indices


def prepare_annotations(class_ids, boxes, anchors, num_classes):
    """Dense GT ``[A, num_classes + 9]`` = ``[mask, x1,y1,x2,y2, dx,dy,dw,dh, one-hot]``."""
    deltas, anchor_indices = compute_deltas(boxes, anchors)
    gt = np.zeros((anchors.shape[0], num_classes + 9), dtype=np.float32)
    gt[anchor_indices, 0] = 1.
    gt[anchor_indices, 1:5] = boxes
    gt[anchor_indices, 5:9] = deltas
    gt[anchor_indices, 9 + np.asarray(class_ids)] = 1.
    return gt


def boxes_postprocess(boxes, image_meta):
    """Undo, in place, the geometric pre-processing recorded in ``image_meta`` so that ``boxes`` ([n,4] xyxy, network-input
    coordinates) land in original-image coordinates; returns ``boxes``.  Same transforms, order and float arithmetic as
    src/utils/boxes.py:138-168: un-scale (``scales`` = (sy, sx)), un-pad, un-crop (both (top, bottom, left, right)),
    mirror back if ``flipped``, un-drift (``drifts`` = (dy, dx)).  The eval path only ever carries ``scales`` and zero
    ``drifts``; the device path folds that division into the detect kernel."""
    xs, ys = boxes[:, 0::2], boxes[:, 1::2]            # strided views: (x1, x2) and (y1, y2) columns
    if 'scales' in image_meta:
        sy, sx = image_meta['scales'][0], image_meta['scales'][1]
        xs /= sx
        ys /= sy
    for key, sign in (('padding', -1.0), ('crops', 1.0)):
        if key in image_meta:
            top, left = image_meta[key][0], image_meta[key][2]
            xs += sign * left
            ys += sign * top
    if image_meta.get('flipped', False):
        size = image_meta['drifted_size'] if 'drifted_size' in image_meta else image_meta['orig_size']
        extent = boxes[:, 2] - boxes[:, 0] + 1.
        boxes[:, 0] = size[1] - 1 - boxes[:, 2]
        boxes[:, 2] = boxes[:, 0] + extent - 1.
    if 'drifts' in image_meta:
        dy, dx = image_meta['drifts'][0], image_meta['drifts'][1]
        xs += dx
        ys += dy
    return boxes
