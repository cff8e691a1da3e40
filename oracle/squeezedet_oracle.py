"""CPU restatement of the SqueezeDet hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Every function cites the reference file:line it follows (paths relative to
/root/reference).  The restatement is functional (no nn.Module), fp32, NCHW, and uses
``torch.nn.functional`` on CPU for the convolutions/pools -- the same third-party
arithmetic library the reference delegates to (``torch.nn.Conv2d`` / ``MaxPool2d``) --
and numpy for the integer/index work of the detection filter and the GT encoder.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    "KITTI_ANCHOR_SEED", "KITTI_INPUT_SIZE", "arch_layers", "param_shapes", "generate_anchors",
    "backbone_forward", "resolve_predictions", "inference_head", "nms", "filter_detections",
    "boxes_postprocess", "multitask_loss", "compute_deltas", "encode_gt", "train_step_reference",
    "xyxy_to_xywh", "xywh_to_xyxy", "KITTI_RGB_MEAN", "KITTI_RGB_STD", "resize_linear_f32", "preprocess_image",
    "crop_or_pad_image", "boxes_unpad_uncrop",
]

EPSILON = 1e-10  # src/model/modules.py:3, src/utils/boxes.py:9

# src/datasets/kitti.py:15,27-29
KITTI_INPUT_SIZE = (384, 1248)
KITTI_ANCHOR_SEED = np.array([[34, 30], [75, 45], [38, 90], [127, 68], [80, 174], [196, 97],
                              [194, 178], [283, 156], [381, 185]], dtype=np.float32)


# --------------------------------------------------------------------------------------
# architecture tables -- src/model/squeezedet.py:32-75
# --------------------------------------------------------------------------------------
def arch_layers(arch: str):
    """Layer list of ``SqueezeDetBase.features`` as (kind, *args) tuples, index == position in
    the reference's nn.Sequential (src/model/squeezedet.py:33-49 and :51-67)."""
    if arch == "squeezedet":
        return [("conv", 3, 64, 3, 2, 1), ("relu",), ("pool",),
                ("fire", 64, 16, 64, 64), ("fire", 128, 16, 64, 64), ("pool",),
                ("fire", 128, 32, 128, 128), ("fire", 256, 32, 128, 128), ("pool",),
                ("fire", 256, 48, 192, 192), ("fire", 384, 48, 192, 192),
                ("fire", 384, 64, 256, 256), ("fire", 512, 64, 256, 256),
                ("fire", 512, 96, 384, 384), ("fire", 768, 96, 384, 384)]
    if arch == "squeezedetplus":
        return [("conv", 3, 96, 7, 2, 3), ("relu",), ("pool",),
                ("fire", 96, 96, 64, 64), ("fire", 128, 96, 64, 64), ("fire", 128, 192, 128, 128),
                ("pool",),
                ("fire", 256, 192, 128, 128), ("fire", 256, 288, 192, 192),
                ("fire", 384, 288, 192, 192), ("fire", 384, 384, 256, 256), ("pool",),
                ("fire", 512, 384, 256, 256), ("fire", 512, 384, 256, 256),
                ("fire", 512, 384, 256, 256)]
    raise ValueError("Invalid architecture.")  # src/model/squeezedet.py:69


def param_shapes(arch: str, anchors_per_grid: int = 9, num_classes: int = 3) -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape, as produced by the reference modules
    (src/model/squeezedet.py:12-14,34,73-75)."""
    shapes: Dict[str, Tuple[int, ...]] = {}
    for i, l in enumerate(arch_layers(arch)):
        if l[0] == "conv":
            _, ci, co, k, _, _ = l
            shapes[f"base.features.{i}.weight"] = (co, ci, k, k)
            shapes[f"base.features.{i}.bias"] = (co,)
        elif l[0] == "fire":
            _, ci, s, e1, e3 = l
            shapes[f"base.features.{i}.squeeze.weight"] = (s, ci, 1, 1)
            shapes[f"base.features.{i}.squeeze.bias"] = (s,)
            shapes[f"base.features.{i}.expand1x1.weight"] = (e1, s, 1, 1)
            shapes[f"base.features.{i}.expand1x1.bias"] = (e1,)
            shapes[f"base.features.{i}.expand3x3.weight"] = (e3, s, 3, 3)
            shapes[f"base.features.{i}.expand3x3.bias"] = (e3,)
    cin = 768 if arch == "squeezedet" else 512
    cout = anchors_per_grid * (num_classes + 5)
    shapes["base.convdet.weight"] = (cout, cin, 3, 3)
    shapes["base.convdet.bias"] = (cout,)
    return shapes


# --------------------------------------------------------------------------------------
# anchors -- src/utils/boxes.py:37-67
# --------------------------------------------------------------------------------------
def generate_anchors(grid_size: Tuple[int, int], input_size: Tuple[int, int],
                     anchors_seed: np.ndarray) -> np.ndarray:
    """(A,4) float64 anchors in (cx, cy, w, h); A = gh*gw*N ordered (y, x, k).

    Follows src/utils/boxes.py:46-67: centres are
    ``input * (1/(2*grid) + linspace(0,1,grid+1)[:-1])`` (the same expression, so the same
    float64 roundings), shapes are the seed repeated per cell."""
    gh, gw = grid_size
    ih, iw = input_size
    n = anchors_seed.shape[0]
    cx = iw * (1 / (gw * 2) + np.linspace(0, 1, gw + 1)[:-1])
    cy = ih * (1 / (gh * 2) + np.linspace(0, 1, gh + 1)[:-1])
    out = np.empty((gh, gw, n, 4), dtype=np.float64)
    out[..., 0] = cx[None, :, None]
    out[..., 1] = cy[:, None, None]
    out[..., 2] = anchors_seed[None, None, :, 0]
    out[..., 3] = anchors_seed[None, None, :, 1]
    return out.reshape(-1, 4)


# --------------------------------------------------------------------------------------
# backbone + ConvDet -- src/model/squeezedet.py:9-23, 33-49, 79-87
# --------------------------------------------------------------------------------------
def _fire(x, p, prefix):
    # src/model/squeezedet.py:17-23
    s = F.relu(F.conv2d(x, p[prefix + ".squeeze.weight"], p[prefix + ".squeeze.bias"]))
    a = F.relu(F.conv2d(s, p[prefix + ".expand1x1.weight"], p[prefix + ".expand1x1.bias"]))
    b = F.relu(F.conv2d(s, p[prefix + ".expand3x3.weight"], p[prefix + ".expand3x3.bias"], padding=1))
    return torch.cat([a, b], dim=1)


def backbone_forward(image: torch.Tensor, params: Dict[str, torch.Tensor], arch: str = "squeezedet",
                     num_classes: int = 3, drop_mask: Optional[torch.Tensor] = None,
                     capture: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """``SqueezeDetBase.forward`` (src/model/squeezedet.py:79-87): features -> [dropout] ->
    ConvDet -> NHWC permute -> view [B, A, C+5].

    ``drop_mask`` (same shape as the feature map, already scaled by 1/(1-p)) replaces
    nn.Dropout so that train-mode runs are reproducible; ``None`` == eval / dropout_prob=0.
    ``capture`` (optional dict) receives the NCHW output of every layer for per-layer tests."""
    x = image
    for i, l in enumerate(arch_layers(arch)):
        if l[0] == "conv":
            x = F.conv2d(x, params[f"base.features.{i}.weight"], params[f"base.features.{i}.bias"],
                         stride=l[4], padding=l[5])                       # :34 / :52
        elif l[0] == "relu":
            x = F.relu(x)                                                 # :35
        elif l[0] == "pool":
            x = F.max_pool2d(x, kernel_size=3, stride=2, ceil_mode=True)  # :36,39,42
        else:
            x = _fire(x, params, f"base.features.{i}")
        if capture is not None:
            capture[f"features.{i}"] = x
    if drop_mask is not None:
        x = x * drop_mask                                                 # :81-82
    x = F.conv2d(x, params["base.convdet.weight"], params["base.convdet.bias"], padding=1)  # :83
    if capture is not None:
        capture["convdet"] = x
    b = x.shape[0]
    x = x.permute(0, 2, 3, 1).contiguous()                                # :85
    return x.view(b, -1, num_classes + 5)                                 # :87


# --------------------------------------------------------------------------------------
# prediction decode -- src/model/squeezedet.py:109-120, src/model/modules.py:17-45,66-68
# --------------------------------------------------------------------------------------
def xywh_to_xyxy(b: torch.Tensor) -> torch.Tensor:
    """src/model/modules.py:17-24 (asserts w,h > 0 like :18)."""
    assert bool(torch.all(b[..., 2:4] > 0)), "non-positive box size"
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * (w - 1), cy - 0.5 * (h - 1), cx + 0.5 * (w - 1), cy + 0.5 * (h - 1)], -1)


def xyxy_to_xywh(b: torch.Tensor) -> torch.Tensor:
    """src/model/modules.py:6-14."""
    x1, y1, x2, y2 = b.unbind(-1)
    return torch.stack([(x1 + x2) / 2., (y1 + y2) / 2., x2 - x1 + 1., y2 - y1 + 1.], -1)


def resolve_predictions(pred: torch.Tensor, anchors: np.ndarray, input_size: Tuple[int, int],
                        num_classes: int = 3, log_softmax: bool = False):
    """``PredictionResolver.forward`` (src/model/squeezedet.py:109-120).

    Returns (class_probs [B,A,C], log_class_probs|None, scores [B,A,1], deltas [B,A,4],
    boxes_xyxy [B,A,4])."""
    anc = torch.from_numpy(np.asarray(anchors)).unsqueeze(0).float()      # :106
    logits = pred[..., :num_classes]
    z = logits - logits.max(dim=-1, keepdim=True)[0]                      # modules.py:67
    e = torch.exp(z)
    probs = e / e.sum(dim=-1, keepdim=True)                               # modules.py:68
    logp = torch.log_softmax(logits, dim=-1) if log_softmax else None     # :111-112
    scores = torch.sigmoid(pred[..., num_classes:num_classes + 1])        # :114
    deltas = pred[..., num_classes + 1:]                                  # :116
    ax, ay, aw, ah = anc.unbind(-1)
    dx, dy, dw, dh = deltas.unbind(-1)
    xywh = torch.stack([ax + aw * dx, ay + ah * dy, aw * torch.exp(dw), ah * torch.exp(dh)], -1)  # modules.py:34-39
    xyxy = xywh_to_xyxy(xywh)                                             # modules.py:41
    x1, y1, x2, y2 = xyxy.unbind(-1)
    wmax, hmax = input_size[1] - 1, input_size[0] - 1
    boxes = torch.stack([x1.clamp(0, wmax), y1.clamp(0, hmax), x2.clamp(0, wmax), y2.clamp(0, hmax)], -1)  # :42-43
    return probs, logp, scores, deltas, boxes


def inference_head(pred: torch.Tensor, anchors: np.ndarray, input_size: Tuple[int, int], num_classes: int = 3):
    """``SqueezeDet.forward`` after the base (src/model/squeezedet.py:199-206):
    probs *= score; class_ids = argmax; scores = max."""
    probs, _, scores, _, boxes = resolve_predictions(pred, anchors, input_size, num_classes)
    probs = probs * scores                                                # :200
    class_ids = torch.argmax(probs, dim=2)                                # :201
    best = torch.max(probs, dim=2)[0]                                     # :202
    return class_ids, best, boxes


# --------------------------------------------------------------------------------------
# detection filter -- src/engine/detector.py:87-122 (+ torchvision.ops.nms, third party)
# --------------------------------------------------------------------------------------
def nms(boxes: np.ndarray, scores: np.ndarray, iou_threshold: float) -> np.ndarray:
    """Greedy NMS, restating torchvision.ops.nms (torchvision==0.3.0 per requirements.txt:27;
    source NOT under /root/reference -- published algorithm restated, parity unpinned):
    visit boxes by descending score (stable: equal scores keep input order); a visited,
    unsuppressed box suppresses every later box whose IoU with it is > threshold, with
    IoU = inter / (area_i + area_j - inter), area = (x2-x1)*(y2-y1) (no +1), inter sides
    clamped at 0.  All arithmetic in float32.  Returns kept indices in descending score order."""
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    order = np.argsort(-scores, kind="stable")
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    thr = np.float32(iou_threshold)
    suppressed = np.zeros(len(order), dtype=bool)
    keep: List[int] = []
    zero = np.float32(0)
    for a in range(len(order)):
        if suppressed[a]:
            continue
        i = order[a]
        keep.append(int(i))
        for b in range(a + 1, len(order)):
            if suppressed[b]:
                continue
            j = order[b]
            w = max(zero, min(x2[i], x2[j]) - max(x1[i], x1[j]))
            h = max(zero, min(y2[i], y2[j]) - max(y1[i], y1[j]))
            inter = np.float32(w * h)
            with np.errstate(divide="ignore", invalid="ignore"):
                ovr = np.float32(inter / np.float32(np.float32(areas[i] + areas[j]) - inter))
            if ovr > thr:
                suppressed[b] = True
    return np.asarray(keep, dtype=np.int64)


def filter_detections(class_ids, scores, boxes, keep_top_k: int = 64, nms_thresh: float = 0.4,
                      score_thresh: float = 0.3, num_classes: int = 3) -> Optional[Dict[str, np.ndarray]]:
    """``Detector.filter`` for ONE image (src/engine/detector.py:87-122).

    top-k by score (:88, ties -> lower anchor index first: the build's documented rule, the
    reference's torch.argsort tie order is unspecified), class-wise NMS in class order
    0..C-1 (:95-108), concatenation (:110-112), score threshold (:114), ``None`` when nothing
    survives (:115-116).  Additionally returns ``anchor_idx`` (the index into the A anchors of
    every kept detection) which the reference does not return."""
    class_ids = np.asarray(class_ids).reshape(-1)
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    order = np.argsort(-scores, kind="stable")[:keep_top_k]               # :88
    c, s, b = class_ids[order], scores[order], boxes[order]               # :89-91
    out_c, out_s, out_b, out_i = [], [], [], []
    for cls in range(num_classes):                                        # :95
        m = np.nonzero(c == cls)[0]                                       # :96
        if m.size == 0:                                                   # :97
            continue
        keep = nms(b[m], s[m], nms_thresh)                                # :104
        sel = m[keep]
        out_c.append(c[sel]); out_s.append(s[sel]); out_b.append(b[sel]); out_i.append(order[sel])
    if not out_c:
        return None
    c = np.concatenate(out_c); s = np.concatenate(out_s); b = np.concatenate(out_b, 0); i = np.concatenate(out_i)
    keep = s > np.float32(score_thresh)                                   # :114
    if keep.sum() == 0:                                                   # :115
        return None
    return {"class_ids": c[keep].astype(np.int64), "scores": s[keep], "boxes": b[keep],
            "anchor_idx": i[keep].astype(np.int64)}


def boxes_postprocess(boxes: np.ndarray, scales: Sequence[float]) -> np.ndarray:
    """Eval-time branch of src/utils/boxes.py:138-168: x /= scales[1], y /= scales[0]
    (``scales = [target_h/H0, target_w/W0]`` float32, src/utils/image.py:79); drifts are
    [0,0] at eval (src/datasets/base.py:48) so the remaining branches are inactive."""
    out = np.array(boxes, dtype=np.float32, copy=True)
    sc = np.asarray(scales, dtype=np.float32)
    out[:, [0, 2]] /= sc[1]
    out[:, [1, 3]] /= sc[0]
    return out


def boxes_unpad_uncrop(boxes: np.ndarray, padding, crops) -> np.ndarray:
    """The padding / crops branches of src/utils/boxes.py:149-155 (the ``cfg.forbid_resize`` pre-processing records both, no
    ``scales``): x -= padding[2], y -= padding[0]; then x += crops[2], y += crops[0]; float32, in that order.  Pinned by
    tests/golden/padcrop.npz (the reference's own ``boxes_postprocess`` outputs)."""
    out = np.array(boxes, dtype=np.float32, copy=True)
    out[:, [0, 2]] -= padding[2]
    out[:, [1, 3]] -= padding[0]
    out[:, [0, 2]] += crops[2]
    out[:, [1, 3]] += crops[0]
    return out


# --------------------------------------------------------------------------------------
# loss -- src/model/squeezedet.py:133-174, src/model/modules.py:48-63
# --------------------------------------------------------------------------------------
def _overlaps(b1: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    # src/model/modules.py:56-63 (keeps the trailing singleton dim like the reference)
    lr = torch.clamp_min(torch.min(b1[..., 2:3], b2[..., 2:3]) - torch.max(b1[..., 0:1], b2[..., 0:1]), 0)
    tb = torch.clamp_min(torch.min(b1[..., 3:4], b2[..., 3:4]) - torch.max(b1[..., 1:2], b2[..., 1:2]), 0)
    inter = lr * tb
    union = (b1[..., 2:3] - b1[..., 0:1]) * (b1[..., 3:4] - b1[..., 1:2]) + \
            (b2[..., 2:3] - b2[..., 0:1]) * (b2[..., 3:4] - b2[..., 1:2]) - inter
    return inter / (union + EPSILON)


def multitask_loss(pred: torch.Tensor, gt: torch.Tensor, anchors: np.ndarray, input_size: Tuple[int, int],
                   num_classes: int = 3, class_w: float = 1., pos_w: float = 3.75, neg_w: float = 100.,
                   bbox_w: float = 6.):
    """``Loss.forward`` (src/model/squeezedet.py:133-174).  Differentiable w.r.t. ``pred``
    through torch autograd, including the un-detached IoU path (:144,:151-152).
    Returns (loss [B], stats dict of [B])."""
    num_anchors = pred.shape[1]
    mask = gt[..., :1]                                                    # :135
    gt_boxes = gt[..., 1:5]                                               # :136
    gt_deltas = gt[..., 5:9]                                              # :137
    onehot = gt[..., 9:]                                                  # :138
    _, logp, scores, deltas, boxes = resolve_predictions(pred, anchors, input_size, num_classes, log_softmax=True)
    n_obj = mask.sum(dim=[1, 2])                                          # :143
    iou = _overlaps(gt_boxes, boxes) * mask                               # :144
    class_loss = (class_w * mask * onehot * (-logp)).sum(dim=[1, 2]) / n_obj              # :146-149
    pos = (pos_w * mask * (iou - scores) ** 2).sum(dim=[1, 2]) / n_obj                   # :151-154
    neg = (neg_w * (1 - mask) * (iou - scores) ** 2).sum(dim=[1, 2]) / (num_anchors - n_obj)  # :156-159
    bbox = (bbox_w * mask * (deltas - gt_deltas) ** 2).sum(dim=[1, 2]) / n_obj           # :161-164
    loss = class_loss + pos + neg + bbox                                  # :166
    return loss, {"loss": loss, "class_loss": class_loss, "score_loss": pos + neg, "bbox_loss": bbox}


# --------------------------------------------------------------------------------------
# GT encoding (hot-path *input*, CPU) -- src/utils/boxes.py:12-34,70-135, src/datasets/base.py:61-76
# --------------------------------------------------------------------------------------
def compute_deltas(boxes_xyxy: np.ndarray, anchors_xywh: np.ndarray, ties: str = "argsort"):
    """Greedy unique anchor assignment + regression targets (src/utils/boxes.py:84-135).

    ``ties="argsort"`` keeps the reference's literal ``np.argsort`` walk (tie order = whatever numpy's unstable sort
    yields on this machine); ``ties="lowest"`` is the rule the HIP encoder implements: among free anchors with
    exactly equal overlap (distance) the lowest anchor index wins.  Both pick an anchor of maximal free overlap."""
    if ties == "lowest":
        return _compute_deltas_lowest(boxes_xyxy, anchors_xywh)
    boxes_xyxy = np.asarray(boxes_xyxy)
    A = anchors_xywh.shape[0]
    bx = np.stack([(boxes_xyxy[:, 0] + boxes_xyxy[:, 2]) / 2., (boxes_xyxy[:, 1] + boxes_xyxy[:, 3]) / 2.,
                   boxes_xyxy[:, 2] - boxes_xyxy[:, 0] + 1., boxes_xyxy[:, 3] - boxes_xyxy[:, 1] + 1.], 1)  # :17-22
    ax = np.stack([anchors_xywh[:, 0] - 0.5 * (anchors_xywh[:, 2] - 1), anchors_xywh[:, 1] - 0.5 * (anchors_xywh[:, 3] - 1),
                   anchors_xywh[:, 0] + 0.5 * (anchors_xywh[:, 2] - 1), anchors_xywh[:, 1] + 0.5 * (anchors_xywh[:, 3] - 1)], 1)  # :29-34
    taken = set()
    idxs, deltas = [], []
    for i in range(boxes_xyxy.shape[0]):
        box = boxes_xyxy[i]
        lr = np.maximum(np.minimum(ax[:, 2], box[2]) - np.maximum(ax[:, 0], box[0]), 0)   # :76
        tb = np.maximum(np.minimum(ax[:, 3], box[3]) - np.maximum(ax[:, 1], box[1]), 0)   # :77
        inter = lr * tb
        union = (ax[:, 2] - ax[:, 0]) * (ax[:, 3] - ax[:, 1]) + (box[2] - box[0]) * (box[3] - box[1]) - inter
        ov = inter / (union + EPSILON)                                                   # :81
        chosen = A
        for j in np.argsort(-ov):                                                        # :104
            if ov[j] <= 0:                                                               # :106
                break
            if j not in taken:                                                           # :108
                taken.add(j); chosen = j
                break
        if chosen == A:                                                                  # :115
            dist = np.sum((bx[i] - anchors_xywh) ** 2, axis=1)                           # :116
            for j in np.argsort(dist):
                if j not in taken:
                    taken.add(j); chosen = j
                    break
        idxs.append(chosen)
        deltas.append([(bx[i, 0] - anchors_xywh[chosen, 0]) / anchors_xywh[chosen, 2],   # :125-128
                       (bx[i, 1] - anchors_xywh[chosen, 1]) / anchors_xywh[chosen, 3],
                       np.log(bx[i, 2] / anchors_xywh[chosen, 2]),
                       np.log(bx[i, 3] / anchors_xywh[chosen, 3])])
    return np.array(deltas, dtype=np.float32), np.array(idxs, dtype=np.int32)            # :132-133


def _compute_deltas_lowest(boxes_xyxy: np.ndarray, anchors_xywh: np.ndarray):
    """Same arithmetic as ``compute_deltas`` (same numpy expressions, hence the same float32/float64 promotion),
    selection by masked arg-max / arg-min (numpy returns the first, i.e. lowest-index, extremum)."""
    boxes_xyxy = np.asarray(boxes_xyxy)
    A = anchors_xywh.shape[0]
    bx = np.stack([(boxes_xyxy[:, 0] + boxes_xyxy[:, 2]) / 2., (boxes_xyxy[:, 1] + boxes_xyxy[:, 3]) / 2.,
                   boxes_xyxy[:, 2] - boxes_xyxy[:, 0] + 1., boxes_xyxy[:, 3] - boxes_xyxy[:, 1] + 1.], 1)
    ax = np.stack([anchors_xywh[:, 0] - 0.5 * (anchors_xywh[:, 2] - 1), anchors_xywh[:, 1] - 0.5 * (anchors_xywh[:, 3] - 1),
                   anchors_xywh[:, 0] + 0.5 * (anchors_xywh[:, 2] - 1), anchors_xywh[:, 1] + 0.5 * (anchors_xywh[:, 3] - 1)], 1)
    taken = np.zeros(A, dtype=bool)
    idxs, deltas = [], []
    for i in range(boxes_xyxy.shape[0]):
        box = boxes_xyxy[i]
        lr = np.maximum(np.minimum(ax[:, 2], box[2]) - np.maximum(ax[:, 0], box[0]), 0)
        tb = np.maximum(np.minimum(ax[:, 3], box[3]) - np.maximum(ax[:, 1], box[1]), 0)
        inter = lr * tb
        union = (ax[:, 2] - ax[:, 0]) * (ax[:, 3] - ax[:, 1]) + (box[2] - box[0]) * (box[3] - box[1]) - inter
        ov = inter / (union + EPSILON)
        if taken.all():
            idxs.append(A); deltas.append([0., 0., 0., 0.])
            continue
        j = int(np.argmax(np.where(taken, -1.0, ov)))
        if not ov[j] > 0:
            dist = np.sum((bx[i] - anchors_xywh) ** 2, axis=1)
            j = int(np.argmin(np.where(taken, np.inf, dist)))
        taken[j] = True
        idxs.append(j)
        deltas.append([(bx[i, 0] - anchors_xywh[j, 0]) / anchors_xywh[j, 2], (bx[i, 1] - anchors_xywh[j, 1]) / anchors_xywh[j, 3],
                       np.log(bx[i, 2] / anchors_xywh[j, 2]), np.log(bx[i, 3] / anchors_xywh[j, 3])])
    return np.array(deltas, dtype=np.float32).reshape(-1, 4), np.array(idxs, dtype=np.int32)


def encode_gt(class_ids: np.ndarray, boxes_xyxy: np.ndarray, anchors_xywh: np.ndarray, num_classes: int = 3,
              ties: str = "argsort") -> np.ndarray:
    """Dense gt [A, C+9] = [mask, x1,y1,x2,y2, dx,dy,dw,dh, onehot] (src/datasets/base.py:61-76)."""
    deltas, idx = compute_deltas(boxes_xyxy, anchors_xywh, ties=ties)
    gt = np.zeros((anchors_xywh.shape[0], num_classes + 9), dtype=np.float32)
    gt[idx, 0] = 1.
    gt[idx, 1:5] = boxes_xyxy
    gt[idx, 5:9] = deltas
    gt[idx, 9 + np.asarray(class_ids)] = 1.
    return gt


# --------------------------------------------------------------------------------------
# one training step -- src/engine/trainer.py:42-50, src/train.py:32-35
# --------------------------------------------------------------------------------------
def train_step_reference(params: Dict[str, torch.Tensor], momentum_buf: Optional[Dict[str, torch.Tensor]],
                         image: torch.Tensor, gt: torch.Tensor, anchors: np.ndarray, input_size,
                         arch: str = "squeezedet", lr: float = 0.01, momentum: float = 0.9,
                         weight_decay: float = 1e-4, grad_norm: float = 5.0, num_classes: int = 3,
                         loss_weights=(1., 3.75, 100., 6.), global_batch: Optional[int] = None,
                         drop_mask: Optional[torch.Tensor] = None):
    """fwd -> loss.mean() -> backward -> clip_grad_norm_(5.0) -> SGD(momentum, wd) step
    (src/engine/trainer.py:42-50; optimiser per src/train.py:32-35).  Functional: returns
    (new_params, new_momentum, grads, clipped_total_norm, loss_vec, stats).
    ``global_batch`` lets a data-parallel shard scale its loss by 1/B_global (the
    reference's ``loss.mean()`` runs over the gathered global vector, trainer.py:43)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    pred = backbone_forward(image, leaves, arch, num_classes, drop_mask)
    loss_vec, stats = multitask_loss(pred, gt, anchors, input_size, num_classes, *loss_weights)
    denom = loss_vec.numel() if global_batch is None else global_batch
    (loss_vec.sum() / denom).backward()
    names = list(leaves.keys())
    grads = {k: leaves[k].grad.detach().clone() for k in names}
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    coef = min(1.0, grad_norm / (total + 1e-6))                           # torch.nn.utils.clip_grad_norm_
    new_p, new_m = {}, {}
    for k in names:
        g = grads[k] * coef
        d = g + weight_decay * params[k]
        if momentum_buf is None or k not in momentum_buf:
            buf = d.clone()                                               # torch.optim.SGD first step
        else:
            buf = momentum * momentum_buf[k] + d
        new_m[k] = buf
        new_p[k] = params[k] - lr * buf
    return new_p, new_m, grads, total, loss_vec.detach(), {k: v.detach() for k, v in stats.items()}


# --------------------------------------------------------------------------------------
# input pipeline -- src/utils/image.py:9-19,77-88, src/datasets/base.py:43-59,
# src/engine/detector.py:132-142; KITTI statistics src/datasets/kitti.py:17-18
# --------------------------------------------------------------------------------------
KITTI_RGB_MEAN = np.array([93.877, 98.801, 95.923], dtype=np.float32)
KITTI_RGB_STD = np.array([78.782, 80.130, 81.200], dtype=np.float32)


def resize_linear_f32(img: np.ndarray, dst_hw: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(img, (W, H)) with the default INTER_LINEAR on float32 HWC data.  OpenCV is third party and absent
    here (source not under /root/reference): its published algorithm restated -- parity unpinned against cv2 itself.
    fx = (x + 0.5) * (W0 / W) - 0.5 in float64, cast to float32; sx = floor(fx), fx -= sx; clamped to the border with
    weight 0 beyond it; horizontal pass then vertical pass, float32 arithmetic."""
    img = np.asarray(img, dtype=np.float32)
    H0, W0 = img.shape[:2]
    H, W = dst_hw

    def coords(n_dst, n_src):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * (float(n_src) / float(n_dst)) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        s[lo] = 0; f[lo] = 0
        hi = s >= n_src - 1
        s[hi] = n_src - 1; f[hi] = 0
        return s, np.minimum(s + 1, n_src - 1), f

    sx, sx1, fx = coords(W, W0)
    sy, sy1, fy = coords(H, H0)
    ax0 = (np.float32(1) - fx)[None, :, None]; ax1 = fx[None, :, None]
    rows0 = img[sy][:, sx] * ax0 + img[sy][:, sx1] * ax1
    rows1 = img[sy1][:, sx] * ax0 + img[sy1][:, sx1] * ax1
    ay0 = (np.float32(1) - fy)[:, None, None]; ay1 = fy[:, None, None]
    return (rows0 * ay0 + rows1 * ay1).astype(np.float32)


def preprocess_image(img_u8: np.ndarray, input_size: Tuple[int, int], mean=KITTI_RGB_MEAN, std=KITTI_RGB_STD):
    """Eval-time ``DataWrapper.__getitem__`` for one image: float32 cast (kitti.py:52), whiten (image.py:17), resize
    (image.py:77-86), HWC->CHW (detector.py:140).  Returns (image float32 [3,H,W], scales float32 [2])."""
    x = img_u8.astype(np.float32)
    x = (x - np.asarray(mean, np.float32).reshape(1, 1, 3)) / np.asarray(std, np.float32).reshape(1, 1, 3)
    scales = np.array([input_size[0] / x.shape[0], input_size[1] / x.shape[1]], dtype=np.float32)
    y = resize_linear_f32(x, input_size)
    return np.ascontiguousarray(y.transpose(2, 0, 1)), scales


def crop_or_pad_image(img_u8: np.ndarray, input_size: Tuple[int, int], mean=KITTI_RGB_MEAN, std=KITTI_RGB_STD):
    """The input pipeline's ``cfg.forbid_resize`` branch for one image (src/datasets/base.py:51-54): float32 cast (kitti.py:52),
    whiten (image.py:17), ``crop_or_pad`` (image.py:91-124) -- per axis a smaller image is zero-padded AFTER whitening (floor half of
    the difference in front, ``np.pad`` constant 0), a larger one centre-cropped (floor half cut in front) -- HWC->CHW
    (detector.py:140).  Returns (image float32 [3,H,W], padding int16 [4], crops int16 [4]), both (top, bottom, left, right).
    Pinned against the reference's own functions by tests/golden/padcrop.npz."""
    x = img_u8.astype(np.float32)
    x = (x - np.asarray(mean, np.float32).reshape(1, 1, 3)) / np.asarray(std, np.float32).reshape(1, 1, 3)
    H, W = int(input_size[0]), int(input_size[1])
    h, w = x.shape[:2]
    padding, crops = np.zeros(4, np.int16), np.zeros(4, np.int16)
    if h < H:
        padding[0] = (H - h) // 2; padding[1] = (H - h) - padding[0]
    elif h > H:
        crops[0] = (h - H) // 2; crops[1] = (h - H) - crops[0]
    if w < W:
        padding[2] = (W - w) // 2; padding[3] = (W - w) - padding[2]
    elif w > W:
        crops[2] = (w - W) // 2; crops[3] = (w - W) - crops[2]
    out = np.zeros((H, W, 3), np.float32)
    ys, xs = int(crops[0]), int(crops[2])                       # first source row / column kept
    nh, nw = min(h, H), min(w, W)
    out[int(padding[0]):int(padding[0]) + nh, int(padding[2]):int(padding[2]) + nw] = x[ys:ys + nh, xs:xs + nw]
    return np.ascontiguousarray(out.transpose(2, 0, 1)), padding, crops
