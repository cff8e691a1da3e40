"""Seeded synthetic weights / images / ground truth for tests and the benchmark.

There is no network, no KITTI data and no pretrained checkpoint in the build environment
(SURVEY.md section 0), and the reference's own ``init_weights`` (N(0, 0.005^2), src/model/
squeezedet.py:89-97) collapses every score to exactly 1/6.  These generators give
Kaiming-scale weights whose outputs span the full score range.  Everything is drawn from
``numpy.random.RandomState`` (bit-stable across numpy / torch versions) so committed golden
vectors stay reproducible.
"""
from __future__ import annotations

import numpy as np
import torch

from .boxes import prepare_annotations


def layer_table(arch):
    """(kind, ...) per position of ``SqueezeDetBase.features`` (src/model/squeezedet.py:33-67)."""
    if arch == 'squeezedet':
        return [('conv', 3, 64, 3, 2, 1), ('relu',), ('pool',),
                ('fire', 64, 16, 64, 64), ('fire', 128, 16, 64, 64), ('pool',),
                ('fire', 128, 32, 128, 128), ('fire', 256, 32, 128, 128), ('pool',),
                ('fire', 256, 48, 192, 192), ('fire', 384, 48, 192, 192),
                ('fire', 384, 64, 256, 256), ('fire', 512, 64, 256, 256),
                ('fire', 512, 96, 384, 384), ('fire', 768, 96, 384, 384)]
    if arch == 'squeezedetplus':
        return [('conv', 3, 96, 7, 2, 3), ('relu',), ('pool',),
                ('fire', 96, 96, 64, 64), ('fire', 128, 96, 64, 64), ('fire', 128, 192, 128, 128),
                ('pool',),
                ('fire', 256, 192, 128, 128), ('fire', 256, 288, 192, 192),
                ('fire', 384, 288, 192, 192), ('fire', 384, 384, 256, 256), ('pool',),
                ('fire', 512, 384, 256, 256), ('fire', 512, 384, 256, 256), ('fire', 512, 384, 256, 256)]
    raise ValueError('Invalid architecture.')


def convdet_in_channels(arch):
    return 768 if arch == 'squeezedet' else 512


def make_state_dict(arch='squeezedet', seed=1234, anchors_per_grid=9, num_classes=3):
    """Synthetic checkpoint ``state_dict`` with the reference's keys and OIHW fp32 shapes.

    Backbone convs: N(0, 2/fan_in) (Kaiming normal), biases N(0, 0.05^2); ConvDet: zero-mean rows
    with per-output-type gains and N(0, 0.3^2) biases (confidence bias -2), tuned so that scores
    span 0..0.95, the top-64 mixes classes and class-wise NMS has real work to do."""
    rs = np.random.RandomState(seed)
    sd = {}

    def conv(name, co, ci, k, gain=2.0):
        fan_in = ci * k * k
        sd[name + '.weight'] = torch.from_numpy(
            (rs.standard_normal((co, ci, k, k)) * np.sqrt(gain / fan_in)).astype(np.float32))
        sd[name + '.bias'] = torch.from_numpy((rs.standard_normal(co) * 0.05).astype(np.float32))

    for i, l in enumerate(layer_table(arch)):
        if l[0] == 'conv':
            conv(f'base.features.{i}', l[2], l[1], l[3])
        elif l[0] == 'fire':
            _, ci, s, e1, e3 = l
            conv(f'base.features.{i}.squeeze', s, ci, 1)
            conv(f'base.features.{i}.expand1x1', e1, s, 1)
            conv(f'base.features.{i}.expand3x3', e3, s, 3)
    cout = anchors_per_grid * (num_classes + 5)
    conv('base.convdet', cout, convdet_in_channels(arch), 3, gain=1.0)
    # ConvDet rows: zero-mean (so the constant part of the post-ReLU features does not turn into a
    # per-anchor offset that would put the whole top-k in one anchor/class), then per-type gains:
    # class logits x2, confidence x1.5, box deltas x0.3 (|delta| ~ 0.4 like a trained net).
    w = sd['base.convdet.weight']
    w -= w.mean(dim=(1, 2, 3), keepdim=True)
    wv = w.view(anchors_per_grid, num_classes + 5, -1)
    wv[:, :num_classes] *= 2.0
    wv[:, num_classes] *= 1.5
    wv[:, num_classes + 1:] *= 0.3
    b = sd['base.convdet.bias'].view(anchors_per_grid, num_classes + 5)
    b[:] = torch.from_numpy((rs.standard_normal((anchors_per_grid, num_classes + 5)) * 0.3).astype(np.float32))
    b[:, num_classes] += -2.0          # most anchors below the 0.3 score threshold
    return sd


def make_images(batch, input_size, seed=0):
    """Whitened-image-like fp32 NCHW batch (unit variance per image, cf. src/utils/image.py:17)
    with spatial structure: white noise plus blocky noise at 8/32/96-pixel scales, so the
    feature map -- and hence the detections -- vary over the image like a real scene."""
    rs = np.random.RandomState(seed)
    h, w = input_size
    img = np.zeros((batch, 3, h, w), dtype=np.float32)
    for scale, weight in ((1, 0.5), (8, 1.0), (32, 1.0), (96, 1.0)):
        n = rs.standard_normal((batch, 3, -(-h // scale), -(-w // scale))).astype(np.float32)
        if scale > 1:
            n = np.repeat(np.repeat(n, scale, axis=2), scale, axis=3)[:, :, :h, :w]
        img += weight * n
    img /= img.reshape(batch, -1).std(axis=1).reshape(batch, 1, 1, 1)
    return torch.from_numpy(img)


def make_gt_boxes(batch, input_size, num_classes=3, seed=1, min_boxes=3, max_boxes=8):
    """Random sparse annotations: per-image lists (class_ids [n], xyxy boxes [n,4] float32); uniform centres,
    log-uniform sizes 20..min(400, dim/2) px, uniform class."""
    rs = np.random.RandomState(seed)
    h, w = input_size
    cls_list, box_list = [], []
    for b in range(batch):
        n = rs.randint(min_boxes, max_boxes + 1)
        hi_w, hi_h = min(400., w / 2.), min(400., h / 2.)
        bw = np.exp(rs.uniform(np.log(min(20., hi_w / 2)), np.log(hi_w), n))
        bh = np.exp(rs.uniform(np.log(min(20., hi_h / 2)), np.log(hi_h), n))
        cx = rs.uniform(0, w - 1, n)
        cy = rs.uniform(0, h - 1, n)
        x1 = np.clip(cx - bw / 2, 0, w - 2); x2 = np.clip(cx + bw / 2, x1 + 1, w - 1)
        y1 = np.clip(cy - bh / 2, 0, h - 2); y2 = np.clip(cy + bh / 2, y1 + 1, h - 1)
        box_list.append(np.stack([x1, y1, x2, y2], 1).astype(np.float32))
        cls_list.append(rs.randint(0, num_classes, n))
    return cls_list, box_list


def make_gt(batch, anchors, input_size, num_classes=3, seed=1, min_boxes=3, max_boxes=8):
    """Dense gt ``[B, A, C+9]`` from ``make_gt_boxes``, encoded on the host with ``prepare_annotations``."""
    cls_list, box_list = make_gt_boxes(batch, input_size, num_classes, seed, min_boxes, max_boxes)
    out = np.zeros((batch, anchors.shape[0], num_classes + 9), dtype=np.float32)
    for b in range(batch):
        out[b] = prepare_annotations(cls_list[b], box_list[b], anchors, num_classes)
    return torch.from_numpy(out)
