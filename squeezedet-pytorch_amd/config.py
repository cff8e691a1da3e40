"""The ``cfg`` fields the hot path reads, with the reference's defaults.

The reference threads one mutable ``argparse.Namespace`` through every layer
(src/utils/config.py:5-131, src/utils/misc.py:14).  This module does not rebuild its CLI;
it only produces a namespace carrying exactly the fields the model / detector / trainer read
(SURVEY.md section 8b), so the mirrored classes accept either this or the reference's own cfg.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from .boxes import KITTI_ANCHORS_SEED, KITTI_INPUT_SIZE, generate_anchors


def make_cfg(arch='squeezedet', input_size=KITTI_INPUT_SIZE, anchors_seed=KITTI_ANCHORS_SEED,
             num_classes=3, class_names=('Car', 'Pedestrian', 'Cyclist'), device='cuda', **overrides):
    """Namespace with the reference's defaults (src/utils/config.py:23-85) plus the
    dataset-derived fields of ``Config.update_dataset_info`` (:121-131)."""
    grid_size = tuple(x // 16 for x in input_size)           # src/datasets/kitti.py:26
    anchors = generate_anchors(grid_size, input_size, np.asarray(anchors_seed))
    cfg = SimpleNamespace(
        mode='eval', arch=arch, dropout_prob=0.5,
        lr=0.01, momentum=0.9, weight_decay=1e-4, grad_norm=5., batch_size=20,
        class_loss_weight=1., positive_score_loss_weight=3.75,
        negative_score_loss_weight=100., bbox_loss_weight=6.,
        nms_thresh=0.4, score_thresh=0.3, keep_top_k=64,
        gpus=[0], chunk_sizes=[20], num_iters=-1, print_interval=10, debug=0, num_workers=4, forbid_resize=False,
        inflight=2,                  # batches in flight on the device in Detector.stream / detect_dataset (lanes.DetectStream)
        input_size=tuple(input_size), num_classes=num_classes, class_names=tuple(class_names),
        anchors=anchors, anchors_per_grid=int(np.asarray(anchors_seed).shape[0]),
        num_anchors=int(anchors.shape[0]), grid_size=grid_size, device=device,
    )
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg
