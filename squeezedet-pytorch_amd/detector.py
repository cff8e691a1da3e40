"""``Detector`` with the reference's interface (src/engine/detector.py:14-122): ``detect(batch)``
returns one dict per image (numpy arrays, boxes mapped back to original-image coordinates),
``filter(det)`` filters one image's dense detections.

Where the reference loops over the batch in Python with >= 10 host syncs per image (boolean-mask
indexing, ``torch.sum(..) == 0``, per-image ``.cpu()``), this runs ONE fused kernel for the whole
batch (decode -> top-k -> class-wise NMS -> threshold -> box un-scaling) straight from ``pred`` and
ONE device-to-host copy of the compact result.  Each result additionally carries ``anchor_idx``.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .boxes import boxes_postprocess


class Detector(object):
    def __init__(self, model, cfg):
        self.model = model.to(cfg.device)
        self.model.eval()
        self.cfg = cfg

    # ---- device-side batched path (no host sync) ----
    @torch.no_grad()
    def detect_device(self, image, scales=None, out=None):
        """image [B,3,H,W] on the GPU -> (count [B] i32, class_ids [B,K] i64, scores [B,K], boxes [B,K,4],
        anchor_idx [B,K] i32), all on the GPU; rows >= count[b] are padding."""
        cfg = self.cfg
        pred = self.model.base(image)
        anchors = self.model.resolver.anchors_on(pred.device)
        return ops.detect(pred, anchors, cfg.input_size, cfg.num_classes, cfg.keep_top_k, cfg.nms_thresh,
                          cfg.score_thresh, scales=scales, out=out)

    @torch.no_grad()
    def detect(self, batch):
        image = batch['image']
        B = image.shape[0]
        meta = batch.get('image_meta', {})
        metas = [{k: (v[b].cpu().numpy() if isinstance(v, torch.Tensor) else
                      (np.asarray(v[b]) if isinstance(v, np.ndarray) else v[b])) for k, v in meta.items()}
                 for b in range(B)]
        # fold the eval-time scale division (boxes_postprocess, src/utils/boxes.py:145-147) into the kernel
        # when that is the only active transform; otherwise post-process on the host like the reference
        simple = all(set(m.keys()) <= {'scales', 'index', 'image_id', 'orig_size', 'rgb_mean', 'rgb_std', 'drifts',
                                       'drifted_size', 'flipped'}
                     and not m.get('flipped', False) and not np.any(m.get('drifts', 0)) for m in metas)
        scales = None
        if simple and all('scales' in m for m in metas):
            scales = torch.tensor(np.stack([np.asarray(m['scales'], dtype=np.float32) for m in metas]),
                                  device=image.device, dtype=torch.float32)
        cnt, cls, sc, bx, idx = self.detect_device(image, scales=scales)
        cnt, cls, sc, bx, idx = (t.cpu().numpy() for t in (cnt, cls, sc, bx, idx))   # one D2H round
        results = []
        for b in range(B):
            n = int(cnt[b])
            if n == 0:
                results.append({'image_meta': metas[b]})
                continue
            det = {'class_ids': cls[b, :n].copy(), 'scores': sc[b, :n].copy(), 'boxes': bx[b, :n].copy(),
                   'anchor_idx': idx[b, :n].astype(np.int64)}
            if scales is None:
                det['boxes'] = boxes_postprocess(det['boxes'], metas[b])
            det['image_meta'] = metas[b]
            results.append(det)
        return results

    @torch.no_grad()
    def detect_images(self, images, image_ids=None):
        """Raw path: list of uint8 HWC RGB images (any sizes) -> detections in original-image coordinates.
        Upload of the uint8 pixels, GPU pre-processing (whiten + resize + CHW), backbone, fused detection with the
        per-image scale division, one compact D2H copy."""
        from .preprocess import preprocess_batch
        cfg = self.cfg
        mean = getattr(cfg, 'rgb_mean', None)
        std = getattr(cfg, 'rgb_std', None)
        kw = {} if mean is None or std is None else {'rgb_mean': np.asarray(mean).reshape(-1), 'rgb_std': np.asarray(std).reshape(-1)}
        image, scales, meta = preprocess_batch(images, cfg.input_size, device=cfg.device, **kw)
        cnt, cls, sc, bx, idx = (t.cpu().numpy() for t in self.detect_device(image, scales=scales))
        results = []
        for b in range(len(images)):
            n = int(cnt[b])
            m = {'orig_size': meta['orig_size'][b], 'scales': meta['scales'][b], 'index': b,
                 'image_id': image_ids[b] if image_ids is not None else str(b)}
            if n == 0:
                results.append({'image_meta': m})
                continue
            results.append({'class_ids': cls[b, :n].copy(), 'scores': sc[b, :n].copy(), 'boxes': bx[b, :n].copy(),
                            'anchor_idx': idx[b, :n].astype(np.int64), 'image_meta': m})
        return results

    def filter(self, det):
        """One image's dense ``{'class_ids' [A], 'scores' [A], 'boxes' [A,4]}`` (GPU tensors) ->
        filtered dict of GPU tensors (plus ``anchor_idx``) or ``None``."""
        cfg = self.cfg
        cnt, cls, sc, bx, idx = ops.filter_dense(det['class_ids'][None], det['scores'][None], det['boxes'][None],
                                                 cfg.num_classes, cfg.keep_top_k, cfg.nms_thresh, cfg.score_thresh)
        n = int(cnt[0].item())
        if n == 0:
            return None
        return {'class_ids': cls[0, :n], 'scores': sc[0, :n], 'boxes': bx[0, :n], 'anchor_idx': idx[0, :n].long()}
