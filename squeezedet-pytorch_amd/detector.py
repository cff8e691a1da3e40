"""``Detector`` with the reference's interface (src/engine/detector.py:14-122): ``detect(batch)``
returns one dict per image (numpy arrays, boxes mapped back to original-image coordinates),
``filter(det)`` filters one image's dense detections, ``detect_dataset(dataset)`` is the reference's inference driver
(:52-85) on the GPU input pipeline and the lane executor (``stream`` / ``detect_stream``, lanes.py), ``DataWrapper`` its
annotation-free dataset view (:125-145).

Where the reference loops over the batch in Python with >= 10 host syncs per image (boolean-mask
indexing, ``torch.sum(..) == 0``, per-image ``.cpu()``), this runs ONE fused kernel for the whole
batch (decode -> top-k -> class-wise NMS -> threshold -> box un-scaling) straight from ``pred`` and
ONE device-to-host copy of the compact result.  Each result additionally carries ``anchor_idx``.
"""
from __future__ import annotations

import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import torch.utils.data

from . import ops
from .boxes import boxes_postprocess


class Detector(object):
    def __init__(self, model, cfg):
        self.model = model.to(cfg.device)
        self.model.eval()
        self.cfg = cfg

    # ---- device-side batched path (no host sync) ----
    @torch.no_grad()
    def detect_device(self, image, scales=None, out=None, shifts=None):
        """image [B,3,H,W] on the GPU -> (count [B] i32, class_ids [B,K] i64, scores [B,K], boxes [B,K,4],
        anchor_idx [B,K] i32), all on the GPU; rows >= count[b] are padding.  ``scales`` / ``shifts`` [B,2]: ``boxes_postprocess``'
        scale division and padding / crops terms, folded into the detect kernel."""
        cfg = self.cfg
        pred = self.model.base(image)
        anchors = self.model.resolver.anchors_on(pred.device)
        return ops.detect(pred, anchors, cfg.input_size, cfg.num_classes, cfg.keep_top_k, cfg.nms_thresh,
                          cfg.score_thresh, scales=scales, out=out, shifts=shifts)

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
                                       'drifted_size', 'flipped', 'padding', 'crops'}
                     and not m.get('flipped', False) and not np.any(m.get('drifts', 0)) for m in metas)
        scales = shifts = None
        padcrop = simple and all('padding' in m and 'crops' in m and 'scales' not in m for m in metas)
        if simple and not padcrop and all('scales' in m and 'padding' not in m and 'crops' not in m for m in metas):
            scales = torch.tensor(np.stack([np.asarray(m['scales'], dtype=np.float32) for m in metas]),
                                  device=image.device, dtype=torch.float32)
        if padcrop:
            # the forbid_resize branch (src/datasets/base.py:53-54): un-pad / un-crop = one add per axis (only one of the two is non-zero)
            shifts = torch.tensor(np.stack([[float(m['crops'][0]) - float(m['padding'][0]), float(m['crops'][2]) - float(m['padding'][2])]
                                            for m in metas]).astype(np.float32), device=image.device)
        cnt, cls, sc, bx, idx = self.detect_device(image, scales=scales, shifts=shifts)
        cnt, cls, sc, bx, idx = (t.cpu().numpy() for t in (cnt, cls, sc, bx, idx))   # one D2H round
        results = []
        for b in range(B):
            n = int(cnt[b])
            if n == 0:
                results.append({'image_meta': metas[b]})
                continue
            det = {'class_ids': cls[b, :n].copy(), 'scores': sc[b, :n].copy(), 'boxes': bx[b, :n].copy(),
                   'anchor_idx': idx[b, :n].astype(np.int64)}
            if scales is None and shifts is None:
                det['boxes'] = boxes_postprocess(det['boxes'], metas[b])
            det['image_meta'] = metas[b]
            results.append(det)
        return results

    @torch.no_grad()
    def detect_images(self, images, image_ids=None, rgb_mean=None, rgb_std=None):
        """Raw path: list of uint8 HWC RGB images (any sizes) -> detections in original-image coordinates.
        Upload of the uint8 pixels, GPU pre-processing (whiten + resize -- or, with ``cfg.forbid_resize``, whiten + crop_or_pad --
        + CHW), backbone, fused detection with the per-image scale division (or un-pad / un-crop shift), one compact D2H copy.
        ``rgb_mean`` / ``rgb_std``: the dataset's whitening statistics (default: ``cfg.rgb_mean`` / ``cfg.rgb_std``, else KITTI's)."""
        from .preprocess import preprocess_batch
        cfg = self.cfg
        mean = rgb_mean if rgb_mean is not None else getattr(cfg, 'rgb_mean', None)
        std = rgb_std if rgb_std is not None else getattr(cfg, 'rgb_std', None)
        kw = {} if mean is None or std is None else {'rgb_mean': np.asarray(mean).reshape(-1), 'rgb_std': np.asarray(std).reshape(-1)}
        forbid = bool(getattr(cfg, 'forbid_resize', False))
        image, aux, meta = preprocess_batch(images, cfg.input_size, device=cfg.device, forbid_resize=forbid, **kw)
        dets = self.detect_device(image, shifts=aux) if forbid else self.detect_device(image, scales=aux)
        cnt, cls, sc, bx, idx = (t.cpu().numpy() for t in dets)
        results = []
        for b in range(len(images)):
            n = int(cnt[b])
            m = {'orig_size': meta['orig_size'][b], 'index': b, 'image_id': image_ids[b] if image_ids is not None else str(b)}
            m.update({'padding': meta['padding'][b], 'crops': meta['crops'][b]} if forbid else {'scales': meta['scales'][b]})
            if n == 0:
                results.append({'image_meta': m})
                continue
            results.append({'class_ids': cls[b, :n].copy(), 'scores': sc[b, :n].copy(), 'boxes': bx[b, :n].copy(),
                            'anchor_idx': idx[b, :n].astype(np.int64), 'image_meta': m})
        return results

    # ---- the inference driver's execution mode: lanes of captured steps (lanes.DetectStream) ----
    def stream(self, lanes=None, graph=True, rgb_mean=None, rgb_std=None):
        """The (cached) ``lanes.DetectStream`` of this detector: ``lanes`` batches in flight on the device, each lane replaying a
        captured hipGraph of preprocess -> backbone -> fused detect; uint8 upload on a copy stream, ONE packed result copy per
        batch into pinned memory, results handed out late.  ``cfg.inflight`` (default 2) is the default lane count."""
        from .lanes import DetectStream
        lanes = int(lanes if lanes is not None else getattr(self.cfg, 'inflight', 2))
        key = (lanes, bool(graph), None if rgb_mean is None else tuple(np.asarray(rgb_mean, np.float32).reshape(-1).tolist()),
               None if rgb_std is None else tuple(np.asarray(rgb_std, np.float32).reshape(-1).tolist()),
               bool(getattr(self.cfg, 'forbid_resize', False)), int(getattr(self.cfg, 'batch_size', 20)))
        cache = self.__dict__.setdefault('_streams', {})
        ex = cache.get(key)
        if ex is None:
            if len(cache) > 4:
                cache.clear()
            ex = cache[key] = DetectStream(self, lanes=lanes, graph=graph, rgb_mean=rgb_mean, rgb_std=rgb_std)
        return ex

    def detect_stream(self, batches, lanes=None, rgb_mean=None, rgb_std=None):
        """Generator: an iterable of batches (lists of uint8 HWC RGB images of any sizes, or ``(images, image_ids)`` pairs) -> per
        batch, in order, the list of per-image result dicts ``detect_images`` would return -- bit for bit -- while the batches
        overlap on the device (see ``stream``).  Results trail the input by up to ``2 * lanes - 1`` batches."""
        ex = self.stream(lanes=lanes, rgb_mean=rgb_mean, rgb_std=rgb_std)
        if ex.pending():
            raise RuntimeError('detect_stream: the detector\'s stream still holds un-fetched batches')
        return ex.run(batches)

    def detect_dataset(self, dataset):
        """The reference's inference driver (src/engine/detector.py:52-85: DataLoader over ``DataWrapper(dataset)`` -> ``detect`` per
        batch -> timing lines -> list of per-image results) in this package's execution mode (``stream``): ``cfg.num_workers``
        threads load ``dataset.load_image(i)`` one batch ahead AND pack the raw pixels straight into the pinned staging buffer of
        the batch; the main thread only enqueues (one H2D copy, one captured step on the next lane, one D2H copy) and collects the
        results of batches that have finished, ``cfg.inflight`` (default 2) batches overlapping on the device.  Images whose pixels
        are not uint8-representable (a dataset that hands out pre-whitened floats) take the reference's own route instead:
        ``dataset.preprocess`` on the host, then ``detect``.  Same printed lines and the same result dicts
        (``image_meta['index']`` = dataset index) as the reference; identical, bit for bit, to ``detect_images`` batch by batch."""
        cfg = self.cfg
        n = len(dataset)
        bs = max(1, int(getattr(cfg, 'batch_size', 1)))
        every = max(1, int(getattr(cfg, 'print_interval', 10)))
        batches = [list(range(i, min(i + bs, n))) for i in range(0, n, bs)]
        t_start = time.time()
        results = []
        data_s = net_s = 0.0
        workers = int(getattr(cfg, 'num_workers', 4))
        ex = self.stream(rgb_mean=getattr(dataset, 'rgb_mean', None), rgb_std=getattr(dataset, 'rgb_std', None)) if batches else None
        if ex is not None and ex.pending():
            raise RuntimeError('detect_dataset: the detector\'s stream still holds un-fetched batches')

        def load_into(st, b0, idxs):
            out = []
            for b, i in enumerate(idxs, b0):
                im, iid = dataset.load_image(i)
                out.append((im, iid, st.put(b, im)))
            return out

        def collect_results(force=False):
            while ex.pending() and (force or ex.pending() > 2 * len(ex._lanes) - 1 or ex.oldest_ready()):
                _t, r = ex.fetch()
                out = r.per_image()
                for d, i in zip(out, r.tag):
                    d['image_meta']['index'] = i
                results.extend(out)

        # One future per worker and batch (a contiguous run of the batch's images each: a future per IMAGE cost the main thread ~30 us
        # apiece, 0.6 ms per 20-image batch), submitted straight to the pool -- a per-batch task that itself waits on the pool's workers
        # would deadlock a one-worker pool; num_workers = 0 (the reference's "load in the main process") loads inline.
        with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
            def begin(idxs):
                st = ex.stage(len(idxs))
                if workers > 0:
                    per = -(-len(idxs) // min(workers, len(idxs)))
                    return st, [pool.submit(load_into, st, b0, idxs[b0:b0 + per]) for b0 in range(0, len(idxs), per)]
                return st, idxs

            def finish(st, pend):
                if workers > 0:
                    return [item for f in pend for item in f.result()]
                return load_into(st, 0, pend)
            pending = begin(batches[0]) if batches else None
            try:
                for it, idxs in enumerate(batches):
                    t0 = time.time()
                    st, pend = pending
                    loaded = finish(st, pend)                          # [(image, image_id, packed), ...]
                    pending = begin(batches[it + 1]) if it + 1 < len(batches) else None
                    data_s = time.time() - t0
                    t0 = time.time()
                    ids = [iid for _, iid, _ in loaded]
                    if all(ok for _, _, ok in loaded):
                        ex.submit(st, image_ids=ids, tag=idxs)
                    else:
                        ex.discard(st)
                        collect_results(force=True)                    # keep the dataset order
                        results.extend(self.detect(_host_batch(dataset, [np.asarray(im) for im, _, _ in loaded], ids, idxs, cfg.device)))
                    collect_results()
                    net_s = time.time() - t0
                    if it % every == 0:
                        print('eval: [{0}/{1}] | data {2:.3f}s | net {3:.3f}s'.format(it, len(batches), data_s, net_s))
            except BaseException:
                if ex is not None:                                 # leave the cached executor reusable: nothing open, nothing queued
                    ex._open.clear()
                    ex.drain()
                raise
            if ex is not None:
                collect_results(force=True)
        total = time.time() - t_start
        tpi = total / max(n, 1)
        print('Elapsed {:.2f}min ({:.1f}ms/image, {:.1f}frames/s)'.format(total / 60., tpi * 1000., 1. / max(tpi, 1e-12)))
        print('-' * 80)
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


def _host_batch(dataset, images, image_ids, indices, device):
    """One ``detect`` batch built the reference's way (``DataWrapper.__getitem__`` + default collate, src/engine/detector.py:125-145):
    ``dataset.preprocess`` on the host, images stacked CHW, every ``image_meta`` field stacked per key."""
    items = [DataWrapper.item(dataset, im, iid, i) for im, iid, i in zip(images, image_ids, indices)]
    meta = {}
    for k in items[0]['image_meta']:
        vals = [it['image_meta'][k] for it in items]
        meta[k] = vals if isinstance(vals[0], str) else np.stack([np.asarray(v) for v in vals])
    return {'image': torch.from_numpy(np.stack([it['image'] for it in items])).to(device), 'image_meta': meta}


class DataWrapper(torch.utils.data.Dataset):
    """A ``Dataset`` view that bypasses the annotations (src/engine/detector.py:125-145) for callers that drive ``detect`` with
    their own ``DataLoader``: item = ``{'image': CHW float32 (the dataset's own ``preprocess``), 'image_meta': {...}}``."""

    def __init__(self, dataset):
        super().__init__()
        self.dataset = dataset

    @staticmethod
    def item(dataset, image, image_id, index):
        meta = {'index': index, 'image_id': image_id, 'orig_size': np.array(np.asarray(image).shape, dtype=np.int32)}
        image, meta, _ = dataset.preprocess(image, meta)
        return {'image': np.ascontiguousarray(np.asarray(image).transpose(2, 0, 1)), 'image_meta': meta}

    def __getitem__(self, index):
        image, image_id = self.dataset.load_image(index)
        return self.item(self.dataset, image, image_id, index)

    def __len__(self):
        return len(self.dataset)
