"""``DetectStream``: the execution mode of the inference driver (reference: ``Detector.detect_dataset``, src/engine/detector.py:52-85 --
a ``DataLoader`` feeding ``detect`` one batch at a time, each batch blocking on per-image ``.cpu()`` copies, :37).

Here a queue of batches runs over ``lanes`` (default 2) independent *lanes*.  A lane owns a compute stream, the device-side staging
buffer of the raw uint8 pixels, the network input, one packed result buffer and -- per batch size it has seen twice -- a captured
hipGraph of its step (backbone -> fused detect; ``preprocess_kernel`` runs in front of it as the one eager launch, so that the
raw-pixel buffer is released to the next upload the moment it has been read).  Consecutive batches go to consecutive lanes, so
the serial tail of batch i (the last round of every persistent kernel, the 160-workgroup detect launch) overlaps the head of
batch i + 1.  Around the lanes:

  * raw pixels are packed by any thread (``Staging.put``) into one of ``lanes + 1`` pinned host buffers -- header (byte offsets +
    sizes of the images) and pixels in ONE allocation -- and leave with ONE host-to-device copy on a copy stream;
  * the five result tensors of a batch are views of ONE allocation (``ops.det_buffers_packed``) and come back with ONE
    device-to-host copy on a third stream into one of two pinned slots per lane;
  * results are handed out late (``fetch`` / ``run``): the host only ever waits for the OLDEST batch, while up to ``2 * lanes - 1``
    younger ones are queued on the device, so it never blocks the lane it is about to refill.

A batch shape a lane sees for the first time runs as eager launches on the lane's stream (which also warms the allocator);
the second time it is captured; afterwards it replays.  The ragged last batch of a dataset therefore costs no capture.  Graph
replay, eager launches and ``Detector.detect_images`` produce the same bits (tests/test_lanes_gpu.py).  A failed capture keeps
that shape on eager launches and sets ``degraded`` (bench.py reports it next to ``value``).
"""
from __future__ import annotations

import ctypes
from collections import deque

import numpy as np
import torch

from . import _native as nat
from . import ops
from .preprocess import KITTI_RGB_MEAN, KITTI_RGB_STD

_HDR_ALIGN = 256


def _header_bytes(cap):
    """Bytes in front of the pixels: int64 offsets [cap] then int32 sizes [cap][2], rounded up to 256."""
    return -(-(16 * cap) // _HDR_ALIGN) * _HDR_ALIGN


def streams_alias(a, b, spin_us=500):
    """Whether work enqueued on stream ``b`` waits behind work on stream ``a`` -- i.e. the two HIP streams were mapped onto the same
    hardware queue (the runtime shares a few hardware queues, 4 by default, among all streams of the process; a queue is in-order
    across every stream it carries).  Probe: a kernel that idles ``spin_us`` on ``a``, an empty one on ``b``; if ``a``'s is already
    over when ``b``'s completes, ``b`` queued behind it."""
    lib = nat.lib()
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    nat.check(lib.sqd_spin_us(int(spin_us), nat.c_p(a.cuda_stream)), 'sqd_spin_us')
    ea.record(a)
    nat.check(lib.sqd_spin_us(0, nat.c_p(b.cuda_stream)), 'sqd_spin_us')
    eb.record(b)
    eb.synchronize()
    alias = ea.query()
    ea.synchronize()
    return bool(alias)


def pick_streams(device, need, tries=40):
    """``need`` streams of ``device`` that do not share a hardware queue with each other, taken from torch's stream pool (a
    candidate that queues behind an already chosen stream in two probes out of two is skipped).  Returns (streams, distinct): when
    the process cannot get ``need`` separate queues the list is completed with streams that do alias and ``distinct`` is False."""
    torch.cuda.synchronize(device)
    chosen, spare = [], []
    with torch.cuda.device(device):
        for _ in range(tries):
            c = torch.cuda.Stream(device)
            if any(c.cuda_stream == s.cuda_stream for s in chosen + spare):
                continue
            if all(not (streams_alias(s, c) and streams_alias(s, c)) for s in chosen):
                chosen.append(c)
                if len(chosen) == need:
                    return chosen, True
            else:
                spare.append(c)
        while len(chosen) < need:
            chosen.append(spare.pop(0) if spare else torch.cuda.Stream(device))
    return chosen, False


class BatchResult:
    """Compact detections of one batch on the host (copies, not views of the pinned slot): ``count`` int32 [n], ``class_ids`` int64
    [n,K], ``scores`` [n,K], ``boxes`` [n,K,4], ``anchor_idx`` int32 [n,K]; rows >= count[b] are padding."""
    __slots__ = ('count', 'class_ids', 'scores', 'boxes', 'anchor_idx', 'meta', 'tag')

    def __init__(self, arrays, meta=None, tag=None):
        self.count, self.class_ids, self.scores, self.boxes, self.anchor_idx = arrays
        self.meta = meta
        self.tag = tag

    def per_image(self):
        """One dict per image, as ``Detector.detect_images`` returns them (``image_meta`` from the staging, if any)."""
        out = []
        for b in range(len(self.count)):
            n = int(self.count[b])
            m = dict(self.meta[b]) if self.meta is not None else {'index': b}
            if n == 0:
                out.append({'image_meta': m})
                continue
            out.append({'class_ids': self.class_ids[b, :n].copy(), 'scores': self.scores[b, :n].copy(), 'boxes': self.boxes[b, :n].copy(),
                        'anchor_idx': self.anchor_idx[b, :n].astype(np.int64), 'image_meta': m})
        return out


class Staging:
    """One batch of raw images being packed into a pinned host buffer.  ``put(b, image)`` may be called from any thread (one
    thread per slot ``b``); an image that is not uint8 HWC RGB is refused (returns False), one that does not fit its slot is kept by
    reference and re-packed by ``DetectStream.submit`` after the buffer has grown."""

    def __init__(self, owner, index, n):
        self.owner, self.index, self.n = owner, index, n
        self.sizes = np.zeros((n, 2), dtype=np.int32)
        self.filled = [False] * n
        self.overflow = {}
        self.refused = {}

    def view(self, b, h, w):
        """Writable uint8 [h, w, 3] view of slot ``b`` in the pinned buffer, for a loader / decoder that produces its pixels in place
        (no packing copy at all); None if an image of that size does not fit the slot (use ``put`` then: it grows the buffers)."""
        hb = self.owner._host[self.index]
        nbytes = int(h) * int(w) * 3
        if h < 1 or w < 1 or nbytes > hb['slot']:
            return None
        self.sizes[b] = (h, w)
        self.filled[b] = True
        off = hb['hdr'] + b * hb['slot']
        return hb['np'][off:off + nbytes].reshape(int(h), int(w), 3)

    def put(self, b, image):
        hb = self.owner._host[self.index]
        im = np.asarray(image)
        if im.ndim != 3 or im.shape[2] != 3 or im.shape[0] < 1 or im.shape[1] < 1:
            self.refused[b] = im
            return False
        if im.dtype != np.uint8:
            u8 = im.astype(np.uint8)
            if not np.array_equal(u8, im):               # (pre-whitened floats: the caller takes the reference's host route)
                self.refused[b] = im
                return False
            im = u8
        self.sizes[b] = im.shape[:2]
        self.filled[b] = True
        nbytes = im.shape[0] * im.shape[1] * 3
        if nbytes > hb['slot']:
            self.overflow[b] = im
            return True
        off = hb['hdr'] + b * hb['slot']
        hb['np'][off:off + nbytes] = np.ascontiguousarray(im).reshape(-1)
        return True


class _Lane:
    def __init__(self, device, index, stream):
        self.index = index
        self.comp = stream
        self.graphs = {}             # key -> CUDAGraph | 'eager'
        self.seen = {}               # key -> submissions so far
        self.bufs = {}               # n -> dict(img, aux, out, flat, secs, res=[pinned, pinned], copied=[Event, Event])
        self.src = None              # device staging of the raw pixels (+ header)
        self.src_cap = 0
        self.done = torch.cuda.Event()
        self.done.record(self.comp)
        self.consumed = torch.cuda.Event()
        self.consumed.record(self.comp)
        self.last_copied = None      # event of the latest result copy out of this lane's packed buffer
        self.pool = None             # memory pool shared by this lane's captured graphs
        self.uses = 0


class DetectStream:
    """See the module docstring.  ``detector``: a ``Detector``; ``lanes``: batches in flight on the device; ``graph=False`` keeps every
    batch on eager launches; ``rgb_mean`` / ``rgb_std``: whitening statistics of the raw-image path (default ``cfg``'s, else KITTI's)."""

    def __init__(self, detector, lanes=2, graph=True, rgb_mean=None, rgb_std=None, capacity=None):
        cfg = detector.cfg
        self.det, self.cfg = detector, cfg
        self.device = torch.device(cfg.device)
        if self.device.type != 'cuda':
            raise RuntimeError('DetectStream runs on the MI355X HIP kernels only (cfg.device must be a CUDA/HIP device)')
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        if lanes < 1:
            raise ValueError('DetectStream: lanes must be >= 1')
        self.graph = bool(graph)
        self.max_ragged_graphs_per_lane = 1    # captured shapes per lane besides the full batch size
        self.max_graphs_per_lane = 8           # captured steps a lane keeps (distinct input tensors of submit_device count separately)
        self.degraded = False
        self.captures = 0
        self.eager_batches = 0
        self.replayed_batches = 0
        self.forbid = bool(getattr(cfg, 'forbid_resize', False))
        mean = rgb_mean if rgb_mean is not None else getattr(cfg, 'rgb_mean', None)
        std = rgb_std if rgb_std is not None else getattr(cfg, 'rgb_std', None)
        mean = KITTI_RGB_MEAN if mean is None else np.asarray(mean, np.float32).reshape(-1)
        std = KITTI_RGB_STD if std is None else np.asarray(std, np.float32).reshape(-1)
        self.rgb_mean, self.rgb_std = mean, std
        self._mean_c = (ctypes.c_float * 3)(*[float(v) for v in mean])
        self._std_c = (ctypes.c_float * 3)(*[float(v) for v in std])
        self.cap = int(capacity if capacity is not None else max(1, int(getattr(cfg, 'batch_size', 20))))
        H, W = int(cfg.input_size[0]), int(cfg.input_size[1])
        self._slot = -(-(H * W * 3) // 4096) * 4096          # first guess: an original the size of the network input
        # lanes + upload + download each on a hardware queue of their own: two of them on one queue serialise behind each other's
        # barrier packets (measured: 1.46 ms per batch of 20 with separate queues, 2.24 ms when the copy stream shares a lane's)
        streams, self.queues_distinct = pick_streams(self.device, lanes + 2)
        if not self.queues_distinct:
            # fewer hardware queues than streams (4 per process by default): give up the download stream first -- the packed
            # results then leave on the lane's own stream, behind its step -- and keep lanes and upload apart
            streams, self.queues_distinct = pick_streams(self.device, lanes + 1)
            streams.append(None)
        with torch.cuda.device(self.device):
            self._lanes = [_Lane(self.device, i, streams[i]) for i in range(lanes)]
            self._copy, self._back = streams[lanes], streams[lanes + 1]
        self._host = [None] * (lanes + 1)                    # pinned staging buffers, round robin
        self._next_host = 0
        self._open = set()
        self._next_lane = 0
        self._outstanding = deque()                          # tickets in submission order
        self._ready = deque()                                # results fetched early (to free a result slot), not yet handed out
        self._seq = 0
        self._eager_done = None                              # (event, lane index) of the latest eager run: see _launch
        self._weights = self._weights_signature()

    # ------------------------------------------------------------------------------------------------------------------
    # raw-image path
    # ------------------------------------------------------------------------------------------------------------------
    def _alloc_host(self, index, cap, slot):
        hdr = _header_bytes(cap)
        t = torch.empty(hdr + cap * slot, dtype=torch.uint8, pin_memory=True)
        ev = torch.cuda.Event()
        self._host[index] = {'t': t, 'np': t.numpy(), 'hdr': hdr, 'slot': slot, 'cap': cap, 'uploaded': ev, 'used': False}

    def stage(self, n):
        """A ``Staging`` for the next batch of ``n`` raw images: the next pinned buffer of the ring, once its previous upload has left."""
        if n < 1:
            raise ValueError('DetectStream.stage: empty batch')
        i = self._next_host
        if i in self._open:
            raise RuntimeError('DetectStream.stage: every staging buffer is open (submit one first)')
        self._next_host = (i + 1) % len(self._host)
        hb = self._host[i]
        if hb is not None and hb['used']:
            hb['uploaded'].synchronize()
        if hb is None or hb['cap'] < max(n, self.cap) or hb['slot'] < self._slot:
            self._alloc_host(i, max(n, self.cap), self._slot)
        self._open.add(i)
        return Staging(self, i, n)

    def submit(self, st, image_ids=None, tag=None):
        """Enqueue a filled ``Staging``: header -> ONE H2D copy -> the lane's step -> ONE D2H copy.  Returns the ticket number."""
        if st.refused:
            self._open.discard(st.index)
            raise ValueError(f'DetectStream.submit: images {sorted(st.refused)} are not uint8-representable [H,W,3] pixels')
        if not all(st.filled):
            self._open.discard(st.index)
            raise ValueError('DetectStream.submit: staging has empty slots')
        n = st.n
        hb = self._host[st.index]
        if st.overflow:
            # an image larger than its slot: grow every staging buffer to the new slot size (rare: once per dataset), re-pack this one
            need = max(im.shape[0] * im.shape[1] * 3 for im in st.overflow.values())
            self._slot = -(-need // 4096) * 4096
            old = hb
            self._alloc_host(st.index, max(n, self.cap), self._slot)
            hb = self._host[st.index]
            for b in range(n):
                nbytes = int(st.sizes[b, 0]) * int(st.sizes[b, 1]) * 3
                dst = hb['hdr'] + b * hb['slot']
                if b in st.overflow:
                    hb['np'][dst:dst + nbytes] = np.ascontiguousarray(st.overflow[b]).reshape(-1)
                else:
                    so = old['hdr'] + b * old['slot']
                    hb['np'][dst:dst + nbytes] = old['np'][so:so + nbytes]
        cap, hdr, slot = hb['cap'], hb['hdr'], hb['slot']
        head = hb['np'][:hdr]
        head[:8 * cap].view(np.int64)[:n] = hdr + slot * np.arange(n, dtype=np.int64)
        head[8 * cap:16 * cap].view(np.int32).reshape(cap, 2)[:n] = st.sizes
        total = hdr + n * slot
        lane = self._take_lane()
        if lane.src is None or lane.src.numel() < hb['t'].numel() or lane.src_cap != cap:
            if lane.src is not None:
                torch.cuda.synchronize(self.device)         # (growth, rare: nothing may still read the buffer that is replaced)
            lane.src = torch.empty(hb['t'].numel(), dtype=torch.uint8, device=self.device)
            lane.src_cap = cap
        with torch.cuda.stream(self._copy):
            self._copy.wait_event(lane.consumed)              # the lane's previous batch has been pre-processed out of lane.src
            lane.src[:total].copy_(hb['t'][:total], non_blocking=True)
            hb['uploaded'].record(self._copy)
            hb['used'] = True
            uploaded = hb['uploaded']
        self._open.discard(st.index)
        nb = self._bufs(lane, n)
        key = ('u8', n, self.forbid)
        lane.comp.wait_event(uploaded)
        # the pre-processing kernel is launched eagerly in front of the captured step: the lane's raw-pixel buffer is free again as
        # soon as THAT kernel is through (``lane.consumed``), so the next upload into it overlaps the whole network
        with torch.cuda.stream(lane.comp):
            src = lane.src
            d_off = nat.c_p(src.data_ptr())
            d_sizes = nat.c_p(src.data_ptr() + 8 * cap)
            H, W = int(self.cfg.input_size[0]), int(self.cfg.input_size[1])
            if self.forbid:
                rc = nat.lib().sqd_preprocess_u8_padcrop_fwd(nat.ptr(src), d_off, d_sizes, nat.ptr(nb['img']), nat.ptr(nb['aux']), None,
                                                             self._mean_c, self._std_c, n, H, W, nat.stream_handle(self.device))
                nat.check(rc, 'sqd_preprocess_u8_padcrop_fwd')
            else:
                rc = nat.lib().sqd_preprocess_u8_fwd(nat.ptr(src), d_off, d_sizes, nat.ptr(nb['img']), nat.ptr(nb['aux']),
                                                     self._mean_c, self._std_c, n, H, W, nat.stream_handle(self.device))
                nat.check(rc, 'sqd_preprocess_u8_fwd')
            lane.consumed.record(lane.comp)

        def compute():
            if self.forbid:
                self.det.detect_device(nb['img'], shifts=nb['aux'], out=nb['out'])
            else:
                self.det.detect_device(nb['img'], scales=nb['aux'], out=nb['out'])
        meta = self._image_meta(st.sizes, image_ids)
        return self._launch(lane, nb, key, compute, meta, tag)

    def submit_images(self, images, image_ids=None, tag=None):
        """Convenience: pack a list of uint8 HWC images (any sizes) on the calling thread + the packing pool and submit them."""
        from .preprocess import _pack_pool
        st = self.stage(len(images))
        if len(images) >= 4:
            list(_pack_pool().map(lambda b: st.put(b, images[b]), range(len(images))))
        else:
            for b, im in enumerate(images):
                st.put(b, im)
        return self.submit(st, image_ids=image_ids, tag=tag)

    def _image_meta(self, sizes, image_ids):
        """The per-image ``image_meta`` of the GPU input pipeline (the keys ``Detector.detect_images`` hands out)."""
        H, W = int(self.cfg.input_size[0]), int(self.cfg.input_size[1])
        metas = []
        for b, (h0, w0) in enumerate(sizes):
            m = {'orig_size': np.array([int(h0), int(w0), 3], dtype=np.int32), 'index': b,
                 'image_id': image_ids[b] if image_ids is not None else str(b)}
            if self.forbid:
                pad, crop = np.zeros(4, np.int16), np.zeros(4, np.int16)       # the same integers as the kernel (image.py:99-115)
                for size, target, k in ((int(h0), H, 0), (int(w0), W, 2)):
                    if size < target:
                        pad[k] = (target - size) // 2; pad[k + 1] = (target - size) - pad[k]
                    elif size > target:
                        crop[k] = (size - target) // 2; crop[k + 1] = (size - target) - crop[k]
                m.update(padding=pad, crops=crop)
            else:
                m['scales'] = np.array([H / h0, W / w0], dtype=np.float32)
            metas.append(m)
        return metas

    # ------------------------------------------------------------------------------------------------------------------
    # device-resident path
    # ------------------------------------------------------------------------------------------------------------------
    def submit_device(self, image, scales=None, shifts=None, meta=None, tag=None):
        """Enqueue one batch that is already on the GPU (fp32 NCHW, as ``Detector.detect_device`` takes it).  The captured step reads
        ``image`` / ``scales`` / ``shifts`` where they are: a caller that refills the same tensors replays, new tensors run eagerly
        (and are captured on their second use).  The lane reads the tensors asynchronously: do not overwrite them before the batch
        has been fetched (or use one set of tensors per lane)."""
        if not (isinstance(image, torch.Tensor) and image.is_cuda and image.dtype == torch.float32 and image.dim() == 4 and image.is_contiguous()):
            raise ValueError('DetectStream.submit_device: image must be a contiguous fp32 CUDA/HIP tensor [B,3,H,W]')
        n = image.shape[0]
        lane = self._take_lane()
        nb = self._bufs(lane, n, with_input=False)
        key = ('dev', n, image.data_ptr(), tuple(image.shape), None if scales is None else scales.data_ptr(),
               None if shifts is None else shifts.data_ptr())
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != lane.comp.cuda_stream and not cur.query():
            # whatever is still producing the tensors on the caller's stream.  Skipped when that stream is idle: the marker of an
            # unconditional wait sits in a hardware queue the caller's stream shares with one of the lanes and would serialise them
            lane.comp.wait_stream(cur)

        def compute():
            self.det.detect_device(image, scales=scales, shifts=shifts, out=nb['out'])
        return self._launch(lane, nb, key, compute, meta, tag)

    # ------------------------------------------------------------------------------------------------------------------
    # lanes
    # ------------------------------------------------------------------------------------------------------------------
    def _take_lane(self):
        # keep the host at most 2 * lanes - 1 batches ahead: the result slot this submission will use must have been read
        while len(self._outstanding) >= 2 * len(self._lanes):
            self._ready.append(self._fetch_oldest())
        lane = self._lanes[self._next_lane]
        self._next_lane = (self._next_lane + 1) % len(self._lanes)
        return lane

    def _bufs(self, lane, n, with_input=True):
        nb = lane.bufs.get(n)
        K = int(self.cfg.keep_top_k)
        if nb is None:
            with torch.cuda.stream(lane.comp):
                out, flat = ops.det_buffers_packed(n, K, self.device, int(self.cfg.num_anchors))
            secs, total = ops.det_packed_layout(n, K)
            nb = {'out': out, 'flat': flat, 'secs': secs, 'img': None, 'aux': None,
                  'res': [torch.empty(total, dtype=torch.uint8, pin_memory=True) for _ in range(2)],
                  'copied': [torch.cuda.Event(), torch.cuda.Event()], 'uses': 0}
            lane.bufs[n] = nb
        if with_input and nb['img'] is None:
            H, W = int(self.cfg.input_size[0]), int(self.cfg.input_size[1])
            with torch.cuda.stream(lane.comp):
                nb['img'] = torch.empty(n, 3, H, W, device=self.device, dtype=torch.float32)
                nb['aux'] = torch.empty(n, 2, device=self.device, dtype=torch.float32)
        return nb

    def _launch(self, lane, nb, key, compute, meta, tag):
        sig = self._weights_signature()
        if sig != self._weights:
            # the parameters changed since the graphs were captured (an optimizer step, load_state_dict): the captured steps hold
            # packed-weight buffers that are rebuilt, not all in place -- drop them; every shape is re-captured on its second use
            torch.cuda.synchronize(self.device)
            for ln in self._lanes:
                ln.graphs.clear(); ln.seen.clear()
            self._weights = sig
        if self._eager_done is not None and self._eager_done[1] != lane.index:
            # an eager run may have (re)packed weights or created workspaces on ITS lane's stream: order this lane behind it.  Only
            # first-time shapes and ragged batches run eagerly, so the steady state never takes this edge
            lane.comp.wait_event(self._eager_done[0])
        if lane.last_copied is not None:
            lane.comp.wait_event(lane.last_copied)            # the previous results have left the lane's packed buffer
        seen = lane.seen.get(key, 0)
        lane.seen[key] = seen + 1
        g = lane.graphs.get(key)
        if g is None and self.graph and seen >= 1:
            # the full batch size is always captured, of other (ragged) sizes only the first one a lane sees twice: a stream of many
            # different ragged batch sizes stays on eager launches instead of capturing (a device-wide synchronisation each) for ever.
            # A lane's graphs share one memory pool; their number is bounded as well (a caller of submit_device that keeps handing in
            # new tensors would otherwise collect one graph per tensor): the oldest one goes
            others = sum(1 for k2, v in lane.graphs.items() if v != 'eager' and k2[1] != self.cap)
            if key[1] == self.cap or others < self.max_ragged_graphs_per_lane:
                captured = [k2 for k2, v in lane.graphs.items() if v != 'eager']
                if len(captured) >= self.max_graphs_per_lane:
                    torch.cuda.synchronize(self.device)           # (nothing may still be replaying the graph that is dropped)
                    del lane.graphs[captured[0]]
                    lane.seen.pop(captured[0], None)
                g = self._capture(lane, compute)
                lane.graphs[key] = g
        if len(lane.seen) > 4096:                                 # (keys of tensors seen once and never again)
            lane.seen = {k2: v for k2, v in lane.seen.items() if k2 in lane.graphs}
            lane.seen[key] = seen + 1
        with torch.cuda.stream(lane.comp):
            if g is not None and g != 'eager':
                g.replay()
                self.replayed_batches += 1
            else:
                with torch.no_grad():
                    compute()
                self.eager_batches += 1
                ev = torch.cuda.Event()
                ev.record(lane.comp)
                self._eager_done = (ev, lane.index)
            lane.done.record(lane.comp)
        r = nb['uses'] & 1
        nb['uses'] += 1
        back = self._back if self._back is not None else lane.comp
        with torch.cuda.stream(back):                         # compact results -> pinned memory on a stream of their own: neither
            back.wait_event(lane.done)                        # the lanes nor the uploads wait for the copy
            nb['res'][r].copy_(nb['flat'], non_blocking=True)
            nb['copied'][r].record(back)
        lane.last_copied = nb['copied'][r]
        self._seq += 1
        self._outstanding.append((self._seq, nb, r, meta, tag))
        return self._seq

    def _weights_signature(self):
        return tuple((p._version, p.data_ptr()) for p in self.det.model.parameters())

    def discard(self, st):
        """Give an un-submitted ``Staging`` back (its batch takes another route)."""
        self._open.discard(st.index)

    def _capture(self, lane, compute):
        """One hipGraph of the lane's step for this key, captured on the lane's stream (its eager run has already sized the pools);
        'eager' (and ``degraded``) if the capture fails."""
        try:
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            # thread_local: loader threads, a collective library's watchdog ... may touch the runtime while this thread captures
            if lane.pool is None:
                lane.pool = torch.cuda.graph_pool_handle()
            # one memory pool per LANE: its graphs never run concurrently (one stream) and keep nothing alive between replays (the result
            # buffers live outside), so a second shape or a second input tensor reuses the first graph's activation memory
            with torch.cuda.graph(g, stream=lane.comp, pool=lane.pool, capture_error_mode='thread_local'):
                with torch.no_grad():
                    compute()
            self.captures += 1
            return g
        except Exception as e:  # noqa: BLE001 -- the batch still runs, on eager launches; the caller can see it in .degraded
            import sys
            print(f'[DetectStream] lane {lane.index}: step not captured ({type(e).__name__}: {e}); eager launches', file=sys.stderr)
            torch.cuda.synchronize(self.device)
            self.degraded = True
            return 'eager'

    # ------------------------------------------------------------------------------------------------------------------
    # results
    # ------------------------------------------------------------------------------------------------------------------
    def _fetch_oldest(self):
        seq, nb, r, meta, tag = self._outstanding.popleft()
        nb['copied'][r].synchronize()
        host = nb['res'][r].numpy()
        arrays = []
        for off, cnt, dt, shp in nb['secs']:
            nbytes = cnt * torch.empty(0, dtype=dt).element_size()
            arrays.append(host[off:off + nbytes].view(_NP[dt]).reshape(shp).copy())
        return seq, BatchResult(tuple(arrays), meta, tag)

    def pending(self):
        """Batches submitted and not yet fetched."""
        return len(self._outstanding) + (len(self._ready) if self._ready else 0)

    def oldest_ready(self):
        """Whether ``fetch`` would return without waiting."""
        if self._ready:
            return True
        if not self._outstanding:
            return False
        _seq, nb, r, _m, _t = self._outstanding[0]
        return nb['copied'][r].query()

    def fetch(self):
        """(ticket, BatchResult) of the OLDEST un-fetched batch; waits for its result copy only."""
        if self._ready:
            return self._ready.popleft()
        if not self._outstanding:
            raise RuntimeError('DetectStream.fetch: nothing submitted')
        return self._fetch_oldest()

    def drain(self):
        """Every remaining result, in submission order."""
        out = []
        while self.pending():
            out.append(self.fetch())
        return out

    def run(self, batches, image_ids=None):
        """Generator over an iterable of batches (each a list of uint8 HWC images, or a ``(images, image_ids)`` pair): yields, in
        order and as late as the queue allows, one list of per-image result dicts per batch."""
        for item in batches:
            images, ids = item if (isinstance(item, tuple) and len(item) == 2 and not isinstance(item[0], np.ndarray)) else (item, None)
            self.submit_images(list(images), image_ids=ids)
            while self.pending() > 2 * len(self._lanes) - 1 or (self.pending() and self.oldest_ready()):
                yield self.fetch()[1].per_image()
        while self.pending():
            yield self.fetch()[1].per_image()


_NP = {torch.int32: np.int32, torch.int64: np.int64, torch.float32: np.float32}
