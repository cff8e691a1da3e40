"""Training-side plumbing of the hot path: the data-parallel gradient exchange and a ``Trainer`` with the
reference's interface.

What the reference does (src/engine/trainer.py:18-92, src/utils/data_parallel.py:46-156): ONE Python process
replicates the module onto every GPU each step, scatters the batch by ``chunk_sizes``, gathers the per-sample loss
vectors on GPU 0 and back-propagates ``loss.mean()`` of that gathered vector through the replicas.

What runs here: one process per GPU (``torch.distributed`` over RCCL / xGMI), weights replicated once, and the
gradient of that same GLOBAL mean assembled by an all-reduce that the backward itself drives:

* ``attach_data_parallel(model)`` hangs a ``GradientExchange`` on the model's ``SqueezeDetBase``.  From then on
  ``loss.backward()`` leaves the global-mean gradient in every ``p.grad`` -- the reference's own
  ``src/engine/trainer.py`` loop (``zero_grad -> backward -> clip_grad_norm_ -> step``) runs unchanged on top of it.
* The backward (``backward.py``) reports each finished slice of its flat gradient buffer -- ConvDet + the 24x78 Fire
  modules first (86 % of the parameters), then the 48x156 and the 96x312 stages -- and the exchange all-reduces that
  bucket on a side stream while the remaining layers are still being differentiated.
* Shards may be unequal: every rank scales its bucket by its local image count, one extra float at the end of the
  buffer carries that count through the same all-reduce, and the sum is divided by the global count on the device --
  exactly ``mean`` over the gathered vector (trainer.py:43) for any ``chunk_sizes``, with no extra collective and no
  host sync.  ``clip_grad_norm_`` then sees the same global norm on every rank, SGD is local and deterministic.
"""
from __future__ import annotations

import time

import torch


def _dist(group=None):
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def shard_sizes(batch_size, world):
    """Images per rank for a global batch (first ranks take the remainder): the even-split case of the
    reference's ``chunk_sizes`` (src/utils/config.py:102-110)."""
    q, r = divmod(int(batch_size), int(world))
    return [q + (i < r) for i in range(world)]


class GradientExchange:
    """All-reduce of the flat gradient buffer in buckets, overlapped with the backward that fills it.

    Driven by ``backward.run_backbone_backward``: ``begin(flat, total, local_batch)`` once the buffer exists (``flat``
    holds ``total`` gradient floats plus one trailing count slot), ``ready(lo, hi)`` whenever the slice ``[lo, hi)`` is
    final on the current stream, ``finish()`` before the gradients are handed to autograd.  Works on any backend;
    the side stream is only used for CUDA/HIP buffers."""

    def __init__(self, group=None, overlap=True):
        self.group = group
        self.overlap = overlap
        self.force = False                     # run the collectives even in a one-rank group (bench.py --force-dist)
        self._side = None
        self._flat = None
        self._pending = []
        self._captured_works = []
        self.buckets_last_step = []            # [(lo, hi)] of the latest backward, for tests / logging

    def world(self):
        d = _dist()
        return d.get_world_size(self.group) if d is not None else 1

    def begin(self, flat, total, local_batch):
        self._flat, self._total, self._b = flat, int(total), float(local_batch)
        self._pending = []
        self.buckets_last_step = []
        self._active = self.world() > 1 or (self.force and _dist() is not None)
        # weight of this rank's gradient in the global mean: the backward multiplies it in where the gradients are produced (the slab
        # reduction's ``scale``, ``scale_slice`` for the stem) -- no element-wise pass over the buckets
        self.scale = self._b if self._active else 1.0
        self._native = bool(self._active and flat.is_cuda)
        if self._active:
            if self._native:
                from . import _native as nat
                # the count slot, through the library (one thread): no torch kernel in the exchange
                nat.check(nat.lib().sqd_grad_scale(None, 0, 1.0, None, nat.c_p(flat.data_ptr() + 4 * self._total), self._b,
                                                   nat.stream_handle(flat.device)), 'sqd_grad_scale')
            else:
                flat[total:].fill_(self._b)
            if flat.is_cuda and self.overlap and self._side is None:
                self._side = torch.cuda.Stream(device=flat.device)

    def scale_slice(self, lo, hi):
        """Weight ``flat[lo:hi]`` (a gradient that did not come out of the slab reduction: the stem's) by this rank's image count."""
        if not self._active or hi <= lo:
            return
        if self._native:
            from . import _native as nat
            nat.check(nat.lib().sqd_grad_scale(nat.c_p(self._flat.data_ptr() + 4 * int(lo)), int(hi - lo), self.scale, None, None, 0.0,
                                               nat.stream_handle(self._flat.device)), 'sqd_grad_scale')
        else:
            self._flat[lo:hi].mul_(self.scale)

    def _divide(self, lo, grad_hi):
        """flat[lo:grad_hi] /= the summed image count (device scalar in the count slot), on the current stream."""
        from . import _native as nat
        nat.check(nat.lib().sqd_grad_scale(nat.c_p(self._flat.data_ptr() + 4 * int(lo)), int(grad_hi - lo), 1.0,
                                           nat.c_p(self._flat.data_ptr() + 4 * self._total), None, 0.0,
                                           nat.stream_handle(self._flat.device)), 'sqd_grad_scale')

    def ready(self, lo, hi):
        """``flat[lo:hi]`` is final on the current stream AND already weighted by ``scale``: all-reduce it (SUM) -- on the side stream when
        overlapping.  ``finish`` divides by the summed image count (the count slot travels with the first bucket handed over, the tail
        of the buffer)."""
        if not self._active or hi <= lo:
            return
        d = _dist()
        if hi == self._total:
            hi = self._flat.numel()             # the count slot travels with the tail bucket
        elif not self.buckets_last_step and self._native:
            raise RuntimeError('GradientExchange.ready: the first bucket must be the tail of the buffer (it carries the count slot)')
        self.buckets_last_step.append((int(lo), int(hi)))
        grad_hi = min(hi, self._total)
        if self._side is not None:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                work = d.all_reduce(self._flat[lo:hi], op=d.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            work = d.all_reduce(self._flat[lo:hi], op=d.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, lo, grad_hi))
        if self._flat.is_cuda and torch.cuda.is_current_stream_capturing():
            # a collective issued while a hipGraph is being captured: its Work object (and the events inside, recorded in the capturing
            # stream) is kept alive for the life of the exchange.  Released, those events go back to the process group's event cache and
            # the next EAGER collective (a barrier right behind the capture) reuses one; the group's watchdog thread has been seen to
            # query such an event while it still counted as "recorded in a capturing stream" and to abort the process (about one run
            # in ten of `bench.py --force-dist`).  A handful of objects per captured step.
            self._captured_works.append(work)

    def finish(self):
        if not self._active:
            return
        for w, _lo, _hi in self._pending:
            w.wait()                            # CUDA: the current stream waits for the collective; CPU: blocks
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        if self._native:
            # ONE launch over the whole buffer, behind the last collective (the summed count is in the slot by then).  (Dividing every
            # bucket on the side stream right behind its own collective -- side waits on the collective's stream, the next collective
            # waits on side again -- captured into a graph that crashed hipStreamEndCapture on ROCm 7.2; 8 MB take ~5 us here.)
            self._divide(0, self._total)
        else:
            flat, total = self._flat, self._total
            flat[:total].div_(flat[total])      # global image count, summed by the same all-reduce
        self._pending = []
        self._flat = None


def broadcast_parameters(module, src=0, group=None):
    """Identical weights on every rank, ONE broadcast of a flat copy (not one per tensor); written back with
    ``copy_`` so version counters move and the packed-weight caches of the HIP plan refresh."""
    d = _dist()
    if d is None or d.get_world_size(group) == 1:
        return
    params = [p for p in module.parameters()]
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        d.broadcast(flat, src=src, group=group)
        off = 0
        for p in params:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))
            off += n


def find_base(model):
    """The ``SqueezeDetBase`` inside ``model`` (``SqueezeDetWithLoss`` / ``SqueezeDet`` / the base itself / a ``.module``
    wrapper)."""
    from .model import SqueezeDetBase
    for m in model.modules():
        if isinstance(m, SqueezeDetBase):
            return m
    raise TypeError('no SqueezeDetBase inside the model')


# this process's random streams already differ from the other ranks' (seeded here or restored), and the seed they had then: a later
# torch.manual_seed() by the user makes all ranks equal again, which the next attach notices by the changed initial seed
_RANK_STREAMS = {'set': False, 'seed': None}


def mark_rank_streams_set(flag=True):
    """Tell ``attach_data_parallel`` that this process's random streams are already rank-specific (``checkpoint.load_checkpoint``
    calls this after restoring the rank's own streams, so a later ``Trainer(...)`` / ``attach_data_parallel`` leaves them alone);
    ``False``: they are NOT (a checkpoint with fewer streams than ranks was loaded) and the next attach offsets them again."""
    _RANK_STREAMS['set'] = bool(flag)
    _RANK_STREAMS['seed'] = torch.initial_seed() if flag else None


def attach_data_parallel(model, optimizer=None, group=None, overlap=True, broadcast=True, seed_offset=True):
    """Process-per-GPU replacement of the reference's ``DataParallel`` wrapper (src/engine/trainer.py:83-85): after this
    call ``loss.mean().backward()`` on the local shard leaves the gradient of the GLOBAL batch mean in ``p.grad`` on every
    rank.  Also replicates rank 0's weights (and optimizer state tensors) once and de-correlates the dropout streams of
    the ranks -- ONCE per process: a second call (the INTEGRATION.md recipe followed by ``Trainer``), or a call after
    ``load_checkpoint`` has restored this rank's own streams (the reference's resume order: load, then ``Trainer(...)``), does
    not touch the generators again.  No-op without an initialised process group.  Returns the ``GradientExchange`` (or None)."""
    d = _dist()
    base = find_base(model)
    if d is None:
        base.grad_sync = None
        return None
    ex = GradientExchange(group=group, overlap=overlap)
    base.grad_sync = ex
    if broadcast:
        broadcast_parameters(model, 0, group)
        if optimizer is not None:
            tensors = [v for st in optimizer.state.values() for v in st.values() if isinstance(v, torch.Tensor) and v.is_floating_point()]
            if tensors:
                with torch.no_grad():
                    flat = torch.cat([t.reshape(-1) for t in tensors])
                    d.broadcast(flat, src=0, group=group)
                    off = 0
                    for t in tensors:
                        t.copy_(flat[off:off + t.numel()].view_as(t)); off += t.numel()
    if seed_offset and (not _RANK_STREAMS['set'] or _RANK_STREAMS['seed'] != torch.initial_seed()):
        rank = d.get_rank(group)
        if rank:
            torch.manual_seed(torch.initial_seed() + rank)     # seeds the CPU and every GPU generator of this process
        _RANK_STREAMS['set'] = True
        _RANK_STREAMS['seed'] = torch.initial_seed()           # (the dropout in front of ConvDet derives its stream from this seed)
    return ex


def detach_data_parallel(model):
    find_base(model).grad_sync = None


def allreduce_gradients(params, world=None, group=None, local_batch=None):
    """Stand-alone form for gradients that did NOT come out of the HIP backward (tests with oracle gradients, foreign
    modules): averages ``p.grad`` over the ranks as one flat bucket -- weighted by ``local_batch`` when given (exact
    global mean for unequal shards), else a plain mean over ranks.  Consecutive views of one buffer (what the HIP
    backward emits) are reduced in place.  Returns the flat bucket."""
    d = _dist()
    params = [p for p in params if p.grad is not None]
    if not params:
        return None
    if world is None:
        world = d.get_world_size(group) if d is not None else 1
    g0 = params[0].grad
    off, in_place = 0, g0.is_contiguous()
    for p in params:
        g = p.grad
        if not (in_place and g.is_contiguous() and g.dtype == g0.dtype and g.device == g0.device
                and g.untyped_storage().data_ptr() == g0.untyped_storage().data_ptr()
                and g.storage_offset() == g0.storage_offset() + off):
            in_place = False
            break
        off += g.numel()
    if in_place:
        flat = g0.as_strided((off,), (1,), g0.storage_offset())
    else:
        flat = torch.cat([p.grad.reshape(-1) for p in params])
    if d is not None and world > 1:
        if local_batch is None:
            d.all_reduce(flat, op=d.ReduceOp.SUM, group=group)
            flat.mul_(1.0 / world)
        else:
            cnt = torch.full((1,), float(local_batch), dtype=flat.dtype, device=flat.device)
            flat.mul_(float(local_batch))
            d.all_reduce(flat, op=d.ReduceOp.SUM, group=group)
            d.all_reduce(cnt, op=d.ReduceOp.SUM, group=group)
            flat.div_(cnt)
    if not in_place:
        off = 0
        for p in params:
            n = p.numel()
            p.grad.copy_(flat[off:off + n].view_as(p))
            off += n
    return flat


def encode_sparse_batch(batch, cfg):
    """Collate helper: a batch carrying sparse annotations (``gt_boxes`` / ``gt_class_ids``: per-image lists, network-input
    coordinates) instead of the dense ``gt`` gets the dense tensor built ON THE GPU (``annotations.encode_annotations``;
    the reference does it per image in DataLoader workers, src/datasets/base.py:61-76).  Dense batches pass through."""
    if 'gt' in batch or 'gt_boxes' not in batch:
        return batch
    from .annotations import encode_annotations
    out = {k: v for k, v in batch.items() if k not in ('gt_boxes', 'gt_class_ids')}
    out['gt'] = encode_annotations(batch['gt_class_ids'], batch['gt_boxes'], cfg.anchors, cfg.num_classes, device=cfg.device)
    return out


class _Mean:
    """Weighted running mean of a logged quantity."""
    __slots__ = ('last', 'total', 'weight')

    def __init__(self):
        self.last, self.total, self.weight = 0.0, 0.0, 0.0

    def add(self, value, weight=1.0):
        self.last = value
        self.total += value * weight
        self.weight += weight

    @property
    def mean(self):
        return self.total / self.weight if self.weight else 0.0


LOSS_KEYS = ('loss', 'class_loss', 'score_loss', 'bbox_loss')


class Trainer(object):
    """Same constructor, ``train_epoch`` / ``val_epoch`` / ``run_epoch`` / ``set_device`` surface and return values as the
    reference's ``Trainer`` (src/engine/trainer.py:9-92).  Differences underneath: process-per-GPU gradient exchange
    (``attach_data_parallel``) instead of the DataParallel wrapper, sparse annotations are encoded on the GPU, and the
    four logged losses cost ONE host sync per iteration (one stacked copy) instead of four ``.item()`` calls."""

    def __init__(self, model, optimizer, lr_scheduler, cfg):
        self.model, self.optimizer, self.lr_scheduler, self.cfg = model, optimizer, lr_scheduler, cfg
        self.metrics = list(LOSS_KEYS)
        self.set_device(cfg.gpus, cfg.chunk_sizes, cfg.device)

    def set_device(self, gpus, chunk_sizes, device):
        """``gpus`` / ``chunk_sizes`` describe the global job as in the reference; this process only owns ``device``.
        Multi-GPU = a process group initialised by the launcher (torchrun)."""
        d = _dist()
        self.world = d.get_world_size() if d is not None else 1
        self.rank = d.get_rank() if d is not None else 0
        self.model = self.model.to(device)
        for st in self.optimizer.state.values():
            for name, v in list(st.items()):
                if isinstance(v, torch.Tensor):
                    st[name] = v.to(device=device, non_blocking=True)
        self.exchange = attach_data_parallel(self.model, self.optimizer)

    def _to_device(self, batch):
        batch = encode_sparse_batch(batch, self.cfg)
        return {k: (v.to(device=self.cfg.device, non_blocking=True) if ('image_meta' not in k and isinstance(v, torch.Tensor)) else v)
                for k, v in batch.items()}

    def _iteration(self, batch, train):
        fused = train and hasattr(self.model, 'forward_mean')
        if fused:
            # loss.mean() and its backward inside the loss kernels (no torch reduction / elementwise launches in the step)
            mean_loss, parts = self.model.forward_mean(batch)
            per_image_loss = parts['loss']
        else:
            per_image_loss, parts = self.model(batch)
        if train:
            self.optimizer.zero_grad()
            if fused:
                if getattr(self, '_one', None) is None or self._one.device != mean_loss.device:
                    self._one = torch.ones((), device=mean_loss.device)
                mean_loss.backward(self._one)      # local shard mean; the gradient exchange turns it into the global mean
            else:
                per_image_loss.mean().backward()
            if not (isinstance(self.optimizer, FusedClipSGD) and self.optimizer.max_norm > 0):      # (it clips inside its one launch)
                torch.nn.utils.clip_grad_norm_([p for p in self.model.parameters() if p.requires_grad], self.cfg.grad_norm)
            self.optimizer.step()
        n = per_image_loss.shape[0]
        logged = torch.stack([parts[k].detach().sum() for k in self.metrics] + [per_image_loss.new_tensor(float(n))])
        if self.world > 1:
            _dist().all_reduce(logged)             # 5 floats: global sums and the global image count
        logged = logged.tolist()                   # the iteration's single host sync
        return [v / logged[-1] for v in logged[:-1]], n

    def run_epoch(self, phase, epoch, data_loader):
        t_epoch = time.time()
        train = phase == 'train'
        self.model.train(train)
        means = {k: _Mean() for k in self.metrics}
        limit = len(data_loader) if self.cfg.num_iters < 0 else self.cfg.num_iters
        t_mark = time.time()
        for it, batch in enumerate(data_loader):
            if it >= limit:
                break
            batch = self._to_device(batch)
            t_data = time.time() - t_mark
            values, n = self._iteration(batch, train)
            for k, v in zip(self.metrics, values):
                means[k].add(v, n)
            t_net = time.time() - t_mark - t_data
            if self.rank == 0 and it % self.cfg.print_interval == 0:
                losses = ' '.join(f'| {k} {v:.3f}' for k, v in zip(self.metrics, values))
                print(f'epoch {epoch}: {phase:<5s} [{it}/{limit}] {losses} | data {1e3 * t_data:.1f}ms | net {1e3 * t_net:.1f}ms')
            t_mark = time.time()
        if train:
            self.lr_scheduler.step()
        out = {k: m.mean for k, m in means.items()}
        out['epoch_time'] = (time.time() - t_epoch) / 60.0
        return out

    def train_epoch(self, epoch, data_loader):
        return self.run_epoch('train', epoch, data_loader)

    @torch.no_grad()
    def val_epoch(self, epoch, data_loader):
        return self.run_epoch('val', epoch, data_loader)


class FusedClipSGD(torch.optim.Optimizer):
    """``torch.nn.utils.clip_grad_norm_(params, max_norm)`` + ``torch.optim.SGD(params, lr, momentum, weight_decay).step()``
    (src/engine/trainer.py:47-50) as ONE launch over all parameter tensors (csrc/optim.hip, sqd_sgd_clip_step): torch runs the pair
    as ~10 foreach / elementwise launches over the 64 tensors (about 0.15 ms of a 6.3 ms training step).  Same arithmetic in the
    same order per element; the gradient norm is one reduction over the backward's flat gradient buffer when the gradients are
    consecutive views of it (the HIP backward's layout), else torch's foreach norm; it is read on the device (no host sync, hipGraph
    capturable).

    A ``torch.optim.Optimizer`` with ONE parameter group: ``param_groups[0]`` carries ``lr`` / ``momentum`` / ``weight_decay`` /
    ``max_norm`` and is read at every ``step()``, so ``torch.optim.lr_scheduler.StepLR(opt, 60, 0.5)`` (the reference's schedule,
    src/train.py:36, stepped at src/engine/trainer.py:67) drives it like ``torch.optim.SGD``.  ``max_norm > 0`` clips INSIDE
    ``step()``: a caller that keeps the reference's explicit ``clip_grad_norm_`` line must construct it with ``max_norm=0`` (or
    drop the line, as ``Trainer._iteration`` does) -- otherwise the gradient is clipped twice.  The hyper-parameters are launch
    arguments: a hipGraph replay of a captured step keeps the values of the capture (re-capture after an LR change)."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, max_norm=0.0, flat_grad=None):
        params = [p for p in params if p.requires_grad]
        if not params or any(p.dtype != torch.float32 or not p.is_cuda for p in params):
            raise ValueError('FusedClipSGD: fp32 parameters on the GPU only (the product path has no CPU fallback)')
        super().__init__(params, dict(lr=float(lr), momentum=float(momentum), weight_decay=float(weight_decay), max_norm=float(max_norm)))
        if len(self.param_groups) != 1:
            raise ValueError('FusedClipSGD: one parameter group (the kernel applies one set of hyper-parameters to all tensors)')
        self.flat_grad = flat_grad                     # callable -> the flat gradient tensor the .grad views live in (or None)
        dev = self.params[0].device
        self.total = sum(p.numel() for p in self.params)
        self.momentum_flat = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self._bufs, off = [], 0
        for p in self.params:
            self._bufs.append(self.momentum_flat[off:off + p.numel()]); off += p.numel()
            self.state[p]['momentum_buffer'] = self._bufs[-1]       # views: attach_data_parallel broadcasts optimizer state tensors in place
        self._table = self._table_key = self._chunks = None
        self._norm_ws = None                           # partial sums of squares + the norm (device)
        self.last_norm = None

    @property
    def params(self):
        return self.param_groups[0]['params']

    def _hyper(self, name):
        return float(self.param_groups[0][name])

    lr = property(lambda self: self._hyper('lr'))
    momentum = property(lambda self: self._hyper('momentum'))
    weight_decay = property(lambda self: self._hyper('weight_decay'))
    max_norm = property(lambda self: self._hyper('max_norm'))

    def add_param_group(self, param_group):
        if getattr(self, 'param_groups', None):
            raise ValueError('FusedClipSGD: one parameter group only')
        super().add_param_group(param_group)

    def _flat_base(self, grads):
        """The flat gradient tensor if the gradients are consecutive views of it (the HIP backward's layout), else None."""
        flat = self.flat_grad() if self.flat_grad is not None else None
        if flat is None or not flat.is_contiguous() or flat.numel() < self.total:
            return None
        ptr = flat.data_ptr()
        for g in grads:
            if not g.is_contiguous() or g.data_ptr() != ptr:
                return None
            ptr += g.numel() * 4
        return flat

    @torch.no_grad()
    def step(self, closure=None):
        """One clip + SGD step.  Returns the total gradient norm (``clip_grad_norm_``'s return value) as a 0-dim DEVICE tensor that
        ALIASES a persistent workspace (also kept in ``last_norm``): the next ``step`` overwrites it in place -- a copy per step would
        be the one torch kernel of an otherwise torch-free captured training step.  Callers that keep norms across iterations take
        the value at once (``float(norm)`` / ``norm.item()``) or ``norm.clone()`` (``norm_value()`` = the float of the latest step)."""
        from . import _native as nat
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        params = self.params
        grads = [p.grad for p in params]
        if any(g is None for g in grads):
            raise RuntimeError('FusedClipSGD.step: a parameter has no gradient')
        flat = self._flat_base(grads)
        if flat is None:
            grads = [g if g.is_contiguous() else g.contiguous() for g in grads]
        # the descriptor table holds the parameter and momentum addresses and, per tensor, its offset into the flat gradient
        # buffer (whose own address is a launch argument) or the gradient's address.  Everything baked into it is part of the key:
        # a parameter whose storage was replaced after the first step (model.to() / p.data = ...) rebuilds the table instead of
        # letting the kernel write through a stale pointer
        key = (('flat',) if flat is not None else tuple(g.data_ptr() for g in grads),
               tuple(p.data_ptr() for p in params), self.momentum_flat.data_ptr())
        if key != self._table_key:
            off, rows = 0, []
            for p, g, b in zip(params, grads, self._bufs):
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise RuntimeError('FusedClipSGD.step: parameters must stay contiguous fp32 GPU tensors')
                rows.append([p.data_ptr(), off if flat is not None else g.data_ptr(), b.data_ptr(), p.numel()]); off += p.numel()
            self._table = torch.tensor(rows, dtype=torch.int64).to(params[0].device)
            # equal chunks of the 64 tensors (16 .. 500 k elements): one workgroup per chunk
            ce = int(nat.lib().sqd_sgd_chunk_elems())
            chunks = [[i, e] for i, p in enumerate(params) for e in range(0, p.numel(), ce)]
            self._chunks = torch.tensor(chunks, dtype=torch.int64).to(params[0].device)
            self._table_key = key
        norm = None
        max_norm = self.max_norm
        if max_norm > 0 and flat is not None:
            # the norm of the flat gradient buffer: partial sums of squares by one small launch, finished (fixed order) inside the step
            if self._norm_ws is None:
                self._norm_ws = torch.empty(int(nat.lib().sqd_grad_sumsq_parts()) + 1, device=flat.device, dtype=torch.float32)
            parts, norm = self._norm_ws[:-1], self._norm_ws[-1]
            nat.check(nat.lib().sqd_grad_sumsq(nat.ptr(flat), self.total, nat.ptr(parts), nat.stream_handle(flat.device)), 'sqd_grad_sumsq')
            nat.check(nat.lib().sqd_sgd_clip_step_chunked(nat.ptr(self._table), nat.ptr(self._chunks), self._chunks.shape[0], nat.ptr(flat),
                                                          nat.ptr(parts), nat.ptr(norm), max_norm, self.lr, self.momentum, self.weight_decay,
                                                          nat.stream_handle(params[0].device)), 'sqd_sgd_clip_step_chunked')
        else:
            if max_norm > 0:
                norm = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads)))
            nat.check(nat.lib().sqd_sgd_clip_step(nat.ptr(self._table), len(params), nat.ptr(flat), nat.ptr(norm), max_norm, self.lr,
                                                  self.momentum, self.weight_decay, 64, nat.stream_handle(params[0].device)),
                      'sqd_sgd_clip_step')
        torch.autograd.graph.increment_version(params)        # (the packed-weight caches key on the version counters)
        self.last_norm = norm
        return norm if closure is None else loss

    def norm_value(self):
        """The latest step's total gradient norm as a Python float (one host sync), or None before the first clipped step."""
        return None if self.last_norm is None else float(self.last_norm)

    def state_dict(self):
        g = self.param_groups[0]
        return {'momentum_flat': self.momentum_flat.clone(), 'lr': g['lr'], 'momentum': g['momentum'], 'weight_decay': g['weight_decay'],
                'max_norm': g['max_norm'], 'initial_lr': g.get('initial_lr')}

    def load_state_dict(self, sd):
        self.momentum_flat.copy_(sd['momentum_flat'].to(self.momentum_flat.device))
        g = self.param_groups[0]
        for k in ('lr', 'momentum', 'weight_decay', 'max_norm'):
            if sd.get(k) is not None:
                g[k] = float(sd[k])
        if sd.get('initial_lr') is not None:
            g['initial_lr'] = float(sd['initial_lr'])


def make_train_step(cfg, state_dict, image, rank, world, dist, gt_seed=1, force_exchange=False, fused_optimizer=True, parts=None):
    """Benchmark helper: returns (step_fn, description, probe_fn).  A step = fwd + loss + bwd (with the bucketed RCCL
    gradient exchange when a process group exists) + clip_grad_norm_(5.0) + SGD(lr .01, momentum .9, wd 1e-4) on a
    device-resident batch.  ``probe_fn()`` -> (gt on the CPU, eval-mode per-image loss of the current weights on the CPU):
    what bench.py's CPU leg checks against the oracle before the first optimizer step."""
    from . import synthetic
    from .model import SqueezeDetWithLoss
    model = SqueezeDetWithLoss(cfg)
    model.load_state_dict(state_dict)
    model = model.to(image.device).train()
    params = [p for p in model.parameters() if p.requires_grad]
    if fused_optimizer:
        base = find_base(model)
        opt = FusedClipSGD(params, lr=cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay, max_norm=cfg.grad_norm,
                           flat_grad=lambda: base.last_grad_flat)
    else:
        opt = torch.optim.SGD(params, lr=cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
    gt_cpu = synthetic.make_gt(image.shape[0], cfg.anchors, cfg.input_size, cfg.num_classes, seed=gt_seed + rank)
    gt = gt_cpu.to(image.device)
    batch = {'image': image, 'gt': gt}
    ex = attach_data_parallel(model, opt) if dist is not None else None
    if ex is not None:
        ex.force = bool(force_exchange)
    if parts is not None:                            # (bench.py diagnostics: the objects behind the closure)
        parts.update(model=model, optimizer=opt, base=find_base(model), exchange=ex)

    one = torch.ones((), device=image.device)       # the root gradient, allocated once (autograd would fill a new tensor every step)

    def step():
        loss, stats = model.forward_mean(batch)     # = model(batch)[0].mean(), the mean and its backward inside the loss kernels
        opt.zero_grad()
        loss.backward(one)
        if not fused_optimizer:
            torch.nn.utils.clip_grad_norm_(params, cfg.grad_norm)
        opt.step()                                   # (fused: clip + SGD in one launch)
        return loss

    def probe():
        model.eval()
        with torch.no_grad():
            lv, _ = model(batch)
        model.train()
        return gt_cpu, lv.detach().cpu()

    net = 'SqueezeDet' if cfg.arch == 'squeezedet' else 'SqueezeDet+'
    desc = f'{net} KITTI 1248x384 bs={image.shape[0]}/GPU training: fwd + multi-task loss + bwd + clip(5.0) + SGD' + (
        ' (clip + SGD: one fused launch)' if fused_optimizer else '')
    if ex is not None:
        desc += f' + RCCL gradient all-reduce over {world} GPU(s) in 3 buckets overlapped with backward'
    return step, desc, probe
