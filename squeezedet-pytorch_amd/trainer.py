"""``Trainer`` with the reference's interface (src/engine/trainer.py:9-92) and process-per-GPU data
parallelism over RCCL/xGMI in place of the reference's single-process ``DataParallel``
(src/utils/data_parallel.py:46-156).

Reference semantics kept: ``loss = loss.mean()`` over the GLOBAL per-sample loss vector
(trainer.py:43; DataParallel gathers the per-sample vectors to GPU 0 first), ``zero_grad`` ->
``backward`` -> ``clip_grad_norm_(params, cfg.grad_norm)`` -> ``optimizer.step()`` (:46-50), StepLR per
epoch (:67-68).  With W ranks each holding b = B/W images the global mean's gradient is
(1/W) * sum_r grad(mean_local_r): one all-reduce(SUM) of the flattened gradient (2,082,120 floats =
8.33 MB for SqueezeDet) per step, then a 1/W scale; the clip then sees the same global norm on every
rank and the SGD step is local and deterministic, so replicas stay bit-identical.  There is no
per-step weight broadcast (the reference's ``replicate``) and no data-path collective in inference.
"""
from __future__ import annotations

import time

import torch
import torch.nn as nn

EPSILON = 1e-10


class MetricLogger(object):
    """Running average (src/utils/misc.py:29-40)."""

    def __init__(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / (self.count + EPSILON)


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def shard_sizes(batch_size, world):
    """Per-rank chunk sizes for a global batch, first ranks take the remainder -- the equal-split case
    of the reference's ``chunk_sizes`` (src/utils/config.py:102-110)."""
    base, rem = divmod(batch_size, world)
    return [base + (1 if r < rem else 0) for r in range(world)]


def allreduce_gradients(params, world=None, group=None):
    """Sum-all-reduce the gradients of ``params`` as ONE flat bucket and divide by the world size.
    Works for any backend (RCCL on GPUs, gloo in the CPU tests).  Returns the flat bucket."""
    dist = _dist()
    params = [p for p in params if p.grad is not None]
    if not params:
        return None
    if world is None:
        world = dist.get_world_size(group) if dist is not None else 1
    # The HIP backward already emits every gradient as a view of one flat buffer (named_parameters order): all-reduce
    # that buffer in place -- no gather copy, no scatter back.
    g0 = params[0].grad
    off, contiguous_views = 0, g0.is_contiguous()
    for p in params:
        g = p.grad
        if not (contiguous_views and g.is_contiguous() and g.dtype == g0.dtype and g.device == g0.device
                and g.untyped_storage().data_ptr() == g0.untyped_storage().data_ptr()
                and g.storage_offset() == g0.storage_offset() + off):
            contiguous_views = False
            break
        off += g.numel()
    if contiguous_views:
        flat = g0.as_strided((off,), (1,), g0.storage_offset()) if off != g0.numel() else g0.reshape(-1)
        if dist is not None and world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            flat.mul_(1.0 / world)
        return flat
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    if dist is not None and world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.mul_(1.0 / world)
    off = 0
    views = []
    for p in params:
        n = p.numel()
        views.append(flat[off:off + n].view_as(p))
        off += n
    torch._foreach_copy_([p.grad for p in params], views)
    return flat


class Trainer(object):
    def __init__(self, model, optimizer, lr_scheduler, cfg):
        self.model = model
        self.optimizer = optimizer
        self.lr_scheduler = lr_scheduler
        self.cfg = cfg
        self.set_device(cfg.gpus, cfg.chunk_sizes, cfg.device)
        self.metrics = ['loss', 'class_loss', 'score_loss', 'bbox_loss']

    def run_epoch(self, phase, epoch, data_loader):
        start_time = time.time()
        if phase == 'train':
            self.model.train()
        else:
            self.model.eval()
            torch.cuda.empty_cache()

        metric_loggers = {m: MetricLogger() for m in self.metrics}
        data_timer, net_timer = MetricLogger(), MetricLogger()
        num_iters = len(data_loader) if self.cfg.num_iters < 0 else self.cfg.num_iters
        end = time.time()
        params = [p for p in self.model.parameters() if p.requires_grad]

        for iter_id, batch in enumerate(data_loader):
            if iter_id >= num_iters:
                break
            if 'gt' not in batch and 'gt_boxes' in batch:
                # sparse annotations (per-image lists of xyxy boxes / class ids): encode the dense gt on the GPU
                # instead of in DataLoader workers (prepare_annotations, src/datasets/base.py:61-76)
                from .annotations import encode_annotations
                batch = dict(batch)
                batch['gt'] = encode_annotations(batch.pop('gt_class_ids'), batch.pop('gt_boxes'), self.cfg.anchors,
                                                 self.cfg.num_classes, device=self.cfg.device)
            for k in batch:
                if 'image_meta' not in k and isinstance(batch[k], torch.Tensor):
                    batch[k] = batch[k].to(device=self.cfg.device, non_blocking=True)
            data_timer.update(time.time() - end)
            end = time.time()

            loss, loss_stats = self.model(batch)
            loss = loss.mean()                       # local shard mean; the all-reduce below makes it the global mean

            if phase == 'train':
                self.optimizer.zero_grad()
                loss.backward()
                if self.world > 1:
                    allreduce_gradients(params, self.world)
                nn.utils.clip_grad_norm_(params, self.cfg.grad_norm)
                self.optimizer.step()

            msg = 'epoch {0:<3s} {1:<5s} [{2}/{3}] '.format(str(epoch) + ':', phase, iter_id, num_iters)
            stats = torch.stack([loss_stats[m].mean() for m in metric_loggers]).detach()
            if self.world > 1:                       # 4 floats: log the global means like the gathered vector would
                _dist().all_reduce(stats)
                stats /= self.world
            stats = stats.tolist()                   # ONE host sync per iteration (the reference does four .item())
            for m, value in zip(metric_loggers, stats):
                metric_loggers[m].update(value, batch['image'].shape[0])
                msg += '| {} {:.3f} '.format(m, value)

            net_timer.update(time.time() - end)
            end = time.time()
            msg += '| data {:.1f}ms | net {:.1f}ms'.format(1000. * data_timer.val, 1000. * net_timer.val)
            if iter_id % self.cfg.print_interval == 0 and self.rank == 0:
                print(msg)
            del loss, loss_stats

        if phase == 'train':
            self.lr_scheduler.step()
        stats = {k: v.avg for k, v in metric_loggers.items()}
        stats.update({'epoch_time': (time.time() - start_time) / 60.})
        return stats

    def train_epoch(self, epoch, data_loader):
        return self.run_epoch('train', epoch, data_loader)

    @torch.no_grad()
    def val_epoch(self, epoch, data_loader):
        return self.run_epoch('val', epoch, data_loader)

    def set_device(self, gpus, chunk_sizes, device):
        """One process per GPU: ``gpus``/``chunk_sizes`` describe the global job like in the reference, but this
        process only ever owns ``device``.  Multi-GPU = torch.distributed already initialised by the launcher."""
        dist = _dist()
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.model = self.model.to(device)
        if self.world > 1:                           # identical initial weights on every rank (once, not per step)
            for p in self.model.parameters():
                dist.broadcast(p.data, src=0)
        for state in self.optimizer.state.values():
            for k, v in state.items():
                if isinstance(v, torch.Tensor):
                    state[k] = v.to(device=device, non_blocking=True)


def make_train_step(cfg, state_dict, image, rank, world, dist, gt_seed=1):
    """Benchmark helper: returns (step_fn, description).  A step = fwd + loss + bwd (+ RCCL all-reduce)
    + clip_grad_norm_(5.0) + SGD(lr .01, momentum .9, wd 1e-4) on a device-resident batch."""
    from . import synthetic
    from .model import SqueezeDetWithLoss
    model = SqueezeDetWithLoss(cfg)
    model.load_state_dict(state_dict)
    model = model.to(image.device).train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
    gt = synthetic.make_gt(image.shape[0], cfg.anchors, cfg.input_size, cfg.num_classes, seed=gt_seed + rank).to(image.device)
    batch = {'image': image, 'gt': gt}

    def step():
        loss, stats = model(batch)
        loss = loss.mean()
        opt.zero_grad()
        loss.backward()
        if world > 1:
            allreduce_gradients(params, world)
        nn.utils.clip_grad_norm_(params, cfg.grad_norm)
        opt.step()
        return loss

    desc = (f'{"SqueezeDet" if cfg.arch == "squeezedet" else "SqueezeDet+"} KITTI 1248x384 bs={image.shape[0]}/GPU training: fwd + multi-task loss + bwd + clip(5.0) + SGD'
            + (f' + RCCL grad all-reduce over {world} GPUs' if world > 1 else ''))
    return step, desc
