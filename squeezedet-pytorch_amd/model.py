"""``nn.Module`` surface of the reference's ``src/model/squeezedet.py`` -- same class names,
constructor arguments, ``forward`` signatures, sub-module attribute names and ``state_dict`` keys /
shapes (OIHW fp32) -- with every forward/backward running on the hand-written gfx950 kernels.

Reference map:
  Fire                 src/model/squeezedet.py:9-23
  SqueezeDetBase       :26-97   (features Sequential, dropout, convdet, init_weights)
  PredictionResolver   :100-120
  Loss                 :123-174
  SqueezeDetWithLoss   :177-187
  SqueezeDet           :190-206

Internally activations are NHWC fp32; the two expand convolutions of a Fire write disjoint channel
ranges of one buffer (no ``torch.cat``), ConvDet's NHWC output *is* the permuted tensor of :85.
Canonical parameters stay OIHW ``nn.Parameter``s (optimizer / clip_grad_norm_ / state_dict touch
them directly); the packed copies the kernels read are a private cache keyed on ``param._version``.
There is no CPU path: modules raise if the input is not on a GPU or the HIP library is missing.
"""
from __future__ import annotations


import numpy as np
import torch
import torch.nn as nn

from . import ops
from .synthetic import layer_table, convdet_in_channels


def _nhwc_input(x, channels, who):
    """Stand-alone sub-module calls take the reference's NCHW tensors: checked, detached, viewed NHWC (no copy when the
    tensor came out of another sub-module, whose NCHW result is a permuted view of an NHWC buffer)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == channels):
        raise RuntimeError(f'{who}: expected an fp32 CUDA/HIP tensor [B,{channels},H,W] (the layers run on the MI355X kernels only)')
    return x.detach().permute(0, 2, 3, 1).contiguous()


class _ConvParams(nn.Module):
    """Parameter holder with nn.Conv2d's state_dict layout (``weight`` OIHW, ``bias``).  Inside ``SqueezeDetBase.forward``
    the layer runs as part of the fused HIP plan; called on its own (``model.base.features[0](x)``, ``fire.squeeze(s)``,
    as the reference's ``nn.Conv2d`` allows) it launches its kernel stand-alone: NCHW in, NCHW out, no activation,
    inference only (not differentiable -- training differentiates through the parent module)."""

    def __init__(self, cin, cout, ksize, stride=1, padding=0):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.kernel_size, self.stride, self.padding = (ksize, ksize), (stride, stride), (padding, padding)
        self.weight = nn.Parameter(torch.empty(cout, cin, ksize, ksize))
        self.bias = nn.Parameter(torch.empty(cout))
        self._own_plan = None

    def extra_repr(self):
        return f'{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, padding={self.padding}'

    def _plan(self, npix):
        k = self.kernel_size[0]
        wc = ops.choose_wino_cfg(self.in_channels, self.out_channels, npix) if k == 3 else None
        cfg_id = ('w', wc) if wc is not None else ('d', ops.choose_cfg(k * k, self.in_channels, self.out_channels, npix))
        ver = (cfg_id, self.weight._version, self.weight.data_ptr(), self.bias._version, self.bias.data_ptr())
        if self._own_plan is None or self._own_plan[0] != ver:
            plan = ops.WinoPlan(self.weight, self.bias, wc) if wc is not None else ops.ConvPlan(self.weight, self.bias, cfg_id[1])
            self._own_plan = (ver, plan)
        return self._own_plan[1]

    def run_nhwc(self, x, relu=False, out=None, out_coff=0):
        """NHWC in -> NHWC out (channel window ``out_coff`` of ``out`` when given)."""
        Bq, H, W, _ = x.shape
        y = out if out is not None else torch.empty(Bq, H, W, self.out_channels, device=x.device, dtype=torch.float32)
        plan = self._plan(Bq * H * W)
        if isinstance(plan, ops.WinoPlan):
            return ops.conv_wino(x, 0, plan, y, out_coff, relu=relu)
        return ops.conv(x, 0, plan, y, out_coff, relu=relu)

    def forward(self, x):
        if self.stride[0] == 2:                           # the stem: reads the NCHW image directly
            if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32):
                raise RuntimeError('stem conv: expected an fp32 CUDA/HIP NCHW image')
            return ops.stem_conv_relu(x.detach(), self.weight, self.bias, relu=False).permute(0, 3, 1, 2)
        return self.run_nhwc(_nhwc_input(x, self.in_channels, 'conv')).permute(0, 3, 1, 2)


class _Marker(nn.Module):
    """Keeps the reference's nn.Sequential indices for the ReLU / MaxPool2d slots (fused away inside
    ``SqueezeDetBase.forward``).  Called on its own it applies that layer: the pool through the HIP kernel, the ReLU as a
    plain elementwise op."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what

    def forward(self, x):
        if self.what.startswith('ReLU'):
            return torch.relu(x.detach())
        return ops.maxpool(_nhwc_input(x, x.shape[1], 'max pool')).permute(0, 3, 1, 2)


class Fire(nn.Module):
    def __init__(self, inplanes, squeeze_planes, expand1x1_planes, expand3x3_planes):
        super().__init__()
        self.squeeze = _ConvParams(inplanes, squeeze_planes, 1)
        self.expand1x1 = _ConvParams(squeeze_planes, expand1x1_planes, 1)
        self.expand3x3 = _ConvParams(squeeze_planes, expand3x3_planes, 3, padding=1)

    def forward(self, x):
        """Stand-alone Fire (src/model/squeezedet.py:17-23): NCHW in, NCHW out; three kernel launches, the two expands
        write the two halves of one buffer (no ``torch.cat``).  Inference only."""
        xh = _nhwc_input(x, self.squeeze.in_channels, 'Fire')
        s = self.squeeze.run_nhwc(xh, relu=True)
        e1, e3 = self.expand1x1.out_channels, self.expand3x3.out_channels
        out = torch.empty(*s.shape[:3], e1 + e3, device=s.device, dtype=torch.float32)
        self.expand1x1.run_nhwc(s, relu=True, out=out, out_coff=0)
        self.expand3x3.run_nhwc(s, relu=True, out=out, out_coff=e1)
        return out.permute(0, 3, 1, 2)


class SqueezeDetBase(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.num_classes = cfg.num_classes
        self.num_anchors = cfg.num_anchors
        self.arch = cfg.arch
        mods = []
        for l in layer_table(cfg.arch):                   # raises ValueError('Invalid architecture.')
            if l[0] == 'conv':
                mods.append(_ConvParams(l[1], l[2], l[3], stride=l[4], padding=l[5]))
            elif l[0] == 'relu':
                mods.append(_Marker('ReLU (fused into the stem kernel)'))
            elif l[0] == 'pool':
                mods.append(_Marker('MaxPool2d(kernel_size=3, stride=2, ceil_mode=True)'))
            else:
                mods.append(Fire(*l[1:]))
        self.features = nn.Sequential(*mods)
        self.dropout_prob = float(cfg.dropout_prob)
        self.dropout = nn.Dropout(cfg.dropout_prob, inplace=True) if cfg.dropout_prob > 0 else None
        self.convdet = _ConvParams(convdet_in_channels(cfg.arch), cfg.anchors_per_grid * (cfg.num_classes + 5), 3, padding=1)
        self._plans = {}
        self._fused_plans = {}
        self._wino_plans = {}
        self._wgrad_batches = {}
        self.last_grad_flat = None                # flat gradient buffer of the latest backward (every .grad is a view of it)
        self.grad_sync = None                     # trainer.GradientExchange when data parallel (attach_data_parallel)
        self.use_winograd = True                  # 3x3 forward convs: Winograd F(2x2,3x3) kernel where tuning.json says it is faster
        self.fuse_expand = True                   # inference forward: expand1x1 + expand3x3 in one launch
        self.fuse_expand_wino = True              # ... in Winograd form (ops.fire_wino) where the table has an X: row
        self.fuse_fire_bridge = True              # ... together with the NEXT Fire's squeeze (ops.fire_bridge) where it has a Y: row
        self.fuse_train_forward = True    # training forward: the bridges in their storing forms
        # inference forward: pool 2 / 3 folded into the following squeeze (ops.pool_squeeze).  Off by default: measured equal
        # to the two separate kernels (0.174 vs 0.18 ms) -- both are bound by the 9x L2 read amplification of the window gather
        self.fuse_pool_squeeze = False
        self.fuse_stem_squeeze = True             # inference forward: the first Fire's squeeze inside the stem launch (ops.stem_pool_squeeze)
        self._pack_table_keepalive = None
        self._forced_drop_mask = None       # tests: NCHW mask (already scaled by 1/(1-p)) instead of RNG
        # counter-based dropout (ops.DropState): applied in the last Fire's expand epilogues, its step advanced by ConvDet's launch,
        # no mask tensor and no torch RNG kernel in the step.  fused_dropout = False draws the mask as a tensor (stand-alone kernel)
        self.fused_dropout = True
        self._drop = None                   # (torch.initial_seed() it was derived from, ops.DropState)
        self._drop_restored = False         # the stream came from set_dropout_rng (a checkpoint), not from torch's seed
        self.fuse_squeeze_bwd = True     # backward: squeeze wgrad + dgrad in one launch
        self.group_wgrad = ops.WINO_WGRAD_GROUP   # backward: the expand3x3 weight gradients of a stage share one launch (ops.conv_wgrad_wino_group)
        self.init_weights()

    # ---- reference: init_weights, src/model/squeezedet.py:89-97 ----
    def init_weights(self):
        for m in self.modules():
            if isinstance(m, _ConvParams):
                nn.init.normal_(m.weight, mean=0.0, std=0.002 if m is self.convdet else 0.005)
                nn.init.constant_(m.bias, 0)

    # ---- packed-weight cache ----
    def plan(self, name, mod, cfg_id, direction='fwd'):
        key = (name, cfg_id, direction)
        ver = (mod.weight._version, mod.weight.data_ptr(), mod.bias._version, mod.bias.data_ptr())
        hit = self._plans.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        p = ops.ConvPlan(mod.weight, mod.bias, cfg_id, dgrad=(direction != 'fwd'))
        self._plans[key] = (ver, p)
        return p

    def fused_expand_plan(self, idx, fire, cfg_id):
        """Packed weights of ``fire``'s expand pair for the one-launch fused expand (inference forward).  Rebuilt when
        either module's parameters change; not part of ``refresh_plans`` (training keeps the two separate kernels, whose
        per-layer activations and packed weights the backward needs anyway)."""
        key = ('fused', idx, cfg_id)
        mods = (fire.expand1x1, fire.expand3x3)
        ver = tuple(v for m in mods for v in (m.weight._version, m.weight.data_ptr(), m.bias._version, m.bias.data_ptr()))
        hit = self._fused_plans.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        p = ops.FusedExpandPlan(fire.expand1x1.weight, fire.expand1x1.bias, fire.expand3x3.weight, fire.expand3x3.bias, cfg_id)
        self._fused_plans[key] = (ver, p)
        return p

    def fire_wino_plan(self, idx, fire, cfg_id):
        """Packed weights of ``fire``'s expand pair for the one-launch Winograd form (inference forward); rebuilt when either
        module's parameters change."""
        key = ('firewino', idx, cfg_id)
        mods = (fire.expand1x1, fire.expand3x3)
        ver = tuple(v for m in mods for v in (m.weight._version, m.weight.data_ptr(), m.bias._version, m.bias.data_ptr()))
        hit = self._fused_plans.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        p = ops.FireWinoPlan(fire.expand1x1.weight, fire.expand1x1.bias, fire.expand3x3.weight, fire.expand3x3.bias, cfg_id)
        self._fused_plans[key] = (ver, p)
        return p

    def fire_bridge_plan(self, idx, fire, nxt, cfg_id, pooled=False):
        """Operands of the one-launch form of ``fire``'s expand pair + ``nxt``'s squeeze (ops.fire_bridge / ops.fire_pool_bridge);
        refreshed in place when any of the three modules' parameters change (``refresh_plans`` does all bridges with two launches)."""
        key = ('firebridge', idx, cfg_id, pooled)
        mods = (fire.expand1x1, fire.expand3x3, nxt.squeeze)
        hit = self._fused_plans.get(key)
        if hit is not None:
            if hit[0] != self._bridge_version(mods):
                self._refresh_bridge_plans([key])
            return self._fused_plans[key][1]
        p = ops.FireBridgePlan(fire.expand1x1.weight, fire.expand1x1.bias, fire.expand3x3.weight, fire.expand3x3.bias,
                               nxt.squeeze.weight, nxt.squeeze.bias, cfg_id, pooled=pooled)
        self._fused_plans[key] = (self._bridge_version(mods), p, mods)
        return p

    @staticmethod
    def _bridge_version(mods):
        return tuple(v for m in mods for v in (m.weight._version, m.weight.data_ptr(), m.bias._version, m.bias.data_ptr()))

    def _refresh_bridge_plans(self, keys=None):
        """Every cached Fire-bridge plan (or just ``keys``) whose parameters changed: operands rewritten in place, two launches."""
        from . import plans as _plans
        stale = []
        for key, val in self._fused_plans.items():
            if key[0] != 'firebridge' or (keys is not None and key not in keys):
                continue
            ver, plan, mods = val
            now = self._bridge_version(mods)
            if now != ver:
                stale.append((key, now, plan, mods))
        if not stale:
            return
        tables = _plans.refresh_bridge_plans([(plan, m[0].weight.detach(), m[0].bias.detach(), m[1].weight.detach(), m[1].bias.detach(),
                                              m[2].weight.detach(), m[2].bias.detach()) for _k, _n, plan, m in stale])
        self._pack_table_keepalive = (self._pack_table_keepalive or [])[-6:] + [tables]
        for key, now, plan, mods in stale:
            self._fused_plans[key] = (now, plan, mods)

    def wino_plan(self, name, mod, cfg_id, direction='fwd'):
        """Transformed-weight cache of the Winograd 3x3 kernel (ops.WinoPlan), re-transformed in place when the parameter
        changed."""
        key = (name, cfg_id, direction)
        ver = (mod.weight._version, mod.weight.data_ptr(), mod.bias._version, mod.bias.data_ptr())
        hit = self._wino_plans.get(key)
        if hit is not None:
            if hit[0] != ver:
                hit[1].repack(mod.weight, mod.bias, dgrad=(direction != 'fwd'))
                self._wino_plans[key] = (ver, hit[1])
            return hit[1]
        p = ops.WinoPlan(mod.weight, mod.bias, cfg_id, dgrad=(direction != 'fwd'))
        self._wino_plans[key] = (ver, p)
        return p

    def conv3x3(self, name, mod, x, x_coff, y, y_coff, relu, ymul=None):
        """Forward 3x3 convolution of ``mod``: the Winograd kernel where the measured table prefers it, else the direct
        implicit-GEMM kernel."""
        Bq, H, W, _ = x.shape
        C, N = mod.in_channels, mod.out_channels
        wc = ops.choose_wino_cfg(C, N, Bq * H * W) if self.use_winograd else None
        if wc is not None and (ymul is None or tuple(ymul.shape) == tuple(y.shape)):
            return ops.conv_wino(x, x_coff, self.wino_plan(name, mod, wc), y, y_coff, relu=relu, ymul=ymul)
        return ops.conv(x, x_coff, self.plan(name, mod, ops.choose_cfg(9, C, N, Bq * H * W)), y, y_coff, relu=relu,
                        ymul=ymul, ymul_coff=y_coff)

    def dgrad3x3(self, name, mod, dy, dy_coff, dx, accumulate=False, ymask=None, ymul=None):
        """dx (=|+=) data gradient of ``mod``'s 3x3 convolution (transposed, tap-flipped weights), ymul / ymask fused into the
        epilogue; Winograd kernel where the measured table prefers it."""
        Bq, H, W, _ = dy.shape
        C, N = mod.out_channels, mod.in_channels                  # the dgrad convolves dY (out channels) into dX (in channels)
        wc = ops.choose_wino_cfg(C, N, Bq * H * W) if self.use_winograd else None
        same_geom = all(t is None or tuple(t.shape) == tuple(dx.shape) for t in (ymask, ymul))
        if wc is not None and same_geom:
            return ops.conv_wino(dy, dy_coff, self.wino_plan(name, mod, wc, 'dgrad'), dx, 0, accumulate=accumulate, ymask=ymask, ymul=ymul)
        return ops.conv(dy, dy_coff, self.plan(name, mod, ops.choose_cfg(9, C, N, Bq * H * W), 'dgrad'), dx, 0,
                        accumulate=accumulate, ymask=ymask, ymul=ymul)

    # ---- dropout state ----
    def drop_state(self, device):
        """The device-side {seed, step} of this module's dropout (created on first use from ``torch.initial_seed()``; re-derived,
        step 0, when the process was re-seeded with ANOTHER seed since -- ``torch.manual_seed`` keeps working as the one seed of a run,
        and the per-rank seed offsets of ``trainer.attach_data_parallel`` give every rank its own masks; re-seeding with the same
        value is not observable here and continues the stream: ``set_dropout_rng(seed, 0, device)`` restarts it explicitly)."""
        seed = torch.initial_seed()
        stale = self._drop is not None and (self._drop[1].state.device != torch.device(device) or self._drop[1].p != self.dropout_prob)
        # a RESTORED stream (set_dropout_rng: checkpoint resume) is not tied to torch's seed any more: the per-rank re-seeding of a later
        # attach_data_parallel, or a torch.manual_seed of the training script, must not silently restart it at step 0
        reseeded = self._drop is not None and not self._drop_restored and self._drop[0] != seed
        if self._drop is None or stale or reseeded:
            self._drop = (seed, ops.DropState(self.dropout_prob, seed ^ 0x5851f42d4c957f2d, device))
            self._drop_restored = False
        return self._drop[1]

    def get_dropout_rng(self):
        """(seed, step) of the dropout stream or None (checkpoints)."""
        return None if self._drop is None else self._drop[1].get()

    def set_dropout_rng(self, seed, step, device):
        """Restore a saved dropout stream (checkpoint resume); it stays in force -- also across a later ``torch.manual_seed`` / the
        per-rank seed offsets of ``attach_data_parallel`` -- until ``set_dropout_rng`` is called again."""
        d = ops.DropState(self.dropout_prob, 0, device)
        d.set(seed, step)
        self._drop = (torch.initial_seed(), d)
        self._drop_restored = True          # (drop_state() keeps it whatever happens to torch's seed; set_dropout_rng again replaces it)

    def invalidate_plans(self):
        """Drop every packed / transformed weight copy.  The caches notice optimizer steps, ``load_state_dict``, ``.to()``
        and any other in-place op on the parameters (version counter / data pointer); a write through ``param.data``
        (EMA, manual ``p.data.copy_``) moves neither -- call this after one."""
        self._plans.clear(); self._fused_plans.clear(); self._wino_plans.clear()
        for m in self.modules():
            if isinstance(m, _ConvParams):
                m._own_plan = None

    def refresh_plans(self):
        """Re-pack every cached plan whose parameter changed since it was packed (after an optimizer step that
        is all of them) with ONE batched kernel launch instead of one launch per plan."""
        stale, dg, keys = [], [], []
        for key, (ver, plan) in self._plans.items():
            name, _cfg, direction = key
            name = name.split('@')[0]                  # 'N.squeeze@pool': the same module packed for the fused pool+squeeze
            mod = self.convdet if name == 'convdet' else getattr(self.features[int(name.split('.')[0])], name.split('.')[1]) \
                if '.' in name else self.features[int(name)]
            now = (mod.weight._version, mod.weight.data_ptr(), mod.bias._version, mod.bias.data_ptr())
            if now != ver:
                stale.append((plan, mod.weight)); dg.append(direction != 'fwd'); keys.append((key, now, mod))
        self._refresh_wino_plans()
        self._refresh_bridge_plans()
        if not stale:
            return
        if self._pack_table_keepalive is not None and len(self._pack_table_keepalive) > 8:
            self._pack_table_keepalive = self._pack_table_keepalive[-4:]
        table = ops.repack_batched(stale, dg)
        self._pack_table_keepalive = (self._pack_table_keepalive or []) + [table]     # keep the descriptor table alive until consumed
        for (key, now, mod), (plan, _w) in zip(keys, stale):
            if plan.bias is not None:
                plan.bias = mod.bias.detach()
            self._plans[key] = (now, plan)

    def _refresh_wino_plans(self):
        """The same for the Winograd plans: every transformed-weight copy whose parameter changed, one launch."""
        stale, dg, keys = [], [], []
        for key, (ver, plan) in self._wino_plans.items():
            name, _cfg, direction = key
            mod = self.convdet if name == 'convdet' else getattr(self.features[int(name.split('.')[0])], name.split('.')[1])
            now = (mod.weight._version, mod.weight.data_ptr(), mod.bias._version, mod.bias.data_ptr())
            if now != ver:
                stale.append((plan, mod.weight)); dg.append(direction != 'fwd'); keys.append((key, now, mod))
        if not stale:
            return
        table = ops.repack_wino_batched(stale, dg)
        self._pack_table_keepalive = (self._pack_table_keepalive or [])[-6:] + [table]
        for (key, now, mod), (plan, _w) in zip(keys, stale):
            if plan.bias is not None:
                plan.bias = mod.bias.detach()
            self._wino_plans[key] = (now, plan)

    def wgrad_batch(self, entries_fn, key):
        """Cached ops.WgradBatch for one set of layer shapes (``entries_fn()`` builds the entry list on a miss)."""
        hit = self._wgrad_batches.get(key)
        if hit is None:
            if len(self._wgrad_batches) > 4:
                self._wgrad_batches.clear()
            entries, slots, total = entries_fn()
            hit = (ops.WgradBatch(entries, self.convdet.weight.device), slots, total)
            self._wgrad_batches[key] = hit
        return hit

    def forward(self, x):
        from .autograd import backbone_apply
        return backbone_apply(self, x)


class PredictionResolver(nn.Module):
    def __init__(self, cfg, log_softmax=False):
        super().__init__()
        self.log_softmax = log_softmax
        self.input_size = cfg.input_size
        self.num_classes = cfg.num_classes
        self.anchors = torch.from_numpy(np.asarray(cfg.anchors)).unsqueeze(0).float()
        self.anchors_per_grid = cfg.anchors_per_grid
        self._dev_anchors = {}

    def anchors_on(self, device):
        a = self._dev_anchors.get(device)
        if a is None:
            a = self.anchors[0].to(device).contiguous()
            self._dev_anchors[device] = a
        return a

    def forward(self, pred):
        """Same five outputs as the reference (src/model/squeezedet.py:109-120): (pred_class_probs [B,A,C],
        pred_log_class_probs | None, pred_scores [B,A,1], pred_deltas [B,A,4], pred_boxes [B,A,4]) -- inference only
        (not differentiable; the training path differentiates through ``Loss``, whose kernel has the analytic backward)."""
        return ops.resolve(pred.detach(), self.anchors_on(pred.device), self.input_size, self.num_classes, self.log_softmax)

    def decode(self, pred):
        """Fused resolver + ``probs *= score; argmax; max`` of ``SqueezeDet.forward`` (:199-202):
        (class_ids int64 [B,A], scores [B,A], boxes [B,A,4])."""
        return ops.decode(pred, self.anchors_on(pred.device), self.input_size, self.num_classes)


class SqueezeDet(nn.Module):
    """ Model for inference """

    def __init__(self, cfg):
        super().__init__()
        self.base = SqueezeDetBase(cfg)
        self.resolver = PredictionResolver(cfg, log_softmax=False)

    def forward(self, batch):
        pred = self.base(batch['image'])
        class_ids, scores, boxes = self.resolver.decode(pred)
        return {'class_ids': class_ids, 'scores': scores, 'boxes': boxes}


class Loss(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.resolver = PredictionResolver(cfg, log_softmax=True)
        self.num_anchors = cfg.num_anchors
        self.class_loss_weight = cfg.class_loss_weight
        self.positive_score_loss_weight = cfg.positive_score_loss_weight
        self.negative_score_loss_weight = cfg.negative_score_loss_weight
        self.bbox_loss_weight = cfg.bbox_loss_weight

    def forward(self, pred, gt):
        from .autograd import loss_apply
        return loss_apply(self, pred, gt)

    def mean_loss(self, pred, gt):
        """``forward(pred, gt)[0].mean()`` and the (detached) statistics as one autograd node whose forward and backward are the loss
        kernels alone (backward.LossMeanFn): -> (mean total loss, 0-dim; stats dict of per-image vectors)."""
        from .backward import LossMeanFn
        mean, vec = LossMeanFn.apply(pred, gt, self.resolver.anchors_on(pred.device), self)
        return mean, {'loss': vec[3], 'class_loss': vec[0], 'score_loss': vec[1], 'bbox_loss': vec[2]}


class SqueezeDetWithLoss(nn.Module):
    """ Model for training """

    def __init__(self, cfg):
        super().__init__()
        self.base = SqueezeDetBase(cfg)
        self.loss = Loss(cfg)

    def forward(self, batch):
        pred = self.base(batch['image'])
        loss, loss_stats = self.loss(pred, batch['gt'])
        return loss, loss_stats

    def forward_mean(self, batch):
        """``forward(batch)[0].mean()`` (the scalar the reference's trainer differentiates, src/engine/trainer.py:43) + the per-image
        statistics, with the mean and its backward inside the loss kernels (no torch kernel between loss and backbone)."""
        pred = self.base(batch['image'])
        return self.loss.mean_loss(pred, batch['gt'])
