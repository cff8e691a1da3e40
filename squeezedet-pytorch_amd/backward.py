"""Backward executors: ``BackboneFn`` (one autograd node for stem..ConvDet) and ``LossFn``.

Reference behaviour reproduced: autograd through ``SqueezeDetBase.forward`` (src/model/squeezedet.py:79-87)
and ``Loss.forward`` (:133-174) as triggered by ``loss.backward()`` in src/engine/trainer.py:47.

Gradient flow convention: every gradient tensor materialised in HBM already carries the ReLU mask of
the activation it belongs to -- the kernel that *produces* it applies the mask in its epilogue
(``ymask``), so the (up to four) consumers of that gradient read it once each without the mask tensor.
"""
from __future__ import annotations

import torch

from . import ops
from .autograd import run_backbone_forward
from .synthetic import layer_table


def _wgrad_layout(base, saved, B, H, W):
    """Flat gradient layout (named_parameters order: every gradient is a view of ONE buffer) and the batched-reduction
    entries of all Fire / ConvDet weight gradients for the shapes of this step."""
    slots, off = {}, 0
    for n, p in base.named_parameters():
        slots[n] = (off, tuple(p.shape)); off += p.numel()
    layers = layer_table(base.arch)
    entries = []
    # expand3x3 weight gradients that share a launch: the Fire modules between two pools (same grid), same tile form
    e3s = [(f'features.{i}.expand3x3', layers[i][4], layers[i][2]) + tuple(saved[f'fire{i}'][2].shape[:3])
           for i in range(len(layers) - 1, 1, -1) if layers[i][0] == 'fire']
    groups = ops.wino_wgrad_groups(e3s, enabled=getattr(base, 'group_wgrad', None))
    # ... and the expand1x1 weight gradients that are too wide for the fused squeeze backward (their own direct-form launch otherwise)
    fsb = bool(getattr(base, 'fuse_squeeze_bwd', False))
    e1s = [(f'features.{i}.expand1x1', layers[i][3], layers[i][2]) + tuple(saved[f'fire{i}'][2].shape[:3])
           for i in range(len(layers) - 1, 1, -1) if layers[i][0] == 'fire' and not (fsb and ops.squeeze_bwd_ok(layers[i][3], layers[i][2]))]
    groups.update(ops.wgrad1x1_groups(e1s, enabled=getattr(base, 'group_wgrad', None)))

    def add(pre, N, C, taps, shp, fused=False):
        entries.append((pre, N, C, taps, shp[0], shp[1], shp[2], slots[pre + '.weight'][0], slots[pre + '.bias'][0], fused,
                        None if fused else groups.get(pre)))
    add('convdet', base.convdet.out_channels, base.convdet.in_channels, 9, (B, H, W))
    for i in range(len(layers) - 1, 1, -1):
        if layers[i][0] != 'fire':
            continue
        _, cin, s, e1, e3 = layers[i]
        shp = saved[f'fire{i}'][2].shape
        add(f'features.{i}.expand1x1', e1, s, 1, shp, fused=bool(getattr(base, 'fuse_squeeze_bwd', False)) and ops.squeeze_bwd_ok(e1, s))
        add(f'features.{i}.expand3x3', e3, s, 9, shp)
        add(f'features.{i}.squeeze', s, cin, 1, shp, fused=bool(getattr(base, 'fuse_squeeze_bwd', False)) and ops.squeeze_bwd_ok(s, cin))
    return entries, slots, off


def run_backbone_backward(base, saved, dpred):
    """dpred: NHWC [B,H,W,anchors_per_grid*(C+5)].  Returns {param_name (relative to base): grad}; every gradient is a
    view of one flat fp32 buffer in ``named_parameters`` order (``base.last_grad_flat``)."""
    layers = layer_table(base.arch)
    feats = base.features
    dpred = dpred.contiguous()
    B, H, W, ncd = dpred.shape
    cd = base.convdet
    a_in = saved['convdet_in']
    cin_cd = a_in.shape[3]
    shape_key = (B, H, W, bool(getattr(base, 'fuse_squeeze_bwd', False)), getattr(base, 'group_wgrad', None)) + tuple(tuple(saved[f'fire{i}'][2].shape) for i in range(len(layers)) if layers[i][0] == 'fire')
    wb, slots, total = base.wgrad_batch(lambda: _wgrad_layout(base, saved, B, H, W), shape_key)
    # + 1: the data-parallel exchange carries this rank's image count through the same all-reduce (trainer.GradientExchange)
    grad_buf = torch.empty(total + 1, device=dpred.device, dtype=torch.float32)
    grad_flat = grad_buf[:total]
    sync = getattr(base, 'grad_sync', None)
    if sync is not None:
        sync.begin(grad_buf, total, B)
    # Stages of the backward = runs of layers between two pools.  With a gradient exchange attached, each stage's slabs
    # are reduced as soon as the stage is done and its slice of the flat buffer (named_parameters order: a stage is a
    # contiguous range, later stages of the network sit at higher offsets) is handed to the all-reduce; without one, all
    # slabs are reduced by a single launch at the end.
    stage = {'row': 0, 'hi': total}
    pending = {}                                           # group id -> the members seen so far (ops.conv_wgrad_wino_group)

    def close_stage(first_param):
        """Everything from ``first_param`` to the previous stage's start is final once the pending slabs are reduced."""
        if sync is None:
            return
        row_hi = wb.row_of[first_param.rsplit('.', 1)[0]] + 1 if first_param is not None else wb.nrows
        if row_hi > stage['row']:
            wb.reduce(grad_flat, stage['row'], row_hi, scale=sync.scale)     # (weighted by this rank's image count on the way out)
            stage['row'] = row_hi
        lo = slots[first_param][0] if first_param is not None else 0
        sync.ready(lo, stage['hi'])
        stage['hi'] = lo

    def gview(name):
        off, shape = slots[name]
        n = 1
        for d in shape:
            n *= d
        return grad_flat[off:off + n].view(shape)
    ops.conv_wgrad(dpred, 0, ncd, a_in, 0, cin_cd, 9, slab=wb.slab('convdet'))
    last = len(layers) - 1
    assert layers[last][0] == 'fire'
    out_last = saved[f'fire{last}'][2]
    dA = torch.empty_like(out_last)
    if saved.get('drop_scale') is not None:
        # fused dropout: out_last IS the dropped ReLU output (> 0 exactly where kept and active), so the gradient is masked by it and
        # scaled by the constant 1 / (1 - p): no mask tensor is read
        if ncd % 8 == 0:
            ops.conv_wino(dpred, 0, base.wino_plan('convdet', cd, ops.WINO_SK_CFG, 'dgrad'), dA, 0, ymask=out_last, yscale=saved['drop_scale'])
        else:
            base.dgrad3x3('convdet', cd, dpred, 0, dA, ymask=out_last)
            dA.mul_(saved['drop_scale'])
    else:
        base.dgrad3x3('convdet', cd, dpred, 0, dA, ymul=saved['drop_mask'], ymask=out_last)
    for i in range(last, 1, -1):
        l = layers[i]
        if l[0] == 'pool':
            close_stage(f'features.{i + 1}.squeeze.weight')    # the Fire modules behind this pool are done
            if i == 2 and 'stem_pool' in saved:
                continue                                   # folded into the stem weight gradient below
            am, (Hi, Wi) = saved[f'pool{i}']
            # the forward recorded the ReLU mask of the pool's input inside the arg-max codes (ops.maxpool(relu_codes=True)):
            # the gradient that leaves here already carries it, and no activation tensor is re-read for its sign
            dA = ops.maxpool_bwd(dA, am, (Hi, Wi))
            continue
        _, cin, s, e1, e3 = l
        fire = feats[i]
        x_in, sq, out = saved[f'fire{i}']
        Bq, Hq, Wq, _ = out.shape
        npix = Bq * Hq * Wq
        pre = f'features.{i}.'
        fused_e1 = bool(wb.fused.get(pre + 'expand1x1'))

        def grouped(key, item, run_group):
            # a member of a launch group runs with the other members, as soon as the last of them has its gradient (dA stays alive until
            # then); returns False for a layer that has its own launch
            grp = wb.group_of.get(key)
            if grp is None:
                return False
            lst = pending.setdefault(grp[0], [])
            lst.append(item)
            if len(lst) == len(grp[3]):
                run_group(lst, grp)
                del pending[grp[0]]
            return True
        if not fused_e1 and not grouped(pre + 'expand1x1', (dA, 0, e1, sq, 0, s, wb.slab(pre + 'expand1x1')),
                                        lambda lst, grp: ops.conv_wgrad_group(lst, grp[1])):
            ops.conv_wgrad(dA, 0, e1, sq, 0, s, 1, slab=wb.slab(pre + 'expand1x1'))
        if not grouped(pre + 'expand3x3', (dA, e1, e3, sq, 0, s, wb.slab(pre + 'expand3x3')),
                       lambda lst, grp: ops.conv_wgrad_wino_group(lst, grp[1], grp[2])):
            ops.conv_wgrad(dA, e1, e3, sq, 0, s, 9, slab=wb.slab(pre + 'expand3x3'))
        dSq = torch.empty_like(sq)
        if fused_e1:
            # narrow expand1x1 (N <= 128: the first four Fires): weight-gradient slabs and the data gradient from ONE pass over the
            # expand1x1 half of dA; the expand3x3 data gradient below accumulates onto it and applies the squeeze's ReLU mask
            ops.squeeze_bwd(dA, sq, fire.expand1x1.weight, wb.slab(pre + 'expand1x1'), dSq, relu_mask=False, dy_coff=0, N=e1)
        else:
            ops.conv(dA, 0, base.plan(f'{i}.expand1x1', fire.expand1x1, ops.choose_cfg(1, e1, s, npix), 'dgrad'), dSq, 0)
        base.dgrad3x3(f'{i}.expand3x3', fire.expand3x3, dA, e1, dSq, accumulate=True, ymask=sq)
        dIn = torch.empty_like(x_in)
        prev_is_fire = layers[i - 1][0] == 'fire'
        if wb.fused.get(pre + 'squeeze'):
            # weight gradient slabs AND the data gradient in one launch: x_in is streamed once (it used to be read by the weight
            # gradient, and again -- for its sign only -- by the data-gradient kernel's ReLU-mask epilogue)
            ops.squeeze_bwd(dSq, x_in, fire.squeeze.weight, wb.slab(pre + 'squeeze'), dIn, relu_mask=prev_is_fire)
        else:
            ops.conv_wgrad(dSq, 0, s, x_in, 0, cin, 1, slab=wb.slab(pre + 'squeeze'))
            ops.conv(dSq, 0, base.plan(f'{i}.squeeze', fire.squeeze, ops.choose_cfg(1, s, cin, npix), 'dgrad'), dIn, 0,
                     ymask=x_in if prev_is_fire else None)
        dA = dIn
    stem = feats[0]
    stem_out = (gview('features.0.weight'), gview('features.0.bias'))
    if 'stem_pool' in saved:
        am, pooled = saved['stem_pool']
        # (the codes carry the ReLU mask: the pooled tensor is not read again)
        ops.stem_wgrad_pooled(dA, None, am, saved['image'].contiguous(), stem.out_channels, stem.kernel_size[0], out=stem_out)
    else:
        ops.stem_wgrad(dA, saved['image'].contiguous(), stem.out_channels, stem.kernel_size[0], out=stem_out)
    if sync is not None:
        lo0 = slots['features.0.weight'][0]
        sync.scale_slice(lo0, lo0 + stem.weight.numel() + stem.bias.numel())      # the stem's gradient does not pass through the slab reduction
    assert not pending, 'a weight-gradient group was left incomplete'
    if sync is None:
        wb.reduce(grad_flat)                               # all 31 Fire / ConvDet slab reductions: one launch
    else:
        close_stage(None)                                  # whatever is left (first stage + stem)
        sync.finish()
    base.last_grad_flat = grad_flat
    return {n: gview(n) for n in slots}


class BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, base, image, drop_mask, drop, *params):
        pred, saved = run_backbone_forward(base, image.detach(), save=True, drop_mask=drop_mask, drop=drop)
        ctx.base = base
        ctx.saved = saved
        ctx.names = [n for n, _ in base.named_parameters()]
        return pred

    @staticmethod
    def backward(ctx, dpred):
        grads = run_backbone_backward(ctx.base, ctx.saved, dpred)
        ctx.saved = None
        return (None, None, None, None) + tuple(grads[n] for n in ctx.names)


class LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, anchors, loss_mod):
        res = loss_mod.resolver
        weights = (loss_mod.class_loss_weight, loss_mod.positive_score_loss_weight,
                   loss_mod.negative_score_loss_weight, loss_mod.bbox_loss_weight)
        losses, nobj = ops.loss_fwd(pred.detach(), gt, anchors, res.input_size, res.num_classes, weights)
        ctx.save_for_backward(pred.detach(), gt, anchors, nobj)
        ctx.meta = (res.input_size, res.num_classes, weights)
        return losses                      # [4,B] = (class, score, bbox, total)

    @staticmethod
    def backward(ctx, g):
        pred, gt, anchors, nobj = ctx.saved_tensors
        input_size, num_classes, weights = ctx.meta
        coef = (g[:3] + g[3:4]).contiguous()       # gradient of `total` reaches all three components
        dpred = ops.loss_bwd(pred, gt, anchors, nobj, coef, input_size, num_classes, weights)
        return dpred, None, None, None


class LossMeanFn(torch.autograd.Function):
    """``Loss(pred, gt)[0].mean()`` (src/engine/trainer.py:43) as ONE autograd node: the batch mean comes out of the loss launch, its
    backward hands the scalar gradient straight to the loss backward kernel -- no torch reduction / select / elementwise kernels
    between the loss and the backbone's backward."""

    @staticmethod
    def forward(ctx, pred, gt, anchors, loss_mod):
        res = loss_mod.resolver
        weights = (loss_mod.class_loss_weight, loss_mod.positive_score_loss_weight,
                   loss_mod.negative_score_loss_weight, loss_mod.bbox_loss_weight)
        losses, nobj, mean4 = ops.loss_mean_fwd(pred.detach(), gt, anchors, res.input_size, res.num_classes, weights)
        ctx.save_for_backward(pred.detach(), gt, anchors, nobj)
        ctx.meta = (res.input_size, res.num_classes, weights)
        ctx.mark_non_differentiable(losses)
        ctx.set_materialize_grads(False)   # (no zero-filled gradient tensor for the statistics output: that is a fill launch per step)
        return mean4[3], losses            # (0-dim view of the total's mean; the per-image vectors for the statistics)

    @staticmethod
    def backward(ctx, g, _gl):
        pred, gt, anchors, nobj = ctx.saved_tensors
        input_size, num_classes, weights = ctx.meta
        return ops.loss_mean_bwd(pred, gt, anchors, nobj, g.reshape(1), input_size, num_classes, weights), None, None, None
