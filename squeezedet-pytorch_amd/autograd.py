"""Forward / backward executors behind ``torch.autograd.Function`` so that ``SqueezeDetBase`` and
``Loss`` keep the reference's ``nn.Module`` surface (src/model/squeezedet.py:79-87, :133-174) while
every FLOP runs in the HIP kernels.

The whole backbone is ONE autograd node: its forward walks the layer table launching
stem -> pool -> (squeeze, expand1x1, expand3x3)* -> ConvDet on the current stream, keeping NHWC
activations; its backward walks the table in reverse (dgrad = the same implicit-GEMM kernel with
transposed/flipped weights and the ReLU mask applied while staging dY; wgrad / bias-grad kernels)
and returns the gradients of the 64 canonical OIHW parameters.
"""
from __future__ import annotations


import torch

from . import ops
from .synthetic import layer_table


def _needs_grad(base):
    return torch.is_grad_enabled() and any(p.requires_grad for p in base.parameters())


def run_backbone_forward(base, image, save=False, drop_mask=None, drop=None):
    """Launch the forward plan.  Returns (pred_nhwc [B,H,W,A_per_cell*(C+5)], saved dict | None).  Dropout in front of ConvDet
    (training): ``drop`` (an ops.DropState) applies it inside the last Fire's expand launches, or ``drop_mask`` (a scaled keep mask,
    NHWC) multiplies it in."""
    if not image.is_cuda:
        raise RuntimeError('SqueezeDetBase runs on the MI355X HIP kernels only: input must be a CUDA/HIP tensor')
    if image.dtype != torch.float32:
        raise RuntimeError('SqueezeDetBase expects fp32 input')
    layers = layer_table(base.arch)
    feats = base.features
    B = image.shape[0]
    saved = {} if save else None
    base.refresh_plans()                       # one batched re-pack if the optimizer touched the parameters
    stem = feats[0]
    first = 2
    if layers[2][0] == 'pool':
        # conv + ReLU + pool fused: the 30.7 MB/image stem output never reaches HBM.  Training keeps the pool
        # argmax; the backward folds ReLU + pool into the stem weight-gradient kernel (ops.stem_wgrad_pooled)
        Hs, Ws = ops.stem_out_size(image.shape[2], image.shape[3], stem.kernel_size[0])
        am = torch.empty(B, *ops.pool_out_size(Hs, Ws), stem.out_channels, device=image.device, dtype=torch.uint8) if save else None
        nxt = layers[3] if len(layers) > 3 else None
        if (not save and base.fuse_stem_squeeze and nxt is not None and nxt[0] == 'fire'
                and ops.stem_pool_squeeze_ok(image.shape, stem.weight.shape, nxt[2])):
            # inference: the first Fire's squeeze rides in the stem launch; the pooled 64-channel tensor (its only consumer) is
            # never written
            fsq = feats[3].squeeze
            stem_sq = ops.stem_pool_squeeze(image, stem.weight, stem.bias, fsq.weight, fsq.bias)
            a = None
        elif (save and base.fuse_train_forward and base.fuse_stem_squeeze and nxt is not None and nxt[0] == 'fire'
                and ops.stem_pool_squeeze_ok(image.shape, stem.weight.shape, nxt[2])):
            # training: the same, but the pooled tensor and its codes are stored as well (the backward reads them)
            fsq = feats[3].squeeze
            stem_sq, a = ops.stem_pool_squeeze(image, stem.weight, stem.bias, fsq.weight, fsq.bias, argmax=am)
        else:
            stem_sq = None
            a = ops.stem_pool(image, stem.weight, stem.bias, argmax=am)
        first = 3
        if save:
            saved['stem_pool'] = (am, a)
    else:
        a = ops.stem_conv_relu(image, stem.weight, stem.bias)
        if save:
            saved['stem_out'] = a
    if save:
        saved['image'] = image
    drop_applied = False
    unpooled = None                            # inference: a pool whose output only feeds the next squeeze is folded into it
    bridged = None                             # inference: the next Fire's squeeze output, produced by the previous Fire's launch
    if layers[2][0] == 'pool' and stem_sq is not None:
        bridged = stem_sq                      # ... or by the stem's
    for i in range(first, len(layers)):
        l = layers[i]
        if l[0] == 'pool':
            if bridged is not None:                # the previous Fire's launch already pooled and squeezed
                continue
            Bq, H, W, C = a.shape
            nxt = layers[i + 1] if i + 1 < len(layers) else None
            if (not save and base.fuse_pool_squeeze and nxt is not None and nxt[0] == 'fire' and ops.pool_squeeze_ok(C, nxt[2])):
                unpooled = a
                continue
            am = torch.empty(Bq, *ops.pool_out_size(H, W), C, device=a.device, dtype=torch.uint8) if save else None
            # training: the codes carry the ReLU mask of the pool's input (a Fire output), the backward reads no mask tensor
            y = ops.maxpool(a, argmax=am, relu_codes=save)
            if save:
                saved[f'pool{i}'] = (am, (H, W))
            a = y
        else:
            _, cin, s, e1, e3 = l
            fire = feats[i]
            if bridged is not None:
                Bq, H, W, _ = bridged.shape
                C = cin
            elif unpooled is not None:
                Bq, Hu, Wu, C = unpooled.shape
                H, W = ops.pool_out_size(Hu, Wu)
            else:
                Bq, H, W, C = a.shape
            assert C == cin, f'layer {i}: expected {cin} channels, got {C}'
            npix = Bq * H * W
            fusable = not save and not ((drop_mask is not None or drop is not None) and i == len(layers) - 1)
            nxt = layers[i + 1] if i + 1 < len(layers) else None
            nxt2 = layers[i + 2] if i + 2 < len(layers) else None
            # which launch takes this Fire's expand pair (pure table look-ups; decided before the squeeze because the squeeze may ride
            # with the expand1x1)
            zseg = ycfg = xcfg = fcfg = None
            if (fusable and nxt is not None and nxt[0] == 'pool' and nxt2 is not None and nxt2[0] == 'fire' and base.fuse_fire_bridge
                    and base.use_winograd):
                zseg = ops.choose_fire_pool_bridge(s, e1, e3, nxt2[2], npix)
            if zseg is None and fusable and nxt is not None and nxt[0] == 'fire' and base.fuse_fire_bridge and base.use_winograd:
                ycfg = ops.choose_fire_bridge_cfg(s, e1, e3, nxt[2], npix)
            # training: the Fire -> Fire bridge in its storing form (the expand output the backward needs is written by the same launch)
            ycfg_t = None
            if (save and base.fuse_train_forward and nxt is not None and nxt[0] == 'fire' and base.fuse_fire_bridge and base.use_winograd
                    and not ((drop_mask is not None or drop is not None) and i == len(layers) - 1)):
                ycfg_t = ops.choose_fire_bridge_cfg(s, e1, e3, nxt[2], npix)
                if ycfg_t is not None and ycfg_t % 1000 != 12:
                    ycfg_t = None
            zseg_t = None
            if (save and base.fuse_train_forward and nxt is not None and nxt[0] == 'pool' and nxt2 is not None and nxt2[0] == 'fire'
                    and base.fuse_fire_bridge and base.use_winograd):
                zseg_t = ops.choose_fire_pool_bridge(s, e1, e3, nxt2[2], npix)
            if zseg is None and ycfg is None:
                xcfg = ops.choose_fire_wino_cfg(s, e1, e3, npix) if (fusable and base.fuse_expand_wino and base.use_winograd) else None
                fcfg = ops.choose_fused_cfg(s, e1, npix) if (fusable and xcfg is None and base.fuse_expand and e1 == e3) else None
            ym = drop_mask if (drop_mask is not None and i == len(layers) - 1) else None
            dr = drop if (drop is not None and i == len(layers) - 1) else None
            dcfg = None
            if dr is not None:
                # the fused form needs a weight-stationary 1x1 configuration and the balanced Winograd kernel (8 | squeeze width);
                # otherwise this step's mask is drawn as a tensor by the stand-alone kernel and multiplied in like a given mask
                dcfg = ops.conv_drop_cfg(s, e1, npix) if (base.fused_dropout and s % 8 == 0 and base.use_winograd) else None
                if dcfg is None:
                    ym = ops.dropout_mask(dr, (Bq, H, W, e1 + e3))
                    drop_mask = ym
                    dr = None
            out = None
            if bridged is not None:
                sq, bridged = bridged, None
            else:
                sq = torch.empty(Bq, H, W, s, device=a.device, dtype=torch.float32)
                if unpooled is not None:
                    ops.pool_squeeze(unpooled, 0, cin, base.plan(f'{i}.squeeze@pool', fire.squeeze, ops.POOL_SQUEEZE_CFG), sq, 0)
                    unpooled = None
                else:
                    ops.conv(a, 0, base.plan(f'{i}.squeeze', fire.squeeze, ops.choose_cfg(1, cin, s, npix)), sq, 0, relu=True)
            if (fusable and nxt is not None and nxt[0] == 'pool' and nxt2 is not None and nxt2[0] == 'fire' and base.fuse_fire_bridge
                    and base.use_winograd):
                if zseg is not None:
                    # inference: expand pair + concat + the max pool + the squeeze of the Fire behind it in one launch
                    bridged = torch.empty(Bq, *ops.pool_out_size(H, W), nxt2[2], device=sq.device, dtype=torch.float32)
                    ops.fire_pool_bridge(sq, 0, base.fire_bridge_plan(i, fire, feats[i + 2], 12, pooled=True), bridged, 0, nseg=zseg)
                    a = None
                    continue
            if fusable and nxt is not None and nxt[0] == 'fire' and base.fuse_fire_bridge and base.use_winograd:
                if ycfg is not None:
                    # inference: this Fire's expand pair AND the next Fire's squeeze in one launch; the concatenated expand
                    # output (the next layer's only consumer is that squeeze) is never written
                    bridged = torch.empty(Bq, H, W, nxt[2], device=sq.device, dtype=torch.float32)
                    ops.fire_bridge(sq, 0, base.fire_bridge_plan(i, fire, feats[i + 1], ycfg), bridged, 0)
                    a = None
                    continue
            if zseg_t is not None:
                # training: expand pair + concat + max pool + the next squeeze in one launch; what the backward reads of this stage -- the
                # pooled tensor (the next squeeze's input) and the pool's arg-max / ReLU codes -- is stored by it, the unpooled expand
                # output is never written
                Hp, Wp = ops.pool_out_size(H, W)
                pooled = torch.empty(Bq, Hp, Wp, e1 + e3, device=sq.device, dtype=torch.float32)
                am = torch.empty(Bq, Hp, Wp, e1 + e3, device=sq.device, dtype=torch.uint8)
                bridged = torch.empty(Bq, Hp, Wp, nxt2[2], device=sq.device, dtype=torch.float32)
                ops.fire_pool_bridge(sq, 0, base.fire_bridge_plan(i, fire, feats[i + 2], 12, pooled=True), bridged, 0, nseg=zseg_t,
                                     save=pooled, codes=am, save_coff1=0, save_coff3=e1)
                saved[f'fire{i}'] = (a, sq, torch.empty(Bq, H, W, e1 + e3, device='meta'))       # (only the shape of the expand output is read)
                saved[f'pool{i + 1}'] = (am, (H, W))
                a = pooled
                continue
            if ycfg_t is not None:
                out = torch.empty(Bq, H, W, e1 + e3, device=sq.device, dtype=torch.float32)
                bridged = torch.empty(Bq, H, W, nxt[2], device=sq.device, dtype=torch.float32)
                ops.fire_bridge(sq, 0, base.fire_bridge_plan(i, fire, feats[i + 1], ycfg_t), bridged, 0, save=out, save_coff1=0, save_coff3=e1)
                saved[f'fire{i}'] = (a, sq, out)
                a = out
                continue
            if out is None:
                out = torch.empty(Bq, H, W, e1 + e3, device=sq.device, dtype=torch.float32)
            if xcfg is not None:
                # inference: both expands in ONE Winograd launch (expand1x1 = the four inner transform positions, riding
                # along as extra channel slices on the same staged squeeze tile)
                ops.fire_wino(sq, 0, base.fire_wino_plan(i, fire, xcfg), out, 0, e1)
            elif fcfg is not None:
                # inference: both expands in one launch (they read the same squeeze tile; the 1x1 rides along as extra
                # channel groups that only run the centre tap)
                ops.fire_expand(sq, 0, base.fused_expand_plan(i, fire, fcfg), out, 0)
            elif dr is not None:
                # dropout in front of ConvDet (reference: squeezedet.py:81-82) inside the two expand launches: the keep decision of an
                # element is a function of (seed, step, its index in `out`), evaluated in the epilogue -- no mask tensor
                ops.conv(sq, 0, base.plan(f'{i}.expand1x1', fire.expand1x1, dcfg), out, 0, relu=True, drop=dr)
                ops.conv_wino(sq, 0, base.wino_plan(f'{i}.expand3x3', fire.expand3x3, ops.WINO_SK_CFG), out, e1, relu=True, drop=dr)
                drop_applied = True
            else:
                # ... or as a given mask: relu(x) * m == relu(x * m) for the non-negative scaled keep mask, so it is the `ymul`
                # epilogue of the last Fire's two expand kernels -- no extra pass
                ops.conv(sq, 0, base.plan(f'{i}.expand1x1', fire.expand1x1, ops.choose_cfg(1, s, e1, npix)), out, 0, relu=True,
                         ymul=ym, ymul_coff=0)
                base.conv3x3(f'{i}.expand3x3', fire.expand3x3, sq, 0, out, e1, relu=True, ymul=ym)
                if ym is not None:
                    drop_applied = True
            if save:
                saved[f'fire{i}'] = (a, sq, out)
            a = out
    if drop is not None and not drop_applied and drop_mask is None:
        drop_mask = ops.dropout_mask(drop, tuple(a.shape))
    if drop_mask is not None and not drop_applied:
        a = a * drop_mask                      # (layer tables that do not end in a Fire: elementwise fallback)
    Bq, H, W, C = a.shape
    cd = base.convdet
    pred = torch.empty(Bq, H, W, cd.out_channels, device=a.device, dtype=torch.float32)
    fused_rng = drop is not None and drop_applied and drop_mask is None
    base.conv3x3('convdet', cd, a, 0, pred, 0, relu=False)
    if drop is not None:
        # this forward's mask is consumed: step += 1 on the device.  (The balanced Winograd kernel can carry the advance inside its
        # launch -- ops.conv_wino(..., drop_advance=) -- but measured inside the step it runs ConvDet's forward slower than the unit
        # kernel of the table, 219-230 against 208-216 us, which costs more than this one-thread launch.)
        ops.dropout_advance(drop)
    if save:
        saved['convdet_in'] = a
        saved['drop_mask'] = drop_mask
        # with the fused form the backward needs no mask: convdet_in > 0 exactly where the element was kept AND its ReLU was active
        saved['drop_scale'] = float(drop.scale) if fused_rng else None
    return pred, saved


def _make_drop_mask(base, like_nhwc_shape, device):
    """An injected mask (tests: NCHW, already scaled by 1 / (1 - p)) in the layout the epilogues read."""
    return base._forced_drop_mask.to(device).permute(0, 2, 3, 1).contiguous()


def _feature_shape(base, image):
    """NHWC shape of the feature map entering ConvDet (needed to draw the dropout mask up front)."""
    h, w = image.shape[2], image.shape[3]
    layers = layer_table(base.arch)
    h, w = ops.stem_out_size(h, w, layers[0][3])
    c = layers[0][2]
    for l in layers[2:]:
        if l[0] == 'pool':
            h, w = ops.pool_out_size(h, w)
        else:
            c = l[3] + l[4]
    return (image.shape[0], h, w, c)


def backbone_apply(base, image):
    train_drop = base.training and base.dropout is not None
    drop_mask = drop = None
    if train_drop:
        if base._forced_drop_mask is not None:
            drop_mask = _make_drop_mask(base, _feature_shape(base, image), image.device)
        else:
            drop = base.drop_state(image.device)
    if _needs_grad(base):
        from .backward import BackboneFn
        params = [p for _, p in base.named_parameters()]
        pred = BackboneFn.apply(base, image, drop_mask, drop, *params)
    else:
        pred, _ = run_backbone_forward(base, image, save=False, drop_mask=drop_mask, drop=drop)
    B = pred.shape[0]
    out = pred.view(B, -1, base.num_classes + 5)
    if out.shape[1] != base.num_anchors:
        raise RuntimeError(f'input size yields {out.shape[1]} anchors but cfg.num_anchors is {base.num_anchors}')
    return out


def loss_apply(loss_mod, pred, gt):
    from .backward import LossFn
    anchors = loss_mod.resolver.anchors_on(pred.device)
    vec = LossFn.apply(pred, gt, anchors, loss_mod)
    # vec: [4, B] = (class, pos+neg score, bbox, total) -- reference returns (loss, stats dict), :166-174
    class_loss, score_loss, bbox_loss, loss = vec[0], vec[1], vec[2], vec[3]
    return loss, {'loss': loss, 'class_loss': class_loss, 'score_loss': score_loss, 'bbox_loss': bbox_loss}
