"""The launch plan of one inference or training step, computed on the host WITHOUT a GPU: which kernel instance (tile
configuration from the measured table ``tuning.json``) every layer launches for a given architecture / batch / input
size.  ``inference_launch_plan`` mirrors the decisions of ``autograd.run_backbone_forward`` + the fused detect launch,
``training_launch_plan`` those of the saving forward, ``LossFn`` and ``backward.run_backbone_backward``; GPU tests assert
both agree with the real launches name for name and shape for shape (tests/test_surface_gpu.py), for the default and for
non-default model flags.  They exist so that profiles can be checked against the code that shipped:
``profiles/traffic.json`` records the launch set it was measured on (inference at top level, training under ``"train"``)
and a CPU test recomputes that set here.

Reference for the layer sequence: src/model/squeezedet.py:33-87, src/engine/detector.py:20-50, src/engine/trainer.py:42-50.
"""
from __future__ import annotations

from collections import Counter

from . import ops
from .synthetic import convdet_in_channels, layer_table


def _conv3x3(batch, H, W, Cin, N, use_winograd):
    npix = batch * H * W
    wc = ops.choose_wino_cfg(Cin, N, npix) if use_winograd else None
    if wc is not None:
        return (ops.wino_kernel_name(wc), f'9tap C{Cin} N{N} {H}x{W}')
    return (ops.cfg_kernel_name(ops.choose_cfg(9, Cin, N, npix)), f'9tap C{Cin} N{N} {H}x{W}')


def _conv1x1(batch, H, W, Cin, N):
    return (ops.cfg_kernel_name(ops.choose_cfg(1, Cin, N, batch * H * W)), f'1tap C{Cin} N{N} {H}x{W}')


def inference_launch_plan(arch='squeezedet', batch=20, input_size=(384, 1248), anchors_per_grid=9, num_classes=3,
                          use_winograd=True, fuse_expand=True, fuse_fire_bridge=True, fuse_expand_wino=True,
                          fuse_pool_squeeze=False, fuse_stem_squeeze=True):
    """-> list of (kernel name as bench.py / KernelTimer prints it, shape tag), in launch order.  The six switches are
    ``SqueezeDetBase``'s attributes of the same names, one to one."""
    layers = layer_table(arch)
    H, W = ops.stem_out_size(input_size[0], input_size[1], layers[0][3])
    C = layers[0][2]
    plan = []
    first = 2
    bridged = False
    if layers[2][0] == 'pool':
        nxt = layers[3] if len(layers) > 3 else None
        if (fuse_stem_squeeze and nxt is not None and nxt[0] == 'fire'
                and ops.stem_pool_squeeze_ok((batch, 3, input_size[0], input_size[1]), (layers[0][2], 3, layers[0][3], layers[0][3]), nxt[2])):
            plan.append((f'stem_pool_sq<{layers[0][3]}>', f'stem+pool+squeeze {input_size[0]}x{input_size[1]} S{nxt[2]}'))
            bridged = True
        else:
            plan.append((f'stem_pool<{layers[0][3]}>', f'stem+pool {input_size[0]}x{input_size[1]}'))
        H, W = ops.pool_out_size(H, W)
        first = 3
    else:
        plan.append((f'stem_conv<{layers[0][3]}>', f'stem {input_size[0]}x{input_size[1]}'))

    unpooled = None                        # (H, W) of the un-pooled map when the pool is folded into the next squeeze
    for i in range(first, len(layers)):
        l = layers[i]
        if l[0] == 'pool':
            nxt = layers[i + 1] if i + 1 < len(layers) else None
            if bridged:
                pass
            elif fuse_pool_squeeze and nxt is not None and nxt[0] == 'fire' and ops.pool_squeeze_ok(C, nxt[2]):
                unpooled = (H, W)
            else:
                plan.append(('maxpool_fwd', f'pool C{C} {H}x{W}'))
            H, W = ops.pool_out_size(H, W)
            continue
        _, cin, s, e1, e3 = l
        npix = batch * H * W
        nxt = layers[i + 1] if i + 1 < len(layers) else None
        nxt2 = layers[i + 2] if i + 2 < len(layers) else None
        zseg = ycfg = xcfg = fcfg = None
        if nxt is not None and nxt[0] == 'pool' and nxt2 is not None and nxt2[0] == 'fire' and fuse_fire_bridge and use_winograd:
            zseg = ops.choose_fire_pool_bridge(s, e1, e3, nxt2[2], npix)
        if zseg is None and nxt is not None and nxt[0] == 'fire' and fuse_fire_bridge and use_winograd:
            ycfg = ops.choose_fire_bridge_cfg(s, e1, e3, nxt[2], npix)
        if zseg is None and ycfg is None:
            xcfg = ops.choose_fire_wino_cfg(s, e1, e3, npix) if (fuse_expand_wino and use_winograd) else None
            fcfg = ops.choose_fused_cfg(s, e1, npix) if (xcfg is None and fuse_expand and e1 == e3) else None
        if bridged:
            pass
        elif unpooled is not None:
            plan.append(('pool_squeeze', f'pool+squeeze C{cin} N{s} {unpooled[0]}x{unpooled[1]}'))
            unpooled = None
        else:
            plan.append(_conv1x1(batch, H, W, cin, s))
        bridged = False
        C = e1 + e3
        if zseg is not None:
            plan.append(('fire_pool_bridge', f'fire C{s} E{e1}+{e3} -> pool -> S{nxt2[2]} {H}x{W}'))
            bridged = True
            continue
        if ycfg is not None:
            plan.append(('fire_bridge', f'fire C{s} E{e1}+{e3} -> S{nxt[2]} {H}x{W}'))
            bridged = True
            continue
        if xcfg is not None:
            plan.append((ops.fire_wino_kernel_name(xcfg), f'fire C{s} E{e1}+{e3} {H}x{W}'))
        elif fcfg is not None:
            plan.append((ops.cfg_kernel_name(fcfg).replace('conv_dma', 'fire_expand'), f'expand C{s} E{e1} {H}x{W}'))
        else:
            plan.append(_conv1x1(batch, H, W, s, e1))
            plan.append(_conv3x3(batch, H, W, s, e3, use_winograd))
    plan.append(_conv3x3(batch, H, W, convdet_in_channels(arch), anchors_per_grid * (num_classes + 5), use_winograd))
    plan.append(('detect', f'detect A{H * W * anchors_per_grid}'))
    return plan


def _wgrad(batch, H, W, N, C, taps):
    if ops.wgrad_uses_wino(N, C, taps, batch, H, W):
        return ('conv_wgrad_wino', f'wgrad 9tap C{C} N{N} {H}x{W}')
    return (f'conv_wgrad<{taps}>', f'wgrad {taps}tap C{C} N{N} {H}x{W}')


def training_launch_plan(arch='squeezedet', batch=20, input_size=(384, 1248), anchors_per_grid=9, num_classes=3,
                         use_winograd=True, data_parallel_stages=False, fuse_squeeze_bwd=True, dropout=True,
                         fused_dropout=True, fuse_train_forward=True, fuse_fire_bridge=True, fuse_stem_squeeze=True, group_wgrad=None):
    """Launches of one training iteration's forward (activations saved; ``fuse_train_forward``: the stem + squeeze launch and the
    two small-C bridges run in their STORING forms -- what the backward reads is written by the fused launch -- where the table has
    their rows; the other inference-only fusions stay off), multi-task loss forward / backward and the backbone backward, as
    (kernel name, shape tag) in launch order.  The optimizer launch and torch's own elementwise kernels (dropout mask,
    ``loss.mean()``) are not KernelTimer-bracketed and not listed.  ``data_parallel_stages``: with a gradient exchange
    attached the slab reduction runs once per backward stage instead of once at the end.  ``fuse_squeeze_bwd`` =
    ``SqueezeDetBase.fuse_squeeze_bwd``.  ``dropout`` (``cfg.dropout_prob > 0``): the counter-based dropout in front of ConvDet rides
    in the last Fire's expand launches (a weight-stationary 1x1 configuration + the balanced Winograd kernel) and ConvDet's data
    gradient runs on the balanced Winograd kernel (mask = its own input, constant scale); where that form does not apply (``fused_dropout`` off, squeeze width not a
    multiple of 8) the mask is drawn by the stand-alone ``dropout_mask`` launch."""
    layers = layer_table(arch)
    ks = layers[0][3]
    H, W = ops.stem_out_size(input_size[0], input_size[1], ks)
    C = layers[0][2]
    plan = []
    fused_stem = layers[2][0] == 'pool'
    first = 2
    bridged = False                                     # the next Fire's squeeze already ran inside the previous launch
    if fused_stem:
        nxt = layers[3] if len(layers) > 3 else None
        if (fuse_train_forward and fuse_stem_squeeze and nxt is not None and nxt[0] == 'fire'
                and ops.stem_pool_squeeze_ok((batch, 3, input_size[0], input_size[1]), (layers[0][2], 3, ks, ks), nxt[2])):
            plan.append((f'stem_pool_sq_train<{ks}>', f'stem+pool+squeeze {input_size[0]}x{input_size[1]} S{nxt[2]}'))
            bridged = True
        else:
            plan.append((f'stem_pool<{ks}>', f'stem+pool {input_size[0]}x{input_size[1]}'))
        H, W = ops.pool_out_size(H, W)
        first = 3
    else:
        plan.append((f'stem_conv<{ks}>', f'stem {input_size[0]}x{input_size[1]}'))
    geo = {}                                            # layer index -> (H, W, C_in) at its input
    for i in range(first, len(layers)):
        l = layers[i]
        geo[i] = (H, W, C)
        if l[0] == 'pool':
            if not bridged:
                plan.append(('maxpool_fwd', f'pool C{C} {H}x{W}'))
            H, W = ops.pool_out_size(H, W)
            continue
        _, cin, s, e1, e3 = l
        # (the last Fire carries the dropout in its expand epilogues: two plain launches)
        is_last = i == len(layers) - 1
        nxt = layers[i + 1] if i + 1 < len(layers) else None
        nxt2 = layers[i + 2] if i + 2 < len(layers) else None
        npix = batch * H * W
        zseg_t = ycfg_t = None
        if fuse_train_forward and fuse_fire_bridge and use_winograd:
            if nxt is not None and nxt[0] == 'pool' and nxt2 is not None and nxt2[0] == 'fire':
                zseg_t = ops.choose_fire_pool_bridge(s, e1, e3, nxt2[2], npix)
            if nxt is not None and nxt[0] == 'fire' and not (is_last and dropout):
                ycfg_t = ops.choose_fire_bridge_cfg(s, e1, e3, nxt[2], npix)
                if ycfg_t is not None and ycfg_t % 1000 != 12:
                    ycfg_t = None
        if zseg_t is not None or ycfg_t is not None:
            if not bridged:
                plan.append(_conv1x1(batch, H, W, cin, s))
            if zseg_t is not None:
                plan.append(('fire_pool_bridge_save', f'fire C{s} E{e1}+{e3} -> pool -> S{nxt2[2]} {H}x{W}'))
            else:
                plan.append(('fire_bridge_save', f'fire C{s} E{e1}+{e3} -> S{nxt[2]} {H}x{W}'))
            bridged = True
            C = e1 + e3
            continue
        dcfg = None
        if is_last and dropout:
            dcfg = ops.conv_drop_cfg(s, e1, batch * H * W) if (fused_dropout and s % 8 == 0 and use_winograd) else None
            if dcfg is None:
                plan.append(('dropout_mask', f'{batch * H * W * (e1 + e3)} elements'))
        if not bridged:
            plan.append(_conv1x1(batch, H, W, cin, s))
        bridged = False
        if dcfg is not None:
            plan.append((ops.cfg_kernel_name(dcfg), f'1tap C{s} N{e1} {H}x{W}'))
            plan.append(('conv_wino_sk', f'9tap C{s} N{e3} {H}x{W}'))
        else:
            plan.append(_conv1x1(batch, H, W, s, e1))
            plan.append(_conv3x3(batch, H, W, s, e3, use_winograd))
        C = e1 + e3
    ncd = anchors_per_grid * (num_classes + 5)
    ccd = convdet_in_channels(arch)
    fused_rng = dropout and layers[-1][0] == 'fire' and (fused_dropout and layers[-1][2] % 8 == 0 and use_winograd
                                                           and ops.conv_drop_cfg(layers[-1][2], layers[-1][3], batch * H * W) is not None)
    plan.append(_conv3x3(batch, H, W, ccd, ncd, use_winograd))
    A = H * W * anchors_per_grid
    plan.append(('loss_fwd', f'loss A{A}'))
    plan.append(('loss_bwd', f'lossbwd A{A}'))
    # ---- backward (backward.run_backbone_backward) ----
    plan.append(_wgrad(batch, H, W, ncd, ccd, 9))
    if fused_rng and ncd % 8 == 0:
        plan.append(('conv_wino_sk', f'9tap C{ncd} N{ccd} {H}x{W}'))                 # ConvDet data gradient (mask = its own input, constant scale)
    else:
        plan.append(_conv3x3(batch, H, W, ncd, ccd, use_winograd))                   # ConvDet data gradient
    rows_total, rows_done = 1, 0                       # slab-reduction records: ConvDet, then 3 per Fire in backward order
    last = len(layers) - 1
    # expand3x3 weight gradients that share a launch (``group_wgrad`` = ``SqueezeDetBase.group_wgrad``): issued when the last member is reached
    groups = ops.wino_wgrad_groups([(i, layers[i][4], layers[i][2], batch, geo[i][0], geo[i][1]) for i in range(last, 1, -1)
                                    if layers[i][0] == 'fire'], enabled=group_wgrad)
    fire_idx = [i for i in range(last, 1, -1) if layers[i][0] == 'fire']
    groups.update(ops.wgrad1x1_groups([(('e1', i), layers[i][3], layers[i][2], batch, geo[i][0], geo[i][1]) for i in fire_idx
                                       if not (fuse_squeeze_bwd and ops.squeeze_bwd_ok(layers[i][3], layers[i][2]))], enabled=group_wgrad))
    pending = {}
    for i in range(last, 1, -1):
        l = layers[i]
        Hi, Wi, Ci = geo[i] if i in geo else (None, None, None)
        if l[0] == 'pool':
            if data_parallel_stages and rows_total > rows_done:         # a stage of the backward is complete: its bucket goes out
                plan.append(('wgrad_reduce_batched', f'{rows_total - rows_done} layers'))
                rows_done = rows_total
            if i == 2 and fused_stem:
                continue
            plan.append(('maxpool_bwd', f'poolbwd C{Ci} {Hi}x{Wi}'))
            continue
        _, cin, s, e1, e3 = l
        fused_e1 = fuse_squeeze_bwd and ops.squeeze_bwd_ok(e1, s)
        if not fused_e1:
            if ('e1', i) in groups:
                gid, _S, _tc, members = groups[('e1', i)]
                pending.setdefault(gid, []).append(f'C{s} N{e1}')
                if len(pending[gid]) == len(members):
                    plan.append(('conv_wgrad_group<1>', f'wgrad 1tap {" + ".join(pending.pop(gid))} {Hi}x{Wi}'))
            else:
                plan.append(_wgrad(batch, Hi, Wi, e1, s, 1))
        if i in groups:
            gid, _S, _tc, members = groups[i]
            pending.setdefault(gid, []).append(f'C{s} N{e3}')
            if len(pending[gid]) == len(members):
                plan.append(('conv_wgrad_wino_group', f'wgrad 9tap {" + ".join(pending.pop(gid))} {Hi}x{Wi}'))
        else:
            plan.append(_wgrad(batch, Hi, Wi, e3, s, 9))
        if fused_e1:
            plan.append(('squeeze_bwd', f'sqbwd C{s} N{e1} {Hi}x{Wi}'))                # expand1x1 weight + data gradient, one launch
        else:
            plan.append(_conv1x1(batch, Hi, Wi, e1, s))                               # expand1x1 data gradient
        plan.append(_conv3x3(batch, Hi, Wi, e3, s, use_winograd))                     # expand3x3 data gradient (accumulates)
        if fuse_squeeze_bwd and ops.squeeze_bwd_ok(s, cin):
            plan.append(('squeeze_bwd', f'sqbwd C{cin} N{s} {Hi}x{Wi}'))              # squeeze weight + data gradient, one launch
        else:
            plan.append(_wgrad(batch, Hi, Wi, s, cin, 1))
            plan.append(_conv1x1(batch, Hi, Wi, s, cin))                              # squeeze data gradient
        rows_total += 3
    Hs, Ws = input_size
    if fused_stem:
        plan.append((f'stem_wgrad_pooled<{ks}>', f'stem wgrad (pooled) {Hs}x{Ws}'))
    else:
        plan.append((f'stem_wgrad<{ks}>', f'stem wgrad {Hs}x{Ws}'))
    if rows_total > rows_done:
        plan.append(('wgrad_reduce_batched', f'{rows_total - rows_done} layers'))
    return plan


def launches_per_kernel(plan):
    """{kernel name: launches per step}."""
    return dict(Counter(name for name, _ in plan))
