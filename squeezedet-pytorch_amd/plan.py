"""The launch plan of one inference step, computed on the host WITHOUT a GPU: which kernel instance (tile
configuration from the measured table ``tuning.json``) every layer of ``SqueezeDetBase.forward`` + the fused detect
launches for a given architecture / batch / input size.  It mirrors the decisions of ``autograd.run_backbone_forward``
(a GPU test asserts the two agree launch for launch) and exists so that profiles can be checked against the code that
shipped: ``profiles/traffic.json`` records the launch set it was measured on, and a CPU test recomputes that set here.

Reference for the layer sequence: src/model/squeezedet.py:33-87, src/engine/detector.py:20-50.
"""
from __future__ import annotations

from collections import Counter

from . import ops
from .synthetic import convdet_in_channels, layer_table


def inference_launch_plan(arch='squeezedet', batch=20, input_size=(384, 1248), anchors_per_grid=9, num_classes=3,
                          use_winograd=True, fuse_expand=True, fuse_fire_bridge=True):
    """-> list of (kernel name as bench.py / KernelTimer prints it, shape tag), in launch order."""
    layers = layer_table(arch)
    H, W = ops.stem_out_size(input_size[0], input_size[1], layers[0][3])
    C = layers[0][2]
    plan = []
    first = 2
    if layers[2][0] == 'pool':
        plan.append((f'stem_pool<{layers[0][3]}>', f'stem+pool {input_size[0]}x{input_size[1]}'))
        H, W = ops.pool_out_size(H, W)
        first = 3
    else:
        plan.append((f'stem_conv<{layers[0][3]}>', f'stem {input_size[0]}x{input_size[1]}'))

    def conv3x3(Cin, N):
        npix = batch * H * W
        wc = ops.choose_wino_cfg(Cin, N, npix) if use_winograd else None
        if wc is not None:
            return (ops.wino_kernel_name(wc), f'9tap C{Cin} N{N} {H}x{W}')
        return (ops.cfg_kernel_name(ops.choose_cfg(9, Cin, N, npix)), f'9tap C{Cin} N{N} {H}x{W}')

    bridged = False
    for i in range(first, len(layers)):
        l = layers[i]
        if l[0] == 'pool':
            if not bridged:
                plan.append(('maxpool_fwd', f'pool C{C} {H}x{W}'))
            H, W = ops.pool_out_size(H, W)
            continue
        _, cin, s, e1, e3 = l
        npix = batch * H * W
        if not bridged:
            plan.append((ops.cfg_kernel_name(ops.choose_cfg(1, cin, s, npix)), f'1tap C{cin} N{s} {H}x{W}'))
        bridged = False
        C = e1 + e3
        nxt = layers[i + 1] if i + 1 < len(layers) else None
        nxt2 = layers[i + 2] if i + 2 < len(layers) else None
        if nxt is not None and nxt[0] == 'pool' and nxt2 is not None and nxt2[0] == 'fire' and fuse_fire_bridge and use_winograd:
            if ops.choose_fire_pool_bridge(s, e1, e3, nxt2[2], npix) is not None:
                plan.append(('fire_pool_bridge', f'fire C{s} E{e1}+{e3} -> pool -> S{nxt2[2]} {H}x{W}'))
                bridged = True
                continue
        if nxt is not None and nxt[0] == 'fire' and fuse_fire_bridge and use_winograd:
            if ops.choose_fire_bridge_cfg(s, e1, e3, nxt[2], npix) is not None:
                plan.append(('fire_bridge', f'fire C{s} E{e1}+{e3} -> S{nxt[2]} {H}x{W}'))
                bridged = True
                continue
        xcfg = ops.choose_fire_wino_cfg(s, e1, e3, npix) if (fuse_expand and use_winograd) else None
        fcfg = ops.choose_fused_cfg(s, e1, npix) if (xcfg is None and fuse_expand and e1 == e3) else None
        if xcfg is not None:
            plan.append((ops.fire_wino_kernel_name(xcfg), f'fire C{s} E{e1}+{e3} {H}x{W}'))
        elif fcfg is not None:
            plan.append((ops.cfg_kernel_name(fcfg).replace('conv_dma', 'fire_expand'), f'expand C{s} E{e1} {H}x{W}'))
        else:
            plan.append((ops.cfg_kernel_name(ops.choose_cfg(1, s, e1, npix)), f'1tap C{s} N{e1} {H}x{W}'))
            plan.append(conv3x3(s, e3))
    plan.append(conv3x3(convdet_in_channels(arch), anchors_per_grid * (num_classes + 5)))
    plan.append(('detect', f'detect A{H * W * anchors_per_grid}'))
    return plan


def launches_per_kernel(plan):
    """{kernel name: launches per step}."""
    return dict(Counter(name for name, _ in plan))
