"""ctypes binding of the C-ABI HIP library (``csrc/libsqdhip.so``, declared in include/sqd_hip.h).

The library takes raw device pointers, explicit sizes and a ``hipStream_t``; PyTorch only
provides device memory and the current stream.  There is NO fallback: if the shared object is
missing or a symbol cannot be resolved, importing/using the ops raises immediately.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

import torch  # noqa: F401  (loads the process-wide HIP runtime, libamdhip64.so.7, before our library)

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
_LIB_PATH = os.environ.get('SQD_HIP_LIBRARY') or os.path.join(_CSRC, 'libsqdhip.so')     # (override: A/B builds of the same ABI)
_lock = threading.Lock()
_lib = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float

# name -> argtypes (restype is always int status); must match include/sqd_hip.h
_SIGNATURES = {
    'sqd_conv_num_cfgs': [],
    'sqd_conv_cfg_info': [c_i, ctypes.POINTER(c_i), ctypes.POINTER(c_i), ctypes.POINTER(c_i), ctypes.POINTER(c_i)],
    'sqd_conv_cfg_is_dma': [c_i],
    'sqd_conv_fwd': [c_p] * 7 + [c_i] * 19 + [c_p],
    'sqd_pack_conv_weight': [c_p, c_p] + [c_i] * 6 + [c_p],
    'sqd_pack_conv_weights_batched': [c_p, c_i, c_i, c_p],
    'sqd_conv_wgrad': [c_p] * 5 + [c_i] * 11 + [c_p],
    'sqd_wgrad_reduce_batched': [c_p, c_i, c_i, c_p, c_p, c_f, c_p],
    'sqd_wgrad_reduce_batched_range': [c_p, c_i, c_i, c_i, c_p, c_p, c_f, c_p],
    'sqd_grad_scale': [c_p, ctypes.c_longlong, c_f, c_p, c_p, c_f, c_p],
    'sqd_conv_wgrad_wino': [c_p] * 5 + [c_i] * 11 + [c_p],
    'sqd_conv_wgrad_wino_group': [c_p] + [c_i] * 6 + [c_p],
    'sqd_conv_wgrad_group': [c_p] + [c_i] * 5 + [c_p],
    'sqd_squeeze_bwd': [c_p] * 5 + [c_i] * 13 + [c_p],
    'sqd_stem_wgrad': [c_p] * 5 + [c_i] * 6 + [c_p],
    'sqd_stem_wgrad_pooled': [c_p] * 7 + [c_i] * 6 + [c_p],
    'sqd_stem_conv_relu_fwd': [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    'sqd_stem_conv_fwd': [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    'sqd_stem_conv_relu_pool_fwd': [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    'sqd_stem_pool_squeeze_fwd': [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    'sqd_stem_pool_squeeze_train_fwd': [c_p] * 8 + [c_i] * 6 + [c_p],
    'sqd_maxpool3x3s2_ceil_fwd': [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    'sqd_maxpool3x3s2_ceil_fwd_relu': [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    'sqd_maxpool3x3s2_ceil_bwd': [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    'sqd_resolve_fwd': [c_p] * 7 + [c_i] * 5 + [c_p],
    'sqd_decode_fwd': [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    'sqd_detect_fwd': [c_p] * 9 + [c_i] * 6 + [c_f, c_f, c_p],
    'sqd_detect_shift_fwd': [c_p] * 10 + [c_i] * 6 + [c_f, c_f, c_p],
    'sqd_filter_fwd': [c_p] * 9 + [c_i] * 4 + [c_f, c_f, c_p],
    'sqd_preprocess_u8_fwd': [c_p] * 5 + [ctypes.POINTER(c_f), ctypes.POINTER(c_f), c_i, c_i, c_i, c_p],
    'sqd_preprocess_u8_padcrop_fwd': [c_p] * 6 + [ctypes.POINTER(c_f), ctypes.POINTER(c_f), c_i, c_i, c_i, c_p],
    'sqd_pool_squeeze_fwd': [c_p] * 4 + [c_i] * 10 + [c_p],
    'sqd_fire_expand_fwd': [c_p] * 4 + [c_i] * 11 + [c_p],
    'sqd_kitti_ap': [c_i] + [c_p] * 12,
    'sqd_wino_num_cfgs': [],
    'sqd_wino_cfg_info': [c_i, c_p, c_p],
    'sqd_conv_wino_fwd': [c_p] * 6 + [c_i] * 13 + [c_p],
    'sqd_conv_wino_vs_fwd': [c_p] * 4 + [c_i] * 11 + [c_p],
    'sqd_wino_sk_grid': [],
    'sqd_wino_sk_schedule': [c_i] * 7 + [c_p, c_p, c_i, c_p, c_p],
    'sqd_conv_wino_sk_fwd': [c_p] * 6 + [c_f] + [c_i] * 12 + [c_p, c_p, c_i, c_i, c_p, c_p] + [c_p, c_i, c_f, c_p] + [c_p],
    'sqd_conv_drop_fwd': [c_p] * 4 + [c_i] * 11 + [c_p, c_i, c_f, c_i, c_p],
    'sqd_dropout_mask_fwd': [c_p, c_i, c_f, c_p, ctypes.c_longlong, c_p],
    'sqd_dropout_advance': [c_p, c_p],
    'sqd_spin_us': [c_i, c_p],
    'sqd_pack_wino_weight': [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    'sqd_pack_wino_weights_batched': [c_p, c_i, c_i, c_p],
    'sqd_pack_wino_fire': [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    'sqd_fire_wino_fwd': [c_p] * 5 + [c_i] * 13 + [c_p],
    'sqd_fire_bridge_fwd': [c_p] * 6 + [c_i] * 13 + [c_p],
    'sqd_fire_bridge_save_fwd': [c_p] * 7 + [c_i] * 15 + [c_p],
    'sqd_gather_pack_batched': [c_p, c_i, c_i, c_p],
    'sqd_sgd_clip_step': [c_p, c_i, c_p, c_p, c_f, c_f, c_f, c_f, c_i, c_p],
    'sqd_grad_sumsq': [c_p, ctypes.c_longlong, c_p, c_p],
    'sqd_grad_sumsq_parts': [],
    'sqd_sgd_chunk_elems': [],
    'sqd_sgd_clip_step_chunked': [c_p, c_p, c_i, c_p, c_p, c_p, c_f, c_f, c_f, c_f, c_p],
    'sqd_sgd_clip_step_parts': [c_p, c_i, c_p, c_p, c_p, c_f, c_f, c_f, c_f, c_i, c_p],
    'sqd_fire_pool_bridge_fwd': [c_p] * 6 + [c_i] * 15 + [c_p],
    'sqd_fire_pool_bridge_save_fwd': [c_p] * 8 + [c_i] * 18 + [c_p],
    'sqd_encode_gt_fwd': [c_p] * 8 + [c_i, c_i, c_i, c_i, c_p],
    'sqd_loss_fwd': [c_p] * 6 + [c_i] * 5 + [c_f] * 4 + [c_p],
    'sqd_loss_bwd': [c_p] * 6 + [c_i] * 5 + [c_f] * 4 + [c_p],
    'sqd_loss_mean_fwd': [c_p] * 7 + [c_i] * 5 + [c_f] * 4 + [c_p],
    'sqd_loss_mean_bwd': [c_p] * 6 + [c_i] * 5 + [c_f] * 4 + [c_p],
}
# symbols added by later build stages; bound when present in the library
_OPTIONAL = {}

_ERRORS = {1: 'bad argument', 2: 'unsupported configuration', 3: 'kernel launch failed'}


class NativeLibraryError(RuntimeError):
    pass


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into csrc/libsqdhip.so (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(['make', '-C', _CSRC, '-j8'], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise NativeLibraryError('building libsqdhip.so failed')
    return _LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise NativeLibraryError(
                    f'{_LIB_PATH} not found: the HIP extension is required (no CPU fallback). '
                    'Run `python -c "import __graft_entry__ as g; g.build()"` or `make -C squeezedet-pytorch_amd/csrc`.')
            l = ctypes.CDLL(_LIB_PATH)
            for name, argtypes in _SIGNATURES.items():
                fn = getattr(l, name)          # AttributeError -> loud failure on a stale library
                fn.argtypes = argtypes
                fn.restype = c_i
            for name, argtypes in _OPTIONAL.items():
                if hasattr(l, name):
                    fn = getattr(l, name)
                    fn.argtypes = argtypes
                    fn.restype = c_i
            _lib = l
    return _lib


def exported_symbols():
    return list(_SIGNATURES) + [n for n in _OPTIONAL if hasattr(lib(), n)]


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f'{what}: {_ERRORS.get(rc, "error")} (status {rc})')


def ptr(t):
    return c_p(t.data_ptr()) if t is not None else c_p(0)


def stream_handle(device=None):
    return c_p(torch.cuda.current_stream(device).cuda_stream)


def conv_cfgs():
    """[(cfg_id, taps, kc, tile_px, bn, is_dma)] as compiled into the library."""
    l = lib()
    out = []
    for i in range(l.sqd_conv_num_cfgs()):
        t, k, px, bn = c_i(), c_i(), c_i(), c_i()
        check(l.sqd_conv_cfg_info(i, ctypes.byref(t), ctypes.byref(k), ctypes.byref(px), ctypes.byref(bn)), 'sqd_conv_cfg_info')
        out.append((i, t.value, k.value, px.value, bn.value, l.sqd_conv_cfg_is_dma(i)))
    return out
