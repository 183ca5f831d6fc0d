"""ctypes binding of libresunet_hip.so (include/resunet_hip.h).  No torch types cross the ABI: tensors are
handed over as raw device pointers + extents, the stream as the hipStream_t integer of torch's current
stream.  The product path is HIP-only: if the library is missing or no GPU is usable, calls raise -- there
is no CPU fallback (the CPU restatement lives in oracle/ and is test infrastructure only)."""
from __future__ import annotations

import ctypes as C
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# RU_LIB_PATH: another build of the same library (A/B timing of kernel changes on one GPU box)
LIB_PATH = os.environ.get("RU_LIB_PATH") or os.path.join(_PKG, "lib", "libresunet_hip.so")

_lib = None
PRECISIONS = {"f32": 0, "bf16x3": 1}      # RU_PREC_F32 / RU_PREC_BF16X3
GRAD_PRECISIONS = {"bf16x3": 1, "bf16": 2}   # ru_unet_set_grad_precision: RU_PREC_BF16X3 / RU_PREC_BF16
FUSE_GN_BWD_STATS, FUSE_GN_BWD_APPLY, FUSE_SIDE_STREAM, FUSE_BATCH_WREDUCE, FUSE_TAIL_FINALIZE, FUSE_PW_DGRAD = 1, 2, 4, 8, 16, 32   # RU_FUSE_*

_vp, _f, _d, _i, _sz = C.c_void_p, C.c_float, C.c_double, C.c_int, C.c_size_t

# name -> (restype, argtypes); mirrors include/resunet_hip.h one to one
SIGNATURES = {
    "ru_last_error": (C.c_char_p, []),
    "ru_version": (_i, []),
    "ru_device_ok": (_i, []),
    "ru_conv3d_workspace_bytes": (_sz, [_i] * 7),
    "ru_conv3d_fwd": (_i, [_vp, _vp, _vp, _vp] + [_i] * 7 + [_vp, _sz, _vp]),
    "ru_conv3d_bwd_data": (_i, [_vp, _vp, _vp] + [_i] * 7 + [_vp, _sz, _vp]),
    "ru_conv3d_fwd_p": (_i, [_vp, _vp, _vp, _vp] + [_i] * 8 + [_vp, _sz, _vp]),
    "ru_conv3d_bwd_data_p": (_i, [_vp, _vp, _vp] + [_i] * 8 + [_vp, _sz, _vp]),
    "ru_conv3d_bwd_weight": (_i, [_vp, _vp, _vp, _vp] + [_i] * 7 + [_vp, _sz, _vp]),
    "ru_conv3d_bwd_weight_p": (_i, [_vp, _vp, _vp, _vp] + [_i] * 8 + [_vp, _sz, _vp]),
    "ru_groupnorm_workspace_bytes": (_sz, [_i, _i, _sz]),
    "ru_groupnorm_fwd": (_i, [_vp] * 7 + [_i, _i, _sz, _i, _f, _f, _vp, _sz, _vp]),
    "ru_groupnorm_bwd": (_i, [_vp] * 9 + [_i, _i, _sz, _i, _f, _vp, _sz, _vp]),
    "ru_leaky_relu_fwd": (_i, [_vp, _vp, _sz, _f, _vp]),
    "ru_leaky_relu_bwd": (_i, [_vp, _vp, _vp, _sz, _f, _vp]),
    "ru_upsample2x_trilinear_fwd": (_i, [_vp, _vp] + [_i] * 5 + [_vp]),
    "ru_upsample2x_trilinear_bwd": (_i, [_vp, _vp] + [_i] * 5 + [_vp]),
    "ru_sigmoid_fwd": (_i, [_vp, _vp, _sz, _vp]),
    "ru_criterion_workspace_bytes": (_sz, [_i, _i, _sz]),
    "ru_criterion_sums": (_i, [_vp, _vp, _vp, _i, _i, _sz, _f, _vp, _sz, _vp]),
    "ru_criterion_grad": (_i, [_vp, _vp, _vp, _d, _f, _f, _f, _f, _vp, _i, _i, _sz, _vp]),
    "ru_criterion_value": (_i, [C.POINTER(_d), _i, _d, _d, C.POINTER(_d), C.POINTER(_d)]),
    "ru_criterion_value_device": (_i, [_vp, _i, _d, _d, _d, _d, _vp, _vp]),
    "ru_adam_amsgrad_step": (_i, [_vp] * 5 + [_sz] + [_f] * 5 + [_i, _vp]),
    "ru_adam_step": (_i, [_vp] * 5 + [_sz] + [_f] * 5 + [_i, _vp]),
    "ru_unet_create": (_vp, [_i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _i]),
    "ru_unet_destroy": (None, [_vp]),
    "ru_unet_set_precision": (_i, [_vp, _i]),
    "ru_unet_freeze_params": (_i, [_vp, _i]),
    "ru_unet_get_precision": (_i, [_vp]),
    "ru_unet_set_grad_precision": (_i, [_vp, _i]),
    "ru_unet_get_grad_precision": (_i, [_vp]),
    "ru_unet_set_fusion": (_i, [_vp, C.c_uint]),
    "ru_unet_probe": (_i, [_vp, _i]),
    "ru_unet_probe_read": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(_i)]),
    "ru_unet_probe_read_families": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(_i), _i]),
    "ru_unet_param_count": (_i, [_vp]),
    "ru_unet_param_name": (C.c_char_p, [_vp, _i]),
    "ru_unet_param_ndim": (_i, [_vp, _i]),
    "ru_unet_param_dim": (_i, [_vp, _i, _i]),
    "ru_unet_param_offset": (_sz, [_vp, _i]),
    "ru_unet_param_total": (_sz, [_vp]),
    "ru_unet_param_is_dead": (_i, [_vp, _i]),
    "ru_unet_workspace_bytes": (_sz, [_vp] + [_i] * 5),
    "ru_unet_forward": (_i, [_vp, _vp, _vp, _vp] + [_i] * 5 + [_vp, _sz, _vp]),
    "ru_unet_backward": (_i, [_vp] * 6),
    "ru_unet_backward_criterion": (_i, [_vp, _vp, _vp, _vp, _d, _f, _f, _f, _f, _vp, _vp, _vp]),
    "ru_unet_gn_stats": (_i, [_vp, _i, _vp, _vp, _vp]),
    "ru_comm_unique_id": (_i, [_vp]),
    "ru_comm_init": (_i, [C.POINTER(_vp), _vp, _i, _i]),
    "ru_comm_destroy": (_i, [_vp]),
    "ru_comm_rank": (_i, [_vp]),
    "ru_comm_world": (_i, [_vp]),
    "ru_allreduce": (_i, [_vp, _vp, _sz, _i, _vp]),
    "ru_comm_group_begin": (_i, []),
    "ru_comm_group_end": (_i, []),
    "ru_tta_merge": (_i, [_vp, _i, C.c_uint, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ru_compose_labels": (_i, [_vp, _vp, C.c_ulonglong, _vp, _sz, _vp]),
    "ru_dice_counts": (_i, [_vp, _vp, _vp, _i, _i, _sz, _vp]),
    "ru_dice_accumulate": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ru_tile_gather": (_i, [_vp, _vp] + [_i] * 6 + [C.POINTER(_i), _i, _i, _i, _vp]),
    "ru_tile_scatter": (_i, [_vp, _vp] + [_i] * 6 + [C.POINTER(_i), _i, _i, _i, C.POINTER(_i), C.POINTER(_i), _vp]),
    "ru_case_bbox": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ru_case_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ru_case_stats": (_i, [_vp, _vp, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), _vp, _sz, _vp]),
    "ru_case_prepare": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _i, C.c_uint, _vp]),
    "ru_tta_merge_box": (_i, [_vp, _i, C.c_uint, _vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), _vp]),
    "ru_cc_workspace_bytes": (_sz, [_i, _i, _i]),
    "ru_cc_reject": (_i, [_vp, _i, _i, _i, _d, _vp, _sz, _vp]),
    "ru_paste_labels": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), _vp]),
    "ru_zscore_workspace_bytes": (_sz, [_i, _sz]),
    "ru_zscore_stats": (_i, [_vp, _vp, _i, _sz, _vp, _sz, _vp]),
    "ru_augment_patch": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "ru_layout_convert": (_i, [_vp, _vp, _i, _i, _sz, _i, _vp]),
    "ru_upsample2x_trilinear_fwd_l": (_i, [_vp, _vp] + [_i] * 5 + [_f, _vp]),
    "ru_upsample2x_trilinear_bwd_l": (_i, [_vp, _vp] + [_i] * 5 + [_vp]),
    "ru_conv3d_fwd_l": (_i, [_vp, _vp, _vp, _vp] + [_i] * 7 + [_vp, _sz, _vp]),
    "ru_conv3d_bwd_weight_l": (_i, [_vp, _vp, _vp] + [_i] * 7 + [_vp, _sz, _vp]),
}


def load():
    """Load the shared library (once).  Raises RuntimeError with the build hint if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libresunet_hip.so not found at %s -- build it with `python -m brats2019_amd.build` "
            "(hipcc, gfx950).  There is no CPU fallback for this path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("RU_LIB_PATH") and not hasattr(lib, name):
            continue                 # A/B timing against an older build: entry points added since are simply absent
        fn = getattr(lib, name)      # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().ru_last_error().decode("utf-8", "replace")


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("libresunet_hip: %s failed (%d): %s" % (what, rc, last_error()))


def require_gpu():
    """The hot path runs on an MI355X or not at all."""
    if not torch.cuda.is_available():
        raise RuntimeError("brats2019_amd: no ROCm GPU visible -- the ResUNet hot path is HIP-only (no CPU fallback)")
    load()


def ptr(t, allow_none=False):
    """Device pointer of a contiguous float32 (or float64 where noted) CUDA tensor."""
    if t is None:
        if allow_none:
            return None
        raise ValueError("null tensor")
    if not t.is_cuda:
        raise RuntimeError("brats2019_amd: expected a ROCm device tensor, got a %s tensor (HIP-only path)" % t.device)
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous (NCDHW)")
    return C.c_void_p(t.data_ptr())


def f32(t):
    if t.dtype != torch.float32:
        raise TypeError("expected float32, got %s" % t.dtype)
    return ptr(t)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
