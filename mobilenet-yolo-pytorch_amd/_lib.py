"""ctypes binding of libmnyolo.so (include/mnyolo.h).  The product has NO fallback: if the
library is missing or a call fails, an exception is raised."""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MNY_LIB") or os.path.join(HERE, "libmnyolo.so")     # MNY_LIB: A/B another build on the same GPU box

ACT_NONE, ACT_RELU6, ACT_LEAKY, ACT_RELU, ACT_HSWISH, ACT_HSIGMOID = 0, 1, 2, 3, 4, 5
ROUTE_TILE_V1, ROUTE_DMA_F32, ROUTE_DMA_X6, ROUTE_THIN, ROUTE_WIDE, ROUTE_WGRAD_STREAM, ROUTE_WAVE16 = 0, 1, 2, 3, 4, 5, 6      # mny_pw_route


class MnyError(RuntimeError):
    pass


class YoloHead(ctypes.Structure):
    _fields_ = [("N", c_int), ("g", c_int), ("A", c_int), ("C", c_int), ("n_anchors_all", c_int),
                ("ignore_thresh", c_float), ("iou_thresh", c_float), ("iou_weighting", c_float)]


P = c_void_p
_SIGS = {
    # name: (restype, argtypes)
    "mny_version": (c_int, []),
    "mny_last_error": (c_char_p, []),
    "mny_max_parts": (c_int, []),
    "mny_stem_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "mny_stem_stat_parts": (c_int, [c_int, c_int, c_int, c_int]),
    "mny_stem_wgrad": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "mny_stem_bnwgrad_supported": (c_int, [c_int]),
    "mny_stem_bnwgrad": (c_int, [P, P, P, P, P, c_int, P, P, P, c_int, c_int, c_int, c_int, P]),
    "mny_stem_wgrad_parts": (c_int, [c_int, c_int, c_int, c_int]),
    "mny_dw_fwd": (c_int, [P, P, P, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "mny_dw_stat_parts": (c_int, [c_int] * 6),
    "mny_dw_stat_parts_x": (c_int, [c_int] * 7),
    "mny_dw_bwd_data": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "mny_dw_bwd_weight": (c_int, [P, P, P, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "mny_dw_wgrad_parts": (c_int, [c_int] * 6),
    "mny_dw_bnbwd_supported": (c_int, [c_int, c_int]),
    "mny_dw_bnbwd_parts": (c_int, [c_int] * 4),
    "mny_dw_bnbwd_parts_k": (c_int, [c_int] * 6),
    "mny_dw_bnbwd": (c_int, [P, P, P, P, c_int, P, P, P, P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "mny_dw_bnbwd_s2_parts": (c_int, [c_int] * 4),
    "mny_dw_bnbwd_s2": (c_int, [P, P, P, P, c_int, P, P, P, P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "mny_dw_bnbwd_red": (c_int, [P, P, P, P, c_int, P, P, P, P, c_int, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "mny_exdw_supported": (c_int, [c_int] * 6),
    "mny_exdw_stat_parts": (c_int, [c_int64, c_int, c_int]),
    "mny_exdw_stats": (c_int, [P, P, P, c_int, P, P, c_int64, c_int, c_int, P]),
    "mny_exdw_fwd_parts": (c_int, [c_int] * 6),
    "mny_exdw_fwd": (c_int, [P, P, P, c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "mny_exdw_bwd_parts": (c_int, [c_int] * 6),
    "mny_exdw_bwd_ws_floats": (c_size_t, [c_int] * 6),
    "mny_exdw_bwd": (c_int, [P, P, P, P, c_int, P, P, P, P, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P,
                             c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "mny_exdw_bwd_red_parts": (c_int, [c_int] * 6),
    "mny_exdw_bwd_red": (c_int, [P, P, P, P, c_int, P, P, P, P, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P,
                                 c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "mny_pj_bwd_supported": (c_int, [c_int64, c_int, c_int, c_int]),
    "mny_pj_bwd_parts": (c_int, [c_int64, c_int, c_int]),
    "mny_pj_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, P, P, P, P, P, c_int64, c_int, c_int, P]),
    "mny_stemdw_supported": (c_int, [c_int] * 6),
    "mny_stemdw_bwd_parts": (c_int, [c_int] * 4),
    "mny_stemdw_bwd_ws_floats": (c_size_t, [c_int] * 4),
    "mny_stemdw_bwd": (c_int, [P, P, P, P, c_int, P, P, P, P, P, P, P, c_int, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "mny_adamw_step": (c_int, [P, c_int, c_double, c_double, c_double, c_double, c_double, c_int64, P]),
    "mny_pw_fwd": (c_int, [P, P, P, c_int, P, P, P, P, P, c_int64, c_int, c_int, P]),
    "mny_pw_stat_parts": (c_int, [c_int64, c_int, c_int]),
    "mny_pw_wgrad": (c_int, [P, P, P, c_int, P, P, P, P, c_int64, c_int, c_int, P]),
    "mny_pw_wgrad_ws_floats": (c_size_t, [c_int64, c_int, c_int]),
    "mny_pw_bnbwd_supported": (c_int, [c_int64, c_int, c_int]),
    "mny_pw_bnbwd_ws_floats": (c_size_t, [c_int64, c_int, c_int]),
    "mny_pw_bnbwd": (c_int, [P, P, P, P, c_int, P, P, P, P, P, P, c_int, P, P, P, P, P, P, P, c_int64, c_int, c_int, P]),
    "mny_pw_dgrad_bnred_supported": (c_int, [c_int64, c_int, c_int, c_int]),
    "mny_pw_dgrad_bnred_parts": (c_int, [c_int64, c_int, c_int]),
    "mny_pw_dgrad_bnred": (c_int, [P, P, P, P, P, P, c_int, P, P, P, c_int64, c_int, c_int, P]),
    "mny_pw_dgrad_bnred_add_supported": (c_int, [c_int64, c_int, c_int, c_int]),
    "mny_pw_dgrad_bnred_add": (c_int, [P, P, P, P, P, P, P, c_int, P, P, P, c_int64, c_int, c_int, P]),
    "mny_pw_route": (c_int, [c_int, c_int, c_int64, c_int, c_int]),
    "mny_pw_last_route": (c_int, []),
    "mny_pw_w6_supported": (c_int, [c_int64, c_int, c_int]),
    "mny_pw_w6_bytes": (c_size_t, [c_int, c_int]),
    "mny_cut3_batch": (c_int, [P, P, c_int, P]),
    "mny_pw_fwd_w6": (c_int, [P, P, P, c_int, P, P, P, P, P, c_int64, c_int, c_int, P]),
    "mny_pw_dgrad_bnred_w6": (c_int, [P, P, P, P, P, P, P, c_int, P, P, P, c_int64, c_int, c_int, P]),
    "mny_transpose_batch": (c_int, [P, P, c_int, P]),
    "mny_reduce_batch": (c_int, [P, P, c_int, P]),
    "mny_pw_wgrad_splits": (c_int, [c_int64, c_int, c_int]),
    "mny_transpose": (c_int, [P, P, c_int, c_int, P]),
    "mny_transpose_pad": (c_int, [P, P, c_int, c_int, c_int, P]),
    "mny_pad_rows": (c_int, [P, P, P, c_int64, c_int, c_int, P]),
    "mny_bn_finalize": (c_int, [P, c_int, c_int64, P, P, c_float, c_float, P, P, P, P, P, P, c_int, P]),
    "mny_bn_eval_coeffs": (c_int, [P, P, P, P, c_float, P, P, c_int, P]),
    "mny_bn_bwd_reduce": (c_int, [P, P, P, P, c_int, P, P, P, c_int64, c_int, P]),
    "mny_bn_bwd_parts": (c_int, [c_int64, c_int]),
    "mny_bn_bwd_finalize": (c_int, [P, c_int, c_int64, P, P, P, P, P, P, c_int, P]),
    "mny_bn_eval_stats": (c_int, [P, P, c_float, P, P, c_int, P]),
    "mny_bn_bwd_finalize_frozen": (c_int, [P, c_int, c_int64, P, P, P, P, P, P, c_int, P]),
    "mny_bn_bwd_apply": (c_int, [P, P, P, P, c_int, P, P, c_int64, c_int, P]),
    "mny_add_views": (c_int, [P, P, P, c_int, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, P]),
    "mny_mul_views": (c_int, [P, P, P, c_int, P, P, P, c_int, P, c_int64, c_int, P]),
    "mny_mul_views_bwd": (c_int, [P, P, P, P, c_int, P, P, c_int64, c_int, P]),
    "mny_partadd_up": (c_int, [P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "mny_slice_channels": (c_int, [P, P, c_int, c_int64, c_int, c_int, P]),
    "mny_upsample_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "mny_axpy": (c_int, [P, P, P, c_int, c_int64, P]),
    "mny_yolo_loss": (c_int, [P, P, P, P, P, ctypes.POINTER(YoloHead), P, P, P, P]),
    "mny_yolo_loss_ws_bytes": (c_size_t, [ctypes.POINTER(YoloHead), c_int]),
    "mny_yolo_decode": (c_int, [P, P, P, ctypes.POINTER(YoloHead), c_float, P, c_int, P, P, P]),
    "mny_nms_per_class": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_double, P, P, P, P, P]),
    "mny_nms_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mny_nms_status_offset": (c_size_t, [c_int, c_int, c_int]),
    "mny_nms_prefix_offset": (c_size_t, [c_int, c_int, c_int]),
    "mny_map_ws_bytes": (c_size_t, [c_int64, c_int64]),
    "mny_map_eval": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int64, c_int64, c_int, P, P, P, P, P, P, P]),
    "mny_seg_loss_ws_bytes": (c_size_t, [c_int64]),
    "mny_seg_loss": (c_int, [P, P, c_int64, P, P, P, P]),
    "mny_seg_sigmoid": (c_int, [P, c_int, c_int, c_int, P, P]),
    "mny_prep_ws_bytes": (c_size_t, [c_int] * 5),
    "mny_prep_batch": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P]),
    "mny_eval_pack": (c_int, [P, c_int64, P, c_int64, P, P, P, P, P, P, P]),
}
# bf16-storage twins (activation tensors bf16, everything else as in the fp32 entry point): identical ctypes signature
BF16_TWINS = ("mny_pj_bwd_supported", "mny_pj_bwd_parts", "mny_pj_bwd", "mny_pw_bnbwd_supported", "mny_pw_bnbwd", "mny_pw_wgrad_splits", "mny_transpose_batch", "mny_pw_dgrad_bnred_supported", "mny_pw_dgrad_bnred_parts", "mny_pw_dgrad_bnred", "mny_pw_dgrad_bnred_add", "mny_pw_dgrad_bnred_add_supported", "mny_stem_bnwgrad", "mny_pad_rows", "mny_transpose_pad", "mny_dw_bnbwd", "mny_dw_bnbwd_s2", "mny_dw_bnbwd_red", "mny_stem_fwd", "mny_stem_wgrad", "mny_dw_fwd", "mny_dw_bwd_data", "mny_dw_bwd_weight", "mny_pw_fwd",
              "mny_pw_stat_parts", "mny_pw_wgrad", "mny_bn_bwd_reduce", "mny_bn_bwd_apply", "mny_add_views", "mny_mul_views",
              "mny_mul_views_bwd", "mny_partadd_up", "mny_slice_channels", "mny_upsample_bwd", "mny_axpy")
for _n in BF16_TWINS:
    _SIGS[_n + "_bf16"] = _SIGS[_n]
_SIGS["mny_gate_supported"] = (c_int, [c_int64, c_int, c_int])
_SIGS["mny_gate_parts"] = (c_int, [c_int64])
_SIGS["mny_gate_bwd_parts"] = (c_int, [c_int64])
_SIGS["mny_gate_bwd_red3_supported"] = (c_int, [c_int, c_int])
_SIGS["mny_gate_wq_bytes"] = (c_size_t, [c_int, c_int])
_SIGS["mny_gate_cut_batch_bf16"] = (c_int, [P, c_int, P])
_SIGS["mny_gate_stats1_bf16"] = (c_int, [P, P, P, P, P, c_int64, c_int, c_int, P])
_SIGS["mny_gate_stats2_bf16"] = (c_int, [P, P, P, P, P, P, P, c_int64, c_int, c_int, P])
_SIGS["mny_gate_fwd_bf16"] = (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_int, P, c_int64, c_int, c_int, P])
_SIGS["mny_gate_bwd1_bf16"] = (c_int, [P] * 12 + [c_int64, c_int, c_int, P])
_SIGS["mny_gate_bwd2_bf16"] = (c_int, [P] * 14 + [c_int64, c_int, c_int, P])
_SIGS["mny_gate_bwd3_bf16"] = (c_int, [P] * 16 + [c_int64, c_int, c_int, P])
# low-rank BatchNorm backward of the wide expand units (csrc/lrbwd.hip)
_SIGS["mny_lr_supported"] = (c_int, [c_int64, c_int, c_int])
_SIGS["mny_lr_gram_parts"] = (c_int, [c_int64, c_int])
_SIGS["mny_lr_gram"] = (c_int, [P, P, P, c_int, P, c_int64, c_int, P])
_SIGS["mny_lr_gram_bf16"] = _SIGS["mny_lr_gram"]
_SIGS["mny_lr_prep"] = (c_int, [P, P, P, P, c_int, c_int, P])
_SIGS["mny_pw_lr_fix_parts"] = (c_int, [c_int64, c_int, c_int])
_SIGS["mny_pw_lr_fix"] = (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, P, P, P, c_int64, c_int, P])
_SIGS["mny_lr_wfix"] = (c_int, [P, P, P, P, c_int, c_int, P])
_SIGS["mny_pw_bnbwd_red_supported"] = (c_int, [c_int64, c_int, c_int])
_SIGS["mny_pw_bnbwd_red_parts"] = (c_int, [c_int64, c_int, c_int])
_SIGS["mny_pw_bnbwd_red"] = (c_int, [P, P, P, P, c_int, P, P, P, P, P, P, c_int, P, P, P, P, P, P, P, P, P, P, c_int, P, P, P, c_int64, c_int, c_int, P])
_SIGS["mny_dw_bnbwd_red_dz_supported"] = (c_int, [c_int, c_int, c_int])
_SIGS["mny_dw_bnbwd_red_dz"] = _SIGS["mny_dw_bnbwd_red"]
_SIGS["mny_dw_bnbwd_s2_red_dz"] = (c_int, [P, P, P, P, c_int, P, P, P, P, c_int, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P])
_SIGS["mny_dw_bnbwd_s2k5_parts"] = (c_int, [c_int] * 4)
_SIGS["mny_dw_bnbwd_s2k5"] = _SIGS["mny_dw_bnbwd_s2_red_dz"]
_SIGS["mny_dw_bnbwd_s2k5_bf16"] = _SIGS["mny_dw_bnbwd_s2_red_dz"]
_SIGS["mny_transpose_bf16"] = _SIGS["mny_transpose"]
_SIGS["mny_cvt_f32_bf16"] = (c_int, [P, P, c_int64, P])
_SIGS["mny_cvt_batch_f32_bf16"] = (c_int, [P, P, c_int, P])
_SIGS["mny_cvt_bf16_f32"] = (c_int, [P, P, c_int64, P])
EXPORTS = tuple(_SIGS)

_lib = None


def load():
    """Load libmnyolo.so (raises MnyError when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MnyError("libmnyolo.so not found at %s — run __graft_entry__.build() (hipcc, gfx950). "
                           "There is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def call(name, *args):
    """Invoke an int-returning entry point; raise MnyError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise MnyError("%s failed (%d): %s" % (name, rc, lib.mny_last_error().decode()))


def query(name, *args):
    v = getattr(load(), name)(*args)
    if isinstance(v, int) and v < 0:
        raise MnyError("%s(%s) -> %d: %s" % (name, args, v, load().mny_last_error().decode()))
    return v
