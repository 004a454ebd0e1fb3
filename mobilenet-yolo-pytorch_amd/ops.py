"""Thin torch-tensor wrappers over the C ABI (include/mnyolo.h).

PyTorch is plumbing here: it owns device memory and the stream; every arithmetic op is a HIP kernel
of libmnyolo.so.  All tensors are contiguous, on a CUDA(=HIP) device; activations are NHWC, fp32 or — selecting the
`*_bf16` twins of the entry points — bf16; weights, BN coefficients, statistics and workspaces are always fp32.
A `view` is the tuple (tensor, scale|None, shift|None, act) — see mnyolo.h "View arguments".
"""
import ctypes

import torch

from . import _lib
from ._lib import ACT_NONE, YoloHead, call, query

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype in (torch.float32, torch.int32, torch.bfloat16) and t.is_contiguous(), \
        "libmnyolo needs contiguous fp32/bf16/int32 device tensors (got %s %s)" % (t.device, t.dtype)
    return ctypes.c_void_p(t.data_ptr())


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _new(*shape, like=None, dtype=torch.float32):
    return torch.empty(shape, device=like.device, dtype=dtype)


def _k(name, t):
    """entry point for the storage type of activation tensor `t`"""
    return name + "_bf16" if t.dtype == torch.bfloat16 else name


# ---- stem -------------------------------------------------------------------------------------------
def stem_fwd(x_nchw, w, want_stats=True, dtype=torch.float32):
    N, _, H, W = x_nchw.shape
    Co = w.shape[0]
    y = _new(N, (H + 1) // 2, (W + 1) // 2, Co, like=x_nchw, dtype=dtype)
    parts = query("mny_stem_stat_parts", N, H, W, Co)
    stats = _new(parts, 2, Co, like=x_nchw) if want_stats else None
    call(_k("mny_stem_fwd", y), _p(x_nchw), _p(w), _p(y), _p(stats), N, H, W, Co, _st())
    return y, stats


def stem_wgrad(x_nchw, dy):
    N, _, H, W = x_nchw.shape
    Co = dy.shape[3]
    parts = query("mny_stem_wgrad_parts", N, H, W, Co)
    ws = _new(parts, Co * 27, like=dy)
    dw = _new(Co, 3, 3, 3, like=dy)
    call(_k("mny_stem_wgrad", dy), _p(x_nchw), _p(dy), _p(dw), _p(ws), N, H, W, Co, _st())
    return dw


def stem_bnwgrad(x_nchw, g, y, scale, shift, act, gamma, mean, invstd):
    """Stem weight gradient from the unit's output gradient g (BN-backward-apply redone on load). -> (dw, dgamma, dbeta)"""
    N, _, H, W = x_nchw.shape
    Co = y.shape[3]
    M = y.numel() // Co
    parts = query("mny_bn_bwd_parts", M, Co)
    red = _new(parts, 2, Co, like=gamma)
    call(_k("mny_bn_bwd_reduce", y), _p(g), _p(y), _p(scale), _p(shift), act, _p(mean), _p(invstd), _p(red), M, Co, _st())
    dgamma, dbeta, coef = _new(Co, like=gamma), _new(Co, like=gamma), _new(3, Co, like=gamma)
    call("mny_bn_bwd_finalize", _p(red), parts, M, _p(gamma), _p(mean), _p(invstd), _p(dgamma), _p(dbeta), _p(coef), Co, _st())
    ws = _new(query("mny_stem_wgrad_parts", N, H, W, Co), Co * 27, like=gamma)
    dw = _new(Co, 3, 3, 3, like=gamma)
    call(_k("mny_stem_bnwgrad", y), _p(x_nchw), _p(g), _p(y), _p(scale), _p(shift), act, _p(coef), _p(dw), _p(ws), N, H, W, Co, _st())
    return dw, dgamma, dbeta


# ---- depthwise --------------------------------------------------------------------------------------
def dw_fwd(view, w, stride, want_stats=True):
    x, sc, sh, act = view
    N, H, W, C = x.shape
    K = w.shape[-1]
    Ho, Wo = (H + 2 * (K // 2) - K) // stride + 1, (W + 2 * (K // 2) - K) // stride + 1
    y = _new(N, Ho, Wo, C, like=x, dtype=x.dtype)
    stats = _new(query("mny_dw_stat_parts_x", N, H, W, C, K, stride, 1 if x.dtype == torch.bfloat16 else 0), 2, C, like=x) if want_stats else None
    call(_k("mny_dw_fwd", x), _p(x), _p(sc), _p(sh), act, _p(w), _p(y), _p(stats), N, H, W, C, K, stride, _st())
    return y, stats


def dw_bwd_data(dy, w, in_hw, stride, addend=None, out=None):
    N, _, _, C = dy.shape
    H, W = in_hw
    K = w.shape[-1]
    dx = out if out is not None else _new(N, H, W, C, like=dy, dtype=dy.dtype)
    call(_k("mny_dw_bwd_data", dy), _p(dy), _p(w), _p(addend), _p(dx), N, H, W, C, K, stride, _st())
    return dx


def dw_bwd_weight(view, dy, K, stride):
    x, sc, sh, act = view
    N, H, W, C = x.shape
    ws = _new(query("mny_dw_wgrad_parts", N, H, W, C, K, stride), C * K * K, like=x)
    dw = _new(C, 1, K, K, like=x)
    call(_k("mny_dw_bwd_weight", x), _p(x), _p(sc), _p(sh), act, _p(dy), _p(dw), _p(ws), N, H, W, C, K, stride, _st())
    return dw


def dw_bnbwd(g, y, scale, shift, act, coef, xview, w, addend=None, in_stats=None):
    """Fused BN-backward-apply + dw weight/data gradient of a 3x3 stride-1 unit -> (dx, dw).
    in_stats = (mean, invstd) of the unit that produced x: also returns that unit's BN-backward partial sums [parts,2,C]."""
    x, xs, xh, xact = xview
    N, H, W, C = x.shape
    k = w.shape[-1]                                     # 3, or 5 (tile form)
    parts = query("mny_dw_bnbwd_parts_k", N, H, W, C, k, (1 if x.dtype == torch.bfloat16 else 0) | (2 if in_stats is not None else 0))
    ws = _new(parts, C * k * k, like=x)
    dx = torch.empty_like(x)
    dw = _new(C, 1, k, k, like=x)
    if in_stats is None:
        call(_k("mny_dw_bnbwd", x), _p(g), _p(y), _p(scale), _p(shift), act, _p(coef), _p(x), _p(xs), _p(xh), xact, _p(w), _p(addend), _p(dx),
             _p(dw), _p(ws), N, H, W, C, k, 1, _st())
        return dx, dw
    red = _new(parts, 2, C, like=x)
    call(_k("mny_dw_bnbwd_red", x), _p(g), _p(y), _p(scale), _p(shift), act, _p(coef), _p(x), _p(xs), _p(xh), xact, _p(in_stats[0]), _p(in_stats[1]),
         _p(w), _p(addend), _p(dx), _p(dw), _p(ws), _p(red), N, H, W, C, k, 1, _st())
    return dx, dw, red


def dw_bnbwd_s2(g, y, scale, shift, act, coef, xview, w, addend=None):
    """Fused BN-backward-apply + weight / data gradient of a 3x3 STRIDE-2 depthwise unit -> (dx, dw)."""
    x, xs, xh, xact = xview
    N, H, W, C = x.shape
    ws = _new(query("mny_dw_bnbwd_s2_parts", N, H, W, C), C * 9, like=x)
    dx = torch.empty_like(x)
    dw = _new(C, 1, 3, 3, like=x)
    call(_k("mny_dw_bnbwd_s2", x), _p(g), _p(y), _p(scale), _p(shift), act, _p(coef), _p(x), _p(xs), _p(xh), xact, _p(w), _p(addend), _p(dx),
         _p(dw), _p(ws), N, H, W, C, _st())
    return dx, dw


def dw_bnbwd_s2k5(g, y, scale, shift, act, coef, xview, w, addend=None, in_stats=None):
    """Fused BN-backward-apply + weight / data gradient of a 5x5 STRIDE-2 depthwise unit -> (dx, dw) or, with in_stats = (mean, invstd) of the
    unit that produced x, (dx, dw, red[parts, 2, C]): that unit's BN-backward sums as partial rows."""
    x, xs, xh, xact = xview
    N, H, W, C = x.shape
    parts = query("mny_dw_bnbwd_s2k5_parts", N, H, W, C)
    ws = _new(parts, C * 25, like=w)
    dx = torch.empty_like(x)
    dw = _new(C, 1, 5, 5, like=w)
    red = _new(parts, 2, C, like=w) if in_stats is not None else None
    mu, istd = in_stats if in_stats is not None else (None, None)
    call(_k("mny_dw_bnbwd_s2k5", x), _p(g), _p(y), _p(scale), _p(shift), act, _p(coef), _p(x), _p(xs), _p(xh), xact, _p(mu), _p(istd), _p(w), _p(addend),
         _p(dx), _p(dw), _p(ws), _p(red), N, H, W, C, _st())
    return (dx, dw) if red is None else (dx, dw, red)


# ---- pointwise --------------------------------------------------------------------------------------
def pw_fwd(view, w2d, bias=None, addend=None, want_stats=True, out=None):
    """bf16 activations take bf16 weights (mny_pw_fwd_bf16)."""
    x, sc, sh, act = view
    assert w2d.dtype == x.dtype, "pw_fwd: weight dtype %s must match the activation storage type %s" % (w2d.dtype, x.dtype)
    K = x.shape[-1]
    M = x.numel() // K
    Nc = w2d.shape[0]
    y = out if out is not None else _new(*x.shape[:-1], Nc, like=x, dtype=x.dtype)
    stats = _new(query(_k("mny_pw_stat_parts", x), M, K, Nc), 2, Nc, like=x) if want_stats else None
    call(_k("mny_pw_fwd", x), _p(x), _p(sc), _p(sh), act, _p(w2d), _p(bias), _p(addend), _p(y), _p(stats), M, K, Nc, _st())
    return y, stats


def pw_wgrad(view, dy, want_dbias=False):
    x, sc, sh, act = view
    K = x.shape[-1]
    M = x.numel() // K
    Nc = dy.shape[-1]
    ws = _new(query("mny_pw_wgrad_ws_floats", M, K, Nc), like=x)
    dw = _new(Nc, K, like=x)
    db = _new(Nc, like=x) if want_dbias else None
    call(_k("mny_pw_wgrad", x), _p(x), _p(sc), _p(sh), act, _p(dy), _p(dw), _p(db), _p(ws), M, K, Nc, _st())
    return dw, db


def pw_bnbwd(g, y, scale, shift, act, mean, invstd, gamma, xview, w2d, addend=None, want_dx=True):
    """Fused BN-backward + wgrad + dgrad of a thin expand unit -> (dx|None, dw, dgamma, dbeta)."""
    x, xs, xh, xact = xview
    K, Nc = x.shape[-1], y.shape[-1]
    M = y.numel() // Nc
    assert query(_k("mny_pw_bnbwd_supported", y), M, K, Nc) == 1
    ws = _new(query("mny_pw_bnbwd_ws_floats", M, K, Nc), like=y, dtype=torch.float32)
    dx = torch.empty_like(x) if want_dx else None
    dw, dgamma, dbeta = _new(Nc, K, like=y, dtype=torch.float32), _new(Nc, like=y, dtype=torch.float32), _new(Nc, like=y, dtype=torch.float32)
    call(_k("mny_pw_bnbwd", y), _p(g), _p(y), _p(scale), _p(shift), act, _p(mean), _p(invstd), _p(gamma), _p(x), _p(xs), _p(xh), xact,
         _p(w2d), _p(addend), _p(dx), _p(dw), _p(dgamma), _p(dbeta), _p(ws), M, K, Nc, _st())
    return dx, dw, dgamma, dbeta


def pw_dgrad_bnred(dy, wT, y_unit, scale, shift, act, mean, invstd, addend=None):
    """dx = dy @ W (wT = [Nc][K]) [+ addend] and the BN-backward partial sums of the unit whose (complete) output gradient dx is.
    -> (dx [M,Nc], red [parts,2,Nc])"""
    M, K = dy.numel() // dy.shape[-1], dy.shape[-1]
    Nc = wT.shape[0]
    parts = query("mny_pw_dgrad_bnred_parts", M, K, Nc)
    dx = torch.empty(*dy.shape[:-1], Nc, device=dy.device, dtype=torch.float32)
    red = torch.empty(parts, 2, Nc, device=dy.device, dtype=torch.float32)
    if addend is None:
        call("mny_pw_dgrad_bnred", _p(dy), _p(wT), _p(dx), _p(y_unit), _p(scale), _p(shift), int(act), _p(mean), _p(invstd), _p(red), M, K, Nc, _st())
    else:
        call("mny_pw_dgrad_bnred_add", _p(dy), _p(wT), _p(addend), _p(dx), _p(y_unit), _p(scale), _p(shift), int(act), _p(mean), _p(invstd), _p(red),
             M, K, Nc, _st())
    return dx, red


def transpose(w2d, dtype=torch.float32):
    R, C = w2d.shape
    out = _new(C, R, like=w2d, dtype=dtype)
    call(_k("mny_transpose", out), _p(w2d), _p(out), R, C, _st())
    return out


# ---- batch norm -------------------------------------------------------------------------------------
def bn_finalize(stats, count, gamma, beta, running_mean=None, running_var=None):
    C = gamma.numel()
    scale, shift, mean, invstd = (_new(C, like=gamma) for _ in range(4))
    call("mny_bn_finalize", _p(stats), stats.shape[0], count, _p(gamma), _p(beta), BN_EPS, BN_MOMENTUM,
         _p(running_mean), _p(running_var), _p(scale), _p(shift), _p(mean), _p(invstd), C, _st())
    return scale, shift, mean, invstd


def bn_eval_coeffs(gamma, beta, rm, rv):
    C = gamma.numel()
    scale, shift = _new(C, like=gamma), _new(C, like=gamma)
    call("mny_bn_eval_coeffs", _p(gamma), _p(beta), _p(rm), _p(rv), BN_EPS, _p(scale), _p(shift), C, _st())
    return scale, shift


def bn_backward(g, y, scale, shift, act, gamma, mean, invstd, out=None):
    """-> (dy, dgamma, dbeta) for z = scale*y+shift, a = act(z), given g = dL/da."""
    C = y.shape[-1]
    M = y.numel() // C
    parts = query("mny_bn_bwd_parts", M, C)
    red = _new(parts, 2, C, like=y)
    call(_k("mny_bn_bwd_reduce", y), _p(g), _p(y), _p(scale), _p(shift), act, _p(mean), _p(invstd), _p(red), M, C, _st())
    dgamma, dbeta, coef = _new(C, like=y), _new(C, like=y), _new(3, C, like=y)
    call("mny_bn_bwd_finalize", _p(red), parts, M, _p(gamma), _p(mean), _p(invstd), _p(dgamma), _p(dbeta), _p(coef), C, _st())
    dy = out if out is not None else torch.empty_like(y)
    call(_k("mny_bn_bwd_apply", y), _p(g), _p(y), _p(scale), _p(shift), act, _p(coef), _p(dy), M, C, _st())
    return dy, dgamma, dbeta


# ---- glue -------------------------------------------------------------------------------------------
def add_views(a, b=None, up=None, out=None):
    x = a[0]
    N, H, W, C = x.shape
    o = out if out is not None else torch.empty_like(x)
    bt, bsc, bsh, bact = b if b is not None else (None, None, None, ACT_NONE)
    call(_k("mny_add_views", x), _p(a[0]), _p(a[1]), _p(a[2]), a[3], _p(bt), _p(bsc), _p(bsh), bact, _p(up), _p(o), N, H, W, C, _st())
    return o


def upsample_bwd(src, dst=None, accumulate=False):
    N, H, W, C = src.shape
    d = dst if dst is not None else _new(N, H // 2, W // 2, C, like=src, dtype=src.dtype)
    call(_k("mny_upsample_bwd", src), _p(src), _p(d), int(accumulate), N, H, W, C, _st())
    return d


def axpy(src, dst, alpha=None, accumulate=False):
    call(_k("mny_axpy", src), _p(src), _p(alpha), _p(dst), int(accumulate), src.numel(), _st())
    return dst


# ---- detection --------------------------------------------------------------------------------------
def make_head(N, g, A, C, n_anchors_all, ignore_thresh, iou_thresh, iou_weighting):
    return YoloHead(N, g, A, C, n_anchors_all, ignore_thresh, iou_thresh, iou_weighting)


def yolo_loss(head, targets, t_off, anchors_all, mask, hp):
    """head [N,g,g,A*(5+C)] -> (out7 [7] device tensor, dhead)."""
    ws = torch.empty(query("mny_yolo_loss_ws_bytes", ctypes.byref(hp), 0), device=head.device, dtype=torch.uint8)
    out7 = _new(7, like=head)
    dhead = torch.empty_like(head)
    call("mny_yolo_loss", _p(head), _p(targets), _p(t_off), _p(anchors_all), _p(mask), ctypes.byref(hp),
         _p(out7), _p(dhead), ctypes.c_void_p(ws.data_ptr()), _st())
    return out7, dhead


def yolo_decode(head, anchors_all, mask, hp, val_conf, rows=None, row_stride=None, base_counts=None):
    cells = hp.A * hp.g * hp.g
    row_stride = row_stride or cells
    if rows is None:
        rows = _new(hp.N, row_stride, 7, like=head)
    counts = _new(hp.N, like=head, dtype=torch.int32)
    call("mny_yolo_decode", _p(head), _p(anchors_all), _p(mask), ctypes.byref(hp), float(val_conf), _p(rows), int(row_stride),
         _p(base_counts), _p(counts), _st())
    return rows, counts


def nms_per_class(rows, seg_begin, seg_count, num_classes, thr=0.45, max_seg_rows=0, gather=True):
    """rows [capacity,7]; seg_begin/seg_count int32 [S] (device).
    -> (out_idx int32 [capacity], out_counts int32 [S], out_rows [capacity,7]|None, prefix int32 [S+1], status int32 [1])"""
    capacity = rows.shape[0]
    S = seg_begin.numel()
    nbytes = query("mny_nms_ws_bytes", S, capacity, num_classes)
    ws = torch.empty(nbytes, device=seg_begin.device, dtype=torch.uint8)
    out_idx = torch.empty(max(capacity, 1), device=seg_begin.device, dtype=torch.int32)
    out_counts = torch.empty(S, device=seg_begin.device, dtype=torch.int32)
    out_rows = torch.empty(max(capacity, 1), 7, device=seg_begin.device) if gather else None
    call("mny_nms_per_class", _p(rows) if capacity else None, _p(seg_begin), _p(seg_count), S, capacity, int(max_seg_rows), num_classes,
         float(thr), _p(out_idx), _p(out_counts), _p(out_rows), ctypes.c_void_p(ws.data_ptr()), _st())
    so = query("mny_nms_status_offset", S, capacity, num_classes)
    po = query("mny_nms_prefix_offset", S, capacity, num_classes)
    status = ws[so:so + 4].view(torch.int32)
    prefix = ws[po:po + 4 * (S + 1)].view(torch.int32)
    return out_idx, out_counts, out_rows, prefix, status
