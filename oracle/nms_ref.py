"""Oracle (test infrastructure, NOT product): ctypes wrapper over nms_ref.c.

`build()` compiles oracle/nms_ref.c -> oracle/_build/libnms_ref.so with gcc.
`nms(boxes, scores, thr)` mirrors torchvision.ops.nms (PARITY UNPINNED: see
nms_ref.c header); `nms_driver(preds, num_classes)` mirrors utils/box.py:11-31.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libnms_ref.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "nms_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-shared", "-fPIC", "-o", _SO, src])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_nms_kernel.restype = ctypes.c_int64
        _lib.oracle_nms_kernel.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                           ctypes.c_int64, ctypes.c_double, ctypes.c_void_p]
        _lib.oracle_nms_per_class.restype = ctypes.c_int64
        _lib.oracle_nms_per_class.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                              ctypes.c_double, ctypes.c_void_p]
    return _lib


def nms(boxes, scores, thr):
    """torchvision.ops.nms(boxes[n,4], scores[n], thr) -> int64 kept indices."""
    lib = _load()
    b = np.ascontiguousarray(boxes.detach().cpu().numpy(), dtype=np.float32).reshape(-1, 4)
    s = np.ascontiguousarray(scores.detach().cpu().numpy(), dtype=np.float32).reshape(-1)
    n = b.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int64)
    k = lib.oracle_nms_kernel(b.ctypes.data, 4, s.ctypes.data, n, float(thr), keep.ctypes.data)
    return torch.from_numpy(keep[:k].copy())


def nms_rows(rows, num_classes, thr=0.45):
    """Per-class NMS over one image's rows [n,7] -> (kept rows [k,7], kept indices)."""
    lib = _load()
    r = np.ascontiguousarray(rows.detach().cpu().numpy(), dtype=np.float32).reshape(-1, 7)
    n = r.shape[0]
    idx = np.empty(max(n, 1), dtype=np.int64)
    k = lib.oracle_nms_per_class(r.ctypes.data, n, int(num_classes), float(thr), idx.ctypes.data)
    idx = torch.from_numpy(idx[:k].copy())
    return rows.detach().cpu().reshape(-1, 7)[idx], idx


def nms_driver(preds, num_classes):
    """utils/box.py:11-31: preds = (head0 rows per image, head1 rows per image)."""
    assert len(preds) == 2                                  # :13
    assert len(preds[0]) == len(preds[1])                   # :14
    out = []
    for b in range(len(preds[0])):
        rows = torch.cat((preds[0][b], preds[1][b]), 0)     # :17
        out.append(nms_rows(rows, num_classes)[0] if rows.shape[0] else torch.zeros(0, 7))
    return out
