#!/usr/bin/env python3
"""Determinism / correctness stress of the 128 x 256 weight-gradient tile (pw_wgrad_dma_kernel<0,2,4,1>, 254 VGPRs): 40 launches each of the
shapes the headline plan sends to it, every result compared bit for bit with the first and (first launch) with an fp64 reference of a row sample."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mobilenet_yolo_pytorch_amd import _lib  # noqa: E402

P = ctypes.c_void_p
ptr = lambda t: P(t.data_ptr()) if t is not None else None  # noqa: E731
dev = torch.device("cuda:0")
st = P(torch.cuda.current_stream().cuda_stream)
bad = 0
for M, K, N in ((123904, 512, 512), (30976, 1280, 512), (30976, 512, 1024), (30976, 512, 512), (7744 + 13, 512, 512), (99999, 256, 128)):
    torch.manual_seed(M + K)
    x = torch.randn(M, K, device=dev)
    dy = torch.randn(M, N, device=dev) * 0.1
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.2
    splits = _lib.query("mny_pw_wgrad_splits", M, K, N)
    ws = torch.empty(int(_lib.query("mny_pw_wgrad_ws_floats", M, K, N)), device=dev)
    dw = torch.empty(N, K, device=dev)
    first = None
    for it in range(40):
        ws.fill_(float("nan"))
        _lib.call("mny_pw_wgrad", ptr(x), ptr(sc), ptr(sh), 1, ptr(dy), ptr(dw), None, ptr(ws), M, K, N, st)
        torch.cuda.synchronize()
        if first is None:
            first = dw.clone()
            a = torch.clamp(x.double() * sc.double() + sh.double(), 0, 6)
            ref = dy.double().t() @ a
            err = ((first.double() - ref).abs().max() / ref.abs().max()).item()
            print("M%d K%d N%d splits %d: rel. error vs fp64 %.2e, finite %s" % (M, K, N, splits, err, bool(torch.isfinite(first).all())))
            bad += int(err > 2e-6 or not torch.isfinite(first).all())
        elif not torch.equal(first, dw):
            bad += 1
            print("  launch %d differs: %d elements" % (it, int((first != dw).sum())))
print("FAILED" if bad else "all launches bit-identical and correct")
sys.exit(1 if bad else 0)
