"""Error of mny_pw_fwd against an fp64 product, native fp32 MFMA (MNY_X6=0) vs the six-product bf16 form (MNY_X6=1):
    MNY_X6=0 python tools/x6_precision.py ; MNY_X6=1 python tools/x6_precision.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, ".")
from mobilenet_yolo_pytorch_amd import _lib  # noqa: E402

st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
for M, K, N in ((4096, 512, 512), (4096, 96, 576), (4096, 960, 160), (4096, 64, 384)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    y = torch.empty(M, N, device="cuda")
    _lib.call("mny_pw_fwd", p(x), None, None, 0, p(w), None, None, p(y), None, M, K, N, st)
    ref = x.double() @ w.double().t()
    err = (y.double() - ref).abs()
    print("MNY_X6=%s M=%d K=%d N=%d: max abs err %.3e, rms err %.3e (rms of the result %.3f)" % (
        os.environ.get("MNY_X6", "unset"), M, K, N, err.max().item(), err.pow(2).mean().sqrt().item(), ref.pow(2).mean().sqrt().item()))
