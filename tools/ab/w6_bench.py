"""Micro-benchmark of the pre-cut-planes forward GEMM (mny_pw_fwd_w6) on the fat shapes of the headline plan; prints time, TF/s and a
checksum per shape (same-box A/B of two library builds: MNY_LIB=tools/ab/libmnyolo_prev.so python tools/ab/w6_bench.py).
Also checks the result against mny_pw_fwd (fp32 MFMA path, MNY_X6=0 in a child is not needed: tolerance 1e-5 relative to max)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from mobilenet_yolo_pytorch_amd import _lib

p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = ((123904, 512, 512, 1), (123904, 512, 512, 0), (30976, 1280, 512, 1), (30976, 512, 1024, 1), (30976, 512, 512, 1), (123904, 96, 576, 1),
          (123904, 64, 384, 1), (30976, 160, 960, 1), (30976, 320, 1280, 1), (123904, 576, 96, 1), (123904, 384, 64, 1), (30976, 960, 160, 1))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for M, K, N, act in shapes:
    if _lib.query("mny_pw_w6_supported", M, K, N) != 1:
        print("M%d K%d N%d: not a planes shape" % (M, K, N)); continue
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    sc, sh = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.3 * torch.randn(K, generator=g)).cuda()
    planes = torch.empty(_lib.query("mny_pw_w6_bytes", K, N), device="cuda", dtype=torch.uint8)
    nb = (N * ((K + 15) // 16) * 2 + 255) // 256
    jt = np.array([(w.data_ptr(), planes.data_ptr(), N, K, 0, 0)], dtype=np.dtype([("src", np.uint64), ("dst", np.uint64), ("R", np.int32), ("C", np.int32), ("b0", np.int32), ("pad", np.int32)]))
    jd = torch.from_numpy(jt.view(np.uint8).copy()).cuda(); bj = torch.zeros(nb, dtype=torch.int32, device="cuda")
    _lib.call("mny_cut3_batch", p(jd), p(bj), nb, stream)
    parts = _lib.query("mny_pw_stat_parts", M, K, N)
    st = torch.empty(parts, 2, N, device="cuda"); y = torch.empty(M, N, device="cuda")
    a_sc, a_sh = (p(sc), p(sh)) if act else (None, None)
    run = lambda: _lib.call("mny_pw_fwd_w6", p(x), a_sc, a_sh, act, p(planes), None, None, p(y), p(st), M, K, N, stream)
    run(); run(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    y2 = torch.empty(M, N, device="cuda"); st2 = torch.empty(parts, 2, N, device="cuda")
    _lib.call("mny_pw_fwd", p(x), a_sc, a_sh, act, p(w), None, None, p(y2), p(st2), M, K, N, stream)
    torch.cuda.synchronize()
    err = float((y - y2).abs().max() / y2.abs().max())
    serr = float((st.sum(0) - st2.sum(0)).abs().max() / st2.sum(0).abs().max())
    print("M%-7d K%-5d N%-5d act%d: %.4f ms  %6.1f TF/s  y-err %.1e stats-err %.1e  sum %.6e" % (M, K, N, act, ms, 2 * M * K * N / ms / 1e9, err, serr, float(y.double().sum())))
