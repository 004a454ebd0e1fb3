"""Per-shape timing of the short-reduction pointwise kernels (pwthin.hip) against the matrix-core tile kernel:
    python tools/bench_thin.py            # routed as shipped
    MNY_NO_THIN=1 python tools/bench_thin.py
Shapes: the K <= 32 pointwise convs of the headline step (MobileNetV2-YOLO 352x352, batch 256) and of configs[3] (bf16)."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from mobilenet_yolo_pytorch_amd import _lib  # noqa: E402

FWD = [(7929856, 32, 16, 1), (7929856, 16, 96, 0), (1982464, 24, 144, 0), (495616, 32, 192, 0)]
RED = [(7929856, 16, 32, 1), (1982464, 24, 96, 1), (1982464, 24, 144, 1), (495616, 32, 144, 1), (495616, 32, 192, 1)]


def p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for dtype, sfx in ((torch.float32, ""), (torch.bfloat16, "_bf16")):
        es = 4 if dtype == torch.float32 else 2
        for M, K, N, act in FWD:
            x = torch.randn(M, K, device="cuda").to(dtype)
            w = torch.randn(N, K, device="cuda").to(dtype)
            sc, sh = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
            y = torch.empty(M, N, device="cuda", dtype=dtype)
            parts = _lib.query("mny_pw_stat_parts" + sfx, M, K, N)
            stt = torch.empty(parts, 2, N, device="cuda")
            ms = timeit(lambda: _lib.call("mny_pw_fwd" + sfx, p(x), p(sc), p(sh), act, p(w), None, None, p(y), p(stt), M, K, N, st))
            print("fwd%-5s M=%8d K=%2d N=%3d  %.3f ms  %.0f GB/s" % (sfx, M, K, N, ms, M * (K + N) * es / ms / 1e6))
        for M, K, N, act in RED:
            dy = torch.randn(M, K, device="cuda").to(dtype)
            w = torch.randn(N, K, device="cuda").to(dtype)
            yraw = torch.randn(M, N, device="cuda").to(dtype)
            c = [torch.ones(N, device="cuda") for _ in range(4)]
            dx = torch.empty(M, N, device="cuda", dtype=dtype)
            parts = _lib.query("mny_pw_dgrad_bnred_parts" + sfx, M, K, N)
            red = torch.empty(parts, 2, N, device="cuda")
            ms = timeit(lambda: _lib.call("mny_pw_dgrad_bnred" + sfx, p(dy), p(w), p(dx), p(yraw), p(c[0]), p(c[1]), act, p(c[2]), p(c[3]), p(red), M, K, N, st))
            print("red%-5s M=%8d K=%2d N=%3d  %.3f ms  %.0f GB/s" % (sfx, M, K, N, ms, M * (K + 2 * N) * es / ms / 1e6))


if __name__ == "__main__":
    main()
