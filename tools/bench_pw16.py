#!/usr/bin/env python3
"""bf16-storage pointwise GEMM timings at MobileNetV3-YOLO 512x512 bs-64 shapes: forward with / without the BN view and the statistics
epilogue (what the entry point costs beyond the matrix product).  python tools/bench_pw16.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mobilenet_yolo_pytorch_amd import _lib  # noqa: E402

P = ctypes.c_void_p
ptr = lambda t: P(t.data_ptr()) if t is not None else None  # noqa: E731


def timeit(fn, reps=30):
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
    dev = torch.device("cuda:0")
    st = P(torch.cuda.current_stream().cuda_stream)
    shapes = [(65536, 672, 160), (65536, 160, 672), (65536, 112, 672), (65536, 672, 112), (65536, 320, 640), (16384, 960, 960), (16384, 960, 160), (16384, 160, 960),
              (262144, 40, 240), (262144, 40, 120), (262144, 120, 40), (262144, 240, 80), (65536, 80, 480), (65536, 80, 200), (65536, 200, 80), (1048576, 24, 72), (1048576, 72, 24),
              (4194304, 16, 64), (4194304, 16, 16)]
    for M, K, N in shapes:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        parts = _lib.query("mny_pw_stat_parts_bf16", M, K, N)
        stats = torch.zeros(parts * 2 * N, device=dev)
        res = []
        for act, use_stats in ((0, False), (4, False), (4, True), (1, True)):
            a_sc, a_sh = (None, None) if act == 0 else (sc, sh)
            fn = lambda: _lib.call("mny_pw_fwd_bf16", ptr(x), ptr(a_sc), ptr(a_sh), act, ptr(w), None, None, ptr(y), ptr(stats) if use_stats else None, M, K, N, st)  # noqa: E731
            res.append(timeit(fn))
        gb = 2 * (M * K + M * N) / 1e9
        print("M%-8d K%-4d N%-4d  plain %.3f  hswish-view %.3f  +stats %.3f  relu6-view+stats %.3f ms   (%.0f GB/s, %.0f TF/s at the last; HBM floor %.3f ms)" % (
            M, K, N, res[0], res[1], res[2], res[3], gb / res[3] * 1e3, 2 * M * K * N / res[3] / 1e9, gb / 6.0))


if __name__ == "__main__":
    main()
