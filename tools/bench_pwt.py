#!/usr/bin/env python3
"""Timing of the thin pointwise convs of the MobileNetV3-YOLO 512x512 bs-64 plan on bf16 storage (K <= 48 at >= 131072 pixels: the wave-per-16-pixels
kernel of csrc/gate.hip; MNY_NO_PWT=1: the kernels they took before).  python tools/bench_pwt.py  -> ms and algorithmic GB/s per shape, forward with
BN statistics behind a ReLU view, and the plain data gradient."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mobilenet_yolo_pytorch_amd import ops  # noqa: E402


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
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    total = 0.0
    for M, K, N, count in ((4194304, 16, 64, 1), (4194304, 16, 16, 3), (1048576, 24, 72, 2), (1048576, 24, 64, 1), (262144, 40, 120, 2), (262144, 40, 240, 1)):
        torch.manual_seed(K * 1000 + N)
        x = torch.randn(1, 1, M, K, device=dev).to(bf)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
        sc, sh = 1 + 0.2 * torch.randn(K, device=dev), 0.3 * torch.randn(K, device=dev)
        y = torch.empty(1, 1, M, N, device=dev, dtype=bf)
        t = timeit(lambda: ops.pw_fwd((x, sc, sh, 3), w, want_stats=True, out=y))
        gb = 2 * M * (K + N) / 1e9
        yy, st = ops.pw_fwd((x, sc, sh, 3), w, want_stats=True, out=y)
        print("M%-8d K%-3d N%-4d fwd+stats %.3f ms  %6.0f GB/s   checksum y %.6e stats %.6e" % (M, K, N, t, gb / t * 1e3, yy.float().double().sum().item(),
                                                                                       st.double().sum().item()), flush=True)
        total += count * t
    print("plan share (forward launches x count): %.3f ms" % total)


if __name__ == "__main__":
    main()
