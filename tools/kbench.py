#!/usr/bin/env python3
"""Single-kernel micro-benchmarks through the C ABI (for rocprofv3 runs).
usage: python tools/kbench.py pw M K N [reps] | wgrad M K N | dw N H W C stride | dwbw ..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mobilenet_yolo_pytorch_amd import ops  # noqa: E402


def timeit(fn, reps):
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
    kind = sys.argv[1]
    args = [int(v) for v in sys.argv[2:]]
    dev = "cuda"
    if kind in ("pw", "pwx", "wgrad"):
        M, K, N = args[:3]
        reps = args[3] if len(args) > 3 else 20
        x = torch.randn(1, 1, M, K, device=dev)
        w = torch.randn(N, K, device=dev) * K ** -0.5
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        if kind == "pw":
            ms = timeit(lambda: ops.pw_fwd((x, None, None, 0), w, want_stats=False), reps)
        elif kind == "pwx":
            ms = timeit(lambda: ops.pw_fwd((x, sc, sh, 1), w, want_stats=True), reps)
        else:
            dy = torch.randn(1, 1, M, N, device=dev)
            ms = timeit(lambda: ops.pw_wgrad((x, sc, sh, 1), dy), reps)
        print("%s M%d K%d N%d: %.3f ms  %.1f TF/s  %.1f GB/s" % (kind, M, K, N, ms, 2 * M * K * N / ms / 1e9, 4 * (M * K + M * N) / ms / 1e6))
    elif kind == "bnbwd":
        M, K, N = args[:3]
        reps = args[3] if len(args) > 3 else 10
        x = torch.randn(1, 1, M, K, device=dev)
        w = torch.randn(N, K, device=dev) * K ** -0.5
        xs, xh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        gamma, beta = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1
        y, st = ops.pw_fwd((x, xs, xh, 1), w)
        scale, shift, mean, invstd = ops.bn_finalize(st, M, gamma, beta)
        g = torch.randn(1, 1, M, N, device=dev)
        ms = timeit(lambda: ops.pw_bnbwd(g, y, scale, shift, 1, mean, invstd, gamma, (x, xs, xh, 1), w), reps)
        print("bnbwd M%d K%d N%d: %.3f ms  %.1f GB/s (4 passes over Y-sized tensors)" % (M, K, N, ms, 4 * (4 * M * N + 2 * M * K) / ms / 1e6))
    elif kind in ("dw", "dwplain", "dwbw", "dwbd"):
        N, H, W, C, s = args[:5]
        reps = args[5] if len(args) > 5 else 20
        x = torch.randn(N, H, W, C, device=dev)
        w = torch.randn(C, 1, 3, 3, device=dev) * 0.3
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        dy = torch.randn(N, Ho, Wo, C, device=dev)
        if kind == "dw":
            ms = timeit(lambda: ops.dw_fwd((x, sc, sh, 1), w, s), reps)
        elif kind == "dwplain":
            ms = timeit(lambda: ops.dw_fwd((x, None, None, 0), w, s, want_stats=False), reps)
        elif kind == "dwbw":
            ms = timeit(lambda: ops.dw_bwd_weight((x, sc, sh, 1), dy, 3, s), reps)
        else:
            ms = timeit(lambda: ops.dw_bwd_data(dy, w, (H, W), s), reps)
        print("%s N%d H%d C%d s%d: %.3f ms  %.1f GB/s" % (kind, N, H, C, s, ms, 4 * (x.numel() + dy.numel()) / ms / 1e6))


if __name__ == "__main__":
    main()
