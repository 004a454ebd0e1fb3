#!/usr/bin/env python3
"""Timing of the depthwise forward kernels (csrc/dwconv.hip: dw3_fwd_kernel / dw5_fwd_kernel) at the shapes of a plan.
    python tools/bench_dwfwd.py [bf16|f32]
bf16: the 21 depthwise layers of the MobileNetV3-YOLO 512x512 bs-64 plan (BASELINE configs[3]); f32: those of the MobileNetV2-YOLO
352x352 bs-256 plan.  Prints ms, algorithmic GB/s and an fp64 checksum of the output and of the statistics rows per shape (A/B runs of
two builds / two MNY_DW_PF settings must print identical checksums).  Knobs are read once per process: run once per setting."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mobilenet_yolo_pytorch_amd import _lib  # noqa: E402

P = ctypes.c_void_p
ptr = lambda t: P(t.data_ptr()) if t is not None else None  # noqa: E731
RELU, RELU6, HSWISH = 3, 1, 4


def timeit(fn, reps):
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
    bf = len(sys.argv) < 2 or sys.argv[1] == "bf16"
    dt = torch.bfloat16 if bf else torch.float32
    sfx = "_bf16" if bf else ""
    if bf:      # (C, H, k, stride, input activation, count) — models/mobilenetv3.py:84-102 at 512x512, bs 64, + the neck (mbv3_yolo.py)
        bs = 64
        shapes = ((16, 256, 3, 1, RELU, 1), (64, 256, 3, 2, RELU, 1), (72, 128, 3, 1, RELU, 1), (72, 128, 5, 2, RELU, 1), (120, 64, 5, 1, RELU, 2),
                  (240, 64, 3, 2, HSWISH, 1), (200, 32, 3, 1, HSWISH, 1), (184, 32, 3, 1, HSWISH, 2), (480, 32, 3, 1, HSWISH, 1),
                  (672, 32, 3, 1, HSWISH, 1), (672, 32, 5, 1, HSWISH, 1), (672, 32, 5, 2, HSWISH, 1), (960, 16, 5, 1, HSWISH, 1),
                  (960, 16, 3, 1, 2, 1), (160, 32, 3, 1, 2, 2), (320, 32, 3, 1, 2, 1), (320, 16, 3, 1, 2, 2))
    else:
        bs = 256
        shapes = ((32, 176, 3, 1, RELU6, 1), (96, 176, 3, 2, RELU6, 1), (144, 88, 3, 1, RELU6, 1), (144, 88, 3, 2, RELU6, 1), (192, 44, 3, 1, RELU6, 2),
                  (192, 44, 3, 2, RELU6, 1), (384, 22, 3, 1, RELU6, 4), (576, 22, 3, 1, RELU6, 2), (576, 22, 3, 2, RELU6, 1), (960, 11, 3, 1, RELU6, 3))
    dev = torch.device("cuda:0")
    st = P(torch.cuda.current_stream().cuda_stream)
    total = 0.0
    for C, H, k, s, act, count in shapes:
        torch.manual_seed(C + H)
        x = torch.randn(bs, H, H, C, device=dev).to(dt)
        w = (torch.randn(C, 1, k, k, device=dev) * 0.3).contiguous()
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        Ho = (H - 1) // s + 1
        y = torch.empty(bs, Ho, Ho, C, device=dev, dtype=dt)
        parts = _lib.query("mny_dw_stat_parts_x", bs, H, H, C, k, s, 1 if bf else 0)
        stats = torch.zeros(parts, 2, C, device=dev)
        fn = lambda: _lib.call("mny_dw_fwd" + sfx, ptr(x), ptr(sc), ptr(sh), act, ptr(w), ptr(y), ptr(stats), bs, H, H, C, k, s, st)  # noqa: E731
        ms = timeit(fn, 20)
        eb = 2 if bf else 4
        gb = eb * (x.numel() + y.numel()) / 1e9
        total += ms * count
        print("dw%d C%-4d H%-4d s%d act%d x%d: %7.3f ms  %7.1f GB/s   y %.9e  stats %.9e" % (
            k, C, H, s, act, count, ms, gb / ms * 1e3, y.double().sum().item(), stats.double().sum().item()))
    print("plan share (sum of ms x count): %.3f ms" % total)


if __name__ == "__main__":
    main()
