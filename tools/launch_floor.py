#!/usr/bin/env python3
"""What does a dependent launch cost?  The smallest entry point (mny_bn_eval_stats, 16 channels, one workgroup) 2 000 times in one stream,
and mny_bn_finalize at a few (parts, C): time per launch with HIP events around the whole run."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mobilenet_yolo_pytorch_amd import _lib  # noqa: E402

P = ctypes.c_void_p
ptr = lambda t: P(t.data_ptr())  # noqa: E731


def per_launch(fn, n=2000):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    st = P(torch.cuda.current_stream().cuda_stream)
    C = 16
    rm, rv, m, i = (torch.ones(C, device=dev) for _ in range(4))
    print("mny_bn_eval_stats C=16 (1 workgroup of 256): %.2f us per launch" % per_launch(lambda: _lib.call("mny_bn_eval_stats", ptr(rm), ptr(rv), 1e-5, ptr(m), ptr(i), C, st)))
    for parts, C in ((768, 32), (768, 672), (256, 160), (64, 16)):
        stats = torch.rand(parts * 2 * C, device=dev)
        g, b, rm, rv = (torch.ones(C, device=dev) for _ in range(4))
        co = torch.zeros(4, C, device=dev)
        fn = lambda: _lib.call("mny_bn_finalize", ptr(stats), parts, 100000, ptr(g), ptr(b), 1e-5, 0.1, ptr(rm), ptr(rv), ptr(co[0]), ptr(co[1]), ptr(co[2]), ptr(co[3]), C, st)  # noqa: E731
        print("mny_bn_finalize parts=%d C=%d: %.2f us per launch" % (parts, C, per_launch(fn)))


if __name__ == "__main__":
    main()
