#!/usr/bin/env python3
"""Same-box timing of mny_stemdw_bwd against the three launches it replaces at the bs-256 / 352x352 shape.  usage: python tools/bench_stemdw.py [bs] [size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from mobilenet_yolo_pytorch_amd import _lib  # noqa: E402
from test_gpu_stemdw import forward_on_gpu, ptr, stream  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(2):
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
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 352
    dev, C = torch.device("cuda:0"), 32
    torch.manual_seed(0)
    x = torch.randn(N, 3, H, W, device=dev)
    ws, wd = torch.randn(C, 3, 3, 3, device=dev) * 0.3, torch.randn(C, 3, 3, device=dev) * 0.4
    gs, bs, gd_, bd_ = (torch.rand(C, device=dev) + 0.5 for _ in range(4))
    Ho, Wo = H // 2, W // 2
    M = N * Ho * Wo
    s, sc, d, dc = forward_on_gpu(x, ws, wd, gs, bs, gd_, bd_, dev)
    st = stream()
    gout = torch.randn(N, Ho, Wo, C, device=dev)
    dcoef = torch.stack((torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.05, torch.randn(C, device=dev) * 0.05)).contiguous()
    dparts = _lib.query("mny_dw_bnbwd_parts", N, Ho, Wo, C)
    wsb = torch.zeros(max(dparts * C * 9, _lib.query("mny_stem_wgrad_parts", N, H, W, C) * C * 27), device=dev)
    inred = torch.zeros(dparts * 2 * C, device=dev)
    gsb = torch.empty(N, Ho, Wo, C, device=dev)
    dwd0, dgs0, dbs0, scoef, dws0 = torch.zeros(C, 3, 3, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(3, C, device=dev), torch.zeros(C, 3, 3, 3, device=dev)

    def three():
        _lib.call("mny_dw_bnbwd_red", ptr(gout), ptr(d), ptr(dc[0]), ptr(dc[1]), _lib.ACT_RELU6, ptr(dcoef), ptr(s), ptr(sc[0]), ptr(sc[1]), _lib.ACT_RELU6,
                  ptr(sc[2]), ptr(sc[3]), ptr(wd), None, ptr(gsb), ptr(dwd0), ptr(wsb), ptr(inred), N, Ho, Wo, C, 3, 1, st)
        _lib.call("mny_bn_bwd_finalize", ptr(inred), dparts, M, ptr(gs), ptr(sc[2]), ptr(sc[3]), ptr(dgs0), ptr(dbs0), ptr(scoef), C, st)
        _lib.call("mny_stem_bnwgrad", ptr(x), ptr(gsb), ptr(s), ptr(sc[0]), ptr(sc[1]), _lib.ACT_RELU6, ptr(scoef), ptr(dws0), ptr(wsb), N, H, W, C, st)

    parts = _lib.query("mny_stemdw_bwd_parts", N, H, W, C)
    ws1 = torch.zeros(int(_lib.query("mny_stemdw_bwd_ws_floats", N, H, W, C)), device=dev)
    dwws = torch.zeros(parts * C * 9, device=dev)
    dws1, dgs1, dbs1, dwd1 = torch.zeros(C, 3, 3, 3, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, 3, 3, device=dev)

    def one():
        _lib.call("mny_stemdw_bwd", ptr(gout), ptr(d), ptr(dc[0]), ptr(dc[1]), _lib.ACT_RELU6, ptr(dcoef), ptr(s), ptr(sc[0]), ptr(sc[1]), ptr(sc[2]), ptr(sc[3]),
                  ptr(gs), _lib.ACT_RELU6, ptr(x), ptr(ws), ptr(wd), ptr(dws1), ptr(dgs1), ptr(dbs1), ptr(dwd1), ptr(dwws), ptr(ws1), N, H, W, C, st)

    t3, t1 = timeit(three), timeit(one)
    print("N%d %dx%d: dw_bnbwd_red + finalize + stem_bnwgrad %.3f ms  |  stemdw_bwd %.3f ms" % (N, H, W, t3, t1))
    for name, a, b in (("dw_stem", dws0, dws1), ("dgamma_s", dgs0, dgs1), ("dbeta_s", dbs0, dbs1), ("dw_dw", dwd0, dwd1)):
        print("    %-9s max |three - one| = %.3e (max |ref| %.3e)" % (name, (a.double() - b.double()).abs().max().item(), a.abs().max().item()))


if __name__ == "__main__":
    main()
