#!/usr/bin/env python3
"""Same-box timing of mny_pj_bwd against mny_bn_bwd_apply + mny_pw_dgrad_bnred + mny_pw_wgrad at the project-conv shapes of the bs-256 / 352x352
plan.  usage: python tools/bench_pjbwd.py [bs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from mobilenet_yolo_pytorch_amd import _lib  # noqa: E402
from test_gpu_pjbwd import make, ptr, stream  # noqa: E402


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
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda:0")
    act = _lib.ACT_RELU6
    tot3 = tot1 = 0.0
    for Ki, No, H, count in ((32, 16, 176, 1), (96, 24, 88, 1), (144, 24, 88, 1), (144, 32, 44, 1), (192, 32, 44, 2)):
        M = bs * H * H
        G, Y, coef, D, dsc, dsh, dmu, dis, W = make(M, Ki, No, seed=Ki, dev=dev)
        st = stream()
        dY = torch.empty_like(G)
        one, zero = torch.ones(No, device=dev), torch.zeros(No, device=dev)
        wT = W.t().contiguous()
        rparts = _lib.query("mny_pw_dgrad_bnred_parts", M, No, Ki)
        rbuf = torch.zeros(rparts, 2, Ki, device=dev)
        gd0 = torch.empty(M, Ki, device=dev)
        dw0 = torch.zeros(No, Ki, device=dev)
        ws = torch.zeros(int(_lib.query("mny_pw_wgrad_ws_floats", M, Ki, No)) + 16, device=dev)

        def three():
            _lib.call("mny_bn_bwd_apply", ptr(G), ptr(Y), ptr(one), ptr(zero), _lib.ACT_NONE, ptr(coef), ptr(dY), M, No, st)
            _lib.call("mny_pw_dgrad_bnred", ptr(dY), ptr(wT), ptr(gd0), ptr(D), ptr(dsc), ptr(dsh), act, ptr(dmu), ptr(dis), ptr(rbuf), M, No, Ki, st)
            _lib.call("mny_pw_wgrad", ptr(D), ptr(dsc), ptr(dsh), act, ptr(dY), ptr(dw0), None, ptr(ws), M, Ki, No, st)

        parts = _lib.query("mny_pj_bwd_parts", M, Ki, No)
        gd = torch.empty(M, Ki, device=dev)
        dw = torch.zeros(No, Ki, device=dev)
        dws = torch.zeros(parts * No * Ki, device=dev)
        red = torch.zeros(parts, 2, Ki, device=dev)

        def one_pass():
            _lib.call("mny_pj_bwd", ptr(G), ptr(Y), ptr(coef), ptr(D), ptr(dsc), ptr(dsh), ptr(dmu), ptr(dis), act, ptr(W), ptr(gd), ptr(dw), ptr(dws), ptr(red),
                      M, Ki, No, st)

        t3, t1 = timeit(three), timeit(one_pass)
        gb = 4 * M * (2 * No + 2 * Ki) / 1e9
        print("Ki%-3d No%-2d M%-8d: apply + dgrad_bnred + wgrad %.3f ms  |  pj_bwd %.3f ms (%.0f GB/s)   max |dgd| %.2e  max |ddw| %.2e (ref %.2e)" % (
            Ki, No, M, t3, t1, gb / t1 * 1e3, (gd - gd0).abs().max().item(), (dw - dw0).abs().max().item(), dw0.abs().max().item()), flush=True)
        tot3 += count * t3
        tot1 += count * t1
    print("plan share: %.3f ms -> %.3f ms" % (tot3, tot1))


if __name__ == "__main__":
    main()
