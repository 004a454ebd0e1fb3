#!/usr/bin/env python3
"""Same-box timing of the expand + depthwise unit (csrc/exdw.hip) against the materialised kernels it replaces, at the three
stride-2 shapes of the bs-256 / 352x352 plan.  usage: python tools/bench_exdw.py [fwd|bwd|all] [bs] [K]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mobilenet_yolo_pytorch_amd import _lib  # noqa: E402

P = ctypes.c_void_p
ptr = lambda t: P(t.data_ptr()) if t is not None else None  # noqa: E731


def timeit(fn, reps=int(os.environ.get('EXDW_REPS', '10'))):
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
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = torch.device("cuda:0")
    st = P(torch.cuda.current_stream().cuda_stream)
    only_k = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    for K, H in ((16, 176), (24, 88), (32, 44)):
        if only_k and K != only_k:
            continue
        N, W, C = bs, H, 6 * K
        M = N * H * W
        Ho, Wo = H // 2, W // 2
        torch.manual_seed(K)
        x = torch.randn(N, H, W, K, device=dev)
        isc, ish = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.2
        w = torch.randn(C, K, device=dev) * K ** -0.5
        wd = torch.randn(C, 3, 3, device=dev) * 0.4
        gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.5 + 1.0
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        coef = torch.zeros(4, C, device=dev)
        y = torch.empty(N, H, W, C, device=dev)
        z = torch.empty(N, Ho, Wo, C, device=dev)
        z2 = torch.empty_like(z)
        stats = torch.zeros(1024 * 2 * C, device=dev)
        pparts = _lib.query("mny_pw_stat_parts", M, K, C)
        sparts = _lib.query("mny_exdw_stat_parts", M, K, C)

        def fin(parts):
            _lib.call("mny_bn_finalize", ptr(stats), parts, M, ptr(gam), ptr(bet), 1e-5, 0.1, ptr(rm), ptr(rv), ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), C, st)

        def pw():
            _lib.call("mny_pw_fwd", ptr(x), ptr(isc), ptr(ish), 0, ptr(w), None, None, ptr(y), ptr(stats), M, K, C, st)

        def dw():
            _lib.call("mny_dw_fwd", ptr(y), ptr(coef[0]), ptr(coef[1]), 1, ptr(wd), ptr(z), ptr(stats), N, H, W, C, 3, 2, st)

        def xs():
            _lib.call("mny_exdw_stats", ptr(x), ptr(isc), ptr(ish), 0, ptr(w), ptr(stats), M, K, C, st)

        def xf():
            _lib.call("mny_exdw_fwd", ptr(x), ptr(isc), ptr(ish), 0, ptr(w), ptr(coef[0]), ptr(coef[1]), ptr(wd), ptr(z2), ptr(stats), N, H, W, K, C, 2, st)

        pw(); fin(pparts)
        if what in ("fwd", "all"):
            t = [timeit(f) for f in (pw, dw, xs, xf)]
            print("K%d H%d fwd: pw_fwd %.3f + dw_fwd %.3f = %.3f ms   |   exdw_stats %.3f + exdw_fwd %.3f = %.3f ms" % (
                K, H, t[0], t[1], t[0] + t[1], t[2], t[3], t[2] + t[3]), flush=True)
        if what in ("bwd", "all"):
            dw()
            zparts = _lib.query("mny_dw_stat_parts", N, H, W, C, 3, 2)
            zc = torch.zeros(4, C, device=dev)
            zg, zb = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
            _lib.call("mny_bn_finalize", ptr(stats), zparts, N * Ho * Wo, ptr(zg), ptr(zb), 1e-5, 0.1, ptr(rm), ptr(rv), ptr(zc[0]), ptr(zc[1]), ptr(zc[2]), ptr(zc[3]), C, st)
            gz = torch.randn(N, Ho, Wo, C, device=dev)
            red = torch.zeros(2048 * 2 * C, device=dev)
            rparts = _lib.query("mny_bn_bwd_parts", N * Ho * Wo, C)
            zcoef = torch.zeros(3, C, device=dev)
            dgz, dbz = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            _lib.call("mny_bn_bwd_reduce", ptr(gz), ptr(z), ptr(zc[0]), ptr(zc[1]), 1, ptr(zc[2]), ptr(zc[3]), ptr(red), N * Ho * Wo, C, st)
            _lib.call("mny_bn_bwd_finalize", ptr(red), rparts, N * Ho * Wo, ptr(zg), ptr(zc[2]), ptr(zc[3]), ptr(dgz), ptr(dbz), ptr(zcoef), C, st)
            gy = torch.empty(N, H, W, C, device=dev)
            dwd = torch.zeros(C, 3, 3, device=dev)
            ws1 = torch.zeros(max(1024 * C * 9, _lib.query("mny_pw_bnbwd_ws_floats", M, K, C)), device=dev)
            dx = torch.empty(N, H, W, K, device=dev)
            dwe, dge, dbe = torch.zeros(C, K, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)

            def dwb():
                _lib.call("mny_dw_bnbwd_s2", ptr(gz), ptr(z), ptr(zc[0]), ptr(zc[1]), 1, ptr(zcoef), ptr(y), ptr(coef[0]), ptr(coef[1]), 1, ptr(wd),
                          None, ptr(gy), ptr(dwd), ptr(ws1), N, H, W, C, st)

            def pwb():
                _lib.call("mny_pw_bnbwd", ptr(gy), ptr(y), ptr(coef[0]), ptr(coef[1]), 1, ptr(coef[2]), ptr(coef[3]), ptr(gam), ptr(x), ptr(isc), ptr(ish), 0,
                          ptr(w), None, ptr(dx), ptr(dwe), ptr(dge), ptr(dbe), ptr(ws1), M, K, C, st)

            ws2 = torch.zeros(max(int(_lib.query("mny_exdw_bwd_ws_floats", N, H, W, K, C, 2)), 1), device=dev)
            dws = torch.zeros(_lib.query("mny_exdw_bwd_parts", N, H, W, K, C, 2) * C * 9, device=dev)
            dx2 = torch.empty_like(dx)
            dwd2, dwe2, dge2, dbe2 = torch.zeros_like(dwd), torch.zeros_like(dwe), torch.zeros_like(dge), torch.zeros_like(dbe)

            def xb():
                _lib.call("mny_exdw_bwd", ptr(gz), ptr(z), ptr(zc[0]), ptr(zc[1]), 1, ptr(zcoef), ptr(x), ptr(isc), ptr(ish), 0, ptr(w),
                          ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), ptr(gam), ptr(wd), None, ptr(dx2), ptr(dwe2), ptr(dge2), ptr(dbe2),
                          ptr(dwd2), ptr(dws), ptr(ws2), N, H, W, K, C, 2, st)

            t = [timeit(f) for f in (dwb, pwb, xb)]
            print("K%d H%d bwd: dw_bnbwd_s2 %.3f + pw_bnbwd %.3f = %.3f ms   |   exdw_bwd %.3f ms" % (K, H, t[0], t[1], t[0] + t[1], t[2]), flush=True)
            for name, a, b in (("dx", dx, dx2), ("dw_exp", dwe, dwe2), ("dgamma", dge, dge2), ("dbeta", dbe, dbe2), ("dw_dw", dwd, dwd2)):
                d = (a.double() - b.double()).abs().max().item()
                print("    %-7s max |materialised - fused| = %.3e (max |ref| %.3e)" % (name, d, a.abs().max().item()))


if __name__ == "__main__":
    main()
