#!/usr/bin/env python3
"""Timing of the fused depthwise 3x3 stride-1 backward (csrc/dwbwd.hip) at the shapes of the bs-256 / 352x352 plan.  The knobs of
that file (MNY_DWB_TH, MNY_DWB_RES, MNY_DWB_XCD, MNY_STENCIL_CGB) are read once per process, so run it once per setting:
    python tools/bench_dwbwd.py [bs] [f32|bf16] [K]   prints ms, algorithmic GB/s and fp64 checksums of dX / dW / producer sums per shape
bf16: the shapes of the MobileNetV3-YOLO 512x512 bs-64 plan.  K = 5: the 5x5 stride-1 units of that plan (tile form, csrc/dwtile.hip) next to
the launches they replace (bn_bwd_apply + dw_bwd_weight + dw_bwd_data + the producer's bn_bwd_reduce); K = 3 with MNY_DWT3=1: the tile form on 3x3."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mobilenet_yolo_pytorch_amd import _lib  # noqa: E402

P = ctypes.c_void_p
ptr = lambda t: P(t.data_ptr()) if t is not None else None  # noqa: E731


def timeit(fn, reps):
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
    bf = len(sys.argv) > 2 and sys.argv[2] == "bf16"
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dt = torch.bfloat16 if bf else torch.float32
    sfx = "_bf16" if bf else ""
    shapes = ((16, 256, 1, 1), (72, 128, 0, 1), (120, 64, 1, 2), (200, 32, 1, 1), (184, 32, 1, 2), (480, 32, 1, 1), (672, 32, 1, 1), (960, 16, 1, 2), (160, 32, 0, 2),
              (320, 32, 0, 1), (320, 16, 0, 2)) if bf else ((32, 176, 0, 1), (144, 88, 1, 1), (192, 44, 1, 2), (384, 22, 1, 4), (576, 22, 1, 3), (960, 11, 1, 3))
    if K == 5:
        shapes = ((120, 64, 1, 2), (672, 32, 1, 1), (960, 16, 1, 1)) if bf else ((120, 44, 1, 2), (672, 22, 1, 1), (960, 11, 1, 1))
    KK = K * K
    act = 4 if K == 5 else 1                      # h-swish units (5x5 of MobileNetV3's tail) / ReLU6
    dev = torch.device("cuda:0")
    st = P(torch.cuda.current_stream().cuda_stream)
    total = 0.0
    for C, H, red, count in shapes:
        N, W = bs, H
        torch.manual_seed(C)
        g, y, x = (torch.randn(N, H, W, C, device=dev).to(dt) for _ in range(3))
        add = torch.randn(N, H, W, C, device=dev).to(dt)
        mk = lambda a, b: (a + b * torch.randn(C, device=dev))  # noqa: E731
        scale, shift = mk(1.0, 0.2), mk(0.0, 0.3)
        coef = torch.stack((mk(1.0, 0.2), mk(0.0, 0.05), mk(0.0, 0.05))).contiguous()
        xs, xh, xm, xi = mk(1.0, 0.2), mk(0.0, 0.3), mk(0.0, 0.2), mk(1.0, 0.1).abs()
        w = torch.randn(C, K, K, device=dev) * 0.4
        parts0 = _lib.query("mny_dw_bnbwd_parts_k", N, H, W, C, K, 1 if bf else 0)
        parts = _lib.query("mny_dw_bnbwd_parts_k", N, H, W, C, K, (1 if bf else 0) | 2)        # with producer sums (the form may differ)
        ws = torch.zeros(max(parts, parts0) * C * KK, device=dev)
        inred = torch.zeros(parts * 2 * C, device=dev)
        dx, dw = torch.empty_like(x), torch.zeros(C, K, K, device=dev)

        def plain():
            _lib.call("mny_dw_bnbwd" + sfx, ptr(g), ptr(y), ptr(scale), ptr(shift), act, ptr(coef), ptr(x), ptr(xs), ptr(xh), act, ptr(w), ptr(add), ptr(dx), ptr(dw),
                      ptr(ws), N, H, W, C, K, 1, st)

        def withred():
            _lib.call("mny_dw_bnbwd_red" + sfx, ptr(g), ptr(y), ptr(scale), ptr(shift), act, ptr(coef), ptr(x), ptr(xs), ptr(xh), act, ptr(xm), ptr(xi), ptr(w), ptr(add),
                      ptr(dx), ptr(dw), ptr(ws), ptr(inred), N, H, W, C, K, 1, st)

        t0 = timeit(plain, 10)
        cs = (dx.double().sum().item(), dx.double().abs().sum().item(), (dw.double() * torch.arange(KK, device=dev).view(1, K, K)).sum().item())
        t1 = timeit(withred, 10)
        rs = inred.view(parts, 2, C).double().sum(0)
        gb = 5 * N * H * W * C * (2 if bf else 4) / 1e9
        un = ""
        if K == 5:
            # the launches the tile form replaces (the engine's un-fused route): apply -> weight + data gradient -> the producer's reduce
            M = N * H * W
            dy = torch.empty_like(g)
            wparts = _lib.query("mny_dw_wgrad_parts", N, H, W, C, K, 1)
            wws = torch.zeros(max(wparts, 1) * C * KK, device=dev)
            dw2, dx2 = torch.zeros(C, K, K, device=dev), torch.empty_like(x)
            rparts = _lib.query("mny_bn_bwd_parts", M, C)
            rws = torch.zeros(rparts * 2 * C, device=dev)

            def unfused():
                _lib.call("mny_bn_bwd_apply" + sfx, ptr(g), ptr(y), ptr(scale), ptr(shift), act, ptr(coef), ptr(dy), M, C, st)
                _lib.call("mny_dw_bwd_weight" + sfx, ptr(x), ptr(xs), ptr(xh), act, ptr(dy), ptr(dw2), ptr(wws), N, H, W, C, K, 1, st)
                _lib.call("mny_dw_bwd_data" + sfx, ptr(dy), ptr(w), ptr(add), ptr(dx2), N, H, W, C, K, 1, st)
                _lib.call("mny_bn_bwd_reduce" + sfx, ptr(dx2), ptr(x), ptr(xs), ptr(xh), act, ptr(xm), ptr(xi), ptr(rws), M, C, st)

            t2 = timeit(unfused, 10)
            ddx = (dx.float() - dx2.float()).abs().max().item() / max(dx2.float().abs().max().item(), 1e-30)
            ddw = (dw - dw2).abs().max().item() / max(dw2.abs().max().item(), 1e-30)
            rr = rws.view(rparts, 2, C).double().sum(0)
            drd = ((rs - rr).abs().max() / rr.abs().max()).item()
            un = "  | un-fused chain %.3f ms; rel. diff dx %.2e dw %.2e sums %.2e" % (t2, ddx, ddw, drd)
        print("C%-4d %3dx%-3d parts %4d/%4d: plain %.3f ms (%.0f GB/s)  with producer sums %.3f ms (%.0f GB/s) | dx %.9e %.9e dw %.9e red %.9e %.9e" % (
            C, H, W, parts0, parts, t0, gb / t0 * 1e3, t1, gb / t1 * 1e3, cs[0], cs[1], cs[2], rs[0].sum().item(), rs[1].sum().item()) + un, flush=True)
        total += count * (t1 if red else t0)
    print("step share (plan's launch counts): %.3f ms" % total)


if __name__ == "__main__":
    main()
