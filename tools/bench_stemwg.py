#!/usr/bin/env python3
"""Timing of the stem weight-gradient kernels (csrc/stem.hip) at the stem shapes of the two benchmark plans.
    python tools/bench_stemwg.py            # matrix-core form (stem_wgrad_mfma_kernel)
    MNY_STEM_WGRAD_VALU=1 python tools/bench_stemwg.py   # vector-ALU form (stem_tile_kernel MODE 2)
Prints ms, algorithmic GB/s and the distance of the result to an fp64 reference (torch conv2d weight gradient of the rebuilt dY)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mobilenet_yolo_pytorch_amd import _lib  # noqa: E402

P = ctypes.c_void_p
ptr = lambda t: P(t.data_ptr()) if t is not None else None  # noqa: E731


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
    st = P(torch.cuda.current_stream().cuda_stream)
    for name, N, H, Co, act, dt in (("mbv3 512^2 bs64 bf16", 64, 512, 16, 4, torch.bfloat16), ("mbv3 512^2 bs64 f32", 64, 512, 16, 4, torch.float32),
                                    ("mbv2 352^2 bs256 f32", 256, 352, 32, 1, torch.float32)):
        torch.manual_seed(1)
        sfx = "_bf16" if dt == torch.bfloat16 else ""
        x = torch.randn(N, 3, H, H, device=dev)
        Ho = H // 2
        y = torch.randn(N, Ho, Ho, Co, device=dev).to(dt)
        g = torch.randn(N, Ho, Ho, Co, device=dev).to(dt)
        scale, shift = torch.rand(Co, device=dev) + 0.5, torch.randn(Co, device=dev) * 0.2
        coef = torch.stack([torch.rand(Co, device=dev) + 0.5, torch.randn(Co, device=dev) * 0.05, torch.randn(Co, device=dev) * 0.05]).contiguous()
        parts = _lib.query("mny_stem_wgrad_parts", N, H, H, Co)
        ws = torch.zeros(parts, Co * 27, device=dev)
        dw = torch.zeros(Co, 3, 3, 3, device=dev)
        fn = lambda: _lib.call("mny_stem_bnwgrad" + sfx, ptr(x), ptr(g), ptr(y), ptr(scale), ptr(shift), act, ptr(coef), ptr(dw), ptr(ws), N, H, H, Co, st)  # noqa: E731
        ms = timeit(fn)
        gb = (x.numel() * 4 + (g.numel() + y.numel()) * g.element_size()) / 1e9
        # fp64 reference on a slice of the batch (the whole batch for the error would need 3 GB of fp64)
        nb = min(N, 8)
        xs, ys, gs = x[:nb].double(), y[:nb].double(), g[:nb].double()
        z = ys * scale.double() + shift.double()
        if act == 4:
            d = torch.where(z <= -3, torch.zeros_like(z), torch.where(z >= 3, torch.ones_like(z), (2 * z + 3) / 6))
        else:
            d = ((z > 0) & (z < 6)).double()
        dY = coef[0].double() * gs * d + coef[1].double() * ys + coef[2].double()
        wref = torch.nn.grad.conv2d_weight(xs, (Co, 3, 3, 3), dY.permute(0, 3, 1, 2).contiguous(), stride=2, padding=1)
        ws2 = torch.zeros(_lib.query("mny_stem_wgrad_parts", nb, H, H, Co), Co * 27, device=dev)
        dw2 = torch.zeros(Co, 3, 3, 3, device=dev)
        _lib.call("mny_stem_bnwgrad" + sfx, ptr(x[:nb].contiguous()), ptr(g[:nb].contiguous()), ptr(y[:nb].contiguous()), ptr(scale), ptr(shift), act, ptr(coef),
                  ptr(dw2), ptr(ws2), nb, H, H, Co, st)
        torch.cuda.synchronize()
        err = ((dw2.double() - wref).abs().max() / wref.abs().max()).item()
        print(f"{name:24s} parts {parts:5d}  {ms:7.4f} ms  {gb / ms * 1e3:7.0f} GB/s   rel.err vs fp64 (first {nb} images) {err:.2e}   sum {dw.double().sum().item():.6f}")


if __name__ == "__main__":
    main()
