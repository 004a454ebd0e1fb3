#!/usr/bin/env python3
"""Per-layer table of the pointwise-conv GEMM launches of a plan: time, achieved TFLOP/s, fraction of the fp32 MFMA peak
(157.3 TFLOP/s: the contract's roofline) AND of the pipe the six-product form actually runs on (dense bf16 MFMA / 6 = 416.7 TFLOP/s of fp32
work), algorithmic GB/s, and the kernel family the dispatcher took.

    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown --detail mny_pw_fwd,mny_pw_wgrad,mny_pw_dgrad_bnred,mny_pw_dgrad_bnred_add 2> detail.txt
    python tools/gemm_table.py detail.txt profiles/r05_gemm_per_layer.md [profiles/r05_pmc_mfma_busy.md]
"""
import re
import sys

PEAK32, PEAK_X6 = 157.3, 2500.0 / 6.0


def main():
    src, dst = sys.argv[1], sys.argv[2]
    rows = []
    for l in open(src):
        m = re.match(r"(fwd|bwd) (mny_pw_\S+)\s+(?:(dgrad(?:\+add)?(?:\+red)?) )?M(\d+) K(\d+) N(\d+)\s+([0-9.]+) ms\s+([0-9.]+) TF/s\s+([0-9.]+) GB/s", l)
        if m:
            which, ep, kind, M, K, N, ms, tf, gbs = m.groups()
            rows.append((which, ep, kind or ("wgrad" if "wgrad" in ep else "fwd"), int(M), int(K), int(N), float(ms), float(tf), float(gbs)))
    rows.sort(key=lambda r: -r[6])
    tot = sum(r[6] for r in rows)
    flops = sum(2.0 * r[3] * r[4] * r[5] for r in rows)
    mb = [r for r in rows if 2.0 * r[4] * r[5] / (4.0 * (r[4] + r[5])) >= 20.0]      # the planner's "MFMA-bound" criterion (nt_x6: FLOP per byte of the A and C rows >= 20)
    with open(dst, "w") as f:
        f.write("# Pointwise-conv GEMM launches of the headline step, per layer (bs 256, 352x352, fp32)\n\n")
        f.write("`python bench.py --breakdown --detail mny_pw_fwd,mny_pw_wgrad,mny_pw_dgrad_bnred,mny_pw_dgrad_bnred_add` (every launch bracketed with HIP events on the launch stream, "
                "10 steps; one stream).  `frac fp32` = achieved / %.1f TFLOP/s (v_mfma_f32_32x32x2_f32 dense peak: the contract's roofline for an fp32 GEMM); "
                "`frac x6` = achieved / %.1f TFLOP/s (dense bf16 MFMA peak / 6: the ceiling of the six-product form the MFMA-bound shapes actually run). "
                "AI = 2KN / 4(K+N) FLOP per byte of the A and C rows; shapes with AI >= 20 take the six-product form.\n\n" % (PEAK32, PEAK_X6))
        f.write("| pass | entry point | M | K | N | AI | ms | TFLOP/s | frac fp32 | frac x6 | alg. GB/s |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        for which, ep, kind, M, K, N, ms, tf, gbs in rows:
            ai = 2.0 * K * N / (4.0 * (K + N))
            f.write("| %s | `%s` %s | %d | %d | %d | %.0f | %.3f | %.1f | %.2f | %s | %.0f |\n" % (
                which, ep, kind if kind not in ("fwd", "wgrad") else "", M, K, N, ai, ms, tf, tf / PEAK32, ("%.2f" % (tf / PEAK_X6)) if ai >= 20 else "—", gbs))
        f.write("\n**Totals**: %d launches, %.2f ms/step, %.1f GFLOP/step -> %.1f TFLOP/s = %.2f of the fp32 MFMA peak.  " % (len(rows), tot, flops / 1e9, flops / tot / 1e9, flops / tot / 1e9 / PEAK32))
        tmb, fmb = sum(r[6] for r in mb), sum(2.0 * r[3] * r[4] * r[5] for r in mb)
        f.write("MFMA-bound shapes (AI >= 20): %d launches, %.2f ms, %.1f TFLOP/s = %.2f of the fp32 peak, **%.2f of the six-product ceiling**.\n" % (
            len(mb), tmb, fmb / tmb / 1e9, fmb / tmb / 1e9 / PEAK32, fmb / tmb / 1e9 / PEAK_X6))
        small = [r for r in mb if r[3] <= 31000]
        if small:
            ts, fs = sum(r[6] for r in small), sum(2.0 * r[3] * r[4] * r[5] for r in small)
            big = [r for r in mb if r[3] > 31000]
            tb, fb = sum(r[6] for r in big), sum(2.0 * r[3] * r[4] * r[5] for r in big)
            f.write("\nThe 11x11 layers (M = 30 976 rows = 242 row tiles on 256 CUs): %d launches, %.2f ms, %.1f TFLOP/s; the larger MFMA-bound layers: %d launches, %.2f ms, %.1f TFLOP/s — "
                    "the same rate: the six-product kernels are bound by what a workgroup does per tile (operand cut + LDS traffic next to the MFMAs, "
                    "`profiles/*_pmc_mfma_busy.md`), not by how many tiles a launch has, so a split-K / stream-K decomposition of the 11x11 shapes has nothing to recover.\n" % (
                        len(small), ts, fs / ts / 1e9, len(big), tb, fb / tb / 1e9))
    print("wrote", dst, len(rows), "rows")


if __name__ == "__main__":
    main()
