"""MFMA-pipe utilisation of the GEMM kernels from rocprofv3 PMC counters (tools/prof_round.sh, 4th run).

usage: python tools/prof_mfma.py <mfma_dir> <steps> <tag>
SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (profiles/r01_pmc_nt_gemm_512.md), so
    MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)
is the fraction of SIMD-cycles the matrix pipe was executing an MFMA while the kernel ran.  Per kernel family and per
(family, grid size) = per layer shape class; kernels without MFMA work are omitted.
"""
import csv
import sys
from collections import defaultdict

sys.path.insert(0, "tools")
from prof_summary import family  # noqa: E402


def main():
    d, steps, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    busy, act, n = defaultdict(float), defaultdict(float), defaultdict(int)
    with open(d + "/run_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            k = (family(r["Kernel_Name"]), int(r["Grid_Size"]))
            v = float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                busy[k] += v
                n[k] += 1
            elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                act[k] += v
    fam_b, fam_a, fam_n = defaultdict(float), defaultdict(float), defaultdict(int)
    for k in busy:
        fam_b[k[0]] += busy[k]; fam_a[k[0]] += act[k]; fam_n[k[0]] += n[k]
    with open("profiles/%s_pmc_mfma_busy.md" % tag, "w") as o:
        o.write("# MFMA-pipe busy fraction of the GEMM kernels (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE)\n\n"
                "`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nms` (bs=256, 352x352, fp32), counters in their own run.  "
                "busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).  The fp32 `v_mfma_f32_32x32x2_f32` "
                "occupies the pipe 64 cycles per issue, so busy == achieved / peak for a kernel that does nothing but MFMA.\n\n")
        o.write("## per kernel family\n\n| kernel | launches/step | MFMA busy |\n|---|---|---|\n")
        for k in sorted(fam_b, key=lambda k: -fam_b[k]):
            if fam_b[k] <= 0:
                continue
            o.write("| `%s` | %.1f | %.1f %% |\n" % (k, fam_n[k] / steps, 100 * fam_b[k] / (128 * fam_a[k])))
        o.write("\n## per (kernel, grid size) = per layer-shape class\n\n| kernel | grid (threads) | launches/step | MFMA busy |\n|---|---|---|---|\n")
        for k in sorted(busy, key=lambda k: -busy[k]):
            if busy[k] <= 0:
                continue
            o.write("| `%s` | %d | %.1f | %.1f %% |\n" % (k[0], k[1], n[k] / steps, 100 * busy[k] / (128 * act[k])))
        tot_b = sum(fam_b[k] for k in fam_b if k.startswith("pw_gemm_nt") or k.startswith("pw_wgrad") or k.startswith("pw_bnbwd"))
        tot_a = sum(fam_a[k] for k in fam_a if fam_b[k] > 0 and (k.startswith("pw_gemm_nt") or k.startswith("pw_wgrad") or k.startswith("pw_bnbwd")))
        o.write("\nAll pointwise GEMM kernels together (forward, data gradient, weight gradient, fused BN-backward units): "
                "**%.1f %% MFMA busy** over their own run time.\n" % (100 * tot_b / (128 * tot_a)))
    print(open("profiles/%s_pmc_mfma_busy.md" % tag).read())


if __name__ == "__main__":
    main()
