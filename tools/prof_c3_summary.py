"""configs[3] (MobileNetV3-YOLO 512x512 bs 64 bf16): per-kernel time + HBM counters of the whole step -> profiles/<tag>_c3_*.

usage: python tools/prof_c3_summary.py <stats_dir> <fetch_dir> <write_dir> <steps_stats> <steps_pmc> <tag>

Counter calibration (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts a 128-B request as 64 B for 16-B/lane streams (x2); other
widths and WRITE_SIZE are uncalibrated -> calibrated HERE, on this run's own bn_bwd_apply_kernel<bf16> launches (a pure stream with a
known byte count: reads G and Y, writes dY, 8 B per lane): read factor = known read bytes / FETCH_SIZE, write factor = known write
bytes / WRITE_SIZE.  The LDS-DMA GEMM / weight-gradient kernels (16 B per lane) take the documented x2, everything else the
calibrated 8-B/lane factor; writes take the calibrated write factor everywhere.
"""
import csv
import json
import re
import shutil
import sys
from collections import defaultdict


def family(name):
    n = re.sub(r"^void ", "", name)
    n = re.sub(r"^mny::", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n.replace("float", "f32").replace("mny::bf16_t", "bf16")[:64]


def pmc(dirname, counter):
    tot = defaultdict(float)
    with open(dirname + "/run_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                tot[family(r["Kernel_Name"])] += float(r["Counter_Value"])
    return tot


def main():
    stats_dir, fetch_dir, write_dir, steps_stats, steps_pmc, tag = sys.argv[1:7]
    steps_stats, steps_pmc = int(steps_stats), int(steps_pmc)
    shutil.copy(stats_dir + "/run_kernel_stats.csv", "profiles/%s_c3_rocprofv3_kernel_stats.csv" % tag)
    line = json.loads([ln for ln in open(stats_dir + "/../prof_c3.json") if ln.startswith("{")][-1])
    dur, calls = defaultdict(float), defaultdict(int)
    with open(stats_dir + "/run_kernel_stats.csv") as f:
        for r in csv.DictReader(f):
            k = family(r["Name"])
            dur[k] += float(r["TotalDurationNs"])
            calls[k] += int(r["Calls"])
    fe, wr = pmc(fetch_dir, "FETCH_SIZE"), pmc(write_dir, "WRITE_SIZE")
    KIB = 1024.0
    # calibration on the apply kernel: algorithmic bytes = 3 passes (2 read, 1 written) of its tensors
    apply_k = [k for k in dur if k.startswith("bn_bwd_apply_kernel<bf16")]
    alg = line["algorithmic_bytes_per_step"].get("mny_bn_bwd_apply_bf16", 0)
    f_raw = sum(fe[k] for k in apply_k) * KIB / steps_pmc
    w_raw = sum(wr[k] for k in apply_k) * KIB / steps_pmc
    rf = (alg * 2 / 3) / f_raw if f_raw else 2.0
    wf = (alg / 3) / w_raw if w_raw else 1.0
    dma = lambda k: k.startswith("pw_gemm_nt_dma_kernel") or k.startswith("pw_wgrad_bf16_kernel") or k.startswith("pw_wgrad_dma_kernel")   # noqa: E731
    rows, tot_ms, tot_b = [], 0.0, 0.0
    for k in dur:
        ms = dur[k] / steps_stats / 1e6
        fb = fe.get(k, 0.0) * KIB / steps_pmc * (2.0 if dma(k) else rf)
        wb = wr.get(k, 0.0) * KIB / steps_pmc * wf
        rows.append((ms, k, calls[k] / steps_stats, fb, wb))
        tot_ms += ms
        tot_b += fb + wb
    rows.sort(reverse=True)
    with open("profiles/%s_c3_kernels_time_and_hbm.md" % tag, "w") as o:
        o.write("# configs[3] — MobileNetV3-YOLO 512x512, bs 64, bf16 storage: per-kernel time and HBM traffic per training step\n\n")
        o.write("`python3 bench.py --arch mbv3 --size 512 --batch 64 --dtype bf16 --no-cpu-baseline --no-nms` under rocprofv3: `--kernel-trace --stats` "
                "(%d steps), `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (%d steps each, separate runs; `tools/prof_c3.sh`).\n\n" % (steps_stats, steps_pmc))
        o.write("Calibration on this run's own `bn_bwd_apply_kernel<bf16>` launches (a pure 8-B/lane stream, %.3f GB algorithmic per step): "
                "read bytes = FETCH_SIZE x %.3f, written bytes = WRITE_SIZE x %.3f; the LDS-DMA GEMM / weight-gradient kernels (16 B per lane) take the "
                "guide's x2 on FETCH_SIZE.\n\n" % (alg / 1e9, rf, wf))
        o.write("| kernel | launches/step | ms/step | avg us | read GB | written GB | GB/s |\n|---|---|---|---|---|---|---|\n")
        for ms, k, n, fb, wb in rows:
            if ms < 0.04:
                continue
            o.write("| `%s` | %.0f | %.3f | %.1f | %.3f | %.3f | %.0f |\n" % (k, n, ms, ms * 1e3 / max(n, 1), fb / 1e9, wb / 1e9, (fb + wb) / ms / 1e6 if ms else 0))
        o.write("\nTotal per step: %.0f kernel launches, kernel time %.2f ms, HBM traffic %.2f GB -> %.2f TB/s average over the kernel time "
                "(bench line of the same run: %.2f ms/step, %.0f images/s).\n" % (sum(r[2] for r in rows), tot_ms, tot_b / 1e9, tot_b / tot_ms / 1e9,
                                                                             line["ms_per_step"], line["value"]))
    sig = line["plan_signatures"]["__step__"]
    with open("profiles/%s_traffic_config3.json" % tag, "w") as o:
        json.dump({"config": "configs[3]: MobileNetV3-YOLO 512x512 bs 64 bf16 storage, whole training step", "calls_per_step": sig[0], "step_signature": sig[1],
                   "kernel_launches_per_step_rocprof": sum(r[2] for r in rows), "kernel_ms_per_step_rocprof": tot_ms, "hbm_bytes_per_step": tot_b,
                   "read_factor_8B_per_lane": rf, "write_factor": wf,
                   "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs, counter unit KiB) summed over every kernel of a step; "
                             "calibrated on the run's own bn_bwd_apply_kernel<bf16> launches (known byte count), x2 on FETCH_SIZE for the 16-B/lane LDS-DMA kernels"},
                  o, indent=1)
    print("c3: %.2f ms kernel time, %.2f GB/step, read factor %.3f write factor %.3f" % (tot_ms, tot_b / 1e9, rf, wf))


if __name__ == "__main__":
    main()
