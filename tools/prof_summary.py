"""Summarise rocprofv3 CSV output (kernel stats + separate FETCH_SIZE / WRITE_SIZE PMC passes) into profiles/.

usage: python tools/prof_summary.py <stats_dir> <fetch_dir> <write_dir> <steps_stats> <steps_pmc> <tag>
  *_dir: rocprofv3 -d directories (with run_kernel_stats.csv / run_counter_collection.csv)
  steps_*: total bench steps (warmup + timed) of the run, to normalise per step
Counter unit is KiB; on gfx950 FETCH_SIZE counts the 128-B requests of wide coalesced streams as 64 B (MI355X_MICROARCH.md,
HBM section) -> `fetch x2` is the corrected read volume for 16-B/lane streams.
"""
import csv
import re
import shutil
import sys
from collections import defaultdict


def family(name):
    n = re.sub(r"^void ", "", name)
    n = re.sub(r"^mny::", "", n)
    n = re.sub(r"\(.*$", "", n)
    n = n.replace("float", "f32").replace("mny::bf16_t", "bf16")
    return n[:64]


def thin_red(k):
    """RED template argument (4th) of a pw_thin_kernel<T, K, XF, RED, R, STATS, ADD> family name, or None."""
    m = re.match(r"pw_thin_kernel<[^,]+, \d+, \d+, (\d+),", k)
    return int(m.group(1)) if m else None


def nt_args(k):
    """template arguments <TN, XF, BF, RED[, X6]> of a pw_gemm_nt_dma_kernel family name, or None"""
    m = re.match(r"pw_gemm_nt_dma_kernel<([^>]*)>", k)
    if not m:
        return None
    a = [int(v) for v in m.group(1).split(",")]
    return a + [0] * (5 - len(a))


def wide_mode(k):
    """MODE template argument (4th) of a pw_wide_kernel<KS, TNB, XF, MODE> family name (0/1 forward, 2/3 data gradient + BN sums), or None."""
    m = re.match(r"pw_wide_kernel<\d+, \d+, \d+, (\d+)>", k)
    return int(m.group(1)) if m else None


def is_pw_fwd(k):
    """kernels behind the mny_pw_fwd entry point: the tile GEMMs without a BN-backward epilogue (fp32-MFMA or six-product bf16 form) +
    the register-staged fallback + the short-reduction stream kernel (RED = 0)"""
    a = nt_args(k)
    return (a is not None and a[3] == 0) or (k.startswith("pw_gemm_nt_kernel")) or thin_red(k) == 0 or wide_mode(k) in (0, 1)


def is_pw_dgrad_bnred(k):
    a = nt_args(k)
    return (a is not None and a[3] == 1) or thin_red(k) == 1 or wide_mode(k) == 2


def pmc(dirname, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(dirname + "/run_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = family(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


def main():
    stats_dir, fetch_dir, write_dir, steps_stats, steps_pmc, tag = sys.argv[1:7]
    steps_stats, steps_pmc = int(steps_stats), int(steps_pmc)
    shutil.copy(stats_dir + "/run_kernel_stats.csv", "profiles/%s_rocprofv3_kernel_stats.csv" % tag)
    dur, calls = {}, {}
    with open(stats_dir + "/run_kernel_stats.csv") as f:
        for r in csv.DictReader(f):
            k = family(r["Name"])
            dur[k] = dur.get(k, 0.0) + float(r["TotalDurationNs"])
            calls[k] = calls.get(k, 0) + int(r["Calls"])
    fe, fc = pmc(fetch_dir, "FETCH_SIZE")
    wr, _ = pmc(write_dir, "WRITE_SIZE")
    rows = []
    for k in dur:
        ms = dur[k] / steps_stats / 1e6
        f_gb = fe.get(k, 0.0) * 1024 / steps_pmc / 1e9
        w_gb = wr.get(k, 0.0) * 1024 / steps_pmc / 1e9
        rows.append((ms, k, calls[k] / steps_stats, f_gb, w_gb))
    rows.sort(reverse=True)
    with open("profiles/%s_kernels_time_and_hbm.md" % tag, "w") as o:
        o.write("# Per-kernel time (rocprofv3 --kernel-trace --stats) and HBM traffic (separate --pmc FETCH_SIZE / WRITE_SIZE passes)\n\n")
        o.write("`python bench.py` bs=256, 352x352, fp32, 1 MI355X; per training step.  Counter unit KiB; `fetch x2` applies the gfx950 "
                "correction for wide coalesced streams (MI355X_MICROARCH.md); WRITE_SIZE is uncalibrated.  "
                "`GB/s` = (fetch x2 + write) / time.\n\n")
        o.write("| kernel | launches/step | ms/step | avg us | FETCH_SIZE GB | fetch x2 GB | WRITE_SIZE GB | GB/s |\n|---|---|---|---|---|---|---|---|\n")
        tms = tf = tw = 0.0
        for ms, k, n, f_gb, w_gb in rows:
            tms += ms; tf += f_gb; tw += w_gb
            if ms < 0.05:
                continue
            o.write("| `%s` | %.0f | %.3f | %.1f | %.2f | %.2f | %.2f | %.0f |\n" % (k, n, ms, ms * 1e3 / max(n, 1), f_gb, 2 * f_gb, w_gb,
                                                                               (2 * f_gb + w_gb) / ms * 1e3 if ms else 0))
        o.write("\nTotal per step: kernel time %.2f ms; FETCH_SIZE %.1f GB (x2 = %.1f GB), WRITE_SIZE %.1f GB -> %.2f TB/s average.\n" % (
            tms, tf, 2 * tf, tw, (2 * tf + tw) / tms))
        # entry-point aggregate bench.py's roofline object must agree with: mny_pw_fwd = every NT GEMM kernel (forward + data gradient)
        # (the RED = 1 instantiations belong to mny_pw_dgrad_bnred: data gradient + BN-backward reduction, a separate entry point)
        g = [(ms, n, f_gb, w_gb) for ms, k, n, f_gb, w_gb in rows if is_pw_fwd(k)]
        g_ms, g_n = sum(r[0] for r in g), sum(r[1] for r in g)
        g_f, g_w = sum(r[2] for r in g), sum(r[3] for r in g)
        o.write("\n`mny_pw_fwd` entry point (all `pw_gemm_nt*`, `pw_thin_kernel*` and `pw_wide_kernel*` kernels except the `RED = 1 / 2` / `MODE = 2 / 3` instantiations of `mny_pw_dgrad_bnred[_add]`): %.0f launches/step, %.3f ms/step, avg %.1f us; HBM %.2f GB/step "
                "(fetch x2 %.2f + write %.2f) = %.0f MB per launch.\n" % (g_n, g_ms, g_ms * 1e3 / max(g_n, 1), 2 * g_f + g_w, 2 * g_f, g_w,
                                                                          (2 * g_f + g_w) * 1e3 / max(g_n, 1)))
    import json
    # per entry point: HBM bytes per launch from the counters + the launch-list fingerprint bench.py printed in the SAME stats run
    # (bench.py refuses the figure when its plan no longer launches exactly these kernels)
    sigs = {}
    try:
        line = [ln for ln in open(stats_dir + "/../prof_stats.json") if ln.startswith("{")][-1]
        sigs = json.loads(line).get("plan_signatures", {})
    except (OSError, IndexError, ValueError):
        print("warning: no bench JSON line next to the stats directory: traffic files carry no plan signature")
    groups = {
        "mny_pw_fwd": is_pw_fwd,
        "mny_pw_dgrad_bnred": is_pw_dgrad_bnred,                                                          # (RED = 2 = mny_pw_dgrad_bnred_add, not priced)
        "mny_pw_wgrad": lambda k: k.startswith("pw_wgrad"),
        "mny_dw_fwd": lambda k: k.startswith("dw3_fwd_kernel") or bool(re.match(r"dw_slide_kernel<f32, \d, \d, 0,", k)),
        # the fused backward of the thin expand units: stage 1 + finalize + stage 2 (the entry point's own partial combine rides along)
        "mny_pw_bnbwd": lambda k: k.startswith("pw_bnbwd_"),
        # the fused depthwise backward (round 4: priced) and the expand + depthwise unit (exdw.hip; its backward = pass 1 + 2; the shared
        # finalize / combine kernels of pwgemm.hip it launches are counted under mny_pw_bnbwd's name filter when that entry point also runs)
        "mny_dw_bnbwd": lambda k: bool(re.match(r"dw_bnbwd_s1k3_kernel<.*, false>$", k)),
        "mny_dw_bnbwd_red": lambda k: bool(re.match(r"dw_bnbwd_s1k3_kernel<.*, true>$", k)),
        "mny_dw_bnbwd_s2": lambda k: k.startswith("dw_bnbwd_s2k3_kernel"),
        "mny_exdw_stats": lambda k: k.startswith("exdw_stats") or k.startswith("exdw_gram_stats"),
        "mny_exdw_fwd": lambda k: k.startswith("exdw_fwd"),
        "mny_exdw_bwd": lambda k: k.startswith("exdw_bwd") or k.startswith("exdw_dxfix"),
        "mny_stemdw_bwd": lambda k: k.startswith("stemdw_bwd"),
        "mny_pj_bwd": lambda k: k.startswith("pj_bwd"),
    }
    for entry, pred in groups.items():
        g = [(ms, n, f_gb, w_gb) for ms, k, n, f_gb, w_gb in rows if pred(k)]
        g_ms, g_n = sum(r[0] for r in g), sum(r[1] for r in g)
        g_f, g_w = sum(r[2] for r in g), sum(r[3] for r in g)
        sig = sigs.get(entry, [None, None])
        if sig[0] is not None and abs(sig[0] - g_n) > 0.01:
            print("warning: %s: %s kernel launches/step in the trace vs %s calls in the plan (helper kernels or a wrong name filter)" % (entry, g_n, sig[0]))
        with open("profiles/%s_traffic_%s.json" % (tag, entry), "w") as o:
            json.dump({"entry_point": entry, "launches_per_step": sig[0] if sig[0] is not None else g_n, "plan_signature": sig[1],
                       "kernel_launches_per_step_rocprof": g_n, "ms_per_step_rocprof": g_ms,
                       "hbm_bytes_per_step": (2 * g_f + g_w) * 1e9, "hbm_bytes_per_launch": (2 * g_f + g_w) * 1e9 / max(sig[0] or g_n, 1),
                       "fetch_size_x2_bytes_per_step": 2 * g_f * 1e9, "write_size_bytes_per_step": g_w * 1e9,
                       "method": "rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE and --pmc WRITE_SIZE (three separate runs, tools/prof_round.sh) of "
                                 "`python3 bench.py --no-cpu-baseline --no-nms`; counter unit KiB; FETCH_SIZE x2 = gfx950 correction for wide coalesced "
                                 "streams (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated"}, o, indent=1)
    print(open("profiles/%s_kernels_time_and_hbm.md" % tag).read())


if __name__ == "__main__":
    main()
