# kernel-trace of configs[3] (single stream, un-bracketed): start/end timestamps -> gaps between consecutive kernels
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export MNY_SIDE_STREAM=0
rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/r5/gaps -o run -- python3 $REPO/bench.py --arch mbv3 --size 512 --batch 64 --dtype bf16 --no-cpu-baseline --no-nms --steps 6 --warmup 3 --roofline-pass after > $REPO/gpurun_out/r5/gaps.json 2> $REPO/gpurun_out/r5/gaps.err
ls $REPO/gpurun_out/r5/gaps
python3 - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5/gaps/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# take the last 3000 kernels (steady state)
rows = rows[-3400:]
gaps = []; dur = 0
by = collections.defaultdict(lambda: [0, 0, 0])
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = s1 - e0
    gaps.append(g); dur += e0 - s0
    k = n1.split("<")[0].split("(")[0][-40:]
    by[k][0] += 1; by[k][1] += g; by[k][2] += e1 - s1
span = rows[-1][1] - rows[0][0]
print("kernels %d span %.3f ms, sum durations %.3f ms, sum gaps %.3f ms (positive only %.3f), median gap %.2f us" % (
    len(rows), span / 1e6, dur / 1e6, sum(gaps) / 1e6, sum(g for g in gaps if g > 0) / 1e6, sorted(gaps)[len(gaps) // 2] / 1e3))
import statistics
small = [g for g in gaps if 0 < g < 50000]
print("gap percentiles us: p10 %.2f p50 %.2f p90 %.2f" % tuple(sorted(small)[int(len(small) * p)] / 1e3 for p in (0.1, 0.5, 0.9)))
for k, (n, g, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-42s n=%4d gap-before avg %.2f us, dur avg %.2f us" % (k, n, g / n / 1e3, d / n / 1e3))
PY
