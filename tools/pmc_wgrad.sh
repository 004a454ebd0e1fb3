# HBM read traffic of the weight-gradient kernels with and without the XCD-aware block order (rocprofv3 --pmc FETCH_SIZE, own runs)
set -e
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/pmc_wg_xcd -o run -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nms > /dev/null 2> $REPO/gpurun_out/pmc_wg_xcd.err
export MNY_WGRAD_NO_XCD=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/pmc_wg_plain -o run -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nms > /dev/null 2> $REPO/gpurun_out/pmc_wg_plain.err
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/"
for tag in ("pmc_wg_xcd", "pmc_wg_plain"):
    f = glob.glob(R + tag + "/**/*counter_collection.csv", recursive=True)[0]
    tot = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "wgrad" in r["Kernel_Name"]:
            tot[r["Kernel_Name"].split("(")[0][-60:]] += float(r["Counter_Value"])
    print(tag, "FETCH_SIZE x2 GB per step over the wgrad kernels: %.2f" % (sum(tot.values()) * 1024 * 2 / 6 / 1e9))
    for k, v in tot.most_common(8): print("   %-60s %.2f" % (k, v * 1024 * 2 / 6 / 1e9))
PY
