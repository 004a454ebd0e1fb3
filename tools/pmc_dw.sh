# FETCH_SIZE / WRITE_SIZE of the depthwise forward variants (separate --pmc passes, program directly after `--`)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() {  # tag, counter
  rocprofv3 --pmc $2 --output-format csv -d $R/gpurun_out/pmc_$1_$2 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nms > /dev/null 2> $R/gpurun_out/pmc_$1_$2.err
}
run A FETCH_SIZE; run A WRITE_SIZE
export MNY_DW_XCD=0; run B FETCH_SIZE; unset MNY_DW_XCD
export MNY_DW_V1=1; run D FETCH_SIZE; run D WRITE_SIZE; unset MNY_DW_V1
cd $R
python3 - <<'PY'
import csv, glob, collections, os, re
for d in sorted(glob.glob('gpurun_out/pmc_*_*_SIZE')):
    f = glob.glob(d + '/**/run_counter_collection.csv', recursive=True)
    if not f: print(d, 'no csv'); continue
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        if 'dw3_fwd' in k or 'dw_slide' in k:
            k = re.sub(r'^void mny::', '', k); k = re.sub(r'\(.*', '', k)
            tot[k] += float(r['Counter_Value']); cnt[k] += 1
    for k in sorted(tot): print(os.path.basename(d), k, cnt[k], 'launches', round(tot[k] * 1024 / 3 / 1e9, 3), 'GB/step (raw counter, KiB units)')
PY
