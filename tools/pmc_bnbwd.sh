# wave-level stall picture of the fused expand-unit backward kernels (own PMC runs; program directly after `--`)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for shape in "7929856 16 96" "1982464 24 144"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_bnw_$tag -o run -- python3 $R/tools/kbench.py bnbwd $shape 5 > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc_bnw2_$tag -o run -- python3 $R/tools/kbench.py bnbwd $shape 5 > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc_bnw3_$tag -o run -- python3 $R/tools/kbench.py bnbwd $shape 5 > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_bnw*_*')):
    f = glob.glob(d + '/**/run_counter_collection.csv', recursive=True)
    if not f:
        g = glob.glob(d + '/**/run_kernel_stats.csv', recursive=True)
        if g:
            for r in csv.DictReader(open(g[0])):
                if 'bnbwd' in r['Name']: print(d, r['Name'][:60], r['Calls'], r['AverageNs'])
        continue
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].split('(')[0][-40:]
        if 'bnbwd' in k:
            tot[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
    for k in tot: print(d, k, {c: round(v / n[k][c]) for c, v in sorted(tot[k].items())})
PY
