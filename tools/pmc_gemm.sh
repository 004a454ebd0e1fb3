# wave-level stall picture of two mid-size GEMM shapes (own PMC run; program directly after `--`)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for shape in "123904 64 384" "123904 512 512"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_gemm_$tag -o run -- python3 $R/tools/kbench.py pwx $shape 10 > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc_gemm2_$tag -o run -- python3 $R/tools/kbench.py pwx $shape 10 > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_gemm*_*')):
    f = glob.glob(d + '/**/run_counter_collection.csv', recursive=True)
    if not f: print(d, 'no csv'); continue
    tot = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if 'pw_gemm_nt_dma' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    print(d, {k: round(v / n[k]) for k, v in sorted(tot.items())})
PY
