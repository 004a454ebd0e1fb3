python -m pytest tests/test_gpu_exdw.py -m gpu -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fix -o fix -- python3 $GRAFT_REPO_ROOT/tools/bench_exdw.py bwd 256 > $GRAFT_REPO_ROOT/gpurun_out/bench_exdw_fix.txt 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/bench_exdw_fix.txt | grep -v amdgpu.ids
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_fix/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'exdw' in r['Name']:
        print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
