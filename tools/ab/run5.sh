python -m pytest tests/test_gpu_exdw.py -m gpu -x -q 2>&1 | tail -2
export MNY_SIDE_STREAM=0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fix -o fix -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-nms --steps 6 --warmup 2 --roofline-pass after > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_fix/**/*kernel_stats.csv',recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    tot+=float(r['TotalDurationNs'])
    if 'exdw' in r['Name']:
        print(r['Name'][:70], r['Calls'], r['AverageNs'])
print(tot/1e6)
PY
