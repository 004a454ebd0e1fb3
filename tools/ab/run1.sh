set -x
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_exdw.py tests/test_gpu_net.py -m gpu -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-nms --roofline-pass after --steps 20"
for wb in 1536 768 1024 2048 1536; do
  echo "WG_BLOCKS=$wb"; MNY_WG_BLOCKS=$wb $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fin -o fin -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-nms --roofline-pass after --steps 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_fin/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'finalize' in r['Name'] or 'reduce_parts' in r['Name'] or 'copyBuffer' in r['Name']:
        print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
