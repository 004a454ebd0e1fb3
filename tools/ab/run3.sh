cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fin -o fin -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-nms --bracket-every 1 --steps 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_fin/**/*kernel_stats.csv',recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    tot+=float(r['TotalDurationNs'])
    if 'finalize' in r['Name'] or 'reduce_parts' in r['Name'] or 'copyBuffer' in r['Name'] or 'fillBuffer' in r['Name']:
        print(r['Name'][:60], r['Calls'], r['AverageNs'])
print('total kernel ms', tot/1e6)
PY
