# wave-level instruction / stall counters of EVERY kernel of the headline step (own PMC run over bench.py; program directly after `--`)
R=$GRAFT_REPO_ROOT
ARGS=${1:-}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_step -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nms $ARGS > /dev/null 2> $R/gpurun_out/pmc_step.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc_step_t -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nms $ARGS > /dev/null 2>> $R/gpurun_out/pmc_step.err
cd $R
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob('gpurun_out/pmc_step/**/run_counter_collection.csv', recursive=True)[0]
g = glob.glob('gpurun_out/pmc_step_t/**/run_kernel_stats.csv', recursive=True)[0]
def short(n): return re.sub(r'\(.*', '', n).replace('void mny::', '').replace('mny::', '')
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = short(r['Kernel_Name']); tot[k][r['Counter_Name']] += float(r['Counter_Value'])
times = {short(r['Name']): (int(r['TotalDurationNs']), int(r['Calls'])) for r in csv.DictReader(open(g))}
rows = []
for k, c in tot.items():
    t, n = times.get(k, (0, 1))
    wc = c['SQ_WAVE_CYCLES'] or 1
    rows.append((t, k, n, c['SQ_INSTS_VALU'], c['SQ_INSTS_SALU'], c['SQ_INSTS_MFMA'], c['SQ_INSTS_LDS'], c['SQ_ACTIVE_INST_ANY'] / wc, c['SQ_WAIT_INST_ANY'] / wc, c['SQ_WAIT_ANY'] / wc))
rows.sort(reverse=True)
print("%-52s %5s %9s %6s %6s %6s | %6s %6s %6s" % ("kernel", "calls", "ms total", "SALU/V", "MFMA/V", "LDS/V", "active", "istall", "wait"))
for t, k, n, v, s, m, l, a, wi, wa in rows[:70]:
    v = v or 1
    print("%-52s %5d %9.3f %6.2f %6.3f %6.2f | %6.2f %6.2f %6.2f" % (k[:52], n, t / 1e6, s / v, m / v, l / v, a, wi, wa))
PY
