# second counter set of the whole headline step: LDS / TA back-pressure, instruction fetch, co-execution (own PMC runs)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL --output-format csv -d $R/gpurun_out/pmc_s2a -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nms > /dev/null 2> $R/gpurun_out/pmc_s2.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/pmc_s2b -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nms > /dev/null 2>> $R/gpurun_out/pmc_s2.err
cd $R
python3 - <<'PY'
import csv, glob, collections, re
def short(n): return re.sub(r'\(.*', '', n).replace('void mny::', '').replace('mny::', '')
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ('pmc_s2a', 'pmc_s2b'):
    f = glob.glob('gpurun_out/%s/**/run_counter_collection.csv' % d, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        tot[short(r['Kernel_Name'])][r['Counter_Name']] += float(r['Counter_Value'])
rows = sorted(tot.items(), key=lambda kv: -kv[1]['SQ_BUSY_CYCLES'])
names = ['SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_DATA_FIFO_FULL', 'SQ_LDS_CMD_FIFO_FULL', 'SQ_VMEM_TA_ADDR_FIFO_FULL', 'SQ_VMEM_TA_CMD_FIFO_FULL', 'SQ_IFETCH_LEVEL',
         'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_INST_CYCLES_VMEM_RD', 'SQ_VALU_MFMA_COEXEC_CYCLES', 'SQ_INSTS_BRANCH']
print("per SQ_WAVE_CYCLES (x1000):", " ".join(n.replace('SQ_', '')[:14] for n in names))
for k, c in rows[:45]:
    wc = c['SQ_WAVE_CYCLES'] or 1
    print("%-46s busy %9.0f |" % (k[:46], c['SQ_BUSY_CYCLES']), " ".join("%7.1f" % (1000 * c[n] / wc) for n in names))
PY
