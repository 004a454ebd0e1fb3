# wave-level picture of the stem + depthwise backward kernel and of the kernels it replaces (own PMC runs; program directly after `--`).  usage: bash tools/pmc_stemdw.sh
R=$GRAFT_REPO_ROOT
K=${1:-16}
cd /tmp && export TMPDIR=/tmp && export EXDW_REPS=2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_sd1 -o run -- python3 $R/tools/bench_stemdw.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/pmc_sd2 -o run -- python3 $R/tools/bench_stemdw.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT --output-format csv -d $R/gpurun_out/pmc_sd3 -o run -- python3 $R/tools/bench_stemdw.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc_sd4 -o run -- python3 $R/tools/bench_stemdw.py > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, re
def short(n): return re.sub(r'\(.*', '', n).replace('void mny::', '').replace('mny::', '')[:44]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for d in ('pmc_sd1', 'pmc_sd2', 'pmc_sd3'):
    for f in glob.glob('gpurun_out/%s/**/run_counter_collection.csv' % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if 'stemdw' in k or 'dw_bnbwd_s1k3' in k or 'stem_tile' in k or 'pj_bwd' in k:
                tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k in sorted(tot):
    c = {n: v / cnt[k][n] for n, v in tot[k].items()}
    wc = c.get('SQ_WAVE_CYCLES', 1) or 1
    print(k)
    print("   per launch: busy_cycles %.0f waves %.0f wave_cycles %.3g | insts: valu %.3g salu %.3g lds %.3g vmem %.3g mfma %.3g branch %.3g" % (
        c.get('SQ_BUSY_CYCLES', 0), c.get('SQ_WAVES', 0), wc, c.get('SQ_INSTS_VALU', 0), c.get('SQ_INSTS_SALU', 0), c.get('SQ_INSTS_LDS', 0), c.get('SQ_INSTS_VMEM', 0), c.get('SQ_INSTS_MFMA', 0), c.get('SQ_INSTS_BRANCH', 0)))
    print("   share of wave cycles: wait_any %.2f wait_inst_any %.2f active_any %.2f | active valu %.3f lds %.3f vmem %.3f sca %.3f misc %.3f | wait_lds %.3f wait_vmem %.3f | mfma_busy/busy %.2f lds_conflict/wave_cyc %.3f" % (
        c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_VALU', 0) / wc, c.get('SQ_ACTIVE_INST_LDS', 0) / wc,
        c.get('SQ_ACTIVE_INST_VMEM', 0) / wc, c.get('SQ_ACTIVE_INST_SCA', 0) / wc, c.get('SQ_ACTIVE_INST_MISC', 0) / wc, c.get('SQ_WAIT_INST_LDS', 0) / wc, c.get('SQ_WAIT_INST_VMEM', 0) / wc,
        c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (c.get('SQ_BUSY_CYCLES', 1) or 1) / 4, c.get('SQ_LDS_BANK_CONFLICT', 0) / wc))
for f in glob.glob('gpurun_out/pmc_sd4/**/run_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(t in r['Name'] for t in ('stemdw', 'dw_bnbwd_s1k3', 'stem_tile', 'pj_bwd')): print("%-60s calls %s avg %.1f us" % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3))
PY
