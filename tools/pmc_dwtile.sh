# wave-level picture of the tile-form depthwise backward kernels (csrc/dwtile.hip) next to the register forms / un-fused launches they replace
# (own PMC runs; program directly after `--`).  usage: bash tools/pmc_dwtile.sh    -> gpurun_out/pmc_dwtile.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
pass() {  # tag, K, counters...
  local tag=$1 k=$2; shift 2
  rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_dwt_$tag -o run -- python3 $R/tools/bench_dwbwd.py 64 bf16 $k > /dev/null 2>&1
}
pass a5 5 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES
pass b5 5 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM
pass a3 3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES
pass b3 3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM
export MNY_DWT3=0
pass a3r 3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES
pass b3r 3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM
unset MNY_DWT3
cd $R
python3 - > gpurun_out/pmc_dwtile.txt <<'PY'
import csv, glob, collections, re
def short(n): return re.sub(r'\(.*', '', n).replace('void mny::', '').replace('mny::', '')[:70]
print("wave-level counters per launch, bench_dwbwd.py 64 bf16 {5,3} (MobileNetV3 512x512 bs-64 shapes); averaged over the launches of a kernel name")
for tags, title in ((('a5', 'b5'), '5x5 stride 1: tile form + the un-fused launches it replaces'), (('a3', 'b3'), '3x3 stride 1, routing rule (tile form where it is faster)'),
                    (('a3r', 'b3r'), '3x3 stride 1, register form everywhere (MNY_DWT3=0)')):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for t in tags:
        for f in glob.glob('gpurun_out/pmc_dwt_%s/**/run_counter_collection.csv' % t, recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                if any(s in k for s in ('dwb_tile', 'dw_bnbwd', 'dw5_wgrad', 'dw5_fwd', 'bn_bwd_apply', 'bn_bwd_reduce')):
                    tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
    print("\n== " + title)
    for k in sorted(tot):
        c = {n: v / cnt[k][n] for n, v in tot[k].items()}
        wc = c.get('SQ_WAVE_CYCLES', 1) or 1
        print(k, "(%d launches)" % max(cnt[k].values()))
        print("   insts per launch: valu %.3g salu %.3g lds %.3g vmem %.3g | waves %.0f, busy cycles %.3g" % (c.get('SQ_INSTS_VALU', 0), c.get('SQ_INSTS_SALU', 0), c.get('SQ_INSTS_LDS', 0),
              c.get('SQ_INSTS_VMEM', 0), c.get('SQ_WAVES', 0), c.get('SQ_BUSY_CYCLES', 0)))
        print("   share of wave cycles: waiting (any) %.2f, issue-stalled %.2f, active %.2f | active valu %.3f lds %.3f | waiting on lds %.3f | lds bank conflicts / wave cycle %.3f" % (
              c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_VALU', 0) / wc,
              c.get('SQ_ACTIVE_INST_LDS', 0) / wc, c.get('SQ_WAIT_INST_LDS', 0) / wc, c.get('SQ_LDS_BANK_CONFLICT', 0) / wc))
PY
cat gpurun_out/pmc_dwtile.txt | cut -c1-220
