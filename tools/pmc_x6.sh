# wave-level picture of the six-product GEMM kernels inside the headline step ("first the evidence"): VALU / MFMA / LDS
# activity, waits, issue stalls, LDS bank conflicts, waves — own PMC runs over a short `python bench.py`, program directly after `--`.
# usage: bash tools/pmc_x6.sh   -> gpurun_out/pmc_x6.txt  (copy to profiles/r06_pmc_x6.md)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export MNY_SIDE_STREAM=0
pass() {  # tag, counters...
  local tag=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_x6_$tag -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nms > /dev/null 2> $R/gpurun_out/pmc_x6_$tag.err
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM
pass c SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM
cd $R
python3 - > gpurun_out/pmc_x6.txt <<'PY'
import csv, glob, collections, re
def short(n): return re.sub(r', 0>$', '>', re.sub(r'\(.*', '', n).replace('void mny::', '').replace('mny::', ''))[:80] if 'pw_gemm_nt_dma' in n else re.sub(r'\(.*', '', n).replace('void mny::', '').replace('mny::', '')[:80]
want = ('pw_gemm_nt_dma_kernel<4, 1, 0, 0, 3>', 'pw_gemm_nt_dma_kernel<4, 0, 0, 0, 3>', 'pw_wgrad_dma_kernel<0, 2, 2, 1>', 'pw_gemm_nt_dma_kernel<3, 0, 0, 1, 3>',
        'pw_gemm_nt_dma_kernel<5, 1, 0, 0, 3>', 'pw_wgrad_stream_kernel<2, 3>', 'pw_wide_kernel<6, 3, 0, 2>')
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for t in 'abc':
    for f in glob.glob('gpurun_out/pmc_x6_%s/**/run_counter_collection.csv' % t, recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if k in want:
                tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
print("# Wave-level counters of the six-product GEMM kernels inside the headline step (`python bench.py --steps 2 --warmup 1`, single stream)\n")
print("rocprofv3 --pmc, three separate passes (tools/pmc_x6.sh); values are averages PER LAUNCH over every launch of the kernel name in the run.\n")
for k in want:
    if k not in tot: continue
    c = {n: v / cnt[k][n] for n, v in tot[k].items()}
    wc = c.get('SQ_WAVE_CYCLES', 1) or 1
    busy = c.get('SQ_BUSY_CYCLES', 1) or 1
    print("`%s` (%d launches counted)" % (k, max(cnt[k].values())))
    print("   instructions per launch: valu %.3g, mfma %.3g, lds %.3g, salu %.3g, vmem %.3g; waves %.0f" % (c.get('SQ_INSTS_VALU', 0), c.get('SQ_INSTS_MFMA', 0),
          c.get('SQ_INSTS_LDS', 0), c.get('SQ_INSTS_SALU', 0), c.get('SQ_INSTS_VMEM', 0), c.get('SQ_WAVES', 0)))
    print("   share of wave cycles: waiting (any) %.2f, issue-stalled %.2f, active %.2f | active valu %.3f, lds %.3f, scalar %.3f, misc %.3f | waiting on lds %.3f | "
          "lds bank conflicts / wave cycle %.4f" % (c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_ANY', 0) / wc,
          c.get('SQ_ACTIVE_INST_VALU', 0) / wc, c.get('SQ_ACTIVE_INST_LDS', 0) / wc, c.get('SQ_ACTIVE_INST_SCA', 0) / wc, c.get('SQ_ACTIVE_INST_MISC', 0) / wc,
          c.get('SQ_WAIT_INST_LDS', 0) / wc, c.get('SQ_LDS_BANK_CONFLICT', 0) / wc))
    print("   matrix pipe: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES = %.3f; valu : mfma instructions = %.1f : 1\n" % (
          c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / busy, c.get('SQ_INSTS_VALU', 0) / max(c.get('SQ_INSTS_MFMA', 1), 1)))
PY
cat gpurun_out/pmc_x6.txt | cut -c1-260
