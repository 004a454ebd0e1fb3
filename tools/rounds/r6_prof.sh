# round-6 evidence: headline kernel stats + FETCH / WRITE / MFMA counters (prof_round.sh), configs[3] (prof_c3.sh), six-product GEMM wave counters (pmc_x6.sh)
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
bash tools/prof_round.sh > gpurun_out/prof_round.log 2>&1
bash tools/prof_c3.sh > gpurun_out/prof_c3.log 2>&1
bash tools/pmc_x6.sh > gpurun_out/pmc_x6.log 2>&1
# keep what the summaries need small enough to come back (<= 64 MiB): the per-dispatch trace CSVs are large
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out | tail -1
