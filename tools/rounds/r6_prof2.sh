# round-6 evidence at the final kernels: headline kernel stats + FETCH / WRITE / MFMA counters (prof_round.sh), configs[3] (prof_c3.sh)
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
bash tools/prof_round.sh > gpurun_out/prof_round.log 2>&1
bash tools/prof_c3.sh > gpurun_out/prof_c3.log 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out | tail -1
