# headline kernel stats + FETCH / WRITE / MFMA counters at the final kernels (the 128 x 256 weight-gradient tile changes mny_pw_wgrad's traffic)
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
bash tools/prof_round.sh > gpurun_out/prof_round.log 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out | tail -1
