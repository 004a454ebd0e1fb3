mkdir -p gpurun_out/r6
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
bash tools/pmc_exdw.sh 16 > gpurun_out/r6/pmc_exdw.txt 2>&1
rm -rf gpurun_out/pmc_ex1 gpurun_out/pmc_ex2 gpurun_out/pmc_ex3 gpurun_out/pmc_ex4
