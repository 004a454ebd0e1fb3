#!/bin/bash
# round 6: exdw backward with the per-row 64-bit index arithmetic hoisted: same-box A/B against the previous build (tools/ab/libmnyolo_prev.so)
mkdir -p gpurun_out/r6
{
for i in 1 2; do
echo "== new"; EXDW_REPS=20 python tools/bench_exdw.py bwd 256
echo "== prev"; MNY_LIB=$GRAFT_REPO_ROOT/tools/ab/libmnyolo_prev.so EXDW_REPS=20 python tools/bench_exdw.py bwd 256
done
python -m pytest tests/test_gpu_exdw.py -q -m gpu 2>&1 | tail -3
} > gpurun_out/r6/exdw_ab.txt 2>&1
