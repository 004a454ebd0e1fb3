#!/bin/bash
# timing builds of exdw_bwd (WRONG results): which side of the producer / consumer split is the longer one.  Libraries: tools/ab/lib_exdw_<variant>.so
mkdir -p gpurun_out/r6
{
for v in base NOPROD NOCONS NODX NOCONTRACT; do
  echo "== $v"
  if [ $v = base ]; then EXDW_REPS=20 python tools/bench_exdw.py bwd 256 2>&1 | grep "exdw_bwd"
  else MNY_LIB=$GRAFT_REPO_ROOT/tools/ab/lib_exdw_$v.so EXDW_REPS=20 python tools/bench_exdw.py bwd 256 2>&1 | grep "exdw_bwd"; fi
done
} > gpurun_out/r6/exdw_parts.txt 2>&1
