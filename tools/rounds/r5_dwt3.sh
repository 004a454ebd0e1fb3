#!/bin/bash
mkdir -p gpurun_out/r5
for v in base pd2 pd1w2 pd2w2 pd3w2; do
  if [ $v = base ]; then L=""; else L="MNY_LIB=$PWD/tools/ab/lib_$v.so"; fi
  echo "== $v K5"; env $L timeout 300 python tools/bench_dwbwd.py 64 bf16 5 2>&1 | grep -v amdgpu.ids | cut -c1-112
  echo "== $v K3 tile"; env $L MNY_DWT3=1 timeout 300 python tools/bench_dwbwd.py 64 bf16 3 2>&1 | grep -v amdgpu.ids | grep -E "C120|C160|C480|C672|C960|share" | cut -c1-112
done
