#!/bin/bash
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "fused_dw_unit_backward" > gpurun_out/r5/dwt_tests.txt 2>&1; echo "kernel tests rc=$?"; tail -3 gpurun_out/r5/dwt_tests.txt
echo "== register form"; MNY_DWT3=0 timeout 300 python tools/bench_dwbwd.py 64 bf16 3 2>&1 | grep -v amdgpu.ids | grep -E "^C16 |^C72 |share" | cut -c1-120
echo "== rule"; timeout 300 python tools/bench_dwbwd.py 64 bf16 3 2>&1 | grep -v amdgpu.ids | grep -E "^C16 |^C72 |share" | cut -c1-120
bash tools/rounds/r5_ab.sh "MNY_DWT3=0 MNY_X=0" c3 2>&1
