#!/bin/bash
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "dgrad_with_fused_bn or short_reduction or thin" > gpurun_out/r5/pwt2_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5/pwt2_tests.txt
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_mbv3.py -q -x > gpurun_out/r5/pwt2_mbv3.txt 2>&1; echo "mbv3 rc=$?"; tail -2 gpurun_out/r5/pwt2_mbv3.txt
bash tools/rounds/r5_ab.sh "MNY_NO_PWT=1 MNY_X=0" c3 2>&1
