#!/bin/bash
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "fused_dw_unit_backward" > gpurun_out/r5/dwt_tests.txt 2>&1; echo "kernel tests rc=$?"; tail -2 gpurun_out/r5/dwt_tests.txt
MNY_DWT3=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "fused_dw_unit_backward" > gpurun_out/r5/dwt_tests3.txt 2>&1; echo "kernel tests (3x3 tile) rc=$?"; tail -2 gpurun_out/r5/dwt_tests3.txt
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_mbv3.py -q -x > gpurun_out/r5/dwt_mbv3.txt 2>&1; echo "mbv3/bf16 tests rc=$?"; tail -3 gpurun_out/r5/dwt_mbv3.txt
bash tools/rounds/r5_ab.sh "MNY_NO_DWT5=1,MNY_DWT3=0 MNY_DWT3=0 MNY_X=0" c3 2>&1 | tee gpurun_out/r5/dwt_ab.txt
