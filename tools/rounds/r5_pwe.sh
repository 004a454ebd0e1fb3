#!/bin/bash
mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -k "expand_unit" > gpurun_out/r5/pwe_tests.txt 2>&1; echo "expand-unit tests rc=$?"; tail -5 gpurun_out/r5/pwe_tests.txt
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_mbv3.py -q -x > gpurun_out/r5/pwe_mbv3.txt 2>&1; echo "mbv3 rc=$?"; tail -2 gpurun_out/r5/pwe_mbv3.txt
bash tools/rounds/r5_ab.sh "MNY_NO_PWE=1 MNY_X=0" c3 2>&1
