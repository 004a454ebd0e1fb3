#!/bin/bash
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -x -k "test_pw_bf16" > gpurun_out/r5/pwt_tests.txt 2>&1; echo "pw bf16 tests rc=$?"; tail -4 gpurun_out/r5/pwt_tests.txt
bash tools/rounds/r5_ab.sh "MNY_NO_PWT=1 MNY_X=0" c3 2>&1 | tee gpurun_out/r5/pwt_ab.txt
