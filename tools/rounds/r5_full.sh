#!/bin/bash
mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r5/full_gpu_tests.txt 2>&1; echo "full gpu suite rc=$?"; tail -3 gpurun_out/r5/full_gpu_tests.txt
bash tools/rounds/r5_ab.sh "MNY_DWTF=0 MNY_X=0" c3 2>&1 | tee gpurun_out/r5/dwtf_ab.txt
