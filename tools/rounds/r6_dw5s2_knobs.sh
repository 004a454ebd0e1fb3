#!/bin/bash
mkdir -p gpurun_out/r6
{
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "dw5_stride2" 2>&1 | tail -3
bash tools/rounds/r5_ab.sh "MNY_DWB5_TH=16 MNY_DWB5_TH=8 MNY_DWB5_TH=32 MNY_DWB5_CGB=32 MNY_DWB5_CGB=128 MNY_DWB5_RES=512" c3
} > gpurun_out/r6/dw5s2_knobs.txt 2>&1
