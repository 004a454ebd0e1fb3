#!/bin/bash
# round 6: fused 5x5 stride-2 depthwise backward: unit tests, configs[3] A/B against the unfused launches (same box)
mkdir -p gpurun_out/r6
{
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "dw5_stride2" 2>&1 | tail -4
echo "== c3 A/B"; bash tools/rounds/r5_ab.sh "MNY_NO_DWFUSE5S2=1 MNY_NO_DWFUSE5S2=0" c3
} > gpurun_out/r6/dw5s2.txt 2>&1
