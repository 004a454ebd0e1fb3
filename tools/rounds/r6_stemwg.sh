#!/bin/bash
# round 6: stem weight gradient on the matrix cores vs the vector-ALU form (same box), + the stem unit tests + the configs[3] A/B
mkdir -p gpurun_out/r6
{
echo "== mfma"; python tools/bench_stemwg.py
echo "== valu"; MNY_STEM_WGRAD_VALU=1 python tools/bench_stemwg.py
echo "== tests"; python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py tests/test_gpu_stemdw.py -q -m gpu -k "stem" 2>&1 | tail -5
echo "== c3 A/B"; bash tools/rounds/r5_ab.sh "MNY_STEM_WGRAD_VALU=1 MNY_STEM_WGRAD_VALU=0" c3
} > gpurun_out/r6/stemwg.txt 2>&1
