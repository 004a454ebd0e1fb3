#!/bin/bash
# round 6: 128 x 256 workgroup tile for the six-product weight gradients (MNY_WG_TJ4=1) vs the 128 x 128 tile: kernel tests + same-box A/B of the headline
mkdir -p gpurun_out/r6
{
MNY_WG_TJ4=1 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wgrad or six_product or pointwise" 2>&1 | tail -4
bash tools/rounds/r5_ab.sh "MNY_WG_TJ4=0 MNY_WG_TJ4=1 MNY_WG_TJ4=1,MNY_WG_BLOCKS=1024 MNY_WG_TJ4=1,MNY_WG_BLOCKS=2048" c1
} > gpurun_out/r6/wgwide.txt 2>&1
