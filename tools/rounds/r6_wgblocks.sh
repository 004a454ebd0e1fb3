#!/bin/bash
# split budget of the weight gradients, 1536 (old default) vs 1024, both benchmark configurations (same box)
mkdir -p gpurun_out/r6
{
bash tools/rounds/r5_ab.sh "MNY_WG_BLOCKS=1536 MNY_WG_BLOCKS=1024" both
} > gpurun_out/r6/wgblocks.txt 2>&1
