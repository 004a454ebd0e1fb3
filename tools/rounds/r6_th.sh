#!/bin/bash
# round 6: strip heights of the register-form depthwise backward kernels on configs[3] (small maps, bs 64: few strips per launch)
mkdir -p gpurun_out/r6
{
bash tools/rounds/r5_ab.sh "MNY_DWB_TH=32 MNY_DWB_TH=16 MNY_DWB_TH=8 MNY_DWB2_TH=8 MNY_DWB5_TH=8,MNY_DWB_TH=8,MNY_DWB2_TH=8" c3
} > gpurun_out/r6/th.txt 2>&1
