#!/bin/bash
# tile-form depthwise backward: kernel tests, then the micro-benchmarks at 3 and 2 waves per SIMD
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "fused_dw_unit_backward" > gpurun_out/r5/dwt_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r5/dwt_tests.txt
tail -5 gpurun_out/r5/dwt_tests.txt
MNY_DWT3=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "fused_dw_unit_backward" > gpurun_out/r5/dwt_tests3.txt 2>&1
echo "tests (3x3 on the tile form) rc=$?" >> gpurun_out/r5/dwt_tests3.txt
tail -3 gpurun_out/r5/dwt_tests3.txt
timeout 300 python tools/bench_dwbwd.py 64 bf16 5 > gpurun_out/r5/dwt_b5_w3.txt 2>&1
MNY_DWT_WPE=2 timeout 300 python tools/bench_dwbwd.py 64 bf16 5 > gpurun_out/r5/dwt_b5_w2.txt 2>&1
MNY_DWT3=1 timeout 300 python tools/bench_dwbwd.py 64 bf16 3 > gpurun_out/r5/dwt_b3_tile.txt 2>&1
MNY_DWT3=1 MNY_DWT_WPE=2 timeout 300 python tools/bench_dwbwd.py 64 bf16 3 > gpurun_out/r5/dwt_b3_tile_w2.txt 2>&1
for f in dwt_b5_w3 dwt_b5_w2 dwt_b3_tile dwt_b3_tile_w2; do echo "== $f"; grep -v amdgpu.ids gpurun_out/r5/$f.txt | cut -c1-118,250-340; done
