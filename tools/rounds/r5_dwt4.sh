#!/bin/bash
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "dw_forward" > gpurun_out/r5/dwf_tests.txt 2>&1; echo "fwd tests rc=$?"; tail -4 gpurun_out/r5/dwf_tests.txt
echo "== forward, register forms (MNY_DWTF=0)"; MNY_DWTF=0 timeout 300 python tools/bench_dwfwd.py bf16 2>&1 | grep -v amdgpu.ids | cut -c1-150
echo "== forward, tile rule"; timeout 300 python tools/bench_dwfwd.py bf16 2>&1 | grep -v amdgpu.ids | cut -c1-150
echo "== forward, tile everywhere (MNY_DWTF=1)"; MNY_DWTF=1 timeout 300 python tools/bench_dwfwd.py bf16 2>&1 | grep -v amdgpu.ids | cut -c1-150
