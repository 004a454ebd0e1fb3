python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_dw" 2>&1 | tail -2
echo "== coop (default, bf16)"; python tools/bench_dwbwd.py 64 bf16 2>&1 | grep -v amdgpu | cut -c1-120
echo "== MNY_DWB_COOP=0"; MNY_DWB_COOP=0 python tools/bench_dwbwd.py 64 bf16 2>&1 | grep -v amdgpu | cut -c1-120
echo "== coop cgb 16"; MNY_DWB_COOP_CGB=16 python tools/bench_dwbwd.py 64 bf16 2>&1 | grep -v amdgpu | cut -c1-120
