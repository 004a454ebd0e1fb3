python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_dw" 2>&1 | tail -3
echo "== coop (default)"; python tools/bench_dwbwd.py 256
echo "== MNY_DWB_COOP=0"; MNY_DWB_COOP=0 python tools/bench_dwbwd.py 256
echo "== coop cgb 64"; MNY_DWB_COOP_CGB=64 python tools/bench_dwbwd.py 256
echo "== coop cgb 16"; MNY_DWB_COOP_CGB=16 python tools/bench_dwbwd.py 256
