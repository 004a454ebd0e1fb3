mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wide_tile_weight" 2>&1 | tail -3 > gpurun_out/r6/wgtest.txt
