mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_net.py tests/test_gpu_lr.py tests/test_gpu_00_dp.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r6/wgblocks_tests.txt
