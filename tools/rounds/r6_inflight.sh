mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_net.py tests/test_gpu_train_loop.py tests/test_gpu_00_dp.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r6/inflight_tests.txt
