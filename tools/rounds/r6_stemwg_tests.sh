mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py tests/test_gpu_stemdw.py -q -m gpu -k "stem" 2>&1 | tail -8 > gpurun_out/r6/stemwg_tests.txt
python -m pytest tests/test_gpu_net.py tests/test_gpu_bf16.py -q -m gpu -x 2>&1 | tail -5 >> gpurun_out/r6/stemwg_tests.txt
