mkdir -p gpurun_out/r6
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r6/full_gpu_tests.txt
