mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_lr.py tests/test_gpu_kernels.py -q 2>&1 | tail -5 > gpurun_out/r6/bnw_tests.txt
python -m pytest tests/test_gpu_net.py tests/test_gpu_00_dp.py -q -k "not 256" 2>&1 | tail -4 >> gpurun_out/r6/bnw_tests.txt
bash tools/rounds/r5_ab.sh "MNY_NO_BNW_RED=1 MNY_NO_BNW_RED=0" c1 > gpurun_out/r6/bnw_ab.txt 2>&1
