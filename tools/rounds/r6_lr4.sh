mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_lr.py -q 2>&1 | tail -5 > gpurun_out/r6/lr_tests.txt
python -m pytest tests/test_gpu_net.py tests/test_gpu_00_dp.py -q -k "not 256" 2>&1 | tail -5 > gpurun_out/r6/lr_net_tests.txt
bash tools/rounds/r5_ab.sh "MNY_NO_LR_S2=1 MNY_NO_LR_S2=0 MNY_NO_LR=1" c1 > gpurun_out/r6/lr_ab.txt 2>&1
