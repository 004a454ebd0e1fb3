mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_net.py tests/test_gpu_00_dp.py tests/test_gpu_mbv3.py tests/test_gpu_seg.py -q -k "not 256" 2>&1 | tail -4 > gpurun_out/r6/lane_tests.txt
bash tools/rounds/r5_ab.sh "MNY_NO_LANE2=1 MNY_NO_LANE2=0" both > gpurun_out/r6/lane_ab.txt 2>&1
