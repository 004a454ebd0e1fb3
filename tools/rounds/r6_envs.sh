# fallback routes stay green: every round-6 switch off, one stream
mkdir -p gpurun_out/r6
for e in "MNY_SIDE_STREAM=0" "MNY_NO_LR=1" "MNY_NO_ADDRED=1 MNY_NO_BNW_RED=1 MNY_NO_LR_S2=1" "MNY_NO_DEAD_SIDE=1 MNY_NO_LOSS_SIDE=1 MNY_LR_PREP_MAIN=1 MNY_LR_FIX_DMA=1" "MNY_LANE2=1"; do
  echo "== $e"; env $e python -m pytest tests/test_gpu_net.py tests/test_gpu_00_dp.py tests/test_gpu_mbv3.py -q -k "not 256" 2>&1 | tail -1
done > gpurun_out/r6/envs.txt 2>&1
