# fallback routes of the late-round switches stay green (whole-network tests incl. bf16 storage)
mkdir -p gpurun_out/r6
for e in "MNY_NO_DWFUSE5S2=1 MNY_STEM_WGRAD_VALU=1" "MNY_WG_TJ4=0 MNY_WG_BLOCKS=1536"; do
  echo "== $e"; env $e python -m pytest tests/test_gpu_net.py tests/test_gpu_mbv3.py tests/test_gpu_bf16.py -q -k "not 256" 2>&1 | tail -1
done > gpurun_out/r6/envs2.txt 2>&1
