mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_gate.py tests/test_gpu_pjbwd.py -q -s -k "default_init or unfused or bs16 or matches_oracle or gate or bf16_storage" 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r6/t1.txt
