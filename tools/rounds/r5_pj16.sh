#!/bin/bash
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_pjbwd.py tests/test_gpu_gate.py -q > gpurun_out/r5/pj16_tests.txt 2>&1; echo "pj/gate tests rc=$?"; tail -2 gpurun_out/r5/pj16_tests.txt
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_mbv3.py -q -x > gpurun_out/r5/pj16_mbv3.txt 2>&1; echo "mbv3 rc=$?"; tail -2 gpurun_out/r5/pj16_mbv3.txt
python bench.py --arch mbv3 --size 512 --batch 64 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown --detail all > gpurun_out/r5/c3_f.json 2> gpurun_out/r5/c3_f.txt
grep -E "mny_pj_bwd_bf16" gpurun_out/r5/c3_f.txt | cut -c1-110
bash tools/rounds/r5_ab.sh "MNY_X=0" c3 2>&1
