#!/bin/bash
# end-of-round sequence: full GPU suite, configs[3] profile, the driver-style line, the standalone configs[3] lines
mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r5/full_gpu_tests.txt 2>&1; echo "full gpu suite rc=$?"; tail -3 gpurun_out/r5/full_gpu_tests.txt
bash tools/prof_c3.sh > gpurun_out/r5/prof_c3.log 2>&1; echo "prof_c3 rc=$?"
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r5/bench_n1.json 2> gpurun_out/r5/bench_n1.err; echo "bench rc=$?"
python bench.py --arch mbv3 --size 512 --batch 64 --dtype bf16 --no-nms > gpurun_out/r5/bench_mbv3_bf16.json 2> gpurun_out/r5/bench_mbv3_bf16.err; echo "c3 rc=$?"
python bench.py --arch mbv3 --size 512 --batch 64 --no-nms > gpurun_out/r5/bench_mbv3_f32.json 2> gpurun_out/r5/bench_mbv3_f32.err; echo "c3 fp32 rc=$?"
