mkdir -p gpurun_out/r6
python tools/bench_dwbwd.py 256 f32 3 > gpurun_out/r6/dwbwd_base.txt 2>&1
