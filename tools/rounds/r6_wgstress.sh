mkdir -p gpurun_out/r6
python tools/ab/stress_wgrad_wide.py > gpurun_out/r6/wgstress.txt 2>&1
