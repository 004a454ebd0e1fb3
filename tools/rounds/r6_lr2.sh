mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_lr.py -x -q 2>&1 | tail -5 > gpurun_out/r6/lr_tests.txt
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-nms --breakdown --detail all > gpurun_out/r6/lr_c1.json 2> gpurun_out/r6/lr_c1.txt
