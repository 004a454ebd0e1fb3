mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_lr.py -q 2>&1 | tail -3 > gpurun_out/r6/lr_tests.txt
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-nms --breakdown --detail mny_lr_gram > /dev/null 2> gpurun_out/r6/lr_c1.txt
