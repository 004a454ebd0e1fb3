mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6/smoke2.txt 2>&1
python -m pytest tests/test_gpu_exdw.py tests/test_gpu_stemdw.py -q -m gpu 2>&1 | tail -2 >> gpurun_out/r6/smoke2.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['config']['loss'])" >> gpurun_out/r6/smoke2.txt
