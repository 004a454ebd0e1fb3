python -m pytest tests/test_gpu_net.py tests/test_gpu_train_loop.py -m gpu -x -q 2>&1 | tail -3
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-nms --roofline-pass after"
for i in 1 2; do
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gram', d['ms_per_step'], d['config']['loss'])"
  MNY_EXDW_STATS=direct $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('direct', d['ms_per_step'], d['config']['loss'])"
done
