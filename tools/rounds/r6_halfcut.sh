# timing experiment: the weight-gradient kernel with half its operand cuts (wrong results) = the bound of sharing the cut between the waves of a workgroup
mkdir -p gpurun_out/r6
for lib in "" "tools/experiments/libmnyolo_halfcut.so"; do
  for shape in "123904 512 512" "30976 1280 512" "30976 512 1024"; do
    env MNY_LIB=$lib python tools/kbench.py wgrad $shape 20
  done
done > gpurun_out/r6/halfcut.txt 2>&1
for rep in 1 2; do
env MNY_LIB= python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-nms --roofline-pass after 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('full', d['ms_per_step'])"
env MNY_LIB=tools/experiments/libmnyolo_halfcut.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-nms --roofline-pass after 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('halfcut', d['ms_per_step'])"
done >> gpurun_out/r6/halfcut.txt 2>&1
