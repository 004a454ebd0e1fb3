# configs[3] under one-variable settings, same box (ms/step; default interleaved)
B="python bench.py --arch mbv3 --size 512 --batch 64 --dtype bf16 --steps 15 --warmup 4 --no-cpu-baseline --no-nms --roofline-pass after"
run() { env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-22s %.3f' % ('$1', d['ms_per_step']))"; }
run X=0
for s in "$@"; do run $s; run X=0; done
