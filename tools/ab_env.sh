# same-box A/B of one environment switch: usage  bash tools/ab_env.sh VAR=VALUE [bench flags]
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-nms $2"
for i in 1 2 3; do
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'])"
  env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"
done
