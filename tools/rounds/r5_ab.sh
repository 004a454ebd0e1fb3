# usage: bash tools/rounds/r5_ab.sh "VAR=a VAR=b ..." [c1|c3|both]  -> ms/step of the un-bracketed bench for each setting (same box)
SETS="$1"; WHICH="${2:-both}"
c3() { env ${1//,/ } python bench.py --arch mbv3 --size 512 --batch 64 --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-nms --roofline-pass after 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 $*', d['ms_per_step'], d['config']['loss'])"; }
c1() { env ${1//,/ } python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-nms --roofline-pass after 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c1 $*', d['ms_per_step'], d['config']['loss'])"; }
for rep in 1 2; do for s in $SETS; do
  if [ "$WHICH" != "c1" ]; then c3 $s; fi
  if [ "$WHICH" != "c3" ]; then c1 $s; fi
done; done
