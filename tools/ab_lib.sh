# same-box A/B of two builds of the library: tools/ab/libmnyolo_prev.so (MNY_LIB) against the in-tree one; alternating runs
R=$GRAFT_REPO_ROOT/gpurun_out
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-nms --roofline-pass after"
for i in 1 2; do
  $B > $R/ab_new_$i.json 2> /dev/null
  MNY_LIB=$GRAFT_REPO_ROOT/tools/ab/libmnyolo_prev.so $B > $R/ab_prev_$i.json 2> /dev/null
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown > /dev/null 2> $R/ab_new_bd.txt
MNY_LIB=$GRAFT_REPO_ROOT/tools/ab/libmnyolo_prev.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown > /dev/null 2> $R/ab_prev_bd.txt
python - <<'PY'
import json,glob,os
R=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'
for k in ('new','prev'):
    print(k, [json.load(open(f))['ms_per_step'] for f in sorted(glob.glob(R+'ab_%s_?.json'%k))])
PY
paste <(grep "ms/step" $R/ab_new_bd.txt | awk '{print $1,$2}') <(grep "ms/step" $R/ab_prev_bd.txt | awk '{print $1,$2}')
