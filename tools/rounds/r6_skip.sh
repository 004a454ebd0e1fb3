# timing experiment (stale results): which low-rank pieces cost what in the real two-stream step
mkdir -p gpurun_out/r6
run() { env SKIP_CALLS=$1 python tools/experiments/skip_calls.py --steps 20 --warmup 10 --no-cpu-baseline --no-nms --roofline-pass after 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('skip=$1', d['ms_per_step'])"; }
for rep in 1 2; do
run none
run mny_lr_gram,mny_lr_wfix
run mny_pw_lr_fix
run mny_lr_prep
run mny_lr_gram,mny_lr_wfix,mny_pw_lr_fix,mny_lr_prep
env MNY_NO_LR=1 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-nms --roofline-pass after 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NO_LR', d['ms_per_step'])"
done
