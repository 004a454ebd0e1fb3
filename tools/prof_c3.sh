set -e
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_c3 -o run -- python3 $REPO/bench.py --arch mbv3 --size 512 --batch 64 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-nms > $REPO/gpurun_out/prof_c3.json 2> $REPO/gpurun_out/prof_c3.err
ls $REPO/gpurun_out/prof_c3
