set -e
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_stats -o run -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms > $REPO/gpurun_out/prof_stats.json 2> $REPO/gpurun_out/prof_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/prof_fetch -o run -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nms > /dev/null 2> $REPO/gpurun_out/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/prof_write -o run -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nms > /dev/null 2> $REPO/gpurun_out/prof_write.err
ls -la $REPO/gpurun_out/prof_stats $REPO/gpurun_out/prof_fetch | head -20
