# configs[3] (MobileNetV3-YOLO 512x512 bs 64, bf16 storage): kernel stats + HBM counters, each in its own run (program directly after `--`)
#   then, in the build container:  python tools/prof_c3_summary.py gpurun_out/prof_c3 gpurun_out/prof_c3_fetch gpurun_out/prof_c3_write 23 6 r04   (3 warm-up + 10 timed + 10 second-pass steps = 23; 1 + 2 + 3 = 6)
set -e
REPO=$GRAFT_REPO_ROOT
C3="--arch mbv3 --size 512 --batch 64 --dtype bf16 --no-cpu-baseline --no-nms"
cd /tmp && export TMPDIR=/tmp
# one stream in EVERY step of the profiled runs (warm-up and second-pass steps too): with the weight gradients on the side stream a small
# kernel that shares the CUs with one of them is recorded with a stretched duration (bn_bwd_finalize: 5.6 -> 10.3 us average)
export MNY_SIDE_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_c3 -o run -- python3 $REPO/bench.py $C3 --steps 10 --warmup 3 --bracket-every 1 --full-json > $REPO/gpurun_out/prof_c3.json 2> $REPO/gpurun_out/prof_c3.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/prof_c3_fetch -o run -- python3 $REPO/bench.py $C3 --steps 2 --warmup 1 > /dev/null 2> $REPO/gpurun_out/prof_c3_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/prof_c3_write -o run -- python3 $REPO/bench.py $C3 --steps 2 --warmup 1 > /dev/null 2> $REPO/gpurun_out/prof_c3_write.err
ls $REPO/gpurun_out/prof_c3 $REPO/gpurun_out/prof_c3_fetch | head
