# One round of profiling evidence for profiles/ (run through gpurun; program directly after `--`, counters in their own runs).
#   stats : rocprofv3 --kernel-trace --stats              bench.py --steps 10 --warmup 3 --bracket-every 1   (3 + 10 timed + 10 second-pass = 23 steps; every step bracketed = single stream, so a small kernel's duration is not stretched by a side-stream kernel sharing the CUs)
#   fetch / write : --pmc FETCH_SIZE / --pmc WRITE_SIZE   bench.py --steps 2 --warmup 1    (1 + 2 + 3 = 6 steps)
#   mfma  : --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (same short run)
# then, in the build container:  python tools/prof_summary.py gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write 23 6 r02
#                                python tools/prof_mfma.py gpurun_out/prof_mfma 6 r02
set -e
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# one stream in EVERY step of the profiled runs (warm-up and second-pass steps too): with the weight gradients on the side stream a small
# kernel that shares the CUs with one of them is recorded with a stretched duration (bn_bwd_finalize: 5.6 -> 10.3 us average)
export MNY_SIDE_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_stats -o run -- python3 $REPO/bench.py --steps 10 --warmup 3 --bracket-every 1 --no-cpu-baseline --no-nms --full-json > $REPO/gpurun_out/prof_stats.json 2> $REPO/gpurun_out/prof_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/prof_fetch -o run -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nms > /dev/null 2> $REPO/gpurun_out/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/prof_write -o run -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nms > /dev/null 2> $REPO/gpurun_out/prof_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/prof_mfma -o run -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nms > /dev/null 2> $REPO/gpurun_out/prof_mfma.err
ls -la $REPO/gpurun_out/prof_stats $REPO/gpurun_out/prof_fetch $REPO/gpurun_out/prof_mfma | head -30
