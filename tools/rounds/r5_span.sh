# usage: bash tools/rounds/r5_span.sh TAG "ENV=.. ENV=.." [c1|c3]: kernel-trace of the un-bracketed, single-stream bench -> span / sum of durations / gaps per step
TAG=$1; ENVS="$2"; WHICH=${3:-c3}
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export MNY_SIDE_STREAM=0
for e in $ENVS; do export $e; done
if [ "$WHICH" = "c3" ]; then ARGS="--arch mbv3 --size 512 --batch 64 --dtype bf16"; else ARGS=""; fi
rm -rf $REPO/gpurun_out/r5/span_$TAG
rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/r5/span_$TAG -o run -- python3 $REPO/bench.py $ARGS --no-cpu-baseline --no-nms --steps 8 --warmup 3 --roofline-pass after > /dev/null 2> $REPO/gpurun_out/r5/span_$TAG.err
python3 - "$TAG" <<'PY'
import csv, glob, os, sys, collections
tag = sys.argv[1]
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5/span_%s/**/*kernel_trace.csv" % tag, recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
# step boundaries: the stem kernel of the forward pass (stem_tile_kernel<.., 0>) starts a step
idx = [i for i, r in enumerate(rows) if "stem_tile_kernel" in r[2] and ", 0>" in r[2]]
idx = idx[-7:]
steps = len(idx) - 1
sel = rows[idx[0]:idx[-1]]
dur = sum(e - s for s, e, _ in sel)
span = rows[idx[-1]][0] - rows[idx[0]][0]
print("%s: %d steps, %.1f kernels/step, span %.3f ms/step, durations %.3f ms/step, gaps %.3f ms/step" % (tag, steps, len(sel) / steps, span / steps / 1e6, dur / steps / 1e6, (span - dur) / steps / 1e6))
PY
rm -rf $REPO/gpurun_out/r5/span_$TAG
