#!/bin/bash
# kernel-level view of the split count next to the wide weight-gradient tile (bracketed single-stream steps)
mkdir -p gpurun_out/r6
{
for v in 1536 1024 2048 768; do
echo "== MNY_WG_BLOCKS=$v"; MNY_WG_BLOCKS=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown --detail mny_pw_wgrad 2>&1 >/dev/null | grep -E "mny_pw_wgrad  |reduce_batch|M123904 K512 N512|K1280 N512|K512 N1024|K960 N320" | head -9
done
} > gpurun_out/r6/wgwide3.txt 2>&1
