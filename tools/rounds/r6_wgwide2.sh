#!/bin/bash
# kernel-level view of the wide weight-gradient tile: per-entry-point table (bracketed single-stream steps) for both settings, twice
mkdir -p gpurun_out/r6
{
for rep in 1 2; do for v in 0 1; do
echo "== MNY_WG_TJ4=$v"; MNY_WG_TJ4=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown --detail mny_pw_wgrad 2>&1 >/dev/null | grep -E "mny_pw_wgrad|M[0-9]+ K(512|1280|1024|256)" | head -24
done; done
} > gpurun_out/r6/wgwide2.txt 2>&1
