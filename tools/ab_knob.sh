# usage: bash tools/ab_knob.sh ENTRY VAR v1 v2 ...   -> per-value ms/step of one entry point and the step
E=$1; V=$2; shift 2
for v in default "$@"; do
  if [ "$v" = default ]; then X="X=1"; else X="$V=$v"; fi
  r=$(env $X python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown 2>&1 >/dev/null | grep "^$E \|^sum of")
  echo "$V=$v: $(echo "$r" | awk '{printf "%s %s | ", $2, $NF=="ms/step" ? "" : $(NF-1)}')"
done
