# usage: bash tools/ab_detail.sh ENTRY VAR=VALUE  -> per-call time table of one entry point, default vs the switch (same box)
E=$1; X=$2
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown --detail $E 2>&1 >/dev/null | grep -E "^fwd |^bwd " | awk '{print $1, $3, $4, $5, $6, $7}' | sed 's/ [0-9.]* ms.*//' > /tmp/ab_a.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown --detail $E 2>&1 >/dev/null | grep -E "^fwd |^bwd " > /tmp/ab_a_full.txt
env $X python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown --detail $E 2>&1 >/dev/null | grep -E "^fwd |^bwd " > /tmp/ab_b_full.txt
python - <<'PY'
import re, collections
def load(p):
    d = collections.defaultdict(list)
    for l in open(p):
        m = re.match(r"(fwd|bwd) \S+\s+(.*?)\s+([0-9.]+) ms", l)
        if m: d[(m.group(1), m.group(2).strip())].append(float(m.group(3)))
    return d
a, b = load("/tmp/ab_a_full.txt"), load("/tmp/ab_b_full.txt")
rows = []
for k in a:
    if k in b: rows.append((sum(a[k]), sum(b[k]), len(a[k]), k))
rows.sort(reverse=True)
print("%-6s %-34s %3s %9s %9s" % ("pass", "shape", "n", "default", "switch"))
for sa, sb, n, k in rows[:40]: print("%-6s %-34s %3d %9.3f %9.3f" % (k[0], k[1], n, sa, sb))
print("total default %.3f switch %.3f" % (sum(r[0] for r in rows), sum(r[1] for r in rows)))
PY
