# usage: bash tools/ab_lib_detail.sh ENTRY -> per-call time table of one entry point, tools/ab/libmnyolo_prev.so (MNY_LIB) against the in-tree build (same box)
E=$1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown --detail $E 2>&1 >/dev/null | grep -E "^fwd |^bwd " > /tmp/ab_b_full.txt
MNY_LIB=$GRAFT_REPO_ROOT/tools/ab/libmnyolo_prev.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --breakdown --detail $E 2>&1 >/dev/null | grep -E "^fwd |^bwd " > /tmp/ab_a_full.txt
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
print("%-6s %-34s %3s %9s %9s" % ("pass", "shape", "n", "prev", "new"))
for sa, sb, n, k in rows[:40]: print("%-6s %-34s %3d %9.3f %9.3f" % (k[0], k[1], n, sa, sb))
print("total prev %.3f new %.3f" % (sum(r[0] for r in rows), sum(r[1] for r in rows)))
PY
