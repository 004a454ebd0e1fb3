"""Timing experiment ONLY (upper bounds of "what if these launches were free"): run bench.py with the entry points named in
SKIP_CALLS left out of every replay after the first 8.  The results of such a step are STALE / WRONG by construction — this lives
here, outside the product engine, and patches CallList.run in this process only.

usage: SKIP_CALLS=mny_bn_finalize,mny_bn_bwd_finalize python tools/experiments/skip_calls.py [bench.py arguments]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mobilenet_yolo_pytorch_amd import _lib, engine

SKIP = tuple(v for v in os.environ.get("SKIP_CALLS", "").split(",") if v)
runs = [0]


def run(self, begin=0, end=None):
    lib = _lib.load()
    runs[0] += 1
    for fn, args, name, _ in self.calls[begin:end]:
        if runs[0] > 8 and name in SKIP:
            continue
        rc = fn(*args)
        if rc:
            raise engine.MnyError("%s failed (%d): %s" % (name, rc, lib.mny_last_error().decode()))


if SKIP:
    sys.stderr.write("skip_calls: %s are NOT launched after warm-up: results are stale, timing experiment only\n" % (SKIP,))
    engine.CallList.run = run
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
