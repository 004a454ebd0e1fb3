"""Build libmnyolo.so (gfx950) in-tree with hipcc.  `python -m mobilenet_yolo_pytorch_amd.build`
or `build()`; objects are rebuilt only when their source (or common.h / mnyolo.h) is newer."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmnyolo.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
BASE_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# detect.hip must round like the CPU kernels it replaces (bit-exact NMS): no FMA contraction there
EXTRA = {"detect.hip": ["-ffp-contract=off"], "evalmap.hip": ["-ffp-contract=off"], "prep.hip": ["-ffp-contract=off"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "x6.h"), os.path.join(HERE, "..", "include", "mnyolo.h")]
    jobs, objs = [], []
    for src in sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + BASE_FLAGS + EXTRA.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
