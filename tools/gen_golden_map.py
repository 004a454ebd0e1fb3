#!/usr/bin/env python3
"""Generate tests/golden/map_*.npz by running the REAL reference's utils/eval_mAP.py (build container only).

Outputs are data: seeded packed inputs + the reference's `calculate_mAP` results and, for the small case, the
per-(image,class) TP/FP flags of `eval_single_image_recall`.   usage: python tools/gen_golden_map.py
"""
import contextlib
import io
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

import torch  # noqa: E402

import gen_golden  # noqa: E402   (tools/gen_golden.py: the import stubs for progress / cv2 / torchvision)
gen_golden._install_stubs()
from utils import eval_mAP as R  # noqa: E402   (the reference)
from mobilenet_yolo_pytorch_amd import synthetic  # noqa: E402


def run_reference(case, n_classes):
    db, dl, ds, do, tb, tl, td, to = case
    sp = lambda a, off: [torch.from_numpy(a[off[i]:off[i + 1]].copy()) for i in range(len(off) - 1)]
    names = ["background"] + ["c%d" % i for i in range(1, n_classes)]
    with contextlib.redirect_stdout(io.StringIO()):
        aps, m, tp, fp = R.calculate_mAP(sp(db, do), sp(dl, do), sp(ds, do), sp(tb, to), sp(tl, to), sp(td, to), names)
    f = lambda d: np.array([d[n] for n in names[1:]], np.float32)
    return f(aps), np.float32(m), f(tp), f(fp)


def flags(case, n_classes):
    """TP/FP of every detection (packed order), from eval_single_image_recall per (image, class)."""
    db, dl, ds, do, tb, tl, td, to = case
    tpf, fpf = np.zeros(len(dl), np.float32), np.zeros(len(dl), np.float32)
    for i in range(len(do) - 1):
        a, b, c, d = do[i], do[i + 1], to[i], to[i + 1]
        for cls in range(1, n_classes):
            tsel, dsel = torch.from_numpy(tl[c:d] == cls), torch.from_numpy(dl[a:b] == cls)
            tp, fp, _, _ = R.eval_single_image_recall(tsel, dsel, torch.from_numpy(tb[c:d].copy()), torch.from_numpy(td[c:d].copy()),
                                                      torch.from_numpy(db[a:b].copy()), torch.from_numpy(ds[a:b].copy()))
            idx = np.nonzero(dl[a:b] == cls)[0] + a
            tpf[idx], fpf[idx] = tp.numpy(), fp.numpy()
    return tpf, fpf


def save(name, case, n_classes, with_flags):
    ap, m, tp, fp = run_reference(case, n_classes)
    keys = ("det_boxes", "det_labels", "det_scores", "det_off", "true_boxes", "true_labels", "true_diff", "true_off")
    out = dict(zip(keys, case), n_classes=np.int32(n_classes), ap=ap, mean_ap=m, tp=tp, fp=fp)
    if with_flags:
        out["tp_flags"], out["fp_flags"] = flags(case, n_classes)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB  mAP", float(m), "tp", tp.sum(), "fp", fp.sum())


if __name__ == "__main__":
    save("map_small.npz", synthetic.map_case(n_images=12, n_classes=7, seed=3, mean_gt=3.0, crafted=True), 7, True)
    save("map_voc.npz", synthetic.map_case(n_images=80, n_classes=21, seed=5, mean_gt=2.5), 21, False)
    save("map_nodet.npz", synthetic.map_case(n_images=5, n_classes=4, seed=7, mean_gt=2.0, no_detections=True), 4, False)
