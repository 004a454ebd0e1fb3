#!/usr/bin/env python3
"""Caller-side counterpart of the reference's validation pass (train.py:test(), :333-424) on synthetic data, everything
after JPEG decode on the GPU:  decoded uint8 photos -> BatchPrep (Resize/ToTensor/Normalize of collate_fn) ->
model.eval() forward (network + decode + per-class NMS) -> Evaluator (VOC07 11-point mAP) -> adjust_confidence.

    python examples/eval_synthetic.py [--images 256] [--batch 64] [--size 352]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilenet_yolo_pytorch_amd import synthetic, yolo  # noqa: E402
from mobilenet_yolo_pytorch_amd.evalmap import Evaluator, adjust_confidence  # noqa: E402
from mobilenet_yolo_pytorch_amd.prep import BatchPrep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=352)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = yolo(synthetic.VOC_CONFIG).to(dev).eval()
    for hs in model.yolo_losses:
        hs.val_conf = 0.3
    classes_name = ["background"] + ["class%d" % i for i in range(1, 21)]          # train.py:57-58
    prep = BatchPrep([(a.size, a.size)], [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], device=dev)
    ev = Evaluator(classes_name)
    r = np.random.RandomState(0)
    t0 = time.perf_counter()
    for b in range(0, a.images, a.batch):
        n = min(a.batch, a.images - b)
        photos = synthetic.photos([(int(r.randint(300, 500)), int(r.randint(300, 500))) for _ in range(n)], seed=b)   # "decoded JPEGs"
        targets = synthetic.targets(n, seed=b + 1, empty_every=8)
        images = prep(photos)                                                      # folder2lmdb.py:223-256 on the device
        ev.add(model(images), targets)                                             # train.py:364-395
    aps, mAP, tp, fp = ev.compute()                                                # train.py:421
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    conf = adjust_confidence(ev.gt_box, ev.pred_box, model.yolo_losses[0].val_conf)   # train.py:417-418
    print("%d images in %.2f s; %d ground-truth boxes, %d detections; mAP %.4f (random weights); next val_conf %.2f" % (
        a.images, dt, ev.gt_box, ev.pred_box, mAP, conf))
    assert len(aps) == 20 and 0.0 <= mAP <= 1.0


if __name__ == "__main__":
    main()
