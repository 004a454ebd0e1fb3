#!/usr/bin/env python3
"""Caller-side counterpart of inference.py:109-126 / train.py:357-388: eval forward -> decode -> per-class NMS on the
GPU, detections returned as a python list of [k_i,7] = (x1,y1,x2,y2,conf,cls_score,cls_idx) tensors, plus a checkpoint
round trip in the reference's format (train.py:425-433: {'model': state_dict, 'conf': val_conf, ...})."""
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilenet_yolo_pytorch_amd import synthetic, yolo  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = yolo(synthetic.VOC_CONFIG).to(dev)
    buf = io.BytesIO()
    torch.save({"epoch": 0, "model": model.state_dict(), "conf": 0.3}, buf)       # train.py:425-433
    buf.seek(0)
    ck = torch.load(buf)
    model.load_state_dict(ck["model"])                                              # train.py:141
    for h in model.yolo_losses:
        h.val_conf = ck["conf"]                                                     # train.py:149-150, inference.py:46-47
    model.eval()
    for bs in (2, 64):
        x = synthetic.images(bs, 352, 352, seed=3).to(dev)
        det = model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            det = model(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print("bs=%d: %.2f ms / batch (%.0f img/s), detections per image: %s" % (bs, dt * 1e3, bs / dt, [len(d) for d in det][:8]))
        assert len(det) == bs and all(d.shape[1] == 7 for d in det)


if __name__ == "__main__":
    main()
