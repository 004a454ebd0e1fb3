"""Synthetic workload of the headline benchmark: seeded N(0,1) images, 3 boxes per image
(label~U{1..C}, cx,cy = 0.1+0.8U, w,h = 0.05+0.5U, every 16th image empty) and the VOC model config
(values of the reference's models/voc/config.yaml:1-31)."""
import numpy as np
import torch

VOC_CONFIG = {
    "img_h": 352, "img_w": 352,
    "iou_weighting": 0.021830872589525777,
    "yolo": {
        "num_classes": 20, "num_anchors": 3,
        "ignore_thresh": [0.6076333316652263, 0.5623606200028424],
        "iou_thresh": 0.5497280113447018,
        "anchors": [[143, 265], [153, 121], [280, 279], [20, 37], [49, 94], [73, 201]],
        "classes": 20,
        "mask": [[0, 1, 2], [3, 4, 5]],
    },
}


def images(n, h, w, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 3, h, w, generator=g)


def targets(n, num_classes=20, boxes_per_image=3, seed=1, empty_every=16):
    r = np.random.RandomState(seed)
    out = []
    for i in range(n):
        if empty_every and i % empty_every == empty_every - 1:
            out.append(torch.zeros(0, 5))
            continue
        lab = r.randint(1, num_classes + 1, size=(boxes_per_image, 1)).astype(np.float32)
        cxy = (0.1 + 0.8 * r.rand(boxes_per_image, 2)).astype(np.float32)
        wh = (0.05 + 0.5 * r.rand(boxes_per_image, 2)).astype(np.float32)
        out.append(torch.from_numpy(np.concatenate((lab, cxy, wh), 1)))
    return out
