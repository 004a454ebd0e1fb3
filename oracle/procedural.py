"""Oracle (test infrastructure, NOT product): deterministic, name-keyed tensor fill.

Both sides of a parity check (the real reference inside tools/gen_golden.py, the
oracle, and the HIP product in tests) regenerate identical weights from tensor
NAMES, so the 19.7 MB parameter set never has to be committed.
"""
import zlib

import numpy as np
import torch


def _rng(name, salt):
    return np.random.RandomState((zlib.crc32(name.encode()) ^ salt) & 0x7FFFFFFF)


def fill_state_dict_(module, salt=0):
    """Overwrite every parameter/buffer of `module` in place.

    conv weight  : N(0, sqrt(2/fan_in)) — keeps activations O(1) through 50+ layers
    conv bias    : U(-0.1, 0.1)
    BN weight    : U(0.8, 1.2)      BN bias       : U(-0.2, 0.2)
    running_mean : U(-0.1, 0.1)     running_var   : U(0.8, 1.2)
    num_batches_tracked untouched.
    """
    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            r = _rng(name, salt)
            if name.endswith("num_batches_tracked"):
                continue
            if t.dim() == 4:
                fan_in = t.shape[1] * t.shape[2] * t.shape[3]
                v = r.standard_normal(t.shape) * np.sqrt(2.0 / fan_in)
            elif name.endswith("running_var"):
                v = r.uniform(0.8, 1.2, t.shape)
            elif name.endswith("running_mean"):
                v = r.uniform(-0.1, 0.1, t.shape)
            elif name.endswith(".bias") and t.dim() == 1 and (name.replace(".bias", ".weight") in sd
                                                             and sd[name.replace(".bias", ".weight")].dim() == 4):
                v = r.uniform(-0.1, 0.1, t.shape)
            elif name.endswith(".weight"):
                v = r.uniform(0.8, 1.2, t.shape)
            else:
                v = r.uniform(-0.2, 0.2, t.shape)
            t.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
    return module


def images(n, h, w, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 3, h, w, generator=g)


def targets(n, num_classes=20, boxes_per_image=3, seed=1, empty_every=16):
    """SURVEY §8d synthetic targets: label~U{1..C}, cx,cy=0.1+0.8U, w,h=0.05+0.5U;
    every `empty_every`-th image has no boxes."""
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
