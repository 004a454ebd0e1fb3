#!/usr/bin/env python3
"""Generate tests/golden/prep_*.npz with the real third-party pieces of the reference's collate_fn
(folder2lmdb.py:223-256) run in the build container: Pillow's `Image.resize(..., BILINEAR)` (what
torchvision.transforms.Resize does to a PIL image; torchvision itself is not installed anywhere), then the tensor ops of
ToTensor / Normalize in torch.  Outputs are data: seeded uint8 images, the resized uint8 images and the normalised
float batch.   usage: python tools/gen_golden_prep.py
"""
import os
import sys

import numpy as np
import torch
from PIL import Image

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
from mobilenet_yolo_pytorch_amd import synthetic  # noqa: E402

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]          # models/voc/config.yaml:13-15


def reference_collate(images, size):
    out = []
    for im in images:
        r = Image.fromarray(im).resize((size[1], size[0]), Image.BILINEAR)       # transforms.Resize(size=(h,w), interpolation=BILINEAR)
        t = torch.from_numpy(np.asarray(r).copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)   # ToTensor
        mean, std = torch.as_tensor(MEAN, dtype=t.dtype), torch.as_tensor(STD, dtype=t.dtype)
        t.sub_(mean[:, None, None]).div_(std[:, None, None])                      # Normalize
        out.append((np.asarray(r).copy(), t))
    return [o[0] for o in out], torch.stack([o[1] for o in out]).numpy()


if __name__ == "__main__":
    print("Pillow", Image.__version__ if hasattr(Image, "__version__") else "")
    for name, sizes, size in (("prep_small.npz", [(40, 56), (64, 64), (90, 33), (64, 100), (17, 64), (200, 180)], (64, 64)),
                              ("prep_rect.npz", [(50, 70), (96, 32), (31, 47)], (32, 96))):
        imgs = synthetic.photos(sizes, seed=3)
        u8, batch = reference_collate(imgs, size)
        arrs = {"mean": np.array(MEAN, np.float32), "std": np.array(STD, np.float32), "size": np.array(size), "batch": batch}
        for i, (a, b) in enumerate(zip(imgs, u8)):
            arrs["img%d" % i], arrs["u8_%d" % i] = a, b
        path = os.path.join(OUT, name)
        np.savez_compressed(path, **arrs)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")
