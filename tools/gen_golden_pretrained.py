#!/usr/bin/env python3
"""Pin `yolo.load_pretrained_backbone` to the REAL loader, models/mobilenetv2.py:161-181 (build container only).

The reference downloads d-li14's `mobilenetv2-c5e733a8.pth` (mbv2_yolo.py:116); there is no network, so the download call is
replaced by a locally built state dict with that checkpoint's KEY LAYOUT (`synthetic.dli14_mobilenetv2_keys`: features.0-17,
conv.0/1, classifier — written from the architecture table, not from the reference's remap) in which every tensor is filled
with its own ordinal.  After the reference's `mobilenetv2(url)` ran, the ordinal found in each backbone tensor says which
checkpoint key landed there.  Output: tests/golden/pretrained_map.json = [[backbone key, checkpoint key | null], ...].
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

import torch  # noqa: E402

import gen_golden  # noqa: E402  (import stubs for the reference)
from mobilenet_yolo_pytorch_amd import synthetic  # noqa: E402


def main():
    gen_golden._install_stubs()
    import models.mobilenetv2 as B
    spec = synthetic.dli14_mobilenetv2_keys()
    ckpt, names = {}, []
    for i, (k, shape) in enumerate(spec):
        if i % 7 == 3:
            k = "module." + k                                   # the loader strips DataParallel prefixes (:169)
        ckpt[k] = torch.full(shape, float(i + 1))
        names.append(k)
    B.load_state_dict_from_url = lambda *_a, **_k: ckpt
    torch.manual_seed(0)
    model = B.mobilenetv2("local://no-download")
    out = []
    for k, v in model.state_dict().items():
        val = v.double().flatten()
        first = float(val[0]) if val.numel() else 0.0
        hit = None
        if val.numel() and bool((val == first).all()) and first == int(first) and 1 <= int(first) <= len(names) \
                and tuple(ckpt[names[int(first) - 1]].shape) == tuple(v.shape) and (v.numel() > 1 or "num_batches" in k):
            hit = names[int(first) - 1]
        out.append([k, hit])
    path = os.path.join(REPO, "tests", "golden", "pretrained_map.json")
    json.dump({"spec_len": len(spec), "map": out}, open(path, "w"), indent=0)
    print("wrote", path, sum(1 for _k, h in out if h), "of", len(out), "keys loaded")


if __name__ == "__main__":
    main()
