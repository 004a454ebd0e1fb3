#!/usr/bin/env python3
"""Generate tests/golden/seg_*.npz by running the REAL reference (build container only): `SegLoss` alone on seeded
tensors, and the BDD100K-config network (detection + drivable-area segmentation head) — one train step and one eval
forward on procedural weights.  Outputs are data only.   usage: python tools/gen_golden_seg.py
"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

import torch  # noqa: E402

import gen_golden as GG  # noqa: E402
from oracle import procedural  # noqa: E402

GG._install_stubs()


def seg_maps(n, h, w, c, seed):
    r = np.random.RandomState(seed)
    m = r.rand(n, h, w, c).astype(np.float32)
    m[r.rand(n, h, w, c) < 0.4] = 0.0                     # mostly background, soft edges elsewhere
    return torch.from_numpy(m)


def gen_loss_table():
    from models.seg_loss import SegLoss
    out = {}
    for tag, (n, c, h, w) in {"a": (2, 2, 5, 7), "b": (3, 1, 4, 4)}.items():
        g = torch.Generator().manual_seed(17)
        x = (torch.randn(n, c, h, w, generator=g) * 2).requires_grad_(True)
        t = seg_maps(n, h, w, c, 3)
        if tag == "b":
            t = torch.zeros_like(t)                        # no "object" pixel: mean of an empty selection (NaN)
        loss, obj, noobj = SegLoss(c)(x, t)
        loss.backward()
        out.update({"x_" + tag: GG._np(x), "t_" + tag: GG._np(t), "res_" + tag: np.array([loss.item(), obj, noobj], np.float64),
                    "dx_" + tag: GG._np(x.grad)})
    with torch.no_grad():
        out["eval_a"] = SegLoss(2)(torch.from_numpy(out["x_a"]))
    GG._save("seg_loss.npz", **out)


def gen_net():
    man = json.load(open(os.path.join(OUT, "state_keys_bdd100k.json")))
    cfg = man["config"]
    m = GG._ref_model(cfg)
    procedural.fill_state_dict_(m)
    import models.seg_loss as SL
    import models.yolo_loss as YL
    import utils.box as UB
    UB.device = YL.device = SL.device = torch.device("cpu")
    grabbed = {}
    m.seg_headS16.register_forward_hook(lambda mod, i, o: grabbed.__setitem__("out2", o))
    nc = cfg["yolo"]["num_classes"]

    m.train()
    x = procedural.images(4, 128, 128, seed=21)
    tg = procedural.targets(4, num_classes=nc, seed=6, empty_every=4)
    sm = seg_maps(4, 8, 8, cfg["seg"]["num_classes"], 9)
    res, seg_out = m(x, [t.clone() for t in tg], sm)
    loss = sum(r[0] for r in res) + seg_out[0]
    loss.backward()
    names, gnorm = [], []
    for k, p in m.named_parameters():
        names.append(k)
        gnorm.append(-1.0 if p.grad is None else p.grad.double().norm().item())
    tr = {"tuple0": np.array([float(v) for v in res[0]]), "tuple1": np.array([float(v) for v in res[1]]),
          "seg_out": np.array([float(seg_out[0]), seg_out[1], seg_out[2]]), "out2": GG._np(grabbed["out2"]), "seg_maps": GG._np(sm),
          "gnorm": np.array(gnorm), "t_counts": np.array([len(t) for t in tg]), "t_all": GG._np(torch.cat(tg)),
          "g_seghead_w": GG._np(m.seg_headS16[3].weight.grad), "g_seghead_b": GG._np(m.seg_headS16[3].bias.grad),
          "g_segconv_dw": GG._np(m.seg_conv_for_S16[0].conv.weight.grad), "g_stem": GG._np(m.backbone.features[0][0].weight.grad)}
    GG._save("seg_net_train.npz", **tr)
    with open(os.path.join(OUT, "seg_net_names.json"), "w") as f:
        json.dump({"params": names, "grad_none": [n for n, g in zip(names, gnorm) if g < 0]}, f)

    procedural.fill_state_dict_(m)                         # the train step moved the BN running statistics
    m.eval()
    for l in m.yolo_losses:
        l.val_conf = 0.3
    x = procedural.images(2, 96, 96, seed=22)
    with torch.no_grad():
        det, seg = m(x)
    GG._save("seg_net_eval.npz", seg=np.asarray(seg), out2=GG._np(grabbed["out2"]), det_counts=np.array([len(d) for d in det]))


if __name__ == "__main__":
    gen_loss_table()
    gen_net()
