#!/usr/bin/env python3
"""Generate tests/golden/* by importing the REAL reference from /root/reference.

Runs only in the build container (the reference does not exist on the GPU box).
Nothing from the reference is copied: the outputs are data — seeded inputs and
the reference's numeric outputs on them.

Import recipe: three stub modules (`progress.bar`, `cv2`,
`torchvision.ops.nms`) written to a temp dir, the pretrained-download factory
patched to build an un-initialised backbone, `yaml.safe_load` for configs.
`torchvision.ops.nms` is NOT available anywhere (unpinned third-party dep), so
the stub routes to oracle/nms_ref.c: fixtures that pass through it pin the
*driver* of utils/box.py, not the inner kernel (parity unpinned there).

usage: python tools/gen_golden.py   (from the repo root)
"""
import json
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

import torch  # noqa: E402
import yaml  # noqa: E402

from oracle import nms_ref, procedural  # noqa: E402


def _install_stubs():
    d = tempfile.mkdtemp(prefix="mny_stubs_")
    os.makedirs(os.path.join(d, "progress"))
    open(os.path.join(d, "progress", "__init__.py"), "w").close()
    with open(os.path.join(d, "progress", "bar.py"), "w") as f:
        f.write("class Bar:\n    def __init__(self,*a,**k): pass\nclass IncrementalBar(Bar): pass\n")
    with open(os.path.join(d, "cv2.py"), "w") as f:
        f.write("")
    os.makedirs(os.path.join(d, "torchvision"))
    with open(os.path.join(d, "torchvision", "__init__.py"), "w") as f:
        f.write("from . import ops\n")
    with open(os.path.join(d, "torchvision", "ops.py"), "w") as f:
        f.write("from oracle.nms_ref import nms\n")
    sys.path.insert(0, REF)
    sys.path.insert(0, d)


def _ref_model(cfg):
    import models.mbv2_yolo as M
    import models.mobilenetv2 as B
    M.mobilenetv2 = lambda *_a, **_k: B.MobileNetV2()      # skip the download (Q11)
    torch.manual_seed(0)
    return M.yolo(cfg)


def _np(t):
    return t.detach().cpu().numpy()


def _save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def crafted_targets():
    """bs=4 targets covering the cases of SURVEY §8c F3 (VOC anchors, 352 input)."""
    a = np.array(procedural.VOC_CONFIG["yolo"]["anchors"], dtype=np.float32) / 352.0
    t0 = torch.zeros(0, 5)                                              # empty image
    t1 = torch.tensor([[3, 0.52, 0.48, a[4][0], a[4][1]],              # two GT in one cell (both grids)
                       [7, 0.53, 0.49, a[4][0] * 1.05, a[4][1] * 0.95],
                       [12, 0.20, 0.80, a[1][0], a[1][1]]])             # best anchor lives in head 0
    t2 = torch.tensor([[20, 0.999, 0.999, 0.10, 0.12],                  # coordinate at the edge
                       [1, 0.31, 0.62, (a[3][0] + a[4][0]) / 2, (a[3][1] + a[4][1]) / 2],   # multi-anchor hit
                       [5, 0.75, 0.25, (a[0][0] + a[2][0]) / 2, (a[0][1] + a[2][1]) / 2]])
    t3 = torch.tensor([[9, 0.1, 0.1, 0.6, 0.6], [9, 0.9, 0.12, 0.07, 0.3],
                       [15, 0.5, 0.5, a[5][0], a[5][1]], [2, 0.45, 0.55, a[0][0], a[0][1]]])
    return [t.float() for t in (t0, t1, t2, t3)]


def gen_state_keys():
    for name in ("voc", "bdd100k"):
        cfg = yaml.safe_load(open(os.path.join(REF, "models", name, "config.yaml")))
        m = _ref_model(cfg)
        keys = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        with open(os.path.join(OUT, "state_keys_%s.json" % name), "w") as f:
            json.dump({"config": cfg, "keys": keys,
                       "num_params": sum(p.numel() for p in m.parameters())}, f)
        print(name, len(keys), "keys")


def gen_iou_tables():
    from utils.iou import find_jaccard_overlap
    from models.yolo_loss import YOLOLoss
    r = np.random.RandomState(3)
    a = r.rand(9, 4).astype(np.float32)
    a[:, 2:] = a[:, :2] + r.rand(9, 2).astype(np.float32) * 0.5
    b = r.rand(7, 4).astype(np.float32)
    b[:, 2:] = b[:, :2] + r.rand(7, 2).astype(np.float32) * 0.5
    a[0] = [0.2, 0.2, 0.2, 0.2]                       # degenerate
    b[0] = [0.2, 0.2, 0.2, 0.2]                       # 0/0 -> NaN
    b[1] = a[1]                                       # identical
    iou = find_jaccard_overlap(torch.from_numpy(a), torch.from_numpy(b))
    L = YOLOLoss(procedural.VOC_CONFIG["yolo"]["anchors"], [0, 1, 2], 20, [352, 352], 0.6, 0.55)
    ci = np.zeros((9, 7, 2), np.float32)
    for i in range(9):
        for j in range(7):
            x, y = L.box_ciou(torch.from_numpy(a[i:i + 1]), torch.from_numpy(b[j:j + 1]))
            ci[i, j] = [x.item(), y.item()]
    # Q1: weighted_mse_loss broadcast [5,1] x [5]
    x = torch.tensor([[0.1], [0.5], [0.9], [0.3], [0.7]])
    w = torch.tensor([1.2, 1.9, 1.5, 1.1, 1.8])
    q1 = L.weighted_mse_loss(x, torch.ones_like(x), w)
    _save("iou_tables.npz", a=a, b=b, iou=_np(iou), ciou=ci, q1_x=_np(x), q1_w=_np(w), q1=_np(q1))


def gen_loss_and_decode():
    from models.yolo_loss import YOLOLoss
    cfg = procedural.VOC_CONFIG
    y = cfg["yolo"]
    tg = crafted_targets()
    out = {"t_counts": np.array([len(t) for t in tg]), "t_all": _np(torch.cat(tg))}
    for hi, g in enumerate((11, 22)):
        L = YOLOLoss(y["anchors"], y["mask"][hi], 20, [352, 352], y["ignore_thresh"][hi],
                     y["iou_thresh"], iou_weighting=cfg["iou_weighting"])
        L.img_size = [352, 352]
        gen = torch.Generator().manual_seed(100 + hi)
        head = (torch.randn(4, 75, g, g, generator=gen) * 0.6).requires_grad_(True)
        res = L(head, [t.clone() for t in tg])
        res[0].backward()
        out["head%d" % hi] = _np(head)
        out["grad%d" % hi] = _np(head.grad)
        out["tuple%d" % hi] = np.array([float(v) for v in res], dtype=np.float64)
        for vc in (0.1, 0.3, 0.5):
            L.val_conf = vc
            with torch.no_grad():
                rows = L(head.detach())
            out["dec%d_%d_counts" % (hi, int(vc * 10))] = np.array([len(r) for r in rows])
            out["dec%d_%d_rows" % (hi, int(vc * 10))] = _np(torch.cat(rows)) if sum(len(r) for r in rows) else np.zeros((0, 7), np.float32)
    _save("loss_decode.npz", **out)

    # utils/box.py driver over the decoded rows (val_conf 0.3), restated kernel underneath
    from utils.box import nms as ref_nms_driver
    import utils.box as UB
    UB.device = torch.device("cpu")
    preds = []
    for hi in range(2):
        cnt = out["dec%d_3_counts" % hi]
        rows = torch.from_numpy(out["dec%d_3_rows" % hi])
        preds.append(list(torch.split(rows, cnt.tolist())))
    kept = ref_nms_driver(tuple(preds), 20)
    _save("nms_driver.npz", counts=np.array([len(k) for k in kept]),
          rows=_np(torch.cat(kept)) if sum(len(k) for k in kept) else np.zeros((0, 7), np.float32))


def gen_net():
    cfg = yaml.safe_load(open(os.path.join(REF, "models", "voc", "config.yaml")))
    m = _ref_model(cfg)
    procedural.fill_state_dict_(m)

    # eval forward: raw heads through a hook (the reference only returns post-NMS rows)
    grabbed = {}
    m.yolo_headS32.register_forward_hook(lambda mod, i, o: grabbed.__setitem__("out0", o))
    m.yolo_headS16.register_forward_hook(lambda mod, i, o: grabbed.__setitem__("out1", o))
    m.backbone.register_forward_hook(lambda mod, i, o: grabbed.__setitem__("feat", o))
    m.eval()
    for l in m.yolo_losses:
        l.val_conf = 0.3
    import utils.box as UB
    import models.yolo_loss as YL
    UB.device = YL.device = torch.device("cpu")
    ev = {}
    for tag, (n, s) in {"a": (2, 96), "b": (1, 352)}.items():
        x = procedural.images(n, s, s, seed=10)
        with torch.no_grad():
            det = m(x)
        ev["out0_" + tag] = _np(grabbed["out0"])
        ev["out1_" + tag] = _np(grabbed["out1"])
        ev["f1sum_" + tag] = np.array([grabbed["feat"][0].double().sum().item(), grabbed["feat"][0].double().abs().sum().item()])
        ev["f2sum_" + tag] = np.array([grabbed["feat"][1].double().sum().item(), grabbed["feat"][1].double().abs().sum().item()])
        ev["det_counts_" + tag] = np.array([len(d) for d in det])
        ev["det_rows_" + tag] = _np(torch.cat(det)) if sum(len(d) for d in det) else np.zeros((0, 7), np.float32)
    _save("net_eval.npz", **ev)

    # one train step, bs=4 @128x128 (grids 4 and 8)
    m.train()
    x = procedural.images(4, 128, 128, seed=11)
    tg = procedural.targets(4, seed=5, empty_every=4)
    res = m(x, [t.clone() for t in tg], None)
    loss = sum(r[0] for r in res)
    loss.backward()
    names, gnorm, gnone = [], [], []
    for k, p in m.named_parameters():
        names.append(k)
        if p.grad is None:
            gnone.append(k)
            gnorm.append(-1.0)
        else:
            gnorm.append(p.grad.double().norm().item())
    tr = {"tuple0": np.array([float(v) for v in res[0]]), "tuple1": np.array([float(v) for v in res[1]]),
          "gnorm": np.array(gnorm), "out0": _np(grabbed["out0"]), "out1": _np(grabbed["out1"]),
          "t_counts": np.array([len(t) for t in tg]), "t_all": _np(torch.cat(tg)),
          "g_stem": _np(m.backbone.features[0][0].weight.grad),
          "g_head16_w": _np(m.yolo_headS16[3].weight.grad), "g_head32_b": _np(m.yolo_headS32[3].bias.grad),
          "g_f5_dw": _np(m.backbone.features[5].conv[3].weight.grad),
          "g_f5_bn": _np(m.backbone.features[5].conv[1].weight.grad)}
    sd = m.state_dict()
    rs_names = [k for k in sd if k.endswith("running_mean") or k.endswith("running_var")]
    tr["rs_norm"] = np.array([sd[k].double().norm().item() for k in rs_names])
    _save("net_train.npz", **tr)
    with open(os.path.join(OUT, "net_train_names.json"), "w") as f:
        json.dump({"params": names, "grad_none": gnone, "running": rs_names}, f)


def _ref_model_v3(cfg):
    """mbv3_yolo.py imports `models.voc.*` (non-existent) and loads a local weight file: alias + patch."""
    import types
    import models.mobilenetv3 as B3
    import models.yolo_loss as YL
    voc = types.ModuleType("models.voc")
    sys.modules["models.voc"] = voc
    sys.modules["models.voc.mobilenetv3"] = B3
    sys.modules["models.voc.yolo_loss"] = YL
    B3.MobileNetV3 = lambda *_a, **_k: B3.MobileNetV3_Large()
    import models.mbv3_yolo as M3
    M3.MobileNetV3 = B3.MobileNetV3
    torch.manual_seed(0)
    return M3.yolo(cfg)


def gen_net_v3():
    cfg = yaml.safe_load(open(os.path.join(REF, "models", "voc", "config.yaml")))
    m = _ref_model_v3(cfg)
    keys = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    with open(os.path.join(OUT, "state_keys_mbv3.json"), "w") as f:
        json.dump({"config": cfg, "keys": keys, "num_params": sum(p.numel() for p in m.parameters())}, f)
    print("mbv3", len(keys), "keys")
    procedural.fill_state_dict_(m)
    grabbed = {}
    m.yolo_headS32.register_forward_hook(lambda mod, i, o: grabbed.__setitem__("out0", o))
    m.yolo_headS16.register_forward_hook(lambda mod, i, o: grabbed.__setitem__("out1", o))
    import utils.box as UB
    import models.yolo_loss as YL
    UB.device = YL.device = torch.device("cpu")
    m.eval()
    for l in m.yolo_losses:
        l.val_conf = 0.3
    x = procedural.images(2, 128, 128, seed=20)
    with torch.no_grad():
        det = m(x)
    out = {"ev_out0": _np(grabbed["out0"]), "ev_out1": _np(grabbed["out1"]), "ev_counts": np.array([len(d) for d in det])}
    m.train()
    x = procedural.images(2, 128, 128, seed=21)
    tg = procedural.targets(2, seed=6, empty_every=0)
    res = m(x, [t.clone() for t in tg])
    sum(r[0] for r in res).backward()
    names = [k for k, _ in m.named_parameters()]
    out.update({"tr_out0": _np(grabbed["out0"]), "tr_out1": _np(grabbed["out1"]),
                "tuple0": np.array([float(v) for v in res[0]]), "tuple1": np.array([float(v) for v in res[1]]),
                "gnorm": np.array([p.grad.double().norm().item() for _, p in m.named_parameters()]),
                "t_counts": np.array([len(t) for t in tg]), "t_all": _np(torch.cat(tg)),
                "g_shared_dw": _np(m.connect_for_S16.conv[0].conv.weight.grad),
                "g_gate": _np(m.backbone.bneck[3].se.se[3].weight.grad)})
    sd = m.state_dict()
    rs_names = [k for k in sd if k.endswith("running_mean") or k.endswith("running_var")]
    out["rs_norm"] = np.array([sd[k].double().norm().item() for k in rs_names])
    _save("net_v3.npz", **out)
    with open(os.path.join(OUT, "net_v3_names.json"), "w") as f:
        json.dump({"params": names, "running": rs_names}, f)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    _install_stubs()
    if "--only-v3" not in sys.argv:
        gen_state_keys(); gen_iou_tables(); gen_loss_and_decode(); gen_net()
    gen_net_v3()
