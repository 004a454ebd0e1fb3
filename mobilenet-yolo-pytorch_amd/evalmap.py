"""Evaluation consumer of the detection path: VOC07 11-point mAP on the device.

Host-side mirror of the reference's interface — `calculate_mAP` keeps the name, argument list and return
value of utils/eval_mAP.py:134-187, `adjust_confidence` is train.py:434-440 — over `mny_map_eval`
(csrc/evalmap.hip).  `Evaluator` is the loop body of train.py:test() (:359-395) without the per-image Python:
detections stay in the packed [k,7] rows the NMS stage wrote and are converted by `mny_eval_pack`.
There is no CPU fallback: without libmnyolo.so every call raises MnyError.
"""
import ctypes

import torch

from ._lib import call, query


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() else None


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t):
    return t.detach().to(dtype=torch.float32).contiguous()


def map_eval(det_boxes, det_labels, det_scores, det_off, true_boxes, true_labels, true_diff, true_off, n_classes):
    """Packed inputs on one CUDA device (float32; offsets int32 [n_images+1]); n_classes counts the background entry.
    -> dict of device tensors: ap [n-1], tp [n-1], fp [n-1], prec11 [n-1,11], mean_ap [1].  No host sync."""
    dev = det_off.device
    if dev.type != "cuda":
        raise RuntimeError("map_eval needs CUDA tensors (the HIP path is the only path)")
    D, T, n_img = int(det_labels.numel()), int(true_labels.numel()), int(det_off.numel()) - 1
    args = [_f32(det_boxes), _f32(det_labels), _f32(det_scores), det_off.to(torch.int32).contiguous(),
            _f32(true_boxes), _f32(true_labels), _f32(true_diff), true_off.to(torch.int32).contiguous()]
    if true_off.numel() != n_img + 1:
        raise ValueError("det_off and true_off describe different numbers of images")
    nb = query("mny_map_ws_bytes", D, T)
    ws = torch.empty(max(int(nb), 256), device=dev, dtype=torch.uint8)
    out = torch.empty(14 * (n_classes - 1) + 1, device=dev, dtype=torch.float32)
    n = n_classes - 1
    ap, tp, fp, p11, mean = out[:n], out[n:2 * n], out[2 * n:3 * n], out[3 * n:14 * n], out[14 * n:]
    call("mny_map_eval", *[_p(a) for a in args], n_img, D, T, int(n_classes), _p(ap), _p(tp), _p(fp), _p(p11), _p(mean),
         ctypes.c_void_p(ws.data_ptr()), _st())
    return {"ap": ap, "tp": tp, "fp": fp, "prec11": p11.view(n, 11), "mean_ap": mean, "_keepalive": (args, ws)}


def _pack_list(ts, width, dev):
    ts = [t.reshape(-1, width) if width else t.reshape(-1) for t in ts]
    off = torch.tensor([0] + [int(t.shape[0]) for t in ts], dtype=torch.int64).cumsum(0).to(torch.int32)
    if ts:
        flat = torch.cat([t.to(dev, torch.float32) for t in ts])
    else:
        flat = torch.zeros((0, width) if width else (0,), device=dev)
    return flat, off.to(dev)


def calculate_mAP(det_boxes, det_labels, det_scores, true_boxes, true_labels, true_difficulties, classes_name, device=None):
    """Drop-in for utils/eval_mAP.py:134-187: lists with one tensor per image, `classes_name` with 'background' first.
    -> (average_precisions {name: AP}, mean_average_precision, class_true_positive {name: n}, class_false_positive {name: n})."""
    assert len(det_boxes) == len(det_labels) == len(det_scores) == len(true_boxes) == len(true_labels) == len(true_difficulties)
    dev = torch.device(device) if device is not None else next((t.device for t in list(det_boxes) + list(true_boxes) if t.is_cuda), torch.device("cuda:0"))
    db, do = _pack_list(det_boxes, 4, dev)
    dl, _ = _pack_list(det_labels, 0, dev)
    ds, _ = _pack_list(det_scores, 0, dev)
    tb, to = _pack_list(true_boxes, 4, dev)
    tl, _ = _pack_list(true_labels, 0, dev)
    td, _ = _pack_list(true_difficulties, 0, dev)
    r = map_eval(db, dl, ds, do, tb, tl, td, to, len(classes_name))
    ap, tp, fp, m = r["ap"].tolist(), r["tp"].tolist(), r["fp"].tolist(), float(r["mean_ap"])
    names = list(classes_name)[1:]
    return dict(zip(names, ap)), m, dict(zip(names, tp)), dict(zip(names, fp))


def adjust_confidence(gt_box_num, pred_box_num, conf):
    """train.py:434-440 (host logic)."""
    if pred_box_num > gt_box_num * 3:
        conf = conf + 0.01
    elif pred_box_num < gt_box_num * 2 and conf > 0.01:
        conf = conf - 0.01
    return conf


class Evaluator:
    """Accumulates an evaluation set on the device and scores it once (train.py:359-421).

        ev = Evaluator(classes_name)
        for images, targets in loader:
            ev.add(model(images), targets)           # list of [k_i,7] rows per image, list of [t_i,5] targets
        aps, mAP, tp, fp = ev.compute()
    """

    def __init__(self, classes_name):
        self.classes_name = list(classes_name)
        self.rows, self.row_counts, self.tg, self.tg_counts = [], [], [], []
        self.gt_box = self.pred_box = 0

    def add(self, detections, targets):
        dets = [d if d is not None else None for d in detections]
        dev = next((d.device for d in dets if d is not None), torch.device("cuda:0"))
        for d, t in zip(dets, targets):
            t = torch.as_tensor(t, dtype=torch.float32).reshape(-1, 5)
            k = 0 if d is None else int(d.shape[0])
            if k:
                self.rows.append(d.reshape(-1, 7).to(dev, torch.float32))
            self.row_counts.append(k)
            self.tg.append(t)
            self.tg_counts.append(int(t.shape[0]))
            self.gt_box += int(t.shape[0])
            self.pred_box += k

    def add_packed(self, rows, counts, targets):
        """rows [sum(counts),7] on the device (what the NMS stage wrote), counts: python ints per image."""
        if rows.shape[0]:
            self.rows.append(rows.reshape(-1, 7).to(torch.float32))
        self.row_counts += [int(c) for c in counts]
        for t in targets:
            t = torch.as_tensor(t, dtype=torch.float32).reshape(-1, 5)
            self.tg.append(t)
            self.tg_counts.append(int(t.shape[0]))
            self.gt_box += int(t.shape[0])
        self.pred_box += int(rows.shape[0])

    def compute_device(self, device=None):
        dev = torch.device(device) if device is not None else (self.rows[0].device if self.rows else torch.device("cuda:0"))
        rows = torch.cat(self.rows) if self.rows else torch.zeros(0, 7, device=dev)
        tg = (torch.cat(self.tg) if self.tg else torch.zeros(0, 5)).to(dev)
        D, T = int(rows.shape[0]), int(tg.shape[0])
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        db, dl, ds, tb, tl, td = f(max(D, 1), 4), f(max(D, 1)), f(max(D, 1)), f(max(T, 1), 4), f(max(T, 1)), f(max(T, 1))
        call("mny_eval_pack", _p(rows), D, _p(tg), T, _p(db), _p(dl), _p(ds), _p(tb), _p(tl), _p(td), _st())
        off = lambda c: torch.tensor([0] + c, dtype=torch.int64).cumsum(0).to(torch.int32).to(dev)
        return map_eval(db[:D], dl[:D], ds[:D], off(self.row_counts), tb[:T], tl[:T], td[:T], off(self.tg_counts), len(self.classes_name))

    def compute(self, device=None):
        r = self.compute_device(device)
        names = self.classes_name[1:]
        return dict(zip(names, r["ap"].tolist())), float(r["mean_ap"]), dict(zip(names, r["tp"].tolist())), dict(zip(names, r["fp"].tolist()))
