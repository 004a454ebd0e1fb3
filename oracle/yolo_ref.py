"""Oracle (test infrastructure, NOT product): CPU restatement of the detection math.

Follows /root/reference models/yolo_loss.py and utils/iou.py; every function
cites the lines it restates.  torch-CPU fp32, autograd supplies dL/dhead for
the training path.  Pinned against tests/golden/loss_*.npz, decode_*.npz and
iou_tables.npz (captured from the real reference by tools/gen_golden.py).

Head layout convention used throughout the repo
-----------------------------------------------
The reference heads are NCHW ``[N, A*(5+C), g, g]`` with channel = a*(5+C)+attr
(yolo_loss.py:84).  The oracle accepts that layout (``layout="nchw"``) and the
product's channels-last layout ``[N, g, g, A*(5+C)]`` (``layout="nhwc"``).
"""
import math

import numpy as np
import torch


class YoloHeadSpec:
    """Static per-head hyper-parameters (yolo_loss.py:33-50)."""

    def __init__(self, anchors, mask, num_classes, ignore_thresh, iou_thresh,
                 iou_weighting, val_conf=0.1):
        self.anchors = [tuple(a) for a in anchors]      # all anchors, pixels (w,h)
        self.mask = list(mask)                          # indices of this head's anchors
        self.num_classes = int(num_classes)
        self.ignore_thresh = float(ignore_thresh)
        self.iou_thresh = float(iou_thresh)
        self.iou_weighting = float(iou_weighting)
        self.val_conf = float(val_conf)

    @property
    def attrs(self):
        return 5 + self.num_classes


def specs_from_config(cfg):
    """The two YOLOLoss objects of mbv2_yolo.py:132-135."""
    y = cfg["yolo"]
    return [YoloHeadSpec(y["anchors"], y["mask"][i], y["num_classes"],
                         y["ignore_thresh"][i], y["iou_thresh"],
                         cfg["iou_weighting"]) for i in range(2)]


# ----------------------------------------------------------------------------
# utils/iou.py
# ----------------------------------------------------------------------------
def pair_iou(a, b):
    """IoU of every box of a[n1,4] with every box of b[n2,4] (x1y1x2y2).

    utils/iou.py:32-49 (find_jaccard_overlap) incl. find_intersection :4-13.
    No epsilon: a 0/0 pair yields NaN exactly like the reference.
    """
    lo = torch.maximum(a[:, None, :2], b[None, :, :2])
    hi = torch.minimum(a[:, None, 2:], b[None, :, 2:])
    d = (hi - lo).clamp(min=0)
    inter = d[..., 0] * d[..., 1]
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def cxcywh_to_corners_(t):
    """In-place (cx,cy,w,h) -> (x1,y1,x2,y2) with the reference's exact order of
    operations: x1 = cx - w/2, then x2 = w + x1 (yolo_loss.py:243-247, Q6)."""
    t[..., 0] = t[..., 0] - t[..., 2] / 2
    t[..., 1] = t[..., 1] - t[..., 3] / 2
    t[..., 2] = t[..., 2] + t[..., 0]
    t[..., 3] = t[..., 3] + t[..., 1]
    return t


def ciou_pair(gt, pred):
    """(iou - ciou_term, iou) for one gt box vs one predicted box, each [1,4].

    yolo_loss.py:257-293.  Quirks kept: the distance term is divided by the
    *area* of the enclosing box (:264,:277); alpha is not detached (:283);
    falls back to plain iou when the enclosing area is 0 (:286-287).
    """
    ex1 = torch.minimum(gt[:, 0], pred[:, 0])
    ey1 = torch.minimum(gt[:, 1], pred[:, 1])
    ex2 = torch.maximum(gt[:, 2], pred[:, 2])
    ey2 = torch.maximum(gt[:, 3], pred[:, 3])
    c = ((ex2 - ex1) * (ey2 - ey1))[:, None]                       # :249-256, :264
    iou = pair_iou(gt, pred)                                        # :265
    w1, h1 = (gt[:, 2] - gt[:, 0])[:, None], (gt[:, 3] - gt[:, 1])[:, None]
    w2, h2 = (pred[:, 2] - pred[:, 0])[:, None], (pred[:, 3] - pred[:, 1])[:, None]
    cx1, cy1 = (gt[:, 2] + gt[:, 0])[:, None] / 2, (gt[:, 1] + gt[:, 3])[:, None] / 2
    cx2, cy2 = (pred[:, 2] + pred[:, 0])[:, None] / 2, (pred[:, 1] + pred[:, 3])[:, None] / 2
    u = (cx1 - cx2) * (cx1 - cx2) + (cy1 - cy2) * (cy1 - cy2)       # :272
    d = u / c                                                       # :277
    at = torch.atan(w2 / h2) - torch.atan(w1 / h1)                  # :279-282
    v = 4 / (math.pi * math.pi) * at * at
    alpha = v / (1 - iou + v + 0.000001)                            # :283
    term = d + alpha * v                                            # :284
    degenerate = (c == 0)
    term = term * (~degenerate) + iou * degenerate                  # :286-287
    return iou - term, iou


class _DeltaSigmoid(torch.autograd.Function):
    """yolo_loss.py:15-32: forward 1/(1+exp(-x)), backward = identity (Q2)."""

    @staticmethod
    def forward(ctx, x):
        return 1.0 / (1.0 + torch.exp(-x))

    @staticmethod
    def backward(ctx, g):
        return g.clone()


def _as_pred(head, spec, layout):
    """-> contiguous [N, A, g, g, 5+C] (yolo_loss.py:84 / :186)."""
    A, T = len(spec.mask), spec.attrs
    if layout == "nchw":
        n, _, gh, gw = head.shape
        return head.view(n, A, T, gh, gw).permute(0, 1, 3, 4, 2).contiguous()
    n, gh, gw, _ = head.shape
    return head.view(n, gh, gw, A, T).permute(0, 3, 1, 2, 4).contiguous()


def _grid_and_anchor_maps(spec, img_size, n, g):
    """yolo_loss.py:62-75 and :214 — anchors scaled by img_size (Q8), float32."""
    scaled = np.array([(aw / img_size[0], ah / img_size[1]) for aw, ah in spec.anchors])
    mine = torch.tensor(scaled[spec.mask], dtype=torch.float32)     # [A,2]
    A = len(spec.mask)
    ar = torch.linspace(0, g - 1, g)
    gx = ar.view(1, 1, 1, g).expand(n, A, g, g)
    gy = ar.view(1, 1, g, 1).expand(n, A, g, g)
    grid = torch.stack((gx, gy), 4)
    anch = mine.view(1, A, 1, 1, 2).expand(n, A, g, g, 2)
    return grid, anch, scaled


def decode_boxes(pred, xy, wh, spec, img_size):
    """Corner boxes [N,A,g,g,4] from activated xy / wh (yolo_loss.py:89-92,:192-196)."""
    n, A, g = pred.shape[0], pred.shape[1], pred.shape[2]
    grid, anch, scaled = _grid_and_anchor_maps(spec, img_size, n, g)
    boxes = torch.cat(((xy + grid) / torch.tensor([g, g], dtype=torch.float32),
                       wh * anch), 4)
    return cxcywh_to_corners_(boxes), scaled


def decode_rows(head, spec, img_size, layout="nchw"):
    """Eval path: per-image list of rows [x1,y1,x2,y2,conf,cls_score,cls_idx]
    kept where conf > val_conf, in (anchor,row,col) order (yolo_loss.py:180-204).
    Uses torch.sigmoid, not the delta sigmoid (Q7)."""
    pred = _as_pred(head, spec, layout)
    xy = torch.sigmoid(pred[..., 0:2])
    wh = torch.exp(pred[..., 2:4])
    cc = torch.sigmoid(pred[..., 4:])
    boxes, _ = decode_boxes(pred, xy, wh, spec, img_size)
    score, idx = torch.max(cc[..., 1:], dim=4)
    rows = torch.cat((boxes, cc[..., 0:1], score[..., None], idx.float()[..., None]), 4)
    keep = rows[..., 4] > spec.val_conf
    return [rows[b, keep[b]] for b in range(rows.shape[0])]


def loss_forward(head, targets, spec, img_size, layout="nchw"):
    """Training path of YOLOLoss.forward (yolo_loss.py:206-236, get_target :77-178).

    head     : tensor (may require grad) in `layout`
    targets  : list (len N) of float32 [n_i,5] = (label 1..C, cx, cy, w, h)
    returns  : (loss, recall, avg_iou, obj, no_obj, cls_score, count_per_image)
               with the same python/tensor types as the reference.
    """
    pred = _as_pred(head, spec, layout)
    n, A, g = pred.shape[0], pred.shape[1], pred.shape[2]
    C = spec.num_classes
    xy = _DeltaSigmoid.apply(pred[..., 0:2])                         # :85
    wh = torch.exp(pred[..., 2:4])                                   # :86
    out = _DeltaSigmoid.apply(pred[..., 4:])                         # :87
    boxes, scaled = decode_boxes(pred, xy, wh, spec, img_size)      # :89-92

    tgt = out.clone()                                                # :97
    wts = torch.zeros(n, A, g, g, C + 1)                             # :82
    no_obj = torch.sum(out[..., 0])                                  # :98
    cells = out[..., 0].numel()                                      # :99
    anchor_boxes = torch.tensor(
        np.concatenate((np.zeros((len(spec.anchors), 2)), scaled), 1), dtype=torch.float32)  # :102
    dims = torch.tensor([g, g], dtype=torch.float32)                 # :105
    count = 0
    recall = 0
    iou_sum = 0.0
    obj = 0.0
    cls_score = 0.0
    box_terms = []                                                   # iou_loss rows (:159)
    box_wts = []                                                     # iou_weight rows (:162)
    y_true = (1 - 0.1) + 0.5 * 0.1                                   # :426
    y_false = 0.5 * 0.1                                              # :427

    for b in range(n):
        t_b = targets[b]
        if len(t_b) == 0:                                            # :108-111
            wts[b, ..., 0] = 1
            tgt[b, ..., 0] = 0
            continue
        gt_corners = cxcywh_to_corners_(t_b[:, 1:].clone().detach())  # :112-113
        every = boxes[b].reshape(A * g * g, 4)
        best = pair_iou(gt_corners, every).max(0)[0].view(A, g, g)   # :116-120
        below = best < spec.ignore_thresh                            # :123
        wts[b, ..., 0][below] = 1                                    # :124
        tgt[b, ..., 0][below] = 0                                    # :125

        centre = t_b[:, 1:3] * dims                                  # :128
        shape_boxes = torch.cat((torch.zeros(len(t_b), 2), t_b[:, 3:5]), 1)   # :129-130
        a_iou = pair_iou(shape_boxes, anchor_boxes)                  # :132
        best_anchor = torch.argmax(a_iou, 1)                         # :133
        for t in range(len(t_b)):
            gi = int(centre[t, 0])                                   # :136
            gj = int(centre[t, 1])                                   # :137
            mine = a_iou[t][spec.mask]                               # :138
            over = (mine > spec.iou_thresh).tolist()                 # :139
            bn = int(best_anchor[t])
            k_best = spec.mask.index(bn) if bn in spec.mask else -1  # :140-142
            cls = int(t_b[t, 0] - 1)                                 # :131,:147 (labels 1-based, Q9)
            for k in range(A):
                if not (k == k_best or over[k]):                     # :145
                    continue
                count += 1
                tgt[b, k, gj, gi, 0] = 1                             # :149
                wts[b, k, gj, gi, 0] = 1                             # :150
                conf = out[b, k, gj, gi, 0].item()                   # :151
                obj += conf
                no_obj = no_obj - conf                               # :153
                g_box = gt_corners[t][None]
                p_box = boxes[b, k, gj, gi][None]
                term, iou = ciou_pair(g_box, p_box)                  # :157
                box_terms.append(term)
                box_wts.append(2.0 - (g_box[:, 2] - g_box[:, 0]) * (g_box[:, 3] - g_box[:, 1]))  # :160-162
                if iou > spec.ignore_thresh:                         # :163-164
                    recall += 1
                iou_sum += iou.item()                                # :165
                # class_loss :425-434
                if wts[b, k, gj, gi, 1 + cls] > 0:
                    tgt[b, k, gj, gi, 1 + cls] = y_true
                    wts[b, k, gj, gi, 1 + cls] = 1
                else:
                    tgt[b, k, gj, gi, 1:] = y_false
                    wts[b, k, gj, gi, 1:] = 1
                    tgt[b, k, gj, gi, 1 + cls] = y_true
                cls_score += out[b, k, gj, gi, 1 + cls].item()       # :169

    if count > 0:                                                    # :170-177
        obj_avg = obj / count
        cls_avg = cls_score / count
        no_obj = no_obj / (cells - count)
        avg_iou = iou_sum / count
        recall = recall / count
    else:
        recall = obj_avg = cls_avg = no_obj = avg_iou = 0

    # weighted_mse_loss :53-60 on conf+class
    loss = torch.sum((out - tgt) ** 2 * wts / torch.sum(wts))        # :219
    box_loss = torch.zeros(1)
    if box_terms:                                                    # :223-224
        x = torch.cat(box_terms)                                     # [P,1]
        w = torch.cat(box_wts)                                       # [P]
        # [P,1]*[P] broadcasts to [P,P]; the area weights cancel up to rounding (Q1)
        box_loss = torch.sum((x - 1) ** 2 * w / torch.sum(w)) / x.numel()
    loss = loss + box_loss * spec.iou_weighting                      # :234
    return loss, recall, avg_iou, obj_avg, no_obj, cls_avg, count / n
