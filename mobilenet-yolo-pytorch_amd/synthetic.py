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


def map_case(n_images, n_classes, seed=0, mean_gt=2.5, crafted=False, no_detections=False, det_per_gt=1.6, clutter=6.0):
    """Evaluation-set shaped input of the mAP stage (packed, float32): per image ~Poisson(mean_gt) ground-truth boxes
    (10 % difficult), detections = jittered copies of ground truth (some twice: the duplicate-hit case, some with the
    wrong label) + low-score clutter, stored per image in per-class, score-descending order like the NMS stage emits
    them (every 5th image shuffled: matching follows stored order).  n_classes counts the background entry.
    -> det_boxes [D,4], det_labels [D], det_scores [D], det_off [n+1], true_boxes [T,4], true_labels [T], true_diff [T], true_off [n+1]"""
    r = np.random.RandomState(seed)
    f = np.float32
    db, dl, ds, do, tb, tl, td, to = [], [], [], [0], [], [], [], [0]
    for i in range(n_images):
        k = 0 if (crafted and i == 2) else r.poisson(mean_gt)
        cxy, wh = 0.1 + 0.8 * r.rand(k, 2), 0.05 + 0.4 * r.rand(k, 2)
        gtb = np.concatenate((cxy - wh / 2, cxy + wh / 2), 1).astype(f)
        gtl = r.randint(1, n_classes, size=k).astype(f)
        gtd = (r.rand(k) < 0.1).astype(f)
        if crafted and i == 0 and k >= 2:                       # two objects of one class on top of each other (first-max tie-break)
            gtb[1], gtl[1] = gtb[0], gtl[0]
        boxes, labels, scores = [], [], []
        for j in range(k):
            for _ in range(r.poisson(det_per_gt)):
                jit = r.randn(4) * 0.03 * (1 + 3 * (r.rand() < 0.25))
                boxes.append(gtb[j] + jit.astype(f))
                labels.append(gtl[j] if r.rand() > 0.1 else f(r.randint(1, n_classes)))
                scores.append(0.3 + 0.7 * r.rand())
        for _ in range(r.poisson(clutter)):
            c, s = 0.1 + 0.8 * r.rand(2), 0.05 + 0.4 * r.rand(2)
            boxes.append(np.concatenate((c - s / 2, c + s / 2)).astype(f))
            labels.append(f(r.randint(1, n_classes)))
            scores.append(0.02 + 0.4 * r.rand())
        if no_detections or (crafted and i == 4):
            boxes, labels, scores = [], [], []
        b, l, s = np.array(boxes, f).reshape(-1, 4), np.array(labels, f), np.array(scores, f)
        order = np.lexsort((-s, l)) if i % 5 != 3 else r.permutation(len(s))
        db.append(b[order]); dl.append(l[order]); ds.append(s[order]); do.append(do[-1] + len(s))
        tb.append(gtb); tl.append(gtl); td.append(gtd); to.append(to[-1] + k)
    cat = lambda xs, shape: np.concatenate(xs).astype(f) if xs else np.zeros(shape, f)
    return (cat(db, (0, 4)).reshape(-1, 4), cat(dl, (0,)), cat(ds, (0,)), np.array(do, np.int32),
            cat(tb, (0, 4)).reshape(-1, 4), cat(tl, (0,)), cat(td, (0,)), np.array(to, np.int32))


def photos(sizes, seed=0):
    """Decoded-JPEG stand-ins for the input-prep stage: uint8 RGB [h,w,3] per (h,w) in `sizes` — smooth gradients + blocks +
    noise, so resampling has real structure to average (pure noise would hide coefficient errors behind rounding)."""
    r = np.random.RandomState(seed)
    out = []
    for h, w in sizes:
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        img = np.zeros((h, w, 3), np.float32)
        for c in range(3):
            fx, fy, ph = r.uniform(0.01, 0.2), r.uniform(0.01, 0.2), r.uniform(0, 6.28)
            img[..., c] = 127 + 90 * np.sin(fx * xx + fy * yy + ph)
        for _ in range(4):
            y0, x0 = r.randint(0, h), r.randint(0, w)
            img[y0:y0 + r.randint(2, max(3, h // 3)), x0:x0 + r.randint(2, max(3, w // 3))] = r.randint(0, 256, 3)
        img += r.randn(h, w, 3) * 12
        out.append(np.clip(img, 0, 255).astype(np.uint8))
    return out


def dli14_mobilenetv2_keys(width=1.0):
    """(key, shape) list with the layout of the ImageNet MobileNetV2 checkpoint the reference downloads (d-li14
    `mobilenetv2-c5e733a8.pth`, mbv2_yolo.py:116): ONE `features` Sequential of 18 modules (stem + 17 inverted-residual
    blocks), the 1x1 `conv` to 1280 and a `classifier` — the input side of `yolo.load_pretrained_backbone`.
    Written from the architecture table (t, c, n, s); used by tests and tools to build local stand-in checkpoints."""
    def bn(prefix, c):
        return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)),
                (prefix + ".running_var", (c,)), (prefix + ".num_batches_tracked", ())]
    out = [("features.0.0.weight", (32, 3, 3, 3))] + bn("features.0.1", 32)
    cin, idx = 32, 1
    for t, c, n, _s in [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]:
        for _ in range(n):
            p = "features.%d.conv" % idx
            hid, j = cin * t, 0
            if t != 1:
                out += [(p + ".0.weight", (hid, cin, 1, 1))] + bn(p + ".1", hid)
                j = 3
            out += [(p + ".%d.weight" % j, (hid, 1, 3, 3))] + bn(p + ".%d" % (j + 1), hid)
            out += [(p + ".%d.weight" % (j + 3), (c, hid, 1, 1))] + bn(p + ".%d" % (j + 4), c)
            cin, idx = c, idx + 1
    out += [("conv.0.weight", (1280, cin, 1, 1))] + bn("conv.1", 1280)
    out += [("classifier.weight", (1000, 1280)), ("classifier.bias", (1000,))]
    return out
