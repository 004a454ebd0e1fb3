"""TEST INFRASTRUCTURE — CPU restatement of the reference's batch input preparation (folder2lmdb.py:223-265 collate_fn):
`transforms.Resize(size, BILINEAR)` on a PIL image -> `ToTensor()` -> `Normalize(mean, std)` -> stack, with one
`random.choice(train_img_size)` per batch.  Only tests/, __graft_entry__.smoke() and bench.py's CPU-baseline leg may
import this.

The resize itself lives in a third-party dependency that is not vendored in /root/reference: torchvision's Resize on a
PIL image is `img.resize((w, h), Image.BILINEAR)`, i.e. Pillow's ImagingResample (requirements: `Pillow`, unpinned;
this container has Pillow 12.2.0).  Restated below from its published algorithm (src/libImaging/Resample.c: antialiased
triangle filter whose support grows with the down-scale factor, 22-bit fixed-point coefficients, a horizontal then a
vertical pass, each rounded to uint8) and PINNED bit-exactly against Pillow itself run in this container
(tools/gen_golden_prep.py -> tests/golden/prep_*.npz; tests/test_oracle_prep.py also checks it live when PIL imports).
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def coefficients(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter over the whole
    axis.  -> bounds int32 [out,2] (first tap, tap count), kk int32 [out, ksize]"""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(xmax, np.float64)
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64)        # all coefficients are >= 0 for this filter
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One ImagingResample pass over uint8 HWC: int32 accumulate from 1 << (PRECISION_BITS-1), arithmetic shift, clip to 0..255."""
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((len(bounds),) + img.shape[1:], np.int64)
    for i, (lo, n) in enumerate(bounds):
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for t in range(n):
            acc += img[lo + t] * int(kk[i, t])
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out.astype(np.uint8), 0, axis)


def resize_bilinear_u8(img_hwc, out_h, out_w):
    """Image.resize((out_w, out_h), BILINEAR) of an RGB uint8 image: horizontal pass first, a pass whose size does not
    change is skipped (Resample.c ImagingResampleInner)."""
    h, w = img_hwc.shape[:2]
    out = img_hwc
    if w != out_w:
        out = _pass(out, *coefficients(w, out_w), axis=1)
    if h != out_h:
        out = _pass(out, *coefficients(h, out_h), axis=0)
    return np.ascontiguousarray(out)


def to_tensor_normalize(img_hwc_u8, mean, std):
    """transforms.ToTensor (uint8 HWC -> float CHW / 255) then transforms.Normalize: (x - mean) / std, all fp32."""
    f = np.float32
    x = (img_hwc_u8.astype(f) / f(255)).transpose(2, 0, 1)
    return ((x - np.asarray(mean, f)[:, None, None]) / np.asarray(std, f)[:, None, None]).astype(f)


def collate(images, size, mean, std):
    """folder2lmdb.py:223-256 for the image half of a batch: every image -> (size[0], size[1]) -> tensor -> normalise -> stack."""
    return np.stack([to_tensor_normalize(resize_bilinear_u8(im, size[0], size[1]), mean, std) for im in images])


def choose_size(rng, train_img_size):
    """folder2lmdb.py:227: random.choice(self.transform_size) — one size per batch."""
    return rng.choice(train_img_size)
