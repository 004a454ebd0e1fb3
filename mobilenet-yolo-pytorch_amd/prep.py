"""Device-side batch input preparation — host mirror of the image half of the reference's
`collate_fn` (folder2lmdb.py:223-256): one `random.choice(train_img_size)` per batch, every decoded image resized with
Pillow's BILINEAR semantics, ToTensor, Normalize, stacked — here as ONE packed upload + `mny_prep_batch` (csrc/prep.hip),
returning the NCHW fp32 batch on the GPU.  JPEG decoding, augmentation and the seg-map resize (cv2 INTER_AREA; cv2 is not
available to pin it) stay with the caller.  No CPU fallback: without libmnyolo.so every call raises MnyError."""
import ctypes
import random

import numpy as np
import torch

from ._lib import call, query

DESC = np.dtype([("offset", np.int64), ("h", np.int32), ("w", np.int32)])       # mny_image_desc


class BatchPrep:
    """prep = BatchPrep(config["train_img_size"], config["normalize"]["mean"], config["normalize"]["std"])
       images = prep(list_of_uint8_hwc_arrays)            # -> [N,3,H,W] float32 on `device`, size drawn per batch
    """

    def __init__(self, train_img_size, mean, std, device="cuda:0", rng=None):
        self.sizes = [tuple(int(v) for v in s) for s in train_img_size]
        self.mean = (ctypes.c_float * 3)(*[float(v) for v in mean])
        self.std = (ctypes.c_float * 3)(*[float(v) for v in std])
        self.device = torch.device(device)
        self.rng = rng or random                    # the reference draws from the global `random` (folder2lmdb.py:227)
        self._stage = None                          # pinned staging buffer, grown on demand
        self._status = None

    def choose_size(self):
        return self.rng.choice(self.sizes)          # folder2lmdb.py:227

    def pack(self, images):
        """uint8 HWC RGB arrays/tensors -> (pinned uint8 buffer view, desc array, max_h, max_w)."""
        arrs = []
        for im in images:
            a = im.numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise ValueError("expected decoded RGB uint8 images [h,w,3], got %s %s" % (a.dtype, a.shape))
            arrs.append(a)
        desc = np.zeros(len(arrs), DESC)
        off = 0
        for i, a in enumerate(arrs):
            desc[i] = (off, a.shape[0], a.shape[1])
            off += (a.size + 15) // 16 * 16
        if self._stage is None or self._stage.numel() < off:
            self._stage = torch.empty(max(off, 1 << 20), dtype=torch.uint8)
            if torch.cuda.is_available():
                self._stage = self._stage.pin_memory()
        buf = self._stage.numpy()
        for d, a in zip(desc, arrs):
            buf[d["offset"]:d["offset"] + a.size] = a.reshape(-1)
        return self._stage[:off], desc, int(desc["h"].max()), int(desc["w"].max())

    def run_device(self, src, desc_dev, n, max_h, max_w, size, out=None):
        """Everything already on the device: src uint8 buffer, desc_dev = the mny_image_desc table as a uint8/int64 tensor."""
        oh, ow = int(size[0]), int(size[1])
        if out is None:
            out = torch.empty(n, 3, oh, ow, device=self.device, dtype=torch.float32)
        ws = torch.empty(query("mny_prep_ws_bytes", n, max_h, max_w, oh, ow), device=self.device, dtype=torch.uint8)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        call("mny_prep_batch", p(src), p(desc_dev), n, max_h, max_w, oh, ow, self.mean, self.std, p(out), p(ws),
             ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        self._status = ws[:4].view(torch.int32)
        self._keep = (src, desc_dev, ws)
        return out

    def __call__(self, images, size=None):
        if len(images) == 0:
            raise ValueError("empty batch")
        size = size or self.choose_size()
        stage, desc, mh, mw = self.pack(images)
        src = stage.to(self.device, non_blocking=True)
        desc_dev = torch.from_numpy(desc.view(np.uint8).copy()).to(self.device, non_blocking=True)
        return self.run_device(src, desc_dev, len(images), mh, mw, size)

    def check(self):
        """Host sync: raise if the last batch held an image outside the declared bounds."""
        if self._status is not None and int(self._status.item()) != 0:
            raise RuntimeError("input prep: image %d is empty or larger than the declared maximum" % (int(self._status.item()) - 1))
