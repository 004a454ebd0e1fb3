"""Fused AdamW over libmnyolo's multi-tensor kernel — a drop-in for `torch.optim.AdamW` at train.py:134,283.

    optimizer = AdamW(model.parameters(), lr=7e-4, weight_decay=4e-4)      # same arguments as torch.optim.AdamW
    loss.backward(); optimizer.step()

One HIP launch per step updates every parameter (the reference issues ~200 small tensor updates).  State layout matches
torch's (`state[p] = {"step", "exp_avg", "exp_avg_sq"}`; the two moments are views into flat buffers), so
`state_dict()` / `load_state_dict()` interoperate with `torch.optim.AdamW` checkpoints.  Parameters whose `.grad` is
None are skipped exactly like upstream (the seg branch, Q10).  fp32 CUDA(HIP) parameters only; no CPU fallback.
"""
import ctypes

import numpy as np
import torch

from . import _lib

CHUNK = 65536
_CHUNK_DT = np.dtype([("p", np.uint64), ("g", np.uint64), ("m", np.uint64), ("v", np.uint64), ("n", np.int32), ("vec4", np.int32)])


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        if amsgrad:
            raise _lib.MnyError("fused AdamW: amsgrad is not implemented (the reference does not use it)")
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._tables = {}           # group index -> (signature, [one chunk table per distinct step count: table, nchunks, t, params])

    def _ensure_state(self, group):
        new = [p for p in group["params"] if p.grad is not None and len(self.state[p]) == 0]
        if not new:
            return
        total = sum((p.numel() + 3) // 4 * 4 for p in new)
        dev = new[0].device
        flat_m = torch.zeros(total, device=dev, dtype=torch.float32)
        flat_v = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        for p in new:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise _lib.MnyError("fused AdamW needs contiguous fp32 CUDA(HIP) parameters — there is no CPU fallback")
            n = p.numel()
            st = self.state[p]
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = flat_m[off:off + n].view_as(p)
            st["exp_avg_sq"] = flat_v[off:off + n].view_as(p)
            off += (n + 3) // 4 * 4

    def _table(self, gi, group):
        """Device chunk tables of group `gi`, rebuilt only when a parameter / gradient pointer changes (a step costs one pass
        over the parameters reading two pointers each — the moments only move through load_state_dict, which drops the cache).
        One table (= one launch) per distinct step count: torch keeps the step per parameter, and parameters that first receive
        a gradient later (a branch enabled or unfrozen mid-run, a torch checkpoint with mixed counts) carry their own."""
        live = [p for p in group["params"] if p.grad is not None]
        sig = tuple((p.data_ptr(), p.grad.data_ptr()) for p in live)
        cached = self._tables.get(gi)
        if cached is not None and cached[0] == sig:
            return cached[1], live
        self._sync_steps(gi)                                      # counters of the parameters the old tables covered
        by_step = {}
        for p, (pp, gp) in zip(live, sig):
            g, st = p.grad, self.state[p]
            if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()):
                raise _lib.MnyError("fused AdamW needs contiguous fp32 CUDA(HIP) gradients")
            mp, vp = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            n = p.numel()
            sub = by_step.setdefault(int(float(st["step"])), {"rows": [], "ids": set(), "params": []})
            sub["ids"].add(id(p))
            sub["params"].append(p)
            for o in range(0, n, CHUNK):
                c = min(CHUNK, n - o)
                ptrs = (pp + 4 * o, gp + 4 * o, mp + 4 * o, vp + 4 * o)
                sub["rows"].append(ptrs + (c, int(all(q % 16 == 0 for q in ptrs))))
        subs = []
        for t in sorted(by_step):
            sub = by_step[t]
            host = np.array(sub["rows"], dtype=_CHUNK_DT)
            subs.append({"table": torch.from_numpy(host.view(np.uint8).reshape(-1)).to(live[0].device), "nchunks": len(sub["rows"]), "t": t,
                         "ids": sub["ids"], "params": sub["params"]})
        self._tables[gi] = (sig, subs)
        return subs, live

    def _sync_steps(self, only=None):
        """write the step counters back into the per-parameter `step` tensors (torch's state layout)"""
        for gi in list(self._tables):
            if only is not None and gi != only:
                continue
            for sub in self._tables[gi][1]:
                for p in sub["params"]:
                    self.state[p]["step"].fill_(float(sub["t"]))

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables.clear()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            self._ensure_state(group)
            if not any(p.grad is not None for p in group["params"]):
                continue
            subs, live = self._table(gi, group)
            b1, b2 = group["betas"]
            st = ctypes.c_void_p(torch.cuda.current_stream(live[0].device).cuda_stream)
            for sub in subs:                                      # normally one; one launch per distinct step count otherwise
                sub["t"] += 1
                _lib.call("mny_adamw_step", ctypes.c_void_p(sub["table"].data_ptr()), sub["nchunks"], float(group["lr"]), float(b1), float(b2),
                          float(group["eps"]), float(group["weight_decay"]), sub["t"], st)
        return loss
