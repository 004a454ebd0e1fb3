"""Data-parallel training: one process per GPU, replicated parameters, the minibatch sharded by image,
gradients averaged with RCCL all-reduce over xGMI (torch.distributed backend "nccl" == RCCL on ROCm).

The reference has no distributed code (train.py:14-18 import torch.distributed and never use it); the
semantics implemented here are "what DistributedDataParallel without SyncBN would compute" (SURVEY §8e):
BatchNorm statistics and the loss normaliser stay per rank, gradients are averaged.

Design for xGMI (point-to-point, 7 links x ~153 GB/s): the whole fp32 gradient set is 19.7 MB, so an
all-reduce is latency- not bandwidth-bound (~0.2 ms).  The engine lays the gradients out in ONE flat arena
in backward-production order (heads first).  `plan_buckets` cuts it into a few contiguous buckets at
backward call boundaries; bucket k's all-reduce is enqueued the moment the backward call that completes
it has been launched, so it rides under the remaining backward kernels, and the last one under the next
step's forward (the arena is only waited for before it is overwritten or read by the optimizer).
"""
import weakref

import torch
import torch.distributed as dist


def plan_buckets(slot_sizes, ready_call, n_buckets=4):
    """slot_sizes: floats of every gradient slot in arena order; ready_call[i]: index of the backward call
    after which slot i is final (non-decreasing).  Returns [(float_begin, float_end, call_index)], contiguous,
    balanced by bytes, cut only where the ready index changes."""
    assert len(slot_sizes) == len(ready_call) and slot_sizes
    total = sum(slot_sizes)
    target = total / max(1, n_buckets)
    out, begin, acc, pos = [], 0, 0, 0
    for i, n in enumerate(slot_sizes):
        acc += n
        pos += n
        last = i == len(slot_sizes) - 1
        boundary = last or ready_call[i + 1] != ready_call[i]
        if boundary and (acc >= target or last) :
            out.append((begin, pos, ready_call[i]))
            begin, acc = pos, 0
    return out


class BucketedAllReduce:
    """Average contiguous slices of one flat tensor across the process group, asynchronously."""

    def __init__(self, flat, buckets, group=None):
        self.flat, self.buckets, self.group = flat, buckets, group
        self.world = dist.get_world_size(group)
        self.pending = []
        # the reduction op follows the BACKEND, not the tensor's device: RCCL ("nccl") averages in the collective; gloo has no
        # AVG (SUM, then scale) and — the way two ranks share the one GPU of a test box — is fed through a host staging copy
        self.backend = str(dist.get_backend(group)).lower()
        self.avg = getattr(dist.ReduceOp, "AVG", None) if self.backend == "nccl" else None
        self.stage_host = self.avg is None and flat.is_cuda
        if self.stage_host:
            self.host = torch.empty(max(e - b for b, e, _ in buckets), dtype=flat.dtype).pin_memory()

    def launch(self, k):
        b, e, _ = self.buckets[k]
        view = self.flat[b:e]
        if self.avg is not None:
            w = dist.all_reduce(view, op=self.avg, group=self.group, async_op=True)
            self.pending.append((w, None))
        elif self.stage_host:       # gloo over device memory (tests): stream-ordered copy out, blocking reduce, copy back
            h = self.host[:e - b]
            h.copy_(view)           # D2H on the current stream: ordered behind the backward calls that produced the bucket
            torch.cuda.current_stream(self.flat.device).synchronize()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            h.mul_(1.0 / self.world)
            view.copy_(h)
            torch.cuda.current_stream(self.flat.device).synchronize()      # `host` is reused by the next bucket
        else:                       # gloo on CPU tensors: SUM then scale
            w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.pending.append((w, view))

    def wait(self):
        for w, view in self.pending:
            w.wait()                # NCCL: makes the current stream wait; gloo: blocks the host
            if view is not None:
                view.mul_(1.0 / self.world)
        self.pending = []


class PlanReducer:
    """Glue between a training NetPlan and BucketedAllReduce: replays the backward call list in
    segments and enqueues each bucket's all-reduce right after the segment that completes it."""

    def __init__(self, plan, group=None, n_buckets=4):
        sizes = self.slot_extents(plan)
        ready = self._ready_calls(plan)
        self._plan = weakref.ref(plan)            # the plan owns its reducer (plan._dp), never the other way round
        self.owner = lambda: None
        self.buckets = plan_buckets(sizes, ready, n_buckets)
        if self.buckets[0][0] != 0 or self.buckets[-1][1] != plan.gflat.numel():
            raise RuntimeError("gradient buckets [%d, %d) do not cover the arena of %d floats" % (
                self.buckets[0][0], self.buckets[-1][1], plan.gflat.numel()))
        self.ar = BucketedAllReduce(plan.gflat, self.buckets, group)

    @staticmethod
    def slot_extents(plan):
        """Floats each gradient slot occupies in the arena, in arena order, taken from the REAL layout: next slot's offset
        minus this slot's, the arena's end closing the last one.  (The engine pads slots to 16 bytes and reserves slack behind
        the padded detection-head gradients — engine._build_backward — so sizes recomputed from numel() fall short.)"""
        offs = [plan.grad_slots[n][0] for n in plan.grad_params]
        assert offs == sorted(offs) and (not offs or offs[0] == 0), "arena slots must be laid out in grad_params order"
        ends = offs[1:] + [plan.gflat.numel()]
        return [e - b for b, e in zip(offs, ends)]

    @staticmethod
    def _ready_calls(plan):
        """For every gradient slot: index (exclusive end) of the backward call that writes it last."""
        ptr_to_slot = {plan.gviews[n].data_ptr(): i for i, n in enumerate(plan.grad_params)}
        ready = [0] * len(plan.grad_params)
        for ci, (_fn, args, _name, meta) in enumerate(plan.bwd.calls):
            written = [getattr(a, "value", None) for a in args] + list((meta or {}).get("writes", ()))   # `writes`: destinations named inside a device job table (mny_reduce_batch)
            for v in written:
                if v in ptr_to_slot:
                    ready[ptr_to_slot[v]] = ci + 1
        # arena order == production order, but make the sequence monotone for safety
        for i in range(1, len(ready)):
            ready[i] = max(ready[i], ready[i - 1])
        return ready

    @property
    def plan(self):
        return self._plan()

    def run_backward(self):
        self.ar.wait()                          # previous step's reduction must finish before the arena is rewritten
        plan = self._plan()
        pos = 0
        for k, (_b, _e, call_end) in enumerate(self.buckets):
            plan.run_bwd_segment(pos, call_end)
            pos = call_end
            self.ar.launch(k)
        plan.run_bwd_segment(pos, None)

    def wait(self):
        self.ar.wait()


class _ModelReducer:
    """Per-model handle returned by attach_data_parallel.  The PlanReducer of a plan is OWNED BY THE PLAN (`plan._dp`); this object
    only keeps weak references, so a plan evicted from the model's plan cache (multi-scale training: one ~40 GB plan per size at
    bs 256) frees its memory — round 2 kept every plan it had ever seen alive through a strong by_plan table."""

    def __init__(self, model, group, n_buckets):
        self.model, self.group, self.n_buckets = weakref.ref(model), group, n_buckets
        self.live = weakref.WeakSet()              # PlanReducers of the plans that are still alive
        self._hook = None

    def for_plan(self, plan):
        r = getattr(plan, "_dp", None)
        if r is None or r.owner() is not self:
            r = PlanReducer(plan, self.group, self.n_buckets)
            r.owner = weakref.ref(self)
            plan._dp = r
            self.live.add(r)
        return r

    def wait(self):
        for r in list(self.live):
            r.wait()

    def detach(self):
        if self._hook is not None:
            self._hook.remove()
            self._hook = None


def attach_data_parallel(model, group=None, n_buckets=4):
    """Broadcast rank 0's parameters/buffers and make `loss.backward()` average gradients across ranks.

    The last gradient bucket's all-reduce is left in flight when backward() returns (it overlaps whatever comes next).  So that
    the literal train loop of the reference (`loss.backward(); optimizer.step()`, train.py:282-283) is safe with ANY
    torch.optim.Optimizer — the fused `optim.AdamW` of this package included — a global optimizer-step pre-hook makes the current
    stream wait for the pending all-reduces before the optimizer reads `p.grad`.  Code that reads `p.grad` by other means right after
    backward (gradient clipping, logging) calls `.wait()` on the returned object first."""
    with torch.no_grad():
        for t in model.state_dict().values():
            dist.broadcast(t, src=0, group=group)
    red = _ModelReducer(model, group, n_buckets)
    model.dp_reducer = red
    ref = weakref.ref(red)

    def _before_optimizer_step(_opt, _args, _kwargs):
        r = ref()
        if r is not None:
            r.wait()
    try:
        from torch.optim.optimizer import register_optimizer_step_pre_hook
        red._hook = register_optimizer_step_pre_hook(_before_optimizer_step)
    except ImportError:                               # very old torch: the caller waits by hand
        red._hook = None
    return red
