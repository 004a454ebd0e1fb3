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
        self.plan = plan
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

    def run_backward(self):
        self.ar.wait()                          # previous step's reduction must finish before the arena is rewritten
        pos = 0
        for k, (_b, _e, call_end) in enumerate(self.buckets):
            self.plan.run_bwd_segment(pos, call_end)
            pos = call_end
            self.ar.launch(k)
        self.plan.run_bwd_segment(pos, None)

    def wait(self):
        self.ar.wait()


class _ModelReducer:
    def __init__(self, model, group, n_buckets):
        self.model, self.group, self.n_buckets = model, group, n_buckets
        self.by_plan = {}

    def for_plan(self, plan):
        r = self.by_plan.get(id(plan))
        if r is None or r.plan is not plan:
            r = PlanReducer(plan, self.group, self.n_buckets)
            self.by_plan[id(plan)] = r
        return r

    def wait(self):
        for r in self.by_plan.values():
            r.wait()


def attach_data_parallel(model, group=None, n_buckets=4):
    """Broadcast rank 0's parameters/buffers and make `loss.backward()` average gradients across ranks.
    Returns an object whose `.wait()` must be called before the optimizer reads `p.grad`."""
    with torch.no_grad():
        for t in model.state_dict().values():
            dist.broadcast(t, src=0, group=group)
    red = _ModelReducer(model, group, n_buckets)
    model.dp_reducer = red
    return red
