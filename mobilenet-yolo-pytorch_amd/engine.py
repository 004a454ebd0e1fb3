"""Static execution plans for the MobileNet-YOLO graph: one flat list of C-ABI calls for the forward
pass, one for the backward pass, built once per (batch, height, width, mode) and replayed every step.

All device buffers (raw conv outputs, BN coefficients, gradients, workspaces) are allocated when the
plan is built and stay resident — at bs=256/352x352 that is ~60 GB of the 288 GB HBM — so a step does
no allocation, no host<->device synchronisation and no Python work besides walking the call list
(which is also what makes the whole step hipGraph-capturable).

Backward dataflow (per conv+BN+act unit, given G = dL/d(activated output)):
    bn_bwd_reduce(G, Y) -> bn_bwd_finalize -> dgamma, dbeta, coef ; bn_bwd_apply -> dY
    wgrad(view(input), dY) -> dW        dgrad(dY, W) (+ addend) -> G of the input
Residual adds alias G to both operands; a second contribution to a value is fused into the producing
kernel's `addend` epilogue, so no separate accumulation pass exists.
"""
import ctypes
import itertools
import os
import warnings

import torch

from . import _lib
from ._lib import ACT_NONE, MnyError, YoloHead

_FWD_TICK = itertools.count()            # process-wide order of training forwards (NetPlan.last_fwd_tick)

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
_vp = ctypes.c_void_p


def _ptr(t):
    return _vp(t.data_ptr()) if t is not None else None


class CallList:
    """A replayable sequence of C-ABI calls."""

    def __init__(self):
        self.calls = []
        self.keep = []           # tensors referenced by raw pointers
        self.marks = {}          # name -> call index (for DP bucket events)

    def add(self, name, *args, meta=None, label=None):
        """`label`: the entry point this call stands for in timing tables / launch fingerprints (the _w6 twins of the GEMM entry points)."""
        fn = getattr(_lib.load(), name)
        conv = []
        for a in args:
            if isinstance(a, torch.Tensor):
                assert a.is_contiguous() and a.is_cuda, name
                self.keep.append(a)
                conv.append(_vp(a.data_ptr()))
            else:
                conv.append(a)
        self.calls.append((fn, tuple(conv), label or name, meta))   # meta: algorithmic {flops, bytes} of the call

    def add_py(self, fn, name="py"):
        """A host-side step of the list (stream fork / join): any callable returning None."""
        self.calls.append((fn, (), name, None))

    def run(self, begin=0, end=None):
        lib = _lib.load()
        for fn, args, name, _ in self.calls[begin:end]:
            rc = fn(*args)
            if rc:
                raise MnyError("%s failed (%d): %s" % (name, rc, lib.mny_last_error().decode()))

    def run_recording(self, routes, begin=0, end=None):
        """run(), and for every pointwise-conv call note the kernel family its dispatcher took (mny_pw_last_route) in routes[call index]."""
        lib = _lib.load()
        end = len(self.calls) if end is None else end
        for idx in range(begin, end):
            fn, args, name, _ = self.calls[idx]
            rc = fn(*args)
            if rc:
                raise MnyError("%s failed (%d): %s" % (name, rc, lib.mny_last_error().decode()))
            if name.startswith(("mny_pw_fwd", "mny_pw_dgrad_bnred", "mny_pw_wgrad")):
                routes[idx] = lib.mny_pw_last_route()

    def run_timed(self, events, only=None, begin=0, end=None):
        """Like run(), bracketing each call (or only the entry points named in `only`) with HIP events
        recorded on torch's current stream — the stream the kernels are launched on.
        Appends (call index, name, start, end) to `events`."""
        lib = _lib.load()
        end = len(self.calls) if end is None else end
        stream = _vp(torch.cuda.current_stream().cuda_stream)
        for idx in range(begin, end):
            fn, args, name, _ = self.calls[idx]
            timed = only is None or name in only
            if timed:
                a, b = HipEvent.take(), HipEvent.take()
                a.record(stream)
            rc = fn(*args)
            if rc:
                raise MnyError("%s failed (%d): %s" % (name, rc, lib.mny_last_error().decode()))
            if timed:
                b.record(stream)
                events.append((idx, name, a, b))


class _TorchEvent:
    """Fallback bracket event (see HipEvent._new)."""
    __slots__ = ("e",)

    def __init__(self):
        self.e = torch.cuda.Event(enable_timing=True)

    def record(self, stream):
        self.e.record()

    def elapsed_time(self, other):
        return self.e.elapsed_time(other.e)


class HipEvent:
    """Timing events straight on the HIP runtime (hipEventCreate / hipEventRecord / hipEventElapsedTime through ctypes), drawn from
    a pool that is filled BEFORE the timed region: a `torch.cuda.Event` is created lazily inside its first `record()`, which put
    ~10 us of host work per bracket into the measured step (invisible at bs=256, 15 % of the step at bs=64)."""
    _hip = None
    _checked = False
    _pool = []
    _next = 0
    __slots__ = ("h",)

    @classmethod
    def _rt(cls):
        if cls._hip is None:
            path = "libamdhip64.so"
            try:                                   # the copy of the runtime TORCH runs on: a handle from another mapped copy crashes instead of failing
                tlib = os.path.join(os.path.dirname(os.path.abspath(torch.__file__)), "lib")
                cands = []
                with open("/proc/self/maps") as f:
                    for line in f:
                        if "libamdhip64.so" in line:
                            cands.append(line.split()[-1])
                mine = [c for c in cands if os.path.dirname(os.path.abspath(c)) == tlib]
                uniq = sorted(set(cands))
                if mine:
                    path = mine[0]                 # torch's bundled runtime
                elif len(uniq) == 1:
                    path = uniq[0]                 # one runtime in the process (system ROCm): unambiguous
                elif uniq:
                    raise OSError("several HIP runtimes mapped (%s), none of them torch's" % ", ".join(uniq))
            except OSError as e:
                warnings.warn("raw HIP events unavailable (%s); timing brackets use torch.cuda.Event" % (e,))
                cls._hip = False
                raise MnyError("no unambiguous HIP runtime")
            cls._hip = ctypes.CDLL(path)
            cls._hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
            cls._hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            cls._hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        return cls._hip

    def __init__(self):
        h = ctypes.c_void_p()
        if self._rt().hipEventCreate(ctypes.byref(h)) != 0:
            raise MnyError("hipEventCreate failed")
        self.h = h

    @classmethod
    def _new(cls):
        """A raw HIP event; if the runtime cannot be reached through ctypes (or is not the one torch runs on), a torch event."""
        if cls._hip is not False:
            try:
                ev = cls()
                if not cls._checked:               # one record / elapsed round trip on torch's stream before trusting the handle
                    ev2 = cls()
                    st = _vp(torch.cuda.current_stream().cuda_stream)
                    ev.record(st); ev2.record(st)
                    torch.cuda.current_stream().synchronize()
                    ev.elapsed_time(ev2)
                    cls._checked = True
                return ev
            except (OSError, AttributeError, MnyError, RuntimeError) as e:
                warnings.warn("raw HIP events unavailable (%s); timing brackets fall back to torch.cuda.Event" % (e,))
                cls._hip = False
        return _TorchEvent()

    @classmethod
    def reserve(cls, n):
        """Rewind the pool and make sure it holds n events (call outside the timed region)."""
        cls._next = 0
        while len(cls._pool) < n:
            cls._pool.append(cls._new())

    @classmethod
    def take(cls):
        if cls._next == len(cls._pool):
            cls._pool.append(cls._new())
        cls._next += 1
        return cls._pool[cls._next - 1]

    def record(self, stream):
        if self._hip.hipEventRecord(self.h, stream) != 0:
            raise MnyError("hipEventRecord failed")

    def elapsed_time(self, other):
        ms = ctypes.c_float()
        if self._hip.hipEventElapsedTime(ctypes.byref(ms), self.h, other.h) != 0:
            raise MnyError("hipEventElapsedTime failed (events not complete?)")
        return float(ms.value)


class _Unit:
    """Resident state of one conv+BN+act unit."""
    __slots__ = ("Y", "coef4", "scale", "shift", "mean", "invstd", "act", "C", "M", "shape")


class NetPlan:
    def __init__(self, net, N, H, W, training, act_dtype=torch.float32, bn_batch=None, frozen_bwd=False):
        """training: build the loss calls (else decode + NMS).  bn_batch (default = training): BatchNorm on batch statistics
        with the running-statistics update, and the backward list; False with training=True is the "validation loss" plan
        (running statistics, buffers untouched) — forward only, or with frozen_bwd the backward list of frozen BatchNorm
        (the statistics are constants of the step: mbv2_yolo.py:157 under model.eval() with gradients)."""
        self.net, self.N, self.H, self.W, self.training = net, N, H, W, training
        bn_batch = training if bn_batch is None else bool(bn_batch)      # (bn_batch without training: decode + NMS on batch statistics, model.train()(images))
        self.bn_batch = bn_batch
        self.frozen = bool(training and not bn_batch and frozen_bwd)
        self.fwd_gen = 0
        self.inflight_gen = None                # model._InFlight: forward generation whose backward is still owed
        self.last_fwd_tick = -1                 # process-wide order of training forwards (model._grad_plan picks the oldest when every slot is busy)
        self.resident_bytes = 0
        dev = net.device
        self.dev = dev
        # storage type of activations / activation gradients.  bf16 selects the `*_bf16` twins of the entry points
        # (same arithmetic in fp32 registers, half the HBM bytes); weights, BN statistics, parameter gradients and the
        # detection heads seen by loss / decode stay fp32.
        assert act_dtype in (torch.float32, torch.bfloat16), act_dtype
        self.adt = act_dtype
        self.bf16 = act_dtype == torch.bfloat16
        self.eb = 2 if self.bf16 else 4
        assert H % 32 == 0 and W % 32 == 0, "input height/width must be multiples of 32 (got %dx%d)" % (H, W)
        g = net.graph
        self.stream = _vp(0)
        self.x_ptr = _vp(0)
        self.timing = None
        self.routes = {"fwd": {}, "bwd": {}}     # call index -> kernel family the dispatcher actually took (recorded at the first replay)
        self._recorded = set()                   # (which, begin, end) segments whose first replay has been recorded
        self.reducer = None
        # optional hipGraph replay (MNY_HIPGRAPH=1): after two eager steps (which also run every one-time HIP attribute /
        # occupancy query) the call lists are captured once per segment and replayed.  Measured on MI355X: no gain
        # (68.1 vs 67.7 ms/step) — the eager list already keeps the GPU queue full — so it is off by default.
        self.use_graphs = training and os.environ.get("MNY_HIPGRAPH", "0") == "1"
        # weight-gradient kernels on a side stream: they feed nothing downstream in the backward pass, so they can overlap the
        # kernels of the following layers.  Fork = the side stream waits for what the main stream has enqueued (dY ready); join
        # before every batched combine and at the end of every replayed segment.  Measured (same-box A/B): no gain at
        # bs=256/352x352, where every kernel fills the chip (49.4 vs 49.5-49.8 ms), +3 % on MobileNetV3 512x512 bs=64 bf16, whose
        # 10-100 us kernels leave CUs idle (17.7 -> 17.2 ms).  Round 4 (same-box, un-bracketed steps): bs 256 / 352x352 38.19 -> 37.98 and
        # 38.07 -> 37.91 ms, so it is on up to 48 M input pixels; fewer resident depthwise-backward workgroups to make room for the side
        # kernels lose more than the overlap gains (MNY_DWB_RES=512: +0.3 ms).  A step whose launches are bracketed with HIP events runs
        # single-stream (the brackets must mean something): bench.py brackets every 4th timed step (--bracket-every), so three quarters of
        # the steps behind its `value` run with the side stream, one quarter without.  MNY_SIDE_STREAM=0/1 forces.
        env_side = os.environ.get("MNY_SIDE_STREAM")
        auto_side = N * H * W <= 48 * 1000 * 1000
        self.side_on = training and not self.use_graphs and (env_side == "1" or (env_side is None and auto_side))
        self.stream_side = _vp(0)
        self._side_stream = torch.cuda.Stream(dev) if self.side_on else None
        self._ev_fork = torch.cuda.Event() if self.side_on else None
        self._ev_join = torch.cuda.Event() if self.side_on else None
        # a second side stream for the SHORT dependent kernels of the low-rank BN backward (mny_lr_prep): they must not queue behind the weight
        # gradients of the first one, and the main stream waits for exactly them (its own event pair)
        self.stream_side2 = _vp(0)
        self._side2_stream = torch.cuda.Stream(dev) if self.side_on else None
        self._ev_fork2 = torch.cuda.Event() if self.side_on else None
        self._ev_join2 = torch.cuda.Event() if self.side_on else None
        self._side_used = False
        self.graphs = {}
        self.eager_steps = 0
        self.x_static = None
        self.fwd = CallList()
        self.cvt_batch = os.environ.get("MNY_NO_CBATCH") != "1"
        self._cvt_jobs = []
        self._cut_jobs = []           # (fp32 matrix, plane buffer): weights of the six-product GEMMs, cut once per pass (mny_cut3_batch)
        self.head32 = {}         # bf16 storage: value id -> fp32 copy of a detection head
        self.units = {}          # value id -> _Unit
        self.reals = {}          # value id -> tensor
        f32 = dict(device=dev, dtype=torch.float32)
        act = dict(device=dev, dtype=self.adt)
        K = self.K
        eb = self.eb
        max_parts = _lib.query("mny_max_parts")
        maxC = max(v.C for v in g.values)
        self.stats_ws = torch.empty(max_parts * 2 * maxC, **f32)
        P = net.param_tensors
        self.param_ptrs = [(t, t.data_ptr()) for t in net.all_state_tensors()]

        def shape(v):
            return (N, H // v.down, W // v.down, v.C)

        def view(v):
            if v.kind == "unit":
                u = self.units[v.id]
                assert u.Y is not None, "%s is the un-materialised expand output of an exdw unit" % v.name
                return (u.Y, u.scale, u.shift, v.act)
            return (self.reals[v.id], None, None, ACT_NONE)

        self._view = view
        self._shape = shape
        # expand + depthwise units (csrc/exdw.hip): a thin expand conv (16 / 24 / 32 -> 6x channels, ReLU6) whose ONLY consumer is a 3x3
        # stride-2 depthwise conv (ReLU6) runs as one unit that never writes the 6x-wide tensor or its gradient: every pass recomputes it
        # from the thin input.  exdw_pw[expand node out id] = depthwise node, exdw_dw[depthwise node out id] = expand node.
        self.exdw_pw, self.exdw_dw = {}, {}
        if not self.bf16 and not self.frozen:           # (the frozen-BatchNorm backward runs on the generic, materialised kernels)
            ks = set(int(v) for v in os.environ.get("MNY_EXDW_K", "16,24").split(",") if v)
            cons = {}
            for nd in g.nodes:
                for v in nd.ins:
                    cons.setdefault(v.id, []).append(nd)
            out_ids = {v.id for v in list(g.outputs) + ([g.seg_out] if g.seg_out is not None else [])}
            for nd in g.nodes:
                if nd.op != "pw" or nd.out.id in out_ids or len(cons.get(nd.out.id, ())) != 1:
                    continue
                d = cons[nd.out.id][0]
                i = nd.ins[0]
                ish = shape(i)
                if (d.op == "dw" and d.k == 3 and d.stride == 2 and nd.out.act == _lib.ACT_RELU6 and d.out.act == _lib.ACT_RELU6 and not nd.bias
                        and i.act in (ACT_NONE, _lib.ACT_RELU6, _lib.ACT_LEAKY, _lib.ACT_RELU) and i.C in ks
                        and _lib.query("mny_exdw_supported", N, ish[1], ish[2], i.C, nd.out.C, 2) == 1):
                    self.exdw_pw[nd.out.id] = d
                    self.exdw_dw[d.out.id] = nd
        # per-pixel gates as one unit (csrc/gate.hip, bf16 storage): t -> pw(C -> C/4, ReLU) -> pw(C/4 -> C, h-sigmoid) -> t * gate [-> + shortcut]
        # (mobilenetv3.py:26-41,69-72).  gates[emit node out id] = dict(t, se0, se3, mul, add): the two hidden units are skipped where they stand
        # and the whole unit is emitted at the multiply (or, when the residual add is its only consumer, at the add, which it absorbs).
        self.gates, self.gate_units, self.gate_absorbed = {}, {}, set()
        if self.bf16 and not self.frozen and os.environ.get("MNY_NO_GATE") != "1":
            cons = {}
            for nd in g.nodes:
                for v in nd.ins:
                    cons.setdefault(v.id, []).append(nd)
            out_ids = {v.id for v in list(g.outputs) + ([g.seg_out] if g.seg_out is not None else [])}
            for nd in g.nodes:
                if nd.op != "mul":
                    continue
                tv_, sv = nd.ins
                se3 = sv.node
                se0 = se3.ins[0].node if (se3 is not None and se3.op == "pw") else None
                if (se0 is None or se0.op != "pw" or se0.ins[0] is not tv_ or tv_.kind != "unit" or tv_.act != ACT_NONE or sv.act != _lib.ACT_HSIGMOID
                        or se0.out.act != _lib.ACT_RELU or se0.bias or se3.bias or len(cons.get(se0.out.id, ())) != 1 or len(cons.get(sv.id, ())) != 1
                        or len(cons.get(tv_.id, ())) != 2 or tv_.id in out_ids or se0.out.id in out_ids or sv.id in out_ids or nd.out.id in out_ids):
                    continue                                    # (t feeds the gate and the multiply, nothing else: the unit's backward yields its whole gradient)
                tsh = shape(tv_)
                if _lib.query("mny_gate_supported", tsh[0] * tsh[1] * tsh[2], tv_.C, se0.out.C) != 1:
                    continue
                gate = dict(t=tv_, se0=se0, se3=se3, mul=nd, add=None,
                            wq=torch.empty(int(_lib.query("mny_gate_wq_bytes", tv_.C, se0.out.C)), device=dev, dtype=torch.uint8))
                emit_at = nd
                mc = cons.get(nd.out.id, [])
                if len(mc) == 1 and mc[0].op == "add" and mc[0].k == 1 and mc[0].ins[0] is nd.out and os.environ.get("MNY_GATE_NOADD") != "1":
                    gate["add"] = mc[0]                         # out = t * gate + view(other operand): the add node is absorbed
                    emit_at = mc[0]
                    self.gate_absorbed.add(nd.out.id)
                self.gates[emit_at.out.id] = gate
                self.gate_units[se0.out.id] = gate
                self.gate_units[se3.out.id] = gate
        # Nodes no loss (or detection output) depends on — MobileNetV2-YOLO's always-on seg branch under a config without a `seg` section
        # (mbv2_yolo.py:155-156: computed, BatchNorm running statistics updated, result dropped) — feed nothing on the main stream: in a
        # training plan they run on the SIDE stream next to the rest of the forward pass (their own statistics workspace), joined behind the heads.
        live = set()
        stack = list(g.outputs) + ([g.seg_out] if g.seg_out is not None else [])
        while stack:
            v = stack.pop()
            if v.node is None or v.id in live:
                continue
            live.add(v.id)
            stack.extend(v.node.ins)
        self.dead_side = bool(self.side_on and bn_batch and os.environ.get("MNY_NO_DEAD_SIDE") != "1"
                              and any(nd.out.id not in live and nd.op in ("dw", "pw", "add") for nd in g.nodes))
        self.stats_ws_side = torch.empty(max_parts * 2 * maxC, **f32) if self.dead_side else self.stats_ws
        dead_forked = False
        # A second forward LANE (round 6): the stride-16 neck / head chain (conv_for_S16, connect_for_S16, yolo_headS16: mbv2_yolo.py:146-153,
        # mbv3_yolo.py:133-138) is independent of the stride-32 chain (features2 / bneck2, conv_for_S32, connect_for_S32, yolo_headS32) except for
        # ONE edge (the upsampled stride-32 feature).  Every conv of a chain is followed by its BatchNorm finalize — a dependent ~6 us launch + the
        # dispatch gap behind it — so two chains on two streams fill each other's bubbles.  Lane-2 nodes run on the second side stream with their own
        # statistics workspace; a lane-2 node whose input was produced on the main stream since the last fork waits for it (fork2), the main stream
        # waits for lane 2 before the losses (join2).  MEASURED SLOWER (same-box A/B, parity-green): headline 34.73 / 34.78 -> 34.78 / 34.89 ms,
        # MobileNetV3 512x512 bf16 13.86 / 13.85 -> 13.99 / 14.02 ms — the two chains' kernels share the CUs and the cross-stream waits cost more than the
        # finalize bubbles they fill.  OFF by default; MNY_LANE2=1 turns it on for A/B.
        lane2_on = bool(self.side_on and bn_batch and os.environ.get("MNY_LANE2") == "1")
        lane2_prefix = ("conv_for_S16", "connect_for_S16", "yolo_headS16")
        self.stats_ws_lane2 = torch.empty(max_parts * 2 * maxC, **f32) if lane2_on else self.stats_ws
        made = {}                     # value id -> (lane, index in the call list behind which it exists)
        prev_node = None
        last_fork2 = -1
        lane2_used = False
        for nd in g.nodes:
            o = nd.out
            shp = shape(o)
            M = shp[0] * shp[1] * shp[2]
            dead = self.dead_side and o.id not in live and nd.op in ("dw", "pw", "add") and o.id not in self.exdw_pw and o.id not in self.exdw_dw
            if prev_node is not None:
                made[prev_node[0]] = (prev_node[1], len(self.fwd.calls))   # the previous node's value exists behind the calls appended so far
            path = nd.conv or (nd.out.name if nd.op in ("add", "partadd") else "")
            lane2 = (lane2_on and not dead and nd.op in ("dw", "pw", "pwb", "add", "partadd") and path.startswith(lane2_prefix)
                     and o.id not in self.exdw_pw and o.id not in self.exdw_dw and o.id not in self.gates and o.id not in self.gate_units
                     and o.id not in self.gate_absorbed)
            st_n = self.stream_side if dead else (self.stream_side2 if lane2 else self.stream)          # the stream and statistics workspace of this node's calls
            sws_n = self.stats_ws_side if dead else (self.stats_ws_lane2 if lane2 else self.stats_ws)
            if dead and not dead_forked:
                self.fwd.add_py(self._fork_side, "fork")                # the side stream waits for what the main stream has enqueued (the branch's input)
                dead_forked = True
            if lane2:
                if any(made.get(v.id, (0, -1))[0] == 0 and made.get(v.id, (0, -1))[1] > last_fork2 for v in nd.ins) or not lane2_used:
                    self.fwd.add_py(self._fork_side2, "fork")           # an input made on the main stream since the last fork: lane 2 waits for it
                    last_fork2 = len(self.fwd.calls)
                lane2_used = True
            elif not dead and any(made.get(v.id, (0, -1))[0] == 2 for v in nd.ins):
                self.fwd.add_py(self._join_side2, "join")               # (no graph of this package has such an edge; kept correct for any)
                for k_ in [k_ for k_, mv in made.items() if mv[0] == 2]:
                    made[k_] = (0, made[k_][1])
            prev_node = (o.id, 2 if lane2 else (1 if dead else 0))
            if nd.op == "pw" and o.id in self.gate_units:
                u = _Unit()                                   # a hidden unit of a gate: its BN coefficients live here, the calls come with the gate
                u.Y = None                                    # never materialised
                u.coef4 = torch.empty(4, o.C, **f32)
                u.scale, u.shift, u.mean, u.invstd = u.coef4[0], u.coef4[1], u.coef4[2], u.coef4[3]
                u.act, u.C, u.M, u.shape = o.act, o.C, M, shp
                self.units[o.id] = u
                continue
            if o.id in self.gate_absorbed:
                continue                                      # the multiply of a gate whose residual add absorbs it
            if o.id in self.gates:
                self._emit_gate_forward(self.gates[o.id], nd, bn_batch)
                continue
            if nd.op in ("stem", "dw", "pw"):
                u = _Unit()
                u.Y = torch.empty(shp, **act) if o.id not in self.exdw_pw else None      # the expand output of an exdw unit is never materialised
                u.coef4 = torch.empty(4, o.C, **f32)
                u.scale, u.shift, u.mean, u.invstd = u.coef4[0], u.coef4[1], u.coef4[2], u.coef4[3]
                u.act, u.C, u.M, u.shape = o.act, o.C, M, shp
                self.units[o.id] = u
                w = P[nd.conv + ".weight"]
                stats = sws_n if bn_batch else None
                if nd.op == "stem":
                    parts = _lib.query("mny_stem_stat_parts", N, H, W, o.C)
                    self.fwd.add(K("mny_stem_fwd"), self.x_ptr, w, u.Y, stats, N, H, W, o.C, self.stream,
                                 meta=dict(flops=2 * M * o.C * 27, bytes=4 * N * 3 * H * W + eb * M * o.C))
                elif nd.op == "dw" and o.id in self.exdw_dw:
                    pn = self.exdw_dw[o.id]                  # the expand node: x = its input (a view), e = its BN coefficients
                    pi, pu = pn.ins[0], self.units[pn.out.id]
                    psh = shape(pi)
                    xv = view(pi)
                    parts = _lib.query("mny_exdw_fwd_parts", N, psh[1], psh[2], pi.C, o.C, 2)
                    Mx = N * psh[1] * psh[2]
                    self.fwd.add("mny_exdw_fwd", xv[0], xv[1], xv[2], xv[3], P[pn.conv + ".weight"], pu.scale, pu.shift, w, u.Y, stats,
                                 N, psh[1], psh[2], pi.C, o.C, 2, self.stream,
                                 meta=dict(flops=2 * Mx * pi.C * o.C + 2 * M * o.C * 9, bytes=eb * (Mx * pi.C + M * o.C), shape="exdw K%d C%d H%d" % (pi.C, o.C, psh[1])))
                elif nd.op == "dw":
                    i = nd.ins[0]
                    ish = shape(i)
                    xv = view(i)
                    parts = _lib.query("mny_dw_stat_parts_x", N, ish[1], ish[2], o.C, nd.k, nd.stride, 1 if self.bf16 else 0)
                    self.fwd.add(K("mny_dw_fwd"), xv[0], xv[1], xv[2], xv[3], w, u.Y, stats, N, ish[1], ish[2], o.C, nd.k, nd.stride, st_n,
                                 meta=dict(flops=2 * M * o.C * nd.k * nd.k, bytes=eb * (N * ish[1] * ish[2] * o.C + M * o.C) + 4 * o.C * nd.k * nd.k,
                                           shape="C%d H%d s%d" % (o.C, ish[1], nd.stride)))
                elif o.id in self.exdw_pw:
                    i = nd.ins[0]
                    xv = view(i)
                    parts = _lib.query("mny_exdw_stat_parts", M, i.C, o.C)
                    if bn_batch:                             # batch statistics of the un-materialised expand output (eval plans use the running ones)
                        self.fwd.add("mny_exdw_stats", xv[0], xv[1], xv[2], xv[3], w, stats, M, i.C, o.C, self.stream,
                                     meta=dict(flops=2 * M * i.C * (i.C + 1), bytes=eb * M * i.C,      # X^T X and colsum(X): one read of the thin X
                                               shape="exdw stats M%d K%d N%d" % (M, i.C, o.C)))
                else:
                    i = nd.ins[0]
                    xv = view(i)
                    parts = _lib.query(K("mny_pw_stat_parts"), M, i.C, o.C)
                    w6 = self._w6_planes(w, M, i.C, o.C)
                    self.fwd.add("mny_pw_fwd_w6" if w6 is not None else K("mny_pw_fwd"), xv[0], xv[1], xv[2], xv[3],
                                 w6 if w6 is not None else self._gemm_weight(w), None, None, u.Y, stats, M, i.C, o.C, st_n, label=K("mny_pw_fwd"),
                                 meta=dict(flops=2 * M * i.C * o.C, bytes=eb * (M * i.C + M * o.C) + 4 * i.C * o.C, shape="M%d K%d N%d" % (M, i.C, o.C)))
                gam, bet = P[nd.bn + ".weight"], P[nd.bn + ".bias"]
                rm, rv = P[nd.bn + ".running_mean"], P[nd.bn + ".running_var"]
                if bn_batch:
                    self.fwd.add("mny_bn_finalize", sws_n, parts, M, gam, bet, BN_EPS, BN_MOMENTUM, rm, rv,
                                 u.scale, u.shift, u.mean, u.invstd, o.C, st_n)
                else:
                    self.fwd.add("mny_bn_eval_coeffs", gam, bet, rm, rv, BN_EPS, u.scale, u.shift, o.C, self.stream)
                    if self.frozen:                          # the backward kernels' yhat = (y - running_mean) * invstd
                        self.fwd.add("mny_bn_eval_stats", rm, rv, BN_EPS, u.mean, u.invstd, o.C, self.stream)
            elif nd.op == "pwb":
                i = nd.ins[0]
                xv = view(i)
                t = torch.empty(shp, **act)
                self.reals[o.id] = t
                w6 = self._w6_planes(P[nd.conv + ".weight"], M, i.C, o.C)
                self.fwd.add("mny_pw_fwd_w6" if w6 is not None else K("mny_pw_fwd"), xv[0], xv[1], xv[2], xv[3],
                             w6 if w6 is not None else self._gemm_weight(P[nd.conv + ".weight"]), P[nd.conv + ".bias"], None, t, None,
                             M, i.C, o.C, st_n, label=K("mny_pw_fwd"),
                             meta=dict(flops=2 * M * i.C * o.C, bytes=eb * (M * i.C + M * o.C) + 4 * i.C * o.C, shape="M%d K%d N%d" % (M, i.C, o.C)))
                if self.bf16:                    # loss / decode read the head in fp32
                    t32 = torch.empty(shp, **f32)
                    self.fwd.add("mny_cvt_bf16_f32", t, t32, t.numel(), st_n)
                    self.head32[o.id] = t32
                if g.outputs and o is g.outputs[0]:
                    self._head0_call = self.fwd.calls[-1]        # the first head exists behind THIS call (found again by identity: batched launches are inserted at the head of the list later)
            elif nd.op == "add":
                a = view(nd.ins[0])
                has_b, has_up = nd.k & 1, nd.k & 2
                b = view(nd.ins[1]) if has_b else (None, None, None, ACT_NONE)
                up = self.reals[nd.ins[-1].id] if has_up else None
                if has_up:
                    assert nd.ins[-1].kind == "real"
                t = torch.empty(shp, **act)
                self.reals[o.id] = t
                self.fwd.add(K("mny_add_views"), a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], up, t, shp[0], shp[1], shp[2], shp[3], st_n)
            elif nd.op == "mul":
                a, b = view(nd.ins[0]), view(nd.ins[1])
                t = torch.empty(shp, **act)
                self.reals[o.id] = t
                self.fwd.add(K("mny_mul_views"), a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], t, M, o.C, self.stream)
            elif nd.op == "partadd":
                a = view(nd.ins[0])
                up = self.reals[nd.ins[1].id]
                t = torch.empty(shp, **act)
                self.reals[o.id] = t
                self.fwd.add(K("mny_partadd_up"), a[0], a[1], a[2], a[3], up, t, shp[0], shp[1], shp[2], nd.ins[0].C, o.C, st_n)
            else:
                raise AssertionError(nd.op)

        if lane2_used:
            self.fwd.add_py(self._join_side2, "join")                   # both lanes meet before the losses / the decode
        self._flush_cvt_jobs()
        self._flush_cut_jobs(self.fwd, at_head=True)
        if self.gates:                                        # the gates' weights as matrix-core operand chunks, one launch per pass for all of them
            import numpy as np
            gl = list(self.gates.values())
            jt = np.array([(P[gt["se0"].conv + ".weight"].data_ptr(), P[gt["se3"].conv + ".weight"].data_ptr(), gt["wq"].data_ptr(), gt["t"].C, gt["se0"].out.C) for gt in gl],
                          dtype=np.dtype([("w1", np.uint64), ("w2", np.uint64), ("wq", np.uint64), ("C", np.int32), ("R", np.int32)]))
            self.gate_jobs = torch.from_numpy(jt.view(np.uint8).copy()).to(dev)
            self.fwd.add("mny_gate_cut_batch_bf16", self.gate_jobs, len(gl), self.stream)
            self.fwd.calls.insert(0, self.fwd.calls.pop())
        self.heads = [self.head32.get(o.id, self.reals[o.id]) for o in g.outputs]
        self.seg_head = self.head32.get(g.seg_out.id, self.reals[g.seg_out.id]) if g.seg_out is not None else None
        self.loss_outputs = list(g.outputs) + ([g.seg_out] if g.seg_out is not None else [])     # values a loss reads
        self._build_detection()
        if dead_forked or getattr(self, "_loss_side", False):
            self.fwd.add_py(self._join_side, "join")                    # the dead branch's tensors (and the workspace it used) are free again before anything reuses them
        if training and (bn_batch or self.frozen):
            self._build_backward()

    # ------------------------------------------------------------------------------------------
    def _emit_gate_forward(self, gate, nd, bn_batch):
        """The forward calls of a per-pixel gate (csrc/gate.hip) at node `nd` (the multiply, or the residual add that absorbs it)."""
        P = self.net.param_tensors
        t, se0, se3 = gate["t"], gate["se0"], gate["se3"]
        u1, u2 = self.units[se0.out.id], self.units[se3.out.id]
        tsh = self._shape(t)
        M, C, R = tsh[0] * tsh[1] * tsh[2], t.C, se0.out.C
        tv = self._view(t)
        wq = gate["wq"]
        out = torch.empty(self._shape(nd.out), device=self.dev, dtype=self.adt)
        self.reals[nd.out.id] = out
        parts = _lib.query("mny_gate_parts", M)
        eb = self.eb
        for un, hn, Cn in ((u1, se0, R), (u2, se3, C)):
            gam, bet = P[hn.bn + ".weight"], P[hn.bn + ".bias"]
            rm, rv = P[hn.bn + ".running_mean"], P[hn.bn + ".running_var"]
            if bn_batch:
                if hn is se0:
                    self.fwd.add("mny_gate_stats1_bf16", tv[0], tv[1], tv[2], wq, self.stats_ws, M, C, R, self.stream,
                                 meta=dict(flops=2 * M * C * R, bytes=eb * M * C, shape="gate stats1 M%d C%d R%d" % (M, C, R)))
                else:
                    self.fwd.add("mny_gate_stats2_bf16", tv[0], tv[1], tv[2], wq, u1.scale, u1.shift, self.stats_ws, M, C, R, self.stream,
                                 meta=dict(flops=4 * M * C * R, bytes=eb * M * C, shape="gate stats2 M%d C%d R%d" % (M, C, R)))
                self.fwd.add("mny_bn_finalize", self.stats_ws, parts, M, gam, bet, BN_EPS, BN_MOMENTUM, rm, rv, un.scale, un.shift, un.mean, un.invstd, Cn, self.stream)
            else:
                self.fwd.add("mny_bn_eval_coeffs", gam, bet, rm, rv, BN_EPS, un.scale, un.shift, Cn, self.stream)
        a = (None, None, None, ACT_NONE)
        if gate["add"] is not None:
            a = self._view(gate["add"].ins[1])
        self.fwd.add("mny_gate_fwd_bf16", tv[0], tv[1], tv[2], wq, u1.scale, u1.shift, u2.scale, u2.shift, a[0], a[1], a[2], a[3], out,
                     M, C, R, self.stream,
                     meta=dict(flops=4 * M * C * R, bytes=eb * M * C * (3 if a[0] is not None else 2), shape="gate M%d C%d R%d%s" % (M, C, R, " +add" if a[0] is not None else "")))

    # ------------------------------------------------------------------------------------------
    def _build_detection(self):
        net, dev, N = self.net, self.dev, self.N
        f32 = dict(device=dev, dtype=torch.float32)
        self.hp, self.anchors, self.masks = [], [], []
        img = [self.H, self.W]                                  # mbv2_yolo.py:139-140 (Q8: [H,W] against (w,h) anchors)
        for hi, hs in enumerate(net.yolo_losses):
            g = self.heads[hi].shape[1]
            assert self.heads[hi].shape[1] == self.heads[hi].shape[2], "square grids only (yolo_loss.py:71)"
            hs.img_size = img
            anchors = torch.tensor([(aw / img[0], ah / img[1]) for aw, ah in hs.anchors], dtype=torch.float32).to(dev)
            self.anchors.append(anchors)
            self.masks.append(torch.tensor(hs.mask, dtype=torch.int32).to(dev))
            self.hp.append(YoloHead(N, g, len(hs.mask), hs.num_classes, len(hs.anchors), hs.ignore_threshold, hs.iou_thresh, hs.iou_weighting))
        if self.training:
            self.out14 = torch.zeros(2, 7, **f32)
            self.dheads = [torch.empty_like(h) for h in self.heads]
            self.t_ptr, self.off_ptr = _vp(0), _vp(0)
            self.loss_ws = []
            for hi in range(2):
                nbytes = _lib.query("mny_yolo_loss_ws_bytes", ctypes.byref(self.hp[hi]), 0)
                ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
                self.loss_ws.append(ws)
                # the first head's loss (five short, latency-bound launches) runs on the side stream from the moment that head exists, next to the
                # second head's branch (round 6; joined at the end of the list); MNY_NO_LOSS_SIDE=1: both losses at the end, on the main stream
                pos = next((k + 1 for k, c in enumerate(self.fwd.calls) if c is getattr(self, "_head0_call", None)), None)
                early = (hi == 0 and self.side_on and self.bn_batch and pos is not None and pos < len(self.fwd.calls)
                         and os.environ.get("MNY_NO_LOSS_SIDE") != "1")
                self.fwd.add("mny_yolo_loss", self.heads[hi], self.t_ptr, self.off_ptr, self.anchors[hi], self.masks[hi],
                             ctypes.byref(self.hp[hi]), self.out14[hi], self.dheads[hi], ws, self.stream_side if early else self.stream)
                if early:
                    loss_call = self.fwd.calls.pop()
                    self.fwd.add_py(self._fork_side, "fork")
                    fork_call = self.fwd.calls.pop()
                    self.fwd.calls[pos:pos] = [fork_call, loss_call]
                    self._loss_side = True
            self.t_dev = torch.zeros(max(4 * N, 64), 5, **f32)
            self.off_dev = torch.zeros(N + 1, device=dev, dtype=torch.int32)
            if self.seg_head is not None:             # mbv2_yolo.py:167-170: SegLoss on the raw seg head
                self.seg_out3 = torch.zeros(3, **f32)
                self.seg_maps = torch.zeros_like(self.seg_head)                 # resident copy of the batch's seg_maps [N,h,w,C]
                self.dheads.append(torch.empty_like(self.seg_head))
                self.seg_ws = torch.empty(max(_lib.query("mny_seg_loss_ws_bytes", self.seg_head.numel()), 8), device=dev, dtype=torch.uint8)
                self.fwd.add("mny_seg_loss", self.seg_head, self.seg_maps, self.seg_head.numel(), self.seg_out3, self.dheads[2], self.seg_ws, self.stream)
        else:
            C = net.num_classes
            self.cells = [hp.A * hp.g * hp.g for hp in self.hp]
            self.cap = sum(self.cells)
            self.rows = torch.zeros(N, self.cap, 7, **f32)
            self.cnt0 = torch.zeros(N, device=dev, dtype=torch.int32)
            self.cnt1 = torch.zeros(N, device=dev, dtype=torch.int32)
            self.seg_begin = (torch.arange(N, dtype=torch.int32) * self.cap).to(dev)
            self.out_idx = torch.zeros(N * self.cap, device=dev, dtype=torch.int32)
            self.out_counts = torch.zeros(N, device=dev, dtype=torch.int32)
            self.out_rows = torch.zeros(N * self.cap, 7, **f32)
            self.nms_ws = torch.empty(_lib.query("mny_nms_ws_bytes", N, N * self.cap, C), device=dev, dtype=torch.uint8)
            self.val_conf = [ctypes.c_float(0.1), ctypes.c_float(0.1)]
            self.det = CallList()
            self.det.add("mny_yolo_decode", self.heads[0], self.anchors[0], self.masks[0], ctypes.byref(self.hp[0]), self.val_conf[0],
                         self.rows, self.cap, None, self.cnt0, self.stream)
            self.det.add("mny_yolo_decode", self.heads[1], self.anchors[1], self.masks[1], ctypes.byref(self.hp[1]), self.val_conf[1],
                         self.rows, self.cap, self.cnt0, self.cnt1, self.stream)
            self.det.add("mny_nms_per_class", self.rows, self.seg_begin, self.cnt1, N, N * self.cap, self.cap, C, ctypes.c_double(0.45),
                         self.out_idx, self.out_counts, self.out_rows, self.nms_ws, self.stream)
            so = _lib.query("mny_nms_status_offset", N, N * self.cap, C)
            self.nms_status = self.nms_ws[so:so + 4].view(torch.int32)
            if self.seg_head is not None:             # mbv2_yolo.py:161-164 -> seg_loss.py:77-80: sigmoid of image 0, [C,h,w]
                _, sh, sw, sc = self.seg_head.shape
                self.seg_eval = torch.zeros(sc, sh, sw, **f32)
                self.det.add("mny_seg_sigmoid", self.seg_head, sh, sw, sc, self.seg_eval, self.stream)

    # ------------------------------------------------------------------------------------------
    def _build_backward(self):
        net, g, N = self.net, self.net.graph, self.N
        dev = self.dev
        f32 = dict(device=dev, dtype=torch.float32)
        act = dict(device=dev, dtype=self.adt)
        K, eb = self.K, self.eb
        P = net.param_tensors
        bwd = self.bwd = CallList()
        view, shape = self._view, self._shape

        # which nodes lie on a path to a loss
        needed = set()
        outs = self.loss_outputs
        stack = [o for o in outs]
        while stack:
            v = stack.pop()
            if v.node is None or v.id in needed:
                continue
            needed.add(v.id)
            stack.extend(v.node.ins)
        order = [nd for nd in reversed(g.nodes) if nd.out.id in needed]

        # flat gradient arena in backward production order (heads first -> DP buckets complete early)
        self.grad_params = []
        off = 0
        slots = {}
        # detection heads: dL/dhead rows are padded to `head_cp` channels for the backward GEMMs (mny_pad_rows); the pad rows of
        # dW / pad entries of dbias land in slack reserved behind the parameter's gradient slot
        align = 8 if self.bf16 else 4
        self.head_cp = {nd.out.id: (nd.out.C + align - 1) // align * align for nd in order
                        if nd.op == "pwb" and nd.out.C % align and os.environ.get("MNY_NO_HEADPAD") != "1"}
        slack = {}
        for nd in order:
            if nd.out.id in self.head_cp:
                cp = self.head_cp[nd.out.id]
                slack[nd.conv + ".weight"] = cp * nd.ins[0].C
                slack[nd.conv + ".bias"] = cp
        for nd in order:
            names = []
            if nd.conv:
                names.append(nd.conv + ".weight")
                if nd.bias:
                    names.append(nd.conv + ".bias")
            if nd.bn:
                names += [nd.bn + ".weight", nd.bn + ".bias"]
            for nm in names:
                if nm in slots:
                    continue                     # module applied twice (mbv3_yolo.py:133-134): one slot, two contributions
                n = P[nm].numel()
                slots[nm] = (off, n)
                self.grad_params.append(nm)
                off += (max(n, slack.get(nm, 0)) + 3) // 4 * 4
        self.gflat = torch.zeros(off, **f32)
        self.gviews = {nm: self.gflat[o:o + n].view(P[nm].shape) for nm, (o, n) in slots.items()}
        self.grad_slots = slots

        written = set()
        self.shared_tmp = []
        fin_name = "mny_bn_bwd_finalize_frozen" if self.frozen else "mny_bn_bwd_finalize"

        def gv(nm):
            """Destination of a parameter gradient.  The first contribution writes the arena slot; a later one (shared
            module) writes a scratch tensor that `flush_shared()` adds into the slot right after the producing call."""
            if nm not in written:
                written.add(nm)
                return self.gviews[nm]
            tmp = torch.empty_like(self.gviews[nm])
            self.shared_tmp.append((tmp, self.gviews[nm]))
            return tmp

        def flush_shared():
            while self.shared_tmp:
                tmp, dst = self.shared_tmp.pop()
                bwd.add("mny_axpy", tmp, None, dst, 1, tmp.numel(), self.stream)

        # Deferred partial combines: every weight-gradient call leaves per-workgroup partial sums; instead of one small combine
        # launch per layer (70 launches of 8-10 us at bs=256, 0.66 ms/step) the layer gets its OWN partial buffer, is called with
        # dw = NULL, and a whole run of layers is combined by one mny_reduce_batch launch (flush_reduce: every `defer_every` jobs, so
        # the data-parallel buckets still complete early).  MNY_NO_DEFER=1: the per-layer combines.
        self.defer = os.environ.get("MNY_NO_DEFER") != "1"
        defer_every = int(os.environ.get("MNY_DEFER_EVERY", "16"))
        self._red_jobs, self._red_keep, self._post_reduce = [], [], []
        uses = {}
        for nd in order:                                          # a module applied twice (mbv3_yolo.py:133-134) adds a second contribution right
            if nd.conv:                                            # after its producing call: its combines stay inline
                uses[nd.conv] = uses.get(nd.conv, 0) + 1
        single = lambda nd: uses.get(nd.conv, 0) == 1             # noqa: E731

        def defer_job(n_floats, dest, nparts, n):
            """-> private partial buffer of a deferred combine writing `dest` (an arena view)."""
            wsl = torch.empty(max(int(n_floats), 1), **f32)
            self._red_jobs.append((wsl, dest, int(nparts), int(n)))
            return wsl

        def flush_reduce(force=False):
            if not self._red_jobs or (not force and len(self._red_jobs) < defer_every):
                return
            import numpy as np
            jobs, block_job = [], []
            for wsl, dest, nparts, n in self._red_jobs:
                jobs.append((wsl.data_ptr(), dest.data_ptr(), n, nparts, len(block_job)))
                block_job += [len(jobs) - 1] * ((n + 31) // 32)
            jt = np.array(jobs, dtype=np.dtype([("parts", np.uint64), ("out", np.uint64), ("n", np.int64), ("nparts", np.int32), ("b0", np.int32)]))
            jdev = torch.from_numpy(jt.view(np.uint8).copy()).to(dev)
            bdev = torch.tensor(block_job, dtype=torch.int32, device=dev)
            self._red_keep += [t for job in self._red_jobs for t in job[:2]]
            # The combine feeds nothing downstream in the backward pass: it runs on the SIDE stream, behind the side-stream weight gradients
            # whose partial rows it reads (same stream = ordered) and, through the fork, behind everything the main stream has enqueued (the
            # fused units leave their partial rows there).  Every replayed segment ends with a join (run_bwd_segment), so gradients are
            # complete before an all-reduce or the optimizer sees them.  (Round 4 joined here and ran the combine on the main stream: 0.3 ms
            # of 4-5 launches on the critical path of both benchmark configurations.)  MNY_REDUCE_MAIN=1: the round-4 placement.
            if self.side_on and os.environ.get("MNY_REDUCE_MAIN") != "1":
                bwd.add_py(self._fork_side, "fork")
                red_stream = self.stream_side
            else:
                if self.side_on:
                    bwd.add_py(self._join_side, "join")           # the partials of side-stream weight gradients must have landed
                red_stream = self.stream
            bwd.add("mny_reduce_batch", jdev, bdev, len(block_job), red_stream, meta=dict(writes=[job[1].data_ptr() for job in self._red_jobs]))
            for hook in self._post_reduce:                        # corrections of combined weight gradients (low-rank BN backward): same stream, right behind the combine
                hook(red_stream)
            self._red_jobs, self._post_reduce = [], []

        # workspaces shared by all layers (single stream => sequential use)
        ws_floats = 1
        max_parts = _lib.query("mny_max_parts")
        for nd in order:
            o = nd.out
            shp = shape(o)
            M = shp[0] * shp[1] * shp[2]
            if nd.op in ("pw", "pwb"):
                ws_floats = max(ws_floats, _lib.query("mny_pw_wgrad_ws_floats", M, nd.ins[0].C, self.head_cp.get(o.id, o.C)),
                                _lib.query("mny_pw_bnbwd_ws_floats", M, nd.ins[0].C, o.C))
            elif nd.op == "dw":
                ws_floats = max(ws_floats, max_parts * o.C * nd.k * nd.k)     # (max_parts = 1024 >= the 768-row grids of the fused kernels)
            elif nd.op == "stem":
                ws_floats = max(ws_floats, max_parts * o.C * 27)
        self.ws = torch.empty(ws_floats, **f32)
        self.ws_side = torch.empty(ws_floats, **f32) if self.side_on else self.ws
        maxC = max(v.C for v in g.values)
        self.red_ws = torch.empty(2048 * 2 * maxC, **f32)      # >= mny_bn_bwd_parts() rows of [2][C]
        self.coef_ws = torch.empty(3 * maxC, **f32)
        self.g_scale = torch.ones(len(outs), **f32)   # upstream dL/dloss_i, written by backward()
        self.wT = {}

        class GS:
            __slots__ = ("buf", "shared")

            def __init__(self):
                self.buf, self.shared = None, False
        gs = {v.id: GS() for v in g.values}
        self.grad_bufs = []

        def alloc(v):
            t = torch.empty(shape(v), **act)
            self.grad_bufs.append(t)
            return t

        def contribute_alias(v, buf):
            s = gs[v.id]
            if s.buf is None:
                s.buf, s.shared = buf, True
                return
            if s.shared:
                nb = alloc(v)
                bwd.add(K("mny_axpy"), s.buf, None, nb, 0, nb.numel(), self.stream)
                s.buf, s.shared = nb, False
            bwd.add(K("mny_axpy"), buf, None, s.buf, 1, s.buf.numel(), self.stream)

        def contribute_kernel(v, emit, inplace_ok=True):
            """emit(out, addend) appends the producing call.  inplace_ok=False: the kernel cannot take addend == out (the two-pass
            expand + depthwise backward overwrites `out` before it reads the addend): an existing gradient becomes the addend of a fresh buffer."""
            s = gs[v.id]
            if s.buf is None:
                s.buf, s.shared = alloc(v), False
                emit(s.buf, None)
            elif not s.shared and inplace_ok:
                emit(s.buf, s.buf)
            else:
                nb = alloc(v)
                emit(nb, s.buf)
                s.buf, s.shared = nb, False

        for hi, o in enumerate(outs):
            if o.id in self.head_cp:
                cp = self.head_cp[o.id]
                shp = shape(o)
                gp = torch.empty(shp[0], shp[1], shp[2], cp, **act)
                bwd.add(K("mny_pad_rows"), self.dheads[hi], self.g_scale[hi:hi + 1], gp, shp[0] * shp[1] * shp[2], o.C, cp, self.stream)
                gs[o.id].buf = gp
                continue
            bwd.add("mny_axpy", self.dheads[hi], self.g_scale[hi:hi + 1], self.dheads[hi], 0, self.dheads[hi].numel(), self.stream)
            if self.bf16:
                d16 = torch.empty(self.dheads[hi].shape, **act)
                bwd.add("mny_cvt_f32_bf16", self.dheads[hi], d16, d16.numel(), self.stream)
                gs[o.id].buf = d16
            else:
                gs[o.id].buf = self.dheads[hi]

        n_consumers = {v.id: 0 for v in g.values}
        for nd in g.nodes:
            for v in nd.ins:
                n_consumers[v.id] += 1
        for v in outs:
            n_consumers[v.id] += 1
        self.fused_red = {}      # value id -> (partial-sum buffer, rows) written by the data-gradient GEMM that produced the value's gradient
        self.stemdw_done = set() # stem output ids whose whole backward (BN sums, weight gradient) left with the depthwise consumer's (mny_stemdw_bwd)
        last_consumer = {}       # value id -> the consumer node whose backward runs LAST (order = reverse topological)
        for nd_ in order:
            for v_ in nd_.ins:
                last_consumer[v_.id] = nd_
        loss_ids = {v_.id for v_ in outs}

        def takes_own_sums(pn):
            """True for the thin expand units handled by mny_pw_bnbwd (their stage 1 forms the BN sums itself)."""
            if pn.op != "pw" or self.frozen or os.environ.get("MNY_NO_BNFUSE") == "1" or (self.bf16 and os.environ.get("MNY_BNFUSE_BF16") == "0"):
                return False                     # (frozen BatchNorm: that unit's own finalize assumes batch statistics)
            po, pi = pn.out, pn.ins[0]
            if pi.act in (_lib.ACT_HSWISH, _lib.ACT_HSIGMOID) or po.act in (_lib.ACT_HSWISH, _lib.ACT_HSIGMOID):
                return False
            psh = shape(po)
            return _lib.query(K("mny_pw_bnbwd_supported"), psh[0] * psh[1] * psh[2], pi.C, po.C) == 1
        def red_target(i, nd):
            """The conv+BN+act unit whose COMPLETE output gradient the data gradient of `nd` w.r.t. its input `i` is — its BN-backward sums can
            leave with that gradient (no mny_bn_bwd_reduce pass) — or None.  `i` itself when it is a unit and `nd` its last consumer; or, looking
            through a residual add (round 6): the add's unit operand that only the add consumes takes the sum's gradient unchanged (an alias)."""
            if i.id in loss_ids or last_consumer.get(i.id) is not nd:
                return None
            t = None
            if i.kind == "unit":
                t = i
            elif i.node is not None and i.node.op == "add" and os.environ.get("MNY_NO_ADDRED") != "1":
                cands = [v for v in i.node.ins[:1 + (i.node.k & 1)] if v.kind == "unit" and n_consumers[v.id] == 1 and v.id not in loss_ids]
                if len(cands) == 1:
                    t = cands[0]
            if (t is None or t.node is None or t.node.op not in ("dw", "pw") or takes_own_sums(t.node) or t.id not in self.units
                    or self.units[t.id].Y is None or gs[t.id].buf is not None and t is not i):
                return None
            return t

        # wide expand units on the low-rank BN backward (csrc/lrbwd.hip): the depthwise unit behind stores ca o G o act'(z), both GEMMs run on
        # that tensor and the BatchNorm terms are K-wide corrections — no bn_bwd_apply pass over the C-wide tensors.  lr_units: value ids
        self.lr_units = set()

        def lr_ok(pn):
            if (self.bf16 or self.frozen or not self.defer or pn.op != "pw" or pn.bias or not single(pn) or pn.out.id in self.head_cp
                    or takes_own_sums(pn) or os.environ.get("MNY_NO_LR") == "1"):
                return False
            pi = pn.ins[0]
            if pi.act != ACT_NONE or pi.kind not in ("unit", "real"):
                return False                     # the thin input's view must be linear: it is folded into the correction's operands
            psh = shape(pn.out)
            return _lib.query("mny_lr_supported", psh[0] * psh[1] * psh[2], pi.C, pn.out.C) == 1
        # W^T of every generic pointwise unit, all in one launch at the head of the backward list (one per layer was 39 launches)
        self.t_batch = os.environ.get("MNY_NO_TBATCH") != "1"
        if self.t_batch:
            import numpy as np
            jobs, block_job = [], []
            for nd in order:
                if nd.op not in ("pw", "pwb") or takes_own_sums(nd) or nd.conv in self.wT:
                    continue
                w = P[nd.conv + ".weight"]
                oc = self.head_cp.get(nd.out.id, nd.out.C)
                wT = torch.empty(nd.ins[0].C, oc, **act)
                self.wT[nd.conv] = wT
                nb = ((nd.ins[0].C + 31) // 32) * ((oc + 31) // 32)
                jobs.append((w.data_ptr(), wT.data_ptr(), nd.out.C, nd.ins[0].C, oc, len(block_job)))
                block_job += [len(jobs) - 1] * nb
            if jobs:
                jt = np.array(jobs, dtype=np.dtype([("src", np.uint64), ("dst", np.uint64), ("R", np.int32), ("Cc", np.int32), ("Rp", np.int32), ("b0", np.int32)]))
                self.t_jobs = torch.from_numpy(jt.view(np.uint8).copy()).to(dev)
                self.t_blocks = torch.tensor(block_job, dtype=torch.int32, device=dev)
                bwd.add(K("mny_transpose_batch"), self.t_jobs, self.t_blocks, len(block_job), self.stream)
                # ... and, for the data-gradient GEMMs that take the six-product bf16 form, the cut of W^T right behind it
                self.wT6 = {}
                for nd in order:
                    if nd.conv in self.wT and nd.conv not in self.wT6 and not takes_own_sums(nd):
                        osh = shape(nd.out)
                        oc = self.head_cp.get(nd.out.id, nd.out.C)
                        pl6 = self._w6_planes(self.wT[nd.conv], osh[0] * osh[1] * osh[2], oc, nd.ins[0].C)
                        if pl6 is not None:
                            self.wT6[nd.conv] = pl6
                self._flush_cut_jobs(bwd)
        for nd in order:
            o = nd.out
            shp = shape(o)
            M = shp[0] * shp[1] * shp[2]
            s = gs[o.id]
            if o.id in self.exdw_pw or o.id in self.stemdw_done:
                continue                                   # handled with its depthwise consumer (mny_exdw_bwd / mny_stemdw_bwd below)
            if o.id in self.gate_units or o.id in self.gate_absorbed:
                continue                                   # hidden units / absorbed multiply of a per-pixel gate: handled where the gate was emitted
            assert s.buf is not None, "no gradient reached %s" % o.name
            G = s.buf
            if o.id in self.gates:
                # per-pixel gate as one unit (csrc/gate.hip): three passes over (y3, dL/d out) with the two BN-backward finalizes in between; the
                # hidden tensors' gradients are never written.  The residual operand (absorbed add) takes G itself.
                gate = self.gates[o.id]
                t, se0, se3 = gate["t"], gate["se0"], gate["se3"]
                if gate["add"] is not None:
                    contribute_alias(gate["add"].ins[1], G)
                u1, u2, u3 = self.units[se0.out.id], self.units[se3.out.id], self.units[t.id]
                tv = view(t)
                C, R = t.C, se0.out.C
                wq = gate["wq"]
                gparts = _lib.query("mny_gate_bwd_parts", M)
                coef2, coef1 = torch.empty(3 * C, **f32), torch.empty(3 * R, **f32)
                gate["coef"] = (coef2, coef1)
                bwd.add("mny_gate_bwd1_bf16", tv[0], tv[1], tv[2], G, wq, u1.scale, u1.shift, u2.scale, u2.shift, u2.mean, u2.invstd, self.red_ws, M, C, R, self.stream,
                        meta=dict(flops=4 * M * C * R, bytes=eb * 2 * M * C, shape="gate bwd1 M%d C%d R%d" % (M, C, R)))
                bwd.add(fin_name, self.red_ws, gparts, M, P[se3.bn + ".weight"], u2.mean, u2.invstd, gv(se3.bn + ".weight"), gv(se3.bn + ".bias"), coef2, C, self.stream)
                dw2 = gv(se3.conv + ".weight")
                ws2 = defer_job(gparts * C * R, dw2, gparts, C * R)          # (always through the batched combine, MNY_NO_DEFER or not)
                bwd.add("mny_gate_bwd2_bf16", tv[0], tv[1], tv[2], G, wq, u1.scale, u1.shift, u1.mean, u1.invstd, u2.scale, u2.shift, coef2, self.red_ws, ws2,
                        M, C, R, self.stream, meta=dict(flops=8 * M * C * R, bytes=eb * 2 * M * C, shape="gate bwd2 M%d C%d R%d" % (M, C, R)))
                bwd.add(fin_name, self.red_ws, gparts, M, P[se0.bn + ".weight"], u1.mean, u1.invstd, gv(se0.bn + ".weight"), gv(se0.bn + ".bias"), coef1, R, self.stream)
                dw1 = gv(se0.conv + ".weight")
                ws1 = defer_job(gparts * C * R, dw1, gparts, C * R)
                rbuf = None
                prod = t.node
                if (_lib.query("mny_gate_bwd_red3_supported", C, R) == 1 and prod is not None and prod.op == "pw" and not takes_own_sums(prod)
                        and os.environ.get("MNY_GATE_NORED3") != "1"):
                    rbuf = torch.empty(gparts * 2 * C, **f32)        # dt is the project unit's complete output gradient: its BN-backward sums leave with it
                    self.fused_red[t.id] = (rbuf, gparts)
                assert gs[t.id].buf is None, "the gate's input has another consumer"
                gs[t.id].buf, gs[t.id].shared = alloc(t), False
                bwd.add("mny_gate_bwd3_bf16", tv[0], tv[1], tv[2], G, wq, u1.scale, u1.shift, u2.scale, u2.shift, coef2, coef1, u3.mean, u3.invstd, gs[t.id].buf, ws1, rbuf,
                        M, C, R, self.stream, meta=dict(flops=12 * M * C * R, bytes=eb * 3 * M * C, shape="gate bwd3 M%d C%d R%d" % (M, C, R)))
                flush_shared()
                flush_reduce()
                bwd.marks[o.name] = len(bwd.calls)
                continue
            if nd.op == "dw" and o.id in self.exdw_dw:
                # expand + depthwise unit: the depthwise unit's BN-backward sums as usual, then ONE entry point yields the depthwise and the
                # expand unit's parameter gradients and the data gradient wrt the thin input; neither the expand output nor its gradient exists
                u = self.units[o.id]
                pn = self.exdw_dw[o.id]
                pi, pu = pn.ins[0], self.units[pn.out.id]
                psh = shape(pi)
                xv = view(pi)
                gam = P[nd.bn + ".weight"]
                red_buf, red_parts = self.fused_red.get(o.id, (None, 0))
                if red_buf is None:
                    red_buf, red_parts = self.red_ws, _lib.query("mny_bn_bwd_parts", M, o.C)
                    bwd.add(K("mny_bn_bwd_reduce"), G, u.Y, u.scale, u.shift, o.act, u.mean, u.invstd, self.red_ws, M, o.C, self.stream,
                            meta=dict(flops=0, bytes=2 * eb * M * o.C, shape="M%d C%d" % (M, o.C)))
                bwd.add(fin_name, red_buf, red_parts, M, gam, u.mean, u.invstd, gv(nd.bn + ".weight"), gv(nd.bn + ".bias"),
                        self.coef_ws, o.C, self.stream)
                dparts = _lib.query("mny_exdw_bwd_parts", N, psh[1], psh[2], pi.C, o.C, 2)
                dwv = gv(nd.conv + ".weight")
                if self.defer and single(nd):
                    dwv_k, dws_k = None, defer_job(dparts * o.C * 9, dwv, dparts, o.C * 9)
                else:
                    dwv_k, dws_k = dwv, torch.empty(dparts * o.C * 9, **f32)
                xws = torch.empty(max(int(_lib.query("mny_exdw_bwd_ws_floats", N, psh[1], psh[2], pi.C, o.C, 2)), 4), **f32)
                dwe, dge, dbe = gv(pn.conv + ".weight"), gv(pn.bn + ".weight"), gv(pn.bn + ".bias")
                Mx = N * psh[1] * psh[2]
                prod = pi.node
                # the thin input is the raw output of a conv+BN unit consumed only here (the project conv in front of the first expand unit):
                # the finished dX is that unit's complete output gradient -> its BN-backward sums leave with it, no separate reduce pass
                if (os.environ.get("MNY_NO_EXRED") != "1" and pi.kind == "unit" and prod is not None and prod.op in ("pw", "dw", "stem") and gs[pi.id].buf is None
                        and n_consumers[pi.id] == 1 and not takes_own_sums(prod) and xv[1] is not None and pi.act not in (_lib.ACT_HSWISH, _lib.ACT_HSIGMOID)
                        and _lib.query("mny_exdw_bwd_red_parts", N, psh[1], psh[2], pi.C, o.C, 2) > 0):
                    ppu = self.units[pi.id]
                    rparts = _lib.query("mny_exdw_bwd_red_parts", N, psh[1], psh[2], pi.C, o.C, 2)
                    rbuf = torch.empty(rparts * 2 * pi.C, **f32)
                    self.fused_red[pi.id] = (rbuf, rparts)
                    contribute_kernel(pi, lambda out, addend, G=G, u=u, xv=xv, pu=pu, ppu=ppu, pn=pn, nd=nd, dwe=dwe, dge=dge, dbe=dbe, dwv_k=dwv_k, dws_k=dws_k,
                                      xws=xws, rbuf=rbuf, psh=psh, Kc=pi.C, C=o.C, M=M, Mx=Mx, act=o.act: bwd.add(
                        "mny_exdw_bwd_red", G, u.Y, u.scale, u.shift, act, self.coef_ws, xv[0], xv[1], xv[2], xv[3], ppu.mean, ppu.invstd, P[pn.conv + ".weight"],
                        pu.scale, pu.shift, pu.mean, pu.invstd, P[pn.bn + ".weight"], P[nd.conv + ".weight"], addend, out, dwe, dge, dbe,
                        dwv_k, dws_k, xws, rbuf, N, psh[1], psh[2], Kc, C, 2, self.stream, label="mny_exdw_bwd",
                        meta=dict(flops=6 * Mx * Kc * C + 6 * M * C * 9, bytes=self.eb * (3 * Mx * Kc + 4 * M * C), shape="exdw K%d C%d H%d +red" % (Kc, C, psh[1]))), inplace_ok=False)
                    flush_shared()
                    flush_reduce()
                    bwd.marks[o.name] = len(bwd.calls)
                    continue
                contribute_kernel(pi, lambda out, addend, G=G, u=u, xv=xv, pu=pu, pn=pn, nd=nd, dwe=dwe, dge=dge, dbe=dbe, dwv_k=dwv_k, dws_k=dws_k, xws=xws,
                                  psh=psh, Kc=pi.C, C=o.C, M=M, Mx=Mx, act=o.act: bwd.add(
                    "mny_exdw_bwd", G, u.Y, u.scale, u.shift, act, self.coef_ws, xv[0], xv[1], xv[2], xv[3], P[pn.conv + ".weight"],
                    pu.scale, pu.shift, pu.mean, pu.invstd, P[pn.bn + ".weight"], P[nd.conv + ".weight"], addend, out, dwe, dge, dbe,
                    dwv_k, dws_k, xws, N, psh[1], psh[2], Kc, C, 2, self.stream,
                    meta=dict(flops=6 * Mx * Kc * C + 6 * M * C * 9, bytes=self.eb * (3 * Mx * Kc + 4 * M * C), shape="exdw K%d C%d H%d" % (Kc, C, psh[1]))), inplace_ok=False)    # algorithmic: the expand output once per pass + P1 + dX; the transposed stencil per pass + dW_dw; X twice + dX once, G_z and Z twice
                flush_shared()
                flush_reduce()
                bwd.marks[o.name] = len(bwd.calls)
                continue
            if nd.op == "add":
                has_b, has_up = nd.k & 1, nd.k & 2
                contribute_alias(nd.ins[0], G)
                if has_b:
                    contribute_alias(nd.ins[1], G)
                if has_up:
                    upv = nd.ins[-1]
                    us = gs[upv.id]
                    if us.buf is None:
                        us.buf, us.shared = alloc(upv), False
                        bwd.add(K("mny_upsample_bwd"), G, us.buf, 0, shp[0], shp[1], shp[2], shp[3], self.stream)
                    else:
                        if us.shared:
                            nb = alloc(upv)
                            bwd.add(K("mny_axpy"), us.buf, None, nb, 0, nb.numel(), self.stream)
                            us.buf, us.shared = nb, False
                        bwd.add(K("mny_upsample_bwd"), G, us.buf, 1, shp[0], shp[1], shp[2], shp[3], self.stream)
                bwd.marks[o.name] = len(bwd.calls)
                continue
            if nd.op == "mul":
                va, vb = view(nd.ins[0]), view(nd.ins[1])
                contribute_kernel(nd.ins[0], lambda out, addend, G=G, vb=vb, M=M, C=o.C: bwd.add(
                    K("mny_mul_views_bwd"), G, vb[0], vb[1], vb[2], vb[3], addend, out, M, C, self.stream))
                contribute_kernel(nd.ins[1], lambda out, addend, G=G, va=va, M=M, C=o.C: bwd.add(
                    K("mny_mul_views_bwd"), G, va[0], va[1], va[2], va[3], addend, out, M, C, self.stream))
                bwd.marks[o.name] = len(bwd.calls)
                continue
            if nd.op == "partadd":
                av, upv = nd.ins[0], nd.ins[1]
                for tgt, emit in ((av, lambda dst, acc, G=G, M=M, Ca=av.C, Cb=o.C: bwd.add(
                                        K("mny_slice_channels"), G, dst, acc, M, Ca, Cb, self.stream)),
                                  (upv, lambda dst, acc, G=G, shp=shp: bwd.add(
                                        K("mny_upsample_bwd"), G, dst, acc, shp[0], shp[1], shp[2], shp[3], self.stream))):
                    ts = gs[tgt.id]
                    if ts.buf is None:
                        ts.buf, ts.shared = alloc(tgt), False
                        emit(ts.buf, 0)
                    else:
                        if ts.shared:
                            nb = alloc(tgt)
                            bwd.add(K("mny_axpy"), ts.buf, None, nb, 0, nb.numel(), self.stream)
                            ts.buf, ts.shared = nb, False
                        emit(ts.buf, 1)
                bwd.marks[o.name] = len(bwd.calls)
                continue
            if nd.op == "pw" and takes_own_sums(nd):
                # thin "expand" unit: BN-backward + wgrad + dgrad from (G, Y, X) in 4 passes, dY never materialised
                u = self.units[o.id]
                i = nd.ins[0]
                xv = view(i)
                dwv, dgv, dbv = gv(nd.conv + ".weight"), gv(nd.bn + ".weight"), gv(nd.bn + ".bias")
                gam = P[nd.bn + ".weight"]
                w = P[nd.conv + ".weight"]
                # the data gradient of this unit completes the output gradient of the unit in front (the previous block's project conv, directly or
                # through the residual add): that unit's BN-backward sums leave with stage 2 (round 6: mny_pw_bnbwd_red, fp32 storage)
                tgt = red_target(i, nd) if (not self.bf16 and os.environ.get("MNY_NO_REDFUSE") != "1" and os.environ.get("MNY_NO_BNW_RED") != "1") else None
                if tgt is not None and (tgt.act in (_lib.ACT_HSWISH, _lib.ACT_HSIGMOID) or _lib.query("mny_pw_bnbwd_red_supported", M, i.C, o.C) != 1):
                    tgt = None
                if tgt is not None:
                    pu = self.units[tgt.id]
                    rparts = _lib.query("mny_pw_bnbwd_red_parts", M, i.C, o.C)
                    rbuf = torch.empty(rparts * 2 * i.C, **f32)
                    self.fused_red[tgt.id] = (rbuf, rparts)
                    contribute_kernel(i, lambda out, addend, G=G, u=u, xv=xv, w=w, gam=gam, dwv=dwv, dgv=dgv, dbv=dbv, M=M, K=i.C, Nc=o.C, act=o.act, pu=pu, rbuf=rbuf, ract=tgt.act:
                                      bwd.add("mny_pw_bnbwd_red", G, u.Y, u.scale, u.shift, act, u.mean, u.invstd, gam, xv[0], xv[1], xv[2], xv[3],
                                              w, addend, out, dwv, dgv, dbv, self.ws, pu.Y, pu.scale, pu.shift, ract, pu.mean, pu.invstd, rbuf, M, K, Nc, self.stream,
                                              label="mny_pw_bnbwd",
                                              meta=dict(flops=6 * M * K * Nc, bytes=self.eb * (3 * M * Nc + 3 * M * K), shape="M%d K%d N%d +red" % (M, K, Nc))))
                    flush_shared()
                    bwd.marks[o.name] = len(bwd.calls)
                    continue
                contribute_kernel(i, lambda out, addend, G=G, u=u, xv=xv, w=w, gam=gam, dwv=dwv, dgv=dgv, dbv=dbv, M=M, K=i.C, Nc=o.C, act=o.act:
                                  bwd.add(self.K("mny_pw_bnbwd"), G, u.Y, u.scale, u.shift, act, u.mean, u.invstd, gam, xv[0], xv[1], xv[2], xv[3],
                                          w, addend, out, dwv, dgv, dbv, self.ws, M, K, Nc, self.stream,
                                          meta=dict(flops=6 * M * K * Nc, bytes=self.eb * (3 * M * Nc + 3 * M * K), shape="M%d K%d N%d" % (M, K, Nc))))      # algorithmic: G, Y (stage 1) + G (stage 2); X twice, dX once
                flush_shared()
                bwd.marks[o.name] = len(bwd.calls)
                continue
            if nd.op == "pw" and o.id in self.lr_units:
                # wide expand unit, low-rank BN backward: G holds dzc = ca o dL/da o act'(z) (written by the depthwise unit behind, whose epilogue also
                # left this unit's BN-backward sums).  dW = dzc^T X + cb o (W X^T X) + cc (x) colsum(X);  dX = dzc W + X Q + r.
                u = self.units[o.id]
                i = nd.ins[0]
                xv = view(i)
                Kc, C = i.C, o.C
                w = P[nd.conv + ".weight"]
                red_buf, red_parts = self.fused_red[o.id]
                coef_u = torch.empty(3 * C, **f32)                 # private: the side-stream weight-gradient correction reads it long after coef_ws is reused
                bwd.add(fin_name, red_buf, red_parts, M, P[nd.bn + ".weight"], u.mean, u.invstd, gv(nd.bn + ".weight"), gv(nd.bn + ".bias"), coef_u, C, self.stream)
                bq, rb = torch.empty(Kc * Kc, **f32), torch.empty(Kc, **f32)
                prep_side = self.side_on and os.environ.get("MNY_LR_PREP_MAIN") != "1"       # Q, r under the main-term GEMM below; joined in front of the correction
                if prep_side:
                    bwd.add_py(self._fork_side2, "fork")
                bwd.add("mny_lr_prep", coef_u, w, bq, rb, C, Kc, self.stream_side2 if prep_side else self.stream)
                dwv = gv(nd.conv + ".weight")
                if self.side_on:
                    bwd.add_py(self._fork_side, "fork")
                st_w = self.stream_side if self.side_on else self.stream
                psplits = _lib.query("mny_pw_wgrad_splits", M, Kc, C)
                pws = defer_job(max(_lib.query("mny_pw_wgrad_ws_floats", M, Kc, C), psplits * C * Kc), dwv, psplits, C * Kc)
                bwd.add("mny_pw_wgrad", xv[0], xv[1], xv[2], xv[3], G, None, None, pws, M, Kc, C, st_w,
                        meta=dict(flops=2 * M * Kc * C, bytes=eb * (M * Kc + M * C) + 4 * Kc * C, shape="M%d K%d N%d" % (M, Kc, C)))
                gparts = _lib.query("mny_lr_gram_parts", M, Kc)
                gsum = torch.empty(Kc * Kc + Kc, **f32)
                gws = defer_job(gparts * (Kc * Kc + Kc), gsum, gparts, Kc * Kc + Kc)
                bwd.add("mny_lr_gram", xv[0], xv[1], xv[2], xv[3], gws, M, Kc, st_w,
                        meta=dict(flops=2 * M * Kc * Kc, bytes=eb * M * Kc, shape="gram M%d K%d" % (M, Kc)))
                self._post_reduce.append(lambda st, dwv=dwv, gsum=gsum, coef_u=coef_u, w=w, C=C, Kc=Kc: bwd.add("mny_lr_wfix", dwv, gsum, coef_u, w, C, Kc, st))
                wT = self.wT[nd.conv]
                wT6 = getattr(self, "wT6", {}).get(nd.conv)
                contribute_kernel(i, lambda out, addend, G=G, wT=wT, wT6=wT6, M=M, K=C, Nc=Kc: bwd.add(
                    "mny_pw_fwd_w6" if wT6 is not None else "mny_pw_fwd", G, None, None, ACT_NONE, wT6 if wT6 is not None else wT, None, addend, out, None,
                    M, K, Nc, self.stream, label="mny_pw_fwd",
                    meta=dict(flops=2 * M * K * Nc, bytes=self.eb * (M * K + M * Nc) + 4 * K * Nc, shape="dgrad M%d K%d N%d" % (M, K, Nc))))
                buf = gs[i.id].buf
                if prep_side:
                    bwd.add_py(self._join_side2, "join")
                tgt = red_target(i, nd) if os.environ.get("MNY_NO_REDFUSE") != "1" else None
                if tgt is not None:
                    pu = self.units[tgt.id]
                    rparts = _lib.query("mny_pw_lr_fix_parts", M, Kc, tgt.act)
                    rbuf = torch.empty(rparts * 2 * Kc, **f32)
                    self.fused_red[tgt.id] = (rbuf, rparts)
                    bwd.add("mny_pw_lr_fix", xv[0], xv[1], xv[2], bq, rb, buf, buf, pu.Y, pu.scale, pu.shift, tgt.act, pu.mean, pu.invstd, rbuf, M, Kc, self.stream,
                            meta=dict(flops=2 * M * Kc * Kc, bytes=eb * 4 * M * Kc, shape="lr fix+red M%d K%d" % (M, Kc)))
                else:
                    bwd.add("mny_pw_lr_fix", xv[0], xv[1], xv[2], bq, rb, buf, buf, None, None, None, 0, None, None, None, M, Kc, self.stream,
                            meta=dict(flops=2 * M * Kc * Kc, bytes=eb * 3 * M * Kc, shape="lr fix M%d K%d" % (M, Kc)))
                flush_shared()
                flush_reduce()
                bwd.marks[o.name] = len(bwd.calls)
                continue
            if nd.op == "pwb":
                dY = G
            else:
                u = self.units[o.id]
                parts = _lib.query("mny_bn_bwd_parts", M, o.C)
                gam = P[nd.bn + ".weight"]
                if (nd.op == "dw" and os.environ.get("MNY_NO_DWFUSE") != "1" and _lib.query("mny_dw_bnbwd_supported", nd.k, nd.stride) == 1
                        and o.act != _lib.ACT_HSIGMOID and nd.ins[0].act != _lib.ACT_HSIGMOID):
                    # 3x3 (register form) / 5x5 (tile form, csrc/dwtile.hip) stride-1 depthwise unit: dY is rebuilt on chip, one pass over
                    # (G, Y, X) yields dX and dW
                    kk = nd.k * nd.k
                    i = nd.ins[0]
                    ish = shape(i)
                    xv = view(i)
                    red_buf, red_parts = self.fused_red.get(o.id, (None, 0))
                    if red_buf is None:
                        red_buf, red_parts = self.red_ws, parts
                        bwd.add(K("mny_bn_bwd_reduce"), G, u.Y, u.scale, u.shift, o.act, u.mean, u.invstd, self.red_ws, M, o.C, self.stream,
                                meta=dict(flops=0, bytes=2 * eb * M * o.C, shape="M%d C%d" % (M, o.C)))
                    bwd.add(fin_name, red_buf, red_parts, M, gam, u.mean, u.invstd, gv(nd.bn + ".weight"), gv(nd.bn + ".bias"),
                            self.coef_ws, o.C, self.stream)
                    n_sh = len(self.shared_tmp)
                    dwv = gv(nd.conv + ".weight")
                    wt = P[nd.conv + ".weight"]
                    dwv_k, ws_k = dwv, self.ws
                    prod = i.node
                    # the input is the raw output of a conv+BN+act unit consumed ONLY here: this kernel's dX is that unit's complete
                    # output gradient, so it also leaves the unit's BN-backward sums (mny_dw_bnbwd_red) and the unit's separate
                    # bn_bwd_reduce pass — a re-read of dX and X — disappears (wide expand units, the stem, the neck's pointwise units)
                    with_red = (os.environ.get("MNY_NO_DWRED") != "1" and i.kind == "unit" and prod is not None and prod.op in ("pw", "stem")
                                and gs[i.id].buf is None and n_consumers[i.id] == 1 and not takes_own_sums(prod) and xv[1] is not None
                                and i.act not in (_lib.ACT_HSIGMOID,))
                    pflags = (1 if self.bf16 else 0) | (2 if with_red else 0)        # (the row count depends on the form that runs: csrc/dwtile.hip dwt_use)
                    if self.defer and single(nd):
                        dparts = _lib.query("mny_dw_bnbwd_parts_k", N, ish[1], ish[2], o.C, nd.k, pflags)
                        dwv_k, ws_k = None, defer_job(dparts * o.C * kk, dwv, dparts, o.C * kk)
                    # stem -> this depthwise unit (MobileNetV2's first two units): ONE pass yields both units' parameter gradients; the stem's
                    # output gradient (its only consumer is the stem's weight gradient) is never written (csrc/stemdw.hip)
                    if (nd.k == 3 and not self.bf16 and not self.frozen and i.kind == "unit" and prod is not None and prod.op == "stem" and gs[i.id].buf is None
                            and n_consumers[i.id] == 1 and not takes_own_sums(prod) and xv[1] is not None and single(nd) and single(prod)
                            and _lib.query("mny_stemdw_supported", N, self.H, self.W, i.C, i.act, o.act) == 1):
                        pu = self.units[i.id]
                        sparts = _lib.query("mny_stemdw_bwd_parts", N, self.H, self.W, i.C)
                        xws = torch.empty(max(int(_lib.query("mny_stemdw_bwd_ws_floats", N, self.H, self.W, i.C)), 4), **f32)
                        if self.defer:
                            self._red_jobs.pop()            # the job registered above counts the rows of mny_dw_bnbwd's partial buffer
                            dwv_k, ws_k = None, defer_job(sparts * o.C * 9, dwv, sparts, o.C * 9)
                        else:
                            dwv_k, ws_k = dwv, torch.empty(sparts * o.C * 9, **f32)
                        bwd.add("mny_stemdw_bwd", G, u.Y, u.scale, u.shift, o.act, self.coef_ws, pu.Y, pu.scale, pu.shift, pu.mean, pu.invstd,
                                P[prod.bn + ".weight"], i.act, self.x_ptr, P[prod.conv + ".weight"], wt,
                                gv(prod.conv + ".weight"), gv(prod.bn + ".weight"), gv(prod.bn + ".bias"), dwv_k, ws_k, xws,
                                N, self.H, self.W, i.C, self.stream,
                                meta=dict(flops=2 * M * o.C * (2 * 9 + 27), bytes=4 * (3 * M * o.C + N * 3 * self.H * self.W), shape="stem+dw C%d H%d" % (o.C, ish[1])))
                        self.stemdw_done.add(i.id)
                        flush_shared()
                        flush_reduce()
                        bwd.marks[o.name] = len(bwd.calls)
                        continue
                    if with_red:
                        pu = self.units[i.id]
                        rparts = _lib.query("mny_dw_bnbwd_parts_k", N, ish[1], ish[2], o.C, nd.k, pflags)
                        rbuf = torch.empty(rparts * 2 * i.C, **f32)
                        self.fused_red[i.id] = (rbuf, rparts)
                        # the producer is a wide expand unit on the low-rank BN backward: it takes its gradient as ca o dX o act'(z) (mny_dw_bnbwd_red_dz)
                        lr = prod.op == "pw" and lr_ok(prod) and _lib.query("mny_dw_bnbwd_red_dz_supported", nd.k, o.C, int(self.bf16)) == 1
                        if lr:
                            self.lr_units.add(i.id)
                        red_name = "mny_dw_bnbwd_red_dz" if lr else self.K("mny_dw_bnbwd_red")
                        contribute_kernel(i, lambda out, addend, G=G, u=u, xv=xv, wt=wt, dwv=dwv_k, wsl=ws_k, ish=ish, C=o.C, act=o.act, M=M, pu=pu, rbuf=rbuf, k=nd.k, red_name=red_name, lr=lr: bwd.add(
                            red_name, G, u.Y, u.scale, u.shift, act, self.coef_ws, xv[0], xv[1], xv[2], xv[3], pu.mean, pu.invstd, wt, addend,
                            out, dwv, wsl, rbuf, N, ish[1], ish[2], C, k, 1, self.stream, label=self.K("mny_dw_bnbwd_red"),
                            meta=dict(flops=4 * M * C * k * k, bytes=self.eb * 4 * M * C, shape="C%d H%d%s +red%s" % (C, ish[1], " k5" if k == 5 else "", " dz" if lr else ""))))
                    else:
                        contribute_kernel(i, lambda out, addend, G=G, u=u, xv=xv, wt=wt, dwv=dwv_k, wsl=ws_k, ish=ish, C=o.C, act=o.act, M=M, k=nd.k: bwd.add(
                            self.K("mny_dw_bnbwd"), G, u.Y, u.scale, u.shift, act, self.coef_ws, xv[0], xv[1], xv[2], xv[3], wt, addend, out, dwv, wsl,
                            N, ish[1], ish[2], C, k, 1, self.stream,
                            meta=dict(flops=4 * M * C * k * k, bytes=self.eb * 4 * M * C, shape="C%d H%d%s" % (C, ish[1], " k5" if k == 5 else ""))))
                    flush_shared()
                    flush_reduce()
                    bwd.marks[o.name] = len(bwd.calls)
                    continue
                if (nd.op == "dw" and nd.k == 3 and nd.stride == 2 and os.environ.get("MNY_NO_DWFUSE2") != "1"
                        and o.act != _lib.ACT_HSIGMOID and nd.ins[0].act != _lib.ACT_HSIGMOID):
                    # 3x3 stride-2 depthwise unit: the same fusion (mny_dw_bnbwd_s2): no dY tensor, one launch instead of three
                    i = nd.ins[0]
                    ish = shape(i)
                    xv = view(i)
                    red_buf, red_parts = self.fused_red.get(o.id, (None, 0))
                    if red_buf is None:
                        red_buf, red_parts = self.red_ws, parts
                        bwd.add(K("mny_bn_bwd_reduce"), G, u.Y, u.scale, u.shift, o.act, u.mean, u.invstd, self.red_ws, M, o.C, self.stream,
                                meta=dict(flops=0, bytes=2 * eb * M * o.C, shape="M%d C%d" % (M, o.C)))
                    bwd.add(fin_name, red_buf, red_parts, M, gam, u.mean, u.invstd, gv(nd.bn + ".weight"), gv(nd.bn + ".bias"),
                            self.coef_ws, o.C, self.stream)
                    dwv = gv(nd.conv + ".weight")
                    wt = P[nd.conv + ".weight"]
                    dwv_k, ws_k = dwv, self.ws
                    if self.defer and single(nd):
                        dparts = _lib.query("mny_dw_bnbwd_s2_parts", N, ish[1], ish[2], o.C)
                        dwv_k, ws_k = None, defer_job(dparts * o.C * 9, dwv, dparts, o.C * 9)
                    prod = i.node
                    # behind a wide expand unit on the low-rank BN backward (round 6): the producer's sums and  ca o dX o relu6'(z)  leave with this pass
                    if (not self.bf16 and i.kind == "unit" and prod is not None and prod.op == "pw" and gs[i.id].buf is None and n_consumers[i.id] == 1
                            and xv[1] is not None and i.act == _lib.ACT_RELU6 and lr_ok(prod) and os.environ.get("MNY_NO_LR_S2") != "1"):
                        pu = self.units[i.id]
                        rparts = _lib.query("mny_dw_bnbwd_s2_parts", N, ish[1], ish[2], o.C)
                        rbuf = torch.empty(rparts * 2 * i.C, **f32)
                        self.fused_red[i.id] = (rbuf, rparts)
                        self.lr_units.add(i.id)
                        contribute_kernel(i, lambda out, addend, G=G, u=u, xv=xv, wt=wt, dwv=dwv_k, wsl=ws_k, ish=ish, C=o.C, act=o.act, M=M, pu=pu, rbuf=rbuf: bwd.add(
                            "mny_dw_bnbwd_s2_red_dz", G, u.Y, u.scale, u.shift, act, self.coef_ws, xv[0], xv[1], xv[2], xv[3], pu.mean, pu.invstd, wt, addend, out, dwv, wsl, rbuf,
                            N, ish[1], ish[2], C, self.stream, label=self.K("mny_dw_bnbwd_s2"),
                            meta=dict(flops=4 * M * C * 9, bytes=self.eb * (2 * M * C + 2 * N * ish[1] * ish[2] * C), shape="C%d H%d s2 +red dz" % (C, ish[1]))))
                        flush_shared()
                        flush_reduce()
                        bwd.marks[o.name] = len(bwd.calls)
                        continue
                    contribute_kernel(i, lambda out, addend, G=G, u=u, xv=xv, wt=wt, dwv=dwv_k, wsl=ws_k, ish=ish, C=o.C, act=o.act, M=M: bwd.add(
                        self.K("mny_dw_bnbwd_s2"), G, u.Y, u.scale, u.shift, act, self.coef_ws, xv[0], xv[1], xv[2], xv[3], wt, addend, out, dwv, wsl,
                        N, ish[1], ish[2], C, self.stream,
                        meta=dict(flops=4 * M * C * 9, bytes=self.eb * (2 * M * C + 2 * N * ish[1] * ish[2] * C), shape="C%d H%d s2" % (C, ish[1]))))
                    flush_shared()
                    flush_reduce()
                    bwd.marks[o.name] = len(bwd.calls)
                    continue
                if (nd.op == "dw" and nd.k == 5 and nd.stride == 2 and os.environ.get("MNY_NO_DWFUSE5S2") != "1" and o.C % 2 == 0
                        and o.act != _lib.ACT_HSIGMOID and nd.ins[0].act != _lib.ACT_HSIGMOID):
                    # 5x5 stride-2 depthwise unit (MobileNetV3's two down-sampling 5x5 blocks): mny_dw_bnbwd_s2k5 — one launch instead of
                    # bn_bwd_apply + dw5_wgrad + dw_bwd_data_s2k5, and the producer's BN-backward sums with it where the unit is its only consumer
                    i = nd.ins[0]
                    ish = shape(i)
                    xv = view(i)
                    red_buf, red_parts = self.fused_red.get(o.id, (None, 0))
                    if red_buf is None:
                        red_buf, red_parts = self.red_ws, parts
                        bwd.add(K("mny_bn_bwd_reduce"), G, u.Y, u.scale, u.shift, o.act, u.mean, u.invstd, self.red_ws, M, o.C, self.stream,
                                meta=dict(flops=0, bytes=2 * eb * M * o.C, shape="M%d C%d" % (M, o.C)))
                    bwd.add(fin_name, red_buf, red_parts, M, gam, u.mean, u.invstd, gv(nd.bn + ".weight"), gv(nd.bn + ".bias"),
                            self.coef_ws, o.C, self.stream)
                    dwv = gv(nd.conv + ".weight")
                    wt = P[nd.conv + ".weight"]
                    dparts = _lib.query("mny_dw_bnbwd_s2k5_parts", N, ish[1], ish[2], o.C)
                    dwv_k, ws_k = dwv, self.ws
                    if self.defer and single(nd):
                        dwv_k, ws_k = None, defer_job(dparts * o.C * 25, dwv, dparts, o.C * 25)
                    prod = i.node
                    with_red = (os.environ.get("MNY_NO_DWRED") != "1" and i.kind == "unit" and prod is not None and prod.op in ("pw", "stem")
                                and gs[i.id].buf is None and n_consumers[i.id] == 1 and not takes_own_sums(prod) and xv[1] is not None
                                and i.act not in (_lib.ACT_HSIGMOID,))
                    pu, rbuf = None, None
                    if with_red:
                        pu = self.units[i.id]
                        rbuf = torch.empty(dparts * 2 * i.C, **f32)
                        self.fused_red[i.id] = (rbuf, dparts)
                    contribute_kernel(i, lambda out, addend, G=G, u=u, xv=xv, wt=wt, dwv=dwv_k, wsl=ws_k, ish=ish, C=o.C, act=o.act, M=M, pu=pu, rbuf=rbuf: bwd.add(
                        self.K("mny_dw_bnbwd_s2k5"), G, u.Y, u.scale, u.shift, act, self.coef_ws, xv[0], xv[1], xv[2], xv[3],
                        pu.mean if pu is not None else None, pu.invstd if pu is not None else None, wt, addend, out, dwv, wsl, rbuf,
                        N, ish[1], ish[2], C, self.stream,
                        meta=dict(flops=4 * M * C * 25, bytes=self.eb * (2 * M * C + 2 * N * ish[1] * ish[2] * C), shape="C%d H%d s2 k5%s" % (C, ish[1], " +red" if rbuf is not None else ""))))
                    flush_shared()
                    flush_reduce()
                    bwd.marks[o.name] = len(bwd.calls)
                    continue
                dY = G if not s.shared else alloc(o)
                red_buf, red_parts = self.fused_red.get(o.id, (None, 0))
                if red_buf is None:
                    red_buf, red_parts = self.red_ws, parts
                    bwd.add(K("mny_bn_bwd_reduce"), G, u.Y, u.scale, u.shift, o.act, u.mean, u.invstd, self.red_ws, M, o.C, self.stream,
                            meta=dict(flops=0, bytes=2 * eb * M * o.C, shape="M%d C%d" % (M, o.C)))
                bwd.add(fin_name, red_buf, red_parts, M, gam, u.mean, u.invstd, gv(nd.bn + ".weight"), gv(nd.bn + ".bias"),
                        self.coef_ws, o.C, self.stream)
                # linear project unit behind a conv+BN+act unit it alone consumes (the thin bottleneck that closes an inverted-residual block):
                # ONE pass rebuilds dY in LDS, reads the wide input once and yields the data gradient, the input unit's BN-backward sums and the
                # weight gradient (csrc/pjbwd.hip) — instead of bn_bwd_apply + pw_dgrad_bnred + pw_wgrad
                if nd.op == "pw" and not self.frozen and o.act == ACT_NONE and not nd.bias and single(nd) and o.id not in self.head_cp:      # (bf16 storage: mny_pj_bwd_bf16, csrc/gate.hip)
                    # (G is only read here: a gradient buffer shared with the residual path is fine)
                    i = nd.ins[0]
                    prod = i.node
                    if (prod is not None and prod.op in ("dw", "pw") and i.kind == "unit" and gs[i.id].buf is None and n_consumers[i.id] == 1
                            and not takes_own_sums(prod) and view(i)[1] is not None and single(prod)
                            and _lib.query(K("mny_pj_bwd_supported"), M, i.C, o.C, i.act) == 1):
                        pu = self.units[i.id]
                        rparts = _lib.query(K("mny_pj_bwd_parts"), M, i.C, o.C)
                        rbuf = torch.empty(rparts * 2 * i.C, **f32)
                        self.fused_red[i.id] = (rbuf, rparts)
                        dwv = gv(nd.conv + ".weight")
                        if self.defer:
                            dwv_k, ws_k = None, defer_job(rparts * o.C * i.C, dwv, rparts, o.C * i.C)
                        else:
                            dwv_k, ws_k = dwv, torch.empty(rparts * o.C * i.C, **f32)
                        contribute_kernel(i, lambda out, addend, G=G, u=u, pu=pu, w=P[nd.conv + ".weight"], dwv=dwv_k, wsl=ws_k, rbuf=rbuf, M=M, Ki=i.C, No=o.C, act_=i.act: bwd.add(
                            self.K("mny_pj_bwd"), G, u.Y, self.coef_ws, pu.Y, pu.scale, pu.shift, pu.mean, pu.invstd, act_, w, out, dwv, wsl, rbuf, M, Ki, No, self.stream,
                            meta=dict(flops=4 * M * Ki * No, bytes=self.eb * (2 * M * No + 2 * M * Ki), shape="project M%d K%d N%d" % (M, Ki, No))))
                        flush_shared()
                        flush_reduce()
                        bwd.marks[o.name] = len(bwd.calls)
                        continue
                stem_fused = (nd.op == "stem" and os.environ.get("MNY_NO_STEMFUSE") != "1" and _lib.query("mny_stem_bnwgrad_supported", o.C) == 1)
                if not stem_fused:
                    bwd.add(K("mny_bn_bwd_apply"), G, u.Y, u.scale, u.shift, o.act, self.coef_ws, dY, M, o.C, self.stream,
                            meta=dict(flops=0, bytes=3 * eb * M * o.C, shape="M%d C%d" % (M, o.C)))
            w = P[nd.conv + ".weight"]
            if nd.op == "stem" and stem_fused:
                # the stem has no data gradient: its only consumer of dY is the weight gradient, which rebuilds dY from (G, Y) on load
                sdw, sws = gv(nd.conv + ".weight"), self.ws
                if self.defer and single(nd):
                    sparts = _lib.query("mny_stem_wgrad_parts", N, self.H, self.W, o.C)
                    sdw, sws = None, defer_job(sparts * o.C * 27, sdw, sparts, o.C * 27)
                bwd.add(K("mny_stem_bnwgrad"), self.x_ptr, G, u.Y, u.scale, u.shift, o.act, self.coef_ws, sdw, sws,
                        N, self.H, self.W, o.C, self.stream)
            elif nd.op == "stem":
                if self.side_on:
                    bwd.add_py(self._fork_side, "fork")
                bwd.add(K("mny_stem_wgrad"), self.x_ptr, dY, gv(nd.conv + ".weight"), self.ws_side, N, self.H, self.W, o.C, self.stream_side)
            elif nd.op == "dw":
                i = nd.ins[0]
                ish = shape(i)
                xv = view(i)
                dwb = eb * (N * ish[1] * ish[2] * o.C + M * o.C)
                n_sh = len(self.shared_tmp)
                dwv_ = gv(nd.conv + ".weight")
                on_side = self.side_on and single(nd)          # (both contributions of a module applied twice stay in order on the main stream)
                if on_side:
                    bwd.add_py(self._fork_side, "fork")
                dws_ = self.ws_side if on_side else self.ws
                if self.defer and single(nd):
                    wparts = _lib.query("mny_dw_wgrad_parts", N, ish[1], ish[2], o.C, nd.k, nd.stride)
                    dws_, dwv_ = defer_job(wparts * o.C * nd.k * nd.k, dwv_, wparts, o.C * nd.k * nd.k), None
                bwd.add(K("mny_dw_bwd_weight"), xv[0], xv[1], xv[2], xv[3], dY, dwv_, dws_, N, ish[1], ish[2], o.C,
                        nd.k, nd.stride, self.stream_side if on_side else self.stream, meta=dict(flops=2 * M * o.C * nd.k * nd.k, bytes=dwb, shape="C%d H%d s%d" % (o.C, ish[1], nd.stride)))
                contribute_kernel(i, lambda out, addend, dY=dY, w=w, ish=ish, nd=nd, C=o.C, M=M, dwb=dwb: bwd.add(
                    K("mny_dw_bwd_data"), dY, w, addend, out, N, ish[1], ish[2], C, nd.k, nd.stride, self.stream,
                    meta=dict(flops=2 * M * C * nd.k * nd.k, bytes=dwb, shape="C%d H%d s%d" % (C, ish[1], nd.stride))))
            else:   # pw / pwb
                i = nd.ins[0]
                xv = view(i)
                n_sh = len(self.shared_tmp)
                db = gv(nd.conv + ".bias") if nd.bias else None
                dwv_ = gv(nd.conv + ".weight")
                on_side = self.side_on and single(nd)          # (both contributions of a module applied twice stay in order on the main stream)
                oc = self.head_cp.get(o.id, o.C)        # channel count of dY as the GEMMs see it (padded for the heads)
                if on_side:
                    bwd.add_py(self._fork_side, "fork")
                pws_ = self.ws_side if on_side else self.ws
                if self.defer and db is None and single(nd):
                    psplits = _lib.query(K("mny_pw_wgrad_splits"), M, i.C, oc)
                    pws_, dwv_ = defer_job(max(_lib.query("mny_pw_wgrad_ws_floats", M, i.C, oc), psplits * oc * i.C), dwv_, psplits, oc * i.C), None
                bwd.add(K("mny_pw_wgrad"), xv[0], xv[1], xv[2], xv[3], dY, dwv_, db, pws_, M, i.C, oc,
                        self.stream_side if on_side else self.stream,
                        meta=dict(flops=2 * M * i.C * o.C, bytes=eb * (M * i.C + M * o.C) + 4 * i.C * o.C, shape="M%d K%d N%d" % (M, i.C, oc)))
                if self.t_batch:
                    wT = self.wT[nd.conv]               # filled by the batched transpose at the head of the list
                else:
                    wT = torch.empty(i.C, oc, **act)    # the data-gradient GEMM reads W^T in the activation storage type
                    self.wT[nd.conv] = wT
                    if oc != o.C:
                        bwd.add(K("mny_transpose_pad"), w, wT, o.C, i.C, oc, self.stream)
                    else:
                        bwd.add(K("mny_transpose"), w, wT, o.C, i.C, self.stream)
                prod = i.node
                wT6 = getattr(self, "wT6", {}).get(nd.conv)      # pre-cut W^T planes: this data gradient takes the six-product bf16 form
                # (bf16 storage: round 2 measured the epilogue's 2-byte loads of the unit's output at what the saved pass cost, 3 281 vs
                # 3 293 img/s on MobileNetV3 512; with round 3's kernels it wins — same-box A/B 17.65-17.70 vs 17.93-17.99 ms/step — and is on;
                # MNY_REDFUSE_BF16=0 turns it off for A/B)
                # the unit whose complete output gradient this data gradient is (i itself, or — through a residual add — the add's unit operand)
                tgt = red_target(i, nd) if (os.environ.get("MNY_NO_REDFUSE") != "1" and (not self.bf16 or os.environ.get("MNY_REDFUSE_BF16") != "0")) else None
                if (tgt is not None and gs[i.id].buf is None
                        and _lib.query(K("mny_pw_dgrad_bnred_supported"), M, oc, i.C, tgt.act) == 1):
                    # this data gradient IS the complete dL/d(output) of a conv+BN+act unit whose backward starts with a BN reduction:
                    # the sums are taken from the GEMM's own output tile (+ the unit's raw output), the separate reduce pass is dropped
                    pu = self.units[tgt.id]
                    rparts = _lib.query(K("mny_pw_dgrad_bnred_parts"), M, oc, i.C)
                    rbuf = torch.empty(rparts * 2 * i.C, **f32)
                    self.fused_red[tgt.id] = (rbuf, rparts)
                    if wT6 is not None:
                        contribute_kernel(i, lambda out, addend, dY=dY, wT6=wT6, M=M, K=oc, Nc=i.C, pu=pu, rbuf=rbuf, act_=tgt.act: bwd.add(
                            "mny_pw_dgrad_bnred_w6", dY, wT6, None, out, pu.Y, pu.scale, pu.shift, act_, pu.mean, pu.invstd, rbuf, M, K, Nc, self.stream,
                            label="mny_pw_dgrad_bnred",
                            meta=dict(flops=2 * M * K * Nc, bytes=self.eb * (M * K + 2 * M * Nc) + 4 * K * Nc, shape="dgrad+red M%d K%d N%d" % (M, K, Nc))))
                    else:
                        contribute_kernel(i, lambda out, addend, dY=dY, wT=wT, M=M, K=oc, Nc=i.C, pu=pu, rbuf=rbuf, act_=tgt.act: bwd.add(
                            self.K("mny_pw_dgrad_bnred"), dY, wT, out, pu.Y, pu.scale, pu.shift, act_, pu.mean, pu.invstd, rbuf, M, K, Nc, self.stream,
                            meta=dict(flops=2 * M * K * Nc, bytes=self.eb * (M * K + 2 * M * Nc) + 4 * K * Nc, shape="dgrad+red M%d K%d N%d" % (M, K, Nc))))
                elif (tgt is not None and os.environ.get("MNY_NO_REDADD") != "1" and gs[i.id].buf is not None
                        and _lib.query(K("mny_pw_dgrad_bnred_add_supported"), M, oc, i.C, tgt.act) == 1):
                    # the LAST contribution to the output gradient of a conv+BN+act unit (a project conv feeding a residual add and the
                    # next block — or the block input that IS that residual sum): the earlier contributions arrive as the addend, the epilogue
                    # sees the complete gradient -> BN sums here
                    pu = self.units[tgt.id]
                    rparts = _lib.query(K("mny_pw_dgrad_bnred_parts"), M, oc, i.C)
                    rbuf = torch.empty(rparts * 2 * i.C, **f32)
                    self.fused_red[tgt.id] = (rbuf, rparts)
                    if wT6 is not None:
                        contribute_kernel(i, lambda out, addend, dY=dY, wT6=wT6, M=M, K=oc, Nc=i.C, pu=pu, rbuf=rbuf, act_=tgt.act: bwd.add(
                            "mny_pw_dgrad_bnred_w6", dY, wT6, addend, out, pu.Y, pu.scale, pu.shift, act_, pu.mean, pu.invstd, rbuf, M, K, Nc, self.stream,
                            label="mny_pw_dgrad_bnred_add",
                            meta=dict(flops=2 * M * K * Nc, bytes=self.eb * (M * K + 3 * M * Nc) + 4 * K * Nc, shape="dgrad+add+red M%d K%d N%d" % (M, K, Nc))))
                    else:
                        contribute_kernel(i, lambda out, addend, dY=dY, wT=wT, M=M, K=oc, Nc=i.C, pu=pu, rbuf=rbuf, act_=tgt.act: bwd.add(
                            self.K("mny_pw_dgrad_bnred_add"), dY, wT, addend, out, pu.Y, pu.scale, pu.shift, act_, pu.mean, pu.invstd, rbuf, M, K, Nc, self.stream,
                            meta=dict(flops=2 * M * K * Nc, bytes=self.eb * (M * K + 3 * M * Nc) + 4 * K * Nc, shape="dgrad+add+red M%d K%d N%d" % (M, K, Nc))))
                else:
                    contribute_kernel(i, lambda out, addend, dY=dY, wT=wT, wT6=wT6, M=M, K=oc, Nc=i.C: bwd.add(
                        "mny_pw_fwd_w6" if wT6 is not None else self.K("mny_pw_fwd"), dY, None, None, ACT_NONE, wT6 if wT6 is not None else wT, None, addend, out, None,
                        M, K, Nc, self.stream, label=self.K("mny_pw_fwd"),
                        meta=dict(flops=2 * M * K * Nc, bytes=self.eb * (M * K + M * Nc) + 4 * K * Nc, shape="dgrad M%d K%d N%d" % (M, K, Nc))))
            flush_shared()
            flush_reduce()
            bwd.marks[o.name] = len(bwd.calls)
        flush_reduce(force=True)

    # ------------------------------------------------------------------------------------------
    def _w6_planes(self, w2d, M, K, Nc):
        """Plane buffer for the weight operand `w2d` ([Nc][K] fp32) of a GEMM that takes the six-product bf16 form (fp32 plans), or None."""
        if self.bf16 or _lib.query("mny_pw_w6_supported", M, K, Nc) != 1:
            return None
        planes = torch.empty(_lib.query("mny_pw_w6_bytes", K, Nc), device=self.dev, dtype=torch.uint8)
        self._cut_jobs.append((w2d, planes, Nc, K))
        return planes

    def _flush_cut_jobs(self, calls, at_head=False):
        """One mny_cut3_batch launch for the pending (matrix, planes) pairs; at_head: placed first in the list (forward weights)."""
        if not self._cut_jobs:
            return
        import numpy as np
        jobs, block_job = [], []
        for src, planes, R, C in self._cut_jobs:
            jobs.append((src.data_ptr(), planes.data_ptr(), R, C, len(block_job), 0))
            block_job += [len(jobs) - 1] * ((R * ((C + 15) // 16) * 2 + 255) // 256)
        jt = np.array(jobs, dtype=np.dtype([("src", np.uint64), ("dst", np.uint64), ("R", np.int32), ("C", np.int32), ("b0", np.int32), ("pad", np.int32)]))
        jd = torch.from_numpy(jt.view(np.uint8).copy()).to(self.dev)
        bj = torch.tensor(block_job, dtype=torch.int32, device=self.dev)
        calls.add("mny_cut3_batch", jd, bj, len(block_job), self.stream)
        if at_head:
            calls.calls.insert(0, calls.calls.pop())
        calls.keep += [t for job in self._cut_jobs for t in job[:2]]
        self._cut_jobs = []

    def _gemm_weight(self, w):
        """Weight operand of a forward pointwise GEMM: the fp32 parameter itself, or (bf16 storage) a bf16 shadow copy
        refreshed by a conversion call placed right before the GEMM — the fp32 master copy is what optimizers update."""
        if not self.bf16:
            return w
        w16 = torch.empty(w.shape, device=self.dev, dtype=torch.bfloat16)
        if self.cvt_batch:
            self._cvt_jobs.append((w, w16))          # one batched conversion at the head of the forward list (_flush_cvt_jobs)
        else:
            self.fwd.add("mny_cvt_f32_bf16", w, w16, w.numel(), self.stream)
        return w16

    def _flush_cvt_jobs(self):
        if not self._cvt_jobs:
            return
        import numpy as np
        jobs, block_job = [], []
        for w, w16 in self._cvt_jobs:
            jobs.append((w.data_ptr(), w16.data_ptr(), w.numel(), len(block_job), 0))
            block_job += [len(jobs) - 1] * ((w.numel() + 4095) // 4096)
        jt = np.array(jobs, dtype=np.dtype([("src", np.uint64), ("dst", np.uint64), ("n", np.int64), ("b0", np.int32), ("pad", np.int32)]))
        self.c_jobs = torch.from_numpy(jt.view(np.uint8).copy()).to(self.dev)
        self.c_blocks = torch.tensor(block_job, dtype=torch.int32, device=self.dev)
        self.fwd.add("mny_cvt_batch_f32_bf16", self.c_jobs, self.c_blocks, len(block_job), self.stream)
        self.fwd.calls.insert(0, self.fwd.calls.pop())       # the shadows must exist before the first GEMM
        self.fwd.keep += [t for pair in self._cvt_jobs for t in pair]

    def K(self, name):
        """Entry point for the plan's activation storage type."""
        return name + "_bf16" if self.bf16 else name

    def kernel_routes(self):
        """[(entry point as called, label, (M, K, N), mny_pw_route family)] for every pointwise-conv call of the plan's two lists —
        which kernel family each launch takes (tests assert that a plan compared with the oracle contains the families the benchmark
        runs; tools/plan_stats.py prints the table)."""
        import re
        out = []
        lists = [("fwd", self.fwd.calls)] + ([("bwd", self.bwd.calls)] if getattr(self, "bwd", None) is not None else [])
        for which, calls in lists:
            for idx, (fn, _args, label, meta) in enumerate(calls):
                base = label[:-5] if label.endswith("_bf16") else label
                op = {"mny_pw_fwd": 0, "mny_pw_dgrad_bnred": 1, "mny_pw_dgrad_bnred_add": 1, "mny_pw_wgrad": 2}.get(base)
                m = re.search(r"M(\d+) K(\d+) N(\d+)", (meta or {}).get("shape", ""))
                if op is None or m is None:
                    continue
                M, K, N = (int(v) for v in m.groups())
                fam = self.routes[which].get(idx)             # what the dispatcher did; the predictor only before the first replay
                if fam is None:
                    fam = _lib.query("mny_pw_route", op, int(self.bf16), M, K, N)
                out.append((getattr(fn, "__name__", label), label, (M, K, N), fam))
        return out

    def stale(self):
        return any(t.data_ptr() != p for t, p in self.param_ptrs)

    def _bind(self, x):
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape) == (self.N, 3, self.H, self.W)
        if self.use_graphs:                      # graphs bake pointers: stage the batch in a resident buffer (0.15 ms at bs=256)
            if self.x_static is None:
                self.x_static = torch.empty_like(x)
            if x.data_ptr() != self.x_static.data_ptr():
                self.x_static.copy_(x)
            x = self.x_static
        self.x_ptr.value = x.data_ptr()
        self.stream.value = torch.cuda.current_stream(self.dev).cuda_stream
        return x

    def _replay(self, which, calls, begin=0, end=None):
        """Run calls[begin:end]: HIP-event bracketed (bench), as a captured hipGraph, or eagerly."""
        if self.timing is not None:
            calls.run_timed(self.timing[which], self.timing["only"], begin, end)
            return
        if not (self.use_graphs and self.eager_steps >= 2):
            key = (which, begin, end)
            if key not in self._recorded:
                calls.run_recording(self.routes[which], begin, end)          # the first replay of every segment: which kernel family each dispatcher took
                self._recorded.add(key)
            else:
                calls.run(begin, end)
            return
        key = (which, begin, end)
        g = self.graphs.get(key)
        if g is None:
            cur = torch.cuda.current_stream(self.dev)
            try:
                g = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream(self.dev)
                with torch.cuda.graph(g, stream=side):
                    self.stream.value = torch.cuda.current_stream(self.dev).cuda_stream
                    calls.run(begin, end)
            except Exception as e:               # noqa: BLE001 — capture is an optimisation, never a requirement
                warnings.warn("hipGraph capture failed (%s); staying on the eager call list" % (e,))
                self.use_graphs = False
                self.graphs.clear()
                self.stream.value = cur.cuda_stream
                calls.run(begin, end)
                return
            finally:
                self.stream.value = cur.cuda_stream
            self.graphs[key] = g
        g.replay()

    def set_targets(self, targets):
        """targets: list (len N) of [n_i,5] float tensors (label,cx,cy,w,h), CPU or device."""
        assert len(targets) == self.N, "need one target tensor per image"
        counts = [int(t.shape[0]) for t in targets]
        total = sum(counts)
        if total > self.t_dev.shape[0]:
            self.t_dev = torch.zeros(2 * total, 5, device=self.dev, dtype=torch.float32)
            self.graphs.clear()                  # captured graphs hold the old buffer's address
        acc, offs = 0, [0]
        for c in counts:
            acc += c
            offs.append(acc)
        off = torch.tensor(offs, dtype=torch.int32)
        if total:
            packed = torch.cat([t.reshape(-1, 5) for t in targets if t.shape[0]]).to(torch.float32)
            self.t_dev[:total].copy_(packed, non_blocking=True)
        self.off_dev.copy_(off, non_blocking=True)
        self.t_ptr.value = self.t_dev.data_ptr()
        self.off_ptr.value = self.off_dev.data_ptr()

    def forward_train(self, x, targets, seg_maps=None):
        x = self._bind(x)
        self.stream_side.value = self._side_stream.cuda_stream if (self.side_on and self.timing is None) else self.stream.value
        self.stream_side2.value = self._side2_stream.cuda_stream if (self.side_on and self.timing is None) else self.stream.value
        self.set_targets(targets)
        if self.seg_head is not None:
            if seg_maps is None:
                raise ValueError("this config has a `seg` section: training needs seg_maps [N,h,w,C] (train.py:257-258)")
            if tuple(seg_maps.shape) != tuple(self.seg_maps.shape):
                raise ValueError("seg_maps must be %s (N, H/16, W/16, seg classes), got %s" % (tuple(self.seg_maps.shape), tuple(seg_maps.shape)))
            self.seg_maps.copy_(seg_maps, non_blocking=True)                    # seg_loss.py:53 (clone().to(device))
        self._replay("fwd", self.fwd)
        self.saved_x = x
        self.fwd_gen += 1
        self.last_fwd_tick = next(_FWD_TICK)
        return self.out14

    def backward(self, g_losses):
        self.stream.value = torch.cuda.current_stream(self.dev).cuda_stream
        # event-bracketed runs (bench breakdown) keep everything on one stream so the brackets mean something
        self.stream_side.value = self._side_stream.cuda_stream if (self.side_on and self.timing is None) else self.stream.value
        self.stream_side2.value = self._side2_stream.cuda_stream if (self.side_on and self.timing is None) else self.stream.value
        self.x_ptr.value = self.saved_x.data_ptr()
        self.g_scale.copy_(g_losses.reshape(self.g_scale.numel()).to(self.g_scale.dtype))
        if self.reducer is not None:
            self.reducer.run_backward()             # segmented replay + bucketed RCCL all-reduce (dp.py)
        else:
            self.run_bwd_segment(0, None)
        self.eager_steps += 1

    def _fork_side(self):
        # one fork and one join event per plan, reused by every fork / join of a step (a wait captures the record that precedes it, so
        # re-recording the same event later is safe): round 2 created ~70 fresh torch.cuda.Event objects per step
        if self.side_on and self.timing is None:
            self._ev_fork.record(torch.cuda.current_stream(self.dev))
            self._side_stream.wait_event(self._ev_fork)
            self._side_used = True

    def _fork_side2(self):
        if self.side_on and self.timing is None:
            self._ev_fork2.record(torch.cuda.current_stream(self.dev))
            self._side2_stream.wait_event(self._ev_fork2)

    def _join_side2(self):
        if self.side_on and self.timing is None:
            self._ev_join2.record(self._side2_stream)
            torch.cuda.current_stream(self.dev).wait_event(self._ev_join2)

    def _join_side(self):
        if self._side_used:
            self._ev_join.record(self._side_stream)
            torch.cuda.current_stream(self.dev).wait_event(self._ev_join)
            self._side_used = False

    def run_bwd_segment(self, begin, end):
        self._replay("bwd", self.bwd, begin, end)
        self._join_side()                                    # gradients of this segment are complete on the main stream

    def enable_timing(self, only=None, steps=1):
        """Bracket calls with HIP events (bench.py roofline leg); disable with disable_timing().  `steps`: how many steps will
        be recorded before the events are read — their events are created here, outside the timed region."""
        sel = set(only) if only else None
        n = sum(1 for cl in (self.fwd, self.bwd) for c in cl.calls if sel is None or c[2] in sel)
        HipEvent.reserve(2 * n * max(int(steps), 1))
        self.timing = {"fwd": [], "bwd": [], "only": sel}

    def disable_timing(self):
        t, self.timing = self.timing, None
        return t

    def forward_eval(self, x, val_conf):
        x = self._bind(x)
        self.fwd.run()
        for i in range(2):
            self.val_conf[i].value = val_conf[i]
        self.det.run()
