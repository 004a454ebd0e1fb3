"""Drop-in `yolo` module: same constructor config, forward contract and `state_dict` keys as the
reference's models/mbv2_yolo.py:105-173, executed entirely by libmnyolo (HIP) through static plans.

    model = yolo(config).cuda()
    outputs = model(images, targets)          # training: ((loss, recall, avg_iou, obj, no_obj, cls, count), (...))
    sum(o[0] for o in outputs).backward()     # fills p.grad for the 202 trainable tensors
    detections = model.eval()(images)         # list (len N) of [k_i,7] tensors after per-class NMS

There is no CPU fallback: calling the module with CPU tensors, or without the built library, raises.
The nn.Conv2d / nn.BatchNorm2d children are parameter holders only (their own forward is never used),
which is what keeps checkpoints written by the reference's train.py:175-226 loadable.
"""
import math

import torch
import torch.nn as nn

from . import _lib
from .arch import mbv2_yolo_graph, mbv3_yolo_graph
from .engine import NetPlan


class Holder(nn.Module):
    """Plain container; children are registered under the reference's names / indices."""

    def forward(self, *a, **k):
        raise RuntimeError("parameter holder: use the top-level yolo module")

    def __getitem__(self, i):          # index like the reference's nn.Sequential children
        return self._modules[str(i)]

    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())


class HeadState:
    """Python-side state of one detection head — the attributes train.py / inference.py touch on
    `model.yolo_losses[i]` (val_conf r/w: train.py:149-150,417-418; img_size: mbv2_yolo.py:139-140)."""

    def __init__(self, anchors, mask, num_classes, img_size, ignore_threshold, iou_thresh, val_conf=0.1, iou_weighting=0.01):
        self.anchors = [tuple(a) for a in anchors]
        self.mask = list(mask)
        self.num_mask = len(mask)
        self.num_anchors = len(anchors)
        self.num_classes = num_classes
        self.bbox_attrs = 5 + num_classes
        self.img_size = img_size
        self.ignore_threshold = ignore_threshold
        self.iou_thresh = iou_thresh
        self.val_conf = val_conf
        self.iou_weighting = iou_weighting


class _InFlight:
    """Marks a plan whose forward has run and whose backward is still owed.  A further differentiable forward of the same shape then runs on a
    SECOND plan (own activations, own gradient arena) instead of overwriting what the pending backward needs — `a = model(x, t); b = model(x, t);
    a_loss.backward(); b_loss.backward()` works like it does under autograd.  Released by backward() or when the autograd graph is dropped."""

    def __init__(self, plan):
        self.plan, self.gen = plan, plan.fwd_gen
        plan.inflight_gen = plan.fwd_gen

    def release(self):
        if self.plan is not None and getattr(self.plan, "inflight_gen", None) == self.gen:
            self.plan.inflight_gen = None
        self.plan = None

    def __del__(self):
        self.release()


class _TrainStep(torch.autograd.Function):
    """One autograd node for the whole network + both losses.  Parameter gradients are written
    straight into the model's flat gradient arena (and exposed as `p.grad` views) instead of being
    returned through autograd, so no per-parameter AccumulateGrad work happens."""

    @staticmethod
    def forward(ctx, x, anchor, model, plan, targets, seg_maps=None):
        out14 = plan.forward_train(x, targets, seg_maps)
        res = out14.clone()
        ctx.model, ctx.plan, ctx.gen = model, plan, plan.fwd_gen
        ctx.token = _InFlight(plan)               # the plan's saved activations are spoken for until backward() ran or this graph is dropped
        losses, metrics = res[:, 0].contiguous(), res[:, 1:].contiguous()
        if plan.seg_head is not None:             # third loss: SegLoss (mbv2_yolo.py:167-170); its two means ride along
            seg3 = plan.seg_out3.clone()
            losses = torch.cat((losses, seg3[:1]))
            metrics = torch.cat((metrics.reshape(-1), seg3[1:]))
        ctx.mark_non_differentiable(metrics)
        return losses, metrics

    @staticmethod
    def backward(ctx, g_losses, _g_metrics):
        if ctx.plan.fwd_gen != ctx.gen:           # the plan's resident activations belong to a later forward
            raise RuntimeError("backward() of a step whose plan has run another forward since (static plans keep ONE set of saved "
                               "activations per (batch, height, width)): call backward before the next forward of the same shape")
        ctx.model._run_backward(ctx.plan, g_losses.contiguous())
        ctx.token.release()
        return None, None, None, None, None, None


class yolo(nn.Module):
    ARCH = "mbv2"

    def __init__(self, config, sync_metrics=False, act_dtype=torch.float32):
        """`act_dtype=torch.bfloat16` stores activations / activation gradients in bf16 (BASELINE config 4); parameters,
        their gradients, BN statistics and the detection heads stay fp32, so optimizers and checkpoints are unchanged."""
        super().__init__()
        assert act_dtype in (torch.float32, torch.bfloat16), "act_dtype must be torch.float32 or torch.bfloat16"
        self.act_dtype = act_dtype
        y = config["yolo"]
        self.num_classes = y["num_classes"]
        self.num_anchors = y["num_anchors"]
        self.seg_num_classes = config["seg"]["num_classes"] if "seg" in config else None
        self.sync_metrics = sync_metrics          # True: metrics as python floats like the reference (forces a device sync)
        if self.ARCH == "mbv2":
            self.graph = mbv2_yolo_graph(self.num_classes, self.num_anchors, self.seg_num_classes)
        else:
            self.graph = mbv3_yolo_graph(self.num_classes, self.num_anchors)
        self.has_seg = self.graph.seg_out is not None
        self._build_modules()
        self.yolo_losses = [HeadState(y["anchors"], y["mask"][i], self.num_classes, [config["img_w"], config["img_h"]],
                                      y["ignore_thresh"][i], y["iou_thresh"], iou_weighting=config["iou_weighting"])
                            for i in range(2)]                                   # plain list: not in state_dict (mbv2_yolo.py:132)
        self.img_size = [config["img_w"], config["img_h"]]
        self._plans = {}
        self._anchor = None
        self.grad_hook = None                      # DP: called as hook(plan) after gradients are complete

    def __getstate__(self):           # torch.save(model) (train.py:431): plans hold raw device pointers
        d = self.__dict__.copy()
        d["_plans"], d["_anchor"], d["grad_hook"] = {}, None, None
        d.pop("dp_reducer", None)
        return d

    # ---- parameters ---------------------------------------------------------------------------
    def _build_modules(self):
        for path, kind, a in self.graph.modules:
            parts = path.split(".")
            m = self
            for p in parts[:-1]:
                if p not in m._modules:
                    m.add_module(p, Holder())
                m = m._modules[p]
            if kind == "conv":
                cin, cout, k, stride, groups, bias = a
                mod = nn.Conv2d(cin, cout, k, stride, k // 2, groups=groups, bias=bias)
                if path.startswith("backbone.") and self.ARCH == "mbv2":      # mobilenetv2.py:146-152
                    mod.weight.data.normal_(0, math.sqrt(2.0 / (k * k * cout)))
                elif not bias:                                                 # BasicConv: mbv2_yolo.py:32-36; mobilenetv3.py:113-116
                    nn.init.kaiming_normal_(mod.weight, mode="fan_out")
                # biased head convs keep nn.Conv2d's default init (mbv2_yolo.py:82)
            else:
                mod = nn.BatchNorm2d(a[0])                                     # weight 1, bias 0 (both inits)
            m.add_module(parts[-1], mod)

    @property
    def device(self):
        return next(self.parameters()).device

    def all_state_tensors(self):
        return [t for t in self.state_dict(keep_vars=True).values()]

    @property
    def param_tensors(self):
        return {k: (v.data if isinstance(v, nn.Parameter) else v) for k, v in self.state_dict(keep_vars=True).items()}

    def load_pretrained_backbone(self, checkpoint, strict_shapes=True):
        """The ImageNet-backbone loader of `mobilenetv2(pretrained)` (models/mobilenetv2.py:161-181, called from
        mbv2_yolo.py:116-117), from a LOCAL state dict or file — never a download.

        Same key rule as the reference: a checkpoint key (with `module.` removed) is matched against every backbone key after
        renaming `features2.{0..3}.` to `features.{14..17}.` (the reference splits torchvision-style `features` into
        `features` [0..13] and `features2` [0..3]); matches are copied, everything else (the checkpoint's classifier, backbone
        keys the checkpoint lacks) is left alone; when several checkpoint keys collapse onto one name the last one wins, as in
        the reference's nested loop.  Returns the list of backbone keys that were loaded."""
        if self.ARCH != "mbv2":
            raise ValueError("load_pretrained_backbone is the MobileNetV2 loader (models/mobilenetv2.py:161); "
                             "the MobileNetV3 reference has none (mbv3_yolo.py:104)")
        if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
            path = str(checkpoint)
            if "://" in path:
                raise ValueError("pass a local file or a state dict: this build never downloads")
            checkpoint = torch.load(path, map_location="cpu")
        own = self.backbone.state_dict()
        alias = {}
        for k2 in own:
            n2 = k2
            for i in range(4):                                                  # mobilenetv2.py:173-176
                n2 = n2.replace("features2.%d." % i, "features.%d." % (14 + i))
            alias[n2] = k2
        picked = {}
        for k1, v1 in checkpoint.items():
            k2 = alias.get(k1.replace("module.", ""))
            if k2 is not None:
                picked[k2] = v1
        for k2, v in picked.items():
            if tuple(v.shape) != tuple(own[k2].shape):
                if strict_shapes:                                               # what load_state_dict (:180) raises on
                    raise RuntimeError("size mismatch for %s: checkpoint %s vs model %s" % (k2, tuple(v.shape), tuple(own[k2].shape)))
                continue
            with torch.no_grad():
                own[k2].copy_(v)                                                # state_dict tensors alias the live parameters
        return sorted(picked)

    # ---- plans --------------------------------------------------------------------------------
    PLAN_BUDGET_FRAC = 0.6        # share of the device's HBM the cached plans may keep resident

    def _plan(self, N, H, W, training, slot=0):
        """`training`: True (loss + backward, batch statistics), False (decode + NMS, running statistics), "traindet" (decode + NMS on BATCH
        statistics with the running-statistics update: `model.train()(images)`, mbv2_yolo.py:158-166 in training mode), "evalloss" (loss on running
        statistics, no statistics update, forward only: `model.eval()(images, targets)` under no_grad) or "evalgrad" (the same with the
        backward list of frozen BatchNorm: `model.eval()(images, targets)` with gradients, mbv2_yolo.py:157)."""
        key = (N, H, W, training) if self.act_dtype == torch.float32 else (N, H, W, training, "bf16")
        if slot:
            key = key + ("slot%d" % slot,)
        p = self._plans.get(key)
        if p is None or p.stale():
            if not torch.cuda.is_available() or self.device.type != "cuda":
                raise _lib.MnyError("yolo runs only on an MI355X (HIP) device: move the module and inputs to cuda; "
                                    "there is no CPU fallback")
            _lib.load()
            self._plans.pop(key, None)
            before = torch.cuda.memory_allocated(self.device)
            p = NetPlan(self, N, H, W, training in (True, "evalloss", "evalgrad"), self.act_dtype, bn_batch=(training is True or training == "traindet"),
                        frozen_bwd=(training == "evalgrad"))
            p.resident_bytes = max(torch.cuda.memory_allocated(self.device) - before, 0)
            self._plans[key] = p
            # multi-scale training keeps one plan per size: bound them by resident BYTES (a bs=256/352x352 training plan holds
            # ~40 GB), oldest first, never the one just built
            budget = self.PLAN_BUDGET_FRAC * torch.cuda.get_device_properties(self.device).total_memory
            evicted = False
            while len(self._plans) > 1 and (len(self._plans) > 8 or sum(q.resident_bytes for q in self._plans.values()) > budget):
                self._plans.pop(next(iter(self._plans)))
                evicted = True
            if evicted:
                import gc
                gc.collect()                                                    # a plan holds reference cycles (its builder closures): free its HBM now, not at some later collection
        else:
            self._plans[key] = self._plans.pop(key)                             # most recently used last
        return p

    MAX_INFLIGHT = 2                               # differentiable forwards of one shape that may await their backward at the same time

    def _grad_plan(self, N, H, W, mode):
        """The plan a differentiable forward runs on: the first one of the shape whose previous step is settled (backward ran, or its graph was
        dropped); all MAX_INFLIGHT busy: the one whose forward is oldest (its pending backward then raises, as a stale step always did)."""
        plans = []
        for slot in range(self.MAX_INFLIGHT):
            p = self._plan(N, H, W, mode, slot)
            if getattr(p, "inflight_gen", None) is None:
                return p
            plans.append(p)
        return min(plans, key=lambda q: q.last_fwd_tick)

    def _check_input(self, x):
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise _lib.MnyError("input must be a CUDA(HIP) tensor — the HIP path has no CPU fallback")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected images [N,3,H,W], got %s" % (tuple(x.shape),))
        return x.contiguous().float()

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, x, targets=None, seg_maps=None):
        x = self._check_input(x)
        N, _, H, W = x.shape
        for hs in self.yolo_losses:
            hs.img_size = [H, W]                                                # mbv2_yolo.py:139-140
        if targets is None:
            return self._forward_eval(x)
        if self.has_seg and seg_maps is not None:
            seg_maps = seg_maps.to(device=x.device, dtype=torch.float32)
        if self.training:
            plan = self._grad_plan(N, H, W, True)
            if self._anchor is None or self._anchor.device != x.device:
                self._anchor = torch.zeros((), device=x.device, requires_grad=True)
            nbt = [b for k, b in self.named_buffers() if k.endswith("num_batches_tracked")]
            torch._foreach_add_(nbt, 1)
            losses, metrics = _TrainStep.apply(x, self._anchor, self, plan, targets, seg_maps if self.has_seg else None)
        elif torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # model.eval()(images, targets) with gradients — frozen-BatchNorm fine-tuning (mbv2_yolo.py:157 returns differentiable losses in
            # eval mode): running statistics in the forward, untouched buffers, and a backward in which they are constants
            plan = self._grad_plan(N, H, W, "evalgrad")
            if self._anchor is None or self._anchor.device != x.device:
                self._anchor = torch.zeros((), device=x.device, requires_grad=True)
            losses, metrics = _TrainStep.apply(x, self._anchor, self, plan, targets, seg_maps if self.has_seg else None)
        else:
            # model.eval()(images, targets) under no_grad — a validation loss: nn.BatchNorm2d normalises with the running statistics and
            # leaves them untouched (mobilenetv2.py:41-84 in eval mode); forward-only plan
            plan = self._plan(N, H, W, "evalloss")
            with torch.no_grad():
                res = plan.forward_train(x, targets, seg_maps if self.has_seg else None).clone()
                losses, metrics = res[:, 0].contiguous(), res[:, 1:].contiguous()
                if plan.seg_head is not None:
                    seg3 = plan.seg_out3.clone()
                    losses = torch.cat((losses, seg3[:1]))
                    metrics = torch.cat((metrics.reshape(-1), seg3[1:]))
        seg_metrics = None
        if self.has_seg:
            seg_metrics, metrics = metrics[12:], metrics[:12].view(2, 6)
        out = []
        if self.sync_metrics:
            m = metrics.tolist()
        for i in range(2):
            if self.sync_metrics:     # reference types: python floats, no_obj a tensor (yolo_loss.py:170-178)
                r = (losses[i], m[i][0], m[i][1], m[i][2], metrics[i, 3], m[i][4], m[i][5])
            else:
                r = (losses[i],) + tuple(metrics[i, j] for j in range(6))
            out.append(r)
        if seg_metrics is not None:               # (loss*0.05, mean(obj).item(), mean(no_obj).item())  seg_loss.py:76
            sm = seg_metrics.tolist() if self.sync_metrics else (seg_metrics[0], seg_metrics[1])
            return tuple(out), (losses[2], sm[0], sm[1])
        return tuple(out)

    def _run_backward(self, plan, g_losses):
        P = dict(self.named_parameters())
        # gradient accumulation (backward twice without zero_grad(set_to_none=True)): keep the old arena
        first = P[plan.grad_params[0]]
        prev = None
        red = getattr(self, "dp_reducer", None)
        plan.reducer = red.for_plan(plan) if red is not None else None
        if first.grad is not None and first.grad.data_ptr() == plan.gviews[plan.grad_params[0]].data_ptr():
            if plan.reducer is not None:
                plan.reducer.wait()               # the previous step's last buckets may still be in flight on the arena
            prev = plan.gflat.clone()
        plan.backward(g_losses)
        if prev is not None:
            if plan.reducer is not None:
                plan.reducer.wait()               # accumulate into AVERAGED gradients, never under a running all-reduce
            plan.gflat.add_(prev)
        for nm in plan.grad_params:
            p = P[nm]
            gview = plan.gviews[nm]
            if p.grad is None:
                p.grad = gview
            elif p.grad.data_ptr() != gview.data_ptr():
                p.grad.add_(gview)                                              # foreign .grad tensor: accumulate into it
        if self.grad_hook is not None:
            self.grad_hook(plan)

    def _forward_eval(self, x):
        N, _, H, W = x.shape
        if self.training:
            # model.train()(images): the reference decodes + runs NMS in any mode (mbv2_yolo.py:158-166); its BatchNorm layers are then in
            # training mode — batch statistics, running statistics and num_batches_tracked updated — and so are these
            plan = self._plan(N, H, W, "traindet")
            nbt = [b for k, b in self.named_buffers() if k.endswith("num_batches_tracked")]
            torch._foreach_add_(nbt, 1)
        else:
            plan = self._plan(N, H, W, False)
        with torch.no_grad():
            plan.forward_eval(x, [float(h.val_conf) for h in self.yolo_losses])
            counts = plan.out_counts.tolist()                                   # the one host sync of the eval path
            if int(plan.nms_status.item()) != 0:
                raise _lib.MnyError("NMS: a bucket of %d boxes exceeded the plan's per-image candidate bound" % int(plan.nms_status.item()))
            total = sum(counts)
            dets = plan.out_rows[:total].clone()
        dets = list(torch.split(dets, counts))                                  # list of [k_i,7] (mbv2_yolo.py:159-166)
        if plan.seg_head is not None:
            return dets, plan.seg_eval.cpu().numpy()                            # (output, seg_out) mbv2_yolo.py:161-164
        return dets
