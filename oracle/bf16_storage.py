"""Oracle (test infrastructure, NOT product): the fp32 CPU restatement with the product's bf16 STORAGE points modelled.

`yolo(config, act_dtype=torch.bfloat16)` (BASELINE configs[3]) keeps every arithmetic step in fp32 and rounds (RNE) where a tensor
is written to HBM.  A randomly initialised ~80-layer network with BatchNorm amplifies a 2^-9 perturbation to 15-35 % rms at the
heads (measured on the fp32 oracle itself, tests/test_gpu_bf16.py), so the distance to the fp32 reference says little; parity for
that plan is therefore checked against THIS model of it: the same reference ops (oracle/net_ref*.py, each citing the reference
lines it restates) with a straight-through bf16 rounding at exactly the product's storage points:

  * the raw output of every convolution (stem, depthwise, pointwise, biased heads — bias added in fp32 first);
  * the A operand of every pointwise GEMM: bf16(act(bn(y))) — the value a materialised bf16 activation would hold — and the
    pointwise weights (bf16 shadow copies of the fp32 masters); depthwise / stem inputs and filters stay fp32;
  * every materialised sum / product (residual adds, the per-pixel gate multiply, PartAdd) via net_ref.STORE;
  * MobileNetV3's per-pixel gates run as one unit (csrc/gate.hip): the operands of their two 1x1 convs are rounded like every GEMM
    operand (whatever the channel count), their OUTPUTS (the hidden tensors) are never stored, hence not rounded, and where the block's
    residual add absorbs the multiply only the sum is rounded.
BatchNorm statistics are then taken over the rounded outputs, as the product does.  Gradients flow straight through the roundings
(the product additionally rounds activation gradients to bf16: a non-chaotic 2^-9-per-layer effect the tests' bounds absorb).
"""
import contextlib
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import net_ref


def q(t):
    """bf16 round-to-nearest-even with a straight-through gradient."""
    return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())


def _conv_forward(self, x):
    w = self.weight
    if self.kernel_size == (1, 1) and self.groups == 1:
        w = q(w)
        if self.in_channels % 8 == 0:         # the bf16-MFMA path; K % 8 != 0 (MobileNetV3's 10/28-channel gates) widens to fp32
            x = q(x)
    return q(F.conv2d(x, w, self.bias, self.stride, self.padding, self.dilation, self.groups))


def _gate_conv_forward(self, x):
    return F.conv2d(q(x), q(self.weight), self.bias, self.stride, self.padding, self.dilation, self.groups)      # operands rounded, output kept in fp32


@contextlib.contextmanager
def bf16_storage(model, gate_fused=True, absorbed=None):
    """Inside the context, `model` (a RefYolo / RefYoloV3) computes what the product's bf16-storage plan computes.

    The two facts about the plan that move rounding points are ARGUMENTS, taken by the tests from the product's plan itself so that the
    model cannot drift from what ran: `gate_fused` — the per-pixel gates run as one unit (plan.gates non-empty; with
    MNY_NO_GATE=1 their two convs are ordinary convs with stored outputs and the multiply is a stored tensor); `absorbed` — names of the
    blocks (e.g. "backbone.bneck.3") whose residual add absorbs the gate's multiply (plan.gates[...]["add"] is not None); None = every
    stride-1 block with a gate (what the default plan does)."""
    from . import net_ref_v3
    gates = [m for m in model.modules() if isinstance(m, net_ref_v3.PixelGate)]
    gate_convs = {id(c) for gt in gates for c in gt.modules() if isinstance(c, nn.Conv2d)} if gate_fused else set()
    convs = [m for m in model.modules() if isinstance(m, nn.Conv2d)]
    for m in convs:
        m.forward = types.MethodType(_gate_conv_forward if id(m) in gate_convs else _conv_forward, m)
    for name, blk in model.named_modules():
        if isinstance(blk, net_ref_v3.V3Block) and blk.se is not None:
            blk.se._absorbed_by_add = gate_fused and (blk.stride == 1 if absorbed is None else name in absorbed)
    old = net_ref.STORE
    net_ref.STORE = q
    try:
        yield model
    finally:
        net_ref.STORE = old
        for m in convs:
            del m.forward
        for gt in gates:
            if hasattr(gt, "_absorbed_by_add"):
                del gt._absorbed_by_add
