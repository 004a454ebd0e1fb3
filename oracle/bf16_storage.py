"""Oracle (test infrastructure, NOT product): the fp32 CPU restatement with the product's bf16 STORAGE points modelled.

`yolo(config, act_dtype=torch.bfloat16)` (BASELINE configs[3]) keeps every arithmetic step in fp32 and rounds (RNE) where a tensor
is written to HBM.  A randomly initialised ~80-layer network with BatchNorm amplifies a 2^-9 perturbation to 15-35 % rms at the
heads (measured on the fp32 oracle itself, tests/test_gpu_bf16.py), so the distance to the fp32 reference says little; parity for
that plan is therefore checked against THIS model of it: the same reference ops (oracle/net_ref*.py, each citing the reference
lines it restates) with a straight-through bf16 rounding at exactly the product's storage points:

  * the raw output of every convolution (stem, depthwise, pointwise, biased heads — bias added in fp32 first);
  * the A operand of every pointwise GEMM: bf16(act(bn(y))) — the value a materialised bf16 activation would hold — and the
    pointwise weights (bf16 shadow copies of the fp32 masters); depthwise / stem inputs and filters stay fp32;
  * every materialised sum / product (residual adds, the per-pixel gate multiply, PartAdd) via net_ref.STORE.
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


@contextlib.contextmanager
def bf16_storage(model):
    """Inside the context, `model` (a RefYolo / RefYoloV3) computes what the product's bf16-storage plan computes."""
    convs = [m for m in model.modules() if isinstance(m, nn.Conv2d)]
    for m in convs:
        m.forward = types.MethodType(_conv_forward, m)
    old = net_ref.STORE
    net_ref.STORE = q
    try:
        yield model
    finally:
        net_ref.STORE = old
        for m in convs:
            del m.forward
