"""Oracle (test infrastructure, NOT product): torch-CPU restatement of the network.

Restates models/mobilenetv2.py (:38-158) and models/mbv2_yolo.py (:16-173) with
stock torch.nn ops, NCHW fp32, and the reference's exact ``state_dict`` keys
(430 for the VOC config).  Pinned against tests/golden/net_*.npz and
state_keys_*.json, captured from the real reference by tools/gen_golden.py.
It is also the ``cpu_baseline`` ("port") leg of bench.py.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import yolo_ref
from . import nms_ref
from . import seg_ref


def _store(t):
    """Storage hook of a MATERIALISED sum (residual adds): identity for the fp32 reference; oracle/bf16_storage.py swaps in a
    straight-through bf16 rounding to model the `act_dtype=bfloat16` plans of the product (BASELINE configs[3])."""
    return STORE(t)


STORE = lambda t: t        # noqa: E731


def _cbr6(cin, cout, k, stride, groups=1):
    """conv -> BN -> ReLU6 triple as three Sequential children (mobilenetv2.py:38-51)."""
    return [nn.Conv2d(cin, cout, k, stride, k // 2, groups=groups, bias=False),
            nn.BatchNorm2d(cout), nn.ReLU6(inplace=True)]


class Bottleneck(nn.Module):
    """mobilenetv2.py:54-91.  Attribute name `conv` fixes the state_dict keys."""

    def __init__(self, cin, cout, stride, t):
        super().__init__()
        hid = round(cin * t)
        self.identity = stride == 1 and cin == cout
        layers = []
        if t != 1:
            layers += _cbr6(cin, hid, 1, 1)
        layers += _cbr6(hid, hid, 3, stride, groups=hid)
        layers += [nn.Conv2d(hid, cout, 1, 1, 0, bias=False), nn.BatchNorm2d(cout)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        y = self.conv(x)
        return _store(x + y) if self.identity else y


class Backbone(nn.Module):
    """mobilenetv2.py:94-158 (width 1.0): features -> x1 (96ch, /16); features2+conv -> x2 (1280ch, /32)."""

    STAGES1 = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1)]
    STAGES2 = [(6, 160, 3, 2), (6, 320, 1, 1)]

    def __init__(self):
        super().__init__()
        c = 32
        first = [nn.Sequential(*_cbr6(3, c, 3, 2))]
        for t, co, n, s in self.STAGES1:
            for i in range(n):
                first.append(Bottleneck(c, co, s if i == 0 else 1, t))
                c = co
        self.features = nn.Sequential(*first)
        second = []
        for t, co, n, s in self.STAGES2:
            for i in range(n):
                second.append(Bottleneck(c, co, s if i == 0 else 1, t))
                c = co
        self.features2 = nn.Sequential(*second)
        self.conv = nn.Sequential(*_cbr6(c, 1280, 1, 1))
        for m in self.modules():                                    # :146-158
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, x):
        x1 = self.features(x)
        return x1, self.conv(self.features2(x1))


class ConvBnLeaky(nn.Module):
    """mbv2_yolo.py:16-44 (BasicConv): conv(no bias, pad k//2) -> BN -> LeakyReLU(0.1)."""

    def __init__(self, cin, cout, k, depthwise=False):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, 1, k // 2, bias=False, groups=cin if depthwise else 1)
        self.bn = nn.BatchNorm2d(cout)
        nn.init.kaiming_normal_(self.conv.weight, mode="fan_out")  # :35
        nn.init.constant_(self.bn.weight, 1)
        nn.init.constant_(self.bn.bias, 0)

    def forward(self, x):
        return F.leaky_relu(self.bn(self.conv(x)), 0.1)


def _dw_pw_pw(cin, cout):                                           # mbv2_yolo.py:70-76
    return nn.Sequential(ConvBnLeaky(cin, cin, 3, True), ConvBnLeaky(cin, cin, 1), ConvBnLeaky(cin, cout, 1))


def _head(mid, cout, cin):                                          # mbv2_yolo.py:77-92
    return nn.Sequential(ConvBnLeaky(cin, cin, 3, True), ConvBnLeaky(cin, cin, 1),
                         ConvBnLeaky(cin, mid, 1), nn.Conv2d(mid, cout, 1))


class Residual(nn.Module):                                          # mbv2_yolo.py:93-104 (Connect)
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Sequential(ConvBnLeaky(c, c, 3, True), ConvBnLeaky(c, c, 1))

    def forward(self, x, extra=None):
        s = torch.add(x, self.conv(x))
        if extra is not None:                                       # mbv2_yolo.py:151 (the sum the product forms in the same pass)
            s = torch.add(s, extra)
        return _store(s)


class RefYolo(nn.Module):
    """mbv2_yolo.py:105-173.  Same constructor config, forward contract and keys;
    never downloads weights (Q11).  A config with a ``seg`` key adds ``seg_headS16`` and the
    segmentation loss (oracle/seg_ref.py) exactly as :110-114,161-171 do."""

    def __init__(self, config):
        super().__init__()
        self.num_classes = config["yolo"]["num_classes"]
        self.num_anchors = config["yolo"]["num_anchors"]
        self.has_seg = "seg" in config
        if self.has_seg:
            self.seg_headS16 = _head(32, config["seg"]["num_classes"], 32)
        out_ch = self.num_anchors * (5 + self.num_classes)
        self.backbone = Backbone()
        self.conv_for_S32 = ConvBnLeaky(1280, 512, 1)
        self.connect_for_S32 = Residual(512)
        self.yolo_headS32 = _head(1024, out_ch, 512)
        self.conv_for_S16 = _dw_pw_pw(96, 512)
        self.seg_conv_for_S16 = _dw_pw_pw(96, 32)
        self.connect_for_S16 = Residual(512)
        self.seg_connect_for_S16 = Residual(32)
        self.yolo_headS16 = _head(512, out_ch, 512)
        self.specs = yolo_ref.specs_from_config(config)             # plain list: not in state_dict
        self.img_size = [config["img_w"], config["img_h"]]

    def heads(self, x):
        f1, f2 = self.backbone(x)
        s32 = self.connect_for_S32(self.conv_for_S32(f2))
        out0 = self.yolo_headS32(s32)
        up = F.interpolate(s32, scale_factor=2, mode="nearest")
        s16 = self.connect_for_S16(self.conv_for_S16(f1), extra=up)                  # :147,:151
        out1 = self.yolo_headS16(s16)
        self.seg_branch = self.seg_connect_for_S16(self.seg_conv_for_S16(f1))         # always executed (Q10)
        return out0, out1

    def forward(self, x, targets=None, seg_maps=None):
        self.img_size = [x.size(2), x.size(3)]                      # :139-140 (Q8)
        out0, out1 = self.heads(x)
        if targets is not None:
            output = tuple(yolo_ref.loss_forward(o, targets, s, self.img_size)
                           for o, s in zip((out0, out1), self.specs))
            if self.has_seg:                                        # :167-170
                self.out2 = self.seg_headS16(self.seg_branch)
                return output, seg_ref.seg_loss(self.out2, seg_maps)
            return output
        rows = tuple(yolo_ref.decode_rows(o, s, self.img_size) for o, s in zip((out0, out1), self.specs))
        output = nms_ref.nms_driver(rows, self.num_classes)
        if self.has_seg:                                            # :161-164
            self.out2 = self.seg_headS16(self.seg_branch)
            return output, seg_ref.seg_eval(self.out2)
        return output
