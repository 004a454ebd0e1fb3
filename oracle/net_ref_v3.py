"""Oracle (test infrastructure, NOT product): torch-CPU restatement of MobileNetV3-Large-YOLO.

Restates models/mobilenetv3.py:14-136 and models/mbv3_yolo.py:16-145 with stock torch.nn ops and the
reference's state_dict keys.  Pinned against tests/golden/net_v3_*.npz / state_keys_mbv3.json, captured from the
real reference (imported with the `models.voc.*` aliases of SURVEY §8c) by tools/gen_golden.py.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nms_ref, yolo_ref
from .net_ref import ConvBnLeaky, Residual, _dw_pw_pw, _head, _store


def hswish(x):
    return x * F.relu6(x + 3) / 6                      # mobilenetv3.py:14-17


class PixelGate(nn.Module):
    """mobilenetv3.py:26-41: named SeModule there, but forward never pools — a per-pixel gate."""

    def __init__(self, c, r=4):
        super().__init__()
        self.se = nn.Sequential(nn.Conv2d(c, c // r, 1, bias=False), nn.BatchNorm2d(c // r), nn.ReLU(inplace=True),
                                nn.Conv2d(c // r, c, 1, bias=False), nn.BatchNorm2d(c), nn.Identity())

    def forward(self, x):
        y = x * (F.relu6(self.se(x) + 3) / 6)
        # bf16-storage model of the product's fused gate unit (oracle/bf16_storage.py sets the flag): where the block's residual add
        # absorbs the multiply, the product is not a stored tensor — only the sum is rounded
        return y if getattr(self, "_absorbed_by_add", False) else _store(y)


class V3Block(nn.Module):
    """mobilenetv3.py:44-74."""

    def __init__(self, k, cin, exp, cout, act, gate, stride):
        super().__init__()
        self.stride, self.act = stride, act
        self.se = PixelGate(cout) if gate else None
        self.conv1 = nn.Conv2d(cin, exp, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(exp)
        self.conv2 = nn.Conv2d(exp, exp, k, stride, k // 2, groups=exp, bias=False)
        self.bn2 = nn.BatchNorm2d(exp)
        self.conv3 = nn.Conv2d(exp, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.shortcut = nn.Sequential()
        if stride == 1 and cin != cout:
            self.shortcut = nn.Sequential(nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        f = F.relu if self.act == "relu" else hswish
        y = f(self.bn1(self.conv1(x)))
        y = f(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        if self.se is not None:
            y = self.se(y)
        return _store(y + self.shortcut(x)) if self.stride == 1 else y


class V3Backbone(nn.Module):
    CFG1 = [(3, 16, 16, "relu", 0, 1), (3, 64, 24, "relu", 0, 2), (3, 72, 24, "relu", 0, 1), (5, 72, 40, "relu", 1, 2),
            (5, 120, 40, "relu", 1, 1), (5, 120, 40, "relu", 1, 1), (3, 240, 80, "hs", 0, 2), (3, 200, 80, "hs", 0, 1),
            (3, 184, 80, "hs", 0, 1), (3, 184, 80, "hs", 0, 1), (3, 480, 112, "hs", 1, 1), (3, 672, 112, "hs", 1, 1),
            (5, 672, 160, "hs", 1, 1)]
    CFG2 = [(5, 672, 160, "hs", 1, 2), (5, 960, 160, "hs", 1, 1)]

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 16, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(16)
        c, blocks = 16, []
        for k, e, o, a, g, s in self.CFG1:
            blocks.append(V3Block(k, c, e, o, a, g, s))
            c = o
        self.bneck = nn.Sequential(*blocks)
        blocks = []
        for k, e, o, a, g, s in self.CFG2:
            blocks.append(V3Block(k, c, e, o, a, g, s))
            c = o
        self.bneck2 = nn.Sequential(*blocks)
        self.conv2 = nn.Conv2d(160, 960, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(960)
        for m in self.modules():                                   # mobilenetv3.py:111-123
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")

    def forward(self, x):
        y = hswish(self.bn1(self.conv1(x)))
        f1 = self.bneck(y)
        return f1, hswish(self.bn2(self.conv2(self.bneck2(f1))))


class RefYoloV3(nn.Module):
    """mbv3_yolo.py:97-145."""

    def __init__(self, config):
        super().__init__()
        self.num_classes = config["yolo"]["num_classes"]
        self.num_anchors = config["yolo"]["num_anchors"]
        out_ch = self.num_anchors * (5 + self.num_classes)
        self.backbone = V3Backbone()
        self.conv_for_S32 = _dw_pw_pw(960, 320)
        self.connect_for_S32 = Residual(320)
        self.yolo_headS32 = _head(960, out_ch, 320)
        self.connect_for_S16 = Residual(160)
        self.yolo_headS16 = _head(640, out_ch, 320)
        self.specs = yolo_ref.specs_from_config(config)
        self.img_size = [config["img_w"], config["img_h"]]

    def heads(self, x):
        f1, f2 = self.backbone(x)
        s32 = self.connect_for_S32(self.conv_for_S32(f2))
        out0 = self.yolo_headS32(s32)
        up = F.interpolate(s32, scale_factor=2, mode="nearest")
        s16 = self.connect_for_S16(self.connect_for_S16(f1))       # the same module twice (Q12)
        n = s16.size(1)                                              # PartAdd :85-96
        s16 = _store(torch.cat((s16 + up[:, :n], up[:, n:]), 1))
        return out0, self.yolo_headS16(s16)

    def forward(self, x, targets=None):
        self.img_size = [x.size(2), x.size(3)]
        out0, out1 = self.heads(x)
        if targets is not None:
            return tuple(yolo_ref.loss_forward(o, targets, s, self.img_size) for o, s in zip((out0, out1), self.specs))
        rows = tuple(yolo_ref.decode_rows(o, s, self.img_size) for o, s in zip((out0, out1), self.specs))
        return nms_ref.nms_driver(rows, self.num_classes)
