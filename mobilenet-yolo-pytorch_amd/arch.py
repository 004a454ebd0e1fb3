"""Network topology of MobileNetV2-YOLO as a small static graph over HIP ops.

A graph *value* is either
  - a ``unit``  : raw conv output Y followed by BatchNorm + activation that are never materialised —
                  consumers read it through a view (Y, scale, shift, act);
  - a ``real``  : a materialised NHWC tensor (residual sums, biased head outputs);
  - the network ``input`` (NCHW).
Nodes: stem | dw | pw (conv+BN+act units), pwb (biased 1x1, no BN), add (a + b [+ upsample2x(up)]).

The module paths attached to every node reproduce the reference's ``state_dict`` keys:
backbone = models/mobilenetv2.py:94-158, neck/heads = models/mbv2_yolo.py:105-135.
"""
from ._lib import ACT_HSIGMOID, ACT_HSWISH, ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_RELU6


class Value:
    __slots__ = ("id", "kind", "C", "down", "act", "node", "name")

    def __init__(self, id, kind, C, down, act=ACT_NONE, node=None, name=""):
        self.id, self.kind, self.C, self.down, self.act, self.node, self.name = id, kind, C, down, act, node, name


class Node:
    __slots__ = ("op", "ins", "out", "conv", "bn", "k", "stride", "bias")

    def __init__(self, op, ins, out, conv=None, bn=None, k=1, stride=1, bias=False):
        self.op, self.ins, self.out, self.conv, self.bn, self.k, self.stride, self.bias = op, ins, out, conv, bn, k, stride, bias


class Graph:
    def __init__(self):
        self.values, self.nodes = [], []
        self.modules = []          # (path, kind, args) in forward order
        self.input = self._val("input", 3, 1, name="x")
        self.outputs = []
        self.seg_out = None        # raw seg head (BDD100K config), a third loss input

    def _val(self, kind, C, down, act=ACT_NONE, name=""):
        v = Value(len(self.values), kind, C, down, act, name=name)
        self.values.append(v)
        return v

    def _node(self, op, ins, out, **kw):
        n = Node(op, ins, out, **kw)
        out.node = n
        self.nodes.append(n)
        return out

    def _register(self, *mods):
        have = {m[0] for m in self.modules}
        self.modules += [m for m in mods if m[0] not in have]        # a module applied twice is registered once

    def stem(self, x, conv, bn, cout, act):
        self._register((conv, "conv", (3, cout, 3, 2, 1, False)), (bn, "bn", (cout,)))
        return self._node("stem", [x], self._val("unit", cout, x.down * 2, act, conv), conv=conv, bn=bn, k=3, stride=2)

    def dw(self, x, conv, bn, act, k=3, stride=1):
        self._register((conv, "conv", (x.C, x.C, k, stride, x.C, False)), (bn, "bn", (x.C,)))
        return self._node("dw", [x], self._val("unit", x.C, x.down * stride, act, conv), conv=conv, bn=bn, k=k, stride=stride)

    def pw(self, x, conv, bn, cout, act):
        self._register((conv, "conv", (x.C, cout, 1, 1, 1, False)), (bn, "bn", (cout,)))
        return self._node("pw", [x], self._val("unit", cout, x.down, act, conv), conv=conv, bn=bn)

    def pwb(self, x, conv, cout):
        self._register((conv, "conv", (x.C, cout, 1, 1, 1, True)))
        return self._node("pwb", [x], self._val("real", cout, x.down, name=conv), conv=conv, bias=True)

    def add(self, a, b=None, up=None, name="add"):
        ins = [a] + ([b] if b is not None else []) + ([up] if up is not None else [])
        n = self._node("add", ins, self._val("real", a.C, a.down, name=name))
        n.node.k = (1 if b is not None else 0) | (2 if up is not None else 0)   # operand presence bits
        return n


    def mul(self, a, b, name="mul"):
        """a * b, both read through their views (MobileNetV3's per-pixel gate)."""
        assert a.C == b.C and a.down == b.down
        return self._node("mul", [a, b], self._val("real", a.C, a.down, name=name))

    def partadd(self, a, up, name="partadd"):
        """PartAdd(a, upsample2x(up)) with a.C <= up.C (mbv3_yolo.py:85-96,135)."""
        assert a.C <= up.C and up.down == 2 * a.down
        return self._node("partadd", [a, up], self._val("real", up.C, a.down, name=name))


def _inverted_residual(g, x, path, cout, stride, t):
    """models/mobilenetv2.py:54-91; Sequential indices follow :63-85."""
    hid = round(x.C * t)
    identity = stride == 1 and x.C == cout
    h, i = x, 0
    if t != 1:
        h = g.pw(h, "%s.conv.0" % path, "%s.conv.1" % path, hid, ACT_RELU6)
        i = 3
    h = g.dw(h, "%s.conv.%d" % (path, i), "%s.conv.%d" % (path, i + 1), ACT_RELU6, 3, stride)
    h = g.pw(h, "%s.conv.%d" % (path, i + 3), "%s.conv.%d" % (path, i + 4), cout, ACT_NONE)
    return g.add(x, h, name=path + ".sum") if identity else h


def _basic(g, x, path, cout, depthwise=False):
    """models/mbv2_yolo.py:16-44 BasicConv: conv -> BN -> LeakyReLU(0.1)."""
    if depthwise:
        return g.dw(x, path + ".conv", path + ".bn", ACT_LEAKY, 3, 1)
    return g.pw(x, path + ".conv", path + ".bn", cout, ACT_LEAKY)


def _dw_pw_pw(g, x, path, cout):                       # mbv2_yolo.py:70-76
    h = _basic(g, x, path + ".0", x.C, True)
    h = _basic(g, h, path + ".1", x.C)
    return _basic(g, h, path + ".2", cout)


def _head(g, x, path, mid, cout):                      # mbv2_yolo.py:77-92
    h = _basic(g, x, path + ".0", x.C, True)
    h = _basic(g, h, path + ".1", x.C)
    h = _basic(g, h, path + ".2", mid)
    return g.pwb(h, path + ".3", cout)


def _connect(g, x, path, up=None):                     # mbv2_yolo.py:93-104 (+ the add of :151 when `up`)
    h = _basic(g, x, path + ".conv.0", x.C, True)
    h = _basic(g, h, path + ".conv.1", x.C)
    return g.add(x, h, up=up, name=path + ".sum")


# registration order of the reference's top-level children (mbv2_yolo.py:111-130)
TOP_ORDER = ["seg_headS16", "backbone", "conv_for_S32", "connect_for_S32", "yolo_headS32", "upsample", "conv_for_S16",
             "seg_conv_for_S16", "connect_for_S16", "seg_connect_for_S16", "yolo_headS16"]


def mbv2_yolo_graph(num_classes, num_anchors, seg_classes=None):
    g = Graph()
    out_ch = num_anchors * (5 + num_classes)
    x = g.stem(g.input, "backbone.features.0.0", "backbone.features.0.1", 32, ACT_RELU6)   # mobilenetv2.py:113
    idx = 1
    for t, c, n, s in [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1)]:      # :98-105
        for i in range(n):
            x = _inverted_residual(g, x, "backbone.features.%d" % idx, c, s if i == 0 else 1, t)
            idx += 1
    f1 = x
    idx = 0
    for t, c, n, s in [(6, 160, 3, 2), (6, 320, 1, 1)]:                                              # :106-110
        for i in range(n):
            x = _inverted_residual(g, x, "backbone.features2.%d" % idx, c, s if i == 0 else 1, t)
            idx += 1
    f2 = g.pw(x, "backbone.conv.0", "backbone.conv.1", 1280, ACT_RELU6)                              # :131

    s32 = _basic(g, f2, "conv_for_S32", 512)                                                          # mbv2_yolo.py:142
    s32 = _connect(g, s32, "connect_for_S32")                                                         # :143
    out0 = _head(g, s32, "yolo_headS32", 1024, out_ch)                                                # :144
    s16 = _dw_pw_pw(g, f1, "conv_for_S16", 512)                                                       # :146
    s16 = _connect(g, s16, "connect_for_S16", up=s32)                                                 # :147,:151
    out1 = _head(g, s16, "yolo_headS16", 512, out_ch)                                                 # :153
    seg = _dw_pw_pw(g, f1, "seg_conv_for_S16", 32)                                                    # :155 (always runs, Q10)
    seg = _connect(g, seg, "seg_connect_for_S16")                                                     # :156
    g.seg_out = _head(g, seg, "seg_headS16", 32, seg_classes) if seg_classes is not None else None   # :113,162,168
    g.outputs = [out0, out1]
    g.modules.sort(key=lambda m: TOP_ORDER.index(m[0].split(".")[0]))   # stable: forward order inside a child
    return g


# ---- MobileNetV3-Large-YOLO (models/mobilenetv3.py:77-136, models/mbv3_yolo.py:97-145) -------------------------
TOP_ORDER_V3 = ["backbone", "conv_for_S32", "connect_for_S32", "yolo_headS32", "upsample", "connect_for_S16", "yolo_headS16"]


def _v3_block(g, x, path, k, exp, cout, nl, se, stride):
    """mobilenetv3.py:44-74.  The "SE" module gates PER PIXEL (its avg_pool is never called, :40-41)."""
    h = g.pw(x, path + ".conv1", path + ".bn1", exp, nl)
    h = g.dw(h, path + ".conv2", path + ".bn2", nl, k, stride)
    h = g.pw(h, path + ".conv3", path + ".bn3", cout, ACT_NONE)
    if se:
        s = g.pw(h, path + ".se.se.0", path + ".se.se.1", cout // 4, ACT_RELU)          # :31-34
        s = g.pw(s, path + ".se.se.3", path + ".se.se.4", cout, ACT_HSIGMOID)           # :35-37
        h = g.mul(h, s, name=path + ".gate")
    if stride == 1:                                                                      # :72
        sc = x if x.C == cout else g.pw(x, path + ".shortcut.0", path + ".shortcut.1", cout, ACT_NONE)   # :61-65
        h = g.add(h, sc, name=path + ".sum")
    return h


def mbv3_yolo_graph(num_classes, num_anchors):
    g = Graph()
    out_ch = num_anchors * (5 + num_classes)
    R, H = ACT_RELU, ACT_HSWISH
    x = g.stem(g.input, "backbone.conv1", "backbone.bn1", 16, H)                         # mobilenetv3.py:80-82
    cfg1 = [(3, 16, 16, R, False, 1), (3, 64, 24, R, False, 2), (3, 72, 24, R, False, 1), (5, 72, 40, R, True, 2),
            (5, 120, 40, R, True, 1), (5, 120, 40, R, True, 1), (3, 240, 80, H, False, 2), (3, 200, 80, H, False, 1),
            (3, 184, 80, H, False, 1), (3, 184, 80, H, False, 1), (3, 480, 112, H, True, 1), (3, 672, 112, H, True, 1),
            (5, 672, 160, H, True, 1)]                                                   # :84-98
    for i, (k, e, c, nl, se, s) in enumerate(cfg1):
        x = _v3_block(g, x, "backbone.bneck.%d" % i, k, e, c, nl, se, s)
    f1 = x
    for i, (k, e, c, nl, se, s) in enumerate([(5, 672, 160, H, True, 2), (5, 960, 160, H, True, 1)]):   # :99-102
        x = _v3_block(g, x, "backbone.bneck2.%d" % i, k, e, c, nl, se, s)
    f2 = g.pw(x, "backbone.conv2", "backbone.bn2", 960, H)                               # :104-106

    s32 = _dw_pw_pw(g, f2, "conv_for_S32", 320)                                          # mbv3_yolo.py:106,127
    s32 = _connect(g, s32, "connect_for_S32")                                            # :128
    out0 = _head(g, s32, "yolo_headS32", 960, out_ch)                                    # :129
    s16 = _connect(g, f1, "connect_for_S16")                                             # :133
    s16 = _connect(g, s16, "connect_for_S16")                                            # :134 the SAME module again (Q12)
    s16 = g.partadd(s16, s32, name="S16.partadd")                                        # :135
    out1 = _head(g, s16, "yolo_headS16", 640, out_ch)                                    # :138
    g.outputs = [out0, out1]
    g.modules.sort(key=lambda m: TOP_ORDER_V3.index(m[0].split(".")[0]))
    return g
