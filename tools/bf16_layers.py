"""Layer-by-layer drift of the bf16-storage plan against the fp32 plan (MI355X)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import procedural

arch, size, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])


def run(dtype):
    from mobilenet_yolo_pytorch_amd import mbv3, yolo
    cls = mbv3.yolo if arch == "mbv3" else yolo
    torch.manual_seed(0)
    m = cls(procedural.VOC_CONFIG, sync_metrics=True, act_dtype=dtype)
    procedural.fill_state_dict_(m)
    m = m.cuda().train()
    x = procedural.images(n, size, size, seed=5).cuda()
    tg = procedural.targets(n, seed=6, empty_every=0)
    res = m(x, tg)
    (res[0][0] + res[1][0]).backward()
    torch.cuda.synchronize()
    key = (n, size, size, True) if dtype == torch.float32 else (n, size, size, True, "bf16")
    return m, m._plans[key]


m32, p32 = run(torch.float32)
m16, p16 = run(torch.bfloat16)
g = m32.graph
for nd in g.nodes:
    o = nd.out
    if o.id in p32.units:
        a, b = p32.units[o.id], p16.units[o.id]
        ya, yb = a.Y.float(), b.Y.float()
        e = (ya - yb).abs().max().item() / (ya.abs().max().item() + 1e-12)
        rms = ((ya - yb).pow(2).mean().sqrt() / (ya.pow(2).mean().sqrt() + 1e-12)).item()
        es = (a.scale - b.scale).abs().max().item() / (a.scale.abs().max().item() + 1e-12)
        print("%-5s %-38s C%-4d M%-7d Y max-rel %.2e rms-rel %.2e  scale rel %.2e" % (nd.op, o.name[-38:], o.C, a.M, e, rms, es))
    elif o.id in p32.reals:
        ya, yb = p32.reals[o.id].float(), p16.reals[o.id].float()
        rms = ((ya - yb).pow(2).mean().sqrt() / (ya.pow(2).mean().sqrt() + 1e-12)).item()
        print("%-5s %-38s C%-4d rms-rel %.2e" % (nd.op, o.name[-38:], o.C, rms))
