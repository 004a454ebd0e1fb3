"""Print a plan's call lists in launch order (entry point, shape, kernel family): usage  python tools/dump_plan.py [mbv2|mbv3] [bs] [size] [f32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobilenet_yolo_pytorch_amd import mbv3, synthetic, yolo

arch = sys.argv[1] if len(sys.argv) > 1 else "mbv2"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
size = int(sys.argv[3]) if len(sys.argv) > 3 else 352
dt = torch.bfloat16 if (len(sys.argv) > 4 and sys.argv[4] == "bf16") else torch.float32
dev = torch.device("cuda:0")
model = (yolo if arch == "mbv2" else mbv3.yolo)(synthetic.VOC_CONFIG, act_dtype=dt).to(dev).train()
out = model(synthetic.images(bs, size, size, seed=0).to(dev), synthetic.targets(bs, seed=1, empty_every=16))
(out[0][0] + out[1][0]).backward()
torch.cuda.synchronize()
plan = list(model._plans.values())[0]
fam = {0: "tile-v1", 1: "dma", 2: "dma-x6", 3: "thin", 4: "wide", 5: "wgrad-stream"}
routes = {(label, shape): f for _fn, label, shape, f in plan.kernel_routes()}
for which, calls in (("fwd", plan.fwd.calls), ("bwd", plan.bwd.calls)):
    for i, (fn, _a, label, meta) in enumerate(calls):
        shape = (meta or {}).get("shape", "")
        import re
        m = re.search(r"M(\d+) K(\d+) N(\d+)", shape)
        r = fam.get(routes.get((label, tuple(int(v) for v in m.groups())))) if m else None
        print("%s %3d %-26s %-34s %s" % (which, i, getattr(fn, "__name__", label), shape, r or ""))
