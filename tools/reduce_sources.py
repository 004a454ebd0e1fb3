"""For every separate bn_bwd_reduce / bn_bwd_apply launch of the headline plan: which backward call(s) produced the unit's output
gradient G (run on the GPU box).  Decides where a BN-sum epilogue / a rebuild-on-load fusion would pay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobilenet_yolo_pytorch_amd import synthetic, yolo
m = yolo(synthetic.VOC_CONFIG).cuda().train()
x = synthetic.images(8, 352, 352).cuda()
out = m(x, synthetic.targets(8))
(out[0][0] + out[1][0]).backward()
plan = m._plans[(8, 352, 352, True)]
calls = plan.bwd.calls
writers = {}
for ci, (fn, args, name, meta) in enumerate(calls):
    ptrs = [getattr(a, "value", None) for a in args]
    if name == "mny_bn_bwd_reduce":
        g = ptrs[0]
        srcs = [(cj, calls[cj][2], (calls[cj][3] or {}).get("shape", "")) for cj in range(ci) if g in [getattr(a, "value", None) for a in calls[cj][1]][2:]]
        print("reduce", (meta or {}).get("shape"), "<-", [(n, s) for _c, n, s in srcs[-3:]])
    if name == "mny_bn_bwd_apply":
        print("apply ", (meta or {}).get("shape"), "-> next:", [calls[cj][2] for cj in range(ci + 1, min(ci + 4, len(calls)))])
