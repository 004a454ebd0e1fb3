# whole-step determinism at the headline size: same batch, fresh gradients, N runs -> loss tuple and every gradient bit-identical
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import procedural
from mobilenet_yolo_pytorch_amd import yolo
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
arch = sys.argv[3] if len(sys.argv) > 3 else "mbv2"          # mbv2 | mbv3 | mbv3bf16 (512x512)
size = 352 if arch == "mbv2" else 512
torch.manual_seed(0)
if arch == "mbv2":
    m = procedural.fill_state_dict_(yolo(procedural.VOC_CONFIG, sync_metrics=True)).cuda().train()
else:
    from mobilenet_yolo_pytorch_amd import mbv3, synthetic
    m = mbv3.yolo(synthetic.VOC_CONFIG, act_dtype=torch.bfloat16 if arch == "mbv3bf16" else torch.float32).cuda().train()
x = procedural.images(bs, size, size, seed=5).cuda()
tg = procedural.targets(bs, seed=6, empty_every=5)
first, nbad = None, 0
for it in range(runs):
    m.zero_grad(set_to_none=True)
    junk = torch.randn(1 << 24, device="cuda")
    res = m(x, tg)
    (res[0][0] + res[1][0]).backward()
    torch.cuda.synchronize()
    del junk
    cur = ([float(v.detach()) if torch.is_tensor(v) else float(v) for r in res for v in r], {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    if first is None:
        first = cur
        continue
    bad = [k for k in first[1] if not torch.equal(cur[1][k], first[1][k])]
    if bad or cur[0] != first[0]:
        nbad += 1
        print("run %d differs: loss equal %s, %d gradients differ, e.g. %s" % (it, cur[0] == first[0], len(bad), bad[:4]))
print("%s bs=%d: %d of %d repeat runs differ from the first" % (arch, bs, nbad, runs - 1))
