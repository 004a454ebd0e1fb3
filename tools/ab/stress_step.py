# whole-step determinism at the headline size: same batch, fresh gradients, N runs -> loss tuple and every gradient bit-identical
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import procedural
from mobilenet_yolo_pytorch_amd import yolo
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.manual_seed(0)
m = procedural.fill_state_dict_(yolo(procedural.VOC_CONFIG, sync_metrics=True)).cuda().train()
x = procedural.images(bs, 352, 352, seed=5).cuda()
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
print("bs=%d: %d of %d repeat runs differ from the first" % (bs, nbad, runs - 1))
