"""Compare the bf16-storage plan against the fp32 plan of the same model (MI355X)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import procedural


def run(arch, dtype, size, n=2):
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
    plan = m._plans[key]
    heads = [h.float().cpu().numpy() for h in plan.heads]
    grads = {k: p.grad.float().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}
    return heads, [[float(v) for v in r] for r in res], grads


for arch, size in (("mbv3", 128), ("mbv2", 96)):
    h32, r32, g32 = run(arch, torch.float32, size)
    h16, r16, g16 = run(arch, torch.bfloat16, size)
    for i in range(2):
        e = np.abs(h16[i] - h32[i]).max() / (np.abs(h32[i]).max() + 1e-12)
        print(arch, "head", i, "rel-to-max err %.3e" % e)
        print(arch, "tuple", i, np.round(r32[i], 4), np.round(r16[i], 4))
    cos = []
    for k in g32:
        a, b = g32[k].ravel().astype(np.float64), g16[k].ravel().astype(np.float64)
        na, nb = np.linalg.norm(a), np.linalg.norm(b)
        if na < 1e-9:
            continue
        cos.append((float(a @ b / (na * nb + 1e-30)), nb / na, k))
    cos.sort()
    print(arch, "worst cosine:", cos[:6])
    print(arch, "median cosine %.5f" % np.median([c[0] for c in cos]), "n", len(cos))
