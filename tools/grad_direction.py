#!/usr/bin/env python3
"""How far is every parameter gradient of a train step from the CPU oracle's, as a TENSOR (not only its norm)?
    python tools/grad_direction.py [bs ...]      (default 8 64; 256 needs ~60 GB of host memory)
Prints, per batch size, the worst and the five largest  ||g - g_ref|| / ||g_ref||  and  1 - cos(g, g_ref)  over the 202 tensors —
the measurement behind the direction bounds of tests/test_gpu_net.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import net_ref, procedural  # noqa: E402


def main():
    from mobilenet_yolo_pytorch_amd import yolo
    sizes = [int(v) for v in sys.argv[1:]] or [8, 64]
    for bs in sizes:
        torch.manual_seed(0)
        m = yolo(procedural.VOC_CONFIG)
        procedural.fill_state_dict_(m)
        m = m.cuda().train()
        ref = procedural.fill_state_dict_(net_ref.RefYolo(procedural.VOC_CONFIG)).train()
        x = procedural.images(bs, 352, 352, seed=5 if bs >= 64 else 3)
        tg = procedural.targets(bs, seed=6 if bs >= 64 else 4, empty_every=5 if bs >= 64 else 4)
        res = m(x.cuda(), tg)
        (res[0][0] + res[1][0]).backward()
        rr = ref(x, tg)
        (rr[0][0] + rr[1][0]).backward()
        rp = dict(ref.named_parameters())
        rows = []
        for k, p in m.named_parameters():
            if rp[k].grad is None:
                continue
            a, b = p.grad.double().cpu().flatten(), rp[k].grad.double().flatten()
            nb = b.norm().item()
            rel = (a - b).norm().item() / (nb + 1e-30)
            cos = (a @ b).item() / (a.norm().item() * nb + 1e-30)
            rows.append((rel, 1.0 - cos, nb, k))
        rows.sort(reverse=True)
        print("bs %d: %d tensors; worst rel diff %.3e, worst 1-cos %.3e" % (bs, len(rows), rows[0][0], max(r[1] for r in rows)))
        for rel, omc, nb, k in rows[:8]:
            print("   %-50s rel %.3e  1-cos %.3e  ||g_ref|| %.3e" % (k, rel, omc, nb))
        big = [r for r in rows if r[2] > 1e-4]
        print("   tensors with ||g_ref|| > 1e-4: worst rel %.3e (%s)" % (max(r[0] for r in big), max(big)[3]))
        del m, ref


if __name__ == "__main__":
    main()
