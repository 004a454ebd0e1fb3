#!/usr/bin/env python3
"""configs[3] integration parity, tensor by tensor: MobileNetV3-YOLO 512x512, reference (default) init, train step at bs 16.
For every parameter gradient: product with bf16 storage / product with fp32 storage / the oracle's bf16-storage model, each against the FP32
oracle as  ||g - g_ref|| / ||g_ref||  — the measurement behind test_mbv3_512_bf16_default_init_train_step_all_tensors (tests/test_gpu_bf16.py).
    python tools/bf16_grad_table.py [bs] [proc]        (proc: procedural weights instead of the reference init)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import bf16_storage, net_ref_v3, procedural  # noqa: E402


def main():
    from mobilenet_yolo_pytorch_amd import mbv3
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    proc = len(sys.argv) > 2 and sys.argv[2] == "proc"
    S = 512
    torch.manual_seed(0)
    ref = net_ref_v3.RefYoloV3(procedural.VOC_CONFIG).train()
    if proc:
        procedural.fill_state_dict_(ref)
    x = procedural.images(bs, S, S, seed=25)
    tg = procedural.targets(bs, seed=26, empty_every=8)
    grads = {}
    plan16 = None
    for tag, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        m = mbv3.yolo(procedural.VOC_CONFIG, sync_metrics=True, act_dtype=dt)
        m.load_state_dict(ref.state_dict())
        m = m.cuda().train()
        res = m(x.cuda(), tg)
        (res[0][0] + res[1][0]).backward()
        grads[tag] = {k: p.grad.double().cpu().flatten() for k, p in m.named_parameters() if p.grad is not None}
        if tag == "bf16":
            plan16 = m._plans[(bs, S, S, True, "bf16")]
            args = dict(gate_fused=bool(plan16.gates), absorbed={g["mul"].out.name[:-len(".gate")] for g in plan16.gates.values() if g["add"] is not None})
        del m
    rf = ref(x, tg)
    (rf[0][0] + rf[1][0]).backward()
    grads["ref"] = {k: p.grad.double().flatten().clone() for k, p in ref.named_parameters() if p.grad is not None}
    ref.zero_grad(set_to_none=True)
    with bf16_storage.bf16_storage(ref, **args):
        rs = ref(x, tg)
        (rs[0][0] + rs[1][0]).backward()
    grads["model"] = {k: p.grad.double().flatten().clone() for k, p in ref.named_parameters() if p.grad is not None}
    gmax = max(v.norm().item() for v in grads["ref"].values())
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-30)).item()      # noqa: E731
    rows = []
    for k, b in grads["ref"].items():
        rows.append((rel(grads["bf16"][k], b), rel(grads["f32"][k], b), rel(grads["model"][k], b), rel(grads["bf16"][k], grads["model"][k]), b.norm().item() / gmax, k))
    rows.sort(reverse=True)
    print("%-44s %10s %10s %10s %12s %10s" % ("tensor", "bf16~ref", "f32~ref", "model~ref", "bf16~model", "|g|/max"))
    for r in rows[:40]:
        print("%-44s %10.4f %10.2e %10.4f %12.4f %10.2e" % (r[5].replace("backbone.", ""), r[0], r[1], r[2], r[3], r[4]))
    sig = [r for r in rows if r[4] >= 1e-3]
    import statistics
    print("significant tensors: %d; median bf16~ref %.4f, median model~ref %.4f, worst f32~ref %.2e" % (
        len(sig), statistics.median(r[0] for r in sig), statistics.median(r[2] for r in sig), max(r[1] for r in sig)))
    conv = [r for r in sig if r[5].endswith("conv.weight") or ".conv" in r[5] and r[5].endswith("weight") and "bn" not in r[5]]
    print("conv weights among them: %d; worst bf16~ref %.4f, worst model~ref %.4f" % (len(conv), max(r[0] for r in conv), max(r[2] for r in conv)))


if __name__ == "__main__":
    main()
