"""MI355X: fused multi-tensor AdamW (mny_adamw_step) against torch.optim.AdamW — the optimizer train.py:134 constructs.
fp32; tolerance 3e-6 relative (+1e-7): the kernel multiplies by 1/sqrt(1-b2^t) where torch divides, otherwise the same update."""
import pytest
import torch

from oracle import procedural

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(32, 3, 3, 3), (96,), (75, 512, 1, 1), (1,), (7,), (300, 1000), (1280, 320, 1, 1)]      # odd sizes, > one 64 Ki chunk
    return [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]


def test_fused_adamw_matches_torch():
    from mobilenet_yolo_pytorch_amd.optim import AdamW
    pa, pb = _params(0), _params(0)
    kw = dict(lr=7e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=4e-4)            # train.py:459-462 defaults
    oa, ob = AdamW(pa, **kw), torch.optim.AdamW(pb, **kw)
    g = torch.Generator().manual_seed(1)
    for it in range(6):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 3 and it < 2:
                a.grad = b.grad = None                     # a parameter without gradient is skipped (seg branch, Q10) ...
                continue
            gr = torch.randn(*a.shape, generator=g).cuda() * (10.0 if i == 2 else 1.0)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()                               # ... and joins later with its OWN step count, like torch's per-parameter `step`
        for a, b in zip(pa, pb):
            torch.testing.assert_close(a.detach(), b.detach(), rtol=3e-6, atol=1e-7)
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"][0]["lr"] == sb["param_groups"][0]["lr"]
    for k in sb["state"]:
        torch.testing.assert_close(sa["state"][k]["exp_avg"], sb["state"][k]["exp_avg"], rtol=3e-6, atol=1e-7)
        torch.testing.assert_close(sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"], rtol=3e-6, atol=1e-12)
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"])
    assert sorted({float(v["step"]) for v in sa["state"].values()}) == [4.0, 6.0]      # the late parameter is two steps behind


def test_state_dict_round_trip_with_torch_adamw():
    from mobilenet_yolo_pytorch_amd.optim import AdamW
    pa, pb = _params(2), _params(2)
    oa, ob = torch.optim.AdamW(pa, lr=1e-3), AdamW(pb, lr=1e-3)
    g = torch.Generator().manual_seed(3)
    def grads():
        for a, b in zip(pa, pb):
            gr = torch.randn(*a.shape, generator=g).cuda()
            a.grad, b.grad = gr.clone(), gr.clone()
    grads(); oa.step(); ob.step()
    import copy
    ob.load_state_dict(copy.deepcopy(oa.state_dict()))     # a torch checkpoint loads into the fused optimizer (deepcopy: torch's
    #                                                        load_state_dict keeps references to same-device tensors)
    grads(); oa.step(); ob.step()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=3e-6, atol=1e-7)


def test_fused_adamw_on_the_model_parameter_set():
    """One step on the real module: gradients are views into the plan's flat arena, the seg-branch parameters have none.
    (Multi-step trajectories of two optimizers cannot be compared: Adam's m/sqrt(v) turns 1e-7 differences into sign flips.)"""
    from mobilenet_yolo_pytorch_amd import yolo
    from mobilenet_yolo_pytorch_amd.optim import AdamW
    cfg = dict(procedural.VOC_CONFIG)
    torch.manual_seed(0)
    m = yolo(cfg).cuda().train()
    x = procedural.images(4, 160, 160, seed=3).cuda()
    tg = procedural.targets(4, seed=4, empty_every=0)
    res = m(x, tg)
    (res[0][0] + res[1][0]).backward()
    ref = [torch.nn.Parameter(p.detach().clone()) for p in m.parameters()]
    for r, p in zip(ref, m.parameters()):
        r.grad = None if p.grad is None else p.grad.detach().clone()
    kw = dict(lr=7e-4, weight_decay=4e-4)
    fused, stock = AdamW(m.parameters(), **kw), torch.optim.AdamW(ref, **kw)
    for _ in range(3):                                      # same gradients three times: exercises the bias corrections
        fused.step(); stock.step()
    n_live = 0
    for r, p in zip(ref, m.parameters()):
        torch.testing.assert_close(p.detach(), r.detach(), rtol=3e-6, atol=2e-7)
        n_live += p.grad is not None
    assert n_live >= 200
    res2 = m(x, tg)                                         # the plan keeps working on the updated parameters
    assert float(res2[0][0] + res2[1][0]) < float(res[0][0] + res[1][0])
