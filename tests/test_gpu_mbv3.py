"""MI355X: MobileNetV3-YOLO ops and whole-network parity (fp32) vs torch-CPU references and the real reference's fixture."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import net_ref_v3, procedural

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.cpu().permute(0, 3, 1, 2).contiguous()


def rnd(*s, seed=0):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed))


def test_gate_multiply_and_partadd_kernels():
    import ctypes
    from mobilenet_yolo_pytorch_amd import _lib, ops
    N, H, W, C = 2, 6, 8, 40
    a, b, g = rnd(N, C, H, W, seed=1), rnd(N, C, H, W, seed=2), rnd(N, C, H, W, seed=3)
    sc, sh = 1 + 0.2 * rnd(C, seed=4), 0.5 * rnd(C, seed=5)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    zb = br * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    out = ar * (F.relu6(zb + 3) / 6)
    out.backward(g)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    ad, bd, gd = nhwc(a), nhwc(b), nhwc(g)
    scd, shd = sc.cuda(), sh.cuda()
    o = torch.empty_like(ad)
    M = N * H * W
    _lib.call("mny_mul_views", p(ad), None, None, 0, p(bd), p(scd), p(shd), _lib.ACT_HSIGMOID, p(o), M, C, st)
    torch.testing.assert_close(nchw(o), out.detach(), rtol=1e-5, atol=1e-6)
    ga = torch.empty_like(ad)
    _lib.call("mny_mul_views_bwd", p(gd), p(bd), p(scd), p(shd), _lib.ACT_HSIGMOID, None, p(ga), M, C, st)
    torch.testing.assert_close(nchw(ga), ar.grad, rtol=1e-5, atol=1e-6)
    # gradient wrt the gate's pre-activation = g * a * hsigmoid'(z): mul_views_bwd gives g*a, the BN-bwd kernels apply act'
    gb = torch.empty_like(ad)
    _lib.call("mny_mul_views_bwd", p(gd), p(ad), None, None, 0, None, p(gb), M, C, st)
    dy = torch.empty_like(ad)
    _lib.call("mny_bn_bwd_apply", p(gb), p(bd), p(scd), p(shd), _lib.ACT_HSIGMOID, None, p(dy), M, C, st)
    torch.testing.assert_close(nchw(dy) * sc.view(1, -1, 1, 1), br.grad, rtol=1e-5, atol=1e-6)
    # PartAdd
    Ca, Cb = 16, 40
    x, up = rnd(N, Ca, H, W, seed=6), rnd(N, Cb, H // 2, W // 2, seed=7)
    u2 = F.interpolate(up, scale_factor=2, mode="nearest")
    ref = torch.cat((x + u2[:, :Ca], u2[:, Ca:]), 1)
    xd, upd = nhwc(x), nhwc(up)
    o2 = torch.empty(N, H, W, Cb, device="cuda")
    _lib.call("mny_partadd_up", p(xd), None, None, 0, p(upd), p(o2), N, H, W, Ca, Cb, st)
    torch.testing.assert_close(nchw(o2), ref, rtol=1e-6, atol=1e-6)
    gg = nhwc(rnd(N, Cb, H, W, seed=8))
    gx = torch.ones(N, H, W, Ca, device="cuda")
    _lib.call("mny_slice_channels", p(gg), p(gx), 1, M, Ca, Cb, st)
    torch.testing.assert_close(gx, gg[..., :Ca] + 1, rtol=1e-6, atol=1e-6)


def test_bn_backward_with_10_channels():
    from mobilenet_yolo_pytorch_amd import ops
    N, H, W, C = 3, 5, 7, 10
    y = rnd(N, C, H, W, seed=1)
    gamma, beta = 1 + 0.3 * rnd(C, seed=2), 0.2 * rnd(C, seed=3)
    yr, gr, br = y.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a = F.relu(F.batch_norm(yr, None, None, gr, br, True, 0.1, 1e-5))
    g = rnd(*a.shape, seed=4)
    a.backward(g)
    yd = nhwc(y)
    M = N * H * W
    st = torch.stack((yd.view(M, C).sum(0), (yd.view(M, C) ** 2).sum(0))).view(1, 2, C).contiguous()
    scale, shift, mean, invstd = ops.bn_finalize(st, M, gamma.cuda(), beta.cuda())
    dy, dgamma, dbeta = ops.bn_backward(nhwc(g), yd, scale, shift, 3, gamma.cuda(), mean, invstd)
    torch.testing.assert_close(nchw(dy), yr.grad, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(dgamma.cpu(), gr.grad, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(dbeta.cpu(), br.grad, rtol=2e-4, atol=2e-4)


def _close(got, ref, rel, what):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = np.abs(ref).max() + 1e-12
    assert np.abs(got - ref).max() <= rel * scale, "%s: rel err %.2e" % (what, np.abs(got - ref).max() / scale)


def _model(train):
    from mobilenet_yolo_pytorch_amd import mbv3
    torch.manual_seed(0)
    m = mbv3.yolo(procedural.VOC_CONFIG, sync_metrics=True)
    procedural.fill_state_dict_(m)
    m = m.cuda()
    return m.train() if train else m.eval()


def test_mbv3_eval_and_train_match_reference_fixture():
    """tolerances as in test_gpu_net.py (fp32, ~80 conv layers with train-mode BN)."""
    z = np.load(os.path.join(G, "net_v3.npz"))
    names = json.load(open(os.path.join(G, "net_v3_names.json")))
    m = _model(False)
    for hs in m.yolo_losses:
        hs.val_conf = 0.3
    det = m(procedural.images(2, 128, 128, seed=20).cuda())
    plan = m._plans[(2, 128, 128, False)]
    _close(plan.heads[0].permute(0, 3, 1, 2).cpu().numpy(), z["ev_out0"], 2e-3, "eval out0")
    _close(plan.heads[1].permute(0, 3, 1, 2).cpu().numpy(), z["ev_out1"], 2e-3, "eval out1")
    assert len(det) == 2
    m = _model(True)
    tg = list(torch.split(torch.from_numpy(z["t_all"]), z["t_counts"].tolist()))
    res = m(procedural.images(2, 128, 128, seed=21).cuda(), tg)
    (res[0][0] + res[1][0]).backward()
    plan = m._plans[(2, 128, 128, True)]
    _close(plan.heads[0].permute(0, 3, 1, 2).cpu().numpy(), z["tr_out0"], 3e-3, "train out0")
    _close(plan.heads[1].permute(0, 3, 1, 2).cpu().numpy(), z["tr_out1"], 3e-3, "train out1")
    for i in range(2):
        np.testing.assert_allclose(np.array([float(v) for v in res[i]]), z["tuple%d" % i], rtol=3e-3, atol=1e-5)
    gp = dict(m.named_parameters())
    gn = np.array([gp[k].grad.double().norm().item() for k in names["params"]])
    bad = [(k, a, b) for k, a, b in zip(names["params"], gn, z["gnorm"]) if abs(a - b) > 2e-2 * abs(b) + 5e-5]
    assert not bad, bad[:6]
    _close(gp["connect_for_S16.conv.0.conv.weight"].grad.cpu().numpy(), z["g_shared_dw"], 3e-2, "shared dw grad (Q12)")
    _close(gp["backbone.bneck.3.se.se.3.weight"].grad.cpu().numpy(), z["g_gate"], 3e-2, "gate conv grad")
    sd = m.state_dict()
    rs = np.array([sd[k].double().norm().item() for k in names["running"]])
    np.testing.assert_allclose(rs, z["rs_norm"], rtol=2e-4)


def test_mbv3_512_matches_oracle():
    ref = procedural.fill_state_dict_(net_ref_v3.RefYoloV3(procedural.VOC_CONFIG)).train()
    m = _model(True)
    x = procedural.images(2, 512, 512, seed=5)
    tg = procedural.targets(2, seed=6, empty_every=0)
    rr = ref(x, tg)
    (rr[0][0] + rr[1][0]).backward()
    res = m(x.cuda(), tg)
    (res[0][0] + res[1][0]).backward()
    for i in range(2):
        np.testing.assert_allclose(np.array([float(v) for v in res[i]]), np.array([float(v) for v in rr[i]]), rtol=3e-3, atol=1e-5)
    rp = dict(ref.named_parameters())
    for k, p in m.named_parameters():
        a, b = p.grad.double().norm().item(), rp[k].grad.double().norm().item()
        assert abs(a - b) <= 3e-2 * b + 5e-5, (k, a, b)


def test_mbv3_eval_mode_losses_are_differentiable_frozen_batchnorm():
    """models/mbv3_yolo.py under model.eval() with gradients (frozen BatchNorm: running statistics, constants of the step) — the
    shared `connect_for_S16` module (two contributions), the squeeze-excite gates and the h-swish units all through the generic
    backward kernels with mny_bn_bwd_finalize_frozen; losses and every gradient norm against the oracle in .eval()."""
    ref = procedural.fill_state_dict_(net_ref_v3.RefYoloV3(procedural.VOC_CONFIG)).train()
    x = procedural.images(2, 256, 256, seed=7)
    tg = procedural.targets(2, seed=8, empty_every=0)
    # running statistics that fit the weights (the procedural ones send the eval-mode activations to inf): one train-mode pass at momentum 1
    bns = [mod for mod in ref.modules() if isinstance(mod, torch.nn.BatchNorm2d)]
    for mod in bns:
        mod.momentum = 1.0
    with torch.no_grad():
        ref(procedural.images(4, 256, 256, seed=9), procedural.targets(4, seed=10, empty_every=0))
    for mod in bns:
        mod.momentum = 0.1
    ref.eval()
    m = _model(False)
    m.load_state_dict(ref.state_dict())
    before = {k: v.clone() for k, v in m.state_dict().items()}
    rr = ref(x, tg)
    (rr[0][0] + rr[1][0]).backward()
    res = m(x.cuda(), tg)
    (res[0][0] + res[1][0]).backward()
    for i in range(2):
        np.testing.assert_allclose(np.array([float(torch.as_tensor(v).detach()) for v in res[i]]), np.array([float(torch.as_tensor(v).detach()) for v in rr[i]]), rtol=3e-3, atol=1e-5)
    rp = dict(ref.named_parameters())
    n_cmp = 0
    for k, p in m.named_parameters():
        if rp[k].grad is None:
            assert p.grad is None, k
            continue
        a, b = p.grad.double().norm().item(), rp[k].grad.double().norm().item()
        assert abs(a - b) <= 3e-2 * b + 5e-5, (k, a, b)
        n_cmp += 1
    assert n_cmp > 200
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k


def test_recorded_kernel_routes_are_what_the_dispatchers_did():
    """NetPlan.kernel_routes() reports the family each pointwise-conv call's dispatcher TOOK at the plan's first replay
    (mny_pw_last_route), not a prediction: MobileNetV3's weight gradients over h-swish views run the register-staged kernel in fp32
    (pw_wgrad_impl keeps them off the LDS-DMA kernels) although the plain-view predictor mny_pw_route says otherwise."""
    from mobilenet_yolo_pytorch_amd import _lib
    m = _model(True)
    x = procedural.images(2, 256, 256, seed=3).cuda()
    tg = procedural.targets(2, seed=4, empty_every=0)
    res = m(x, tg)
    (res[0][0] + res[1][0]).backward()
    plan = m._plans[(2, 256, 256, True)]
    assert plan.routes["fwd"] and plan.routes["bwd"]
    rec = plan.kernel_routes()
    n_diff = 0
    for fn, label, (M, K, N), fam in rec:
        op = {"mny_pw_fwd": 0, "mny_pw_wgrad": 2}.get(label, 1)
        n_diff += int(fam != _lib.query("mny_pw_route", op, 0, M, K, N))
        assert 0 <= fam <= 5
    hsw = [nd for nd in m.graph.nodes if nd.op == "pw" and nd.ins[0].act == _lib.ACT_HSWISH]
    assert hsw and n_diff > 0           # at least the h-swish-view weight gradients differ from the plain-view prediction
