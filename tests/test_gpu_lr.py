"""GPU: the low-rank BatchNorm backward of a wide expand unit (csrc/lrbwd.hip + mny_pw_lr_fix + mny_dw_bnbwd_red_dz) — autograd of
nn.Conv2d(K, C, 1) + nn.BatchNorm2d + ReLU6 (models/mobilenetv2.py:73-78) without the bn_bwd_apply pass — against torch autograd in fp64
on the CPU and against the un-fused chain (mny_bn_bwd_apply + mny_pw_wgrad + mny_pw_fwd data gradient) it replaces."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

from mobilenet_yolo_pytorch_amd import _lib  # noqa: E402

P = ctypes.c_void_p


def ptr(t):
    return P(t.data_ptr()) if t is not None else None


def stream():
    return P(torch.cuda.current_stream().cuda_stream)


def _unit(M, K, C, seed, view, act):
    """fp64 reference of one expand unit: returns inputs and autograd results."""
    g = torch.Generator().manual_seed(seed)
    xr = (torch.randn(M, K, generator=g) * 1.5 + 0.3).float()
    xs = (torch.rand(K, generator=g) + 0.5).float() if view else None
    xt = (torch.randn(K, generator=g) * 0.4).float() if view else None
    W = (torch.randn(C, K, generator=g) * K ** -0.5).float()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).float(), (torch.randn(C, generator=g) * 0.5 + 1.0).float()
    G = torch.randn(M, C, generator=g).float()
    xv = (xr.double() * xs.double() + xt.double()) if view else xr.double()
    xv.requires_grad_(True)
    Wd = W.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = xv @ Wd.t()
    mean, var = y.mean(0), y.var(0, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    z = (y - mean) * invstd * gd + bd
    if act == _lib.ACT_RELU6:
        a = z.clamp(0, 6)
    elif act == _lib.ACT_HSWISH:
        a = z * (z + 3).clamp(0, 6) / 6
    else:
        a = z
    (a * G.double()).sum().backward()
    scale = (gd * invstd).detach()
    shift = (bd - mean * scale).detach()
    zz = z.detach()
    if act == _lib.ACT_RELU6:
        dact = ((zz > 0) & (zz < 6)).double()
    elif act == _lib.ACT_HSWISH:
        dact = torch.where(zz <= -3, torch.zeros_like(zz), torch.where(zz >= 3, torch.ones_like(zz), (2 * zz + 3) / 6))
    else:
        dact = torch.ones_like(zz)
    dz = G.double() * dact
    return dict(xr=xr, xs=xs, xt=xt, W=W, gamma=gamma, G=G, y=y.detach(), mean=mean.detach(), invstd=invstd.detach(), scale=scale, shift=shift, dz=dz,
                dW=Wd.grad, dX=xv.grad, dgamma=gd.grad, dbeta=bd.grad)


@pytest.mark.parametrize("M,K,C,view,act", [(3000, 64, 384, True, 1), (2 * 22 * 22 + 5, 96, 576, False, 1), (1500, 160, 960, True, 1), (777, 320, 1280, True, 1),
                                            (4096, 24, 72, True, 1), (2500, 40, 120, False, 4), (1300, 112, 672, True, 4), (70000, 64, 384, True, 1),
                                            (32 * 300, 64, 384, True, 1), (32 * 280, 96, 576, False, 1), (32 * 260, 96, 576, True, 1)])
def test_low_rank_bn_backward_matches_autograd(M, K, C, view, act):
    dev = torch.device("cuda:0")
    act = {1: _lib.ACT_RELU6, 4: _lib.ACT_HSWISH}[act]
    assert _lib.query("mny_lr_supported", M, K, C) == 1
    r = _unit(M, K, C, seed=M + K, view=view, act=act)
    st = stream()
    f = lambda t: t.float().contiguous().to(dev)      # noqa: E731
    xr, W = f(r["xr"]), f(r["W"])
    xs, xt = (f(r["xs"]), f(r["xt"])) if view else (None, None)
    mean, invstd, gamma = f(r["mean"]), f(r["invstd"]), f(r["gamma"])
    # what the depthwise backward in front stores (mny_dw_bnbwd_red_dz) and the partial row of sums it leaves
    dzc = f(r["dz"] * r["scale"])
    yhat = (r["y"] - r["mean"]) * r["invstd"]
    red = f(torch.stack((r["dz"].sum(0), (r["dz"] * yhat).sum(0))).reshape(1, 2, C))
    coef = torch.empty(3, C, device=dev)
    dgam, dbet = torch.empty(C, device=dev), torch.empty(C, device=dev)
    _lib.call("mny_bn_bwd_finalize", ptr(red), 1, M, ptr(gamma), ptr(mean), ptr(invstd), ptr(dgam), ptr(dbet), ptr(coef), C, st)
    # weight gradient: main term, Gram + column sums of the viewed input, in-place correction
    dw = torch.zeros(C, K, device=dev)
    ws = torch.zeros(int(_lib.query("mny_pw_wgrad_ws_floats", M, K, C)) + 16, device=dev)
    _lib.call("mny_pw_wgrad", ptr(xr), ptr(xs), ptr(xt), _lib.ACT_NONE, ptr(dzc), ptr(dw), None, ptr(ws), M, K, C, st)
    gp = _lib.query("mny_lr_gram_parts", M, K)
    gparts = torch.full((gp, K * K + K), float("nan"), device=dev)
    _lib.call("mny_lr_gram", ptr(xr), ptr(xs), ptr(xt), _lib.ACT_NONE, ptr(gparts), M, K, st)
    torch.cuda.synchronize()
    assert torch.isfinite(gparts).all()
    gs = gparts.double().sum(0).float().contiguous()
    xv = (r["xr"].double() * r["xs"].double() + r["xt"].double()) if view else r["xr"].double()
    gram_ref = xv.t() @ xv
    e = (gs[:K * K].double().cpu().reshape(K, K) - gram_ref).abs().max().item()
    assert e <= 2e-5 * gram_ref.abs().max().item(), ("gram", e)
    e = (gs[K * K:].double().cpu() - xv.sum(0)).abs().max().item()
    assert e <= 2e-5 * xv.sum(0).abs().max().item() + 1e-3, ("colsum", e)
    _lib.call("mny_lr_wfix", ptr(dw), ptr(gs), ptr(coef), ptr(W), C, K, st)
    # data gradient: main term on dzc, prep, correction (+ the BN-backward sums of a linear unit whose raw output is xr)
    wT = W.t().contiguous()
    dx = torch.empty(M, K, device=dev)
    _lib.call("mny_pw_fwd", ptr(dzc), None, None, _lib.ACT_NONE, ptr(wT), None, None, ptr(dx), None, M, C, K, st)
    bq, rb = torch.empty(K, K, device=dev), torch.empty(K, device=dev)
    _lib.call("mny_lr_prep", ptr(coef), ptr(W), ptr(bq), ptr(rb), C, K, st)
    pmean, pinv = f(r["xr"].double().mean(0)), f((r["xr"].double().var(0, unbiased=False) + 1e-5).rsqrt())
    ones, zeros = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    psc, psh = (xs, xt) if view else (ones, zeros)
    rp = _lib.query("mny_pw_lr_fix_parts", M, K, _lib.ACT_NONE)
    rbuf = torch.full((rp, 2, K), float("nan"), device=dev)
    _lib.call("mny_pw_lr_fix", ptr(xr), ptr(xs), ptr(xt), ptr(bq), ptr(rb), ptr(dx), ptr(dx), ptr(xr), ptr(psc), ptr(psh), _lib.ACT_NONE, ptr(pmean), ptr(pinv),
              ptr(rbuf), M, K, st)
    q_ref = (r["W"].double().t() * coef[1].double().cpu()) @ r["W"].double()
    assert (bq.double().cpu() - q_ref).abs().max().item() <= 2e-5 * q_ref.abs().max().item() + 1e-7, "Q"
    torch.cuda.synchronize()
    tol = lambda ref: 3e-4 * ref.abs().max().item() + 1e-4      # noqa: E731
    for name, got, ref in (("dgamma", dgam, r["dgamma"]), ("dbeta", dbet, r["dbeta"]), ("dW", dw, r["dW"]), ("dX", dx, r["dX"])):
        assert torch.isfinite(got).all(), name
        e = (got.double().cpu() - ref).abs().max().item()
        assert e <= tol(ref), (name, e, ref.abs().max().item())
    # relative Frobenius error (a permuted tile or a missing term cannot hide in the max norm of a large tensor)
    for name, got, ref in (("dW", dw, r["dW"]), ("dX", dx, r["dX"])):
        rel = ((got.double().cpu() - ref).norm() / ref.norm()).item()
        assert rel <= 2e-4, (name, rel)
    s = rbuf.double().sum(0).cpu()
    xhat = (r["xr"].double() - pmean.double().cpu()) * pinv.double().cpu()
    for name, got, ref in (("s1", s[0], r["dX"].sum(0)), ("s2", s[1], (r["dX"] * xhat).sum(0))):
        e = (got - ref).abs().max().item()
        # (the sums of a BatchNorm input's gradient vanish: the bound is the fp32 summation error of M terms of the gradient's size)
        assert e <= 3e-4 * ref.abs().max().item() + 1e-6 * M * r["dX"].abs().mean().item() + 2e-3, (name, e, ref.abs().max().item())


def test_low_rank_fix_without_reduction_target_and_with_a_separate_addend():
    """mny_pw_lr_fix with red = NULL (the input has other consumers) and addend != dx."""
    dev = torch.device("cuda:0")
    M, K = 5000, 96
    g = torch.Generator().manual_seed(5)
    x, bq, rb, add = (torch.randn(M, K, generator=g), torch.randn(K, K, generator=g) * 0.1, torch.randn(K, generator=g), torch.randn(M, K, generator=g))
    xd, bd, rd, ad = (t.to(dev).contiguous() for t in (x, bq, rb, add))
    out = torch.empty(M, K, device=dev)
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g)
    scd, shd = sc.to(dev), sh.to(dev)
    _lib.call("mny_pw_lr_fix", ptr(xd), ptr(scd), ptr(shd), ptr(bd), ptr(rd), ptr(ad), ptr(out), None, None, None, 0, None, None, None, M, K, stream())
    torch.cuda.synchronize()
    ref = (x.double() * sc.double() + sh.double()) @ bq.double().t() + rb.double() + add.double()
    assert (out.double().cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-5


@pytest.mark.parametrize("C,H,W,in_act", [(384, 22, 22, 1), (96, 30, 17, 1), (576, 11, 11, 1)])
def test_depthwise_backward_stores_the_masked_scaled_gradient(C, H, W, in_act):
    """mny_dw_bnbwd_red_dz == mny_dw_bnbwd_red with dx replaced by in_scale o dx o act'(in_scale x + in_shift); same dw, same producer sums."""
    dev = torch.device("cuda:0")
    N = 3
    assert _lib.query("mny_dw_bnbwd_red_dz_supported", 3, C, 0) == 1
    g = torch.Generator().manual_seed(C + H)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev).contiguous()      # noqa: E731
    G, Y, X = rnd(N, H, W, C), rnd(N, H, W, C), rnd(N, H, W, C) * 2
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
    coef = torch.stack((torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1)).to(dev).contiguous()
    xsc, xsh = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.5 + 1.0).to(dev)
    xmu, xis = (torch.randn(C, generator=g) * 0.2).to(dev), (torch.rand(C, generator=g) + 0.5).to(dev)
    wt = (torch.randn(C, 1, 3, 3, generator=g) / 3).to(dev).contiguous()
    parts = _lib.query("mny_dw_bnbwd_parts_k", N, H, W, C, 3, 2)
    res = {}
    for name in ("mny_dw_bnbwd_red", "mny_dw_bnbwd_red_dz"):
        dx, dw = torch.full((N, H, W, C), float("nan"), device=dev), torch.zeros(C, 1, 3, 3, device=dev)
        ws, red = torch.zeros(parts * C * 9, device=dev), torch.full((parts, 2, C), float("nan"), device=dev)
        _lib.call(name, ptr(G), ptr(Y), ptr(sc), ptr(sh), _lib.ACT_RELU6, ptr(coef), ptr(X), ptr(xsc), ptr(xsh), _lib.ACT_RELU6, ptr(xmu), ptr(xis),
                  ptr(wt), None, ptr(dx), ptr(dw), ptr(ws), ptr(red), N, H, W, C, 3, 1, stream())
        torch.cuda.synchronize()
        res[name] = (dx, dw, red)
    (dx0, dw0, red0), (dx1, dw1, red1) = res["mny_dw_bnbwd_red"], res["mny_dw_bnbwd_red_dz"]
    assert torch.equal(dw0, dw1) and torch.equal(red0, red1)
    z = X * xsc + xsh
    want = dx0 * ((z > 0) & (z < 6)).float() * xsc
    assert torch.isfinite(dx1).all()
    assert (dx1 - want).abs().max().item() <= 1e-6 * want.abs().max().item()


@pytest.mark.parametrize("C,H,W,act", [(576, 22, 22, 1), (96, 31, 17, 1), (192, 12, 12, 1)])
def test_stride2_depthwise_backward_stores_the_masked_scaled_gradient_and_the_producer_sums(C, H, W, act):
    """mny_dw_bnbwd_s2_red_dz == mny_dw_bnbwd_s2 with dx replaced by in_scale o dx o relu6'(in_scale x + in_shift), same dw, and its partial rows
    sum to what mny_bn_bwd_reduce computes over (that dx, x) — odd sizes included (the masked last row / column)."""
    dev = torch.device("cuda:0")
    N = 3
    g = torch.Generator().manual_seed(C + H)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev).contiguous()      # noqa: E731
    G, Y, X = rnd(N, Ho, Wo, C), rnd(N, Ho, Wo, C), rnd(N, H, W, C) * 2
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
    coef = torch.stack((torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1)).to(dev).contiguous()
    xsc, xsh = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.5 + 1.0).to(dev)
    xmu, xis = (torch.randn(C, generator=g) * 0.2).to(dev), (torch.rand(C, generator=g) + 0.5).to(dev)
    wt = (torch.randn(C, 1, 3, 3, generator=g) / 3).to(dev).contiguous()
    parts = _lib.query("mny_dw_bnbwd_s2_parts", N, H, W, C)
    dx0, dw0 = torch.full((N, H, W, C), float("nan"), device=dev), torch.zeros(C, 1, 3, 3, device=dev)
    ws = torch.zeros(parts * C * 9, device=dev)
    _lib.call("mny_dw_bnbwd_s2", ptr(G), ptr(Y), ptr(sc), ptr(sh), _lib.ACT_RELU6, ptr(coef), ptr(X), ptr(xsc), ptr(xsh), _lib.ACT_RELU6, ptr(wt), None,
              ptr(dx0), ptr(dw0), ptr(ws), N, H, W, C, stream())
    dx1, dw1 = torch.full((N, H, W, C), float("nan"), device=dev), torch.zeros(C, 1, 3, 3, device=dev)
    red = torch.full((parts, 2, C), float("nan"), device=dev)
    _lib.call("mny_dw_bnbwd_s2_red_dz", ptr(G), ptr(Y), ptr(sc), ptr(sh), _lib.ACT_RELU6, ptr(coef), ptr(X), ptr(xsc), ptr(xsh), _lib.ACT_RELU6, ptr(xmu), ptr(xis),
              ptr(wt), None, ptr(dx1), ptr(dw1), ptr(ws), ptr(red), N, H, W, C, stream())
    torch.cuda.synchronize()
    assert torch.equal(dw0, dw1)
    z = X * xsc + xsh
    mask = ((z > 0) & (z < 6)).float()
    want = dx0 * mask * xsc
    assert torch.isfinite(dx1).all()
    assert (dx1 - want).abs().max().item() <= 1e-6 * want.abs().max().item()
    dz = (dx0 * mask).double()
    s = red.double().sum(0)
    s1, s2 = dz.sum((0, 1, 2)), (dz * ((X.double() - xmu.double()) * xis.double())).sum((0, 1, 2))
    assert (s[0] - s1).abs().max().item() <= 2e-5 * dz.abs().sum((0, 1, 2)).max().item() + 1e-4
    assert (s[1] - s2).abs().max().item() <= 2e-5 * (dz * ((X.double() - xmu.double()) * xis.double())).abs().sum((0, 1, 2)).max().item() + 1e-4


@pytest.mark.parametrize("M,K,Nc,ract", [(70001, 24, 144, 0), (50000, 32, 192, 0), (40003, 16, 96, 1), (8192, 32, 192, 0)])
def test_thin_expand_unit_backward_leaves_the_sums_of_the_unit_in_front(M, K, Nc, ract):
    """mny_pw_bnbwd_red == mny_pw_bnbwd (same dx, dW, dgamma, dbeta, bit for bit) + partial rows that sum to what mny_bn_bwd_reduce computes over
    (that dx, the raw output of the unit in front) — the expand conv of a residual block handing the project conv of the previous block its BN sums."""
    dev = torch.device("cuda:0")
    from mobilenet_yolo_pytorch_amd import ops
    act = _lib.ACT_RELU6
    r_act = {0: _lib.ACT_NONE, 1: _lib.ACT_RELU6}[ract]
    assert _lib.query("mny_pw_bnbwd_red_supported", M, K, Nc) == 1
    g = torch.Generator().manual_seed(M + K)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev).contiguous()      # noqa: E731
    x = rnd(1, 1, M, K)
    w = (rnd(Nc, K) * K ** -0.5).contiguous()
    xs, xh = (torch.rand(K, generator=g) + 0.5).to(dev), (torch.randn(K, generator=g) * 0.3).to(dev)
    gamma, beta = (torch.rand(Nc, generator=g) + 0.5).to(dev), (torch.randn(Nc, generator=g) * 0.2).to(dev)
    yd, st_ = ops.pw_fwd((x, xs, xh, _lib.ACT_NONE), w)
    scale, shift, mean, invstd = ops.bn_finalize(st_, M, gamma, beta)
    G, add = rnd(1, 1, M, Nc), rnd(1, 1, M, K)
    dx0, dw0, dg0, db0 = ops.pw_bnbwd(G, yd, scale, shift, act, mean, invstd, gamma, (x, xs, xh, _lib.ACT_NONE), w, addend=add)
    ry = rnd(M, K) * 2
    rsc, rsh = (torch.rand(K, generator=g) + 0.5).to(dev), (torch.randn(K, generator=g) * 0.5 + 1.0).to(dev)
    rmu, ris = (torch.randn(K, generator=g) * 0.2).to(dev), (torch.rand(K, generator=g) + 0.5).to(dev)
    ws = torch.empty(int(_lib.query("mny_pw_bnbwd_ws_floats", M, K, Nc)), device=dev)
    dx1 = torch.full((M, K), float("nan"), device=dev)
    dw1, dg1, db1 = torch.empty(Nc, K, device=dev), torch.empty(Nc, device=dev), torch.empty(Nc, device=dev)
    parts = _lib.query("mny_pw_bnbwd_red_parts", M, K, Nc)
    red = torch.full((parts, 2, K), float("nan"), device=dev)
    _lib.call("mny_pw_bnbwd_red", ptr(G), ptr(yd), ptr(scale), ptr(shift), act, ptr(mean), ptr(invstd), ptr(gamma), ptr(x), ptr(xs), ptr(xh), _lib.ACT_NONE,
              ptr(w), ptr(add), ptr(dx1), ptr(dw1), ptr(dg1), ptr(db1), ptr(ws), ptr(ry), ptr(rsc), ptr(rsh), r_act, ptr(rmu), ptr(ris), ptr(red), M, K, Nc, stream())
    torch.cuda.synchronize()
    assert torch.equal(dx1, dx0.view(M, K)) and torch.equal(dw1, dw0) and torch.equal(dg1, dg0) and torch.equal(db1, db0)
    z = ry.double() * rsc.double() + rsh.double()
    dact = torch.ones_like(z) if ract == 0 else ((z > 0) & (z < 6)).double()
    dz = dx1.double() * dact
    xhat = (ry.double() - rmu.double()) * ris.double()
    s = red.double().sum(0)
    assert torch.isfinite(s).all()
    assert (s[0] - dz.sum(0)).abs().max().item() <= 2e-5 * dz.abs().sum(0).max().item() + 1e-4
    assert (s[1] - (dz * xhat).sum(0)).abs().max().item() <= 2e-5 * (dz * xhat).abs().sum(0).max().item() + 1e-4
