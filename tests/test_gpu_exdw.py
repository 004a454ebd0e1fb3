"""GPU: the expand + depthwise unit (csrc/exdw.hip, include/mnyolo.h "expand + depthwise as one unit") against plain torch fp32/fp64
ops on the CPU — models/mobilenetv2.py:73-85 (1x1 conv + BN + ReLU6 + depthwise 3x3 stride 2 + BN) and its autograd — and against the
materialised kernels it replaces (bit-identical expand output: same fmaf chain)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from mobilenet_yolo_pytorch_amd import _lib  # noqa: E402

P = ctypes.c_void_p
EPS = 1e-5


def ptr(t):
    return P(t.data_ptr()) if t is not None else None


def stream():
    return P(torch.cuda.current_stream().cuda_stream)


def make_case(N, H, W, K, seed, affine_in):
    g = torch.Generator().manual_seed(seed)
    C = 6 * K
    x = torch.randn(N, H, W, K, generator=g)
    w = torch.randn(C, K, generator=g) * (1.0 / K ** 0.5)
    wd = torch.randn(C, 3, 3, generator=g) * 0.4
    gam = torch.rand(C, generator=g) + 0.5
    bet = torch.randn(C, generator=g) * 0.5 + 1.0
    if affine_in:
        isc, ish = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    else:
        isc = ish = None
    return x, w, wd, gam, bet, isc, ish


def ref_forward(x, w, wd, gam, bet, isc, ish, dtype=torch.float64):
    """-> (Y [N,H,W,C], mean, var (biased), a, Z [N,H/2,W/2,C]) in `dtype` on the CPU"""
    X = x.to(dtype)
    if isc is not None:
        X = X * isc.to(dtype) + ish.to(dtype)
    Y = X @ w.to(dtype).t()
    mean = Y.mean(dim=(0, 1, 2))
    var = Y.var(dim=(0, 1, 2), unbiased=False)
    sc = gam.to(dtype) / torch.sqrt(var + EPS)
    sh = bet.to(dtype) - mean * sc
    a = torch.clamp(Y * sc + sh, 0.0, 6.0)
    Z = F.conv2d(a.permute(0, 3, 1, 2), wd.to(dtype)[:, None], stride=2, padding=1, groups=a.shape[-1]).permute(0, 2, 3, 1).contiguous()
    return X, Y, mean, var, sc, sh, a, Z


def gpu_stats(x, isc, ish, w, dev):
    N, H, W, K = x.shape
    C = w.shape[0]
    M = N * H * W
    parts = _lib.query("mny_exdw_stat_parts", M, K, C)
    st = torch.zeros(parts, 2, C, device=dev)
    _lib.call("mny_exdw_stats", ptr(x), ptr(isc), ptr(ish), 0, ptr(w), ptr(st), M, K, C, stream())
    return st.double().sum(0)


@pytest.mark.parametrize("K,N,H,W,affine", [(16, 2, 20, 20, True), (16, 3, 36, 44, False), (24, 2, 28, 36, True), (32, 2, 20, 24, False),
                                            (16, 40, 64, 64, True), (24, 9, 88, 88, False), (32, 5, 44, 44, True)])
def test_forward_matches_torch_and_statistics_match(K, N, H, W, affine):
    dev = torch.device("cuda:0")
    x, w, wd, gam, bet, isc, ish = make_case(N, H, W, K, seed=K + N + H, affine_in=affine)
    C = 6 * K
    X, Y, mean, var, sc, sh, a, Z = ref_forward(x, w, wd, gam, bet, isc, ish)
    xd, wdv, wdd = x.to(dev), w.to(dev), wd.to(dev).contiguous()
    iscd = isc.to(dev) if isc is not None else None
    ishd = ish.to(dev) if ish is not None else None
    assert _lib.query("mny_exdw_supported", N, H, W, K, C, 2) == 1
    s = gpu_stats(xd, iscd, ishd, wdv, dev).cpu()
    M = N * H * W
    assert torch.allclose(s[0] / M, mean, rtol=1e-5, atol=2e-6)
    assert torch.allclose(s[1] / M - (s[0] / M) ** 2, var, rtol=2e-5, atol=1e-6)
    # the forward consumes the coefficients mny_bn_finalize would produce: give it the reference's
    scd, shd = sc.float().to(dev), sh.float().to(dev)
    z = torch.full((N, H // 2, W // 2, C), float("nan"), device=dev)
    parts = _lib.query("mny_exdw_fwd_parts", N, H, W, K, C, 2)
    zst = torch.zeros(parts, 2, C, device=dev)
    _lib.call("mny_exdw_fwd", ptr(xd), ptr(iscd), ptr(ishd), 0, ptr(wdv), ptr(scd), ptr(shd), ptr(wdd), ptr(z), ptr(zst),
              N, H, W, K, C, 2, stream())
    torch.cuda.synchronize()
    zc = z.cpu().double()
    assert torch.isfinite(zc).all()
    err = (zc - Z).abs().max().item()
    assert err <= 2e-5 * Z.abs().max().item() + 1e-5, err
    zs = zst.double().sum(0).cpu()
    assert torch.allclose(zs[0], zc.sum(dim=(0, 1, 2)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(zs[1], (zc * zc).sum(dim=(0, 1, 2)), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("K,N,H,W", [(16, 4, 48, 40), (24, 3, 40, 56), (32, 3, 36, 28)])
def test_forward_is_bitwise_the_materialised_path(K, N, H, W):
    """mny_pw_fwd (thin vector-ALU kernel) -> mny_bn_finalize -> mny_dw_fwd on the stored tensor == mny_exdw_stats -> mny_bn_finalize ->
    mny_exdw_fwd, bit for bit in Z: the recomputed expand output is the same fmaf chain and the stencil keeps the tap order."""
    dev = torch.device("cuda:0")
    x, w, wd, gam, bet, isc, ish = make_case(N, H, W, K, seed=3 * K + H, affine_in=True)
    C, M = 6 * K, N * H * W
    xd, wv, wdd, gamd, betd, iscd, ishd = (t.to(dev).contiguous() for t in (x, w, wd, gam, bet, isc, ish))
    st = stream()
    # materialised
    y = torch.empty(N, H, W, C, device=dev)
    parts = _lib.query("mny_pw_stat_parts", M, K, C)
    stats = torch.zeros(max(parts, _lib.query("mny_exdw_stat_parts", M, K, C)) * 2 * C, device=dev)
    _lib.call("mny_pw_fwd", ptr(xd), ptr(iscd), ptr(ishd), 0, ptr(wv), None, None, ptr(y), ptr(stats), M, K, C, st)
    coef = torch.zeros(4, C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    _lib.call("mny_bn_finalize", ptr(stats), parts, M, ptr(gamd), ptr(betd), EPS, 0.1, ptr(rm), ptr(rv), ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), C, st)
    z_ref = torch.empty(N, H // 2, W // 2, C, device=dev)
    _lib.call("mny_dw_fwd", ptr(y), ptr(coef[0]), ptr(coef[1]), _lib.ACT_RELU6, ptr(wdd), ptr(z_ref), None, N, H, W, C, 3, 2, st)
    # un-materialised (its own statistics pass: another partition of the same sums, so the coefficients may differ in the last bit —
    # feed the forward the materialised path's coefficients to compare the arithmetic, then check the statistics separately)
    z = torch.empty_like(z_ref)
    _lib.call("mny_exdw_fwd", ptr(xd), ptr(iscd), ptr(ishd), 0, ptr(wv), ptr(coef[0]), ptr(coef[1]), ptr(wdd), ptr(z), None, N, H, W, K, C, 2, st)
    torch.cuda.synchronize()
    assert torch.equal(z, z_ref)
    s = gpu_stats(xd, iscd, ishd, wv, dev)
    yd = y.double()
    assert torch.allclose(s[0], yd.sum(dim=(0, 1, 2)), rtol=1e-6, atol=1e-3)
    assert torch.allclose(s[1], (yd * yd).sum(dim=(0, 1, 2)), rtol=1e-6, atol=1e-3)


@pytest.mark.parametrize("K,N,H,W,shift", [(16, 8, 64, 64, 0.3), (24, 5, 44, 88, 0.0), (32, 3, 40, 36, 0.3), (16, 6, 48, 48, 4.0), (24, 3, 36, 44, 4.0),
                                           (16, 1, 6, 10, 0.3)])
def test_gram_form_statistics_equal_the_summed_expand_output(K, N, H, W, shift, monkeypatch):
    """mny_exdw_stats derives (sum y, sum y^2) per output channel from the K x K second-moment matrix of the viewed input
    (sum y^2 = w^T (X^T X) w); MNY_EXDW_STATS=direct is the first form, which recomputes Y and sums it.  Same partial-row contract, same
    values: both against the fp64 reference, including inputs whose mean is 4 standard deviations away from zero (the variance is then
    the difference of two sums 17 times its size)."""
    dev = torch.device("cuda:0")
    x, w, wd, gam, bet, isc, ish = make_case(N, H, W, K, seed=7 * K + H, affine_in=True)
    ish = ish + shift
    X, Y, mean, var, sc, sh, a, Z = ref_forward(x, w, wd, gam, bet, isc, ish)
    xd, wv, iscd, ishd = (t.to(dev).contiguous() for t in (x, w, isc, ish))
    M = N * H * W
    got = {}
    for form in ("gram", "direct"):
        monkeypatch.setenv("MNY_EXDW_STATS", form)
        s = gpu_stats(xd, iscd, ishd, wv, dev).cpu()
        got[form] = s
        m = s[0] / M
        v = s[1] / M - m * m
        assert torch.allclose(m, mean, rtol=1e-5, atol=2e-6), (form, (m - mean).abs().max().item())
        assert torch.allclose(v, var, rtol=1e-4 if shift > 1 else 2e-5, atol=1e-6), (form, ((v - var) / var).abs().max().item())
    assert torch.allclose(got["gram"][0], got["direct"][0], rtol=2e-6, atol=1e-3 * M ** 0.5)
    assert torch.allclose(got["gram"][1], got["direct"][1], rtol=2e-6, atol=1e-3 * M ** 0.5)


def ref_backward(x, w, wd, gam, bet, zgam, zbet, isc, ish, gz):
    """autograd (fp64, CPU) of  X -> Y = X W^T -> BN(batch) -> ReLU6 -> dw3x3 s2 -> BN(batch) -> ReLU6, given gz = dL/d(output).
    -> dict of gradients wrt the VIEWED input X (what the unit's dx is), w, gamma/beta of the expand BN, wd, plus the forward tensors"""
    dt = torch.float64
    X = x.to(dt)
    if isc is not None:
        X = X * isc.to(dt) + ish.to(dt)
    X = X.clone().requires_grad_(True)
    W_, Wd_, G_, Bt_ = (t.to(dt).clone().requires_grad_(True) for t in (w, wd, gam, bet))
    Y = X @ W_.t()
    mean, var = Y.mean(dim=(0, 1, 2)), Y.var(dim=(0, 1, 2), unbiased=False)
    a = torch.clamp((Y - mean) / torch.sqrt(var + EPS) * G_ + Bt_, 0.0, 6.0)
    Z = F.conv2d(a.permute(0, 3, 1, 2), Wd_[:, None], stride=2, padding=1, groups=a.shape[-1]).permute(0, 2, 3, 1)
    zm, zv = Z.mean(dim=(0, 1, 2)), Z.var(dim=(0, 1, 2), unbiased=False)
    out = torch.clamp((Z - zm) / torch.sqrt(zv + EPS) * zgam.to(dt) + zbet.to(dt), 0.0, 6.0)
    out.backward(gz.to(dt))
    return dict(dx=X.grad, dw=W_.grad, dgamma=G_.grad, dbeta=Bt_.grad, dwd=Wd_.grad, Z=Z.detach(), zmean=zm.detach(), zvar=zv.detach(),
                mean=mean.detach(), var=var.detach())


@pytest.mark.parametrize("K,N,H,W,affine,addend", [(16, 2, 20, 20, True, False), (16, 3, 36, 44, False, True), (24, 2, 28, 36, True, True),
                                                   (32, 2, 20, 24, False, False), (16, 12, 64, 48, True, False)])
def test_backward_matches_autograd(K, N, H, W, affine, addend):
    dev = torch.device("cuda:0")
    x, w, wd, gam, bet, isc, ish = make_case(N, H, W, K, seed=7 * K + N + W, affine_in=affine)
    C, M, Ho, Wo = 6 * K, N * H * W, H // 2, W // 2
    g = torch.Generator().manual_seed(K + H)
    zgam, zbet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.5 + 1.0
    gz = torch.randn(N, Ho, Wo, C, generator=g)
    add = torch.randn(N, H, W, K, generator=g) if addend else None
    R = ref_backward(x, w, wd, gam, bet, zgam, zbet, isc, ish, gz)
    st = stream()
    d = lambda t: t.to(dev).contiguous() if t is not None else None  # noqa: E731
    xd, wv, wdd, gamd, betd, iscd, ishd, zgd, zbd, gzd, addd = (d(t) for t in (x, w, wd, gam, bet, isc, ish, zgam, zbet, gz, add))
    # forward through the unit (its own statistics)
    parts = _lib.query("mny_exdw_stat_parts", M, K, C)
    stats = torch.zeros(2048 * 2 * C, device=dev)
    _lib.call("mny_exdw_stats", ptr(xd), ptr(iscd), ptr(ishd), 0, ptr(wv), ptr(stats), M, K, C, st)
    ec = torch.zeros(4, C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    _lib.call("mny_bn_finalize", ptr(stats), parts, M, ptr(gamd), ptr(betd), EPS, 0.1, ptr(rm), ptr(rv), ptr(ec[0]), ptr(ec[1]), ptr(ec[2]), ptr(ec[3]), C, st)
    z = torch.empty(N, Ho, Wo, C, device=dev)
    zparts = _lib.query("mny_exdw_fwd_parts", N, H, W, K, C, 2)
    _lib.call("mny_exdw_fwd", ptr(xd), ptr(iscd), ptr(ishd), 0, ptr(wv), ptr(ec[0]), ptr(ec[1]), ptr(wdd), ptr(z), ptr(stats), N, H, W, K, C, 2, st)
    zc = torch.zeros(4, C, device=dev)
    Mz = N * Ho * Wo
    _lib.call("mny_bn_finalize", ptr(stats), zparts, Mz, ptr(zgd), ptr(zbd), EPS, 0.1, ptr(rm), ptr(rv), ptr(zc[0]), ptr(zc[1]), ptr(zc[2]), ptr(zc[3]), C, st)
    # the depthwise unit's BN backward coefficients
    red = torch.zeros(2048 * 2 * C, device=dev)
    rparts = _lib.query("mny_bn_bwd_parts", Mz, C)
    zcoef = torch.zeros(3, C, device=dev)
    dgz, dbz = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    _lib.call("mny_bn_bwd_reduce", ptr(gzd), ptr(z), ptr(zc[0]), ptr(zc[1]), 1, ptr(zc[2]), ptr(zc[3]), ptr(red), Mz, C, st)
    _lib.call("mny_bn_bwd_finalize", ptr(red), rparts, Mz, ptr(zgd), ptr(zc[2]), ptr(zc[3]), ptr(dgz), ptr(dbz), ptr(zcoef), C, st)
    ws = torch.zeros(int(_lib.query("mny_exdw_bwd_ws_floats", N, H, W, K, C, 2)), device=dev)
    bparts = _lib.query("mny_exdw_bwd_parts", N, H, W, K, C, 2)
    dws = torch.zeros(bparts * C * 9, device=dev)
    dx = torch.full((N, H, W, K), float("nan"), device=dev)
    dwe, dge, dbe, dwd = torch.zeros(C, K, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, 3, 3, device=dev)
    _lib.call("mny_exdw_bwd", ptr(gzd), ptr(z), ptr(zc[0]), ptr(zc[1]), 1, ptr(zcoef), ptr(xd), ptr(iscd), ptr(ishd), 0, ptr(wv),
              ptr(ec[0]), ptr(ec[1]), ptr(ec[2]), ptr(ec[3]), ptr(gamd), ptr(wdd), ptr(addd), ptr(dx), ptr(dwe), ptr(dge), ptr(dbe),
              ptr(dwd), ptr(dws), ptr(ws), N, H, W, K, C, 2, st)
    torch.cuda.synchronize()

    def close(name, got, want, rtol):
        got, want = got.cpu().double(), want.double()
        assert torch.isfinite(got).all(), name
        err = (got - want).abs().max().item()
        assert err <= rtol * want.abs().max().item() + 1e-6, (name, err, want.abs().max().item())

    want_dx = R["dx"] + (add.double() if add is not None else 0.0)
    close("dx", dx, want_dx, 2e-4)
    close("dw_exp", dwe, R["dw"], 2e-4)
    close("dgamma", dge, R["dgamma"], 2e-4)
    close("dbeta", dbe, R["dbeta"], 2e-4)
    close("dw_dw", dwd, R["dwd"], 2e-4)
    # partial rows of the depthwise weight gradient (the engine's deferred combine reads these)
    close("dw_dw partial rows", dws.view(bparts, C, 3, 3).double().sum(0), R["dwd"], 2e-4)


@pytest.mark.parametrize("K,N,H,W", [(16, 6, 40, 36), (24, 3, 28, 44), (32, 2, 20, 24)])
def test_backward_with_the_input_units_bn_sums_equals_a_separate_reduce_pass(K, N, H, W):
    """mny_exdw_bwd_red: same dx / parameter gradients as mny_exdw_bwd, and its partial rows sum to what mny_bn_bwd_reduce computes over
    the finished dx and the raw input (the unit in front: sum dz, sum dz * yhat)."""
    dev = torch.device("cuda:0")
    x, w, wd, gam, bet, isc, ish = make_case(N, H, W, K, seed=5 * K + H, affine_in=True)
    C, M, Ho, Wo = 6 * K, N * H * W, H // 2, W // 2
    g = torch.Generator().manual_seed(K + W)
    zgam, zbet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.5 + 1.0
    gz = torch.randn(N, Ho, Wo, C, generator=g)
    imean, iinv = torch.randn(K, generator=g) * 0.2, torch.rand(K, generator=g) + 0.5
    st = stream()
    xd, wv, wdd, gamd, betd, iscd, ishd, zgd, zbd, gzd, imd, iid = (t.to(dev).contiguous() for t in (x, w, wd, gam, bet, isc, ish, zgam, zbet, gz, imean, iinv))
    parts = _lib.query("mny_exdw_stat_parts", M, K, C)
    stats = torch.zeros(2048 * 2 * C, device=dev)
    _lib.call("mny_exdw_stats", ptr(xd), ptr(iscd), ptr(ishd), 0, ptr(wv), ptr(stats), M, K, C, st)
    ec = torch.zeros(4, C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    _lib.call("mny_bn_finalize", ptr(stats), parts, M, ptr(gamd), ptr(betd), EPS, 0.1, ptr(rm), ptr(rv), ptr(ec[0]), ptr(ec[1]), ptr(ec[2]), ptr(ec[3]), C, st)
    z = torch.empty(N, Ho, Wo, C, device=dev)
    zparts = _lib.query("mny_exdw_fwd_parts", N, H, W, K, C, 2)
    _lib.call("mny_exdw_fwd", ptr(xd), ptr(iscd), ptr(ishd), 0, ptr(wv), ptr(ec[0]), ptr(ec[1]), ptr(wdd), ptr(z), ptr(stats), N, H, W, K, C, 2, st)
    zc = torch.zeros(4, C, device=dev)
    Mz = N * Ho * Wo
    _lib.call("mny_bn_finalize", ptr(stats), zparts, Mz, ptr(zgd), ptr(zbd), EPS, 0.1, ptr(rm), ptr(rv), ptr(zc[0]), ptr(zc[1]), ptr(zc[2]), ptr(zc[3]), C, st)
    red = torch.zeros(2048 * 2 * C, device=dev)
    zcoef = torch.zeros(3, C, device=dev)
    dgz, dbz = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    _lib.call("mny_bn_bwd_reduce", ptr(gzd), ptr(z), ptr(zc[0]), ptr(zc[1]), 1, ptr(zc[2]), ptr(zc[3]), ptr(red), Mz, C, st)
    _lib.call("mny_bn_bwd_finalize", ptr(red), _lib.query("mny_bn_bwd_parts", Mz, C), Mz, ptr(zgd), ptr(zc[2]), ptr(zc[3]), ptr(dgz), ptr(dbz), ptr(zcoef), C, st)
    ws = torch.zeros(int(_lib.query("mny_exdw_bwd_ws_floats", N, H, W, K, C, 2)), device=dev)
    bparts = _lib.query("mny_exdw_bwd_parts", N, H, W, K, C, 2)
    outs = []
    for with_red in (False, True):
        dws = torch.zeros(bparts * C * 9, device=dev)
        dx = torch.full((N, H, W, K), float("nan"), device=dev)
        dwe, dge, dbe, dwd = torch.zeros(C, K, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, 3, 3, device=dev)
        if with_red:
            rparts = _lib.query("mny_exdw_bwd_red_parts", N, H, W, K, C, 2)
            ired = torch.full((rparts, 2, K), float("nan"), device=dev)
            _lib.call("mny_exdw_bwd_red", ptr(gzd), ptr(z), ptr(zc[0]), ptr(zc[1]), 1, ptr(zcoef), ptr(xd), ptr(iscd), ptr(ishd), 0, ptr(imd), ptr(iid), ptr(wv),
                      ptr(ec[0]), ptr(ec[1]), ptr(ec[2]), ptr(ec[3]), ptr(gamd), ptr(wdd), None, ptr(dx), ptr(dwe), ptr(dge), ptr(dbe),
                      ptr(dwd), ptr(dws), ptr(ws), ptr(ired), N, H, W, K, C, 2, st)
        else:
            _lib.call("mny_exdw_bwd", ptr(gzd), ptr(z), ptr(zc[0]), ptr(zc[1]), 1, ptr(zcoef), ptr(xd), ptr(iscd), ptr(ishd), 0, ptr(wv),
                      ptr(ec[0]), ptr(ec[1]), ptr(ec[2]), ptr(ec[3]), ptr(gamd), ptr(wdd), None, ptr(dx), ptr(dwe), ptr(dge), ptr(dbe),
                      ptr(dwd), ptr(dws), ptr(ws), N, H, W, K, C, 2, st)
        torch.cuda.synchronize()
        outs.append((dx, dwe, dge, dbe, dwd))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    # a separate reduce pass over (dx, raw x) with the same view and statistics
    rparts2 = _lib.query("mny_bn_bwd_parts", M, K)
    red2 = torch.zeros(rparts2, 2, K, device=dev)
    _lib.call("mny_bn_bwd_reduce", ptr(outs[1][0]), ptr(xd), ptr(iscd), ptr(ishd), 0, ptr(imd), ptr(iid), ptr(red2), M, K, st)
    torch.cuda.synchronize()
    got, want = ired.double().sum(0).cpu(), red2.double().sum(0).cpu()
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-3 * float(want.abs().max())), (got, want)


def test_backward_refuses_an_addend_aliased_to_dx():
    """the backward is two-pass — pass 1 overwrites dx, only the remainder kernel reads the addend — so addend == dx would lose
    the earlier gradient silently.  The entry point refuses it (and the engine never emits it: contribute_kernel(..., inplace_ok=False))."""
    dev = torch.device("cuda:0")
    K, N, H, W = 16, 2, 20, 20
    C, Ho, Wo = 6 * K, H // 2, W // 2
    z = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
    ws = z(int(_lib.query("mny_exdw_bwd_ws_floats", N, H, W, K, C, 2)))
    dws = z(_lib.query("mny_exdw_bwd_parts", N, H, W, K, C, 2) * C * 9)
    dx = z(N, H, W, K)
    v = [z(C) for _ in range(8)]
    args = lambda addend: (ptr(z(N, Ho, Wo, C)), ptr(z(N, Ho, Wo, C)), ptr(v[0]), ptr(v[1]), 1, ptr(z(3, C)), ptr(z(N, H, W, K)), None, None, 0,  # noqa: E731
                           ptr(z(C, K)), ptr(v[2]), ptr(v[3]), ptr(v[4]), ptr(v[5]), ptr(v[6]), ptr(z(C, 3, 3)), ptr(addend), ptr(dx), ptr(z(C, K)),
                           ptr(v[7]), ptr(z(C)), ptr(z(C, 3, 3)), ptr(dws), ptr(ws), N, H, W, K, C, 2, stream())
    with pytest.raises(_lib.MnyError, match="alias"):
        _lib.call("mny_exdw_bwd", *args(dx))
    _lib.call("mny_exdw_bwd", *args(z(N, H, W, K)))          # a separate addend is fine
    torch.cuda.synchronize()
