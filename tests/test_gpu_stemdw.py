"""GPU: mny_stemdw_bwd (csrc/stemdw.hip) — the backward of stem conv + BN + ReLU6 -> depthwise 3x3 + BN + ReLU6 (models/mobilenetv2.py:40,65-67)
as one pass that never writes the stem's output gradient — against the three launches it replaces (mny_dw_bnbwd_red -> mny_bn_bwd_finalize
-> mny_stem_bnwgrad, each checked against torch autograd in test_gpu_kernels.py) and against torch autograd (fp64, CPU) directly."""
import ctypes

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


def forward_on_gpu(x, ws, wd, gs, bs, gd_, bd_, dev):
    """stem -> BN -> ReLU6 -> dw3x3 -> BN with the library's own kernels; returns the raw tensors and coefficient vectors"""
    N, _, H, W = x.shape
    C = 32
    Ho, Wo = H // 2, W // 2
    M = N * Ho * Wo
    st = stream()
    s = torch.empty(N, Ho, Wo, C, device=dev)
    parts = _lib.query("mny_stem_stat_parts", N, H, W, C)
    stats = torch.zeros(max(parts, _lib.query("mny_dw_stat_parts", N, Ho, Wo, C, 3, 1)) * 2 * C, device=dev)
    _lib.call("mny_stem_fwd", ptr(x), ptr(ws), ptr(s), ptr(stats), N, H, W, C, st)
    sc = torch.zeros(4, C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    _lib.call("mny_bn_finalize", ptr(stats), parts, M, ptr(gs), ptr(bs), EPS, 0.1, ptr(rm), ptr(rv), ptr(sc[0]), ptr(sc[1]), ptr(sc[2]), ptr(sc[3]), C, st)
    d = torch.empty(N, Ho, Wo, C, device=dev)
    dparts = _lib.query("mny_dw_stat_parts", N, Ho, Wo, C, 3, 1)
    _lib.call("mny_dw_fwd", ptr(s), ptr(sc[0]), ptr(sc[1]), _lib.ACT_RELU6, ptr(wd), ptr(d), ptr(stats), N, Ho, Wo, C, 3, 1, st)
    dc = torch.zeros(4, C, device=dev)
    _lib.call("mny_bn_finalize", ptr(stats), dparts, M, ptr(gd_), ptr(bd_), EPS, 0.1, ptr(rm), ptr(rv), ptr(dc[0]), ptr(dc[1]), ptr(dc[2]), ptr(dc[3]), C, st)
    return s, sc, d, dc


@pytest.mark.parametrize("N,H,W,seed", [(2, 64, 64, 1), (3, 48, 80, 2), (1, 32, 124, 3), (5, 96, 64, 4), (2, 16, 16, 5)])
def test_stem_and_first_depthwise_backward_in_one_pass(N, H, W, seed):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    C = 32
    x = torch.randn(N, 3, H, W, generator=g)
    ws = torch.randn(C, 3, 3, 3, generator=g) * 0.3
    wd = torch.randn(C, 3, 3, generator=g) * 0.4
    gs, bs = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.5 + 1.0
    gd_, bd_ = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.5 + 1.0
    Ho, Wo = H // 2, W // 2
    M = N * Ho * Wo
    gout = torch.randn(N, Ho, Wo, C, generator=g)
    assert _lib.query("mny_stemdw_supported", N, H, W, C, _lib.ACT_RELU6, _lib.ACT_RELU6) == 1
    xd, wsd, wdd, gsd, bsd, gdd, bdd, god = (t.to(dev).contiguous() for t in (x, ws, wd, gs, bs, gd_, bd_, gout))
    s, sc, d, dc = forward_on_gpu(xd, wsd, wdd, gsd, bsd, gdd, bdd, dev)
    st = stream()
    # the depthwise unit's BN-backward coefficients
    rparts = _lib.query("mny_bn_bwd_parts", M, C)
    red = torch.zeros(rparts * 2 * C, device=dev)
    _lib.call("mny_bn_bwd_reduce", ptr(god), ptr(d), ptr(dc[0]), ptr(dc[1]), _lib.ACT_RELU6, ptr(dc[2]), ptr(dc[3]), ptr(red), M, C, st)
    dgd, dbd, dcoef = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(3, C, device=dev)
    _lib.call("mny_bn_bwd_finalize", ptr(red), rparts, M, ptr(gdd), ptr(dc[2]), ptr(dc[3]), ptr(dgd), ptr(dbd), ptr(dcoef), C, st)
    # the three launches
    dparts = _lib.query("mny_dw_bnbwd_parts", N, Ho, Wo, C)
    wsb = torch.zeros(max(dparts * C * 9, _lib.query("mny_stem_wgrad_parts", N, H, W, C) * C * 27), device=dev)
    inred = torch.zeros(dparts * 2 * C, device=dev)
    gsb = torch.empty(N, Ho, Wo, C, device=dev)
    dwd0 = torch.zeros(C, 3, 3, device=dev)
    _lib.call("mny_dw_bnbwd_red", ptr(god), ptr(d), ptr(dc[0]), ptr(dc[1]), _lib.ACT_RELU6, ptr(dcoef), ptr(s), ptr(sc[0]), ptr(sc[1]), _lib.ACT_RELU6,
              ptr(sc[2]), ptr(sc[3]), ptr(wdd), None, ptr(gsb), ptr(dwd0), ptr(wsb), ptr(inred), N, Ho, Wo, C, 3, 1, st)
    dgs0, dbs0, scoef = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(3, C, device=dev)
    _lib.call("mny_bn_bwd_finalize", ptr(inred), dparts, M, ptr(gsd), ptr(sc[2]), ptr(sc[3]), ptr(dgs0), ptr(dbs0), ptr(scoef), C, st)
    dws0 = torch.zeros(C, 3, 3, 3, device=dev)
    _lib.call("mny_stem_bnwgrad", ptr(xd), ptr(gsb), ptr(s), ptr(sc[0]), ptr(sc[1]), _lib.ACT_RELU6, ptr(scoef), ptr(dws0), ptr(wsb), N, H, W, C, st)
    # one pass
    parts = _lib.query("mny_stemdw_bwd_parts", N, H, W, C)
    ws1 = torch.full((int(_lib.query("mny_stemdw_bwd_ws_floats", N, H, W, C)),), float("nan"), device=dev)
    dwws = torch.full((parts * C * 9,), float("nan"), device=dev)
    dws1, dgs1, dbs1, dwd1 = torch.zeros(C, 3, 3, 3, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, 3, 3, device=dev)
    _lib.call("mny_stemdw_bwd", ptr(god), ptr(d), ptr(dc[0]), ptr(dc[1]), _lib.ACT_RELU6, ptr(dcoef), ptr(s), ptr(sc[0]), ptr(sc[1]), ptr(sc[2]), ptr(sc[3]),
              ptr(gsd), _lib.ACT_RELU6, ptr(xd), ptr(wsd), ptr(wdd), ptr(dws1), ptr(dgs1), ptr(dbs1), ptr(dwd1), ptr(dwws), ptr(ws1), N, H, W, C, st)
    torch.cuda.synchronize()
    for name, a, b in (("dw_stem", dws0, dws1), ("dgamma_s", dgs0, dgs1), ("dbeta_s", dbs0, dbs1), ("dw_dw", dwd0, dwd1)):
        assert torch.isfinite(b).all(), name
        err = (a.double() - b.double()).abs().max().item()
        assert err <= 2e-4 * a.abs().max().item() + 1e-4, (name, err, a.abs().max().item())

    # torch autograd, fp64, CPU: the same chain
    dt = torch.float64
    X = x.to(dt)
    Ws, Wd = ws.to(dt).clone().requires_grad_(True), wd.to(dt).clone().requires_grad_(True)
    Gs, Bs = gs.to(dt).clone().requires_grad_(True), bs.to(dt).clone().requires_grad_(True)
    S = F.conv2d(X, Ws, stride=2, padding=1)
    a = torch.clamp(F.batch_norm(S, None, None, Gs, Bs, True, 0.1, EPS), 0.0, 6.0)
    D = F.conv2d(a, Wd[:, None], stride=1, padding=1, groups=C)
    out = torch.clamp(F.batch_norm(D, None, None, gd_.to(dt), bd_.to(dt), True, 0.1, EPS), 0.0, 6.0)
    out.backward(gout.to(dt).permute(0, 3, 1, 2))
    for name, ref, got in (("dw_stem", Ws.grad, dws1), ("dgamma_s", Gs.grad, dgs1), ("dbeta_s", Bs.grad, dbs1), ("dw_dw", Wd.grad, dwd1)):
        err = (ref - got.cpu().double()).abs().max().item()
        assert err <= 1e-3 * ref.abs().max().item() + 1e-3, (name, err, ref.abs().max().item())
