"""GPU: mny_pj_bwd (csrc/pjbwd.hip) — the backward of a project conv + BN unit (models/mobilenetv2.py:69-70,83-84) as one pass — against
the same arithmetic in fp64 on the CPU (BN-backward-apply of a linear unit, data gradient, the BN-backward sums of the unit in front,
weight gradient) and against the three launches it replaces."""
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


def make(M, Ki, No, seed, dev):
    g = torch.Generator().manual_seed(seed)
    G, Y = torch.randn(M, No, generator=g), torch.randn(M, No, generator=g)
    coef = torch.stack((torch.rand(No, generator=g) + 0.5, torch.randn(No, generator=g) * 0.1, torch.randn(No, generator=g) * 0.1)).contiguous()
    D = torch.randn(M, Ki, generator=g) * 2.0
    dsc, dsh = torch.rand(Ki, generator=g) + 0.5, torch.randn(Ki, generator=g) * 0.5 + 1.0
    dmu, dis = torch.randn(Ki, generator=g) * 0.2, torch.rand(Ki, generator=g) + 0.5
    W = torch.randn(No, Ki, generator=g) * Ki ** -0.5
    return [t.to(dev).contiguous() for t in (G, Y, coef, D, dsc, dsh, dmu, dis, W)]


def run_fused(G, Y, coef, D, dsc, dsh, dmu, dis, W, act):
    dev = G.device
    M, No = G.shape
    Ki = D.shape[1]
    parts = _lib.query("mny_pj_bwd_parts", M, Ki, No)
    gd = torch.full((M, Ki), float("nan"), device=dev)
    dw = torch.zeros(No, Ki, device=dev)
    dws = torch.full((parts * No * Ki,), float("nan"), device=dev)
    red = torch.full((parts, 2, Ki), float("nan"), device=dev)
    _lib.call("mny_pj_bwd", ptr(G), ptr(Y), ptr(coef), ptr(D), ptr(dsc), ptr(dsh), ptr(dmu), ptr(dis), act, ptr(W), ptr(gd), ptr(dw), ptr(dws), ptr(red),
              M, Ki, No, stream())
    torch.cuda.synchronize()
    return gd, dw, red.double().sum(0)


@pytest.mark.parametrize("M,Ki,No,act", [(16 * 40, 32, 16, 1), (2000, 96, 24, 1), (1234, 144, 24, 1), (4096 + 7, 144, 32, 1), (3000, 192, 32, 1),
                                         (100, 32, 16, 0), (50000, 32, 16, 1), (16, 96, 24, 3)])
def test_project_unit_backward_in_one_pass(M, Ki, No, act):
    dev = torch.device("cuda:0")
    acts = {0: _lib.ACT_NONE, 1: _lib.ACT_RELU6, 3: _lib.ACT_LEAKY}
    a_id = acts[act]
    assert _lib.query("mny_pj_bwd_supported", M, Ki, No, a_id) == 1
    T = make(M, Ki, No, seed=M + Ki, dev=dev)
    gd, dw, red = run_fused(*T, a_id)
    G, Y, coef, D, dsc, dsh, dmu, dis, W = (t.double().cpu() for t in T)
    dY = coef[0] * G + coef[1] * Y + coef[2]
    gd_ref = dY @ W
    z = D * dsc + dsh
    slope, hi = {0: (1.0, float("inf")), 1: (0.0, 6.0), 3: (0.1, float("inf"))}[act]
    a = torch.minimum(torch.maximum(z, slope * z), torch.tensor(hi, dtype=torch.float64))
    dact = torch.where(z > 0, torch.ones_like(z), torch.full_like(z, slope)) * (z < hi).double()
    dz = gd_ref * dact
    s1, s2 = dz.sum(0), (dz * (D - dmu) * dis).sum(0)
    dw_ref = dY.t() @ a
    assert torch.isfinite(gd).all()
    err = (gd.double().cpu() - gd_ref).abs().max().item()
    assert err <= 2e-5 * gd_ref.abs().max().item() + 1e-5, err
    for name, got, ref in (("dw", dw.double().cpu(), dw_ref), ("s1", red[0].cpu(), s1), ("s2", red[1].cpu(), s2)):
        e = (got - ref).abs().max().item()
        assert e <= 2e-4 * ref.abs().max().item() + 1e-3, (name, e, ref.abs().max().item())


@pytest.mark.parametrize("M,Ki,No", [(40000, 32, 16), (20000, 144, 24), (12345, 192, 32)])
def test_project_unit_backward_equals_the_three_launches(M, Ki, No):
    """mny_bn_bwd_apply -> mny_pw_dgrad_bnred (+ its BN sums) -> mny_pw_wgrad on the same inputs."""
    dev = torch.device("cuda:0")
    act = _lib.ACT_RELU6
    T = make(M, Ki, No, seed=3 * M + No, dev=dev)
    G, Y, coef, D, dsc, dsh, dmu, dis, W = T
    gd, dw, red = run_fused(*T, act)
    st = stream()
    dY = torch.empty_like(G)
    one, zero = torch.ones(No, device=dev), torch.zeros(No, device=dev)
    _lib.call("mny_bn_bwd_apply", ptr(G), ptr(Y), ptr(one), ptr(zero), _lib.ACT_NONE, ptr(coef), ptr(dY), M, No, st)
    if _lib.query("mny_pw_dgrad_bnred_supported", M, No, Ki, act) == 1:
        wT = W.t().contiguous()
        rparts = _lib.query("mny_pw_dgrad_bnred_parts", M, No, Ki)
        rbuf = torch.zeros(rparts, 2, Ki, device=dev)
        gd0 = torch.empty(M, Ki, device=dev)
        _lib.call("mny_pw_dgrad_bnred", ptr(dY), ptr(wT), ptr(gd0), ptr(D), ptr(dsc), ptr(dsh), act, ptr(dmu), ptr(dis), ptr(rbuf), M, No, Ki, st)
        torch.cuda.synchronize()
        assert (gd - gd0).abs().max().item() <= 2e-5 * gd0.abs().max().item() + 1e-5
        r0 = rbuf.double().sum(0)
        for k in range(2):
            assert (red[k] - r0[k]).abs().max().item() <= 2e-4 * r0[k].abs().max().item() + 1e-3
    dw0 = torch.zeros(No, Ki, device=dev)
    ws = torch.zeros(int(_lib.query("mny_pw_wgrad_ws_floats", M, Ki, No)) + 16, device=dev)
    _lib.call("mny_pw_wgrad", ptr(D), ptr(dsc), ptr(dsh), act, ptr(dY), ptr(dw0), None, ptr(ws), M, Ki, No, st)
    torch.cuda.synchronize()
    assert (dw - dw0).abs().max().item() <= 2e-4 * dw0.abs().max().item() + 1e-3


def _q(t):
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("M,Ki,No,act", [(16 * 40, 16, 16, 3), (2000, 64, 24, 3), (1234, 72, 24, 3), (4096 + 7, 72, 40, 3), (3000, 120, 40, 3), (128 * 9, 120, 40, 4),
                                         (65536 * 2 + 21, 72, 24, 3), (65536 + 4103, 120, 40, 3),       # above the 65 536-pixel cap of the grid, ragged
                                         (50000, 16, 16, 3), (16, 64, 24, 1)])
def test_project_unit_backward_in_one_pass_bf16_storage(M, Ki, No, act):
    """mny_pj_bwd_bf16 (csrc/gate.hip): MobileNetV3's thin project convs (models/mobilenetv3.py:57-58,69) on bf16 storage, against fp64 with the
    unit's rounding points modelled — G, Y, D stored bf16; the matrix operands dY, W and a_d = act(BN(D)) rounded to bf16; gd stored bf16, the
    BN-backward sums of the depthwise unit taken over the STORED gd."""
    dev = torch.device("cuda:0")
    a_id = {1: _lib.ACT_RELU6, 3: _lib.ACT_RELU, 4: _lib.ACT_HSWISH}[act]
    assert _lib.query("mny_pj_bwd_supported_bf16", M, Ki, No, a_id) == 1
    G, Y, coef, D, dsc, dsh, dmu, dis, W = make(M, Ki, No, seed=M + Ki, dev=dev)
    G16, Y16, D16 = G.to(torch.bfloat16), Y.to(torch.bfloat16), D.to(torch.bfloat16)
    parts = _lib.query("mny_pj_bwd_parts_bf16", M, Ki, No)
    gd = torch.full((M, Ki), float("nan"), device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(No, Ki, device=dev)
    dws = torch.full((parts * No * Ki,), float("nan"), device=dev)
    red = torch.full((parts, 2, Ki), float("nan"), device=dev)
    _lib.call("mny_pj_bwd_bf16", ptr(G16), ptr(Y16), ptr(coef), ptr(D16), ptr(dsc), ptr(dsh), ptr(dmu), ptr(dis), a_id, ptr(W), ptr(gd), ptr(dw), ptr(dws), ptr(red),
              M, Ki, No, stream())
    torch.cuda.synchronize()
    Gd, Yd, Dd = G16.double().cpu(), Y16.double().cpu(), D16.double().cpu()
    c, dscd, dshd, dmud, disd, Wd = (t.double().cpu() for t in (coef, dsc, dsh, dmu, dis, W))
    dY = (c[0] * Gd.float() + (c[1] * Yd.float() + c[2])).double()           # fp32-ish; the bf16 rounding below dominates
    gd_ref = _q(dY.float()) @ _q(Wd.float())
    z = Dd * dscd + dshd
    if act == 4:
        a = z * torch.clamp(z + 3, 0, 6) / 6
        dact = torch.where(z <= -3, torch.zeros_like(z), torch.where(z >= 3, torch.ones_like(z), (2 * z + 3) / 6))
    else:
        hi = 6.0 if act == 1 else float("inf")
        a = torch.clamp(z, 0, hi)
        dact = ((z > 0) & (z < hi)).double()
    got = gd.double().cpu()
    assert torch.isfinite(got).all()
    assert (got - gd_ref).abs().max().item() <= 1e-2 * gd_ref.abs().max().item()          # bf16 store
    dz = got * dact                                                                      # sums over the STORED gradient
    s1, s2 = dz.sum(0), (dz * (Dd - dmud) * disd).sum(0)
    dw_ref = _q(dY.float()).t() @ _q(a.float())
    r = red.double().sum(0).cpu()
    for name, g_, ref, tol in (("dw", dw.double().cpu(), dw_ref, 2e-3), ("dw partial rows", dws.view(parts, No, Ki).double().sum(0).cpu(), dw_ref, 2e-3),
                               ("s1", r[0], s1, 2e-3), ("s2", r[1], s2, 2e-3)):
        e = (g_ - ref).abs().max().item()
        assert e <= tol * ref.abs().max().item() + 1e-3, (name, e, ref.abs().max().item())
