"""GPU: MobileNetV3's per-pixel gate as one unit (csrc/gate.hip, include/mnyolo.h "per-pixel gate as one unit") against plain torch ops
on the CPU — models/mobilenetv3.py:26-41 (SeModule: conv C->C/4 + BN + ReLU, conv C/4->C + BN + hsigmoid, x * gate; the avg_pool is
never called) on the project conv's BN output (:69-71) plus the residual add of :72 — with the unit's rounding points modelled: the
operands of both 1x1 convs are rounded to bf16 (activations and weights, as on the bf16-storage GEMM path), everything else is fp32,
the hidden tensors are NOT rounded (they never reach HBM)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

from mobilenet_yolo_pytorch_amd import _lib  # noqa: E402

P = ctypes.c_void_p
EPS = 1e-5
SHAPES = [(40, 10), (112, 28), (160, 40)]


def ptr(t):
    return P(t.data_ptr()) if t is not None else None


def stream():
    return P(torch.cuda.current_stream().cuda_stream)


def q(t):
    return t.to(torch.bfloat16).to(torch.float32)


def make_case(M, C, R, seed, residual):
    g = torch.Generator().manual_seed(seed)
    y3 = (torch.randn(M, C, generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    s3, b3 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.4
    w1 = torch.randn(R, C, generator=g) * (2.0 / C) ** 0.5
    w2 = torch.randn(C, R, generator=g) * (2.0 / R) ** 0.5
    g1, be1 = torch.rand(R, generator=g) + 0.5, torch.randn(R, generator=g) * 0.3
    g2, be2 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.8
    add = None
    if residual == "real":
        add = (torch.randn(M, C, generator=g).to(torch.bfloat16), None, None, 0)
    elif residual == "view":
        add = (torch.randn(M, C, generator=g).to(torch.bfloat16), torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2, 0)
    return y3, s3, b3, w1, w2, g1, be1, g2, be2, add


def ref_forward(y3, s3, b3, w1, w2, g1, be1, g2, be2, add, dtype=torch.float64):
    """The unit's arithmetic in `dtype` with its bf16 operand roundings: -> dict of intermediates."""
    t = (y3.float() * s3 + b3)                               # fp32 like the kernel (one fma)
    tb = q(t).to(dtype)
    h_raw = tb @ q(w1).to(dtype).t()
    m1, v1 = h_raw.mean(0), h_raw.var(0, unbiased=False)
    sc1 = g1.to(dtype) / torch.sqrt(v1 + EPS)
    sh1 = be1.to(dtype) - m1 * sc1
    h = torch.relu(h_raw * sc1 + sh1)
    hb = q(h.float()).to(dtype)
    g_raw = hb @ q(w2).to(dtype).t()
    m2, v2 = g_raw.mean(0), g_raw.var(0, unbiased=False)
    sc2 = g2.to(dtype) / torch.sqrt(v2 + EPS)
    sh2 = be2.to(dtype) - m2 * sc2
    gate = torch.clamp(g_raw * sc2 + sh2 + 3.0, 0.0, 6.0) / 6.0
    out = t.to(dtype) * gate
    if add is not None:
        a = add[0].float().to(dtype)
        if add[1] is not None:
            a = a * add[1].to(dtype) + add[2].to(dtype)
        out = out + a
    return dict(h_raw=h_raw, g_raw=g_raw, m1=m1, v1=v1, m2=m2, v2=v2, sc1=sc1, sh1=sh1, sc2=sc2, sh2=sh2, out=out)


def cut_weights(w1d, w2d, C, R):
    """mny_gate_cut_batch_bf16 for one gate -> the chunk buffer."""
    import numpy as np
    dev = w1d.device
    wq = torch.zeros(int(_lib.query("mny_gate_wq_bytes", C, R)), device=dev, dtype=torch.uint8)
    job = np.array([(w1d.data_ptr(), w2d.data_ptr(), wq.data_ptr(), C, R)],
                   dtype=np.dtype([("w1", np.uint64), ("w2", np.uint64), ("wq", np.uint64), ("C", np.int32), ("R", np.int32)]))
    jd = torch.from_numpy(job.view(np.uint8).copy()).to(dev)
    _lib.call("mny_gate_cut_batch_bf16", ptr(jd), 1, stream())
    torch.cuda.synchronize()
    return wq


def run_forward(case, C, R):
    dev = torch.device("cuda:0")
    y3, s3, b3, w1, w2, g1, be1, g2, be2, add = case
    M = y3.shape[0]
    d = lambda t: t.to(dev).contiguous() if t is not None else None  # noqa: E731
    y3d, s3d, b3d, w1d, w2d, g1d, be1d, g2d, be2d = (d(t) for t in (y3, s3, b3, w1, w2, g1, be1, g2, be2))
    st = stream()
    wq = cut_weights(w1d, w2d, C, R)
    parts = _lib.query("mny_gate_parts", M)
    stats = torch.zeros(parts * 2 * C, device=dev)
    c1 = torch.zeros(4, R, device=dev)
    c2 = torch.zeros(4, C, device=dev)
    rm1, rv1, rm2, rv2 = torch.zeros(R, device=dev), torch.ones(R, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
    _lib.call("mny_gate_stats1_bf16", ptr(y3d), ptr(s3d), ptr(b3d), ptr(wq), ptr(stats), M, C, R, st)
    st1 = stats[:parts * 2 * R].view(parts, 2, R).double().sum(0).cpu()
    _lib.call("mny_bn_finalize", ptr(stats), parts, M, ptr(g1d), ptr(be1d), EPS, 0.1, ptr(rm1), ptr(rv1), ptr(c1[0]), ptr(c1[1]), ptr(c1[2]), ptr(c1[3]), R, st)
    _lib.call("mny_gate_stats2_bf16", ptr(y3d), ptr(s3d), ptr(b3d), ptr(wq), ptr(c1[0]), ptr(c1[1]), ptr(stats), M, C, R, st)
    st2 = stats[:parts * 2 * C].view(parts, 2, C).double().sum(0).cpu()
    _lib.call("mny_bn_finalize", ptr(stats), parts, M, ptr(g2d), ptr(be2d), EPS, 0.1, ptr(rm2), ptr(rv2), ptr(c2[0]), ptr(c2[1]), ptr(c2[2]), ptr(c2[3]), C, st)
    out = torch.full((M, C), float("nan"), device=dev, dtype=torch.bfloat16)
    ax, asc, ash, aact = (d(add[0]), d(add[1]), d(add[2]), add[3]) if add is not None else (None, None, None, 0)
    _lib.call("mny_gate_fwd_bf16", ptr(y3d), ptr(s3d), ptr(b3d), ptr(wq), ptr(c1[0]), ptr(c1[1]), ptr(c2[0]), ptr(c2[1]),
              ptr(ax), ptr(asc), ptr(ash), aact, ptr(out), M, C, R, st)
    torch.cuda.synchronize()
    return dict(st1=st1, st2=st2, c1=c1, c2=c2, out=out.float().cpu(), rm1=rm1.cpu(), rv2=rv2.cpu(), wq=wq,
                dev=dict(y3=y3d, s3=s3d, b3=b3d, g1=g1d, g2=g2d))


@pytest.mark.parametrize("C,R", SHAPES)
@pytest.mark.parametrize("M,residual", [(16 * 37 + 5, None), (4096, "real"), (64 * 64 * 3, "view"), (32768 * 2 + 21, "view")])      # the last: above the capped grid (32 768 pixels), ragged — every workgroup accumulates several 16-pixel tiles into its partial row
def test_gate_forward_matches_torch(C, R, M, residual):
    case = make_case(M, C, R, seed=C + M, residual=residual)
    ref = ref_forward(*case)
    got = run_forward(case, C, R)
    # statistics of W1 t and of W2 h (sums over all M pixels: the ragged last tile must not contribute)
    s1, q1 = ref["h_raw"].sum(0), (ref["h_raw"] ** 2).sum(0)
    assert (got["st1"][0] - s1).abs().max() <= 1e-4 * s1.abs().max() + 1e-3
    assert (got["st1"][1] - q1).abs().max() <= 1e-4 * q1.abs().max()
    assert (got["c1"][0].double().cpu() - ref["sc1"]).abs().max() <= 1e-4 * ref["sc1"].abs().max()
    assert (got["c1"][1].double().cpu() - ref["sh1"]).abs().max() <= 1e-4 * ref["sh1"].abs().max() + 1e-5
    # h is rounded to bf16 as an operand: a value on a rounding boundary may flip by one bf16 ulp -> W2 h moves by ~1e-3 of its scale
    s2 = ref["g_raw"].sum(0)
    assert (got["st2"][0] - s2).abs().max() <= 2e-3 * ref["g_raw"].abs().sum(0).max()
    assert (got["c2"][0].double().cpu() - ref["sc2"]).abs().max() <= 2e-3 * ref["sc2"].abs().max()
    err = (got["out"].double() - ref["out"]).abs()
    assert torch.isfinite(got["out"]).all()
    assert err.max() <= 1.5e-2 * ref["out"].abs().max(), (err.max().item(), ref["out"].abs().max().item())   # bf16 output: 2^-9 relative + the gate's slope
    assert err.mean() <= 2e-3 * ref["out"].abs().mean()
    assert float(got["rm1"].abs().sum()) > 0 and float((got["rv2"] - 1).abs().sum()) > 0                    # running statistics moved (mny_bn_finalize as for every unit)


def test_gate_support_query_and_errors():
    assert _lib.query("mny_gate_supported", 4096, 40, 10) == 1 and _lib.query("mny_gate_supported", 4096, 160, 40) == 1
    assert _lib.query("mny_gate_supported", 4096, 64, 16) == 0 and _lib.query("mny_gate_supported", 0, 40, 10) == 0
    dev = torch.device("cuda:0")
    z = torch.zeros(64, device=dev)
    with pytest.raises(_lib.MnyError, match="not supported"):
        _lib.call("mny_gate_stats1_bf16", ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), 16, 64, 16, stream())
    assert _lib.query("mny_gate_wq_bytes", 160, 40) == 2 * (3 * 5 + 10 * 2) * 64 * 16


def ref_backward(case, dout):
    """fp64 autograd through the unit's arithmetic with straight-through bf16 roundings at its operand points -> gradients wrt t, W1, W2,
    the two BatchNorm affine pairs."""
    y3, s3, b3, w1, w2, g1, be1, g2, be2, add = case
    dt_ = torch.float64
    qs = lambda v: v + (v.detach().float().to(torch.bfloat16).to(dt_) - v.detach())  # noqa: E731
    t = (y3.float() * s3 + b3).to(dt_).requires_grad_(True)
    W1, W2 = w1.to(dt_).requires_grad_(True), w2.to(dt_).requires_grad_(True)
    G1, B1, G2, B2 = (v.to(dt_).requires_grad_(True) for v in (g1, be1, g2, be2))
    hr = qs(t) @ qs(W1).t()
    z1 = (hr - hr.mean(0)) / torch.sqrt(hr.var(0, unbiased=False) + EPS) * G1 + B1
    h = torch.relu(z1)
    gr = qs(h) @ qs(W2).t()
    z2 = (gr - gr.mean(0)) / torch.sqrt(gr.var(0, unbiased=False) + EPS) * G2 + B2
    out = t * (torch.clamp(z2 + 3.0, 0.0, 6.0) / 6.0)
    out.backward(dout.to(dt_))
    return dict(dt=t.grad, dw1=W1.grad, dw2=W2.grad, dg1=G1.grad, db1=B1.grad, dg2=G2.grad, db2=B2.grad)


@pytest.mark.parametrize("C,R", SHAPES)
@pytest.mark.parametrize("M", [16 * 37 + 5, 128 * 40, 32768 * 2 + 21])          # the last: above the capped grid, ragged (multi-tile accumulation of the partial rows and dW partials, tail masking)
def test_gate_backward_matches_autograd(C, R, M):
    dev = torch.device("cuda:0")
    case = make_case(M, C, R, seed=3 * C + M, residual=None)
    g = torch.Generator().manual_seed(C * 7 + M)
    dout = (torch.randn(M, C, generator=g) * 0.7).to(torch.bfloat16)
    ref = ref_backward(case, dout.float())
    fw = run_forward(case, C, R)
    c1, c2, wq, dv = fw["c1"], fw["c2"], fw["wq"], fw["dev"]
    st = stream()
    doutd = dout.to(dev)
    parts = _lib.query("mny_gate_bwd_parts", M)
    red = torch.zeros(parts * 2 * C, device=dev)
    coef2, coef1 = torch.zeros(3, C, device=dev), torch.zeros(3, R, device=dev)
    dg2, db2, dg1, db1 = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(R, device=dev), torch.zeros(R, device=dev)
    _lib.call("mny_gate_bwd1_bf16", ptr(dv["y3"]), ptr(dv["s3"]), ptr(dv["b3"]), ptr(doutd), ptr(wq), ptr(c1[0]), ptr(c1[1]), ptr(c2[0]), ptr(c2[1]),
              ptr(c2[2]), ptr(c2[3]), ptr(red), M, C, R, st)
    _lib.call("mny_bn_bwd_finalize", ptr(red), parts, M, ptr(dv["g2"]), ptr(c2[2]), ptr(c2[3]), ptr(dg2), ptr(db2), ptr(coef2), C, st)
    dw2p = torch.full((parts, C, R), float("nan"), device=dev)
    _lib.call("mny_gate_bwd2_bf16", ptr(dv["y3"]), ptr(dv["s3"]), ptr(dv["b3"]), ptr(doutd), ptr(wq), ptr(c1[0]), ptr(c1[1]), ptr(c1[2]), ptr(c1[3]),
              ptr(c2[0]), ptr(c2[1]), ptr(coef2), ptr(red), ptr(dw2p), M, C, R, st)
    _lib.call("mny_bn_bwd_finalize", ptr(red), parts, M, ptr(dv["g1"]), ptr(c1[2]), ptr(c1[3]), ptr(dg1), ptr(db1), ptr(coef1), R, st)
    dw1p = torch.full((parts, R, C), float("nan"), device=dev)
    dt = torch.full((M, C), float("nan"), device=dev, dtype=torch.bfloat16)
    want_red3 = _lib.query("mny_gate_bwd_red3_supported", C, R) == 1
    red3 = torch.zeros(parts, 2, C, device=dev) if want_red3 else None
    mean3, invstd3 = torch.randn(C, device=dev) * 0.3, torch.rand(C, device=dev) + 0.5
    _lib.call("mny_gate_bwd3_bf16", ptr(dv["y3"]), ptr(dv["s3"]), ptr(dv["b3"]), ptr(doutd), ptr(wq), ptr(c1[0]), ptr(c1[1]), ptr(c2[0]), ptr(c2[1]),
              ptr(coef2), ptr(coef1), ptr(mean3), ptr(invstd3), ptr(dt), ptr(dw1p), ptr(red3), M, C, R, st)
    torch.cuda.synchronize()

    def close(name, got, want, tol):
        got, want = got.double().cpu(), want.double()
        assert torch.isfinite(got).all(), name
        err = (got - want).abs().max().item()
        assert err <= tol * want.abs().max().item() + 1e-7, (name, err, want.abs().max().item())

    # bounds: the operands of the four extra products (dg, dhr, and their transposes) are rounded to bf16 (2^-9) before they are summed
    close("dbeta2", db2, ref["db2"], 2e-3)
    close("dgamma2", dg2, ref["dg2"], 3e-3)
    close("dbeta1", db1, ref["db1"], 1e-2)
    close("dgamma1", dg1, ref["dg1"], 1e-2)
    close("dW2", dw2p.double().sum(0), ref["dw2"], 1e-2)
    close("dW1", dw1p.double().sum(0), ref["dw1"], 1e-2)
    close("dt", dt.float(), ref["dt"], 1.5e-2)
    assert ((dt.float().double().cpu() - ref["dt"]).abs().mean() <= 3e-3 * ref["dt"].abs().mean())
    if want_red3:                                   # the project unit's BN-backward sums over the STORED dt and the raw y3
        dts, y = dt.float().double().cpu(), case[0].float().double()
        yhat = (y - mean3.double().cpu()) * invstd3.double().cpu()
        r3 = red3.double().sum(0).cpu()
        close("red3 sum dt", r3[0], dts.sum(0), 1e-4)
        close("red3 sum dt*yhat", r3[1], (dts * yhat).sum(0), 1e-4)
