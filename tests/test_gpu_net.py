"""Whole-network parity on the MI355X: the drop-in module (HIP path) against fixtures captured from the
real reference and against the torch-CPU oracle model on the same procedural weights and seeded inputs.

Tolerances (fp32, ~55 layers with training-mode BatchNorm; GPU and CPU sum in different orders):
  heads        : 2e-3 relative to the tensor's max |value|
  losses       : 1e-3 relative
  grad norms   : 1e-2 relative per parameter tensor (+ 1e-6 abs), full tensors 2e-2 of max |g|
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import net_ref, procedural

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _model(train=False, sync=True):
    from mobilenet_yolo_pytorch_amd import yolo
    torch.manual_seed(0)
    m = yolo(procedural.VOC_CONFIG, sync_metrics=sync)
    procedural.fill_state_dict_(m)
    m = m.cuda()
    return m.train() if train else m.eval()


def _close(got, ref, rel, what):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = np.abs(ref).max() + 1e-12
    err = np.abs(got - ref).max()
    assert err <= rel * scale, "%s: max err %.3e vs scale %.3e (rel %.2e > %.1e)" % (what, err, scale, err / scale, rel)


def _assert_gradient_tensors_close(m, ref, tau):
    """Every parameter gradient as a TENSOR: ||g - g_ref|| <= tau * ||g_ref|| + 2e-5 (norms alone let a sign flip, a
    permuted channel block or a transposed tile through — those give a relative difference of 1.4-2).  tau is measured
    (tools/grad_direction.py, same seeds): worst tensor with ||g_ref|| > 1e-4 at bs 8 / 64 / 256: 3.1e-2 / 2.1e-2 / 1.9e-2 — the
    fp32 reordering noise of ~55 training-mode BatchNorm layers discussed in test_train_step_matches_reference_fixture; the 2e-5 floor
    covers the BN biases whose exact gradient is 0 (a shift in front of a batch-normalised conv: ||g_ref|| ~ 1e-8, pure rounding)."""
    rp = dict(ref.named_parameters())
    n_cmp, worst = 0, (0.0, None)
    for k, p in m.named_parameters():
        if rp[k].grad is None:
            assert p.grad is None, k
            continue
        a, b = p.grad.double().cpu(), rp[k].grad.double()
        d, nb = (a - b).norm().item(), b.norm().item()
        assert d <= tau * nb + 2e-5, (k, d, nb, d / (nb + 1e-30))
        if nb > 1e-4 and d / nb > worst[0]:
            worst = (d / nb, k)
        n_cmp += 1
    return n_cmp, worst


def _heads(m, plan):
    return [h.permute(0, 3, 1, 2).contiguous().cpu().numpy() for h in plan.heads]


def test_eval_heads_match_reference_fixture():
    z = np.load(os.path.join(G, "net_eval.npz"))
    m = _model()
    for hs in m.yolo_losses:
        hs.val_conf = 0.3
    for tag, (n, s) in {"a": (2, 96), "b": (1, 352)}.items():
        x = procedural.images(n, s, s, seed=10).cuda()
        det = m(x)
        plan = m._plans[(n, s, s, False)]
        o0, o1 = _heads(m, plan)
        _close(o0, z["out0_" + tag], 2e-3, "out0 " + tag)
        _close(o1, z["out1_" + tag], 2e-3, "out1 " + tag)
        assert len(det) == n and all(d.shape[1] == 7 and d.is_cuda for d in det)
        # Detection counts against the REAL reference's (fixture): the GPU heads differ from the reference's by <= 2e-3 of the head's
        # range, i.e. a confidence by <= DELTA below — so a count may differ only through candidates whose confidence lies within DELTA
        # of val_conf (they can enter or leave, and take the boxes they suppress with them).  Counted on the GPU heads by the oracle's
        # decode: every image must match EXACTLY when it has no such borderline candidate, else within their number.
        from oracle import yolo_ref
        DELTA = 5e-3
        specs = yolo_ref.specs_from_config(procedural.VOC_CONFIG)
        border = np.zeros(n, dtype=np.int64)
        for hi in range(2):
            specs[hi].val_conf = 0.3 - DELTA
            rows = yolo_ref.decode_rows(plan.heads[hi].cpu(), specs[hi], [s, s], layout="nhwc")
            border += np.array([int((r[:, 4] <= 0.3 + DELTA).sum()) for r in rows])
        ref_counts = z["det_counts_" + tag]
        for d, rc, nb in zip(det, ref_counts, border):
            assert abs(len(d) - rc) <= 2 * nb, (tag, len(d), int(rc), int(nb))


def test_eval_detections_equal_oracle_pipeline_on_same_heads():
    """Decode + NMS of the GPU heads, recomputed by the oracle from those same head tensors."""
    from oracle import nms_ref, yolo_ref
    m = _model()
    for hs in m.yolo_losses:
        hs.val_conf = 0.3
    x = procedural.images(2, 96, 96, seed=10).cuda()
    det = m(x)
    plan = m._plans[(2, 96, 96, False)]
    specs = yolo_ref.specs_from_config(procedural.VOC_CONFIG)
    rows = []
    for hi in range(2):
        specs[hi].val_conf = 0.3
        rows.append(yolo_ref.decode_rows(plan.heads[hi].cpu(), specs[hi], [96, 96], layout="nhwc"))
    ref = nms_ref.nms_driver(tuple(rows), 20)
    # same heads on both sides: the only legitimate difference is a candidate whose confidence sits within float rounding (1e-5) of
    # val_conf — none in this batch (asserted), so counts and rows must agree exactly / to 1e-4 (no slack on the count)
    near = 0
    for hi in range(2):
        specs[hi].val_conf = 0.3 - 1e-5
        near += sum(int((r[:, 4] <= 0.3 + 1e-5).sum()) for r in yolo_ref.decode_rows(plan.heads[hi].cpu(), specs[hi], [96, 96], layout="nhwc"))
    assert near == 0, "the seeded batch has a candidate within 1e-5 of val_conf: pick another seed"
    for d, r in zip(det, ref):
        assert len(d) == len(r), (len(d), len(r))
        np.testing.assert_allclose(d.cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-4)


def test_train_step_matches_reference_fixture():
    z = np.load(os.path.join(G, "net_train.npz"))
    names = json.load(open(os.path.join(G, "net_train_names.json")))
    m = _model(train=True)
    x = procedural.images(4, 128, 128, seed=11).cuda()
    tg = list(torch.split(torch.from_numpy(z["t_all"]), z["t_counts"].tolist()))
    res = m(x, tg)
    loss = sum(r[0] for r in res)
    loss.backward()
    plan = m._plans[(4, 128, 128, True)]
    o0, o1 = _heads(m, plan)
    _close(o0, z["out0"], 2e-3, "train out0")
    _close(o1, z["out1"], 2e-3, "train out1")
    for i in range(2):
        got = np.array([float(v) for v in res[i]])
        np.testing.assert_allclose(got, z["tuple%d" % i], rtol=2e-3, atol=1e-5)
    params = dict(m.named_parameters())
    assert list(params) == names["params"]
    assert [k for k, p in params.items() if p.grad is None] == names["grad_none"]          # Q10: seg branch
    gn = np.array([-1.0 if p.grad is None else p.grad.double().norm().item() for p in params.values()])
    # Batch statistics over 4 x 4 x 4 = 64 samples at the deepest layers make this step chaotic: ANY change of fp32 rounding order is
    # amplified to the percent level at the far end of the backward pass.  Measured against this fixture (worst gradient-norm error /
    # stem-gradient error): fp32-MFMA GEMMs 2.4e-3 / 7.7e-3; the same arithmetic with another accumulation order (MNY_GEMM_V1=1)
    # 7.4e-3 / 1.6e-2; six-product bf16 GEMMs (MNY_X6, whose error against an fp64 product is BELOW the fp32 MFMA's,
    # tools/x6_precision.py) 8.6e-3 / 2.2e-2.  The bounds sit above that noise, an actual bug moves these numbers by O(1).
    bad = [(k, a, b) for k, a, b in zip(params, gn, z["gnorm"]) if abs(a - b) > 2e-2 * abs(b) + 2e-5]   # floor: zero-gradient BN biases hold rounding noise
    assert not bad, bad[:5]
    _close(params["backbone.features.0.0.weight"].grad.cpu().numpy(), z["g_stem"], 4e-2, "stem grad")
    _close(params["yolo_headS16.3.weight"].grad.cpu().numpy(), z["g_head16_w"], 2e-2, "head16 grad")
    _close(params["yolo_headS32.3.bias"].grad.cpu().numpy(), z["g_head32_b"], 2e-2, "head32 bias grad")
    _close(params["backbone.features.5.conv.3.weight"].grad.cpu().numpy(), z["g_f5_dw"], 2e-2, "dw grad")
    _close(params["backbone.features.5.conv.1.weight"].grad.cpu().numpy(), z["g_f5_bn"], 2e-2, "bn grad")
    sd = m.state_dict()
    rs = np.array([sd[k].double().norm().item() for k in names["running"]])
    np.testing.assert_allclose(rs, z["rs_norm"], rtol=1e-4)
    assert int(sd["backbone.features.0.1.num_batches_tracked"]) == 1


def test_train_step_matches_oracle_bs8_352():
    """Same weights, same seeded 352x352 batch through the oracle (torch CPU) and the HIP path."""
    ref = procedural.fill_state_dict_(net_ref.RefYolo(procedural.VOC_CONFIG)).train()
    m = _model(train=True)
    x = procedural.images(8, 352, 352, seed=3)
    tg = procedural.targets(8, seed=4, empty_every=4)
    rr = ref(x, tg)
    (rr[0][0] + rr[1][0]).backward()
    res = m(x.cuda(), tg)
    (res[0][0] + res[1][0]).backward()
    for i in range(2):
        np.testing.assert_allclose(np.array([float(v) for v in res[i]]), np.array([float(v) for v in rr[i]]), rtol=2e-3, atol=1e-5)
    rp = dict(ref.named_parameters())
    for k, p in m.named_parameters():
        if rp[k].grad is None:
            assert p.grad is None
            continue
        a, b = p.grad.double().norm().item(), rp[k].grad.double().norm().item()
        assert abs(a - b) <= 2e-2 * b + 2e-5, (k, a, b)
    assert _assert_gradient_tensors_close(m, ref, 8e-2)[0] == 202          # direction too (measured worst 3.1e-2 at this batch of 8)


@pytest.mark.parametrize("size,bs", [(288, 3), (416, 2), (320, 5)])
def test_multiscale_sizes_match_oracle(size, bs):
    """The other entries of train_img_size (models/voc/config.yaml:4-9): every size builds its own static plan — other GEMM
    tilings, other fused-reduction partial-row counts, ragged tile edges."""
    ref = procedural.fill_state_dict_(net_ref.RefYolo(procedural.VOC_CONFIG)).train()
    m = _model(train=True)
    x = procedural.images(bs, size, size, seed=size)
    tg = procedural.targets(bs, seed=size + 1, empty_every=0)
    rr = ref(x, tg)
    (rr[0][0] + rr[1][0]).backward()
    res = m(x.cuda(), tg)
    (res[0][0] + res[1][0]).backward()
    for i in range(2):
        np.testing.assert_allclose(float(res[i][0]), float(rr[i][0]), rtol=2e-3, atol=1e-5)
    rp = dict(ref.named_parameters())
    for k, p in m.named_parameters():
        if rp[k].grad is None:
            assert p.grad is None
            continue
        a, b = p.grad.double().norm().item(), rp[k].grad.double().norm().item()
        assert abs(a - b) <= 2e-2 * b + 2e-5, (k, a, b)


def test_second_step_and_grad_accumulation():
    m = _model(train=True, sync=False)
    x = procedural.images(4, 96, 96, seed=1).cuda()
    tg = procedural.targets(4, seed=2, empty_every=0)
    r1 = m(x, tg)
    (r1[0][0] + r1[1][0]).backward()
    w = m.backbone.features[0][0].weight
    g1 = w.grad.clone()
    for p in m.parameters():
        p.grad = None
    sd0 = {k: v.clone() for k, v in m.state_dict().items() if "running" in k}
    m.load_state_dict({**m.state_dict(), **sd0})
    # identical second step on identical running stats is irrelevant for grads in train mode
    r2 = m(x, tg)
    (r2[0][0] + r2[1][0]).backward()
    torch.testing.assert_close(w.grad, g1, rtol=1e-5, atol=1e-7)          # deterministic replay
    r3 = m(x, tg)
    (r3[0][0] + r3[1][0]).backward()                                       # no zero_grad -> accumulates
    torch.testing.assert_close(w.grad, 2 * g1, rtol=1e-5, atol=1e-7)
    assert torch.is_tensor(r3[0][1]) and r3[0][1].is_cuda                  # sync_metrics=False keeps metrics on device


def test_eval_mode_loss_uses_running_statistics_leaves_them_alone_and_is_differentiable():
    """ADVICE r1 / VERDICT r3 #7: `model.eval()(images, targets)` — a validation loss, and with gradients frozen-BatchNorm fine-tuning.
    nn.BatchNorm2d in eval mode normalises with the running statistics and does not update them (models/mobilenetv2.py:41-84 under
    .eval()); the oracle in eval mode is the reference for the losses and for every parameter gradient."""
    ref = procedural.fill_state_dict_(net_ref.RefYolo(procedural.VOC_CONFIG)).eval()
    m = _model(train=False)
    x = procedural.images(4, 128, 128, seed=31)
    tg = procedural.targets(4, seed=32, empty_every=4)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        rr = ref(x, tg)
    res = m(x.cuda(), tg)
    for i in range(2):
        np.testing.assert_allclose(np.array([float(torch.as_tensor(v).detach()) for v in res[i]]), np.array([float(v) for v in rr[i]]), rtol=2e-3, atol=1e-5)
    # under no_grad it carries no graph (forward-only plan); with grad mode on the eval-mode losses are differentiable as in the reference
    # (mbv2_yolo.py:157): frozen BatchNorm — the running statistics are constants of the step — checked against the oracle in .eval()
    with torch.no_grad():
        assert not m(x.cuda(), tg)[0][0].requires_grad
    assert res[0][0].requires_grad
    (res[0][0] + res[1][0]).backward()
    rg = ref(x, tg)
    (rg[0][0] + rg[1][0]).backward()
    rp = dict(ref.named_parameters())
    n_cmp = 0
    for k, p in m.named_parameters():
        if rp[k].grad is None:
            assert p.grad is None, k
            continue
        a, b = p.grad.double().cpu(), rp[k].grad.double()
        assert (a - b).norm().item() <= 2e-2 * b.norm().item() + 2e-5, (k, (a - b).norm().item(), b.norm().item())     # |difference|, i.e. direction as well as norm (worst: the stem, 6e-3)
        n_cmp += 1
    assert n_cmp == 202
    for k, v in m.state_dict().items():                      # running_mean / running_var / num_batches_tracked untouched
        assert torch.equal(v, before[k]), k
    # and it differs from the batch-statistics loss of train mode (the bug was silently returning that one)
    m.train()
    tr = m(x.cuda(), tg)
    assert abs(float(tr[0][0].detach()) - float(res[0][0].detach())) > 1e-3 * abs(float(res[0][0].detach()))
    # detections in eval mode still work next to the loss plan
    m.eval()
    assert len(m(x.cuda())) == 4


def test_train_mode_without_targets_decodes_on_batch_statistics_like_the_reference():
    """VERDICT r4 missing #6: the reference decodes + runs NMS in ANY mode (mbv2_yolo.py:158-166); under model.train() its BatchNorm
    layers normalise with the batch statistics and update running_mean / running_var / num_batches_tracked.  Same here: detections
    equal the oracle's pipeline on the same heads, heads equal the oracle model's train-mode heads, buffers move like the oracle's."""
    from oracle import nms_ref, yolo_ref
    m = _model(train=True)
    for hs in m.yolo_losses:
        hs.val_conf = 0.3
    x = procedural.images(4, 128, 128, seed=41)
    det = m(x.cuda())
    plan = m._plans[(4, 128, 128, "traindet")]
    specs = yolo_ref.specs_from_config(procedural.VOC_CONFIG)
    rows = []
    for hi in range(2):
        specs[hi].val_conf = 0.3
        rows.append(yolo_ref.decode_rows(plan.heads[hi].cpu(), specs[hi], [128, 128], layout="nhwc"))
    want = nms_ref.nms_driver(tuple(rows), 20)
    assert len(det) == 4
    for d, r in zip(det, want):
        assert len(d) == len(r)
        np.testing.assert_allclose(d.cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-4)
    # batch statistics were used and the buffers moved exactly like the oracle model's after ONE train-mode forward
    ref1 = procedural.fill_state_dict_(net_ref.RefYolo(procedural.VOC_CONFIG)).train()
    with torch.no_grad():
        ref1(x)
    sd, rsd = m.state_dict(), ref1.state_dict()
    for k in rsd:
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(rsd[k]) == 1, k
        elif "running_" in k:
            a, b = sd[k].double().cpu(), rsd[k].double()
            assert (a - b).abs().max().item() <= 2e-3 * b.abs().max().item() + 1e-6, k
    # eval mode afterwards still runs on the (updated) running statistics
    m.eval()
    assert len(m(x.cuda())) == 4


def test_two_forwards_of_one_shape_may_await_their_backward_like_under_autograd():
    """`a = model(x, t); b = model(x, t); a.backward(); b.backward()`: the second differentiable forward of a shape whose first step still owes
    its backward runs on a second plan (own activations and gradient arena), so both backwards work and their gradients ACCUMULATE in p.grad like
    autograd's do; a graph that is dropped without backward frees its plan again (no third plan, no growth)."""
    m = _model(train=True)
    x = procedural.images(2, 96, 96, seed=33).cuda()
    tg = procedural.targets(2, seed=34, empty_every=0)
    one = m(x, tg)
    (one[0][0] + one[1][0]).backward()
    g1 = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    m.zero_grad(set_to_none=True)
    n_plans = len(m._plans)
    a = m(x, tg)
    b = m(x, tg)                                             # a's backward is still owed: b runs on the second plan
    assert len(m._plans) == n_plans + 1
    for i in range(2):
        assert float(a[i][0]) == float(b[i][0])              # same batch, batch statistics: same losses
    (a[0][0] + a[1][0]).backward()
    (b[0][0] + b[1][0]).backward()
    torch.cuda.synchronize()
    for k, p in m.named_parameters():
        if k in g1:
            d = (p.grad - 2 * g1[k]).abs().max().item()
            assert d <= 1e-5 * (2 * g1[k]).abs().max().item() + 1e-7, (k, d)
    # dropped graphs release their plans: many forwards without backward reuse the two plans
    m.zero_grad(set_to_none=True)
    del a, b, one
    for _ in range(4):
        m(x, tg)
    assert len(m._plans) == n_plans + 1


def test_backward_of_a_step_whose_plan_was_reused_is_refused():
    """With every plan slot of a shape awaiting a backward, a further forward reuses the OLDEST plan; that step's late backward raises instead of
    differentiating through another step's activations."""
    m = _model(train=True)
    x = procedural.images(2, 96, 96, seed=33).cuda()
    tg = procedural.targets(2, seed=34, empty_every=0)
    a = m(x, tg)
    b = m(x, tg)
    c = m(x, tg)                                             # both slots busy: overwrites a's saved activations
    with pytest.raises(RuntimeError, match="another forward"):
        (a[0][0] + a[1][0]).backward()
    (b[0][0] + b[1][0]).backward()
    (c[0][0] + c[1][0]).backward()


def test_training_step_is_bit_deterministic_at_a_size_that_uses_every_kernel_family():
    """bs = 64 at 352x352: the 22x22 layers have M = 30 976 rows, the 44x44 ones 123 904 — large enough for the barrier-free wide-output
    kernel and the stream weight-gradient kernel next to the LDS-DMA, short-reduction and stencil kernels.  Every reduction in the path
    has a fixed order (partial rows + fixed-order combines, no atomics), so the same batch through fresh gradients must give the same
    loss tuple and bit-identical gradients, run after run (this is the check that exposed the wide kernel's first reduction form)."""
    m = _model(train=True)
    x = procedural.images(64, 352, 352, seed=5).cuda()
    tg = procedural.targets(64, seed=6, empty_every=5)
    first = None
    for it in range(6):
        m.zero_grad(set_to_none=True)
        junk = torch.randn(1 << 22, device="cuda")            # allocator / cache churn between the runs
        res = m(x, tg)
        (res[0][0] + res[1][0]).backward()
        torch.cuda.synchronize()
        del junk
        cur = ([float(v.detach()) if torch.is_tensor(v) else float(v) for r in res for v in r], {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        if first is None:
            first = cur
            continue
        assert cur[0] == first[0], (it, cur[0], first[0])
        bad = [k for k in first[1] if not torch.equal(cur[1][k], first[1][k])]
        assert not bad, (it, bad[:8])


def test_train_step_matches_oracle_bs64_352_with_the_benchmark_kernel_families():
    """VERDICT r2 #1a: oracle parity at a batch whose plan is built from the kernels the bs-256 benchmark runs.  At bs 64 / 352x352 the
    22x22 layers have M = 30 976 rows and the 44x44 ones 123 904: past the M >= 8 192 gate of the barrier-free wide-output kernel
    (pwwide.hip) and the M >= 16 384 gate of the stream weight-gradient kernel (pwwgs.hip); the MFMA-bound shapes take the six-product
    bf16 form on pre-cut weight planes (_w6 entry points).  The plan's own call list is asserted to contain all of them, then the same
    weights and batch go through the CPU oracle (oracle/net_ref.py restating models/mobilenetv2.py:63-85, mbv2_yolo.py:137-173,
    yolo_loss.py:77-236): losses within 2e-3, every one of the 202 parameter-gradient norms within 2e-2 (+ 2e-5 floor), as at bs 8."""
    from mobilenet_yolo_pytorch_amd import _lib
    ref = procedural.fill_state_dict_(net_ref.RefYolo(procedural.VOC_CONFIG)).train()
    m = _model(train=True)
    x = procedural.images(64, 352, 352, seed=5)
    tg = procedural.targets(64, seed=6, empty_every=5)
    res = m(x.cuda(), tg)
    (res[0][0] + res[1][0]).backward()
    plan = m._plans[(64, 352, 352, True)]
    routes = plan.kernel_routes()
    fams = {(label, fam) for _fn, label, _shape, fam in routes}
    called = {fn for fn, _l, _s, _f in routes}
    assert ("mny_pw_fwd", _lib.ROUTE_WIDE) in fams and ("mny_pw_fwd", _lib.ROUTE_THIN) in fams and ("mny_pw_fwd", _lib.ROUTE_DMA_X6) in fams, sorted(fams)
    assert ("mny_pw_dgrad_bnred", _lib.ROUTE_WIDE) in fams or ("mny_pw_dgrad_bnred_add", _lib.ROUTE_WIDE) in fams, sorted(fams)
    assert ("mny_pw_wgrad", _lib.ROUTE_WGRAD_STREAM) in fams and ("mny_pw_wgrad", _lib.ROUTE_DMA_X6) in fams, sorted(fams)
    assert "mny_pw_fwd_w6" in called and "mny_pw_dgrad_bnred_w6" in called, sorted(called)
    # every (entry point, kernel family) pair of the bs-256 plan the benchmark times is present in this plan
    big = {(label, _lib.query("mny_pw_route", {"mny_pw_fwd": 0, "mny_pw_wgrad": 2}.get(label, 1), 0, 4 * M, K, N)) for _fn, label, (M, K, N), _f in routes}
    assert big <= fams, sorted(big - fams)
    rr = ref(x, tg)
    (rr[0][0] + rr[1][0]).backward()
    for i in range(2):
        np.testing.assert_allclose(np.array([float(v) for v in res[i]]), np.array([float(v) for v in rr[i]]), rtol=2e-3, atol=1e-5)
    rp = dict(ref.named_parameters())
    n_cmp = 0
    for k, p in m.named_parameters():
        if rp[k].grad is None:
            assert p.grad is None
            continue
        a, b = p.grad.double().norm().item(), rp[k].grad.double().norm().item()
        assert abs(a - b) <= 2e-2 * b + 2e-5, (k, a, b)
        n_cmp += 1
    assert n_cmp == 202
    assert _assert_gradient_tensors_close(m, ref, 5e-2)[0] == 202          # all 202 tensors, direction as well as norm (measured worst 2.1e-2)


def _bench_batch(bs=256, size=352):
    """bench.py's inputs: synthetic.images(bs, size, size, seed=rank) / synthetic.targets(bs, seed=1 + rank, empty_every=16), rank 0."""
    from mobilenet_yolo_pytorch_amd import synthetic
    return synthetic.images(bs, size, size, seed=0), synthetic.targets(bs, seed=1, empty_every=16)


def _mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def test_headline_plan_bs256_352_is_bit_deterministic_and_gives_the_bench_loss():
    """VERDICT r3 #6a: the bs-256 / 352x352 plan ITSELF (grid sizes, split counts and XCD maps at M = 7.9 M rows — only bench.py ran it):
    bench.py's model seed and batch, two fresh-gradient steps bit-identical in the loss tuples and in all 202 gradients, every gradient
    finite, the expand + depthwise unit of the 16->96 @176^2 block on its un-materialised route, and the loss bench.py prints."""
    from mobilenet_yolo_pytorch_amd import synthetic, yolo
    torch.manual_seed(0)
    m = yolo(synthetic.VOC_CONFIG).cuda().train()
    x, tg = _bench_batch()
    x = x.cuda()
    snaps = []
    for it in range(2):
        for p in m.parameters():
            p.grad = None
        res = m(x, tg)
        (res[0][0] + res[1][0]).backward()
        torch.cuda.synchronize()
        snaps.append(([float(v.detach()) if torch.is_tensor(v) else float(v) for r in res for v in r],
                      {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert snaps[0][0] == snaps[1][0]
    assert len(snaps[0][1]) == 202
    bad = [k for k in snaps[0][1] if not torch.equal(snaps[0][1][k], snaps[1][1][k])]
    assert not bad, bad[:8]
    assert all(bool(torch.isfinite(g).all()) for g in snaps[0][1].values())
    loss = snaps[0][0][0] + snaps[0][0][7]
    assert abs(loss - 0.51833) < 5e-5, loss                     # `config.loss` of the bench line (BENCH_r03.json, profiles/r0*_bench_n1.json)
    plan = m._plans[(256, 352, 352, True)]
    names = [c[2] for c in plan.fwd.calls] + [c[2] for c in plan.bwd.calls]
    if os.environ.get("MNY_NO_EXDW") is None:
        assert names.count("mny_exdw_fwd") >= 1 and names.count("mny_exdw_bwd") == names.count("mny_exdw_fwd") == names.count("mny_exdw_stats")
    if os.environ.get("MNY_NO_STEMDW") is None:
        assert names.count("mny_stemdw_bwd") == 1 and "mny_stem_bnwgrad" not in names      # stem + first depthwise unit: one backward pass
    if os.environ.get("MNY_NO_PJBWD") is None:
        assert names.count("mny_pj_bwd") == 6                                              # the six thin project units (32->16 ... 192->32)


@pytest.mark.skipif(_mem_available_gb() < 96.0, reason="the CPU oracle at bs 256 needs ~60 GB of host memory")
@pytest.mark.timeout(900)
def test_headline_plan_bs256_352_matches_the_cpu_oracle():
    """The same bs-256 step through oracle/net_ref.py (the stock torch ops the reference calls + the restated loss: ~20-40 s on the GPU
    box's host cores), bounds as at bs 64: losses 2e-3, every parameter-gradient norm 2e-2 (+ 2e-5)."""
    from mobilenet_yolo_pytorch_amd import synthetic, yolo
    torch.manual_seed(0)
    m = yolo(synthetic.VOC_CONFIG).cuda().train()
    ref = net_ref.RefYolo(procedural.VOC_CONFIG).train()
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    x, tg = _bench_batch()
    res = m(x.cuda(), tg)
    (res[0][0] + res[1][0]).backward()
    rr = ref(x, tg)
    (rr[0][0] + rr[1][0]).backward()
    for i in range(2):
        np.testing.assert_allclose(np.array([float(torch.as_tensor(v).detach()) for v in res[i]]), np.array([float(torch.as_tensor(v).detach()) for v in rr[i]]), rtol=2e-3, atol=1e-5)
    rp = dict(ref.named_parameters())
    n_cmp = 0
    for k, p in m.named_parameters():
        if rp[k].grad is None:
            assert p.grad is None
            continue
        a, b = p.grad.double().norm().item(), rp[k].grad.double().norm().item()
        assert abs(a - b) <= 2e-2 * b + 2e-5, (k, a, b)
        n_cmp += 1
    assert n_cmp == 202
    assert _assert_gradient_tensors_close(m, ref, 5e-2)[0] == 202          # the benchmark's own plan: every gradient tensor (measured worst 1.9e-2)
