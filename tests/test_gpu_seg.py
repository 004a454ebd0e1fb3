"""BDD100K config on the GPU (detection + drivable-area segmentation head): mny_seg_loss / mny_seg_sigmoid through the
C ABI against the real reference's fixtures (tools/gen_golden_seg.py) and the oracle, and the whole network with the
third loss (train step: losses, seg outputs, per-parameter gradients — every parameter trains now; eval: (dets, seg map))."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from oracle import net_ref, procedural, seg_ref

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _seg_loss_gpu(x_nchw, t_nhwc):
    from mobilenet_yolo_pytorch_amd import _lib
    head = x_nchw.permute(0, 2, 3, 1).contiguous().cuda()
    t = t_nhwc.contiguous().cuda()
    out3, dhead = torch.zeros(3, device="cuda"), torch.zeros_like(head)
    ws = torch.empty(max(_lib.query("mny_seg_loss_ws_bytes", head.numel()), 8), device="cuda", dtype=torch.uint8)
    p = lambda a: ctypes.c_void_p(a.data_ptr())
    _lib.call("mny_seg_loss", p(head), p(t), head.numel(), p(out3), p(dhead), p(ws), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return out3.cpu().numpy(), dhead.permute(0, 3, 1, 2).cpu().numpy()


def test_seg_loss_reference_tables():
    z = np.load(os.path.join(G, "seg_loss.npz"))
    for tag in "ab":
        out3, dx = _seg_loss_gpu(torch.from_numpy(z["x_" + tag]), torch.from_numpy(z["t_" + tag]))
        np.testing.assert_allclose(out3, z["res_" + tag], rtol=2e-6, equal_nan=True)       # fp32 sigmoid, fp64 sums
        np.testing.assert_allclose(dx, z["dx_" + tag], rtol=1e-5, atol=1e-9)
    assert np.isnan(z["res_b"][1])


def test_seg_loss_large_vs_oracle_and_deterministic():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(16, 2, 44, 44, generator=g) * 3
    t = (torch.rand(16, 44, 44, 2, generator=g) > 0.7).float() * torch.rand(16, 44, 44, 2, generator=g)
    xr = x.clone().requires_grad_(True)
    loss, obj, noobj = seg_ref.seg_loss(xr, t)
    loss.backward()
    a3, adx = _seg_loss_gpu(x, t)
    b3, bdx = _seg_loss_gpu(x, t)
    assert np.array_equal(a3, b3) and np.array_equal(adx, bdx)
    np.testing.assert_allclose(a3, [loss.item(), obj, noobj], rtol=1e-5)
    np.testing.assert_allclose(adx, xr.grad.numpy(), rtol=1e-5, atol=1e-10)


def test_seg_sigmoid_eval_branch():
    from mobilenet_yolo_pytorch_amd import _lib
    z = np.load(os.path.join(G, "seg_loss.npz"))
    x = torch.from_numpy(z["x_a"])
    head = x.permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.zeros(x.shape[1], x.shape[2], x.shape[3], device="cuda")
    p = lambda a: ctypes.c_void_p(a.data_ptr())
    _lib.call("mny_seg_sigmoid", p(head), x.shape[2], x.shape[3], x.shape[1], p(out), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    np.testing.assert_allclose(out.cpu().numpy(), z["eval_a"], rtol=2e-6)


def _bdd(act_dtype=torch.float32, sync=True):
    from mobilenet_yolo_pytorch_amd import yolo
    man = json.load(open(os.path.join(G, "state_keys_bdd100k.json")))
    torch.manual_seed(0)
    m = yolo(man["config"], sync_metrics=sync, act_dtype=act_dtype)
    procedural.fill_state_dict_(m)
    return man["config"], m.cuda()


def test_bdd_train_step_matches_reference_fixture():
    z = np.load(os.path.join(G, "seg_net_train.npz"))
    names = json.load(open(os.path.join(G, "seg_net_names.json")))
    cfg, m = _bdd()
    m.train()
    x = procedural.images(4, 128, 128, seed=21).cuda()
    tg = list(torch.split(torch.from_numpy(z["t_all"]), z["t_counts"].tolist()))
    res, seg_out = m(x, tg, torch.from_numpy(z["seg_maps"]))
    (sum(r[0] for r in res) + seg_out[0]).backward()
    assert isinstance(seg_out[1], float) and isinstance(seg_out[2], float) and seg_out[0].requires_grad
    np.testing.assert_allclose([float(seg_out[0]), seg_out[1], seg_out[2]], z["seg_out"], rtol=2e-3)
    plan = m._plans[(4, 128, 128, True)]
    out2 = plan.seg_head.permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(out2 - z["out2"]).max() <= 2e-3 * np.abs(z["out2"]).max()
    for i in range(2):
        np.testing.assert_allclose(np.array([float(v) for v in res[i]]), z["tuple%d" % i], rtol=2e-3, atol=1e-5)
    params = dict(m.named_parameters())
    assert list(params) == names["params"]
    assert [k for k, p in params.items() if p.grad is None] == []               # with the seg loss every parameter trains
    gn = np.array([p.grad.double().norm().item() for p in params.values()])
    np.testing.assert_allclose(gn, z["gnorm"], rtol=2e-2, atol=2e-5)
    for key, ref in (("seg_headS16.3.weight", "g_seghead_w"), ("seg_headS16.3.bias", "g_seghead_b"),
                     ("seg_conv_for_S16.0.conv.weight", "g_segconv_dw"), ("backbone.features.0.0.weight", "g_stem")):
        got = params[key].grad.cpu().numpy()
        assert np.abs(got - z[ref]).max() <= 2e-2 * np.abs(z[ref]).max(), key


def test_bdd_train_step_vs_oracle_352():
    cfg, m = _bdd()
    m.train()
    ref = procedural.fill_state_dict_(net_ref.RefYolo(cfg)).train()
    x = procedural.images(4, 352, 352, seed=31)
    tg = procedural.targets(4, num_classes=cfg["yolo"]["num_classes"], seed=8, empty_every=3)
    r = np.random.RandomState(4)
    sm = torch.from_numpy((r.rand(4, 22, 22, 2) * (r.rand(4, 22, 22, 2) > 0.5)).astype(np.float32))
    res, seg_out = m(x.cuda(), tg, sm.cuda())
    (sum(q[0] for q in res) + seg_out[0]).backward()
    rres, rseg = ref(x, [t.clone() for t in tg], sm)
    (sum(q[0] for q in rres) + rseg[0]).backward()
    np.testing.assert_allclose([float(seg_out[0]), seg_out[1], seg_out[2]], [float(rseg[0]), rseg[1], rseg[2]], rtol=2e-3)
    rp = dict(ref.named_parameters())
    for k, p in m.named_parameters():
        a, b = p.grad.double().norm().item(), rp[k].grad.double().norm().item()
        assert abs(a - b) <= 2e-2 * b + 2e-5, (k, a, b)


def test_bdd_gradient_scales_with_upstream_weight():
    """loss = det + 3 * seg: the seg branch's gradients scale by 3 (g_scale of the third loss)."""
    cfg, m = _bdd()
    m.train()
    x = procedural.images(2, 96, 96, seed=5).cuda()
    tg = procedural.targets(2, num_classes=7, seed=2, empty_every=0)
    sm = torch.rand(2, 6, 6, 2, generator=torch.Generator().manual_seed(1))
    grads = []
    for wgt in (1.0, 3.0):
        for p in m.parameters():
            p.grad = None
        procedural.fill_state_dict_(m)
        res, seg_out = m(x, tg, sm)
        (sum(q[0] for q in res) + wgt * seg_out[0]).backward()
        grads.append(dict(m.named_parameters())["seg_headS16.3.weight"].grad.clone())
    torch.testing.assert_close(grads[1], 3 * grads[0], rtol=1e-4, atol=1e-9)


def test_bdd_eval_matches_reference_fixture():
    z = np.load(os.path.join(G, "seg_net_eval.npz"))
    cfg, m = _bdd()
    m.eval()
    for hs in m.yolo_losses:
        hs.val_conf = 0.3
    det, seg = m(procedural.images(2, 96, 96, seed=22).cuda())
    assert isinstance(seg, np.ndarray) and seg.shape == (2, 6, 6)
    np.testing.assert_allclose(seg, z["seg"], rtol=2e-3, atol=1e-4)
    assert len(det) == 2 and all(abs(len(d) - c) <= 3 for d, c in zip(det, z["det_counts"]))


def test_bdd_missing_seg_maps_raises():
    cfg, m = _bdd()
    m.train()
    with pytest.raises(ValueError, match="seg_maps"):
        m(procedural.images(2, 96, 96, seed=5).cuda(), procedural.targets(2, num_classes=7, seed=2, empty_every=0))


def test_bdd_bf16_storage_trains():
    cfg, m = _bdd(act_dtype=torch.bfloat16)
    m.train()
    x = procedural.images(4, 128, 128, seed=21).cuda()
    z = np.load(os.path.join(G, "seg_net_train.npz"))
    tg = list(torch.split(torch.from_numpy(z["t_all"]), z["t_counts"].tolist()))
    res, seg_out = m(x, tg, torch.from_numpy(z["seg_maps"]))
    (sum(r[0] for r in res) + seg_out[0]).backward()
    np.testing.assert_allclose(float(seg_out[0]), z["seg_out"][0], rtol=5e-2)
    g = dict(m.named_parameters())["seg_headS16.3.weight"].grad
    assert torch.isfinite(g).all() and g.abs().max() > 0


def test_seg_loss_properties_at_full_size():
    """bs=256 BDD-shaped seg head (26x26x2): loss is 0 with zero gradient when sigmoid(head) == truth; shifting every logit
    up raises both monitoring means; the two selections partition the tensor."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(256, 2, 26, 26, generator=g)
    t = torch.sigmoid(x).permute(0, 2, 3, 1).contiguous()
    out3, dx = _seg_loss_gpu(x, t)
    assert out3[0] < 1e-12 and np.abs(dx).max() < 1e-9
    t2 = (torch.rand(256, 26, 26, 2, generator=g) > 0.5).float()
    a, _ = _seg_loss_gpu(x, t2)
    b, _ = _seg_loss_gpu(x + 1.0, t2)
    assert b[1] > a[1] and b[2] > a[2]
    s = torch.sigmoid(x).permute(0, 2, 3, 1)
    n_obj = int((t2 >= 0.5).sum())
    tot = (a[1] * n_obj + a[2] * (t2.numel() - n_obj)) / t2.numel()
    assert abs(tot - float(s.mean())) < 1e-5
