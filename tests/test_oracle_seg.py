"""oracle/seg_ref.py + the seg branch of oracle/net_ref.py against fixtures from the REAL reference
(tools/gen_golden_seg.py): SegLoss alone (value, pass-through gradient, empty-selection NaN, eval branch) and the
BDD100K-config network (one train step with seg maps, one eval forward)."""
import json
import os

import numpy as np
import torch

from oracle import net_ref, procedural, seg_ref

G = os.path.join(os.path.dirname(__file__), "golden")


def test_seg_loss_tables():
    z = np.load(os.path.join(G, "seg_loss.npz"))
    for tag in "ab":
        x = torch.from_numpy(z["x_" + tag]).requires_grad_(True)
        loss, obj, noobj = seg_ref.seg_loss(x, torch.from_numpy(z["t_" + tag]))
        loss.backward()
        np.testing.assert_allclose([loss.item(), obj, noobj], z["res_" + tag], rtol=1e-6, equal_nan=True)
        np.testing.assert_allclose(x.grad.numpy(), z["dx_" + tag], rtol=1e-6, atol=1e-9)
    assert np.isnan(z["res_b"][1])                                               # the fixture holds the empty-selection case
    np.testing.assert_allclose(seg_ref.seg_eval(torch.from_numpy(z["x_a"])), z["eval_a"], rtol=1e-6)
    # the gradient is 2*0.05*(sigmoid - t)/numel: no sigma' factor (seg_loss.py:24-32)
    x = torch.from_numpy(z["x_a"])
    want = 0.1 * (torch.sigmoid(x) - torch.from_numpy(z["t_a"]).permute(0, 3, 1, 2)) / x.numel()
    np.testing.assert_allclose(z["dx_a"], want.numpy(), rtol=1e-5, atol=1e-9)


def _bdd():
    man = json.load(open(os.path.join(G, "state_keys_bdd100k.json")))
    return man["config"], procedural.fill_state_dict_(net_ref.RefYolo(man["config"]))


def test_bdd_train_step_matches_reference():
    z = np.load(os.path.join(G, "seg_net_train.npz"))
    names = json.load(open(os.path.join(G, "seg_net_names.json")))
    cfg, m = _bdd()
    m.train()
    x = procedural.images(4, 128, 128, seed=21)
    tg = list(torch.split(torch.from_numpy(z["t_all"]), z["t_counts"].tolist()))
    res, seg_out = m(x, tg, torch.from_numpy(z["seg_maps"]))
    (sum(r[0] for r in res) + seg_out[0]).backward()
    np.testing.assert_allclose([float(seg_out[0]), seg_out[1], seg_out[2]], z["seg_out"], rtol=1e-4)
    np.testing.assert_allclose(m.out2.detach().numpy(), z["out2"], rtol=1e-3, atol=1e-4)
    for i in range(2):
        np.testing.assert_allclose(np.array([float(v) for v in res[i]]), z["tuple%d" % i], rtol=1e-4, atol=1e-6)
    params = dict(m.named_parameters())
    assert list(params) == names["params"] and names["grad_none"] == []          # with a seg loss every parameter trains
    gn = np.array([p.grad.double().norm().item() for p in params.values()])
    np.testing.assert_allclose(gn, z["gnorm"], rtol=2e-3, atol=2e-5)      # floor: tensors whose true gradient is 0 (conv bias-like terms in front of a BN)
    np.testing.assert_allclose(m.seg_headS16[3].weight.grad.numpy(), z["g_seghead_w"], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(m.seg_headS16[3].bias.grad.numpy(), z["g_seghead_b"], rtol=1e-3, atol=1e-8)


def test_bdd_eval_matches_reference():
    z = np.load(os.path.join(G, "seg_net_eval.npz"))
    cfg, m = _bdd()
    m.eval()
    for s in m.specs:
        s.val_conf = 0.3
    with torch.no_grad():
        det, seg = m(procedural.images(2, 96, 96, seed=22))
    np.testing.assert_allclose(seg, z["seg"], rtol=1e-4, atol=1e-5)
    assert seg.shape == (cfg["seg"]["num_classes"], 6, 6)
    assert [len(d) for d in det] == z["det_counts"].tolist()
