"""oracle/net_ref.py vs the real reference (fixtures from tools/gen_golden.py).  CPU only."""
import json
import os

import numpy as np
import torch

from oracle import net_ref, procedural

G = os.path.join(os.path.dirname(__file__), "golden")


def _model():
    torch.manual_seed(0)
    m = net_ref.RefYolo(procedural.VOC_CONFIG)
    return procedural.fill_state_dict_(m)


def test_state_dict_keys_and_shapes():
    man = json.load(open(os.path.join(G, "state_keys_voc.json")))
    sd = net_ref.RefYolo(man["config"]).state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == man["keys"]
    manb = json.load(open(os.path.join(G, "state_keys_bdd100k.json")))
    sdb = net_ref.RefYolo(manb["config"]).state_dict()
    assert sorted([k, list(v.shape)] for k, v in sdb.items()) == sorted(manb["keys"])


def test_eval_heads_and_detections():
    z = np.load(os.path.join(G, "net_eval.npz"))
    m = _model().eval()
    for s in m.specs:
        s.val_conf = 0.3
    for tag, (n, s) in {"a": (2, 96), "b": (1, 352)}.items():
        x = procedural.images(n, s, s, seed=10)
        with torch.no_grad():
            o0, o1 = m.heads(x)
            det = m(x)
        np.testing.assert_allclose(o0.numpy(), z["out0_" + tag], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(o1.numpy(), z["out1_" + tag], rtol=1e-4, atol=1e-5)
        assert [len(d) for d in det] == z["det_counts_" + tag].tolist()
        np.testing.assert_allclose(torch.cat(det).numpy(), z["det_rows_" + tag], rtol=1e-4, atol=1e-5)


def test_train_step_matches_reference():
    z = np.load(os.path.join(G, "net_train.npz"))
    names = json.load(open(os.path.join(G, "net_train_names.json")))
    m = _model().train()
    x = procedural.images(4, 128, 128, seed=11)
    tg = list(torch.split(torch.from_numpy(z["t_all"]), z["t_counts"].tolist()))
    res = m(x, tg)
    (res[0][0] + res[1][0]).backward()
    for i in range(2):
        np.testing.assert_allclose(np.array([float(v) for v in res[i]]), z["tuple%d" % i], rtol=1e-4, atol=1e-6)
    got_names = [k for k, _ in m.named_parameters()]
    assert got_names == names["params"]
    assert [k for k, p in m.named_parameters() if p.grad is None] == names["grad_none"]      # Q10
    gn = np.array([-1.0 if p.grad is None else p.grad.double().norm().item() for _, p in m.named_parameters()])
    np.testing.assert_allclose(gn, z["gnorm"], rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(m.backbone.features[0][0].weight.grad.numpy(), z["g_stem"], rtol=2e-3, atol=1e-6)
    sd = m.state_dict()
    rs = np.array([sd[k].double().norm().item() for k in names["running"]])
    np.testing.assert_allclose(rs, z["rs_norm"], rtol=1e-5)
