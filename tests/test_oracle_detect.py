"""Oracle pinning: oracle/yolo_ref.py vs fixtures captured from the real reference
(tools/gen_golden.py).  CPU only."""
import json
import os

import numpy as np
import torch

from oracle import procedural, yolo_ref

G = os.path.join(os.path.dirname(__file__), "golden")


def _targets(z):
    return list(torch.split(torch.from_numpy(z["t_all"]), z["t_counts"].tolist()))


def test_iou_and_ciou_tables():
    z = np.load(os.path.join(G, "iou_tables.npz"))
    a, b = torch.from_numpy(z["a"]), torch.from_numpy(z["b"])
    iou = yolo_ref.pair_iou(a, b).numpy()
    assert np.array_equal(np.isnan(iou), np.isnan(z["iou"]))          # 0/0 -> NaN like the reference
    assert np.array_equal(np.nan_to_num(iou), np.nan_to_num(z["iou"]))  # bit-exact
    for i in range(a.shape[0]):
        for j in range(b.shape[0]):
            t, u = yolo_ref.ciou_pair(a[i:i + 1], b[j:j + 1])
            got = np.array([t.item(), u.item()], np.float32)
            np.testing.assert_array_equal(np.isnan(got), np.isnan(z["ciou"][i, j]))
            np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(z["ciou"][i, j]), rtol=0, atol=1e-7)


def test_q1_box_weights_cancel():
    z = np.load(os.path.join(G, "iou_tables.npz"))
    x = torch.from_numpy(z["q1_x"])
    ref = float(z["q1"])
    assert abs(float(((x - 1) ** 2).sum()) - ref) < 1e-6


def test_loss_tuple_and_grad_match_reference():
    z = np.load(os.path.join(G, "loss_decode.npz"))
    specs = yolo_ref.specs_from_config(procedural.VOC_CONFIG)
    tg = _targets(z)
    for hi in range(2):
        head = torch.from_numpy(z["head%d" % hi]).clone().requires_grad_(True)
        res = yolo_ref.loss_forward(head, tg, specs[hi], [352, 352])
        res[0].backward()
        got = np.array([float(v) for v in res])
        np.testing.assert_allclose(got, z["tuple%d" % hi], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(head.grad.numpy(), z["grad%d" % hi], rtol=1e-5, atol=1e-9)
        assert z["tuple%d" % hi][6] > 0                                # fixtures do contain positives


def test_loss_layout_nhwc_equals_nchw():
    z = np.load(os.path.join(G, "loss_decode.npz"))
    specs = yolo_ref.specs_from_config(procedural.VOC_CONFIG)
    head = torch.from_numpy(z["head1"])
    a = yolo_ref.loss_forward(head, _targets(z), specs[1], [352, 352])
    b = yolo_ref.loss_forward(head.permute(0, 2, 3, 1).contiguous(), _targets(z), specs[1], [352, 352], layout="nhwc")
    assert float(a[0]) == float(b[0])


def test_decode_rows_bit_exact():
    z = np.load(os.path.join(G, "loss_decode.npz"))
    specs = yolo_ref.specs_from_config(procedural.VOC_CONFIG)
    for hi in range(2):
        head = torch.from_numpy(z["head%d" % hi])
        for vc in (1, 3, 5):
            specs[hi].val_conf = vc / 10
            rows = yolo_ref.decode_rows(head, specs[hi], [352, 352])
            assert [len(r) for r in rows] == z["dec%d_%d_counts" % (hi, vc)].tolist()
            assert np.array_equal(torch.cat(rows).numpy(), z["dec%d_%d_rows" % (hi, vc)])   # Q13: bit-identical


def test_state_key_manifest_counts():
    for name, n in (("voc", 430), ("bdd100k", 450)):
        m = json.load(open(os.path.join(G, "state_keys_%s.json" % name)))
        assert len(m["keys"]) == n
