"""VOC07 11-point mAP on the GPU (mny_map_eval / mny_eval_pack through the C ABI) against the fixtures captured from the
real reference (tests/golden/map_*.npz) and against the oracle on seeded inputs.  Counts (TP/FP) and the 11 interpolated
precisions are bit-exact; AP / mAP (fp32 means) within 1e-6 of the reference's summation order."""
import os

import numpy as np
import pytest
import torch

from oracle import map_ref

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
KEYS = ("det_boxes", "det_labels", "det_scores", "det_off", "true_boxes", "true_labels", "true_diff", "true_off")


@pytest.fixture(scope="module")
def em():
    assert torch.cuda.is_available()
    from mobilenet_yolo_pytorch_amd import evalmap
    return evalmap


def run(em, case, nc):
    r = em.map_eval(*[torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in case], nc)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in r.items() if not k.startswith("_")}


def check_vs_oracle(em, case, nc):
    r = run(em, case, nc)
    ap, m, tp, fp, p11 = map_ref.calculate_map(*case, nc)
    assert np.array_equal(r["tp"], tp) and np.array_equal(r["fp"], fp)
    assert np.array_equal(r["prec11"], p11)                                     # same fp32 divisions -> bit-exact
    np.testing.assert_allclose(r["ap"], ap, rtol=0, atol=1e-6)
    assert abs(float(r["mean_ap"][0]) - float(m)) < 1e-6
    return r


@pytest.mark.parametrize("name", ["map_small.npz", "map_voc.npz", "map_nodet.npz"])
def test_reference_fixture(em, name):
    z = np.load(os.path.join(G, name))
    r = run(em, [z[k] for k in KEYS], int(z["n_classes"]))
    assert np.array_equal(r["tp"], z["tp"]) and np.array_equal(r["fp"], z["fp"])
    np.testing.assert_allclose(r["ap"], z["ap"], rtol=0, atol=1e-6)
    assert abs(float(r["mean_ap"][0]) - float(z["mean_ap"])) < 1e-6


@pytest.mark.parametrize("seed,n_images,nc", [(0, 1, 2), (1, 7, 3), (2, 40, 21), (3, 150, 21), (4, 64, 81)])
def test_random_vs_oracle(em, seed, n_images, nc):
    from mobilenet_yolo_pytorch_amd import synthetic
    check_vs_oracle(em, synthetic.map_case(n_images, nc, seed=seed), nc)


def test_score_ties_keep_stored_order(em):
    from mobilenet_yolo_pytorch_amd import synthetic
    case = list(synthetic.map_case(30, 5, seed=11))
    case[2] = (np.round(case[2] * 8) / 8).astype(np.float32)                   # 8 distinct scores: ties everywhere
    check_vs_oracle(em, case, 5)


def test_crowded_images_beyond_one_scan_chunk(em):
    """> 1024 detections of one class (several chunks of the block scan), many objects per image."""
    from mobilenet_yolo_pytorch_amd import synthetic
    case = synthetic.map_case(6, 3, seed=12, mean_gt=60.0, det_per_gt=4.0, clutter=300.0)
    assert case[1].size > 3000
    check_vs_oracle(em, case, 3)


def test_labels_outside_the_class_range_are_never_evaluated(em):
    from mobilenet_yolo_pytorch_amd import synthetic
    case = list(synthetic.map_case(20, 6, seed=13))
    dl = case[1].copy()
    dl[::7], dl[1::7], dl[2::7] = 0.0, 2.5, 6.0                                 # background, fractional, == n_classes
    case[1] = dl
    tl = case[5].copy()
    tl[::5] = 0.0
    case[5] = tl
    check_vs_oracle(em, case, 6)


def test_difficult_duplicate_and_degenerate(em):
    tb = np.array([[0, 0, 1, 1], [0, 0, 0, 0], [2, 2, 3, 3], [2, 2, 3, 3]], np.float32)
    td = np.array([1, 0, 0, 0], np.float32)
    db = np.array([[0, 0, 1, 1], [0, 0, 0, 0], [2, 2, 3, 3], [2, 2, 3, 3.1], [2, 2, 3, 3]], np.float32)
    case = (db, np.ones(5, np.float32), np.array([.9, .8, .5, .95, .7], np.float32), np.array([0, 5], np.int32),
            tb, np.ones(4, np.float32), td, np.array([0, 4], np.int32))
    r = check_vs_oracle(em, case, 2)
    # det0 -> difficult (ignored); det1 -> NaN IoU -> FP; det2 -> object 2 (first maximum of the twin objects) TP;
    # det3 -> best is again object 2 -> FP; det4 -> object 2 again -> FP
    assert r["tp"][0] == 1 and r["fp"][0] == 3


def test_no_ground_truth_and_no_detections(em):
    z4, z1, o = np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.array([0, 0, 0], np.int32)
    r = run(em, (z4, z1, z1, o, z4, z1, z1, o), 4)
    assert not r["ap"].any() and r["mean_ap"][0] == 0
    db = np.array([[0, 0, 1, 1]], np.float32)
    r = check_vs_oracle(em, (db, np.ones(1, np.float32), np.array([.5], np.float32), np.array([0, 1, 1], np.int32), z4, z1, z1, o), 3)
    assert r["fp"][0] == 1


def test_image_order_does_not_matter_at_eval_set_size(em):
    """Size-independent property at an evaluation-set size (5 000 images, ~50 k detections): scores are distinct, so
    permuting the images permutes nothing that the metric can see; and every evaluated detection is TP, FP or ignored."""
    from mobilenet_yolo_pytorch_amd import synthetic
    nc = 21
    case = synthetic.map_case(5000, nc, seed=21)
    db, dl, ds, do, tb, tl, td, to = case
    ds = np.linspace(0.02, 0.99, ds.size, dtype=np.float32)[np.argsort(np.argsort(ds, kind="stable"))]   # same ranking, distinct values
    assert np.unique(ds).size == ds.size
    r1 = run(em, (db, dl, ds, do, tb, tl, td, to), nc)
    perm = np.random.RandomState(0).permutation(len(do) - 1)
    cat = lambda a, off: np.concatenate([a[off[i]:off[i + 1]] for i in perm])
    noff = lambda off: np.concatenate([[0], np.cumsum((off[1:] - off[:-1])[perm])]).astype(np.int32)
    r2 = run(em, (cat(db, do), cat(dl, do), cat(ds, do), noff(do), cat(tb, to), cat(tl, to), cat(td, to), noff(to)), nc)
    for k in ("tp", "fp", "prec11", "ap", "mean_ap"):
        assert np.array_equal(r1[k], r2[k]), k
    counts = np.array([(dl == c).sum() for c in range(1, nc)])
    n_easy = np.array([((tl == c) & (td == 0)).sum() for c in range(1, nc)])
    assert (r1["tp"] + r1["fp"] <= counts).all() and (r1["tp"] <= n_easy).all() and r1["tp"].sum() > 1000
    assert 0 < r1["mean_ap"][0] < 1


def test_perfect_detector_scores_one(em):
    from mobilenet_yolo_pytorch_amd import synthetic
    nc = 21
    _, _, _, _, tb, tl, td, to = synthetic.map_case(300, nc, seed=22)
    td = np.zeros_like(td)
    r = run(em, (tb, tl, np.linspace(.9, .1, tl.size, dtype=np.float32), to, tb, tl, td, to), nc)
    present = np.array([(tl == c).any() for c in range(1, nc)])
    assert (r["ap"][present] == 1).all() and not r["fp"].any()


def test_list_interface_matches_reference_fixture(em):
    z = np.load(os.path.join(G, "map_small.npz"))
    nc = int(z["n_classes"])
    sp = lambda a, off: [torch.from_numpy(a[off[i]:off[i + 1]].copy()).cuda() for i in range(len(off) - 1)]
    names = ["background"] + ["c%d" % i for i in range(1, nc)]
    aps, m, tp, fp = em.calculate_mAP(sp(z["det_boxes"], z["det_off"]), sp(z["det_labels"], z["det_off"]), sp(z["det_scores"], z["det_off"]),
                                      sp(z["true_boxes"], z["true_off"]), sp(z["true_labels"], z["true_off"]), sp(z["true_diff"], z["true_off"]), names)
    assert list(aps) == names[1:] and abs(m - float(z["mean_ap"])) < 1e-6
    assert [tp[n] for n in names[1:]] == z["tp"].tolist() and [fp[n] for n in names[1:]] == z["fp"].tolist()
    np.testing.assert_allclose([aps[n] for n in names[1:]], z["ap"], atol=1e-6)


def test_evaluator_from_detector_rows(em):
    """train.py:371-385 glue on the device: [k,7] rows + [t,5] targets -> same answer as the oracle's eval_pack + calculate_map."""
    r = np.random.RandomState(5)
    nc, n_img = 21, 25
    ev = em.Evaluator(["background"] + ["c%d" % i for i in range(1, nc)])
    rows_all, tg_all, do, to = [], [], [0], [0]
    for i in range(n_img):
        t = r.randint(0, 4)
        cxy, wh = 0.2 + 0.6 * r.rand(t, 2), 0.1 + 0.3 * r.rand(t, 2)
        tg = np.concatenate((r.randint(1, nc, (t, 1)), cxy, wh), 1).astype(np.float32)
        k = 0 if i == 3 else t * 2 + r.randint(0, 5)
        rows = np.zeros((k, 7), np.float32)
        for j in range(k):
            if t and j < 2 * t:
                g = tg[j % t]
                rows[j, :4] = [g[1] - g[3] / 2, g[2] - g[4] / 2, g[1] + g[3] / 2, g[2] + g[4] / 2] + r.randn(4) * 0.02
                rows[j, 6] = g[0] - 1
            else:
                c, s = 0.2 + 0.6 * r.rand(2), 0.1 + 0.3 * r.rand(2)
                rows[j, :4] = np.concatenate((c - s / 2, c + s / 2))
                rows[j, 6] = r.randint(0, nc - 1)
            rows[j, 4:6] = 0.2 + 0.8 * r.rand(2)
        ev.add([torch.from_numpy(rows).cuda() if k else None], [tg])
        rows_all.append(rows); tg_all.append(tg); do.append(do[-1] + k); to.append(to[-1] + t)
    aps, m, tp, fp = ev.compute()
    db, dl, ds, tb, tl, td = map_ref.eval_pack(np.concatenate(rows_all), np.concatenate(tg_all))
    ap_o, m_o, tp_o, fp_o, _ = map_ref.calculate_map(db, dl, ds, np.array(do, np.int32), tb, tl, td, np.array(to, np.int32), nc)
    assert list(tp.values()) == tp_o.tolist() and list(fp.values()) == fp_o.tolist() and sum(tp.values()) > 5
    np.testing.assert_allclose(list(aps.values()), ap_o, atol=1e-6)
    assert abs(m - float(m_o)) < 1e-6
    assert ev.gt_box == to[-1] and ev.pred_box == do[-1]
    assert em.adjust_confidence(10, 31, .1) == pytest.approx(.11)


def test_detector_to_map_end_to_end(em):
    """The evaluation loop of train.py:359-421 on the device: eval-mode forward -> decode -> NMS -> Evaluator, scored
    against the oracle fed with the very detections the GPU produced (so the check is the mAP stage, not the network)."""
    from mobilenet_yolo_pytorch_amd import synthetic, yolo
    from oracle import procedural
    torch.manual_seed(0)
    m = yolo(procedural.VOC_CONFIG)
    procedural.fill_state_dict_(m)
    m = m.cuda().eval()
    for hs in m.yolo_losses:
        hs.val_conf = 0.3
    names = ["background"] + ["c%d" % i for i in range(1, 21)]
    ev = em.Evaluator(names)
    rows_all, tg_all, do, to = [], [], [0], [0]
    for b in range(2):
        x = procedural.images(4, 96, 96, seed=30 + b).cuda()
        tg = synthetic.targets(4, seed=40 + b, empty_every=3)
        det = m(x)
        ev.add(det, tg)
        for d, t in zip(det, tg):
            rows_all.append(d.cpu().numpy().reshape(-1, 7)); tg_all.append(t.numpy().reshape(-1, 5))
            do.append(do[-1] + len(rows_all[-1])); to.append(to[-1] + len(tg_all[-1]))
    assert do[-1] > 20
    aps, mAP, tp, fp = ev.compute()
    db, dl, ds, tb, tl, td = map_ref.eval_pack(np.concatenate(rows_all), np.concatenate(tg_all))
    ap_o, m_o, tp_o, fp_o, _ = map_ref.calculate_map(db, dl, ds, np.array(do, np.int32), tb, tl, td, np.array(to, np.int32), 21)
    assert list(tp.values()) == tp_o.tolist() and list(fp.values()) == fp_o.tolist()
    np.testing.assert_allclose(list(aps.values()), ap_o, atol=1e-6)
    assert abs(mAP - float(m_o)) < 1e-6 and sum(fp.values()) > 0
