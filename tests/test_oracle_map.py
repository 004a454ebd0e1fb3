"""oracle/map_ref.py (restated utils/eval_mAP.py) against fixtures produced by the REAL reference
(tools/gen_golden_map.py -> tests/golden/map_*.npz), plus known-answer cases worked by hand."""
import os

import numpy as np
import pytest
import torch

from oracle import map_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KEYS = ("det_boxes", "det_labels", "det_scores", "det_off", "true_boxes", "true_labels", "true_diff", "true_off")


def load(name):
    z = np.load(os.path.join(GOLD, name))
    return z, [z[k] for k in KEYS], int(z["n_classes"])


@pytest.mark.parametrize("name", ["map_small.npz", "map_voc.npz", "map_nodet.npz"])
def test_matches_reference(name):
    z, case, nc = load(name)
    ap, m, tp, fp, _ = map_ref.calculate_map(*case, nc)
    assert np.array_equal(tp, z["tp"]) and np.array_equal(fp, z["fp"])          # counts: exact
    np.testing.assert_allclose(ap, z["ap"], rtol=0, atol=1e-6)                  # fp32 mean of 11 (summation order)
    assert abs(float(m) - float(z["mean_ap"])) < 1e-6


def test_flags_match_reference():
    z, case, nc = load("map_small.npz")
    db, dl, ds, do, tb, tl, td, to = case
    tpf, fpf = np.zeros(len(dl), np.float32), np.zeros(len(dl), np.float32)
    for i in range(len(do) - 1):
        a, b, c, d = do[i], do[i + 1], to[i], to[i + 1]
        for cls in range(1, nc):
            tp, fp, _, _ = map_ref.single_image_recall(cls, tl[c:d], dl[a:b], tb[c:d], td[c:d], db[a:b], ds[a:b])
            idx = np.nonzero(dl[a:b] == cls)[0] + a
            tpf[idx], fpf[idx] = tp, fp
    assert np.array_equal(tpf, z["tp_flags"]) and np.array_equal(fpf, z["fp_flags"])
    assert tpf.sum() > 5 and ((tpf == 0) & (fpf == 0)).sum() > 0               # fixture exercises the "difficult: ignored" branch


def test_recall_thresholds_are_torch_arange():
    assert np.array_equal(map_ref.RECALL_T, torch.arange(start=0, end=1.1, step=.1).numpy())


def test_known_answer():
    """One class, one image, 2 objects; detections (stored order): hit A (0.9), hit A again (0.8), miss (0.7), hit B (0.6).
    sorted = stored; TP,FP = 1,0 / 1,1 / 1,2 / 2,2; precision 1, .5, .333, .5; recall .5 .5 .5 1
    -> p11 = 1 for t <= 0.5, 0.5 for t in 0.6..1.0 -> AP = (6*1 + 5*0.5)/11."""
    tb = np.array([[0, 0, 1, 1], [2, 2, 3, 3]], np.float32)
    db = np.array([[0, 0, 1, 1], [0, 0, 1, .9], [5, 5, 6, 6], [2, 2, 3, 3]], np.float32)
    one = np.ones
    ap, m, tp, fp, p11 = map_ref.calculate_map(db, one(4, np.float32), np.array([.9, .8, .7, .6], np.float32), np.array([0, 4], np.int32),
                                               tb, one(2, np.float32), np.zeros(2, np.float32), np.array([0, 2], np.int32), 2)
    assert tp[0] == 2 and fp[0] == 2
    assert np.allclose(p11[0], [1] * 6 + [.5] * 5)
    assert abs(ap[0] - 8.5 / 11) < 1e-6 and abs(m - 8.5 / 11) < 1e-6


def test_stored_order_decides_the_true_positive():
    """Two detections on one object: the FIRST STORED one is the TP even when its score is lower (eval_mAP.py:33,55-59)."""
    tb = np.array([[0, 0, 1, 1]], np.float32)
    db = np.array([[0, 0, 1, .8], [0, 0, 1, 1]], np.float32)
    args = (np.ones(2, np.float32), np.array([.3, .9], np.float32), np.array([0, 2], np.int32), tb, np.ones(1, np.float32),
            np.zeros(1, np.float32), np.array([0, 1], np.int32), 2)
    ap, _, tp, fp, p11 = map_ref.calculate_map(db, *args)
    # sorted by score: [FP(.9), TP(.3)] -> precision 0, .5; recall 0, 1 -> p11 = .5 everywhere
    assert tp[0] == 1 and fp[0] == 1 and np.allclose(p11[0], .5)


def test_difficult_and_degenerate():
    tb = np.array([[0, 0, 1, 1], [0, 0, 0, 0]], np.float32)
    td = np.array([1, 0], np.float32)
    db = np.array([[0, 0, 1, 1], [0, 0, 0, 0]], np.float32)                     # 2nd: 0/0 IoU with the degenerate object -> NaN -> FP
    ap, _, tp, fp, _ = map_ref.calculate_map(db, np.ones(2, np.float32), np.array([.9, .8], np.float32), np.array([0, 2], np.int32),
                                             tb, np.ones(2, np.float32), td, np.array([0, 2], np.int32), 2)
    assert tp[0] == 0 and fp[0] == 1 and ap[0] == 0


def test_eval_pack_and_adjust_confidence():
    rows = np.array([[.1, .2, .3, .4, .5, .6, 7]], np.float32)
    tg = np.array([[3, .5, .5, .2, .4]], np.float32)
    db, dl, ds, tb, tl, td = map_ref.eval_pack(rows, tg)
    assert dl[0] == 8 and ds[0] == np.float32(.5) * np.float32(.6) and np.array_equal(db[0], rows[0, :4])
    assert np.allclose(tb[0], [.4, .3, .6, .7]) and tl[0] == 3 and td[0] == 0
    assert map_ref.adjust_confidence(10, 31, .1) == pytest.approx(.11)
    assert map_ref.adjust_confidence(10, 19, .1) == pytest.approx(.09)
    assert map_ref.adjust_confidence(10, 25, .1) == .1 and map_ref.adjust_confidence(10, 5, .01) == .01
