"""Detection math on the GPU (loss fwd+bwd, decode+compaction, per-class NMS) through the C ABI,
against the oracle and the fixtures captured from the real reference."""
import os

import numpy as np
import pytest
import torch

from oracle import nms_ref, procedural, yolo_ref

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from mobilenet_yolo_pytorch_amd import ops as o
    return o


def _pack(targets):
    off = np.zeros(len(targets) + 1, np.int32)
    off[1:] = np.cumsum([len(t) for t in targets])
    allt = torch.cat([t.reshape(-1, 5) for t in targets]) if off[-1] else torch.zeros(0, 5)
    if allt.shape[0] == 0:
        allt = torch.zeros(1, 5)
    return allt.float().contiguous().cuda(), torch.from_numpy(off).cuda()


def _head_args(ops, spec, N, g, img=352):
    anchors = torch.tensor([(aw / img, ah / img) for aw, ah in spec.anchors], dtype=torch.float32).cuda()
    mask = torch.tensor(spec.mask, dtype=torch.int32).cuda()
    hp = ops.make_head(N, g, len(spec.mask), spec.num_classes, len(spec.anchors), spec.ignore_thresh,
                       spec.iou_thresh, spec.iou_weighting)
    return anchors, mask, hp


def _run_loss(ops, head_nchw, targets, spec, img=352):
    N, _, g, _ = head_nchw.shape
    anchors, mask, hp = _head_args(ops, spec, N, g, img)
    tg, off = _pack(targets)
    h = head_nchw.permute(0, 2, 3, 1).contiguous().cuda()
    out7, dhead = ops.yolo_loss(h, tg, off, anchors, mask, hp)
    return out7.cpu().double().numpy(), dhead.cpu().permute(0, 3, 1, 2).contiguous()


def test_loss_matches_reference_fixture(ops):
    """fp32 tolerance: loss/metrics 2e-5 relative, dL/dhead 1e-4 relative + 1e-8 absolute
    (GPU expf/atanf differ from the CPU libm by a few ulp)."""
    z = np.load(os.path.join(G, "loss_decode.npz"))
    specs = yolo_ref.specs_from_config(procedural.VOC_CONFIG)
    tg = list(torch.split(torch.from_numpy(z["t_all"]), z["t_counts"].tolist()))
    for hi in range(2):
        out7, dhead = _run_loss(ops, torch.from_numpy(z["head%d" % hi]), tg, specs[hi])
        np.testing.assert_allclose(out7, z["tuple%d" % hi], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(dhead.numpy(), z["grad%d" % hi], rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("N,g,seed", [(8, 11, 0), (16, 22, 1), (3, 4, 2), (2, 13, 3)])
def test_loss_matches_oracle_random(ops, N, g, seed):
    specs = yolo_ref.specs_from_config(procedural.VOC_CONFIG)
    hi = seed % 2
    gen = torch.Generator().manual_seed(seed)
    head = torch.randn(N, 75, g, g, generator=gen) * 0.7
    tg = procedural.targets(N, seed=seed + 10, empty_every=3, boxes_per_image=1 + seed)
    hr = head.clone().requires_grad_(True)
    ref = yolo_ref.loss_forward(hr, tg, specs[hi], [352, 352])
    ref[0].backward()
    out7, dhead = _run_loss(ops, head, tg, specs[hi])
    np.testing.assert_allclose(out7, np.array([float(v) for v in ref]), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(dhead.numpy(), hr.grad.numpy(), rtol=1e-4, atol=1e-8)


def test_loss_all_images_empty(ops):
    specs = yolo_ref.specs_from_config(procedural.VOC_CONFIG)
    head = torch.randn(2, 75, 11, 11)
    tg = [torch.zeros(0, 5), torch.zeros(0, 5)]
    hr = head.clone().requires_grad_(True)
    ref = yolo_ref.loss_forward(hr, tg, specs[0], [352, 352])
    ref[0].backward()
    out7, dhead = _run_loss(ops, head, tg, specs[0])
    np.testing.assert_allclose(out7, np.array([float(v) for v in ref]), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(dhead.numpy(), hr.grad.numpy(), rtol=1e-4, atol=1e-9)


def test_decode_matches_reference_fixture(ops):
    """box coordinates within 1e-4 (north star); counts and order exact away from the threshold."""
    z = np.load(os.path.join(G, "loss_decode.npz"))
    specs = yolo_ref.specs_from_config(procedural.VOC_CONFIG)
    for hi, g in enumerate((11, 22)):
        head = torch.from_numpy(z["head%d" % hi])
        anchors, mask, hp = _head_args(ops, specs[hi], 4, g)
        h = head.permute(0, 2, 3, 1).contiguous().cuda()
        for vc in (1, 3, 5):
            rows, counts = ops.yolo_decode(h, anchors, mask, hp, vc / 10)
            counts = counts.cpu().tolist()
            # second head appended behind the first inside wider per-image slots
            wide, c2 = ops.yolo_decode(h, anchors, mask, hp, vc / 10, row_stride=2 * rows.shape[1] + 3)
            wide, c3 = ops.yolo_decode(h, anchors, mask, hp, vc / 10, rows=wide, row_stride=2 * rows.shape[1] + 3, base_counts=c2)
            assert c3.cpu().tolist() == [2 * c for c in counts]
            for b in range(4):
                assert torch.equal(wide[b, counts[b]:2 * counts[b]], rows[b, :counts[b]])
            assert counts == z["dec%d_%d_counts" % (hi, vc)].tolist()
            got = torch.cat([rows[b, :counts[b]] for b in range(4)]).cpu().numpy()
            ref = z["dec%d_%d_rows" % (hi, vc)]
            np.testing.assert_allclose(got[:, :6], ref[:, :6], rtol=0, atol=1e-4)
            assert np.array_equal(got[:, 6], ref[:, 6])


def _rand_rows(n, C, seed, quant=False):
    r = np.random.RandomState(seed)
    xy = r.rand(n, 2).astype(np.float32)
    wh = (0.02 + 0.28 * r.rand(n, 2)).astype(np.float32)
    conf = r.rand(n, 1).astype(np.float32)
    sc = r.rand(n, 1).astype(np.float32)
    if quant:                                   # many exact score ties
        conf = np.round(conf, 1) + 0.05
        sc = np.ones_like(sc)
    cls = r.randint(0, C, size=(n, 1)).astype(np.float32)
    return torch.from_numpy(np.concatenate((xy - wh / 2, xy + wh / 2, conf, sc, cls), 1).astype(np.float32))


@pytest.mark.parametrize("sizes,C,quant", [([50, 0, 300, 7], 20, False), ([1815] * 6, 20, False), ([400, 33], 7, True),
                                            ([0, 0], 20, False), ([20000], 20, False),
                                            # large-bucket path (segments averaging > 2048 rows): bit-matrix + tile resolve
                                            ([9000, 5000], 3, True), ([30000], 4, False), ([2500, 0, 7000], 5, False)])
def test_nms_indices_bit_exact(ops, sizes, C, quant):
    """Same rows into both implementations -> identical kept indices, order included (Q13)."""
    segs = [_rand_rows(n, C, seed=11 + i, quant=quant) for i, n in enumerate(sizes)]
    rows = torch.cat(segs) if sum(sizes) else torch.zeros(0, 7)
    off = np.zeros(len(sizes) + 1, np.int32)
    off[1:] = np.cumsum(sizes)
    dev_rows = rows.cuda() if rows.shape[0] else torch.zeros(1, 7).cuda()[:0]
    beg, cnt = torch.from_numpy(off[:-1].copy()).cuda(), torch.tensor(sizes, dtype=torch.int32).cuda()
    out_idx, out_counts, out_rows, prefix, status = ops.nms_per_class(dev_rows, beg, cnt, C, 0.45, max_seg_rows=max(sizes))
    assert int(status.cpu()) == 0
    out_idx, out_counts, prefix = out_idx.cpu().numpy(), out_counts.cpu().numpy(), prefix.cpu().numpy()
    assert np.array_equal(prefix, np.concatenate(([0], np.cumsum(out_counts))))
    for s, seg in enumerate(segs):
        ref_rows, ref_idx = nms_ref.nms_rows(seg, C, 0.45)
        got = out_idx[off[s]:off[s] + out_counts[s]] - off[s]
        assert out_counts[s] == len(ref_idx)
        assert np.array_equal(got, ref_idx.numpy())
        assert np.array_equal(out_rows[prefix[s]:prefix[s + 1]].cpu().numpy(), ref_rows.numpy())


def test_nms_large_buckets_dense_overlap(ops):
    """Heavily overlapping boxes (most are suppressed, long suppression chains) through the large-bucket path, and the same
    rows through the one-workgroup path (MNY_NMS_SMALL is read per call): both equal the CPU restatement."""
    r = np.random.RandomState(3)
    n, C = 12000, 3
    ctr = 0.5 + 0.05 * r.randn(n, 2).astype(np.float32)
    wh = (0.2 + 0.1 * r.rand(n, 2)).astype(np.float32)
    rows = torch.from_numpy(np.concatenate((ctr - wh / 2, ctr + wh / 2, r.rand(n, 2), r.randint(0, C, (n, 1))), 1).astype(np.float32))
    ref_rows, ref_idx = nms_ref.nms_rows(rows, C, 0.45)
    beg, cnt = torch.tensor([0], dtype=torch.int32).cuda(), torch.tensor([n], dtype=torch.int32).cuda()
    out_idx, out_counts, out_rows, prefix, status = ops.nms_per_class(rows.cuda(), beg, cnt, C, 0.45)
    assert int(status.cpu()) == 0 and int(out_counts.cpu()[0]) == len(ref_idx)
    assert np.array_equal(out_idx[:len(ref_idx)].cpu().numpy(), ref_idx.numpy())
    assert np.array_equal(out_rows[:len(ref_idx)].cpu().numpy(), ref_rows.numpy())


def test_nms_reference_driver_fixture(ops):
    z = np.load(os.path.join(G, "loss_decode.npz"))
    zn = np.load(os.path.join(G, "nms_driver.npz"))
    per_img = []
    for b in range(4):
        parts = []
        for hi in range(2):
            cnt = z["dec%d_3_counts" % hi]
            o = int(cnt[:b].sum())
            parts.append(torch.from_numpy(z["dec%d_3_rows" % hi][o:o + cnt[b]]))
        per_img.append(torch.cat(parts))                       # utils/box.py:17
    sizes = [len(p) for p in per_img]
    off = np.zeros(5, np.int32)
    off[1:] = np.cumsum(sizes)
    rows = torch.cat(per_img)
    beg, cnt = torch.from_numpy(off[:-1].copy()).cuda(), torch.tensor(sizes, dtype=torch.int32).cuda()
    out_idx, out_counts, out_rows, prefix, status = ops.nms_per_class(rows.cuda(), beg, cnt, 20, 0.45, max_seg_rows=max(sizes))
    assert int(status.cpu()) == 0
    assert out_counts.cpu().tolist() == zn["counts"].tolist()
    assert np.array_equal(out_rows[:int(prefix[-1])].cpu().numpy(), zn["rows"])


def test_nms_buckets_beyond_the_lds_image(ops):
    """utils/box.py:20-29 has no bucket-size limit: one class with 9 000 / 50 000 boxes (more than the 8 192-row LDS image) goes
    through the global-scratch path, mixed with ordinary buckets, on both driver paths — indices identical to the CPU kernel."""
    for n, C, seed, dense in ((9000, 1, 5, False), (50000, 1, 6, True), (30000, 3, 7, False)):
        r = np.random.RandomState(seed)
        if dense:                                   # heavy overlap: long kept list re-scans
            ctr = 0.5 + 0.15 * r.randn(n, 2).astype(np.float32)
            wh = (0.05 + 0.1 * r.rand(n, 2)).astype(np.float32)
            cls = np.zeros((n, 1), np.float32)
        else:
            ctr = r.rand(n, 2).astype(np.float32)
            wh = (0.02 + 0.1 * r.rand(n, 2)).astype(np.float32)
            cls = np.zeros((n, 1), np.float32) if C == 1 else np.where(r.rand(n, 1) < 0.8, 1.0, r.randint(0, C, (n, 1))).astype(np.float32)   # class 1 oversized
        rows = torch.from_numpy(np.concatenate((ctr - wh / 2, ctr + wh / 2, r.rand(n, 2).astype(np.float32), cls), 1).astype(np.float32))
        ref_rows, ref_idx = nms_ref.nms_rows(rows, C, 0.45)
        for segs in ((n,), (n // 3, n - n // 3)):   # one segment (large-bucket driver) and two (still oversized buckets)
            off = np.concatenate(([0], np.cumsum(segs))).astype(np.int32)
            beg, cnt = torch.from_numpy(off[:-1].copy()).cuda(), torch.tensor(segs, dtype=torch.int32).cuda()
            out_idx, out_counts, out_rows, prefix, status = ops.nms_per_class(rows.cuda(), beg, cnt, C, 0.45, max_seg_rows=max(segs))
            assert int(status.cpu()) == 0
            oc = out_counts.cpu().numpy()
            if len(segs) == 1:
                assert oc[0] == len(ref_idx)
                assert np.array_equal(out_idx[:oc[0]].cpu().numpy(), ref_idx.numpy())
                assert np.array_equal(out_rows[:oc[0]].cpu().numpy(), ref_rows.numpy())
            else:
                for si in range(len(segs)):
                    seg = rows[off[si]:off[si + 1]]
                    rr, ri = nms_ref.nms_rows(seg, C, 0.45)
                    got = out_idx[off[si]:off[si] + oc[si]].cpu().numpy() - off[si]
                    assert oc[si] == len(ri) and np.array_equal(got, ri.numpy()), (n, C, si)


def test_nms_wrong_segment_bound_is_reported(ops):
    """A caller's max_seg_rows smaller than a real segment is the one remaining error: reported through the status word."""
    rows = _rand_rows(9000, 1, seed=5)
    beg, cnt = torch.tensor([0], dtype=torch.int32).cuda(), torch.tensor([9000], dtype=torch.int32).cuda()
    _, counts, _, _, status = ops.nms_per_class(rows.cuda(), beg, cnt, 1, 0.45, max_seg_rows=4000)
    assert int(status.cpu()) == 9000 and int(counts.cpu()[0]) == 0


def _config5_rows(n=100000, C=20, seed=2):
    """BASELINE configs[4] / SURVEY §8d C5(ii): one image of n rows, cls~U{0..C-1}, centres U(0,1)^2, sizes U(0.02,0.3)."""
    r = np.random.RandomState(seed)
    ctr = r.rand(n, 2).astype(np.float32)
    wh = (0.02 + 0.28 * r.rand(n, 2)).astype(np.float32)
    return torch.from_numpy(np.concatenate((ctr - wh / 2, ctr + wh / 2, r.rand(n, 2).astype(np.float32),
                                            r.randint(0, C, (n, 1)).astype(np.float32)), 1).astype(np.float32))


def test_nms_config5_100k_boxes_20_classes_indices_equal_cpu(ops):
    """The full-size configs[4] case (the rows bench.py times): kept indices identical to oracle/nms_ref.c, order included."""
    n, C = 100000, 20
    rows = _config5_rows(n, C)
    ref_rows, ref_idx = nms_ref.nms_rows(rows, C, 0.45)
    beg, cnt = torch.tensor([0], dtype=torch.int32).cuda(), torch.tensor([n], dtype=torch.int32).cuda()
    out_idx, out_counts, out_rows, prefix, status = ops.nms_per_class(rows.cuda(), beg, cnt, C, 0.45)
    kept = int(out_counts.cpu()[0])
    assert int(status.cpu()) == 0 and kept == len(ref_idx)
    assert np.array_equal(out_idx[:kept].cpu().numpy(), ref_idx.numpy())
    assert np.array_equal(out_rows[:kept].cpu().numpy(), ref_rows.numpy())
    # size-independent properties: per class, kept boxes are pairwise IoU <= thr and score-descending
    k = out_rows[:kept].cpu()
    for c in range(C):
        kc = k[k[:, 6] == c]
        s = kc[:, 4] * kc[:, 5]
        assert bool((s[:-1] >= s[1:]).all())


@pytest.mark.parametrize("val_conf", [1e-6, 0.5])
def test_decode_config5_full_size_99825_candidates_equal_oracle(ops, val_conf):
    """BASELINE configs[4] / SURVEY §8d C5(i) at the size bench.py times (VERDICT r2 #1c): both heads of 55 images at 352x352 =
    99 825 candidates, second head appended behind the first on the device, against yolo_ref.decode_rows (restating
    models/yolo_loss.py:180-204): per-image counts and row order exact, class index exact, box coordinates / confidences <= 1e-4."""
    y = procedural.VOC_CONFIG["yolo"]
    specs = yolo_ref.specs_from_config(procedural.VOC_CONFIG)
    N, S = 55, 352
    g = torch.Generator().manual_seed(4)
    heads = [torch.randn(N, grid, grid, len(y["mask"][hi]) * (5 + y["num_classes"]), generator=g) for hi, grid in enumerate((S // 32, S // 16))]
    args = [_head_args(ops, specs[hi], N, heads[hi].shape[1], S) for hi in range(2)]
    cap = sum(hp.A * hp.g * hp.g for _a, _m, hp in args)
    assert N * cap == 99825
    rows, c0 = ops.yolo_decode(heads[0].cuda(), args[0][0], args[0][1], args[0][2], val_conf, row_stride=cap)
    rows, c1 = ops.yolo_decode(heads[1].cuda(), args[1][0], args[1][1], args[1][2], val_conf, rows=rows, row_stride=cap, base_counts=c0)
    c0, c1, rows = c0.cpu().numpy(), c1.cpu().numpy(), rows.cpu()
    ref = []
    for hi in range(2):
        specs[hi].val_conf = val_conf
        ref.append(yolo_ref.decode_rows(heads[hi], specs[hi], [S, S], layout="nhwc"))
    assert c0.tolist() == [len(r) for r in ref[0]]
    assert c1.tolist() == [len(a) + len(b) for a, b in zip(ref[0], ref[1])]          # utils/box.py:17: the two heads' rows back to back
    if val_conf < 1e-3:
        assert int(c1.sum()) > 0.999 * N * cap                                        # ~every candidate passes, as in the bench leg
    for b in range(N):
        want = torch.cat((ref[0][b], ref[1][b])).numpy()
        got = rows[b, :c1[b]].numpy()
        np.testing.assert_allclose(got[:, :6], want[:, :6], rtol=0, atol=1e-4)
        assert np.array_equal(got[:, 6], want[:, 6])


def test_nms_reference_shaped_batch_256_images_x_1815_candidates(ops):
    """the reference-shaped NMS case — one eval batch of 256 images with all 1 815 candidates of both heads each
    (464 640 rows, 5 120 (image, class) buckets of ~90 rows: utils/box.py:11-31 runs 256 x 20 torchvision.ops.nms calls here).
    Kept indices per image identical to oracle/nms_ref.c (parity-unpinned restatement of torchvision's CPU kernel), order included."""
    S, n, C = 256, 1815, 20
    segs = [_rand_rows(n, C, seed=1000 + i) for i in range(S)]
    rows = torch.cat(segs)
    off = np.arange(S + 1, dtype=np.int32) * n
    beg, cnt = torch.from_numpy(off[:-1].copy()).cuda(), torch.full((S,), n, dtype=torch.int32).cuda()
    out_idx, out_counts, out_rows, prefix, status = ops.nms_per_class(rows.cuda(), beg, cnt, C, 0.45, max_seg_rows=n)
    assert int(status.cpu()) == 0
    out_idx, oc, prefix = out_idx.cpu().numpy(), out_counts.cpu().numpy(), prefix.cpu().numpy()
    out_rows = out_rows.cpu().numpy()
    assert np.array_equal(prefix, np.concatenate(([0], np.cumsum(oc))))
    for s, seg in enumerate(segs):
        ref_rows, ref_idx = nms_ref.nms_rows(seg, C, 0.45)
        assert oc[s] == len(ref_idx), s
        assert np.array_equal(out_idx[off[s]:off[s] + oc[s]] - off[s], ref_idx.numpy()), s
        assert np.array_equal(out_rows[prefix[s]:prefix[s + 1]], ref_rows.numpy()), s


def _gpu_nms_single(ops, rows, C, thr):
    """One image's rows through mny_nms_per_class -> kept original indices (class-major, as utils/box.py:20-29 emits them)."""
    n = rows.shape[0]
    beg, cnt = torch.tensor([0], dtype=torch.int32).cuda(), torch.tensor([n], dtype=torch.int32).cuda()
    out_idx, out_counts, out_rows, prefix, status = ops.nms_per_class(rows.cuda(), beg, cnt, C, thr)
    assert int(status.cpu()) == 0
    return out_idx[:int(out_counts.cpu()[0])].cpu().tolist()


def _rows(boxes, scores, cls=0):
    b = torch.tensor(boxes, dtype=torch.float32)
    s = torch.tensor(scores, dtype=torch.float32)
    return torch.cat((b, torch.ones(len(s), 1), s[:, None], torch.full((len(s), 1), float(cls))), 1)


def test_nms_threshold_edges_known_answers_on_the_gpu(ops):
    """the strict `>` against a DOUBLE threshold of a FLOAT IoU (torchvision's CPU kernel, utils/box.py:27-28) through
    mny_nms_per_class itself, not only through the oracle: IoU exactly 0.5 against thr 0.5 / 0.4999999 / 0.5000000001 (the cases of
    tests/test_oracle_nms.py:22), and each answer equal to the CPU restatement's."""
    rows = _rows([[0, 0, 2, 1], [1, 0, 3, 1], [0, 0, 1, 1.0]], [0.9, 0.8, 0.7])      # iou(0,1) = 1/3, iou(0,2) = 0.5 exactly
    for thr, want in ((0.5, [0, 1, 2]), (0.4999999, [0, 1]), (0.5000000001, [0, 1, 2])):
        assert _gpu_nms_single(ops, rows, 1, thr) == want, thr
        assert nms_ref.nms_rows(rows, 1, thr)[1].tolist() == want
    # duplicates and exact ties: stable descending order keeps the earlier index (torchvision sorts stably)
    rows = _rows([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.0]], [0.9, 0.8, 0.7, 0.9])
    assert _gpu_nms_single(ops, rows, 1, 0.45) == [0, 2] == nms_ref.nms_rows(rows, 1, 0.45)[1].tolist()
    assert _gpu_nms_single(ops, rows, 1, 0.7) == [0, 1, 2] == nms_ref.nms_rows(rows, 1, 0.7)[1].tolist()


def test_nms_float_iou_rounding_across_the_045_threshold(ops):
    """A pair whose IoU, computed in float as torchvision computes it (inter / (a + b - inter)), lands within a few ulps of float32(0.45):
    whether it is suppressed depends on the float rounding of `ovr` against the DOUBLE 0.45.  Sweep the second box's width through
    that neighbourhood one float at a time: every case must agree with the CPU restatement, and both outcomes must occur."""
    import struct
    outcomes = set()
    w0 = np.float32(0.45 * 2 / 1.45)                    # second box [0, 0, w, 1] inside the unit box: IoU = w, need w around 0.45... use widths near 0.45
    for k in range(-6, 7):
        w = np.float32(0.45)
        bits = struct.unpack("I", struct.pack("f", float(w)))[0] + k
        w = np.float32(struct.unpack("f", struct.pack("I", bits))[0])
        rows = _rows([[0, 0, 1, 1], [0, 0, float(w), 1.0]], [0.9, 0.8])              # IoU = w / 1 (contained box): float ovr == w exactly
        got = _gpu_nms_single(ops, rows, 1, 0.45)
        assert got == nms_ref.nms_rows(rows, 1, 0.45)[1].tolist(), (k, float(w))
        assert got == ([0] if float(w) > 0.45 else [0, 1]), (k, float(w))             # float -> double promotion, strict >
        outcomes.add(len(got))
    assert outcomes == {1, 2}
    del w0


def test_nms_score_ties_across_the_64_candidate_tile_boundary(ops):
    """130 boxes of ONE score in one class, laid out so that suppression depends on stable order across the 64-candidate tiles of the
    greedy loop: box 2m+1 duplicates box 2m (IoU 1) -> exactly the even indices survive, in index order; then the same with the pairs
    straddling tile boundaries (offset by one)."""
    n = 130
    for shift in (0, 1):
        boxes = []
        for i in range(n):
            m = (i + shift) // 2
            boxes.append([3.0 * m, 0.0, 3.0 * m + 2.0, 2.0])
        rows = _rows(boxes, [0.5] * n)
        want = nms_ref.nms_rows(rows, 1, 0.45)[1].tolist()
        assert _gpu_nms_single(ops, rows, 1, 0.45) == want
        first = {}
        for i in range(n):
            first.setdefault((i + shift) // 2, i)
        assert want == sorted(first.values())
