"""Device-side batch input preparation (mny_prep_batch through the C ABI) against fixtures made with the real Pillow /
torch ops (tools/gen_golden_prep.py) and against the oracle: the uint8 resample is integer work -> BIT-EXACT; the fp32
normalisation uses the same two divisions as torch -> bit-exact as well."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import prep_ref

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


@pytest.fixture(scope="module")
def P():
    assert torch.cuda.is_available()
    from mobilenet_yolo_pytorch_amd import prep
    return prep


@pytest.mark.parametrize("name", ["prep_small.npz", "prep_rect.npz"])
def test_reference_fixture_bit_exact(P, name):
    z = np.load(os.path.join(G, name))
    size = tuple(int(v) for v in z["size"])
    imgs = [z["img%d" % i] for i in range(sum(k.startswith("img") for k in z.files))]
    bp = P.BatchPrep([size], z["mean"], z["std"])
    out = bp(imgs).cpu().numpy()
    bp.check()
    assert out.shape == z["batch"].shape and np.array_equal(out, z["batch"])
    # the uint8 stage on its own: undo the normalisation exactly (mean 0, std 1/255 -> out = u8)
    raw = P.BatchPrep([size], [0, 0, 0], [1.0 / 255.0] * 3)(imgs).cpu().numpy()
    for i in range(len(imgs)):
        assert np.array_equal(np.rint(raw[i]).astype(np.uint8).transpose(1, 2, 0), z["u8_%d" % i])


@pytest.mark.parametrize("size", [(352, 352), (288, 288), (416, 416), (96, 160)])
def test_voc_shaped_batch_vs_oracle(P, size):
    from mobilenet_yolo_pytorch_amd import synthetic
    r = np.random.RandomState(size[0])
    sizes = [(int(r.randint(120, 500)), int(r.randint(120, 500))) for _ in range(5)] + [(500, 375), (375, 500), size, (size[0], 333), (33, 47)]
    imgs = synthetic.photos(sizes, seed=size[1])
    out = P.BatchPrep([size], MEAN, STD)(imgs).cpu().numpy()
    ref = prep_ref.collate(imgs, size, MEAN, STD)
    assert np.array_equal(out, ref)


def test_large_downscale_many_taps(P):
    """BDD-shaped source (720x1280 -> 416x416: 3.1x / 1.7x down-scale, 9 and 5 taps)."""
    from mobilenet_yolo_pytorch_amd import synthetic
    imgs = synthetic.photos([(720, 1280), (1280, 720)], seed=2)
    out = P.BatchPrep([(416, 416)], [0.5] * 3, [1.0] * 3)(imgs).cpu().numpy()
    assert np.array_equal(out, prep_ref.collate(imgs, (416, 416), [0.5] * 3, [1.0] * 3))


def test_one_size_draw_per_batch_like_the_reference(P):
    from mobilenet_yolo_pytorch_amd import synthetic
    sizes = [[352, 352], [320, 320], [288, 288], [384, 384], [416, 416]]
    bp = P.BatchPrep(sizes, MEAN, STD, rng=random.Random(7))
    ref_rng = random.Random(7)
    imgs = synthetic.photos([(100, 120), (90, 60)], seed=1)
    for _ in range(4):
        want = prep_ref.choose_size(ref_rng, [tuple(s) for s in sizes])
        assert tuple(bp(imgs).shape) == (2, 3, want[0], want[1])


def test_feeds_the_network(P):
    """The prepared batch is what the stem reads: eval forward on it == eval forward on the oracle's batch."""
    from mobilenet_yolo_pytorch_amd import synthetic, yolo
    from oracle import procedural
    imgs = synthetic.photos([(150, 200), (220, 130)], seed=4)
    x = P.BatchPrep([(96, 96)], MEAN, STD)(imgs)
    torch.manual_seed(0)
    m = yolo(procedural.VOC_CONFIG)
    procedural.fill_state_dict_(m)
    m = m.cuda().eval()
    a = m(x)
    b = m(torch.from_numpy(prep_ref.collate(imgs, (96, 96), MEAN, STD)).cuda())
    assert all(torch.equal(p, q) for p, q in zip(a, b))


def test_out_of_bounds_image_is_flagged_not_overrun(P):
    import ctypes
    from mobilenet_yolo_pytorch_amd import _lib, synthetic
    imgs = synthetic.photos([(40, 40), (80, 90)], seed=1)
    bp = P.BatchPrep([(32, 32)], MEAN, STD)
    stage, desc, mh, mw = bp.pack(imgs)
    src = stage.cuda()
    desc_dev = torch.from_numpy(desc.view(np.uint8).copy()).cuda()
    out = bp.run_device(src, desc_dev, 2, 64, 64, (32, 32))          # the caller's bound is too small for image 1
    with pytest.raises(RuntimeError, match="image 1"):
        bp.check()
    assert not out[1].any() and np.array_equal(out[0].cpu().numpy(), prep_ref.collate(imgs[:1], (32, 32), MEAN, STD)[0])
    with pytest.raises(ValueError):
        bp([np.zeros((4, 4), np.uint8)])
    assert _lib.query("mny_prep_ws_bytes", 2, 64, 64, 32, 32) > 0


def test_full_batch_properties(P):
    """bs=256 at the headline size, through size-independent properties: an image already at the target size passes through
    unresampled; a constant-colour image stays constant; every pixel is a valid normalised uint8 level."""
    from mobilenet_yolo_pytorch_amd import synthetic
    r = np.random.RandomState(0)
    base = synthetic.photos([(352, 352), (375, 500), (500, 333)], seed=9)
    flat = np.empty((281, 499, 3), np.uint8)
    flat[...] = (17, 130, 250)
    imgs = [base[i % 3] for i in range(254)] + [flat, base[0]]
    out = P.BatchPrep([(352, 352)], MEAN, STD)(imgs).cpu().numpy()
    assert out.shape == (256, 3, 352, 352)
    m, s = np.asarray(MEAN, np.float32)[:, None, None], np.asarray(STD, np.float32)[:, None, None]
    ident = (base[0].astype(np.float32).transpose(2, 0, 1) / np.float32(255) - m) / s
    assert np.array_equal(out[0], ident) and np.array_equal(out[255], ident) and np.array_equal(out[3], out[0])
    const = (np.array([17, 130, 250], np.float32)[:, None, None] / np.float32(255) - m) / s
    assert np.array_equal(out[254], np.broadcast_to(const, (3, 352, 352)))
    levels = np.rint((out * s + m) * 255)
    assert np.abs((out * s + m) * 255 - levels).max() < 1e-3 and levels.min() >= 0 and levels.max() <= 255
