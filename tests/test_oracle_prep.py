"""oracle/prep_ref.py (restated Pillow bilinear resample + ToTensor + Normalize) against fixtures produced by the real
Pillow / torch ops in the build container (tools/gen_golden_prep.py), and live against Pillow where it imports."""
import os

import numpy as np
import pytest

from oracle import prep_ref

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["prep_small.npz", "prep_rect.npz"])
def test_fixture_bit_exact(name):
    z = np.load(os.path.join(G, name))
    size = tuple(int(v) for v in z["size"])
    imgs = [z["img%d" % i] for i in range(sum(k.startswith("img") for k in z.files))]
    for i, im in enumerate(imgs):
        assert np.array_equal(prep_ref.resize_bilinear_u8(im, *size), z["u8_%d" % i]), i       # integer work: bit-exact
    batch = prep_ref.collate(imgs, size, z["mean"], z["std"])
    assert batch.dtype == np.float32 and np.array_equal(batch, z["batch"])                     # same fp32 divisions


def test_live_against_pillow():
    Image = pytest.importorskip("PIL.Image")
    r = np.random.RandomState(1)
    for t in range(25):
        h, w = r.randint(4, 300, 2)
        oh, ow = (h, w) if t == 0 else r.choice([7, 32, 96, 352, 416], 2)
        img = (r.rand(h, w, 3) * 256).astype(np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((int(ow), int(oh)), Image.BILINEAR))
        assert np.array_equal(prep_ref.resize_bilinear_u8(img, int(oh), int(ow)), ref), (h, w, oh, ow)


def test_coefficients():
    b, k = prep_ref.coefficients(100, 100)                     # same size: identity taps
    assert (b[:, 0] == np.arange(100)).all() and (k[:, 0] == 1 << 22).all() and not k[:, 1:].any()
    for ins, outs in ((500, 352), (375, 352), (100, 416), (1280, 416)):
        b, k = prep_ref.coefficients(ins, outs)
        assert k.shape[1] == int(np.ceil(max(ins / outs, 1.0))) * 2 + 1
        assert (np.abs(k.sum(1) - (1 << 22)) <= k.shape[1]).all()                   # normalised, up to rounding of each tap
        assert (b[:, 0] >= 0).all() and (b[:, 0] + b[:, 1] <= ins).all() and (b[:, 1] >= 1).all()


def test_choose_size_is_one_draw_per_batch():
    import random
    sizes = [[352, 352], [320, 320], [288, 288], [384, 384], [416, 416]]
    a, b = random.Random(5), random.Random(5)
    assert [prep_ref.choose_size(a, sizes) for _ in range(10)] == [b.choice(sizes) for _ in range(10)]
