"""MI355X: the caller contract of train.py:246-283 with a stock torch optimizer — AdamW stepping on the arena-view
gradients must reduce the loss; zero_grad(set_to_none=False) and multi-scale inputs (train_img_size) must work."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("arch", ["mbv2", "mbv3"])
def test_adamw_reduces_loss(arch):
    from mobilenet_yolo_pytorch_amd import mbv3, synthetic, yolo
    torch.manual_seed(0)
    model = (yolo if arch == "mbv2" else mbv3.yolo)(synthetic.VOC_CONFIG).cuda().train()
    opt = torch.optim.AdamW(model.parameters(), lr=7e-4, weight_decay=4e-4)
    x = synthetic.images(8, 96, 96, seed=1).cuda()
    tg = synthetic.targets(8, seed=2, empty_every=4)
    losses = []
    for step in range(12):
        opt.zero_grad(set_to_none=(step % 2 == 0))          # both zero_grad flavours
        out = model(x, tg)
        loss = out[0][0] + out[1][0]
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.6 * losses[0], losses


def test_multi_scale_batches_share_one_model():
    from mobilenet_yolo_pytorch_amd import synthetic, yolo
    torch.manual_seed(0)
    model = yolo(synthetic.VOC_CONFIG).cuda().train()
    for size in (96, 128, 96, 160):                            # models/voc/config.yaml train_img_size switches per batch
        x = synthetic.images(2, size, size, seed=size).cuda()
        out = model(x, synthetic.targets(2, seed=size, empty_every=0))
        (out[0][0] + out[1][0]).backward()
        assert torch.isfinite(out[0][0]) and model.yolo_losses[0].img_size == [size, size]
    assert len(model._plans) == 3
    model.eval()
    det = model(synthetic.images(2, 128, 128, seed=5).cuda())
    assert len(det) == 2


@pytest.mark.parametrize("arch,bf16", [("mbv2", False), ("mbv3", True)])
def test_step_is_bitwise_deterministic(arch, bf16):
    """Same weights, same batch, five replays: losses and every parameter gradient are bit-identical (no float atomics anywhere;
    this also guards the hand-ordered asm pipelines — raw LDS reads, counted vmcnt, asm-prefetched mask/addend loads — against
    ordering hazards, which would show up as run-to-run differences)."""
    import torch
    from mobilenet_yolo_pytorch_amd import mbv3, yolo
    from oracle import procedural
    torch.manual_seed(0)
    cls = yolo if arch == "mbv2" else mbv3.yolo
    m = cls(procedural.VOC_CONFIG, act_dtype=torch.bfloat16 if bf16 else torch.float32).cuda().train()
    x = procedural.images(16, 224, 224, seed=11).cuda()
    tg = procedural.targets(16, seed=12, empty_every=5)
    ref = None
    for it in range(5):
        for p in m.parameters():
            p.grad = None
        res = m(x, tg)
        (res[0][0] + res[1][0]).backward()
        torch.cuda.synchronize()
        snap = [float(res[0][0]), float(res[1][0])] + [p.grad.clone() for p in m.parameters() if p.grad is not None]
        if ref is None:
            ref = snap
            continue
        assert snap[0] == ref[0] and snap[1] == ref[1], (it, snap[:2], ref[:2])
        for a, b in zip(snap[2:], ref[2:]):
            assert torch.equal(a, b), it
