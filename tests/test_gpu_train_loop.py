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
