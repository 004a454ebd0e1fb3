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
        snap = [float(res[0][0].detach()), float(res[1][0].detach())] + [p.grad.clone() for p in m.parameters() if p.grad is not None]
        if ref is None:
            ref = snap
            continue
        assert snap[0] == ref[0] and snap[1] == ref[1], (it, snap[:2], ref[:2])
        for a, b in zip(snap[2:], ref[2:]):
            assert torch.equal(a, b), it


@pytest.mark.parametrize("arch,bf16,bs,size", [("mbv3", True, 16, 512), ("mbv3", False, 16, 384), ("mbv2", True, 32, 352)])
def test_step_is_bitwise_deterministic_at_benchmark_sized_plans(arch, bf16, bs, size):
    """the whole-step determinism guard of tests/test_gpu_net.py (MobileNetV2 fp32, bs 64) extended to MobileNetV3 (h-swish /
    gate units, 5x5 depthwise, the module applied twice), to bf16 storage and to plans large enough for the M-gated kernel families:
    six replays with allocator churn in between, loss tuples and every parameter gradient bit-identical."""
    from mobilenet_yolo_pytorch_amd import mbv3, yolo
    from oracle import procedural
    torch.manual_seed(0)
    cls = yolo if arch == "mbv2" else mbv3.yolo
    m = cls(procedural.VOC_CONFIG, act_dtype=torch.bfloat16 if bf16 else torch.float32)
    procedural.fill_state_dict_(m)
    m = m.cuda().train()
    x = procedural.images(bs, size, size, seed=21).cuda()
    tg = procedural.targets(bs, seed=22, empty_every=5)
    ref = None
    for it in range(6):
        m.zero_grad(set_to_none=True)
        junk = torch.randn(1 << 22, device="cuda")
        res = m(x, tg)
        (res[0][0] + res[1][0]).backward()
        torch.cuda.synchronize()
        del junk
        snap = [[float(v) for r in res for v in r]] + [p.grad.clone() for p in m.parameters() if p.grad is not None]
        if ref is None:
            ref = snap
            continue
        assert snap[0] == ref[0], (it, snap[0], ref[0])
        bad = [i for i, (a, b) in enumerate(zip(snap[1:], ref[1:])) if not torch.equal(a, b)]
        assert not bad, (it, bad[:8])


@pytest.mark.parametrize("arch,bf16", [("mbv2", False), ("mbv3", False), ("mbv3", True)])
def test_side_stream_weight_gradients_equal_single_stream(arch, bf16, monkeypatch):
    """the weight-gradient kernels of small plans run on a second HIP stream (fork after dY is ready, join before every
    batched combine and at the end of every replayed segment).  The same step with MNY_SIDE_STREAM=0 and =1 (read when a plan is
    built) must give bit-identical losses and gradients — MobileNetV3 covers the module applied twice (its contributions stay in
    order on the main stream), bf16 the shadow-weight path; the data-parallel segments are covered by the 4-bucket replay below."""
    from mobilenet_yolo_pytorch_amd import mbv3, yolo
    from oracle import procedural
    cls = yolo if arch == "mbv2" else mbv3.yolo
    x = procedural.images(8, 160, 160, seed=31).cuda()
    tg = procedural.targets(8, seed=32, empty_every=3)
    snaps = {}
    for side in ("0", "1"):
        monkeypatch.setenv("MNY_SIDE_STREAM", side)
        m = cls(procedural.VOC_CONFIG, act_dtype=torch.bfloat16 if bf16 else torch.float32)
        procedural.fill_state_dict_(m)
        m = m.cuda().train()
        for rep in range(3):                                     # later replays reuse the plan's two events
            m.zero_grad(set_to_none=True)
            res = m(x, tg)
            plan = next(iter(m._plans.values()))
            assert plan.side_on == (side == "1")
            if side == "1" and rep == 2:                         # segmented replay as the data-parallel reducer drives it (joins at segment ends)
                g = torch.ones(plan.g_scale.numel(), device="cuda")
                plan.stream.value = torch.cuda.current_stream().cuda_stream
                plan.stream_side.value = plan._side_stream.cuda_stream
                plan.x_ptr.value = plan.saved_x.data_ptr()
                plan.g_scale.copy_(g)
                n = len(plan.bwd.calls)
                for b, e in ((0, n // 4), (n // 4, n // 2), (n // 2, 3 * n // 4), (3 * n // 4, None)):
                    plan.run_bwd_segment(b, e)
                grads = [plan.gviews[k].clone() for k in plan.grad_params]
            else:
                (res[0][0] + res[1][0]).backward()
                grads = [plan.gviews[k].clone() for k in plan.grad_params]
            torch.cuda.synchronize()
            snap = [[float(v) for r in res for v in r]] + grads
            if side in snaps:
                assert snap[0] == snaps[side][0]
                assert all(torch.equal(a, b) for a, b in zip(snap[1:], snaps[side][1:])), (side, rep)
            snaps[side] = snap
        del m
    assert snaps["0"][0] == snaps["1"][0]
    bad = [i for i, (a, b) in enumerate(zip(snaps["0"][1:], snaps["1"][1:])) if not torch.equal(a, b)]
    assert not bad, bad[:8]
