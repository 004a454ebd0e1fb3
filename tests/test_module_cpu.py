"""CPU: host-side logic of the drop-in module — state_dict contract, pickling, loud failure without a GPU."""
import io
import json
import os

import pytest
import torch

from oracle import net_ref, procedural

G = os.path.join(os.path.dirname(__file__), "golden")


def _mk(cfg=None):
    from mobilenet_yolo_pytorch_amd import yolo
    torch.manual_seed(0)
    return yolo(cfg or procedural.VOC_CONFIG)


def test_state_dict_matches_reference_manifest():
    for name in ("voc", "bdd100k"):
        man = json.load(open(os.path.join(G, "state_keys_%s.json" % name)))
        sd = _mk(man["config"]).state_dict()
        assert [[k, list(v.shape)] for k, v in sd.items()] == man["keys"]
        assert sum(p.numel() for p in _mk(man["config"]).parameters()) == man["num_params"]


def test_checkpoint_roundtrip_with_oracle_model():
    ref = procedural.fill_state_dict_(net_ref.RefYolo(procedural.VOC_CONFIG))
    m = _mk()
    missing, unexpected = m.load_state_dict(ref.state_dict(), strict=True)
    assert not missing and not unexpected
    for k, v in ref.state_dict().items():
        assert torch.equal(m.state_dict()[k], v)


def test_init_statistics_follow_reference():
    m = _mk()
    sd = m.state_dict()
    w = sd["backbone.features.1.conv.3.weight"]                   # 1x1 32->16: std = sqrt(2/(1*1*16))
    assert abs(w.std().item() - (2 / 16) ** 0.5) < 0.05
    w = sd["conv_for_S32.conv.weight"]                            # kaiming fan_out: std = sqrt(2/512)
    assert abs(w.std().item() - (2 / 512) ** 0.5) < 0.01
    assert torch.all(sd["conv_for_S32.bn.weight"] == 1) and torch.all(sd["conv_for_S32.bn.bias"] == 0)
    assert sd["yolo_headS32.3.bias"].abs().max() <= 1 / 1024 ** 0.5 + 1e-6   # nn.Conv2d default


def test_whole_module_pickles():
    m = _mk()
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    assert list(m2.state_dict().keys()) == list(m.state_dict().keys())
    assert m2.yolo_losses[0].val_conf == 0.1


def test_attributes_the_callers_touch():
    m = _mk()
    m.yolo_losses[0].val_conf = 0.3
    m.yolo_losses[1].val_conf = 0.3
    assert m.num_classes == 20 and len(m.yolo_losses) == 2
    assert not any("yolo_losses" in k for k in m.state_dict())


def test_cpu_call_fails_loudly():
    from mobilenet_yolo_pytorch_amd import MnyError
    m = _mk().eval()
    with pytest.raises(MnyError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 96, 96))


def test_pretrained_backbone_loader_follows_reference_remap():
    """a5: models/mobilenetv2.py:161-181.  Fixture = which checkpoint key the REAL loader put into every backbone tensor
    (tools/gen_golden_pretrained.py); the checkpoint is rebuilt here with the same recipe (tensor i filled with i+1,
    every 7th key prefixed `module.`)."""
    from mobilenet_yolo_pytorch_amd import synthetic
    man = json.load(open(os.path.join(G, "pretrained_map.json")))
    spec = synthetic.dli14_mobilenetv2_keys()
    assert len(spec) == man["spec_len"]
    ckpt, fill = {}, {}
    for i, (k, shape) in enumerate(spec):
        if i % 7 == 3:
            k = "module." + k
        ckpt[k] = torch.full(shape, float(i + 1))
        fill[k] = float(i + 1)
    m = _mk()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    loaded = m.load_pretrained_backbone(ckpt)
    sd = m.state_dict()
    for k, src in man["map"]:
        v = sd["backbone." + k]
        if src is None:
            assert torch.equal(v, before["backbone." + k]), k
        else:
            assert bool((v.double() == fill[src]).all()), (k, src)
    assert loaded == sorted(k for k, src in man["map"] if src is not None)
    for k, v in sd.items():                     # neck / heads untouched; the checkpoint's classifier went nowhere
        if not k.startswith("backbone."):
            assert torch.equal(v, before[k]), k
    # the split backbone really maps features2.N <- features.(14+N)
    assert bool((sd["backbone.features2.3.conv.6.weight"] == fill["features.17.conv.6.weight"]).all()
                if "features.17.conv.6.weight" in fill else True)
    # shape mismatch is an error like load_state_dict's; a URL is refused (no downloads)
    bad = dict(ckpt)
    bad["features.0.0.weight"] = torch.zeros(16, 3, 3, 3)
    with pytest.raises(RuntimeError):
        _mk().load_pretrained_backbone(bad)
    with pytest.raises(ValueError):
        _mk().load_pretrained_backbone("https://example.invalid/mobilenetv2.pth")
    # from a file
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".pth") as f:
        torch.save(ckpt, f.name)
        assert _mk().load_pretrained_backbone(f.name) == loaded


def test_plan_slots_for_steps_that_await_their_backward(monkeypatch):
    """Host logic of `yolo._grad_plan` / `_InFlight` (no GPU): a differentiable forward takes the first plan of its shape whose previous step is
    settled; with every slot busy it takes the plan whose forward is oldest; a token releases its plan only if no later forward re-claimed it."""
    import types
    from mobilenet_yolo_pytorch_amd import model as M
    m = _mk()
    made = {}

    def fake_plan(self, N, H, W, mode, slot=0):
        return made.setdefault((N, H, W, mode, slot), types.SimpleNamespace(fwd_gen=0, inflight_gen=None, last_fwd_tick=-1, slot=slot))

    monkeypatch.setattr(M.yolo, "_plan", fake_plan)
    tick = [0]

    def forward_on(p):                      # what NetPlan.forward_train + _TrainStep.forward do to the bookkeeping
        p.fwd_gen += 1
        p.last_fwd_tick = tick[0]
        tick[0] += 1
        return M._InFlight(p)

    p0 = m._grad_plan(2, 96, 96, True)
    t0 = forward_on(p0)
    p1 = m._grad_plan(2, 96, 96, True)
    assert p1 is not p0 and p1.slot == 1                       # the first step still owes its backward
    t1 = forward_on(p1)
    assert m._grad_plan(2, 96, 96, True) is p0                 # both busy: the oldest forward's plan is reused ...
    t2 = forward_on(p0)
    t0.release()                                               # ... and the superseded step's token must not free it
    assert p0.inflight_gen == p0.fwd_gen
    t1.release()
    assert m._grad_plan(2, 96, 96, True) is p1                 # slot 1 settled: it is the free one now
    t2.release()
    assert m._grad_plan(2, 96, 96, True) is p0
    t3 = forward_on(p0)
    del t3                                                     # a dropped autograd graph releases its plan
    assert p0.inflight_gen is None
    assert m._grad_plan(4, 96, 96, True) is not p0             # other shapes have their own slots
