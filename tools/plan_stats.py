"""Print the size of the static plan of the headline config: launches per step and resident bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobilenet_yolo_pytorch_amd import synthetic, yolo

dev = torch.device("cuda:0")
for tag, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    model = yolo(synthetic.VOC_CONFIG, act_dtype=dtype).to(dev).train()
    x = synthetic.images(256, 352, 352, seed=0).to(dev)
    tg = synthetic.targets(256, seed=1, empty_every=16)
    out = model(x, tg)
    (out[0][0] + out[1][0]).backward()
    torch.cuda.synchronize()
    plan = list(model._plans.values())[0]
    print(tag, "fwd launches", len(plan.fwd.calls), "bwd launches", len(plan.bwd.calls),
          "resident GB %.1f" % ((torch.cuda.memory_allocated() - base) / 1e9))
    del model, plan, out
