import sys; sys.path.insert(0, '/root/repo')
import torch
from torch.profiler import profile, ProfilerActivity
from mobilenet_yolo_pytorch_amd import synthetic, yolo
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = yolo(synthetic.VOC_CONFIG).to(dev).train()
x = synthetic.images(64, 352, 352, seed=0).to(dev)
tg = synthetic.targets(64, seed=1, empty_every=16)
def step():
    for p in m.parameters(): p.grad = None
    out = m(x, tg)
    (out[0][0] + out[1][0]).backward()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=60))
