"""rocprofv3 target: per-class NMS on the C5 workload (100k rows, 20 classes), 10 repetitions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mobilenet_yolo_pytorch_amd import ops

r = np.random.RandomState(2)
n, C = 100000, 20
ctr = r.rand(n, 2).astype(np.float32)
wh = (0.02 + 0.28 * r.rand(n, 2)).astype(np.float32)
rows = np.concatenate((ctr - wh / 2, ctr + wh / 2, r.rand(n, 2).astype(np.float32), r.randint(0, C, (n, 1)).astype(np.float32)), 1)
dev_rows = torch.from_numpy(rows.astype(np.float32)).cuda()
beg = torch.zeros(1, dtype=torch.int32, device="cuda")
cnt = torch.full((1,), n, dtype=torch.int32, device="cuda")
for _ in range(10):
    out = ops.nms_per_class(dev_rows, beg, cnt, C)
torch.cuda.synchronize()
print("kept", int(out[1].item()))
