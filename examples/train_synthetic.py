#!/usr/bin/env python3
"""Caller-side counterpart of the reference's training step (train.py:246-283) on synthetic data:
zero_grad -> model(images, targets) -> sum(losses) -> backward -> AdamW.step, with the drop-in HIP module.

    python examples/train_synthetic.py [--arch mbv2|mbv3] [--steps 30] [--batch 32] [--size 352]
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_synthetic.py   # data parallel
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilenet_yolo_pytorch_amd import mbv3, synthetic, yolo  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="mbv2", choices=["mbv2", "mbv3"])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=352)
    a = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    reducer = None
    torch.manual_seed(0)
    model = (yolo if a.arch == "mbv2" else mbv3.yolo)(synthetic.VOC_CONFIG).to(dev).train()
    if world > 1:
        import torch.distributed as dist
        from mobilenet_yolo_pytorch_amd.dp import attach_data_parallel
        dist.init_process_group("nccl", device_id=dev)
        reducer = attach_data_parallel(model)
    from mobilenet_yolo_pytorch_amd.optim import AdamW                            # fused multi-tensor step, torch.optim.AdamW semantics
    opt = AdamW(model.parameters(), lr=7e-4, weight_decay=4e-4)                  # train.py:134,459-462
    first = last = None
    for step in range(a.steps):
        x = synthetic.images(a.batch, a.size, a.size, seed=step % 4 + 10 * rank).to(dev)   # 4 recurring batches: the loss must fall
        tg = synthetic.targets(a.batch, seed=step % 4 + 10 * rank + 1, empty_every=8)
        opt.zero_grad()                                   # train.py:254
        outputs = model(x, tg)                            # train.py:260
        loss = sum(o[0] for o in outputs)                 # train.py:265-276
        loss.backward()                                   # train.py:282
        opt.step()                                        # train.py:283 (data parallel: attach_data_parallel's optimizer-step pre-hook waits for the last gradient bucket)
        v = float(loss.detach())
        first = v if first is None else first
        last = v
        if rank == 0 and (step % 5 == 0 or step == a.steps - 1):
            print("step %3d  loss %.5f  recall %.3f/%.3f  avg_iou %.3f/%.3f" % (
                step, v, float(outputs[0][1]), float(outputs[1][1]), float(outputs[0][2]), float(outputs[1][2])))
    if rank == 0:
        print("loss %.5f -> %.5f" % (first, last))
    assert last < first, "training did not reduce the loss"
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
