"""CPU, world_size 2, gloo: bucket planning and the segmented-backward / bucketed all-reduce protocol of
mobilenet_yolo_pytorch_amd.dp (the RCCL path uses the same code with backend "nccl")."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mobilenet_yolo_pytorch_amd import dp


def test_plan_buckets_contiguous_and_cut_at_call_boundaries():
    sizes = [100, 4, 4, 300, 8, 8, 50, 4, 4, 1000, 12, 12]
    ready = [3, 3, 3, 7, 7, 7, 9, 9, 9, 15, 15, 15]
    b = dp.plan_buckets(sizes, ready, n_buckets=3)
    assert b[0][0] == 0 and b[-1][1] == sum(sizes)
    assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
    assert all(b[i][2] <= b[i + 1][2] for i in range(len(b) - 1))
    assert {e[2] for e in b} <= set(ready)
    # a bucket never ends in the middle of a run of slots that become ready together
    ends = {sum(sizes[:i + 1]) for i in range(len(sizes)) if i == len(sizes) - 1 or ready[i + 1] != ready[i]}
    assert all(e[1] in ends for e in b)
    one = dp.plan_buckets([5], [1], 4)
    assert one == [(0, 5, 1)]


def test_slot_extents_follow_the_real_arena_layout():
    """bucket ranges must come from slot offsets, not from numel() — the padded heads leave slack."""
    plan = FakePlan(0, [75 * 8, 75, 32, 5], slack={0: 76 * 8, 1: 76})
    ext = dp.PlanReducer.slot_extents(plan)
    assert ext == [608, 76, 32, 8] and sum(ext) == plan.gflat.numel()
    naive = [(plan.grad_slots[n][1] + 3) // 4 * 4 for n in plan.grad_params]
    assert sum(naive) < plan.gflat.numel()              # what round 1 computed: stops short of the arena's end


class _V:
    def __init__(self, v):
        self.value = v


class _Calls:
    def __init__(self, calls):
        self.calls = calls


class FakePlan:
    """Mimics engine.NetPlan for dp.PlanReducer: call i 'computes' slot i of the gradient arena."""

    def __init__(self, rank, sizes, slack=None):
        self.rank = rank
        self.grad_params = ["p%d" % i for i in range(len(sizes))]
        self.grad_slots, off = {}, 0
        slack = slack or {}
        for i, (n, s) in enumerate(zip(self.grad_params, sizes)):
            self.grad_slots[n] = (off, s)
            off += (max(s, slack.get(i, 0)) + 3) // 4 * 4       # engine._build_backward: padded detection heads reserve slack
        self.gflat = torch.zeros(off)
        self.gviews = {n: self.gflat[o:o + s] for n, (o, s) in self.grad_slots.items()}
        # two calls per slot: a no-op and the producing call (so ready indices are not trivially i+1)
        self.bwd = _Calls([c for n in self.grad_params for c in ((None, (), "noop", None), (None, (_V(self.gviews[n].data_ptr()),), "wgrad", None))])
        self.log = []

    def run_bwd_segment(self, begin, end):
        end = len(self.bwd.calls) if end is None else end
        self.log.append((begin, end))
        for ci in range(begin, end):
            if ci % 2 == 1:
                i = ci // 2
                self.gviews[self.grad_params[i]].fill_(float((self.rank + 1) * (i + 1)))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sizes = [7, 64, 3, 1000, 12, 5, 256, 33]
        plan = FakePlan(rank, sizes, slack={0: 40, 3: 1100})   # slots longer than their parameter, like the 75->76 channel heads
        red = dp.PlanReducer(plan, n_buckets=3)
        assert red.buckets[0][0] == 0 and red.buckets[-1][1] == plan.gflat.numel()
        assert red.ar.backend == "gloo" and red.ar.avg is None and not red.ar.stage_host
        for step in range(2):                       # second step exercises the wait-before-rewrite path
            red.run_backward()
            red.wait()
            for i, n in enumerate(plan.grad_params):
                want = (1 + 2) / 2 * (i + 1)        # mean over ranks of (rank+1)*(i+1)
                assert torch.allclose(plan.gviews[n], torch.full((sizes[i],), want)), (rank, step, n)
        segs = plan.log[:len(red.buckets) + 1]
        assert segs[0][0] == 0 and all(segs[k][1] == segs[k + 1][0] for k in range(len(segs) - 1))
        assert segs[-1][1] == len(plan.bwd.calls)
        # broadcast helper: rank 0's values win
        m = torch.nn.Linear(3, 2)
        with torch.no_grad():
            m.weight.fill_(float(rank + 5))
        mred = dp.attach_data_parallel(m)
        assert float(m.weight[0, 0]) == 5.0
        # (i) any torch optimizer's step() first waits for the all-reduces still in flight (global step pre-hook), so the
        # literal `loss.backward(); optimizer.step()` of train.py:282-283 cannot read a half-reduced arena; (ii) the per-plan reducer is
        # owned by the PLAN — dropping the plan (plan-cache eviction under multi-scale training) frees it and its arena
        import gc
        import weakref
        plan2 = FakePlan(rank, sizes)
        pr = mred.for_plan(plan2)
        assert mred.for_plan(plan2) is pr and plan2._dp is pr and pr.plan is plan2
        pr.run_backward()
        assert pr.ar.pending, "gloo on CPU tensors reduces asynchronously: the last bucket must still be pending here"
        m.weight.grad = torch.zeros_like(m.weight)
        torch.optim.SGD(m.parameters(), lr=0.0).step()
        assert not pr.ar.pending, "optimizer.step() did not wait for the pending all-reduce"
        for i, n in enumerate(plan2.grad_params):
            assert torch.allclose(plan2.gviews[n], torch.full((sizes[i],), 1.5 * (i + 1))), (rank, n)
        wplan, warena = weakref.ref(plan2), weakref.ref(plan2.gflat)
        del plan2, pr
        gc.collect()
        assert wplan() is None and warena() is None and len(mred.live) == 0, "an evicted plan stayed alive through its reducer"
        mred.detach()
        q.put((rank, "ok"))
    except Exception as e:                          # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_bucketed_allreduce_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
