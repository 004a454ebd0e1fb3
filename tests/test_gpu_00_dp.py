"""MI355X: data-parallel training on the REAL engine (BASELINE configs[2], SURVEY §8e parity definition).

`gpurun` gives one GPU and RCCL refuses two ranks on one device, so the 2-rank case runs two FRESH child processes that share
GPU 0 under `gloo` (the reducer stages each bucket through pinned host memory for that backend); the 1-rank case runs the RCCL
path itself.  Every check happens inside the children — this (parent) process never touches the GPU, and the file sorts before
the other GPU tests so the children are started from a process that has not initialised HIP.

Parity: after `attach_data_parallel` + one `backward()`, every `p.grad` equals the MEAN of the two single-rank gradient sets
computed sequentially (no process group involvement) in the same process on the same two shards — tolerance = one fp32 rounding of
the sum (the kernels' reductions are deterministic).  Also: bucket ranges tile the whole gradient arena (padded detection-head
slack included), the broadcast makes rank 1 start from rank 0's weights, gradient accumulation (two backwards, no zero_grad)
gives twice the mean, BN running statistics stay per rank.
"""
import os
import socket
import traceback

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make(arch, act_dtype, salt=0):
    from mobilenet_yolo_pytorch_amd import mbv3, synthetic, yolo
    from oracle import procedural
    cls = yolo if arch == "mbv2" else mbv3.yolo
    m = cls(synthetic.VOC_CONFIG, act_dtype=act_dtype)
    procedural.fill_state_dict_(m, salt=salt)
    return m.cuda().train()


def _shard(rank, bs, size):
    from mobilenet_yolo_pytorch_amd import synthetic
    return synthetic.images(bs, size, size, seed=40 + rank).cuda(), synthetic.targets(bs, seed=50 + rank, empty_every=3)


def _grads(m):
    return {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in m.named_parameters()}


def _step(m, x, tg):
    out = m(x, tg)
    (out[0][0] + out[1][0]).backward()
    return out


def _assert_mean(got, parts, what, scale=1.0):
    for k, g in got.items():
        if parts[0][k] is None:
            assert g is None, (what, k)
            continue
        want = sum(p[k].double() for p in parts) / len(parts) * scale
        err = (g.double() - want).abs().max().item()
        tol = 4e-7 * want.abs().max().item() + 1e-12        # one fp32 rounding of the sum (+ one of the scale)
        assert err <= tol, (what, k, err, tol)


def _worker_two_rank(rank, world, port, arch, bf16, q):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from mobilenet_yolo_pytorch_amd import dp
        adt = torch.bfloat16 if bf16 else torch.float32
        bs, size = 4, 96
        # the sequential single-rank reference: both shards, one process, no reducer
        ref_grads, ref_state = [], []
        for r in range(world):
            m0 = _make(arch, adt)
            _step(m0, *_shard(r, bs, size))
            ref_grads.append(_grads(m0))
            ref_state.append({k: v.clone() for k, v in m0.state_dict().items()})
            del m0
        # the data-parallel step: rank 1 starts from DIFFERENT weights, the broadcast must overwrite them
        m = _make(arch, adt, salt=0 if rank == 0 else 99)
        red = dp.attach_data_parallel(m, n_buckets=4)
        x, tg = _shard(rank, bs, size)
        _step(m, x, tg)
        red.wait()
        torch.cuda.synchronize()
        _assert_mean(_grads(m), ref_grads, "dp step")
        plan = next(iter(m._plans.values()))
        pr = red.for_plan(plan)
        b = pr.buckets
        assert b[0][0] == 0 and b[-1][1] == plan.gflat.numel(), (b, plan.gflat.numel())
        assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1)) and len(b) >= 2, b
        assert all(b[i][2] <= b[i + 1][2] for i in range(len(b) - 1)) and b[-1][2] <= len(plan.bwd.calls), b
        ext = dp.PlanReducer.slot_extents(plan)
        assert sum(ext) == plan.gflat.numel()
        assert any(e > (plan.grad_slots[n][1] + 3) // 4 * 4 for e, n in zip(ext, plan.grad_params)), \
            "expected slack behind the padded detection-head gradients (the layout the reducer must follow)"
        ready = dp.PlanReducer._ready_calls(plan)
        assert all(1 <= c <= len(plan.bwd.calls) for c in ready), "a gradient slot no backward call writes"
        # BN running statistics are per rank (no SyncBN): equal to this rank's own single-process run
        sd = m.state_dict()
        for k, v in ref_state[rank].items():
            if "running_" in k:
                assert torch.equal(sd[k], v), ("running stats", k)
        # gradient accumulation: a second backward without zero_grad adds the second averaged gradient
        _step(m, x, tg)
        red.wait()
        torch.cuda.synchronize()
        _assert_mean(_grads(m), ref_grads, "accumulated", scale=2.0)
        # zero_grad(set_to_none=False) keeps the arena views: next step overwrites
        for p in m.parameters():
            if p.grad is not None:
                p.grad.zero_()
        m.zero_grad(set_to_none=True)
        _step(m, x, tg)
        red.wait()
        torch.cuda.synchronize()
        _assert_mean(_grads(m), ref_grads, "after zero_grad")
        q.put((rank, "ok"))
    except Exception:                                   # noqa: BLE001
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _worker_one_rank_rccl(rank, world, port, arch, bf16, q):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
        from mobilenet_yolo_pytorch_amd import dp
        adt = torch.bfloat16 if bf16 else torch.float32
        x, tg = _shard(0, 4, 96)
        m0 = _make(arch, adt)
        _step(m0, x, tg)
        want = _grads(m0)
        m = _make(arch, adt)
        red = dp.attach_data_parallel(m, n_buckets=4)
        assert red.for_plan is not None
        for step in range(3):                           # later steps go through wait-before-rewrite
            m.zero_grad(set_to_none=True)
            _step(m, x, tg)
            red.wait()
            torch.cuda.synchronize()
            got = _grads(m)
            if step == 0:
                for k, g in got.items():
                    assert (g is None) == (want[k] is None), k
                    assert g is None or torch.equal(g, want[k]), ("rccl 1-rank", k)
        plan = next(iter(m._plans.values()))
        pr = red.for_plan(plan)
        assert pr.ar.backend == "nccl" and pr.ar.avg is not None
        assert pr.buckets[-1][1] == plan.gflat.numel()
        q.put((rank, "ok"))
    except Exception:                                   # noqa: BLE001
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _worker_two_sizes(rank, world, port, arch, bf16, q):
    """data-parallel + multi-scale training.  The plan cache evicts by resident bytes; the reducer of an evicted plan
    must not keep it (and its activations / gradient buffers) alive."""
    try:
        import gc
        import weakref
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from mobilenet_yolo_pytorch_amd import dp, optim
        m = _make(arch, torch.float32)
        red = dp.attach_data_parallel(m, n_buckets=3)
        opt = optim.AdamW(m.parameters(), lr=1e-4)
        bs = 4
        x, tg = _shard(rank, bs, 128)
        opt.zero_grad(set_to_none=True)
        _step(m, x, tg)
        opt.step()                                           # no explicit red.wait(): the optimizer-step pre-hook waits (train.py:282-283 literally)
        plan1 = m._plans[(bs, 128, 128, True)]
        assert not plan1._dp.ar.pending
        w1 = weakref.ref(plan1)
        bytes1 = plan1.resident_bytes
        assert bytes1 > 0
        del plan1
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        m.PLAN_BUDGET_FRAC = 1e-9                            # any second plan exceeds the budget -> the oldest plan is evicted
        x2, tg2 = _shard(rank, bs, 96)
        opt.zero_grad(set_to_none=True)                      # drops p.grad views into plan 1's arena
        out = _step(m, x2, tg2)
        opt.step()
        del out
        gc.collect()
        torch.cuda.synchronize()
        assert list(m._plans) == [(bs, 96, 96, True)], list(m._plans)
        assert w1() is None, "the evicted plan is still referenced (reducer / autograd context / gradient views)"
        assert len(red.live) == 1
        after = torch.cuda.memory_allocated()
        plan2 = m._plans[(bs, 96, 96, True)]
        assert after <= before - bytes1 + plan2.resident_bytes + (8 << 20), (before, after, bytes1, plan2.resident_bytes)
        # and the second size still trains data-parallel: gradients equal on both ranks
        g = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
        red.wait()
        torch.cuda.synchronize()
        gl = [torch.empty_like(g).cpu() for _ in range(world)]
        dist.all_gather(gl, g.cpu())
        assert torch.equal(gl[0], gl[1])
        q.put((rank, "ok"))
    except Exception:                                   # noqa: BLE001
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _run(target, world, arch, bf16, timeout=420):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, arch, bf16, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=timeout) for _ in procs]
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()                                # exact children started above
    assert sorted(res) == [(r, "ok") for r in range(world)], "\n".join(str(r[1]) for r in res)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("arch,bf16", [("mbv2", False), ("mbv3", False), ("mbv2", True)])
def test_two_ranks_one_gpu_grads_equal_mean_of_single_rank_grads(arch, bf16):
    _run(_worker_two_rank, 2, arch, bf16)


@pytest.mark.timeout(600)
def test_one_rank_rccl_reducer_leaves_gradients_identical():
    _run(_worker_one_rank_rccl, 1, "mbv2", False)


@pytest.mark.timeout(600)
def test_two_ranks_two_sizes_evicted_plan_is_released_and_optimizer_waits():
    _run(_worker_two_sizes, 2, "mbv2", False)
