#!/usr/bin/env python3
"""bench.py — MobileNetV2-YOLO 352x352 fwd+bwd @ bs256 per GPU (BASELINE.json configs[1] / [2]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = zero_grad -> forward(images, targets) (network + both on-device YOLO losses) -> backward, on a
fixed synthetic batch resident in HBM.  With N>1 every rank runs its own 256-image shard
(weak scaling) and the fp32 gradient arena is averaged with RCCL all-reduce buckets launched from inside
the backward pass.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     — the dominant kernel of the step (by measured time), bracketed by HIP events on the
                 launch stream during the timed region (every launch of it in every 4th timed step,
                 --bracket-every); achieved = algorithmic FLOPs (or bytes) of those launches / their
                 summed duration.
  cpu_baseline — the oracle's torch-CPU port of the reference (oracle/net_ref.py), same synthetic
                 distribution, bounded sample, timed on this host's cores (rank 0, N=1 only).
  nms          — BASELINE config 5: 100k boxes x 20 classes per-class NMS, GPU boxes/s and the
                 single-thread C restatement of torchvision's CPU kernel beside it.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def kfd_gpu_count():
    """GPUs of this node as the kernel driver lists them (/sys/class/kfd/kfd/topology/nodes/*/properties, simd_count > 0 = a GPU
    node; CPU nodes carry simd_count 0).  Reads sysfs only: no HIP, no amdsmi, no torch — the launcher parent must not initialise
    the GPU (on this pool a process that has must never be replaced or fork GPU work).  None when the topology is unreadable."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(root)
    except OSError:
        return None
    n = 0
    for d in nodes:
        try:
            for line in open(os.path.join(root, d, "properties")):
                f = line.split()
                if len(f) == 2 and f[0] == "simd_count" and int(f[1]) > 0:
                    n += 1
        except (OSError, ValueError):
            pass
    return n


def self_launch(n):
    """Run this very command under `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` (127.0.0.1 rendezvous, a free
    port) as a CHILD process and return its exit code.  Runs BEFORE `import torch`: the parent makes no torch / HIP / amdsmi call
    at all (round 3 called torch.cuda.device_count() here, which falls back to hipGetDeviceCount = HIP initialisation when amdsmi
    is unavailable); devices are counted from sysfs.  `--launch-check` asserts that (tests/test_host_logic_cpu.py)."""
    import socket
    import subprocess
    check = "--launch-check" in sys.argv
    have = kfd_gpu_count()
    if not check:
        if not have:
            print("bench: --gpus %d but this node exposes %s GPU(s) (kfd topology)" % (n, "no" if have is None else have), file=sys.stderr)
            return 2
        if have < n:                              # partitioned / filtered nodes can miscount: warn, let the ranks fail loudly at set_device
            print("bench: --gpus %d but the kfd topology lists %d GPU node(s); starting the ranks anyway" % (n, have), file=sys.stderr)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL's cross-process buffer sharing needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    if check:
        bad = sorted(m for m in sys.modules if m == "torch" or m.startswith("torch.") or m == "amdsmi" or m.startswith("mobilenet_yolo_pytorch_amd"))
        assert not bad, "launcher parent imported GPU-capable modules before starting its ranks: %s" % bad[:5]
        env["MNY_LAUNCHER_PARENT_CLEAN"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _gpus_arg(argv):
    for i, t in enumerate(argv):
        if t == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if t.startswith("--gpus="):
            return int(t.split("=", 1)[1])
    return 1


if __name__ == "__main__" and "RANK" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) == 1 and _gpus_arg(sys.argv[1:]) > 1:
    # `python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (one per GPU, RCCL over xGMI) before this
    # process has imported torch or made any GPU call, hand their output through (rank 0 prints the one JSON line), exit with their code
    sys.exit(self_launch(_gpus_arg(sys.argv[1:])))

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak
PROFILE_TAG = "r06"               # profiles/<tag>_traffic_*.json: PMC passes of this round (tools/prof_round.sh + tools/prof_summary.py)
BATCH = 256
SIZE = 352

# entry points bracketed by HIP events inside the timed region: the dominant one only (forward + plain data-gradient GEMMs, 62 launches
# per step) — every bracket costs the GPU a few microseconds of lost back-to-back dispatch, so nothing else is bracketed there
# the pure NT GEMM entry point (forward + data gradient).  mny_pw_dgrad_bnred (data gradient + the fed unit's BN-backward
# reduction in the epilogue: MFMA work plus an HBM-bound read of that unit's output) is a different entry point with its own,
# distinct kernel instantiations (pw_gemm_nt_dma_kernel<..., RED = 1>) — it is not priced against the MFMA peak
MFMA_KERNELS = {"mny_pw_fwd", "mny_pw_fwd_bf16"}
# priced in a SECOND, untimed pass of K event-bracketed steps (brackets cost the GPU its back-to-back dispatch, so they stay out
# of the timed region): the depthwise forward against HBM (north-star target >= 60 %), the other two GEMM entry points against MFMA
# round 4: the fused depthwise backward (three entry points, 7.3 ms of the round-3 step, priced nowhere before) against HBM, and the expand +
# depthwise unit (csrc/exdw.hip: instruction-bound recomputation, no longer an HBM stream) against the fp32 matrix / vector peak; its
# statistics pass reads the thin input once (second-moment matrix, no recomputation of the expand output) and is priced against HBM
SECOND_PASS = {"mny_dw_fwd": "hbm", "mny_pw_wgrad": "mfma", "mny_pw_dgrad_bnred": "mfma", "mny_pw_bnbwd": "hbm",
               "mny_dw_bnbwd": "hbm", "mny_dw_bnbwd_red": "hbm", "mny_dw_bnbwd_s2": "hbm",
               "mny_exdw_stats": "hbm", "mny_exdw_fwd": "mfma", "mny_exdw_bwd": "mfma", "mny_stemdw_bwd": "hbm", "mny_pj_bwd": "hbm",
               "mny_dw_fwd_bf16": "hbm", "mny_pw_wgrad_bf16": "mfma", "mny_pw_dgrad_bnred_bf16": "mfma"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="images per GPU (the metric is quoted at 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-nms", action="store_true")
    ap.add_argument("--arch", default="mbv2", choices=["mbv2", "mbv3"], help="mbv3 = BASELINE config 4 topology (fp32 here), not the headline")
    ap.add_argument("--size", type=int, default=SIZE)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="activation storage type; bf16 = BASELINE config 4 (with --arch mbv3 --size 512), never the headline")
    ap.add_argument("--roofline-pass", choices=["inline", "after"], default="inline",
                    help="inline: bracket the MFMA kernels with HIP events inside the timed steps (eager replay); "
                         "after: time K hipGraph-replayed steps, then K more event-bracketed steps for the roofline object")
    ap.add_argument("--bracket-every", type=int, default=4,
                    help="inline roofline pass: bracket the dominant kernels with HIP events in every B-th timed step (1 = every step)")
    ap.add_argument("--breakdown", action="store_true", help="print a per-entry-point time table to stderr")
    ap.add_argument("--launch-check", action="store_true",
                    help="CPU-only self-test of the N-rank launch path: the ranks meet over gloo, rank 0 prints one JSON line (tests/test_host_logic_cpu.py)")
    ap.add_argument("--dp-overhead-child", action="store_true", help="internal: the 1-rank RCCL leg of the default run, in its own process")
    ap.add_argument("--full-json", action="store_true", help="print the verbose line (explanatory strings, plan fingerprints, algorithmic bytes per "
                    "entry point: what tools/prof_*summary.py read) instead of the compact one sized for an 8 kB stdout tail")
    ap.add_argument("--detail", default="", help="comma list of entry points (or `all`): print their per-call table to stderr")
    return ap.parse_args()


def make_batch(batch, rank, device, size=SIZE):
    from mobilenet_yolo_pytorch_amd import synthetic
    x = synthetic.images(batch, size, size, seed=rank).to(device)
    tg = synthetic.targets(batch, seed=1 + rank, empty_every=16)
    return x, tg


def usable_cores():
    """Cores this process may actually use: affinity mask, capped by a cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(budget_s=30.0):
    """Reference semantics on the host CPU, as BASELINE.md §2 states it: oracle/net_ref.py (the stock torch ops the reference
    calls + the restated loss), fwd+bwd at bs=16, all usable cores, 2 warm-up + 5 timed steps, MEDIAN; host CPU model named.
    Bounded: if one step at all cores takes so long that 7 steps would blow the budget (torch's CPU conv kernels thrash far
    below 256 threads), the thread count is halved until a step fits and the sample string says so."""
    import statistics
    from oracle import net_ref, procedural
    torch.manual_seed(0)
    cores = usable_cores()
    m = net_ref.RefYolo(procedural.VOC_CONFIG).train()
    bs = 16
    x = procedural.images(bs, SIZE, SIZE, seed=0)
    tg = procedural.targets(bs, seed=1, empty_every=16)

    def step():
        for p in m.parameters():
            p.grad = None
        t0 = time.perf_counter()
        r = m(x, tg)
        (r[0][0] + r[1][0]).backward()
        return time.perf_counter() - t0
    threads, note = cores, ""
    torch.set_num_threads(threads)
    step()                                     # warm-up 1 (allocator, oneDNN primitive cache)
    w = step()                                 # warm-up 2
    while w * 5 > budget_s and threads > 8:
        threads = max(8, threads // 2)
        torch.set_num_threads(threads)
        step()
        w = step()
        note = "; thread count reduced from %d: a step at all cores took too long for the bounded sample" % cores
    times = [step() for _ in range(5)]
    med = statistics.median(times)
    return {"value": round(bs / med, 3), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": "median of 5 fwd+bwd steps (2 warm-up) at bs=%d, 352x352, oracle/net_ref.py (torch CPU fp32, %d threads; "
                      "host: %s, os.cpu_count()=%d, usable=%d)%s" % (bs, threads, cpu_model_name(), os.cpu_count() or 0, cores, note),
            "min_ms": round(min(times) * 1e3, 1), "median_ms": round(med * 1e3, 1)}


def cpu_baseline_c3(budget_s=30.0):
    """configs[3] beside its GPU number: oracle/net_ref_v3.py (the stock torch ops of models/mobilenetv3.py / mbv3_yolo.py + the restated
    loss) fwd+bwd at bs=4, 512x512, torch CPU fp32 (the CPU port has no bf16 storage mode), 1 warm-up + up to 3 timed steps inside the budget."""
    import statistics
    from oracle import net_ref_v3, procedural
    torch.manual_seed(0)
    cores = usable_cores()
    torch.set_num_threads(cores)
    m = net_ref_v3.RefYoloV3(procedural.VOC_CONFIG).train()
    bs = 4
    x = procedural.images(bs, 512, 512, seed=0)
    tg = procedural.targets(bs, seed=1, empty_every=16)

    def step():
        for p in m.parameters():
            p.grad = None
        t0 = time.perf_counter()
        r = m(x, tg)
        (r[0][0] + r[1][0]).backward()
        return time.perf_counter() - t0
    w = step()
    times = [step()]
    while len(times) < 3 and (len(times) + 1) * max(w, times[0]) < budget_s:
        times.append(step())
    med = statistics.median(times)
    return {"value": round(bs / med, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "median of %d fwd+bwd steps (1 warm-up) at bs=%d, 512x512, oracle/net_ref_v3.py, torch CPU fp32, %d threads" % (len(times), bs, cores),
            "median_ms": round(med * 1e3, 1)}


def c1_leg(device):
    """BASELINE configs[0]: MobileNetV2-YOLO 352x352 bs=2 inference (eval forward + anchor decode + per-class NMS, inference.py) — the
    reference's CPU-runnable case: the product on the GPU and the oracle (torch CPU + oracle/yolo_ref.py + oracle/nms_ref.c) on the host."""
    import statistics
    from mobilenet_yolo_pytorch_amd import yolo
    from oracle import net_ref, procedural
    torch.manual_seed(0)
    cores = usable_cores()
    torch.set_num_threads(cores)
    ref = procedural.fill_state_dict_(net_ref.RefYolo(procedural.VOC_CONFIG)).eval()
    m = yolo(procedural.VOC_CONFIG)
    m.load_state_dict(ref.state_dict())
    m = m.to(device).eval()
    x = procedural.images(2, SIZE, SIZE, seed=3)
    xd = x.to(device)
    with torch.no_grad():
        det_ref = ref(x)
        det = m(xd)
        cpu = []
        for _ in range(5):
            t0 = time.perf_counter()
            ref(x)
            cpu.append(time.perf_counter() - t0)
        for _ in range(5):
            m(xd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            m(xd)
        torch.cuda.synchronize()
        gpu = (time.perf_counter() - t0) / 20
    same = all(len(a) == len(b) for a, b in zip(det, det_ref))
    cm = statistics.median(cpu)
    return {"workload": "MobileNetV2-YOLO 352x352 bs=2 eval forward + decode + per-class NMS (BASELINE configs[0])",
            "value": round(2 / gpu, 1), "unit": "images/s", "ms": round(gpu * 1e3, 3), "detections": int(sum(len(d) for d in det)),
            "counts_match_cpu": bool(same),
            "cpu_baseline": {"value": round(2 / cm, 2), "unit": "images/s", "cores": cores, "kind": "port", "median_ms": round(cm * 1e3, 1),
                             "sample": "median of 5 runs, oracle/net_ref.py eval + yolo_ref decode + nms_ref.c, torch CPU fp32"}}


def h2d_bench(step, x, steps=10):
    """PCIe-inclusive rate (never `value`): the batch starts in pinned host memory and is copied to the device before every
    step, on the compute stream (no overlap) and on a side stream one step ahead (the copy hides behind the previous step)."""
    host = x.cpu().pin_memory()
    dev = x
    out = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        dev.copy_(host, non_blocking=True)
        step()
    torch.cuda.synchronize()
    out["serial_ms_per_step"] = round((time.perf_counter() - t0) / steps * 1e3, 3)
    side = torch.cuda.Stream()
    stage = torch.empty_like(x)
    ev_copy, ev_used = torch.cuda.Event(), torch.cuda.Event()
    ev_used.record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        with torch.cuda.stream(side):
            side.wait_event(ev_used)                  # `stage` may be overwritten once the previous consumer copy is done
            stage.copy_(host, non_blocking=True)
            ev_copy.record(side)
        torch.cuda.current_stream().wait_event(ev_copy)
        dev.copy_(stage, non_blocking=True)          # device-to-device, 0.1 ms
        ev_used.record()
        step()
    torch.cuda.synchronize()
    out["prefetch_ms_per_step"] = round((time.perf_counter() - t0) / steps * 1e3, 3)
    out["batch_mb"] = round(x.numel() * 4 / 1e6, 1)
    out["note"] = "input batch copied from pinned host memory every step; not part of `value`"
    return out


def optimizer_bench(model):
    """AdamW.step() on the model's parameters (train.py:283), reported beside the fwd+bwd metric, never inside it:
    the fused multi-tensor kernel and torch.optim.AdamW on clones of the same tensors."""
    from mobilenet_yolo_pytorch_amd.optim import AdamW
    kw = dict(lr=7e-4, weight_decay=4e-4)
    ref = [torch.nn.Parameter(p.detach().clone()) for p in model.parameters()]
    for r, p in zip(ref, model.parameters()):
        r.grad = None if p.grad is None else p.grad.detach().clone()
    out = {}
    for name, opt in (("fused_ms", AdamW(model.parameters(), **kw)), ("torch_adamw_ms", torch.optim.AdamW(ref, **kw))):
        for _ in range(3):
            opt.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            opt.step()
        torch.cuda.synchronize()
        out[name] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
    out["tensors"] = sum(p.grad is not None for p in model.parameters())
    out["parameters"] = sum(p.numel() for p in model.parameters() if p.grad is not None)
    out["note"] = "not part of `value` (the metric is fwd+loss+bwd)"
    return out


def nms_bench(device):
    """BASELINE config 5: one image-segment of 100,000 rows, 20 classes."""
    import numpy as np
    from mobilenet_yolo_pytorch_amd import ops
    from oracle import nms_ref
    r = np.random.RandomState(2)
    n, C = 100000, 20
    ctr = r.rand(n, 2).astype(np.float32)
    wh = (0.02 + 0.28 * r.rand(n, 2)).astype(np.float32)
    rows = np.concatenate((ctr - wh / 2, ctr + wh / 2, r.rand(n, 2).astype(np.float32), r.randint(0, C, (n, 1)).astype(np.float32)), 1)
    rows_t = torch.from_numpy(rows.astype(np.float32))
    dev_rows = rows_t.to(device)
    beg = torch.zeros(1, dtype=torch.int32, device=device)
    cnt = torch.full((1,), n, dtype=torch.int32, device=device)
    out = ops.nms_per_class(dev_rows, beg, cnt, C)
    torch.cuda.synchronize()
    assert int(out[4].item()) == 0
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        out = ops.nms_per_class(dev_rows, beg, cnt, C)
    torch.cuda.synchronize()
    gpu_dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    _, ref_idx = nms_ref.nms_rows(rows_t, C, 0.45)
    cpu_dt = time.perf_counter() - t0
    kept = int(out[1].item())
    same = kept == len(ref_idx) and torch.equal(out[0][:kept].cpu().long(), ref_idx)
    res = {"workload": "per-class NMS, 100000 boxes x 20 classes, thr 0.45", "boxes_per_s": round(n / gpu_dt, 1),
           "ms": round(gpu_dt * 1e3, 3), "kept": kept, "matches_cpu_indices": bool(same),
           "cpu_boxes_per_s": round(n / cpu_dt, 1), "cpu_kind": "port (oracle/nms_ref.c, 1 thread)"}
    # the reference-shaped case: one eval batch of 256 images x 1 815 candidates (utils/box.py runs 256 x 20 torchvision.ops.nms calls here)
    S, m = 256, 1815
    rr = np.random.RandomState(3)
    ctr2 = rr.rand(S * m, 2).astype(np.float32)
    wh2 = (0.02 + 0.28 * rr.rand(S * m, 2)).astype(np.float32)
    rows2 = torch.from_numpy(np.concatenate((ctr2 - wh2 / 2, ctr2 + wh2 / 2, rr.rand(S * m, 2).astype(np.float32),
                                             rr.randint(0, C, (S * m, 1)).astype(np.float32)), 1).astype(np.float32))
    dev2 = rows2.to(device)
    beg2 = (torch.arange(S, dtype=torch.int32) * m).to(device)
    cnt2 = torch.full((S,), m, dtype=torch.int32, device=device)
    o2 = ops.nms_per_class(dev2, beg2, cnt2, C, max_seg_rows=m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        o2 = ops.nms_per_class(dev2, beg2, cnt2, C, max_seg_rows=m)
    torch.cuda.synchronize()
    g2 = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    same2, kept2 = True, 0
    oc = o2[1].cpu().numpy()
    oi = o2[0].cpu().numpy()
    for si in range(S):
        _, ri = nms_ref.nms_rows(rows2[si * m:(si + 1) * m], C, 0.45)
        kept2 += len(ri)
        same2 = same2 and oc[si] == len(ri) and np.array_equal(oi[si * m:si * m + oc[si]] - si * m, ri.numpy())
    c2 = time.perf_counter() - t0
    res["reference_shape"] = {"workload": "per-class NMS, 256 images x 1815 candidates x 20 classes (one eval batch, utils/box.py:11-31)",
                              "boxes_per_s": round(S * m / g2, 1), "ms": round(g2 * 1e3, 3), "kept": int(kept2), "matches_cpu_indices": bool(same2),
                              "cpu_boxes_per_s": round(S * m / c2, 1), "cpu_kind": "port (oracle/nms_ref.c, 1 thread, incl. the python per-image driver)"}
    # anchor decode (C5 i): both heads of 55 images at 352x352 = 99,825 candidates, confidence threshold low enough that ~all pass
    from mobilenet_yolo_pytorch_amd import synthetic
    y = synthetic.VOC_CONFIG["yolo"]
    N, nc = 55, y["num_classes"]
    anchors = torch.tensor([(aw / SIZE, ah / SIZE) for aw, ah in y["anchors"]], dtype=torch.float32, device=device)
    g = torch.Generator().manual_seed(4)
    heads, hps, masks = [], [], []
    for hi, grid in enumerate((SIZE // 32, SIZE // 16)):
        heads.append(torch.randn(N, grid, grid, len(y["mask"][hi]) * (5 + nc), generator=g).to(device))
        hps.append(ops.make_head(N, grid, len(y["mask"][hi]), nc, len(y["anchors"]), 0.5, 0.213, 0.01))
        masks.append(torch.tensor(y["mask"][hi], dtype=torch.int32, device=device))
    cap = sum(hp.A * hp.g * hp.g for hp in hps)
    rows_d = torch.zeros(N, cap, 7, device=device)

    def decode():
        _, c0 = ops.yolo_decode(heads[0], anchors, masks[0], hps[0], 1e-6, rows=rows_d, row_stride=cap)
        _, c1 = ops.yolo_decode(heads[1], anchors, masks[1], hps[1], 1e-6, rows=rows_d, row_stride=cap, base_counts=c0)
        return c1
    c1 = decode()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        c1 = decode()
    torch.cuda.synchronize()
    dd = (time.perf_counter() - t0) / 10
    res["decode"] = {"workload": "anchor decode + confidence compaction, 2 heads x 55 images @352 (%d candidates)" % (N * cap),
                     "candidates_per_s": round(N * cap / dd, 1), "ms": round(dd * 1e3, 3), "passed": int(c1.sum().item())}
    return res


def map_bench(device):
    """VOC07-test shaped evaluation set (4 952 images, 20 classes + background) scored on the device;
    CPU baseline = the oracle port of utils/eval_mAP.py on a bounded sample of the same set."""
    import numpy as np
    from mobilenet_yolo_pytorch_amd import evalmap, synthetic
    from oracle import map_ref
    nc, n_img = 21, 4952
    case = synthetic.map_case(n_img, nc, seed=9, clutter=30.0)
    dev = [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in case]
    r = evalmap.map_eval(*dev, nc)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        r = evalmap.map_eval(*dev, nc)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    D = int(case[1].size)
    t0 = time.perf_counter()
    ap_o, m_o, tp_o, fp_o, p11_o = map_ref.calculate_map(*case, nc)             # the whole set: ~1.5 s of numpy
    cpu_dt = time.perf_counter() - t0
    same = bool(np.array_equal(r["tp"].cpu().numpy(), tp_o) and np.array_equal(r["fp"].cpu().numpy(), fp_o)
                and np.array_equal(r["prec11"].cpu().numpy(), p11_o) and abs(float(r["mean_ap"]) - float(m_o)) < 1e-6)
    return {"workload": "VOC07 11-point mAP, %d images, %d detections, %d objects, 20 classes" % (n_img, D, int(case[5].size)),
            "detections_per_s": round(D / dt, 1), "ms": round(dt * 1e3, 3), "mean_ap": round(float(r["mean_ap"]), 6),
            "matches_cpu_port": same, "cpu_detections_per_s": round(D / cpu_dt, 1),
            "cpu_kind": "port (oracle/map_ref.py, numpy, 1 thread, same set; the reference's own torch loop ran 17 k detections/s in the build container)"}


def prep_bench(device):
    """one training batch of decoded VOC-shaped photos (256 images, 500x375 / 375x500 / smaller) -> 352x352
    normalised NCHW fp32.  Device leg: sources already in HBM.  CPU: the real Pillow resize + the ToTensor/Normalize tensor
    ops (what collate_fn runs per image in the DataLoader workers), one thread, bounded sample."""
    import numpy as np
    from mobilenet_yolo_pytorch_amd import prep, synthetic
    from oracle import prep_ref
    r = np.random.RandomState(6)
    shapes = [(375, 500), (500, 375), (333, 500), (500, 334), (281, 500), (375, 440)]
    protos = synthetic.photos(shapes, seed=6)
    imgs = [protos[i % len(protos)] for i in range(BATCH)]
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    bp = prep.BatchPrep([(SIZE, SIZE)], mean, std, device=device)
    stage, desc, mh, mw = bp.pack(imgs)
    src = stage.to(device)
    desc_dev = torch.from_numpy(desc.view(np.uint8).copy()).to(device)
    out = torch.empty(BATCH, 3, SIZE, SIZE, device=device)
    bp.run_device(src, desc_dev, BATCH, mh, mw, (SIZE, SIZE), out=out)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        bp.run_device(src, desc_dev, BATCH, mh, mw, (SIZE, SIZE), out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(3):
        x = bp(imgs, (SIZE, SIZE))                    # pack into pinned memory + H2D + kernels
    torch.cuda.synchronize()
    dt_h = (time.perf_counter() - t0) / 3
    k = len(protos)
    try:
        from PIL import Image
        t0 = time.perf_counter()
        n_cpu = 0
        while time.perf_counter() - t0 < 3.0:
            for im in protos:
                rz = np.asarray(Image.fromarray(im).resize((SIZE, SIZE), Image.BILINEAR))
                t = torch.from_numpy(rz.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
                t.sub_(torch.tensor(mean)[:, None, None]).div_(torch.tensor(std)[:, None, None])
            n_cpu += k
        cpu_dt = (time.perf_counter() - t0) / n_cpu
        same = bool(torch.equal(out[k - 1].cpu(), t))
        kind = "reference pieces (Pillow %s resize + torch ToTensor/Normalize ops, 1 thread, %d images)" % (getattr(Image, "__version__", "?"), n_cpu)
    except ImportError:
        t0 = time.perf_counter()
        ref = prep_ref.collate(protos[:2], (SIZE, SIZE), mean, std)
        cpu_dt = (time.perf_counter() - t0) / 2
        same = bool(np.array_equal(out[:2].cpu().numpy(), ref))
        kind = "port (oracle/prep_ref.py, numpy, 2 images)"
    src_bytes = int(sum(im.size for im in imgs))
    return {"workload": "%d decoded photos (%.0f MB uint8) -> %dx%d normalised NCHW fp32" % (BATCH, src_bytes / 1e6, SIZE, SIZE),
            "images_per_s": round(BATCH / dt, 1), "ms": round(dt * 1e3, 3),
            "hbm_gbs": round((src_bytes + 2 * BATCH * 420 * SIZE * 3 + BATCH * 3 * SIZE * SIZE * 4) / dt / 1e9, 1),
            "with_pack_and_h2d_images_per_s": round(BATCH / dt_h, 1), "matches_cpu": same,
            "cpu_images_per_s": round(1.0 / cpu_dt, 1), "cpu_kind": kind}


def roofline_from(events, calls_by_list):
    """Aggregate HIP-event durations per entry point; pick the dominant one."""
    agg = {}
    for which, evs in events.items():
        calls = calls_by_list[which]
        for idx, name, a, b in evs:
            ms = a.elapsed_time(b)
            meta = calls[idx][3] or {}
            d = agg.setdefault(name, {"ms": 0.0, "n": 0, "flops": 0, "bytes": 0})
            d["ms"] += ms
            d["n"] += 1
            d["flops"] += meta.get("flops", 0)
            d["bytes"] += meta.get("bytes", 0)
    return agg


def plan_signature(plan, entry):
    """Stable fingerprint of the launches of one entry point in a plan (shapes in launch order): the committed PMC traffic
    figure is only valid for exactly this launch list."""
    import hashlib
    items = []
    for calls in (plan.fwd.calls, plan.bwd.calls):
        for _fn, _args, name, meta in calls:
            if name == entry:
                items.append((meta or {}).get("shape", "?"))
    return len(items), hashlib.sha256("|".join(items).encode()).hexdigest()[:16]


def step_signature(plan):
    """Fingerprint of the WHOLE step (every call of both lists, in order, with its shape): whole-step counter traffic is only valid
    for exactly this launch list."""
    import hashlib
    # (device launches only: the host-side fork / join steps of the side stream are absent from the single-stream runs the profile
    # scripts make — counting them made every committed whole-step profile look stale against the default plan, rounds 4 and 5)
    # ... and, round 6, as a MULTISET: the single-stream profile runs place the first head's loss at the end of the forward list, the default
    # two-stream plan right behind that head — the same launches in another order)
    items = sorted("%s:%s" % (name, (meta or {}).get("shape", "")) for calls in (plan.fwd.calls, plan.bwd.calls) for _fn, _args, name, meta in calls
                   if name not in ("fork", "join", "py"))
    return len(items), hashlib.sha256("|".join(items).encode()).hexdigest()[:16]


def committed_traffic(plan, entry, headline):
    """HBM bytes per launch of `entry` from the PMC counters.  bench.py cannot run rocprofv3 on itself, so this is the committed
    result of the prescribed separate --pmc passes over this very command (profiles/README.md) — accepted only while the plan
    still launches exactly the kernels that were profiled; otherwise null and a loud note."""
    path = os.path.join(HERE, "profiles", "%s_traffic_%s.json" % (PROFILE_TAG, entry))
    if not headline or not os.path.exists(path):
        return None, None
    tj = json.load(open(path))
    n, sig = plan_signature(plan, entry)
    if tj.get("launches_per_step") != n or tj.get("plan_signature") != sig:
        print("bench: %s is STALE (profiled %s launches / signature %s, plan now has %d / %s): roofline.traffic left null — "
              "re-run tools/prof_round.sh" % (os.path.basename(path), tj.get("launches_per_step"), tj.get("plan_signature"), n, sig), file=sys.stderr)
        return None, "stale: " + os.path.basename(path)
    return round(tj["hbm_bytes_per_launch"]), "profiles/%s: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + --pmc WRITE_SIZE, separate passes" % os.path.basename(path)


def roofline_object(name, d, steps, bound, bf16, measured):
    secs = d["ms"] * 1e-3
    o = {"kernel": name, "bound": bound, "launches_per_step": d["n"] // steps, "ms_per_step": round(d["ms"] / steps, 3),
         "algorithmic_bytes_per_launch": round(d["bytes"] / max(d["n"], 1)), "measured": measured}
    if bound == "mfma":
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
        ach = d["flops"] / secs / 1e12
        o.update(achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                 algorithmic_gflop_per_step=round(d["flops"] / steps / 1e9, 2), algorithmic_hbm_gbs=round(d["bytes"] / secs / 1e9, 1))
    else:
        ach = d["bytes"] / secs / 1e9
        o.update(achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 4),
                 algorithmic_gb_per_step=round(d["bytes"] / steps / 1e9, 3))
    o["traffic"] = None
    return o


def config3_leg(device, steps=12, warmup=4):
    """BASELINE configs[3]: MobileNetV3-YOLO 512x512, bf16 activation storage, bs=64 — one extra leg of the default run so the
    driver's own `python bench.py` carries it.  HBM-bound almost everywhere (SURVEY §8d C4 note: machine balance at bf16 is
    312 FLOP/B, one layer of 82 exceeds it), so the honest roofline is algorithmic bytes / time against 8 TB/s."""
    from mobilenet_yolo_pytorch_amd import mbv3, synthetic
    torch.manual_seed(0)
    model = mbv3.yolo(synthetic.VOC_CONFIG, act_dtype=torch.bfloat16).to(device).train()
    bs, size = 64, 512
    x = synthetic.images(bs, size, size, seed=0).to(device)
    tg = synthetic.targets(bs, seed=1, empty_every=16)

    def step():
        for p in model.parameters():
            p.grad = None
        out = model(x, tg)
        (out[0][0] + out[1][0]).backward()
        return out
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    plan = model._plans[(bs, size, size, True, "bf16")]
    by, fl = 0, 0
    for calls in (plan.fwd.calls, plan.bwd.calls):
        for c in calls:
            if (c[3] or {}).get("flops", 0) > 0:       # convolution work only (forward, data / weight gradient); the separate BN passes
                by += c[3].get("bytes", 0)             # of the un-fused units are overhead, not algorithmic traffic
                fl += c[3]["flops"]
    traffic, traffic_src = None, None
    tpath = os.path.join(HERE, "profiles", "%s_traffic_config3.json" % PROFILE_TAG)
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        n_sig, sig = step_signature(plan)
        if tj.get("calls_per_step") == n_sig and tj.get("step_signature") == sig:
            traffic = round(tj["hbm_bytes_per_step"])
            traffic_src = "profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over the whole step (separate passes), calibrated as the file states" % os.path.basename(tpath)
        else:
            print("bench: %s is STALE (profiled %s calls / %s, plan now %d / %s): config3.roofline.traffic left null" % (
                os.path.basename(tpath), tj.get("calls_per_step"), tj.get("step_signature"), n_sig, sig), file=sys.stderr)
            traffic_src = "stale: " + os.path.basename(tpath)
    res = {"workload": "MobileNetV3-YOLO 512x512 bs=64 fwd+loss+bwd, bf16 activation storage (BASELINE configs[3])",
           "value": round(bs / dt, 1), "unit": "images/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps, "warmup": warmup, "dtype": "bf16",
           "loss": round(float(out[0][0].detach()) + float(out[1][0].detach()), 5),
           "roofline": {"bound": "hbm", "achieved": round(by / dt / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(by / dt / 1e9 / PEAK_HBM_GBS, 4),
                        "traffic": traffic, "traffic_source": traffic_src, "algorithmic_gb_per_step": round(by / 1e9, 3), "algorithmic_tflop_per_step": round(fl / 1e12, 3),
                        "launches_per_step": len(plan.fwd.calls) + len(plan.bwd.calls),
                        "note": "whole step: algorithmic bytes of every convolution call of the plan (forward + data gradient + weight gradient; BN / elementwise passes not counted) over the step time"}}
    del model, plan
    torch.cuda.empty_cache()
    return res


_ROOF_KEEP = ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_bf16x6", "traffic", "launches_per_step", "ms_per_step", "algorithmic_bytes_per_launch")


def compact_line(res):
    """The ONE stdout line, sized for a tail of 8 kB.  A tail keeps the END of a line, so the evidence that must survive comes last: side
    legs first, then configs[0] / configs[3], then the roofline objects (worst fraction first), then the contract's scalar keys with
    `roofline` and `cpu_baseline`.  Explanatory strings (how a figure was measured, where its traffic came from: DESIGN.md §4,
    profiles/README.md), plan fingerprints and per-entry-point algorithmic bytes are in the --full-json line only."""
    def roof(o):
        return {k: o[k] for k in _ROOF_KEEP if k in o} if isinstance(o, dict) else o

    def slim(o, drop=("note", "sample", "measured", "traffic_source", "matches_cpu_note", "host")):
        if isinstance(o, dict):
            return {k: slim(v, drop) for k, v in o.items() if k not in drop}
        if isinstance(o, list):
            return [slim(v, drop) for v in o]
        return o
    out = {}
    for k in ("pcie_inclusive", "prep", "map", "optimizer", "nms", "dp_overhead", "data_parallel"):
        if k in res:
            out[k] = slim(res[k])
    if "config0" in res:
        out["config0"] = slim(res["config0"], drop=("note",))
    if "config3" in res:
        c3 = dict(res["config3"])
        if isinstance(c3.get("roofline"), dict):
            c3["roofline"] = {k: v for k, v in c3["roofline"].items() if k not in ("note", "traffic_source")}
        out["config3"] = c3
    more = sorted(res.get("roofline_more", []), key=lambda o: o.get("frac", 1.0))
    if more:
        out["roofline_more"] = [roof(o) for o in more]
    if "roofline_hbm" in res:
        out["roofline_hbm"] = roof(res["roofline_hbm"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        out[k] = res[k]
    if "cpu_baseline" in res:
        out["cpu_baseline"] = res["cpu_baseline"]
    out["roofline"] = roof(res["roofline"])
    return out


def allreduce_model(payload_bytes, n_buckets):
    """Modelled gradient all-reduce time per step for N = 2, 4, 8 (no multi-GPU box is available to the builder; the driver's
    SCALE run is the measurement).  xGMI is point-to-point, ~153 GB/s per link and direction (MI355X_MICROARCH.md); a ring
    all-reduce moves 2 (N-1)/N of the payload over every GPU's outgoing link and pays 2 (N-1) hop latencies per bucket."""
    link, hop_us = 153e9, 6.0
    out = {"payload_mb": round(payload_bytes / 1e6, 2), "buckets": n_buckets, "link_gbs": link / 1e9, "hop_latency_us": hop_us,
           "model": "single ring per bucket: 2(N-1)/N * bytes / link + 2(N-1) * hop latency per bucket; buckets 1..n-1 overlap the rest of backward, "
                    "the last one the next forward"}
    for n in (2, 4, 8):
        out["n%d_ms" % n] = round((2.0 * (n - 1) / n * payload_bytes / link + n_buckets * 2 * (n - 1) * hop_us * 1e-6) * 1e3, 3)
    return out


def dp_overhead_child(a):
    """`bench.py --dp-overhead-child`: the data-parallel code path on ONE rank (RCCL process group of size 1, bucketed all-reduce
    launched from inside backward) against the plain step, same process, same plan, interleaved.  Prints one JSON line.  Runs in its
    OWN process so that an RCCL / rendezvous stall can only cost this leg (the parent kills it at a hard timeout), never the
    headline line."""
    import datetime
    import socket
    import torch.distributed as dist
    from mobilenet_yolo_pytorch_amd import synthetic, yolo
    from mobilenet_yolo_pytorch_amd.dp import attach_data_parallel
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION"):
        os.environ["NCCL_DEBUG"] = "WARN"
    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    sys.stdout.flush()
    saved_fd = os.dup(1)                       # whatever RCCL writes to the C-level stdout goes to stderr: this process prints ONE JSON line
    os.dup2(2, 1)
    res, red, model = None, None, None
    try:
        torch.manual_seed(0)
        model = yolo(synthetic.VOC_CONFIG).to(device).train()
        x, tg = make_batch(a.batch, 0, device, a.size)

        def step():
            for p in model.parameters():
                p.grad = None
            out = model(x, tg)
            (out[0][0] + out[1][0]).backward()

        def timed(k):
            for _ in range(3):
                step()
            if red is not None:
                red.wait()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                step()
            if red is not None:
                red.wait()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / k * 1e3

        k = max(5, min(a.steps, 10))
        plain1 = timed(k)
        dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(seconds=90))
        red = attach_data_parallel(model)
        dp_ms = timed(k)
        plan = next(iter(model._plans.values()))
        nb = len(red.for_plan(plan).buckets)
        payload = plan.gflat.numel() * 4
        red.detach()
        del model.dp_reducer
        plan._dp = None
        red = None
        plain2 = timed(k)                     # the plain step again, right after, so both numbers see the same clocks / box state
        res = {"one_rank_rccl_ms_per_step": round(dp_ms, 3), "plain_ms_per_step": round(plain2, 3), "plain_before_ms_per_step": round(plain1, 3),
               "overhead_frac": round(dp_ms / plain2 - 1.0, 4), "buckets": nb,
               "note": "1-rank RCCL group on this GPU, own child process with a hard timeout: segmented backward replay + %d all-reduce launches per step; "
                       "not part of `value`" % nb,
               "allreduce_model": allreduce_model(payload, nb)}
    finally:
        try:
            if red is not None:
                red.detach()
            if model is not None and hasattr(model, "dp_reducer"):
                del model.dp_reducer
            if dist.is_initialized():
                dist.destroy_process_group()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    print(json.dumps({"dp_overhead": res}))


def dp_overhead_leg(a, timeout_s=240):
    """Run dp_overhead_child in a fresh child process (this process keeps its plans; HBM holds both) and merge its JSON line in."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--dp-overhead-child", "--steps", str(a.steps), "--batch", str(a.batch), "--size", str(a.size)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": "child exceeded its %d s limit and was killed (RCCL / rendezvous stall?)" % timeout_s}
    for line in reversed(r.stdout.splitlines()):
        if line.startswith("{"):
            return json.loads(line)["dp_overhead"]
    return {"error": "child exit %d: %s" % (r.returncode, r.stderr[-300:])}


def dp_evidence_after_warmup(plan, reducer, out, world, rank, device):
    """N > 1 (or MNY_FORCE_DP=1): what shows on the line that RCCL really ran with N ranks and that the reduced arena is the same
    everywhere.  Called after the warm-up steps (>= 1 reduced step), before the timed region."""
    import torch.distributed as dist
    reducer.wait()
    torch.cuda.synchronize()
    g = plan.gflat.double()
    mine = torch.stack([g.sum(), g.abs().sum(), (out[0][0].detach() + out[1][0].detach()).double().reshape(())]).to(device)
    allr = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    allr = torch.stack(allr).cpu()
    same = bool((allr[:, :2].view(torch.int64) == allr[0, :2].view(torch.int64)).all())
    return {"rccl_ranks": dist.get_world_size(), "backend": str(dist.get_backend()), "grad_checksum_equal_across_ranks": same,
            "grad_checksum_fp64": [float(allr[0, 0]), float(allr[0, 1])], "finite": bool(torch.isfinite(allr).all()),
            "rank_losses": [round(float(v), 5) for v in allr[:, 2]],
            "note": "after the warm-up steps: fp64 sum and abs-sum of the all-reduced gradient arena, all-gathered and compared bit for bit; "
                    "rank_losses differ because every rank runs its own shard (per-rank BN statistics, SURVEY 8e)"}


def measure_allreduce(plan, reducer, world, reps=15):
    """Measured time of each gradient bucket's all-reduce, in isolation, after the timed region: HIP events on the launch stream
    around a blocking (async_op=False) collective on a scratch copy of the arena — the launch stream waits for RCCL's stream, so
    the bracket spans the collective end to end.  MAX over ranks of each rank's median."""
    import statistics
    import torch.distributed as dist
    scratch = plan.gflat.clone()
    r = reducer.for_plan(plan)
    per = []
    for b, e, _ in r.buckets:
        view, ms = scratch[b:e], []
        for _ in range(reps):
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(view, op=dist.ReduceOp.AVG)
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        per.append(statistics.median(ms[3:]))
    t = torch.tensor(per, device=scratch.device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return {"per_bucket_ms": [round(float(v), 4) for v in t.cpu()], "bucket_mb": [round((e - b) * 4 / 1e6, 2) for b, e, _ in r.buckets],
            "sum_ms": round(float(t.sum()), 4), "reps": reps - 3,
            "note": "isolated (nothing overlapping), blocking collectives on a copy of the arena; inside a step all but the last bucket ride under backward"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.dp_overhead_child:
        return dp_overhead_child(a)
    if world != a.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, a.gpus))
    if a.launch_check:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "rank_sum": float(t.item()),
                              "parent_clean": os.environ.get("MNY_LAUNCHER_PARENT_CLEAN") == "1", "rank_cuda_initialized": bool(torch.cuda.is_initialized())}))
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    force_dp = os.environ.get("MNY_FORCE_DP") == "1"      # run the RCCL path even with one rank (validation aid)
    use_dp = world > 1 or force_dp
    if use_dp:
        import torch.distributed as dist
        if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION"):
            os.environ["NCCL_DEBUG"] = "WARN"            # keep RCCL's version banner off stdout: rank 0 prints ONE JSON line
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")   # ... and whatever RCCL still warns about goes to stderr
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=device)

    from mobilenet_yolo_pytorch_amd import synthetic, yolo
    torch.manual_seed(0)                      # identical init on every rank
    adt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    if a.arch == "mbv3":
        from mobilenet_yolo_pytorch_amd import mbv3
        model = mbv3.yolo(synthetic.VOC_CONFIG, act_dtype=adt).to(device).train()
    else:
        model = yolo(synthetic.VOC_CONFIG, act_dtype=adt).to(device).train()
    reducer = None
    if use_dp:
        from mobilenet_yolo_pytorch_amd.dp import attach_data_parallel
        reducer = attach_data_parallel(model)
    x, tg = make_batch(a.batch, rank, device, a.size)

    def step():
        for p in model.parameters():
            p.grad = None                     # optimizer.zero_grad(set_to_none=True), train.py:254
        out = model(x, tg)                    # train.py:260
        (out[0][0] + out[1][0]).backward()    # train.py:276,282
        return out

    for _ in range(max(a.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    plan = model._plans[(a.batch, a.size, a.size, True) + (("bf16",) if a.dtype == "bf16" else ())]
    dp_info = dp_evidence_after_warmup(plan, reducer, out, world, rank, device) if use_dp else None

    # timed region: K steps, barrier + sync on both sides; the dominant kernels are bracketed by HIP events
    inline = a.roofline_pass == "inline" or a.breakdown
    # the brackets cost the GPU its back-to-back dispatch (~0.3 ms per step at 120 event records) and force the step onto one stream, so
    # they sit around the dominant kernels of every `bracket_every`-th timed step only (steps 0, B, 2B, ...: still hundreds of launches,
    # still inside the timed region); the other steps run exactly as a training loop runs them.  --breakdown brackets everything.
    bracket_every = 1 if a.breakdown else max(1, a.bracket_every)
    n_bracketed = (a.steps + bracket_every - 1) // bracket_every
    tstate = None
    if inline:
        plan.enable_timing(only=None if a.breakdown else MFMA_KERNELS, steps=n_bracketed)
        tstate = plan.timing
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(a.steps):
        if inline:
            plan.timing = tstate if it % bracket_every == 0 else None
        out = step()
    if inline:
        plan.timing = tstate
    if reducer is not None:
        reducer.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if not inline:                               # roofline leg: the same K steps again, MFMA kernels bracketed by HIP events
        plan.enable_timing(only=MFMA_KERNELS, steps=a.steps)
        for _ in range(a.steps):
            out = step()
        if reducer is not None:
            reducer.wait()
        torch.cuda.synchronize()
    timing = plan.disable_timing()
    loss = float(out[0][0].detach()) + float(out[1][0].detach())
    # read the brackets NOW: the second pass below draws its events from the same pre-created pool (HipEvent.reserve rewinds it), so the
    # first pass's event objects are re-recorded there.  (Rounds 1-2 read them after the second pass: the first 2 x 850 events of the
    # dominant entry point's 2 x 1 240 then held second-pass kernels — a mixture that happened to average like mny_pw_fwd; found in
    # round 3 when a 0.8 ms entry point joined the second pass.  --breakdown runs, which have no second pass, were never affected.)
    agg = roofline_from({"fwd": timing["fwd"], "bwd": timing["bwd"]}, {"fwd": plan.fwd.calls, "bwd": plan.bwd.calls}) if rank == 0 else None
    per = {}
    if rank == 0 and a.breakdown and a.detail:
        want = None if a.detail == "all" else set(a.detail.split(","))
        for which in ("fwd", "bwd"):
            calls = plan.fwd.calls if which == "fwd" else plan.bwd.calls
            for idx, name, e0, e1 in timing[which]:
                if want is None or name in want:
                    k = (which, idx)
                    per.setdefault(k, [name, calls[idx][3] or {}, 0.0])[2] += e0.elapsed_time(e1)
    second = None
    if world == 1 and not a.breakdown:                # untimed second pass (single GPU only): the other priced entry points, bracketed
        k2 = max(3, min(a.steps, 10))
        plan.enable_timing(only=set(SECOND_PASS), steps=k2)
        for _ in range(k2):
            step()
        if reducer is not None:
            reducer.wait()
        torch.cuda.synchronize()
        t2 = plan.disable_timing()
        second = (roofline_from({"fwd": t2["fwd"], "bwd": t2["bwd"]}, {"fwd": plan.fwd.calls, "bwd": plan.bwd.calls}), k2)     # read at once (see above)

    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if use_dp:
        try:
            dp_info["allreduce_measured"] = measure_allreduce(plan, reducer, world)
        except Exception as e:                                                  # noqa: BLE001 — evidence, never the headline
            dp_info["allreduce_measured"] = {"error": repr(e)[:300]}
        dp_info["allreduce_model"] = allreduce_model(plan.gflat.numel() * 4, len(reducer.for_plan(plan).buckets))

    if rank == 0:
        if a.breakdown:
            tot = sum(d["ms"] for d in agg.values()) / a.steps
            for name, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
                ms = d["ms"] / a.steps
                print("%-22s %8.3f ms/step  %5.1f%%  n=%3d  %7.1f TF/s  %7.1f GB/s" % (
                    name, ms, 100 * ms / tot, d["n"] // a.steps, d["flops"] / d["ms"] / 1e9 if d["ms"] else 0,
                    d["bytes"] / d["ms"] / 1e6 if d["ms"] else 0), file=sys.stderr)
            print("sum of bracketed kernels %.3f ms/step; wall %.3f ms/step" % (tot, dt / a.steps * 1e3), file=sys.stderr)
            for (which, idx), (name, meta, ms) in sorted(per.items(), key=lambda kv: -kv[1][2]):
                ms /= a.steps
                print("%s %-18s %-28s %8.3f ms  %7.1f TF/s %7.1f GB/s  #%d" % (which, name, meta.get("shape", ""), ms,
                      meta.get("flops", 0) / ms / 1e9 if ms else 0, meta.get("bytes", 0) / ms / 1e6 if ms else 0, idx), file=sys.stderr)
            if os.environ.get("MNY_PRINT_MARKS") and getattr(plan, "bwd", None) is not None:
                for nm, ci in sorted(plan.bwd.marks.items(), key=lambda kv: kv[1]):
                    print("mark bwd %-40s ends before call #%d" % (nm, ci), file=sys.stderr)
        dom = max((n for n in agg if n in MFMA_KERNELS), key=lambda n: agg[n]["ms"])
        headline = (a.arch, a.size, a.batch, a.dtype, world) == ("mbv2", SIZE, BATCH, "f32", 1)
        bf16 = a.dtype == "bf16"
        roof = roofline_object(dom, agg[dom], n_bracketed if inline else a.steps, "mfma", bf16,
                               ("HIP events inside the timed region, around every launch of the entry point in %d of the %d timed steps (every %s)" % (
                                   n_bracketed, a.steps, "step" if bracket_every == 1 else "%dth step" % bracket_every)) if inline else
                               "HIP events over %d further steps run right after the timed (hipGraph-replayed) steps" % a.steps)
        roof["traffic"], src = committed_traffic(plan, dom, headline)
        if src:
            roof["traffic_source"] = src
        if dom in ("mny_pw_fwd", "mny_pw_fwd_bf16") and os.environ.get("MNY_NO_THIN") is None:
            # the entry point routes its short-reduction launches (K = 16 / 24 / 32 channels, HBM-bound) to a vector-ALU stream kernel:
            # their FLOPs and time are inside this object, priced against the matrix-core peak like the rest
            roof["kernels"] = ("pw_gemm_nt_dma_kernel (MFMA tiles) + pw_wide_kernel (barrier-free MFMA kernel, K 52..96 into N >= K; csrc/pwwide.hip) + "
                               "pw_thin_kernel (vector ALU, the K <= 32 launches; csrc/pwthin.hip)")
        if not bf16 and os.environ.get("MNY_X6") != "0":
            # fp32 GEMMs with FLOP/byte >= 20 take the six-product bf16 form (csrc/pwgemm.hip, x6_split): say so, and price the same
            # achieved rate against the pipe that actually carries it as well
            roof["matrix_form"] = ("shapes with >= 20 FLOP per byte run as six bf16 partial products per fp32 product on v_mfma_f32_32x32x16_bf16 "
                                   "(every operand cut EXACTLY into three bf16 pieces, fp32 accumulate; error against an fp64 product below the "
                                   "fp32 MFMA's: tools/x6_precision.py); `peak` stays the fp32-MFMA peak of the dtype, `frac_bf16x6` prices the "
                                   "same achieved rate against the bf16 pipe at six products per FMA (%.0f / 6 TFLOP/s)" % PEAK_BF16_MFMA_TFLOPS)
            roof["frac_bf16x6"] = round(roof["achieved"] / (PEAK_BF16_MFMA_TFLOPS / 6.0), 4)
        if bf16:                                       # bf16 storage: the GEMMs are HBM-bound — say so on the line
            roof["note"] = "priced against the dense bf16 MFMA peak; this configuration is HBM-bound: see algorithmic_hbm_gbs / %d GB/s" % int(PEAK_HBM_GBS)
        others = []
        if second is not None:
            agg2, k2 = second
            for name in sorted(agg2, key=lambda n: -agg2[n]["ms"]):
                o = roofline_object(name, agg2[name], k2, SECOND_PASS[name], bf16,
                                    "HIP events over %d further steps run right after the timed region (untimed pass)" % k2)
                o["traffic"], src = committed_traffic(plan, name, headline)
                if src:
                    o["traffic_source"] = src
                others.append(o)
        res = {
            "metric": "images/sec MobileNetV2-YOLO 352x352 fwd+bwd @ bs256" if (a.arch, a.size, a.batch, a.dtype) == ("mbv2", SIZE, BATCH, "f32")
            else "images/sec %s-YOLO %dx%d fwd+bwd @ bs%d (NOT the headline config)" % (a.arch, a.size, a.size, a.batch), "value": round(world * a.batch * a.steps / dt, 2),
            "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "%s-YOLO %dx%d bs=%d/GPU fwd+loss+bwd %s (BASELINE configs[%s])" % (
                "MobileNetV2" if a.arch == "mbv2" else "MobileNetV3", a.size, a.size, a.batch,
                "fp32" if a.dtype == "f32" else "bf16 activation storage, fp32 weights/accumulate",
                ("1" if world == 1 else "2") if a.arch == "mbv2" else ("3" if a.dtype == "bf16" else "3, fp32 instead of bf16")),
                       "global_batch": world * a.batch, "parallelism": "dp%d" % world,
                       "timed_region": "zero_grad + forward(net + 2 on-device YOLO losses) + backward; random-init weights",
                       "loss": round(loss, 5)},
            "roofline": roof,
        }
        if dp_info is not None:
            res["data_parallel"] = dp_info
        res["plan_signatures"] = {e: list(plan_signature(plan, e)) for e in [dom] + sorted(SECOND_PASS) if plan_signature(plan, e)[0]}
        res["plan_signatures"]["__step__"] = list(step_signature(plan))
        alg = {}
        for calls in (plan.fwd.calls, plan.bwd.calls):        # algorithmic bytes per step and entry point (what tools/prof_*summary.py calibrate the counters on)
            for _fn, _args, name, meta in calls:
                if meta and meta.get("bytes"):
                    alg[name] = alg.get(name, 0) + int(meta["bytes"])
        res["algorithmic_bytes_per_step"] = alg
        hb = [o for o in others if o["kernel"].startswith("mny_dw_fwd")]
        if hb:
            res["roofline_hbm"] = hb[0]               # the bandwidth-bound depthwise forward (north-star target: frac >= 0.6)
        if others:
            res["roofline_more"] = [o for o in others if not o["kernel"].startswith("mny_dw_fwd")]   # wgrad, dgrad+reduce (MFMA) and the fused expand-unit backward (HBM)
        if world == 1 and not a.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:                                              # noqa: BLE001 — reported, never at the cost of the line
                res["cpu_baseline"] = {"error": repr(e)[:300]}
        if world == 1 and not a.no_nms and headline and not use_dp:
            try:
                res["dp_overhead"] = dp_overhead_leg(a)
                res["dp_overhead"]["timed_region_ms_per_step"] = round(dt / a.steps * 1e3, 3)
            except Exception as e:                                              # noqa: BLE001 — a side leg must never cost the headline line
                res["dp_overhead"] = {"error": repr(e)[:300]}
        # The headline object is complete here.  Every side leg below is evidence, never the headline: each runs under try/except (its
        # object becomes {"error": ...}), config3 additionally under a wall budget, and the ONE json line is printed whatever they do.
        def side(name, fn, *fargs):
            t_leg = time.perf_counter()
            try:
                res[name] = fn(*fargs)
            except Exception as e:                                              # noqa: BLE001 — incl. the budget alarm's TimeoutError; Ctrl-C / SystemExit propagate (the line still prints in `finally`)
                res[name] = {"error": repr(e)[:300]}
                try:
                    torch.cuda.synchronize()
                except Exception:                                               # noqa: BLE001
                    pass
                import gc
                gc.collect()                                                    # a leg cut mid-way leaves a half-built plan behind: free it before the next leg
                torch.cuda.empty_cache()
            print("bench: side leg %s %.1f s" % (name, time.perf_counter() - t_leg), file=sys.stderr)

        def with_budget(seconds, fn, *fargs):
            """fn(*fargs) under a SIGALRM wall budget (main thread only; a leg stuck inside one HIP call is not interruptible — the
            driver's own timeout covers that — but a leg that merely runs long is cut here and the headline still prints)."""
            import signal

            def on_alarm(_sig, _frm):
                raise TimeoutError("side leg exceeded its %d s wall budget" % seconds)
            old = signal.signal(signal.SIGALRM, on_alarm)
            signal.alarm(seconds)
            try:
                return fn(*fargs)
            finally:
                signal.alarm(0)
                signal.signal(signal.SIGALRM, old)
        try:
            if world == 1 and not a.no_nms and headline:
                del out
                side("config3", with_budget, 240, config3_leg, device)
                if not a.no_cpu_baseline and isinstance(res.get("config3"), dict) and "error" not in res["config3"]:
                    try:
                        res["config3"]["cpu_baseline"] = with_budget(120, cpu_baseline_c3)
                    except Exception as e:                                      # noqa: BLE001
                        res["config3"]["cpu_baseline"] = {"error": repr(e)[:200]}
                side("config0", with_budget, 120, c1_leg, device)
            if world == 1 and not a.no_nms:
                side("nms", nms_bench, device)
                side("map", map_bench, device)
                side("prep", prep_bench, device)
                side("optimizer", optimizer_bench, model)
                side("pcie_inclusive", h2d_bench, step, x)
        finally:
            if a.full_json:
                print(json.dumps(res), flush=True)
            else:                                # (nothing large on stderr either: the driver's 8 kB tail is stdout FOLLOWED by stderr)
                print(json.dumps(compact_line(res), separators=(",", ":")), flush=True)
    if use_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
