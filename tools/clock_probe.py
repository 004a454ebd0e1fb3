"""Sample the GPU clock (rocm-smi) while the fat 512x512 pointwise GEMM runs back to back: is ~105 TF/s a kernel limit or a clock limit?"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from mobilenet_yolo_pytorch_amd import ops
    M, K, N = 123904, 512, 512
    x = torch.randn(1, 352, 352, K, device="cuda")[:, :, :, :].contiguous().view(1, 1, M, K) if False else torch.randn(1, 1, M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / 22
    out = torch.empty(1, 1, M, N, device="cuda")
    for _ in range(5):
        ops.pw_fwd((x, None, None, 0), w, want_stats=False, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 6.0:
        for _ in range(50):
            ops.pw_fwd((x, None, None, 0), w, want_stats=False, out=out)
        torch.cuda.synchronize(); n += 50
    dt = time.perf_counter() - t0
    print("gemm %.1f TF/s over %.1f s" % (2.0 * M * K * N * n / dt / 1e12, dt))
    sys.exit(0)
child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "child"])
time.sleep(3.0)
for i in range(8):
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if "sclk" in l or "Power" in l or "mclk" in l]
    print(" | ".join(l.strip()[:90] for l in lines[:4]))
    time.sleep(0.5)
child.wait()
r = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True)
print("idle:", " | ".join(l.strip()[:90] for l in r.stdout.splitlines() if "sclk" in l))
