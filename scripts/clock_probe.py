"""Sustained clock / power while one kernel family runs in a loop (GPU box): python scripts/clock_probe.py gemm|bn|idle"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
case = sys.argv[1] if len(sys.argv) > 1 else "gemm"
dt = torch.bfloat16
if case == "gemm":
    M, N, Kd = 8192, 4096, 4096
    x = torch.randn(M, Kd, device="cuda").to(dt); w = (torch.randn(N, Kd, device="cuda") * 0.05).to(dt); y = torch.empty(M, N, device="cuda", dtype=dt)
    f = lambda: K.gemm(1, K.A_KC, K.B_KC, x, w, y, M, N, Kd, Kd, Kd, N)
    flops = 2.0 * M * N * Kd
elif case == "bn":
    z = torch.randn(346112, 256, device="cuda").to(dt); y = torch.empty_like(z); ss = torch.ones(256, 2, device="cuda")
    f = lambda: K.bn_apply(z, ss, None, True, y)
    flops = 0
else:
    f = lambda: time.sleep(0.001)
    flops = 0
stop = False
def sampler():
    time.sleep(1.0)
    for _ in range(3):
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
        keep = [l.strip() for l in r.stdout.splitlines() if any(k in l for k in ("sclk", "mclk", "Power", "fclk"))]
        print(case, " | ".join(keep)[:400], flush=True)
        time.sleep(0.7)
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < 4.0:
    for _ in range(20): f()
    if case != "idle": torch.cuda.synchronize()
    n += 20
el = time.time() - t0
th.join()
if flops: print(f"{case}: {flops * n / el / 1e12:.0f} TFLOP/s sustained over {el:.1f} s")
