"""Timing-only ablation of the LDS-DMA GEMM on one conv shape: full / no loads / no MFMA (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
dt = torch.bfloat16
for name, B, HW, Cin, Cout in [("proj 512->512 @52", 32, 52, 512, 512), ("proj 512->256 @104", 32, 104, 512, 256)]:
    M = B * HW * HW
    x = torch.randn(M, Cin, device="cuda").to(dt); w = (torch.randn(Cout, 9 * Cin, device="cuda") * 0.05).to(dt)
    y = torch.empty(M, Cout, device="cuda", dtype=dt)
    fl = 2.0 * M * Cout * 9 * Cin
    for dbg, tag in [(0, "full"), (1, "no-loads"), (2, "no-mfma"), (3, "neither")]:
        K.DEBUG_FLAGS = dbg
        t = timeit(lambda: K.gemm(1, K.A_IM2COL, K.B_KC, x, w, y, M, Cout, 9 * Cin, Cin, 9 * Cin, Cout, conv=(HW, HW, Cin)))
        print(f"{name:22s} {tag:9s} {t*1e3:8.1f} us  ({fl/t/1e9:7.1f} TF/s equivalent)", flush=True)
    n = 4096
    a = torch.randn(n, n, device="cuda").to(dt); b = torch.randn(n, n, device="cuda").to(dt); c = torch.empty(n, n, device="cuda", dtype=dt)
    for dbg, tag in [(0, "full"), (1, "no-loads"), (2, "no-mfma")]:
        K.DEBUG_FLAGS = dbg
        t = timeit(lambda: K.gemm(1, K.A_KC, K.B_KC, a, b, c, n, n, n, n, n, n))
        print(f"{'square 4096':22s} {tag:9s} {t*1e3:8.1f} us  ({2*n**3/t/1e9:7.1f} TF/s equivalent)", flush=True)
K.DEBUG_FLAGS = 0
