"""3x3 forward-shaped implicit GEMMs by output width N: tile order / tile size A-B (GPU box).  K.DEBUG_FLAGS = 16: row tiles fastest."""
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
out = []
for B, HW, Cin, Cout in [(32, 104, 256, 512), (32, 104, 512, 256), (32, 52, 512, 512), (32, 52, 256, 512), (32, 52, 256, 256), (32, 26, 512, 512), (32, 26, 512, 1024), (32, 26, 1024, 512), (32, 26, 256, 256), (32, 13, 512, 512)]:
    M = B * HW * HW
    x = torch.randn(M, Cin, device="cuda").to(dt); w = (torch.randn(Cout, 9 * Cin, device="cuda") * 0.05).to(dt)
    y = torch.empty(M, Cout, device="cuda", dtype=dt)
    fl = 2.0 * M * Cout * 9 * Cin
    t = timeit(lambda: K.gemm(1, K.A_IM2COL, K.B_KC, x, w, y, M, Cout, 9 * Cin, Cin, 9 * Cin, Cout, conv=(HW, HW, Cin)))
    out.append(f"M={M:7d} N={Cout:5d} K={9*Cin:5d}: {t*1e3:7.1f} us {fl/t/1e9:6.1f} TF/s")
print(os.environ.get("TAG", ""), " | ".join(out[:5])); print(" " * len(os.environ.get("TAG", "")), " | ".join(out[5:]))
