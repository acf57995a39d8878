"""The stem's first convolution as a GEMM on its im2col rows (1.38 M x 32 x 32, BatchNorm statistics in the epilogue): the streamed kernel
(csrc/gemm_skinny.hip) against the tiled LDS-DMA kernel (CROG_SKINNY=0).  GPU box; run once per setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
dt = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
M = 32 * 208 * 208
nset = 4
As = [torch.randn(M, 32, device="cuda").to(dt) for _ in range(nset)]
ys = [torch.empty(M, 32, device="cuda", dtype=dt) for _ in range(nset)]
w = (torch.randn(32, 32, device="cuda") * 0.2).to(dt)
stats = torch.zeros(64, 32, 2, device="cuda")
it = [0]
def run(st):
    i = it[0] = (it[0] + 1) % nset
    K.gemm(1, K.A_KC, K.B_KC, As[i], w, ys[i], M, 32, 32, 32, 32, 32, col_stats=stats if st else None, stat_replicas=64 if st else 0)
print(f"M={M}: with statistics {timeit(lambda: run(True)):6.1f} us, without {timeit(lambda: run(False)):6.1f} us  (bytes at 5 TB/s {M*128/5e6:5.1f} us)")
