"""Cost of the BatchNorm-statistics epilogue (atomic replicas) on mid-size 1x1 / 3x3 forward launches, by replica count.  HBM-cold."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
from crog_amd.functional import stat_replicas
from bench_gemm import timeit
dt = torch.bfloat16
for M, N, Kd, conv in [(21632, 1024, 256, 0), (21632, 256, 1024, 0), (86528, 512, 128, 0), (86528, 128, 512, 0), (5408, 2048, 512, 0), (5408, 512, 2048, 0), (21632, 256, 2304, 26), (86528, 128, 1152, 52), (5408, 512, 4608, 13), (346112, 256, 64, 0), (346112, 64, 256, 0)]:
    C = Kd // 9 if conv else Kd
    nset = max(1, int(600e6 / (M * (N + C) * 2)) + 1)
    xs = [torch.randn(M, C, device="cuda").to(dt) for _ in range(nset)]; ys = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(nset)]
    w = (torch.randn(N, Kd, device="cuda") * 0.05).to(dt)
    it = [0]
    def run(R):
        i = it[0] = (it[0] + 1) % nset
        kw = {} if R == 0 else dict(col_stats=stats[R], stat_replicas=R)
        if conv: K.gemm(1, K.A_IM2COL, K.B_KC, xs[i], w, ys[i], M, N, Kd, C, Kd, N, conv=(conv, conv, C), **kw)
        else: K.gemm(1, K.A_KC, K.B_KC, xs[i], w, ys[i], M, N, Kd, Kd, Kd, N, **kw)
    stats = {R: torch.zeros(R, N, 2, device="cuda") for R in (1, 2, 4, 8, 16)}
    n = max(10, 3 * nset)
    res = [f"none {timeit(lambda: run(0), n)*1e3:6.1f}"] + [f"R={R} {timeit(lambda: run(R), n)*1e3:6.1f}" for R in (1, 2, 4, 8, 16)]
    print(f"M={M:6d} N={N:4d} K={Kd:4d} rule R={stat_replicas(K.stat_tiles(M), N)}: " + "  ".join(res) + " us", flush=True)
