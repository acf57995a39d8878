"""Large-M / small-C GEMMs of the stem and layer1 (prologue/epilogue-bound): time, TF/s and algorithmic GB/s per launch,
with and without BatchNorm statistics in the epilogue.  Env: CROG_GEMM_DMA_TILE=m|t|6 forces a tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
from bench_shapes import timeit
dt = torch.bfloat16; dc = 1
tag = os.environ.get("CROG_GEMM_DMA_TILE", "auto")
def report(name, t, fl, by):
    print(f"{tag:5s} {name:44s} {t*1e3:7.1f} us  {fl/t/1e9:6.1f} TF/s  {by/t/1e6:6.0f} GB/s", flush=True)
def conv3(B, HW, Cin, Cout):
    M = B * HW * HW
    x = torch.randn(M, Cin, device="cuda").to(dt); w = (torch.randn(Cout, 9 * Cin, device="cuda") * 0.05).to(dt)
    y = torch.empty(M, Cout, device="cuda", dtype=dt)
    fl = 2.0 * M * Cout * 9 * Cin; by = 2.0 * M * (Cin + Cout)
    for R in (0, 1, 4, 8):
        st = torch.zeros(max(R, 1) * Cout * 2, device="cuda") if R else None
        f = lambda: K.gemm(dc, K.A_IM2COL, K.B_KC, x, w, y, M, Cout, 9 * Cin, Cin, 9 * Cin, Cout, conv=(HW, HW, Cin), col_stats=st, stat_replicas=R)
        report(f"conv3 M={M} {Cin}->{Cout} stats R={R}", timeit(f, 20), fl, by)
def lin(M, Kd, N, res=False):
    x = torch.randn(M, Kd, device="cuda").to(dt); w = (torch.randn(N, Kd, device="cuda") * 0.05).to(dt)
    y = torch.empty(M, N, device="cuda", dtype=dt); dy = torch.randn(M, N, device="cuda").to(dt)
    dx = torch.empty(M, Kd, device="cuda", dtype=dt); r = torch.randn(M, Kd, device="cuda").to(dt)
    fl = 2.0 * M * N * Kd; by = 2.0 * M * (Kd + N)
    for R in (0, 4):
        st = torch.zeros(max(R, 1) * N * 2, device="cuda") if R else None
        f = lambda: K.gemm(dc, K.A_KC, K.B_KC, x, w, y, M, N, Kd, Kd, Kd, N, col_stats=st, stat_replicas=R)
        report(f"1x1 fwd M={M} {Kd}->{N} stats R={R}", timeit(f, 20), fl, by)
    d = lambda: K.gemm(dc, K.A_KC, K.B_NC, dy, w, dx, M, Kd, N, N, Kd, Kd)
    report(f"1x1 dgrad M={M} {N}->{Kd}", timeit(d, 20), fl, by)
    d2 = lambda: K.gemm(dc, K.A_KC, K.B_NC, dy, w, dx, M, Kd, N, N, Kd, Kd, R=r, ldr=Kd)
    report(f"1x1 dgrad+res M={M} {N}->{Kd}", timeit(d2, 20), fl, by + 2.0 * M * Kd)
conv3(32, 104, 64, 64)
conv3(32, 208, 32, 32)
conv3(32, 208, 32, 64)
lin(346112, 64, 256)
lin(346112, 256, 64)
lin(86528, 128, 512)
