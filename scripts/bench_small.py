"""Micro-benchmark of the small (text tower / decoder) GEMM shapes, standalone. GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
from bench_gemm import timeit

dt = torch.bfloat16; dc = K.dcode(dt)
def lin(M, N, Kd):
    x = torch.randn(M, Kd, device="cuda").to(dt); w = (torch.randn(N, Kd, device="cuda") * 0.05).to(dt)
    y = torch.empty(M, N, device="cuda", dtype=dt); dy = torch.randn(M, N, device="cuda").to(dt)
    dx = torch.empty(M, Kd, device="cuda", dtype=dt); dw = torch.zeros(N, Kd, device="cuda")
    f = lambda: K.gemm(dc, K.A_KC, K.B_KC, x, w, y, M, N, Kd, Kd, Kd, N)
    d = lambda: K.gemm(dc, K.A_KC, K.B_NC, dy, w, dx, M, Kd, N, N, Kd, Kd)
    sk = K.pick_splitk(N, Kd, M, 32)
    g = lambda: K.gemm(dc, K.A_MC, K.B_NC, dy, x, dw, N, Kd, M, N, Kd, Kd, splitk=sk, out_mode=K.OUT_F32_ATOMIC)
    fl = 2.0 * M * N * Kd
    tf, td, tg = timeit(f, 50), timeit(d, 50), timeit(g, 50)
    print(f"linear M={M:6d} {Kd:5d}->{N:5d}: fwd {tf*1e3:7.1f} us ({fl/tf/1e9:6.1f} TF/s)  dgrad {td*1e3:7.1f} us ({fl/td/1e9:6.1f})  wgrad(sk={sk}) {tg*1e3:7.1f} us ({fl/tg/1e9:6.1f})")
for M in (640, 5408, 21632):
    for N, Kd in ((1536, 512), (512, 512), (2048, 512), (512, 2048)):
        lin(M, N, Kd)
def bmm(name, bt, M, N, Kd, al, bl):
    a = torch.randn(bt * 704 * 704, device="cuda").to(dt)   # big enough for either orientation
    b = torch.randn(bt * 704 * 704, device="cuda").to(dt)
    lda = Kd if al == K.A_KC else M
    ldb = Kd if bl == K.B_KC else N
    lda = (lda + 7) // 8 * 8; ldb = (ldb + 7) // 8 * 8; ldc = (N + 7) // 8 * 8
    c = torch.empty(bt * M * ldc, device="cuda", dtype=dt)
    ra = (M if al == K.A_KC else Kd); rb = (N if bl == K.B_KC else Kd)
    f = lambda: K.gemm(dc, al, bl, a, b, c, M, N, Kd, lda, ldb, ldc, batch=bt, sA=(ra * lda, 0), sB=(rb * ldb, 0), sC=(M * ldc, 0))
    t = timeit(f, 50); fl = 2.0 * bt * M * N * Kd
    print(f"bmm {name:22s} bt={bt} M={M} N={N} K={Kd}: {t*1e3:7.1f} us ({fl/t/1e9:6.1f} TF/s)")
try:
    bmm("dec self QK^T", 256, 676, 676, 64, K.A_KC, K.B_KC)
    bmm("dec self PV", 256, 676, 64, 676, K.A_KC, K.B_NC)
    bmm("dec self dV=P^T dO", 256, 676, 64, 676, K.A_MC, K.B_NC)
    bmm("text QK^T", 256, 20, 20, 64, K.A_KC, K.B_KC)
    bmm("text PV", 256, 20, 64, 20, K.A_KC, K.B_NC)
    bmm("cross QK^T", 256, 676, 20, 64, K.A_KC, K.B_KC)
except Exception as e:
    print("bmm failed:", e)
