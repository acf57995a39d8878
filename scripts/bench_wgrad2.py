"""Split-K x tile-shape sweep for the atomic-bound small weight gradients (GPU box). Env: CROG_GEMM_DMA_TILE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
from bench_gemm import timeit
dt = torch.bfloat16; dc = 1
tag = os.environ.get("CROG_GEMM_DMA_TILE", "auto")
def lin(M, Kd, N, sks):
    x = torch.randn(M, Kd, device="cuda").to(dt); dy = torch.randn(M, N, device="cuda").to(dt)
    dw = torch.zeros(N, Kd, device="cuda"); fl = 2.0 * M * N * Kd
    out = []
    for sk in sks:
        tg = timeit(lambda: K.gemm(dc, K.A_MC, K.B_NC, dy, x, dw, N, Kd, M, N, Kd, Kd, splitk=sk, out_mode=K.OUT_F32_ATOMIC), 30)
        out.append(f"sk={sk}:{tg*1e3:5.1f}us")
    print(f"{tag:5s} wgrad {Kd:4d}->{N:4d} M={M:6d}  " + "  ".join(out), flush=True)
lin(21632, 512, 512, (4, 8, 12, 16, 24, 28, 48))
lin(21632, 256, 1024, (4, 8, 12, 16, 24, 28, 48))
lin(21632, 512, 1024, (4, 8, 12, 24))
lin(21632, 512, 2048, (3, 6, 12))
lin(5408, 512, 2048, (2, 4, 7, 12))
lin(640, 512, 512, (1, 2, 4, 8))
lin(640, 512, 2048, (1, 2, 4))
lin(86528, 128, 512, (28, 56, 112, 224))
lin(346112, 64, 256, (96, 192, 384))
def conv(B, HW, Cin, Cout, sks):
    M = B * HW * HW
    x = torch.randn(M, Cin, device="cuda").to(dt); dy = torch.randn(M, Cout, device="cuda").to(dt)
    dw = torch.zeros(Cout, 9 * Cin, device="cuda"); fl = 2.0 * M * Cout * 9 * Cin
    out = []
    for sk in sks:
        tg = timeit(lambda: K.gemm(dc, K.A_MC, K.B_NC_IM2COL, dy, x, dw, Cout, 9 * Cin, M, Cout, Cin, 9 * Cin, conv=(HW, HW, Cin), splitk=sk, out_mode=K.OUT_F32_ATOMIC), 30)
        out.append(f"sk={sk}:{tg*1e3:5.1f}us")
    print(f"{tag:5s} c3wgrad {Cin:4d}->{Cout:4d} @{HW:3d}  " + "  ".join(out), flush=True)
conv(32, 104, 64, 64, (51, 102, 153, 256))
conv(32, 52, 128, 128, (14, 28, 56, 85))
conv(32, 208, 32, 32, (128, 256, 512))
conv(32, 208, 32, 64, (128, 256, 512))
conv(32, 52, 256, 256, (4, 7, 14, 21))
conv(32, 26, 512, 512, (1, 2, 5))
