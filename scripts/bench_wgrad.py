"""A/B of split-K choices for weight-gradient GEMMs (GPU box). Env: CROG_GEMM_NO_XCD_SPLITK."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
from bench_gemm import timeit
dt = torch.bfloat16; dc = 1
tag = os.environ.get("TAG", "")
def conv(name, B, HW, Cin, Cout, sks):
    M = B * HW * HW
    x = torch.randn(M, Cin, device="cuda").to(dt); dy = torch.randn(M, Cout, device="cuda").to(dt)
    dw = torch.zeros(Cout, 9 * Cin, device="cuda"); fl = 2.0 * M * Cout * 9 * Cin
    out = []
    for sk in sks:
        tg = timeit(lambda: K.gemm(dc, K.A_MC, K.B_NC_IM2COL, dy, x, dw, Cout, 9 * Cin, M, Cout, Cin, 9 * Cin, conv=(HW, HW, Cin), splitk=sk, out_mode=K.OUT_F32_ATOMIC))
        out.append(f"sk={sk}:{fl/tg/1e9:6.1f}")
    print(f"{tag:6s} c3 {name:22s} " + "  ".join(out), flush=True)
def lin(name, M, Kd, N, sks):
    x = torch.randn(M, Kd, device="cuda").to(dt); dy = torch.randn(M, N, device="cuda").to(dt)
    dw = torch.zeros(N, Kd, device="cuda"); fl = 2.0 * M * N * Kd
    out = []
    for sk in sks:
        tg = timeit(lambda: K.gemm(dc, K.A_MC, K.B_NC, dy, x, dw, N, Kd, M, N, Kd, Kd, splitk=sk, out_mode=K.OUT_F32_ATOMIC))
        out.append(f"sk={sk}:{fl/tg/1e9:6.1f}")
    print(f"{tag:6s} 1x1 {name:21s} " + "  ".join(out), flush=True)
conv("512->256 @104", 32, 104, 512, 256, (8, 10, 16))
conv("512->512 @52", 32, 52, 512, 512, (5, 8, 16))
conv("256->256 @52", 32, 52, 256, 256, (16, 21, 24, 32))
conv("512->512 @26", 32, 26, 512, 512, (5, 8))
conv("64->64 @104", 32, 104, 64, 64, (128, 153, 160))
lin("512->512 M=21632", 21632, 512, 512, (16, 24, 28, 32, 48))
lin("256->1024 M=21632", 21632, 256, 1024, (16, 24, 28, 32, 48))
lin("512->2048 M=21632", 21632, 512, 2048, (8, 12, 16))
lin("256->1280 M=346112", 346112, 256, 1280, (16, 32, 38, 40))
lin("64->256 M=346112", 346112, 64, 256, (128, 256, 384))
