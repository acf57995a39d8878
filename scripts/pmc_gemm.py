"""One GEMM shape, launched N times, for PMC passes (rocprofv3 --pmc ... -- python3 scripts/pmc_gemm.py <case>)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
case = sys.argv[1] if len(sys.argv) > 1 else "conv"
dt = torch.bfloat16; dc = 1
if case == "conv":      # proj.vis.3: 3x3 512 -> 256 at 104^2, B = 32
    B, HW, Cin, Cout = 32, 104, 512, 256
    M = B * HW * HW
    x = torch.randn(M, Cin, device="cuda").to(dt); w = (torch.randn(Cout, 9 * Cin, device="cuda") * 0.05).to(dt)
    y = torch.empty(M, Cout, device="cuda", dtype=dt)
    f = lambda: K.gemm(dc, K.A_IM2COL, K.B_KC, x, w, y, M, Cout, 9 * Cin, Cin, 9 * Cin, Cout, conv=(HW, HW, Cin))
elif case == "wgrad":   # proj.vis.3's weight gradient: dW[256 x 4608] over 346112 pixels on the ping-pong transposed-fragment kernel (gemm_ppt.hip)
    B, HW, Cin, Cout = 32, 104, 512, 256
    Mpix, N = B * HW * HW, 9 * Cin
    x = torch.randn(Mpix, Cin, device="cuda").to(dt); dy = (torch.randn(Mpix, Cout, device="cuda") * 0.1).to(dt)
    g = torch.zeros(Cout, N, device="cuda")
    sk = K.lib().crog_gemm_splitk_hint(K.BF16, K.A_MC, K.B_NC_IM2COL, Cout, N, Mpix)
    f = lambda: K.gemm(dc, K.A_MC, K.B_NC_IM2COL, dy, x, g, Cout, N, Mpix, Cout, Cin, N, splitk=sk, out_mode=K.OUT_F32_ATOMIC, conv=(HW, HW, Cin))
else:                   # plain NT GEMM 8192 x 4096 x 4096
    M, N, Kd = 8192, 4096, 4096
    x = torch.randn(M, Kd, device="cuda").to(dt); w = (torch.randn(N, Kd, device="cuda") * 0.05).to(dt)
    y = torch.empty(M, N, device="cuda", dtype=dt)
    f = lambda: K.gemm(dc, K.A_KC, K.B_KC, x, w, y, M, N, Kd, Kd, Kd, N)
for _ in range(5): f()
torch.cuda.synchronize()
