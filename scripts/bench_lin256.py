"""1x1 / linear forwards and data gradients with a plain epilogue: 128 x 128 vs the 256 x 256 tile (CROG_GEMM_LIN256=1).  HBM-cold."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
from bench_gemm import timeit
dt = torch.bfloat16
out = []
for kind, M, N, Kd in [("fwd", 21632, 1024, 256), ("fwd", 86528, 512, 128), ("fwd", 5408, 2048, 512), ("fwd", 21632, 512, 512), ("dgrad", 21632, 512, 2048), ("dgrad", 21632, 512, 512), ("dgrad", 21632, 256, 1024), ("dgrad", 21632, 2048, 512), ("dgrad", 86528, 512, 128), ("dgrad", 21632, 1536, 512)]:
    nset = max(1, int(600e6 / (M * (N + Kd) * 2)) + 1)
    xs = [torch.randn(M, Kd, device="cuda").to(dt) for _ in range(nset)]; ys = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(nset)]
    w = (torch.randn(N, Kd, device="cuda") * 0.05).to(dt); wn = (torch.randn(Kd, N, device="cuda") * 0.05).to(dt)
    it = [0]
    def run():
        i = it[0] = (it[0] + 1) % nset
        if kind == "fwd": K.gemm(1, K.A_KC, K.B_KC, xs[i], w, ys[i], M, N, Kd, Kd, Kd, N)
        else: K.gemm(1, K.A_KC, K.B_NC, xs[i], wn, ys[i], M, N, Kd, Kd, N, N)
    t = timeit(run, max(10, 3 * nset)) * 1e3
    out.append(f"{kind} {M}x{N}x{Kd}: {t:6.1f} us {2.0*M*N*Kd/t/1e6:5.0f} TF/s")
print(os.environ.get("TAG", ""), " | ".join(out))
