"""What a grouped weight-gradient launch would cost: n equal problems [M x N x K] emulated as ONE ping-pong launch over [M x n N x K]
(same tiles, same bytes per tile) against the n separate launches of today's dispatch (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
dt = torch.bfloat16
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for M, N, Kd, n in [(512, 512, 21632, 12), (1024, 256, 21632, 6), (512, 2048, 21632, 3), (256, 1024, 21632, 5), (1024, 512, 21632, 4)]:
    dy = [(torch.randn(Kd, M, device="cuda") * 0.1).to(dt) for _ in range(n)]
    x = [torch.randn(Kd, N, device="cuda").to(dt) for _ in range(n)]
    xb = torch.randn(Kd, n * N, device="cuda").to(dt)
    g = torch.zeros(M, N, device="cuda"); gb = torch.zeros(M, n * N, device="cuda")
    sk0 = K.lib().crog_gemm_splitk_hint(K.BF16, K.A_MC, K.B_NC, M, N, Kd)
    def sep():
        for i in range(n):
            K.gemm(1, K.A_MC, K.B_NC, dy[i], x[i], g, M, N, Kd, M, N, N, splitk=sk0, out_mode=K.OUT_F32_ATOMIC)
    t_sep = timeit(sep)
    line = f"{n:2d} x dW[{M} x {N}] K={Kd}: separate (sk={sk0}) {t_sep:7.1f} us"
    tiles = -(-M // 256) * -(-n * N // 256)
    for target in (144, 216, 288):
        sk = max(1, target // tiles)
        def grp():
            K.DEBUG_FLAGS = 32768
            K.gemm(1, K.A_MC, K.B_NC, dy[0], xb, gb, M, n * N, Kd, M, n * N, n * N, splitk=sk, out_mode=K.OUT_F32_ATOMIC)
            K.DEBUG_FLAGS = 0
        line += f" | one launch {tiles * sk} blocks (sk={sk}) {timeit(grp):7.1f} us"
    print(line, flush=True)
