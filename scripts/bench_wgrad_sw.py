"""Sliding-window 3x3 weight gradient (csrc/wgrad_sw.hip) against the implicit-GEMM form (debug bit 22) on the tail launches of the step
(layer1 / stem), operand sets rotated past the Infinity Cache (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
dt = torch.bfloat16
def timeit(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for B, HW, Cin, Cout in [(32, 104, 64, 64), (32, 208, 32, 32), (32, 208, 32, 64), (32, 208, 64, 32)]:
    M = B * HW * HW
    N = 9 * Cin
    nset = max(2, int(700e6 / (M * (Cin + Cout) * 2)) + 1)
    xs = [torch.randn(M, Cin, device="cuda").to(dt) for _ in range(nset)]; dys = [torch.randn(M, Cout, device="cuda").to(dt) for _ in range(nset)]
    dw = torch.zeros(Cout, N, device="cuda")
    sk = K.pick_splitk(Cout, N, M, 32, conv=True)
    it = [0]
    def run():
        i = it[0] = (it[0] + 1) % nset
        K.gemm(1, K.A_MC, K.B_NC_IM2COL, dys[i], xs[i], dw, Cout, N, M, Cout, Cin, N, splitk=sk, out_mode=K.OUT_F32_ATOMIC, conv=(HW, HW, Cin))
    K.DEBUG_FLAGS = 4194304; t0 = timeit(run, 3 * nset)
    K.DEBUG_FLAGS = 0; t1 = timeit(run, 3 * nset)
    K.DEBUG_FLAGS = 32; t2 = timeit(run, 3 * nset); K.DEBUG_FLAGS = 0
    ws = torch.empty(256, Cout, N, device="cuda")
    def run_slab():
        i = it[0] = (it[0] + 1) % nset
        K.gemm(1, K.A_MC, K.B_NC_IM2COL, dys[i], xs[i], ws, Cout, N, M, Cout, Cin, N, splitk=256, out_mode=K.OUT_F32, conv=(HW, HW, Cin))
        K.splitk_reduce(ws, 256, Cout, N, N, dw, 0, N, accumulate=True)
    t3 = timeit(run_slab, 3 * nset)
    print(f"dW[{Cout} x {N}] over {M} pixels: implicit GEMM (split {sk}) {t0:6.1f} us, sliding window {t1:6.1f} us (without its atomic adds {t2:6.1f}; 256 slabs + ordered reduction {t3:6.1f}); operands at 5 TB/s {M*(Cin+Cout)*2/5e6:5.1f} us", flush=True)
