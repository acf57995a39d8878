"""The weight gradients that run ALONE at the end of the step (layer1 / stem: tiny outputs, reductions over 346112 / 1.38 M pixels): time at
several split counts, with and without the fp32 atomics (debug bit 5), against the operands' HBM time (GPU box)."""
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
def case(B, HW, Cin, Cout, conv3):
    M = B * HW * HW
    Ncols = 9 * Cin if conv3 else Cin
    nset = max(2, int(700e6 / (M * (Cin + Cout) * 2)) + 1)
    xs = [torch.randn(M, Cin, device="cuda").to(dt) for _ in range(nset)]; dys = [torch.randn(M, Cout, device="cuda").to(dt) for _ in range(nset)]
    dw = torch.zeros(Cout, Ncols, device="cuda")
    it = [0]
    def run(sk):
        i = it[0] = (it[0] + 1) % nset
        if conv3: K.gemm(1, K.A_MC, K.B_NC_IM2COL, dys[i], xs[i], dw, Cout, Ncols, M, Cout, Cin, Ncols, splitk=sk, out_mode=K.OUT_F32_ATOMIC, conv=(HW, HW, Cin))
        else: K.gemm(1, K.A_MC, K.B_NC, dys[i], xs[i], dw, Cout, Ncols, M, Cout, Cin, Ncols, splitk=sk, out_mode=K.OUT_F32_ATOMIC)
    sk0 = K.pick_splitk(Cout, Ncols, M, 32, conv=conv3)
    out = []
    for sk in (max(1, sk0 // 4), max(1, sk0 // 2), sk0, sk0 * 2, sk0 * 4):
        r = []
        for flag in (0, 32):
            K.DEBUG_FLAGS = flag
            r.append(timeit(lambda: run(sk), 2 * nset))
        K.DEBUG_FLAGS = 0
        out.append(f"sk={sk:4d}: {r[0]:6.1f} ({r[1]:6.1f})")
    byt = M * (Cin + Cout) * 2
    print(f"dW[{Cout:3d} x {Ncols:4d}] K={M:7d} {'3x3' if conv3 else '1x1'} (HBM at 5 TB/s {byt/5e6:5.1f} us; hint sk={sk0}): " + "  ".join(out), flush=True)
for c in [(32, 104, 64, 64, True), (32, 104, 64, 64, False), (32, 104, 64, 256, False), (32, 104, 256, 64, False), (32, 104, 256, 128, False), (32, 104, 128, 128, True),
          (32, 208, 32, 32, True), (32, 208, 32, 64, True)]:
    case(*c)
