"""Small-output weight gradients (the tail of the step: layer1 / stem, reductions over 346112 / 1.38 M pixels): split-K through fp32
atomic adds against split-K slabs + crog_splitk_reduce, at several split counts (GPU box).  Operand sets rotated past the Infinity Cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
dt = torch.bfloat16
def timeit(fn, n):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for B, HW, Cin, Cout, conv3 in [(32, 104, 64, 64, True), (32, 104, 128, 128, True), (32, 208, 32, 64, True), (32, 208, 32, 32, True), (32, 52, 128, 128, True),
                                (32, 104, 64, 256, False), (32, 104, 256, 64, False), (32, 52, 512, 128, False), (32, 26, 512, 512, False), (32, 26, 1024, 256, False)]:
    M = B * HW * HW
    N = 9 * Cin if conv3 else Cin
    bl = K.B_NC_IM2COL if conv3 else K.B_NC
    nset = max(2, int(700e6 / (M * (Cin + Cout) * 2)) + 1)
    xs = [torch.randn(M, Cin, device="cuda").to(dt) for _ in range(nset)]; dys = [(torch.randn(M, Cout, device="cuda") * 0.1).to(dt) for _ in range(nset)]
    g = torch.zeros(Cout, N, device="cuda")
    sk0 = K.pick_splitk(Cout, N, M, 32, conv=conv3)
    it = [0]
    conv = (HW, HW, Cin) if conv3 else (0, 0, 0)
    out = []
    for sk in sorted({max(1, sk0 // 4), max(1, sk0 // 2), sk0}):
        ws = torch.empty(sk, Cout, N, device="cuda")
        def atomic():
            i = it[0] = (it[0] + 1) % nset
            K.gemm(1, K.A_MC, bl, dys[i], xs[i], g, Cout, N, M, Cout, Cin, N, splitk=sk, out_mode=K.OUT_F32_ATOMIC, conv=conv)
        def slab():
            i = it[0] = (it[0] + 1) % nset
            K.gemm(1, K.A_MC, bl, dys[i], xs[i], ws, Cout, N, M, Cout, Cin, N, splitk=sk, out_mode=K.OUT_F32, conv=conv)
            K.splitk_reduce(ws, sk, Cout, N, N, g, 0, N, accumulate=True)
        ta, tb = timeit(atomic, 2 * nset), timeit(slab, 2 * nset)
        out.append(f"sk={sk:4d}: atomic {ta:6.1f} slab {tb:6.1f}")
    byt = M * (Cin + Cout) * 2
    print(f"dW[{Cout:3d} x {N:4d}] K={M:7d} {'3x3' if conv3 else '1x1'} (operands at 5 TB/s {byt/5e6:5.1f} us; hint sk={sk0}): " + "   ".join(out), flush=True)
