"""What bounds the small-output 1x1 / linear weight gradients: atomics (debug bit 5 skips them) vs the streaming main loop (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n      # ms per call
dt = torch.bfloat16
def lin(M, Kd, N):
    # rotate over enough operand sets to overflow the 256 MB Infinity Cache: in the training step these operands come from HBM
    nset = max(2, int(800e6 / ((M * Kd + M * N) * 2)) + 1)
    xs = [torch.randn(M, Kd, device="cuda").to(dt) for _ in range(nset)]; dys = [torch.randn(M, N, device="cuda").to(dt) for _ in range(nset)]
    dw = torch.zeros(N, Kd, device="cuda")
    it = [0]
    def run(sk):
        i = it[0] = (it[0] + 1) % nset
        K.gemm(1, K.A_MC, K.B_NC, dys[i], xs[i], dw, N, Kd, M, N, Kd, Kd, splitk=sk, out_mode=K.OUT_F32_ATOMIC)
    sk0 = K.pick_splitk(N, Kd, M, 32)
    out = []
    for sk in (max(1, sk0 // 2), sk0, sk0 * 2):
        r = []
        for flag in (0, 32):
            K.DEBUG_FLAGS = flag
            r.append(timeit(lambda: run(sk), 3 * nset) * 1e3)
        K.DEBUG_FLAGS = 0
        out.append(f"sk={sk:3d}: {r[0]:6.1f} us, no atomics {r[1]:6.1f}")
    byt = (M * Kd + M * N) * 2
    print(f"wgrad {Kd:4d}->{N:4d} M={M:6d} (operands at 4.5 TB/s {byt/4.5e6:5.1f} us)  " + "   ".join(out), flush=True)
for a in [(21632, 512, 512), (21632, 256, 1024), (21632, 1024, 256), (21632, 512, 2048), (21632, 2048, 512), (5408, 2048, 2048), (86528, 128, 512), (86528, 512, 128), (346112, 64, 256), (346112, 256, 64), (346112, 64, 64), (640, 512, 512), (640, 2048, 512)]:
    lin(*a)
