"""A/B of the ping-pong weight-gradient kernel (csrc/gemm_ppt.hip) against the 256 x 256 atomic tile it replaces, on the large weight
gradients of the training step (GPU box).  Interleaved rounds in ONE process, operand sets rotated past the Infinity Cache.
Variants by descriptor debug bits: 65536 = previous kernels, 32768 | d << 12 = ping-pong at DMA distance d; "+slab" = split-K slabs and
the ordered reduction launch instead of atomic adds.  argv[1] = block target (default 144: crog_gemm_splitk_hint's)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K

dt = torch.bfloat16
target = int(sys.argv[1]) if len(sys.argv) > 1 else 0
# (B, HW, Cin, Cout, 3x3?)
shapes = [(32, 26, 512, 512, True), (32, 104, 512, 256, True), (32, 52, 512, 512, True), (32, 26, 1024, 512, True), (32, 26, 256, 256, True),
          (32, 52, 256, 256, True), (32, 52, 512, 256, True), (32, 26, 512, 2048, False), (32, 26, 2048, 512, False), (32, 13, 2048, 2048, False),
          (32, 26, 512, 512, False), (32, 26, 1024, 512, False)]
for B, HW, Cin, Cout, conv3 in shapes:
    Mpix = B * HW * HW
    N = 9 * Cin if conv3 else Cin
    bl = K.B_NC_IM2COL if conv3 else K.B_NC
    sk0 = K.lib().crog_gemm_splitk_hint(K.BF16, K.A_MC, bl, Cout, N, Mpix)
    tiles = -(-Cout // 256) * -(-N // 256)
    sk = max(1, min(target // tiles, Mpix // 64 // 8)) if target else sk0
    nset = max(2, int(600e6 // (Mpix * (Cin + Cout) * 2)) + 1)
    xs = [torch.randn(Mpix, Cin, device="cuda").to(dt) for _ in range(nset)]
    dys = [(torch.randn(Mpix, Cout, device="cuda") * 0.1).to(dt) for _ in range(nset)]
    g = torch.zeros(Cout, N, device="cuda")
    ws = torch.empty(sk, Cout, N, device="cuda")
    conv = (HW, HW, Cin) if conv3 else (0, 0, 0)
    variants = [("prev", 65536, False)] + [(f"ppt/{d}", 32768 | d << 12, False) for d in (3, 4, 5, 6)] + [("ppt/4+slab", 32768 | 4 << 12, True)]

    def run(i, flags, slab):
        K.DEBUG_FLAGS = flags
        if slab:
            K.gemm(1, K.A_MC, bl, dys[i], xs[i], ws, Cout, N, Mpix, Cout, Cin, N, splitk=sk, out_mode=K.OUT_F32, conv=conv)
            K.splitk_reduce(ws, sk, Cout, N, N, g, 0, N, accumulate=True)
        else:
            K.gemm(1, K.A_MC, bl, dys[i], xs[i], g, Cout, N, Mpix, Cout, Cin, N, splitk=sk, out_mode=K.OUT_F32_ATOMIC, conv=conv)
        K.DEBUG_FLAGS = 0
    fl = 2.0 * Mpix * Cout * N
    res = {n: [] for n, _, _ in variants}
    iters = 6
    for rnd in range(5):
        for name, flags, slab in variants:
            for i in range(2): run(i % nset, flags, slab)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(iters): run(i % nset, flags, slab)
            e.record(); torch.cuda.synchronize()
            res[name].append(s.elapsed_time(e) / iters)
    line = f"dW[{Cout:4d} x {N:4d}] K={Mpix:6d} {'3x3' if conv3 else '1x1'} sk={sk:2d} ({tiles * sk:3d} blocks):"
    for name, _, _ in variants:
        v = sorted(res[name]); med = v[len(v) // 2]
        line += f"  {name} {med*1e3:7.1f} us {fl/med/1e9:5.0f} TF"
    print(line, flush=True)
