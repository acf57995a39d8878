"""A/B of the ping-pong 256 x 256 x 64 kernel (csrc/gemm_pp.hip) against the kernels it replaces, on the K-contiguous forward-shaped
GEMMs of the training step (GPU box).  Interleaved rounds in ONE process, random operands, operand sets rotated past the Infinity Cache.
Variants by descriptor debug bits: 2048 = the previous path (gemm_dma16_kernel / 128 x 128), 512 | d<<12 = 256-row ping-pong at DMA
distance d, 1024 | d<<12 = 192-row tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K

dt = torch.bfloat16


def run(x, w, y, M, N, Kd, conv, flags, stats=None):
    K.DEBUG_FLAGS = flags
    if conv:
        K.gemm(1, K.A_IM2COL, K.B_KC, x, w, y, M, N, Kd, conv[2], Kd, N, conv=conv, col_stats=stats, stat_replicas=0 if stats is None else stats.shape[0])
    else:
        K.gemm(1, K.A_KC, K.B_KC, x, w, y, M, N, Kd, Kd, Kd, N, col_stats=stats, stat_replicas=0 if stats is None else stats.shape[0])
    K.DEBUG_FLAGS = 0


# (B, HW, Cin, Cout, 3x3?)  — the 3x3 forwards / data gradients of CROG-R50 at B = 32, then wide 1x1 / linear layers
shapes = [(32, 104, 256, 512, True), (32, 104, 512, 256, True), (32, 52, 512, 512, True), (32, 52, 256, 512, True), (32, 52, 256, 256, True),
          (32, 26, 512, 512, True), (32, 26, 512, 1024, True), (32, 26, 1024, 512, True), (32, 26, 256, 256, True), (32, 104, 128, 256, True),
          (32, 26, 2048, 512, False), (32, 26, 512, 2048, False), (32, 52, 512, 512, False), (32, 26, 1024, 1024, False), (32, 104, 256, 256, False)]
variants = [("prev", 2048)] + [(f"pp256/{d}", 512 | d << 12) for d in (3, 4, 5, 6, 7)] + [(f"pp192/{d}", 1024 | d << 12) for d in (3, 4, 5, 6, 7)]
if "rows" in sys.argv:      # tile heights at the default distance, on the mid-size launches: "prev" here is today's DEFAULT dispatch
    variants = [("default", 0), ("pp256", 512), ("pp192", 1024), ("pp128", 524288)]
    shapes = [(32, 26, 256, 256, True), (32, 13, 512, 512, True), (32, 26, 512, 512, True), (32, 52, 128, 128, True), (32, 26, 1024, 256, False),
              (32, 26, 256, 1024, False), (32, 13, 2048, 512, False), (32, 13, 512, 2048, False), (32, 13, 1024, 2048, False), (32, 26, 512, 256, False)]
if "lin" in sys.argv:       # round 6: the 1x1 / linear shapes of the step that run furthest from their HBM roof (profiles/r06_linfwd_table_serial.txt)
    variants = [("default", 0), ("prev", 2048), ("pp256", 512), ("pp192", 1024), ("pp128", 524288)]
    shapes = [(32, 26, 256, 1024, False), (32, 26, 1024, 256, False), (32, 52, 128, 512, False), (32, 104, 64, 256, False), (32, 26, 512, 512, False),
              (32, 104, 128, 256, False), (32, 26, 512, 1024, False), (32, 52, 256, 512, False), (32, 26, 512, 2048, False), (32, 104, 256, 64, False)]
if len(sys.argv) > 1 and sys.argv[1] == "stats":      # with BatchNorm statistics in the epilogue, as the training step launches them
    with_stats = True
else:
    with_stats = False
for B, HW, Cin, Cout, conv3 in shapes:
    M = B * HW * HW
    Kd = 9 * Cin if conv3 else Cin
    nset = max(2, int(600e6 // (M * (Cin + Cout) * 2)) + 1)
    xs = [torch.randn(M, Cin, device="cuda").to(dt) for _ in range(nset)]
    ws = [(torch.randn(Cout, Kd, device="cuda") * 0.05).to(dt) for _ in range(nset)]
    ys = [torch.empty(M, Cout, device="cuda", dtype=dt) for _ in range(nset)]
    st = torch.zeros(8, Cout, 2, device="cuda") if with_stats else None
    conv = (HW, HW, Cin) if conv3 else None
    fl = 2.0 * M * Cout * Kd
    res = {n: [] for n, _ in variants}
    iters = 6
    for rnd in range(5):
        for name, flags in variants:
            for i in range(2): run(xs[i % nset], ws[i % nset], ys[i % nset], M, Cout, Kd, conv, flags, st)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(iters): run(xs[i % nset], ws[i % nset], ys[i % nset], M, Cout, Kd, conv, flags, st)
            e.record(); torch.cuda.synchronize()
            res[name].append(s.elapsed_time(e) / iters)
    t256 = -(-M // 256) * (Cout // 256); t192 = -(-M // 192) * (Cout // 256)
    line = f"M={M:7d} N={Cout:5d} K={Kd:5d} {'3x3' if conv3 else '1x1'} tiles {t256:4d}/{t192:4d}:"
    for name, _ in variants:
        v = sorted(res[name]); med = v[len(v) // 2]
        line += f"  {name} {med*1e3:7.1f} us {fl/med/1e9:6.0f} TF"
    print(line, flush=True)
