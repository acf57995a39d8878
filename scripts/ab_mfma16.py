"""A/B of the two bf16 MFMA shapes on the K-contiguous forward-shaped GEMMs (GPU box): v_mfma_f32_32x32x16 (K.DEBUG_FLAGS = 0) against
v_mfma_f32_16x16x32 (bit 6; bit 7 adds the 128 x 128 tile).  Interleaved rounds in ONE process, random operands, operand sets rotated
past the Infinity Cache; first a numerical check of the new variant against float64 (outputs and BatchNorm column statistics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K

dt = torch.bfloat16


def run(x, w, y, M, N, Kd, conv, stats=None, flags=0):
    K.DEBUG_FLAGS = flags
    if conv:
        K.gemm(1, K.A_IM2COL, K.B_KC, x, w, y, M, N, Kd, conv[2], Kd, N, conv=conv, col_stats=stats, stat_replicas=0 if stats is None else stats.shape[0])
    else:
        K.gemm(1, K.A_KC, K.B_KC, x, w, y, M, N, Kd, Kd, Kd, N, col_stats=stats, stat_replicas=0 if stats is None else stats.shape[0])
    K.DEBUG_FLAGS = 0


def check(B, HW, Cin, Cout, conv3, flags):
    torch.manual_seed(B * HW + Cin)
    M = B * HW * HW
    Kd = 9 * Cin if conv3 else Cin
    x = torch.randn(M, Cin, device="cuda").to(dt)
    w = (torch.randn(Cout, Kd, device="cuda") * 0.05).to(dt)
    y0 = torch.empty(M, Cout, device="cuda", dtype=dt); y1 = torch.empty_like(y0)
    s0 = torch.zeros(2, Cout, 2, device="cuda"); s1 = torch.zeros_like(s0)
    conv = (HW, HW, Cin) if conv3 else None
    run(x, w, y0, M, Cout, Kd, conv, s0, 256)
    run(x, w, y1, M, Cout, Kd, conv, s1, flags)
    torch.cuda.synchronize()
    if conv3:
        xi = x.float().view(B, HW, HW, Cin).permute(0, 3, 1, 2)
        wi = w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(xi.double(), wi.double(), padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
    else:
        ref = x.double() @ w.double().t()
    e0 = float((y0.double() - ref).abs().max()); e1 = float((y1.double() - ref).abs().max())
    sref = torch.stack([ref.sum(0), (ref * ref).sum(0)], 1)
    t0 = s0.sum(0).double(); t1 = s1.sum(0).double()
    se0 = float(((t0 - sref).abs() / (sref.abs() + 1)).max()); se1 = float(((t1 - sref).abs() / (sref.abs() + 1)).max())
    same = float((y0.float() - y1.float()).abs().max())
    print(f"check B={B} HW={HW} {Cin}->{Cout} conv3={conv3} flags={flags}: out err 32x32 {e0:.3e} 16x16 {e1:.3e} (|y0-y1| {same:.3e}); stats rel err {se0:.2e} / {se1:.2e}", flush=True)
    assert e1 <= max(2 * e0, 0.1) and se1 < 1e-3, "16x16x32 variant is wrong"


check(2, 26, 64, 256, True, 64 | 128)          # 128^2 tile, M = 1352 (ragged last tile is excluded by M % 128: falls back) -> sanity
check(32, 26, 256, 512, True, 64)              # 256^2 tile, M = 21632 = 84.5 tiles: row guard
check(8, 52, 256, 256, True, 64 | 128)         # 128^2 tile
check(8, 52, 256, 512, False, 64 | 128)        # 1x1, 128^2 tile
if len(sys.argv) > 1 and sys.argv[1] == "check":
    sys.exit(0)

shapes = [(32, 104, 256, 512, True), (32, 104, 512, 256, True), (32, 52, 512, 512, True), (32, 52, 256, 512, True), (32, 26, 512, 512, True),
          (32, 26, 512, 1024, True), (32, 26, 1024, 512, True), (32, 52, 256, 256, True), (32, 26, 256, 256, True), (32, 104, 128, 128, True),
          (32, 52, 512, 128, False), (32, 26, 1024, 256, False), (32, 26, 256, 1024, False), (32, 26, 2048, 512, False)]
variants = [("32x32x16", 256), ("16x16x32", 64 | 128)]
for B, HW, Cin, Cout, conv3 in shapes:
    M = B * HW * HW
    Kd = 9 * Cin if conv3 else Cin
    nset = max(2, int(600e6 // (M * (Cin + Cout) * 2)) + 1)
    xs = [torch.randn(M, Cin, device="cuda").to(dt) for _ in range(nset)]
    ws = [(torch.randn(Cout, Kd, device="cuda") * 0.05).to(dt) for _ in range(nset)]
    ys = [torch.empty(M, Cout, device="cuda", dtype=dt) for _ in range(nset)]
    conv = (HW, HW, Cin) if conv3 else None
    fl = 2.0 * M * Cout * Kd
    res = {n: [] for n, _ in variants}
    iters = 8
    for rnd in range(5):
        for name, flags in variants:
            for i in range(2): run(xs[i % nset], ws[i % nset], ys[i % nset], M, Cout, Kd, conv, None, flags)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(iters): run(xs[i % nset], ws[i % nset], ys[i % nset], M, Cout, Kd, conv, None, flags)
            e.record(); torch.cuda.synchronize()
            res[name].append(s.elapsed_time(e) / iters)
    line = f"M={M:7d} N={Cout:5d} K={Kd:5d} {'3x3' if conv3 else '1x1'}:"
    for name, _ in variants:
        v = sorted(res[name]); med = v[len(v) // 2]
        line += f"  {name} med {med*1e3:7.1f} us {fl/med/1e9:7.1f} TF/s (min {v[0]*1e3:7.1f})"
    a = sorted(res["32x32x16"])[2]; b = sorted(res["16x16x32"])[2]
    print(line + f"  ratio {a / b:.3f}", flush=True)
