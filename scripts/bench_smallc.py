"""Small-channel layers of the stem / layer1 at B = 32 (N <= 64 output columns: the 256 x 64 tile): 3x3 and 1x1 forwards with BatchNorm
statistics, operand sets rotated past the Infinity Cache (GPU box).  Run once per library (CROG_LIB) to A/B two builds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
from crog_amd.functional import stat_replicas
dt = torch.bfloat16
for B, HW, Cin, Cout, conv3 in [(32, 208, 32, 32, True), (32, 208, 32, 64, True), (32, 104, 64, 64, True), (32, 104, 256, 64, False), (32, 104, 64, 64, False), (32, 208, 32, 32, False)]:
    M = B * HW * HW
    Kd = 9 * Cin if conv3 else Cin
    nset = max(2, int(600e6 // (M * (Cin + Cout) * 2)) + 1)
    xs = [torch.randn(M, Cin, device="cuda").to(dt) for _ in range(nset)]
    w = (torch.randn(Cout, Kd, device="cuda") * 0.05).to(dt)
    ys = [torch.empty(M, Cout, device="cuda", dtype=dt) for _ in range(nset)]
    R = stat_replicas(K.stat_tiles(M), Cout)
    stats = torch.zeros(R, Cout, 2, device="cuda")
    def run(i):
        if conv3: K.gemm(1, K.A_IM2COL, K.B_KC, xs[i], w, ys[i], M, Cout, Kd, Cin, Kd, Cout, conv=(HW, HW, Cin), col_stats=stats, stat_replicas=R)
        else: K.gemm(1, K.A_KC, K.B_KC, xs[i], w, ys[i], M, Cout, Kd, Cin, Kd, Cout, col_stats=stats, stat_replicas=R)
    ts = []
    for rnd in range(5):
        for i in range(2): run(i % nset)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(10): run(i % nset)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 10)
    t = sorted(ts)[2]
    by = M * (Cin + Cout) * 2
    print(f"{'3x3' if conv3 else '1x1'} M={M:8d} {Cin:3d}->{Cout:3d}: {t*1e3:7.1f} us  {2.0*M*Cout*Kd/t/1e9:6.1f} TF/s  {by/t/1e9:5.2f} TB/s", flush=True)
