"""Timing of the forward ping-pong kernel (csrc/gemm_pp.hip) on four large 3x3 forwards of the step, default dispatch, for A/B builds
(scripts/build_variant.py, CROG_LIB).  GPU box: python scripts/pp_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
dt = torch.bfloat16
for B, HW, Cin, Cout in ((32, 104, 256, 512), (32, 104, 512, 256), (32, 52, 512, 512), (32, 26, 1024, 512)):
    M, N, Kd = B * HW * HW, Cout, 9 * Cin
    nset = max(2, int(600e6 // (M * (Cin + Cout) * 2)) + 1)
    xs = [torch.randn(M, Cin, device="cuda").to(dt) for _ in range(nset)]
    ws = [(torch.randn(N, Kd, device="cuda") * 0.02).to(dt) for _ in range(nset)]
    y = torch.empty(M, N, device="cuda", dtype=dt)
    def run(i):
        K.gemm(1, K.A_IM2COL, K.B_KC, xs[i], ws[i], y, M, N, Kd, Cin, Kd, N, conv=(HW, HW, Cin))
    for i in range(3): run(i % nset)
    torch.cuda.synchronize()
    ts = []
    for rnd in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(6): run(i % nset)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 6)
    t = sorted(ts)[2]
    print(f"[{os.environ.get('CROG_LIB', 'default')}] 3x3 forward {M} x {N} x {Kd}: {t * 1e3:8.1f} us ({2.0 * M * N * Kd / t / 1e9:6.0f} TFLOP/s)", flush=True)
