"""Timing of the ping-pong weight-gradient kernel on its two largest launches, for the CROG_PPT_PROBE builds (scripts/build_variant.py pptN
-DCROG_PPT_PROBE=N; select with CROG_LIB): what a k-tile spends its time on.  GPU box: python scripts/ppt_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
dt = torch.bfloat16
K.DEBUG_FLAGS = int(os.environ.get('PROBE_D', '0')) << 12      # DMA distance 3 .. 7 (0 = the default, 5)
for B, HW, Cin, Cout in ((32, 104, 512, 256), (32, 52, 512, 512), (32, 26, 1024, 512)):
    Mpix, N = B * HW * HW, 9 * Cin
    sk = K.lib().crog_gemm_splitk_hint(K.BF16, K.A_MC, K.B_NC_IM2COL, Cout, N, Mpix)
    nset = max(2, int(600e6 // (Mpix * (Cin + Cout) * 2)) + 1)
    xs = [torch.randn(Mpix, Cin, device="cuda").to(dt) for _ in range(nset)]
    dys = [(torch.randn(Mpix, Cout, device="cuda") * 0.1).to(dt) for _ in range(nset)]
    g = torch.zeros(Cout, N, device="cuda")
    def run(i):
        K.gemm(1, K.A_MC, K.B_NC_IM2COL, dys[i], xs[i], g, Cout, N, Mpix, Cout, Cin, N, splitk=sk, out_mode=K.OUT_F32_ATOMIC, conv=(HW, HW, Cin))
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
    tiles = (Cout // 256) * (N // 256)
    ktb = (Mpix // 64 + sk - 1) // sk
    print(f"[{os.environ.get('CROG_LIB', 'default')} D={os.environ.get('PROBE_D', '-')}] dW[{Cout} x {N}] over {Mpix} pixels, {tiles * sk} blocks x {ktb} k-tiles: {t * 1e3:8.1f} us  = {t * 1e6 / ktb:6.1f} ns per k-tile  ({2.0 * Mpix * Cout * N / t / 1e9:6.0f} TFLOP/s)", flush=True)
