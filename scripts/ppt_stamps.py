"""Where the waves of one block of the ping-pong weight-gradient kernel spend a phase: s_memtime distances summed in the kernel
(build: scripts/build_variant.py pptstamp -DCROG_PPT_STAMP=1; GPU box: CROG_LIB=crog_amd/variants/libcrog_pptstamp.so python scripts/ppt_stamps.py).
The stamps cost registers (scalar spills) and a scalar-memory round trip each: read the SHARES, and the total against the unstamped kernel."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
dt = torch.bfloat16
dense = "dense" in sys.argv
if dense:
    Cout, N, Mpix = 256, 4608, 86528
    x = torch.randn(Mpix, N, device="cuda").to(dt)
else:
    B, HW, Cin, Cout = 32, 104, 512, 256
    Mpix, N = B * HW * HW, 9 * Cin
    x = torch.randn(Mpix, Cin, device="cuda").to(dt)
dy = (torch.randn(Mpix, Cout, device="cuda") * 0.1).to(dt)
g = torch.zeros(Cout, N, device="cuda")
sk = K.lib().crog_gemm_splitk_hint(K.BF16, K.A_MC, K.B_NC if dense else K.B_NC_IM2COL, Cout, N, Mpix)
def run():
    if dense:
        K.gemm(1, K.A_MC, K.B_NC, dy, x, g, Cout, N, Mpix, Cout, N, N, splitk=sk, out_mode=K.OUT_F32_ATOMIC)
    else:
        K.gemm(1, K.A_MC, K.B_NC_IM2COL, dy, x, g, Cout, N, Mpix, Cout, Cin, N, splitk=sk, out_mode=K.OUT_F32_ATOMIC, conv=(HW, HW, Cin))
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); run(); e.record(); torch.cuda.synchronize()
tiles = (Cout // 256) * (N // 256); ktb = (Mpix // 64 + sk - 1) // sk
print(f"{'dense' if dense else '3x3'} dW[{Cout} x {N}] over {Mpix}: {tiles * sk} blocks x {ktb} k-tiles, {s.elapsed_time(e) * 1e3:.1f} us = {s.elapsed_time(e) * 1e6 / ktb:.0f} ns per k-tile")
fn = getattr(K.lib(), "ppt_probe_stamps", None)
if fn is None:
    sys.exit("this library has no stamps (build with -DCROG_PPT_STAMP=1)")
out = (ctypes.c_uint * 64)()
fn.argtypes = [ctypes.c_void_p]
assert fn(out) == 0
names = ["-", "issue (reads + requests)", "vmcnt wait", "lgkmcnt wait", "barrier 1", "MFMAs", "barrier 2"]
for w in range(8):
    v = out[8 * w:8 * w + 8]
    n = max(v[7], 1)
    tot = sum(v[1:7])
    print(f"wave {w} (group {w >> 2}): {n} phases, {tot / n:6.0f} cycles per phase: " + ", ".join(f"{names[i]} {v[i] / n:5.0f}" for i in range(1, 7)))
