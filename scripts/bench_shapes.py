"""A/B micro-benchmark of selected GEMM shapes (GPU box). Env: CROG_LIB, CROG_GEMM_SHAPE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
dt = torch.bfloat16; dc = 1
tag = os.environ.get("TAG", "")
def conv(name, B, HW, Cin, Cout):
    M = B * HW * HW
    x = torch.randn(M, Cin, device="cuda").to(dt); w = (torch.randn(Cout, 9 * Cin, device="cuda") * 0.05).to(dt)
    y = torch.empty(M, Cout, device="cuda", dtype=dt); dy = torch.randn(M, Cout, device="cuda").to(dt)
    dx = torch.empty(M, Cin, device="cuda", dtype=dt); dw = torch.zeros(Cout, 9 * Cin, device="cuda")
    fl = 2.0 * M * Cout * 9 * Cin
    tf = timeit(lambda: K.gemm(dc, K.A_IM2COL, K.B_KC, x, w, y, M, Cout, 9 * Cin, Cin, 9 * Cin, Cout, conv=(HW, HW, Cin)))
    td = timeit(lambda: K.gemm(dc, K.A_IM2COL, K.B_NC_DGRAD, dy, w, dx, M, Cin, 9 * Cout, Cout, Cin, Cin, conv=(HW, HW, Cout)))
    sk = K.pick_splitk(Cout, 9 * Cin, M, 32, conv=True)
    tg = timeit(lambda: K.gemm(dc, K.A_MC, K.B_NC_IM2COL, dy, x, dw, Cout, 9 * Cin, M, Cout, Cin, 9 * Cin, conv=(HW, HW, Cin), splitk=sk, out_mode=K.OUT_F32_ATOMIC))
    print(f"{tag:10s} {name:28s} fwd {fl/tf/1e9:6.1f}  dgrad {fl/td/1e9:6.1f}  wgrad {fl/tg/1e9:6.1f} TF/s", flush=True)
def lin(name, M, Kd, N):
    x = torch.randn(M, Kd, device="cuda").to(dt); w = (torch.randn(N, Kd, device="cuda") * 0.05).to(dt)
    y = torch.empty(M, N, device="cuda", dtype=dt); dy = torch.randn(M, N, device="cuda").to(dt)
    dx = torch.empty(M, Kd, device="cuda", dtype=dt); dw = torch.zeros(N, Kd, device="cuda")
    fl = 2.0 * M * N * Kd
    tf = timeit(lambda: K.gemm(dc, K.A_KC, K.B_KC, x, w, y, M, N, Kd, Kd, Kd, N))
    td = timeit(lambda: K.gemm(dc, K.A_KC, K.B_NC, dy, w, dx, M, Kd, N, N, Kd, Kd))
    sk = K.pick_splitk(N, Kd, M, 32)
    tg = timeit(lambda: K.gemm(dc, K.A_MC, K.B_NC, dy, x, dw, N, Kd, M, N, Kd, Kd, splitk=sk, out_mode=K.OUT_F32_ATOMIC))
    print(f"{tag:10s} {name:28s} fwd {fl/tf/1e9:6.1f}  dgrad {fl/td/1e9:6.1f}  wgrad {fl/tg/1e9:6.1f} TF/s   (fwd {tf*1e3:.0f} us)", flush=True)
conv("proj 512->256 @104", 32, 104, 512, 256)
conv("proj 512->512 @52", 32, 52, 512, 512)
conv("l3 256->256 @52", 32, 52, 256, 256)
conv("neck 512->512 @26", 32, 26, 512, 512)
conv("l1 64->64 @104", 32, 104, 64, 64)
lin("ffn 512->2048 M=21632", 21632, 512, 2048)
lin("1x1 256->1280 @104", 346112, 256, 1280)
lin("text 2048->512 M=640", 640, 2048, 512)
lin("text 512->512 M=640", 640, 512, 512)
lin("square 4096", 4096, 4096, 4096)
