"""Where the time of the skinny-K 1x1 GEMMs goes (GPU box): epilogue variants vs the bytes they move."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
from crog_amd.functional import stat_replicas
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
dt = torch.bfloat16
for M, Kd, N in [(346112, 64, 256), (346112, 256, 64), (346112, 64, 64), (86528, 128, 512), (86528, 512, 128), (21632, 256, 1024), (21632, 1024, 256), (346112, 256, 128)]:
    x = torch.randn(M, Kd, device="cuda").to(dt); w = (torch.randn(N, Kd, device="cuda") * 0.05).to(dt)
    y = torch.empty(M, N, device="cuda", dtype=dt); yf = torch.empty(M, N, device="cuda")
    R = stat_replicas(K.stat_tiles(M), N); stats = torch.zeros(R, N, 2, device="cuda")
    res = torch.randn(M, N, device="cuda").to(dt)
    byt = (M * Kd + M * N) * 2
    t_plain = timeit(lambda: K.gemm(1, K.A_KC, K.B_KC, x, w, y, M, N, Kd, Kd, Kd, N))
    t_stats = timeit(lambda: K.gemm(1, K.A_KC, K.B_KC, x, w, y, M, N, Kd, Kd, Kd, N, col_stats=stats, stat_replicas=R))
    t_f32 = timeit(lambda: K.gemm(1, K.A_KC, K.B_KC, x, w, yf, M, N, Kd, Kd, Kd, N, out_mode=K.OUT_F32))
    t_res = timeit(lambda: K.gemm(1, K.A_KC, K.B_KC, x, w, y, M, N, Kd, Kd, Kd, N, R=res, ldr=N))
    # data-gradient form: dy [M, N] @ W[N, K] -> dx [M, K]
    dx = torch.empty(M, Kd, device="cuda", dtype=dt)
    t_dg = timeit(lambda: K.gemm(1, K.A_KC, K.B_NC, y, w, dx, M, Kd, N, N, Kd, Kd))
    src = torch.empty(byt // 2, device="cuda", dtype=dt); dst = torch.empty_like(src)
    t_copy = timeit(lambda: dst[: M * N].copy_(src[: M * N]))   # moves 2 * M*N*2 bytes
    print(f"M={M:7d} K={Kd:4d} N={N:4d}: fwd {t_plain:6.1f} us ({byt/t_plain/1e6:5.2f} TB/s)  +stats {t_stats:6.1f}  f32-out {t_f32:6.1f} ({(M*Kd*2+M*N*4)/t_f32/1e6:5.2f} TB/s)"
          f"  +res {t_res:6.1f}  dgrad {t_dg:6.1f} ({byt/t_dg/1e6:5.2f} TB/s)   torch copy of the output bytes {t_copy:6.1f} us", flush=True)
