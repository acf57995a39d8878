"""The residual-layer BatchNorm-backward epilogue (crog_gemm bwd_z + bwd_mask + R) on the step's shapes: the fused data gradient against
the plain one (+ R) and against the first pass it replaces (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
from crog_amd.functional import stat_replicas
dt = torch.bfloat16
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for M, N, Kd in [(346112, 256, 64), (86528, 512, 128), (21632, 1024, 256), (5408, 2048, 512)]:
    nset = max(2, int(700e6 // (M * N * 2 * 3)) + 1)
    a = [torch.randn(M, Kd, device="cuda").to(dt) for _ in range(nset)]
    w = (torch.randn(N, Kd, device="cuda") * 0.1).to(dt)
    z = [torch.randn(M, N, device="cuda").to(dt) for _ in range(nset)]
    r = [torch.randn(M, N, device="cuda").to(dt) for _ in range(nset)]
    mask = [torch.randint(0, 256, (M, N // 8), device="cuda", dtype=torch.uint8) for _ in range(nset)]
    out = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(nset)]
    R = stat_replicas(K.stat_tiles(M), N)
    sums = torch.zeros(R, N, 2, device="cuda")
    i = [0]
    def plain():
        j = i[0] % nset; i[0] += 1
        K.gemm(1, K.A_KC, K.B_KC, a[j], w, out[j], M, N, Kd, Kd, Kd, N, R=r[j], ldr=N)
    def fused():
        j = i[0] % nset; i[0] += 1
        K.gemm(1, K.A_KC, K.B_KC, a[j], w, out[j], M, N, Kd, Kd, Kd, N, R=r[j], ldr=N, col_stats=sums, stat_replicas=R, bwd_z=z[j], bwd_mask=mask[j])
    tp, tf = timeit(plain), timeit(fused)
    by_p = M * (Kd + 2 * N) * 2; by_f = M * (Kd + 3 * N) * 2 + M * N // 8
    print(f"M={M} N={N} K={Kd}: plain + R {tp:7.1f} us ({by_p/tp/1e6:4.2f} TB/s) | fused {tf:7.1f} us ({by_f/tf/1e6:4.2f} TB/s)  extra {tf - tp:6.1f} us; "
          f"the first pass it replaces reads {M*N*4.125/1e6:.0f} MB = {M*N*4.125/4.3e6:.1f} us at 4.3 TB/s", flush=True)
