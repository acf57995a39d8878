"""1x1 / linear data gradients: dx = dy @ W as it is launched today (B = W, N-contiguous: transposed LDS reads) against the
forward-shaped form on a transposed copy of W (B = W^T, K-contiguous).  HBM-cold (operand sets rotated).  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n      # ms per call
dt = torch.bfloat16
tot = [0.0, 0.0]
for M, cin, C, n in [(21632, 2048, 512, 3), (640, 512, 2048, 12), (640, 512, 1536, 12), (21632, 512, 512, 12), (21632, 1024, 256, 5), (346112, 256, 64, 2),
                     (21632, 1536, 512, 2), (5408, 2048, 2048, 3), (640, 512, 512, 18), (640, 2048, 512, 12), (346112, 64, 256, 4), (86528, 512, 128, 3),
                     (21632, 256, 1024, 6), (21632, 512, 2048, 3), (86528, 128, 512, 4), (5408, 512, 2048, 3), (5408, 2048, 512, 3)]:
    nset = max(1, int(600e6 / (M * (cin + C) * 2)) + 1) if M > 1000 else 1
    dys = [torch.randn(M, C, device="cuda").to(dt) for _ in range(nset)]
    w = (torch.randn(C, cin, device="cuda") * 0.05).to(dt); wt = w.t().contiguous()
    dxs = [torch.empty(M, cin, device="cuda", dtype=dt) for _ in range(nset)]
    it = [0]
    def nn():
        i = it[0] = (it[0] + 1) % nset
        K.gemm(1, K.A_KC, K.B_NC, dys[i], w, dxs[i], M, cin, C, C, cin, cin)
    def nt():
        i = it[0] = (it[0] + 1) % nset
        K.gemm(1, K.A_KC, K.B_KC, dys[i], wt, dxs[i], M, cin, C, C, C, cin)
    a = timeit(nn, max(10, 3 * nset)) * 1e3; b = timeit(nt, max(10, 3 * nset)) * 1e3
    nn(); r1 = dxs[it[0]].float().clone(); nt(); r2 = dxs[it[0]].float()
    tot[0] += n * a; tot[1] += n * b
    print(f"M={M:7d} cin={cin:5d} C={C:5d} x{n:2d}: NN {a:7.1f} us  NT {b:7.1f} us  ({(a-b)/a*100:+5.1f} %)", flush=True)
print(f"per step: NN {tot[0]/1e3:.2f} ms, NT {tot[1]/1e3:.2f} ms")
