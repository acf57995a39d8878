"""LayerNorm kernels at the decoder's shapes (21632 rows x 512 / 2048 channels, bf16) with and without dropout: time and algorithmic
TB/s (GPU box; CROG_LIB selects the build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
dt = torch.bfloat16
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for M, C in [(21632, 2048), (21632, 512)]:
    nset = 4
    xs = [torch.randn(M, C, device="cuda").to(dt) for _ in range(nset)]
    res = [torch.randn(M, C, device="cuda").to(dt) for _ in range(nset)]
    outs = [torch.empty(M, C, device="cuda", dtype=dt) for _ in range(nset)]
    dxs = [torch.empty(M, C, device="cuda", dtype=dt) for _ in range(nset)]
    gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
    stats = torch.empty(M, 2, device="cuda")
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    rpb = K.ln_bwd_rows_per_block(M)
    part = torch.empty((M + rpb - 1) // rpb, C, 2, device="cuda")
    for p_in, p_out, with_res in [(0.0, 0.0, False), (0.1, 0.0, False), (0.0, 0.1, True)]:
        i = [0]
        def fwd():
            j = i[0] % nset; i[0] += 1
            K.ln_fwd(xs[j], gamma, beta, 1e-5, outs[j], stats, res=res[j] if with_res else None, p_in=p_in, seed_in=5, p_out=p_out, seed_out=7)
        def bwd():
            j = i[0] % nset; i[0] += 1
            K.ln_bwd(res[j], None, xs[j], gamma, stats, dxs[j], part, rpb, p_in=p_in, seed_in=5, p_out=p_out, seed_out=7)
        tf, tb = t(fwd), t(bwd)
        by_f = M * C * 2 * (3 if with_res else 2); by_b = M * C * 2 * 3
        print(f"M={M} C={C} p_in={p_in} p_out={p_out} res={int(with_res)}: ln_fwd {tf:6.1f} us {by_f/tf/1e6:5.2f} TB/s | ln_bwd {tb:6.1f} us {by_b/tb/1e6:5.2f} TB/s", flush=True)
