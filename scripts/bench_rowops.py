"""Standalone bandwidth of the row kernels (LayerNorm, softmax, BN passes, colsum) at CROG-R50 decoder shapes. GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
from bench_gemm import timeit
dt = torch.bfloat16
def ln(M, C, drop):
    x = torch.randn(M, C, device="cuda").to(dt); g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    out = torch.empty_like(x); stats = torch.empty(M, 2, device="cuda"); dy = torch.randn(M, C, device="cuda").to(dt); dx = torch.empty_like(x)
    rpb = K.ln_bwd_rows_per_block(M); part = torch.empty(((M + rpb - 1) // rpb), C, 2, device="cuda")
    p = 0.1 if drop else 0.0
    tf = timeit(lambda: K.ln_fwd(x, g, b, 1e-5, out, stats, p_out=p, seed_out=5), 30)
    tb = timeit(lambda: K.ln_bwd(dy, None, x, g, stats, dx, part, rpb, p_out=p, seed_out=5), 30)
    by = M * C * 2
    print(f"LN   M={M:6d} C={C:5d} drop={int(drop)}: fwd {tf*1e3:6.1f} us ({2*by/tf/1e9:5.2f} TB/s)  bwd {tb*1e3:6.1f} us ({3*by/tb/1e9:5.2f} TB/s)")
def sm(BH, Lq, Lk, drop):
    Lkp = (Lk + 7) // 8 * 8
    S = torch.randn(BH, Lq, Lkp, device="cuda").to(dt); Pd = torch.empty_like(S) if drop else None; dP = torch.randn(BH, Lq, Lkp, device="cuda").to(dt)
    p = 0.1 if drop else 0.0
    tf = timeit(lambda: K.softmax_fwd(S, BH * Lq, Lq, Lk, Lkp, 8, False, None, Pd, p, 7), 30)
    tb = timeit(lambda: K.softmax_bwd(S, dP, BH * Lq, Lk, Lkp, p, 7), 30)
    by = BH * Lq * Lkp * 2
    print(f"SM   BH={BH} Lq={Lq} Lk={Lk} drop={int(drop)}: fwd {tf*1e3:6.1f} us ({(3 if drop else 2)*by/tf/1e9:5.2f} TB/s)  bwd {tb*1e3:6.1f} us ({3*by/tb/1e9:5.2f} TB/s)")
def cs(M, C):
    x = torch.randn(M, C, device="cuda").to(dt); out = torch.zeros(C, device="cuda")
    t = timeit(lambda: K.colsum(x, out), 30)
    print(f"COLSUM M={M:6d} C={C:5d}: {t*1e3:6.1f} us ({M*C*2/t/1e9:5.2f} TB/s)")
for M, C in ((21632, 512), (21632, 2048), (640, 512), (5408, 2048)):
    ln(M, C, False)
ln(21632, 512, True)
sm(256, 676, 676, True); sm(256, 676, 676, False); sm(256, 676, 20, True); sm(256, 20, 20, False)
for M, C in ((21632, 512), (21632, 2048), (21632, 1536), (640, 512), (640, 2048), (346112, 1280)):
    cs(M, C)
