"""Fused attention kernels at the decoder's shape (B = 32, 8 heads, 676 tokens, head_dim 64, dropout 0.1) and the ViT tower's (B = 64, 12
heads, 197 tokens): time per launch of forward and backward (dQ + dK/dV) with CUDA events, and the forward against an fp32 softmax
attention (p = 0).  GPU box; run once per library (CROG_LIB) to A/B two builds."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
dt = torch.bfloat16
for B, heads, L, p in [(32, 8, 676, 0.1), (32, 8, 676, 0.0), (64, 12, 197, 0.0), (32, 8, 169, 0.0)]:
    E = heads * 64
    q, k, v, do = [(torch.randn(B * L, E, device="cuda") * 0.5).to(dt) for _ in range(4)]
    o = torch.empty_like(q); dq = torch.empty_like(q); dk = torch.empty_like(q); dv = torch.empty_like(q)
    lse = torch.empty(B * heads * L, device="cuda"); D = torch.empty_like(lse)
    sc = 1.0 / math.sqrt(64)
    fwd = lambda: K.flash_attn_fwd((q, 0, E), (k, 0, E), (v, 0, E), (o, 0, E), lse, B, heads, L, L, 64, sc, p, 1234, L)
    bwd = lambda: K.flash_attn_bwd((q, 0, E), (k, 0, E), (v, 0, E), (o, 0, E), (do, 0, E), lse, D, (dq, 0, E), (dk, 0, E), (dv, 0, E), B, heads, L, L, 64, sc, p, 1234, L)
    if p > 0 and os.environ.get("CROG_FLASH_KEEP", "1") != "0":      # the forward's dropout decisions as a bit map for the backward kernels
        keep = torch.empty(K.flash_keep_words(B, heads, L, L), device="cuda", dtype=torch.int32)
        fwd_h, bwd_h = fwd, bwd
        fwd = lambda: K.flash_attn_fwd((q, 0, E), (k, 0, E), (v, 0, E), (o, 0, E), lse, B, heads, L, L, 64, sc, p, 1234, L, keep=keep)
        bwd = lambda: K.flash_attn_bwd((q, 0, E), (k, 0, E), (v, 0, E), (o, 0, E), (do, 0, E), lse, D, (dq, 0, E), (dk, 0, E), (dv, 0, E), B, heads, L, L, 64, sc, p, 1234, L, keep=keep)
    fwd(); bwd(); torch.cuda.synchronize()
    err = None
    if p == 0.0:
        qf, kf, vf = [t.float().view(B, L, heads, 64).permute(0, 2, 1, 3) for t in (q, k, v)]
        ref = torch.softmax(qf @ kf.transpose(-1, -2) * sc, -1) @ vf
        err = float((o.float().view(B, L, heads, 64).permute(0, 2, 1, 3) - ref).abs().max())
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    tf, tb = t(fwd), t(bwd)
    if p > 0 and os.environ.get("CROG_FLASH_KEEP", "1") != "0":
        print(f"   (hashing in all three kernels: fwd {t(fwd_h):7.1f} us  bwd {t(bwd_h):7.1f} us)", flush=True)
    fl = 4.0 * B * heads * L * L * 64
    print(f"B={B} heads={heads} L={L} p={p}: fwd {tf:7.1f} us ({fl/tf/1e6:6.1f} TF/s)  bwd {tb:7.1f} us ({2.5*fl/tb/1e6:6.1f} TF/s)" + (f"  fwd max err {err:.2e}" if err is not None else ""), flush=True)
