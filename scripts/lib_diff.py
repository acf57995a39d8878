"""Bitwise A/B of kernel outputs between two builds of the library (GPU box): scripts/lib_diff.py OUT.npz  (run once per CROG_LIB), then
scripts/lib_diff.py A.npz B.npz prints what differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    for k in a.files:
        x, y = a[k], b[k]
        nd = int((x != y).sum())
        print(f"{k:40s} {'same' if nd == 0 else f'{nd} of {x.size} differ, max |d| {np.abs(x.astype(np.float64) - y.astype(np.float64)).max():.3e} (scale {np.abs(x).max():.3e})'}")
    sys.exit(0)
import torch
from crog_amd import kernels as K
dev = "cuda"
out = {}
g = torch.Generator(device=dev).manual_seed(1)
def rnd(*s, dt=torch.float32, sc=1.0):
    return (torch.randn(*s, device=dev, generator=g) * sc).to(dt)
bf = torch.bfloat16
# ---- fused attention, causal text shape and the decoder's 676-token shape, forward + backward
for name, (B, H, L, causal, p) in dict(text=(8, 8, 20, True, 0.0), dec=(4, 8, 676, False, 0.1)).items():
    E, dh = H * 64, 64
    qkv = rnd(B * L, 3 * E, dt=bf, sc=0.5)
    O = torch.empty(B * L, E, device=dev, dtype=bf); lse = torch.empty(B * H * L, device=dev)
    Lkp = (L + 7) // 8 * 8
    K.flash_attn_fwd((qkv, 0, 3 * E), (qkv, E, 3 * E), (qkv, 2 * E, 3 * E), (O, 0, E), lse, B, H, L, L, dh, dh ** -0.5, p, 77, Lkp, causal=causal)
    dO = rnd(B * L, E, dt=bf, sc=0.1); D = torch.empty_like(lse); dqkv = torch.empty_like(qkv)
    K.flash_attn_bwd((qkv, 0, 3 * E), (qkv, E, 3 * E), (qkv, 2 * E, 3 * E), (O, 0, E), (dO, 0, E), lse, D, (dqkv, 0, 3 * E), (dqkv, E, 3 * E), (dqkv, 2 * E, 3 * E),
                     B, H, L, L, dh, dh ** -0.5, p, 77, Lkp, causal=causal)
    out[f"flash_{name}_O"], out[f"flash_{name}_lse"], out[f"flash_{name}_D"], out[f"flash_{name}_dqkv"] = O.float(), lse, D, dqkv.float()
# ---- GEMMs with column statistics: fp32 (generic epilogue), bf16 128 x 128 and ping-pong tiles; a weight gradient with a_sum
for name, (dt, M, N, Kd) in dict(f32=(torch.float32, 1000, 96, 72), bf_mid=(bf, 5408, 256, 512), bf_pp=(bf, 86528, 256, 256)).items():
    A, Bm = rnd(M, Kd, dt=dt), rnd(N, Kd, dt=dt, sc=0.1)
    C = torch.empty(M, N, device=dev, dtype=dt)
    st = torch.zeros(K.stat_tiles(M), N, 2, device=dev)
    K.gemm(K.dcode(dt), K.A_KC, K.B_KC, A, Bm, C, M, N, Kd, Kd, Kd, N, col_stats=st)
    out[f"gemm_{name}_C"], out[f"gemm_{name}_stats"] = C.float(), st
for name, dt in dict(f32=torch.float32, bf16=bf).items():
    dy, x = rnd(3000, 96, dt=dt, sc=0.1), rnd(3000, 64, dt=dt)
    G = torch.zeros(96, 64, device=dev); bsum = torch.zeros(96, device=dev)
    K.gemm(K.dcode(dt), K.A_MC, K.B_NC, dy, x, G, 96, 64, 3000, 96, 64, 64, splitk=1, out_mode=K.OUT_F32_ATOMIC, a_sum=bsum)
    out[f"wgrad_{name}_G"], out[f"wgrad_{name}_asum"] = G, bsum
# ---- softmax rows (group sums), LayerNorm
for LPRname, (rows, Lk) in dict(sm20=(640, 20), sm676=(2704, 676)).items():
    ldp = (Lk + 7) // 8 * 8
    S = rnd(rows, ldp, dt=bf)
    K.softmax_fwd(S, rows, rows, Lk, ldp, 1, False, None, None, 0.0, 0)
    out[f"{LPRname}_P"] = S.float()
x = rnd(640, 512, dt=bf); o = torch.empty_like(x); stt = torch.empty(640, 2, device=dev)
K.ln_fwd(x, torch.ones(512, device=dev), torch.zeros(512, device=dev), 1e-5, o, stt)
out["ln_out"], out["ln_stats"] = o.float(), stt
torch.cuda.synchronize()
np.savez(sys.argv[1], **{k: v.cpu().numpy() for k, v in out.items()})
print("saved", sys.argv[1])
