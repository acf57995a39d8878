"""Which neighbour makes the ds_bpermute build of the LayerNorm backward drop a lane?  (round 5; GPU box)

scripts/bperm_probe.hip found nothing with synthetic aggressors, so this one uses the library's own kernels: the text tower's LayerNorm
backward launch (160 x 512 bf16, fixed operands) repeated on one stream, a chosen kernel of the step repeated on another, every result
compared with the quiet result on the device.  Run it with the A/B library that still sums through ds_bpermute:
    CROG_LIB=$PWD/crog_amd/variants/libcrog_bperm.so python scripts/bperm_hunt.py [launches=3000]
and with the default library (DPP / v_readlane sums) for the control."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = "cuda"
torch.manual_seed(0)
bf = torch.bfloat16
M, C = 160, 512
x = torch.randn(M, C, device=dev).to(bf)
dout = (torch.randn(M, C, device=dev) * 0.02).to(bf)
gamma = torch.rand(C, device=dev) + 0.5
beta = torch.zeros(C, device=dev)
out = torch.empty_like(x)
stats = torch.empty(M, 2, device=dev)
K.ln_fwd(x, gamma, beta, 1e-5, out, stats)
rpb = K.ln_bwd_rows_per_block(M)
nb = (M + rpb - 1) // rpb
partial = torch.empty(nb, C, 2, device=dev)


def ln_bwd(dx):
    K.ln_bwd(dout, None, x, gamma, stats, dx, partial, rpb)


ref = torch.empty_like(x)
ln_bwd(ref)
torch.cuda.synchronize()

# ---- neighbours -----------------------------------------------------------------------------------------------------------------
Bn, H, W = 8, 52, 52
P = Bn * H * W
xa = torch.randn(P, 256, device=dev).to(bf)
w3 = (torch.randn(256, 9 * 256, device=dev) * 0.02).to(bf)
y3 = torch.empty(P, 256, device=dev, dtype=bf)
w1 = (torch.randn(512, 256, device=dev) * 0.05).to(bf)
y1 = torch.empty(P, 512, device=dev, dtype=bf)
dy1 = (torch.randn(P, 512, device=dev) * 0.1).to(bf)
g1 = torch.zeros(512, 256, device=dev)
g3 = torch.zeros(256, 9 * 256, device=dev)
zs = torch.empty(P, 512, device=dev, dtype=bf)
ss = torch.rand(512, 2, device=dev)
xl = torch.randn(21632, 2048, device=dev).to(bf)
ol = torch.empty_like(xl)
sl = torch.empty(21632, 2, device=dev)
gl, bl = torch.ones(2048, device=dev), torch.zeros(2048, device=dev)


def conv3():      # ping-pong LDS-DMA kernel (gemm_pp_kernel<A_IM2COL>), 128 KiB of LDS, s_setprio around its MFMA phases
    K.gemm(K.BF16, K.A_IM2COL, K.B_KC, xa, w3, y3, P, 256, 9 * 256, 256, 9 * 256, 256, conv=(H, W, 256))


def lin_pp():     # 1x1 forward on the ping-pong tile
    K.gemm(K.BF16, K.A_KC, K.B_KC, xa, w1, y1, P, 512, 256, 256, 256, 512)


def wgrad1():     # 128 x 128 LDS-DMA weight gradient with split-K atomics (gemm_dma_kernel)
    K.gemm(K.BF16, K.A_MC, K.B_NC, dy1, xa, g1, 512, 256, P, 512, 256, 256, splitk=16, out_mode=K.OUT_F32_ATOMIC)


def wgrad3():     # 3x3 weight gradient (ping-pong transposed-operand kernel or the 128 x 128 tile, whatever the dispatcher takes)
    K.gemm(K.BF16, K.A_MC, K.B_NC_IM2COL, y3, xa, g3, 256, 9 * 256, P, 256, 256, 9 * 256, splitk=16, out_mode=K.OUT_F32_ATOMIC, conv=(H, W, 256))


def bn_apply():   # HBM-bound elementwise kernel, no LDS
    K.bn_apply(y1, ss, None, True, zs)


def ln_wide():    # LayerNorm(2048) forward: the row kernel (wave sums + LDS slots)
    K.ln_fwd(xl, gl, bl, 1e-5, ol, sl)


neigh = [("none", None), ("3x3 forward, ping-pong LDS-DMA tile", conv3), ("1x1 forward, ping-pong tile", lin_pp), ("1x1 weight gradient, 128 x 128 LDS-DMA tile + atomics", wgrad1),
         ("3x3 weight gradient", wgrad3), ("BatchNorm apply (no LDS)", bn_apply), ("LayerNorm(2048) forward", ln_wide)]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
print(f"library: {os.environ.get('CROG_LIB', 'default (DPP / v_readlane sums)')}; {N} LayerNorm-backward launches (160 x 512) per neighbour", flush=True)
for name, fn in neigh:
    bad = torch.zeros((), device=dev, dtype=torch.int64)
    rows = torch.zeros((), device=dev, dtype=torch.int64)
    torch.cuda.synchronize()
    if fn is not None:
        K.set_stream_override(sa.cuda_stream)
        for _ in range(max(1, N // 6)):
            fn()
        K.set_stream_override(None)
    with torch.cuda.stream(sb):
        dx = torch.empty_like(x)
        for i in range(N):
            ln_bwd(dx)
            d = (dx != ref).any(dim=1)
            bad += d.any()
            rows += d.sum()
            if fn is not None and i % 6 == 0 and sa.query():      # keep the neighbour stream busy for the whole loop
                K.set_stream_override(sa.cuda_stream)
                for _ in range(64):
                    fn()
                K.set_stream_override(None)
    torch.cuda.synchronize()
    print(f"  beside {name:58s}: {int(bad)} of {N} launches wrong ({int(rows)} rows)", flush=True)
