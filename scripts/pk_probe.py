"""Stand-alone reproducer of round 5's second run-to-run difference (LAB_NOTES section 10): the bilinear x2 BACKWARD kernel
(upsample2_bwd_quad_kernel, pure register arithmetic: 36 loads, packed-fp32 FMAs the compiler's SLP vectoriser makes of the scalar code,
4 stores; no LDS, no cross-lane operation) beside a 3x3 weight-gradient GEMM on another stream.  The same launch on the same input is repeated
and compared with its serial result.  usage: pk_probe.py [iterations=3000] [aggressor=wgrad|none]
CROG_LIB=crog_amd/variants/libcrog_noslp_elt.so (scripts/build_variant.py, eltwise.hip with -fno-slp-vectorize) selects the build without
v_pk_*_f32 in the victim."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from crog_amd import kernels as K

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
AGG = sys.argv[2] if len(sys.argv) > 2 else "wgrad"
B, H, W, C = 8, 13, 13, 512
torch.manual_seed(0)
cat = (torch.randn(B, 2 * H, 2 * W, 3 * C, device="cuda") * 0.01).to(torch.bfloat16)      # the gradient of the neck's concat buffer
dy = cat[..., 2 * C:]                                                                     # its third channel slice: the victim's input
dx = torch.empty(B, H, W, C, device="cuda", dtype=torch.bfloat16)
x = (torch.randn(B, 2 * H, 2 * W, C, device="cuda") * 0.5).to(torch.bfloat16)              # f4_proj4's input
G = torch.zeros(C, 9 * C, device="cuda", dtype=torch.float32)
rows = B * 2 * H * 2 * W
sk = K.pick_splitk(C, 9 * C, rows, 64, conv=True)
side = torch.cuda.Stream()


def aggressor():
    side.wait_stream(torch.cuda.current_stream())
    K.set_stream_override(side.cuda_stream)
    try:      # dW[512][9 x 512] += dy_slice^T x im2col(x): A = the SECOND channel slice of the concat gradient (row stride 3C, offset C)
        K.gemm(K.BF16, K.A_MC, K.B_NC_IM2COL, cat, x, G, C, 9 * C, rows, 3 * C, C, 9 * C, a_off=C, conv=(2 * H, 2 * W, C), splitk=sk,
               out_mode=K.OUT_F32_ATOMIC)
    finally:
        K.set_stream_override(None)


for _ in range(3):
    K.upsample2_bwd(dy, dx)
    torch.cuda.synchronize()
ref = dx.clone()
K.upsample2_bwd(dy, dx)
torch.cuda.synchronize()
assert torch.equal(dx, ref)
bad, shown = 0, 0
for it in range(N):
    dx.fill_(7.0)
    if AGG != "none":
        aggressor()
    K.upsample2_bwd(dy, dx)
    torch.cuda.synchronize()
    if not torch.equal(dx, ref):
        bad += 1
        if shown < 6:
            shown += 1
            d = (dx != ref).reshape(-1, C)
            r, c = d.any(1).nonzero().flatten(), d.any(0).nonzero().flatten()
            a, b = dx.reshape(-1, C)[d], ref.reshape(-1, C)[d]
            print(f"  iteration {it}: {int(d.sum())} elements differ, pixel rows {r.tolist()[:8]}, columns {c.tolist()[:10]} .. {c.tolist()[-2:]}; "
                  f"got {a[:4].float().tolist()} want {b[:4].float().tolist()}")
print(f"[{os.environ.get('CROG_LIB', 'default build')}] aggressor {AGG} (splitk {sk}): {bad} of {N} launches of the bilinear backward differ from the serial result", flush=True)
