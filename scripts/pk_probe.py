"""Stand-alone reproducer of round 5's second run-to-run difference (LAB_NOTES section 10): the bilinear x2 BACKWARD kernel
(upsample2_bwd_quad_kernel, pure register arithmetic: 36 loads, packed-fp32 FMAs the compiler's SLP vectoriser makes of the scalar code,
4 stores; no LDS, no cross-lane operation) beside a 3x3 weight-gradient GEMM on another stream.  The same launch on the same input is repeated
and compared with its serial result.  usage: pk_probe.py [iterations=3000] [aggressor=wgrad|wgrad_slab|wgrad_lin|conv_fwd|mfma|copy|tr|tr_mfma|b128_mfma|none]
Round 6: the AGGRESSOR swapped - wgrad = the ping-pong 3x3 weight gradient with atomic adds (LDS-DMA + transposed LDS reads
+ MFMA + fp32 atomics), wgrad_slab = the same with plain slab stores, conv_fwd = the forward ping-pong 3x3 GEMM (LDS-DMA + ds_read_b128 + MFMA,
bf16 stores), mfma = crog_probe_mfma_bf16 (MFMA from registers only: no LDS, no memory), copy = crog_probe_copy (global loads / stores only).
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


ws = torch.empty(sk * C * 9 * C, device="cuda", dtype=torch.float32) if AGG == "wgrad_slab" else None
yf = torch.empty(rows, C, device="cuda", dtype=torch.bfloat16)
wf = (torch.randn(C, 9 * C, device="cuda") * 0.02).to(torch.bfloat16)
sink = torch.empty(512 * 256, device="cuda", dtype=torch.float32)
cp_src, cp_dst = torch.empty(1 << 26, device="cuda", dtype=torch.uint8), torch.empty(1 << 26, device="cuda", dtype=torch.uint8)


linA = (torch.randn(rows, 1024, device="cuda") * 0.1).to(torch.bfloat16) if AGG == "wgrad_lin" else None
linB = (torch.randn(rows, 1024, device="cuda") * 0.1).to(torch.bfloat16) if AGG == "wgrad_lin" else None
linG = torch.zeros(1024, 1024, device="cuda") if AGG == "wgrad_lin" else None
PKLIB = None
if AGG in ("tr", "tr_mfma", "b128_mfma"):
    import ctypes
    PKLIB = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pk_min.so"))
    PKLIB.pk_aggressor_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]


def aggressor():
    side.wait_stream(torch.cuda.current_stream())
    K.set_stream_override(side.cuda_stream)
    try:      # dW[512][9 x 512] += dy_slice^T x im2col(x): A = the SECOND channel slice of the concat gradient (row stride 3C, offset C)
        if AGG == "wgrad":
            K.gemm(K.BF16, K.A_MC, K.B_NC_IM2COL, cat, x, G, C, 9 * C, rows, 3 * C, C, 9 * C, a_off=C, conv=(2 * H, 2 * W, C), splitk=sk,
                   out_mode=K.OUT_F32_ATOMIC)
        elif AGG == "wgrad_slab":
            K.gemm(K.BF16, K.A_MC, K.B_NC_IM2COL, cat, x, ws, C, 9 * C, rows, 3 * C, C, 9 * C, a_off=C, conv=(2 * H, 2 * W, C), splitk=sk, out_mode=K.OUT_F32)
        elif AGG == "conv_fwd":
            for _ in range(2):
                K.gemm(K.BF16, K.A_IM2COL, K.B_KC, x, wf, yf, rows, C, 9 * C, C, 9 * C, C, conv=(2 * H, 2 * W, C))
        elif AGG == "mfma":
            K.check(K.lib().crog_probe_mfma_bf16(sink.data_ptr(), 512, 1500, side.cuda_stream), "probe_mfma")
        elif AGG in ("tr", "tr_mfma", "b128_mfma"):
            assert PKLIB.pk_aggressor_launch({"tr": 4, "tr_mfma": 5, "b128_mfma": 6}[AGG], 256, 4000, sink.data_ptr(), side.cuda_stream) == 0
        elif AGG == "wgrad_lin":      # the ping-pong weight-gradient kernel in its DENSE form (transposed LDS reads, no im2col addressing)
            K.gemm(K.BF16, K.A_MC, K.B_NC, linA, linB, linG, 1024, 1024, rows, 1024, 1024, 1024, splitk=4, out_mode=K.OUT_F32_ATOMIC)
        elif AGG == "copy":
            K.check(K.lib().crog_probe_copy(cp_src.data_ptr(), cp_dst.data_ptr(), 1 << 26, 0, side.cuda_stream), "probe_copy")
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

# ---- the SYNTHETIC victim of scripts/pk_min.hip (one chain of v_pk_fma_f32 / v_pk_mul_f32 + v_pk_add_f32 per lane against the same chain in
# scalar instructions, compared by a second kernel) beside the same GEMM: is a bare packed instruction enough, or does it take the real kernel?
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pk_min.so")
if AGG == "wgrad" and os.path.exists(so) and os.environ.get("PK_SYNTHETIC", "1") == "1":
    import ctypes
    pk = ctypes.CDLL(so)
    pk.pk_victim_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    pk.pk_compare_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    n = 1 << 22
    src = torch.rand(n + 2048, device="cuda") - 0.5
    names = ["v_pk_fma_f32, VGPR pairs", "v_pk_fma_f32 op_sel_hi:[0,1,1]", "v_pk_mul_f32 by an SGPR pair + v_pk_add_f32", "v_pk_fma_f32 as the first reader of just-loaded registers"]
    for blocks in (112, 1024):
        out = torch.empty(blocks * 256 * 4, device="cuda")
        for fl in range(4):
            errs, detail = torch.zeros(16, device="cuda", dtype=torch.int32), torch.zeros(64, device="cuda", dtype=torch.int32)
            main = torch.cuda.current_stream().cuda_stream
            for it in range(min(N, 1000)):
                aggressor()
                for _ in range(4):
                    assert pk.pk_victim_launch(fl, blocks, 256 if fl < 3 else 48, src.data_ptr(), n, out.data_ptr(), main) == 0
                    assert pk.pk_compare_launch(out.data_ptr(), blocks * 256, errs.data_ptr(), detail.data_ptr(), main) == 0
                torch.cuda.synchronize()
            e = int(errs[0])
            print(f"synthetic victim ({names[fl]}, {blocks} blocks) beside the GEMM: {e} of {min(N, 1000) * 4 * blocks * 256} lane results packed != scalar", flush=True)
            if e:
                d = detail.cpu().tolist()
                print("    first: lane", d[0], "packed", [hex(d[1] & 0xffffffff), hex(d[3] & 0xffffffff)], "scalar", [hex(d[2] & 0xffffffff), hex(d[4] & 0xffffffff)])
