"""The text tower's GEMMs (640 tokens = 32 sentences x 20 positions, width 512: 12 layers x {qkv, out, fc, proj}, forward and data gradient)
on the 64 x 64 LDS-DMA tile, stand-alone: microseconds per launch.  A/B builds: scripts/build_variant.py NAME -DCROG_DMA64_NSTAGE=n
(select with CROG_LIB).  GPU box: python scripts/text_gemm_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
dt = torch.bfloat16
M = int(os.environ.get("PROBE_M", "640"))
for N, Kd in ((512, 2048), (512, 512), (2048, 512), (512, 1536), (1536, 512)):
    nset = 24                                        # distinct weights per launch, as the twelve layers have
    xs = [torch.randn(M, Kd, device="cuda").to(dt) for _ in range(nset)]
    ws = [(torch.randn(N, Kd, device="cuda") * 0.05).to(dt) for _ in range(nset)]
    bs = [torch.randn(N, device="cuda") for _ in range(nset)]
    y = torch.empty(M, N, device="cuda", dtype=dt)
    def run(i):
        K.gemm(K.BF16, K.A_KC, K.B_KC, xs[i], ws[i], y, M, N, Kd, Kd, Kd, N, bias=bs[i])
    for i in range(nset): run(i)
    torch.cuda.synchronize()
    ref = (xs[3].float() @ ws[3].float().t() + bs[3])
    run(3); torch.cuda.synchronize()
    err = float((y.float() - ref).norm() / ref.norm())
    ts = []
    for rnd in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(nset): run(i)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / nset)
    t = sorted(ts)[2]
    print(f"[{os.environ.get('CROG_LIB', 'default')}] {M} x {N} x {Kd}: {t * 1e3:6.1f} us per launch back to back ({2.0 * M * N * Kd / t / 1e9:5.1f} TFLOP/s), rel err {err:.1e}", flush=True)
