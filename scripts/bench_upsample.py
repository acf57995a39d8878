"""Bilinear x2 upsampling forward / backward at the step's shapes (projector: 26 -> 52 -> 104 at 512 channels; neck: 13 -> 26), per-output
kernels (CROG_UPSAMPLE_OLD=1) against the patch / pair kernels (GPU box; run once per setting)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
dt = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for B, H, C in [(32, 52, 512), (32, 26, 512), (32, 13, 1024), (32, 13, 512)]:
    nset = 3
    xs = [torch.randn(B, H, H, C, device="cuda").to(dt) for _ in range(nset)]
    ys = [torch.empty(B, 2 * H, 2 * H, C, device="cuda", dtype=dt) for _ in range(nset)]
    it = [0]
    def fwd():
        i = it[0] = (it[0] + 1) % nset
        K.upsample2_fwd(xs[i], ys[i])
    def bwd():
        i = it[0] = (it[0] + 1) % nset
        K.upsample2_bwd(ys[i], xs[i])
    byt = B * H * H * C * 2 * 5
    print(f"{B} x {H} x {H} x {C}: fwd {timeit(fwd):6.1f} us, bwd {timeit(bwd):6.1f} us  (bytes at 5 TB/s {byt/5e6:5.1f} us)", flush=True)
