"""Fused Adam over CROG-R50's 147 M parameters (GPU box): time and TB/s (16 B read + 14 B written per parameter).  CROG_LIB selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
n = 147_112_290 // 4 * 4
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda") * 1e-3; m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
sh = torch.empty(n, device="cuda", dtype=torch.bfloat16)
pr = p.clone(); mr = m.clone(); vr = v.clone()
K.adam_step(p, g, m, v, n, 1e-3, 0.9, 0.999, 1e-8, 1e-4, 1, shadow=sh); torch.cuda.synchronize()
opt_p = torch.nn.Parameter(pr.clone()); opt_p.grad = g.clone()
opt = torch.optim.Adam([opt_p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4); opt.step()
print("max |p - torch.optim.Adam| after one step:", float((p - opt_p.data).abs().max()), " shadow ok:", bool((sh.float() - p).abs().max() < 0.02 * p.abs().max()))
del opt, opt_p, pr, mr, vr
ts = []
for r in range(5):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(10): K.adam_step(p, g, m, v, n, 1e-3, 0.9, 0.999, 1e-8, 1e-4, 2 + i, shadow=sh)
    e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) / 10)
t = sorted(ts)[2]
print(f"adam over {n/1e6:.1f} M parameters: {t*1e3:.1f} us = {n*30/t/1e9:.2f} TB/s")
