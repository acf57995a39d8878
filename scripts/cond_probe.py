"""How ill-conditioned are the quantities two fp32 parity tests pin?  (GPU box)  The same fp32 forward + backward on the fixture's input and on
the input times (1 + 1e-7 noise) - a perturbation of the size of one fp32 rounding: how far do the pinned logits / gradient heads move?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from types import SimpleNamespace
import numpy as np, torch
def err(a, b): return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()
# ---- SSG tiny fp32 (tests/test_ssg_gpu.py)
import test_ssg_gpu as S
from crog_amd.model.ssg import build_ssg
from crog_amd.testing import seeded_state
fx, meta = S.load_case("ssg_tiny_rgb")
cfg = SimpleNamespace(**meta["cfg"])
def ssg_run(eps, seed):
    model = build_ssg(cfg)
    model.load_state_dict(seeded_state({k: tuple(v) for k, v in meta["shapes"].items()}, seed=meta["seed"]))
    model = model.cuda(); model.compute_dtype = torch.float32; model.prepare(); model.train()
    batch = S.synthetic_ssg_batch(meta["B"], cfg.img_size, cfg.with_depth, seed=1234 + meta["seed"], device="cuda")
    if eps:
        g = torch.Generator(device="cuda").manual_seed(seed)
        noise = torch.randn(batch["rgb"].shape, device="cuda", generator=g, dtype=torch.float64)
        batch["rgb"] = (batch["rgb"].double() * (1 + eps * noise)).float()      # (eps of a few fp32 ulps: some pixels move by one ulp, most not at all)
    out, raw = model(batch)
    loss = S.ssg_surrogate_loss(raw, meta["seed"]); loss.backward(); torch.cuda.synchronize()
    from crog_amd.runtime import RT
    RT.join_streams(); torch.cuda.synchronize()
    return {n: p.grad.detach().float().cpu().flatten()[:64].clone() for n, p in model.named_parameters()}, {k: raw[k].detach().float().cpu() for k in S.SSG_OUTPUTS}
g0, r0 = ssg_run(0, 0)
for s in (1, 2, 3, 4):
    g1, r1 = ssg_run(2e-7, s)
    worst = sorted(((float((g1[n] - g0[n]).abs().max()) / max(float(fx["grad::" + n].abs().max()), float(fx["grad_norms"][i]) / max(1.0, 64 ** 0.5), 1e-30), n)
                    for i, n in enumerate(meta["param_names"])), reverse=True)[:3]
    print(f"SSG tiny fp32, input * (1 + 2e-7 noise #{s}): outputs move {max(err(r1[k], r0[k]) for k in r0):.2e} (bound 1e-3 vs fixture); "
          f"gradient heads move (relative to the test's scale, bound 1e-2): {[(round(w, 4), n) for w, n in worst]}", flush=True)
vs_fix = sorted(((float((g0[n] - fx["grad::" + n]).abs().max()) / max(float(fx["grad::" + n].abs().max()), 1e-30), n) for n in meta["param_names"]), reverse=True)[:3]
print("  unperturbed vs fixture, worst three (relative to max |head|):", [(round(w, 4), n) for w, n in vs_fix])
