"""Which parameter gradient goes non-finite first (GPU box): python scripts/nan_hunt.py [batch] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from crog_amd.model import build_crog
from crog_amd.optim import FusedAdam
from crog_amd.testing import make_cfg, synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = make_cfg(); torch.manual_seed(0)
model, groups = build_crog(cfg); model = model.cuda().prepare(); model.train()
opt = FusedAdam(groups, lr=1e-4, store=model.store)
b = synthetic_batch(B, 416, 20, 49408, seed=1234, device="cuda")
for step in range(steps):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred, tgt, loss, _ = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
    opt.zero_grad(); loss.backward(); torch.cuda.synchronize()
    bad = [(n, int((~torch.isfinite(g)).sum()), float(g[torch.isfinite(g)].abs().max()) if torch.isfinite(g).any() else -1) for n, p, o, k, g in model.store.entries if not torch.isfinite(g).all()]
    big = sorted(((float(g.abs().max()), n) for n, p, o, k, g in model.store.entries if torch.isfinite(g).all()), reverse=True)[:3]
    print(f"step {step}: loss {float(loss):.4f}  non-finite grads in {len(bad)} tensors {bad[:6]}  largest finite {big}", flush=True)
    if bad: break
    opt.step()
