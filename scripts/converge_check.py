"""Sanity: 60 optimizer steps on ONE fixed synthetic batch (bf16 autocast, FusedAdam): the loss must fall steadily.  Catches stale
bf16 shadow weights / gradient-buffer bookkeeping errors that single-step parity tests cannot see.  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crog_amd.engine import train_step
from crog_amd.model import build_crog
from crog_amd.optim import FusedAdam
from crog_amd.testing import make_cfg, synthetic_batch
cfg = make_cfg(); torch.manual_seed(0)
model, groups = build_crog(cfg); model = model.cuda().prepare(); model.train()
opt = FusedAdam(groups, lr=1e-4, store=model.store)
batch = synthetic_batch(32, 416, 20, 49408, seed=1, device="cuda")
losses = []
for i in range(60):
    stats, _ = train_step(model, opt, None, batch, cfg)
    if i % 10 == 0 or i == 59: losses.append(round(float(stats[0]), 4))
# the bf16 shadow the next forward would use must equal a fresh cast of the fp32 parameters
st = model.store
fresh = st.P.to(torch.bfloat16)
print("loss every 10 steps:", losses, "| shadow == cast(P):", bool(torch.equal(st.S, fresh)), "| peak GiB", round(torch.cuda.max_memory_allocated() / 2**30, 1))
assert losses[-1] < 0.2 * losses[0] and torch.equal(st.S, fresh)
