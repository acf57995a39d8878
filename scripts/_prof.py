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
for _ in range(3): train_step(model, opt, None, batch, cfg)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train_step(model, opt, None, batch, cfg)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if ("copy" in e.key.lower() or "fill" in e.key.lower() or "zero" in e.key.lower() or "Memcpy" in e.key or "Memset" in e.key)]
for e in sorted(rows, key=lambda e: -e.count)[:25]:
    print(f"{e.count:5d}  {e.key[:40]:40s} {str(e.input_shapes)[:90]}")
