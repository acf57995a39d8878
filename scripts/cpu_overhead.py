"""Host-side issue time of one training step vs GPU time (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from crog_amd.engine import train_step
from crog_amd.model import build_crog
from crog_amd.optim import FusedAdam
from crog_amd.testing import make_cfg, synthetic_batch
cfg = make_cfg(); torch.manual_seed(0)
model, groups = build_crog(cfg); model = model.cuda().prepare()
opt = FusedAdam(groups, lr=1e-4, store=model.store)
batch = synthetic_batch(32, 416, 20, 49408, seed=1, device="cuda"); model.train()
for _ in range(3): train_step(model, opt, None, batch, cfg)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    train_step(model, opt, None, batch, cfg)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host issue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
