"""Per-step (loss, IoU) of the benchmark configuration for the first N steps (GPU box); run from any tree: python scripts/loss_trace.py [N]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from crog_amd.engine import train_step
from crog_amd.model import build_crog
from crog_amd.optim import FusedAdam
from crog_amd.runtime import RT
from crog_amd.testing import make_cfg, synthetic_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg = make_cfg(); torch.manual_seed(0)
model, groups = build_crog(cfg); model = model.cuda().prepare()
opt = FusedAdam(groups, lr=cfg.base_lr, store=model.store)
RT.manual_seed(1234)
batch = synthetic_batch(32, 416, 20, 49408, seed=1234, device="cuda"); model.train()
out = []
for i in range(N):
    stats, _ = train_step(model, opt, None, batch, cfg)
    out.append(stats.tolist())
print(ROOT, " ".join(f"{o[0]:.3f}" for o in out))
