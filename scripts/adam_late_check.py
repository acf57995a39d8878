"""Does the overlapped Adam (FusedAdam.overlap_backward) leave the same parameters as the update done in step()?  Deterministic mode, full-depth
CROG-R50 bf16, B = 4, three steps; prints a checksum of P, m, v.  Run under CROG_ADAM_OVERLAP=0 / CROG_ADAM_LATE=1 / CROG_ADAM_LATE=0 and compare."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd.engine import train_step
from crog_amd.model import build_crog
from crog_amd.optim import FusedAdam
from crog_amd.runtime import RT, set_deterministic
from crog_amd.testing import make_cfg, synthetic_batch
set_deterministic(True)
torch.manual_seed(0)
cfg = make_cfg(dropout=0.1)
model, groups = build_crog(cfg); model = model.cuda().prepare(); model.train()
opt = FusedAdam(groups, lr=1e-4, store=model.store)
b = synthetic_batch(4, 416, 20, 49408, seed=3, device="cuda")
RT.manual_seed(5)
for i in range(4):
    train_step(model, opt, None, b, cfg)
torch.cuda.synchronize()
h = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]
print("P", h(model.store.P), "m", h(opt.m), "v", h(opt.v), "early launches", opt.early_launches, flush=True)
