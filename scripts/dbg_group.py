import os, sys
sys.path.insert(0, os.getcwd())
os.environ["CROG_DBG_GROUP"]="1"
import torch
from crog_amd.engine import train_step
from crog_amd.model import build_crog
from crog_amd.optim import FusedAdam
from crog_amd.testing import make_cfg, synthetic_batch
cfg = make_cfg(); torch.manual_seed(0)
model, groups = build_crog(cfg); model = model.cuda().prepare()
opt = FusedAdam(groups, lr=1e-4, store=model.store)
batch = synthetic_batch(32, 416, 20, 49408, seed=1, device="cuda"); model.train()
for i in range(3):
    print("step", i, flush=True)
    train_step(model, opt, None, batch, cfg)
torch.cuda.synchronize()
print([ (c.off, c.numel) for c in opt._chunks][:6])
