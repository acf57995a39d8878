"""cProfile of the host side of one training step (GPU box)."""
import os, sys, cProfile, pstats
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
torch.autograd.set_multithreading_enabled(False)   # run backward nodes on this thread so cProfile sees them
pr = cProfile.Profile(); pr.enable()
for _ in range(3): train_step(model, opt, None, batch, cfg)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
