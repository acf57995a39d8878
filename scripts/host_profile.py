"""Where the host time of an EAGER training step goes (GPU box): python scripts/host_profile.py [ddp|plain] [st]
`st` runs backward on the calling thread (torch.autograd.set_multithreading_enabled(False)) so that cProfile sees it."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29556")
mode = sys.argv[1] if len(sys.argv) > 1 else "ddp"
st = len(sys.argv) > 2 and sys.argv[2] == "st"
if mode == "ddp":
    dist.init_process_group("nccl", rank=0, world_size=1)
from crog_amd.engine import train_step
from crog_amd.model import build_crog
from crog_amd.optim import FusedAdam
from crog_amd.parallel import DistributedDataParallel, convert_sync_batchnorm
from crog_amd.testing import make_cfg, synthetic_batch
cfg = make_cfg(); torch.manual_seed(0)
model, groups = build_crog(cfg); model = model.cuda().prepare()
net = model
if mode == "ddp":
    convert_sync_batchnorm(model, force=True)
    net = DistributedDataParallel(model, device_ids=[0], force=True)
opt = FusedAdam(groups, lr=1e-4, store=model.store)
batch = synthetic_batch(32, 416, 20, 49408, seed=1, device="cuda"); net.train()
if st:
    torch.autograd.set_multithreading_enabled(False)
step = lambda: train_step(net, opt, None, batch, cfg)
for _ in range(6): step()
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{mode} st={int(st)}: wall {1e3*(t2-t0)/N:.2f} ms/step, host issue {1e3*(t1-t0)/N:.2f} ms/step", flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(N): step()
pr.disable(); torch.cuda.synchronize()
ps = pstats.Stats(pr); ps.sort_stats("tottime").print_stats(45)
if mode == "ddp": dist.destroy_process_group()
