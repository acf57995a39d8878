"""Host issue time and step time with the DDP wrapper forced at world size 1 (GPU box): python scripts/ddp_overhead.py [ddp|nosyncbn|dry|plain] [replay]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
dist.init_process_group("nccl", rank=0, world_size=1)
from crog_amd.engine import train_step
from crog_amd.model import build_crog
from crog_amd.optim import FusedAdam
from crog_amd.parallel import DistributedDataParallel, convert_sync_batchnorm
from crog_amd.testing import make_cfg, synthetic_batch
cfg = make_cfg(); torch.manual_seed(0)
model, groups = build_crog(cfg); model = model.cuda().prepare()
mode = sys.argv[1] if len(sys.argv) > 1 else "ddp"
net = model
if mode == "dry":
    import crog_amd.parallel as P
    P._DRY = True
if mode != "plain":
    if mode != "nosyncbn":
        convert_sync_batchnorm(model, force=True)
    net = DistributedDataParallel(model, device_ids=[0], force=True)
opt = FusedAdam(groups, lr=1e-4, store=model.store)
batch = synthetic_batch(32, 416, 20, 49408, seed=1, device="cuda"); net.train()
step = lambda: train_step(net, opt, None, batch, cfg)
if len(sys.argv) > 2 and sys.argv[2] == "replay":
    from crog_amd.graphs import GraphedTrainStep
    g = GraphedTrainStep(net, opt, cfg, torch.bfloat16)
    step = lambda: g(batch)
for _ in range(6): step()
torch.cuda.synchronize()
N = 20
c0 = time.process_time(); t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter(); c1 = time.process_time()
print(f"{mode}: wall {1e3*(t2-t0)/N:.2f} ms/step, host issue {1e3*(t1-t0)/N:.2f} ms/step, process CPU time {1e3*(c1-c0)/N:.2f} ms/step")
dist.destroy_process_group()
