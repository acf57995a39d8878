"""Why does a second RCCL communicator slow the step at world size 1?  Times the forced-DDP step before / with / after an idle extra
communicator and prints the main thread's CPU affinity around its creation.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1); torch.cuda.set_device(0)
from crog_amd.engine import train_step
from crog_amd.model import build_crog
from crog_amd.optim import FusedAdam
from crog_amd.parallel import DistributedDataParallel, convert_sync_batchnorm
from crog_amd.testing import make_cfg, synthetic_batch
from crog_amd.rccl import RcclComm
cfg = make_cfg(batch_size=32); torch.manual_seed(0)
model, groups = build_crog(cfg); model = model.cuda(); model.prepare(torch.device("cuda", 0))
convert_sync_batchnorm(model, force=True)
net = DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True, force=True)
opt = FusedAdam(groups, lr=1e-4, store=model.store)
batch = synthetic_batch(32, 416, 20, 49408, seed=1, device="cuda"); net.train()
def timed(tag, n=15):
    for _ in range(3): train_step(net, opt, None, batch, cfg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): train_step(net, opt, None, batch, cfg)
    torch.cuda.synchronize(); print(f"{tag:44s} {(time.perf_counter()-t0)/n*1e3:.2f} ms/step  affinity {len(os.sched_getaffinity(0))} cpus  threads {len(os.listdir('/proc/self/task'))}", flush=True)
timed("forced DDP, one communicator")
aff = os.sched_getaffinity(0)
c = RcclComm(None)
print("affinity changed by ncclCommInitRank:", os.sched_getaffinity(0) != aff)
timed("+ idle extra communicator")
os.sched_setaffinity(0, aff)
timed("+ idle extra communicator, affinity restored")
c.close()
timed("extra communicator destroyed")
t = torch.zeros(64 << 20, device="cuda"); del t
g2 = dist.new_group(); x = torch.zeros(8, device="cuda"); dist.all_reduce(x, group=g2); torch.cuda.synchronize()
timed("+ idle torch process group (own comm)")
dist.destroy_process_group()
