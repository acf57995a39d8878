"""Which hardware queue do the communicators' streams land on in the bench's own set-up order?  World size 1, DDP + SyncBN forced:
a collective on the statistics group is timed while a long kernel runs on the main / weight-gradient / text stream or a long
collective on the gradient group.  Blocked = same in-order hardware queue.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
os.environ["CROG_SYNCBN_OWN_GROUP"] = "1"
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1); torch.cuda.set_device(0)
from crog_amd.runtime import RT
from crog_amd.engine import train_step
from crog_amd.model import build_crog
from crog_amd.optim import FusedAdam
from crog_amd.parallel import DistributedDataParallel, convert_sync_batchnorm
from crog_amd.testing import make_cfg, synthetic_batch
cfg = make_cfg(batch_size=8); torch.manual_seed(0)
model, groups = build_crog(cfg); model = model.cuda(); model.prepare(torch.device("cuda", 0))
convert_sync_batchnorm(model, force=True)
net = DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True, force=True)
opt = FusedAdam(groups, lr=1e-4, store=model.store)
batch = synthetic_batch(8, 416, 20, 49408, seed=1, device="cuda"); net.train()
for _ in range(2): train_step(net, opt, None, batch, cfg)
torch.cuda.synchronize()
big = torch.empty(1 << 28, device="cuda"); big2 = torch.empty_like(big); small = torch.zeros(512, device="cuda")
gathered = [torch.empty(1 << 28, device="cuda")]
main = torch.cuda.current_stream()
names = {"main": main, "wgrad": RT._wgrad_stream[0], "text": RT.text_stream}
def long_kernel(s):
    with torch.cuda.stream(s):
        for _ in range(4): big2.copy_(big)
def probe(blocker):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(main)
    if blocker == "grad-group collective":
        w = dist.all_gather(gathered, big, async_op=True)      # 1 GiB copy on the gradient group's stream
    else:
        long_kernel(names[blocker])
        with torch.cuda.stream(names[blocker]): e2.record()
    side = torch.cuda.Stream() if False else None
    # the statistics collective is issued from a stream that is NOT the blocker: use a fresh event-only path
    issue = names["text"] if blocker == "main" else main
    with torch.cuda.stream(issue):
        RT.comm.all_reduce_sum(small)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)
base = probe("text") * 0   # warm
for b in ("main", "wgrad", "text", "grad-group collective"):
    t = min(probe(b) for _ in range(3))
    print(f"statistics collective while a long kernel runs on {b:24s}: done after {t*1e3:8.1f} us  ({'BLOCKED: same hardware queue' if t > 1.0 else 'free'})")
dist.destroy_process_group()
