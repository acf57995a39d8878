"""Which Python call sites issue torch copy / add / fill kernels during one training step (GPU box)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from crog_amd.engine import train_step
from crog_amd.model import build_crog
from crog_amd.optim import FusedAdam
from crog_amd.testing import make_cfg, synthetic_batch
cfg = make_cfg(); torch.manual_seed(0)
model, groups = build_crog(cfg); model = model.cuda().prepare()
opt = FusedAdam(groups, lr=1e-4, store=model.store)
batch = synthetic_batch(32, 416, 20, 49408, seed=1, device="cuda"); model.train()
for _ in range(2): train_step(model, opt, None, batch, cfg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    train_step(model, opt, None, batch, cfg); torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::contiguous", "aten::clone", "aten::mul", "aten::div", "aten::sum", "aten::cat", "aten::stack", "aten::to", "aten::_to_copy"):
        st = [s for s in ev.stack if "crog_amd" in s or "bench" in s or "engine" in s][:2]
        shp = str(ev.input_shapes)[:60]
        agg[(ev.name, " <- ".join(s.split("/")[-1] for s in st), shp)] += 1
for (n, st, shp), c in agg.most_common(60):
    print(f"{c:4d}  {n:18s} {shp:60s} {st}")

print("---- all CPU op names")
names = collections.Counter(ev.name for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CPU)
print(names.most_common(45))
print("---- GPU activities named like copies, with the CPU op that launched them")
byop = collections.Counter()
for ev in prof.events():
    for k in getattr(ev, "kernels", []) or []:
        if "copy" in k.name.lower() or "memcpy" in k.name.lower():
            st = [s for s in ev.stack if "crog_amd" in s or "engine" in s][:3]
            byop[(ev.name, k.name[:40], " <- ".join(s.split("/")[-1] for s in st))] += 1
for k, c in byop.most_common(30): print(c, k)
