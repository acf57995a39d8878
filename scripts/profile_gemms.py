"""Per-shape GEMM time breakdown of one CROG-R50 bf16 training step at B=32 (GPU box)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from crog_amd import kernels as K
from crog_amd.engine import train_step
from crog_amd.model import build_crog
from crog_amd.optim import FusedAdam
from crog_amd.testing import make_cfg, synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = make_cfg(); torch.manual_seed(0)
model, groups = build_crog(cfg); model = model.cuda().prepare()
if os.environ.get("CROG_NO_TEXT_OVERLAP") == "1": model.overlap_text = False
opt = FusedAdam(groups, lr=1e-4, store=model.store)
batch = synthetic_batch(B, 416, 20, 49408, seed=1, device="cuda"); model.train()
for _ in range(2): train_step(model, opt, None, batch, cfg)
torch.cuda.synchronize()
K.PROF = dict(key=None, records=[])
train_step(model, opt, None, batch, cfg)
torch.cuda.synchronize()
recs = K.PROF["records"]; K.PROF = None
agg = collections.OrderedDict()
for e0, e1, fl, key in recs:
    d = e0.elapsed_time(e1)
    a = agg.setdefault(key, [0, 0.0, fl])
    a[0] += 1; a[1] += d
tot = sum(a[1] for a in agg.values())
if os.environ.get("CROG_GEMM_DUMP"):
    import json
    json.dump([dict(key=list(k), n=a[0], ms=a[1], flops=a[2]) for k, a in agg.items()], open(os.environ["CROG_GEMM_DUMP"], "w"))
print(f"GEMM launches {len(recs)} total {tot:.2f} ms")
names = {(0,0):"fwd/NT", (1,0):"conv3 fwd", (0,1):"dgrad/NN", (1,2):"conv3 dgrad", (2,1):"wgrad/TN", (2,3):"conv3 wgrad", (2,0): "TN-kc"}
for key, (n, d, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    al, bl, M, N, Kd, bt, sk = key
    print(f"{d:8.3f} ms  n={n:3d}  {names[(al,bl)]:12s} M={M:7d} N={N:5d} K={Kd:7d} batch={bt:5d} sk={sk:4d}  {fl*n/d/1e9:8.1f} TF/s")
by = collections.defaultdict(float)
for key, (n, d, fl) in agg.items(): by[names[(key[0], key[1])]] += d
print({k: round(v, 2) for k, v in by.items()})
