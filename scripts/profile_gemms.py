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
if os.environ.get("CROG_SINGLE_STREAM") == "1":
    from crog_amd.runtime import RT
    RT.overlap_wgrad = False; model.overlap_text = False
opt = FusedAdam(groups, lr=1e-4, store=model.store)
batch = synthetic_batch(B, 416, 20, 49408, seed=1, device="cuda"); model.train()
for _ in range(2): train_step(model, opt, None, batch, cfg)
torch.cuda.synchronize()
K.PROF = dict(key=None, records=[], descs=[])
train_step(model, opt, None, batch, cfg)
torch.cuda.synchronize()
recs = K.PROF["records"]; descs = K.PROF["descs"]; K.PROF = None
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
def ideal_ms(key, fl):
    """max(MFMA time at the measured 1.9 PFLOP/s peak, operand + output bytes at the measured 4.5 TB/s copy rate); 3x3 operands count once"""
    al, bl, M, N, Kd, bt, sk = key
    a_el = M * Kd / (9 if al == 1 else 1); b_el = N * Kd / (9 if bl in (2, 3) and al != 1 else 1)
    if bl == 3: b_el = N * Kd / 9
    out = M * N * (4 if al == 2 else 2)
    byt = bt * (2 * (a_el + b_el) + out)
    return max(fl / 1.9e12, byt / 4.5e9)
gap = {k: a[1] - a[0] * ideal_ms(k, a[2]) for k, a in agg.items()}
print(f"sum of (measured - ideal) {sum(gap.values()):.2f} ms; ideal total {tot - sum(gap.values()):.2f} ms")
for key, (n, d, fl) in sorted(agg.items(), key=lambda kv: -gap[kv[0]])[:70]:
    al, bl, M, N, Kd, bt, sk = key
    print(f"{d:8.3f} ms gap {gap[key]:6.3f}  n={n:3d}  {names[(al,bl)]:12s} M={M:7d} N={N:5d} K={Kd:7d} batch={bt:5d} sk={sk:4d}  {fl*n/d/1e9:8.1f} TF/s  each {d/n*1e3:7.1f} us ideal {ideal_ms(key, fl)*1e3:6.1f}")
by = collections.defaultdict(float)
for key, (n, d, fl) in agg.items(): by[names[(key[0], key[1])]] += d
print({k: round(v, 2) for k, v in by.items()})

# replay: the same descriptors on the same memory, back to back on an idle GPU (contents are stale; timing only).  A launch that is
# much faster here than inside the step lost its time there to what surrounded it, not to its own code.
import ctypes
rep = collections.OrderedDict()
raw = K.stream()
for (e0, e1, fl, key), d in zip(recs, descs):
    if key in rep: continue
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s_.record()
    for _ in range(3): K.lib().crog_gemm(ctypes.byref(d), raw)
    e_.record(); torch.cuda.synchronize()
    rep[key] = s_.elapsed_time(e_) / 3
print("in-step vs replay (ms total over the step's launches of the shape):")
tr = 0.0
for key, (n, dsum, fl) in sorted(agg.items(), key=lambda kv: -(kv[1][1] - kv[1][0] * rep[kv[0]]))[:40]:
    al, bl, M, N, Kd, bt, sk = key
    print(f"  in-step {dsum:7.3f}  replay {n*rep[key]:7.3f}  n={n:3d} {names[(al,bl)]:12s} M={M:7d} N={N:5d} K={Kd:7d} batch={bt:4d} sk={sk:4d}  each {dsum/n*1e3:7.1f} vs {rep[key]*1e3:7.1f} us")
print(f"in-step total {tot:.2f} ms, replay total {sum(a[0]*rep[k] for k, a in agg.items()):.2f} ms")
print("standalone (replay) time minus ideal, all shapes, by gap:")
rows = sorted(((n * (rep[k] - ideal_ms(k, fl)), n, rep[k], ideal_ms(k, fl), fl, k) for k, (n, d, fl) in agg.items()), reverse=True)
acc = 0.0
for gap_, n, r, idl, fl, key in rows[:80]:
    al, bl, M, N, Kd, bt, sk = key
    acc += gap_
    print(f"  gap {gap_:6.3f} (cum {acc:6.2f})  n={n:3d} {names[(al,bl)]:12s} M={M:7d} N={N:5d} K={Kd:7d} batch={bt:4d} sk={sk:4d}  alone {r*1e3:7.1f} us ideal {idl*1e3:6.1f}  {fl/r/1e9:7.1f} TF/s")
cat = collections.defaultdict(lambda: [0.0, 0.0])
for gap_, n, r, idl, fl, key in rows:
    c = cat[names[(key[0], key[1])]]; c[0] += n * r; c[1] += n * idl
print({k: (round(v[0], 2), round(v[1], 2)) for k, v in cat.items()})
