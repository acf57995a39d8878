"""Deterministic mode, full-depth CROG-R50 bf16: N forward + backward passes from the same weights, how many differ from the first and in
which parameters first (store order).  usage: det_stress.py [B=8] [dropout=0.1] [N=8]   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd.model import build_crog
from crog_amd.runtime import RT, set_deterministic
from crog_amd.testing import make_cfg, synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
N = int(sys.argv[3]) if len(sys.argv) > 3 else 8
set_deterministic(True)
# variants (round 5): DET_VARIANT = all | notext | nowgrad | nofork:conv,linear,mha,ln   (which concurrency is on beside the main stream)
V = os.environ.get("DET_VARIANT", "all")
RT.det_streams = os.environ.get("DET_STREAMS", "all")
if V.startswith("nofork:"):
    RT.no_fork = set(V.split(":", 1)[1].split(","))
elif V == "nowgrad":
    RT.overlap_wgrad = False
elif V.startswith("range:"):      # fork only the conv weight gradients number lo .. hi - 1 of the backward pass (in execution order)
    lo, hi = V.split(":", 1)[1].split("-")
    RT.fork_range = (int(lo), int(hi))
    RT.fork_dummy = os.environ.get("DET_DUMMY", "")
torch.manual_seed(0)
cfg = make_cfg(dropout=p)
model, _ = build_crog(cfg); model = model.cuda().prepare(); model.train()
if V == "notext":
    model.overlap_text = False
b = {k: v.cuda() for k, v in synthetic_batch(B, 416, cfg.word_len, cfg.clip_arch["vocab_size"], seed=9).items()}
sd = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
DUMP = [int(v) for v in os.environ.get("DET_DUMP", "").split(",") if v]      # conv-backward call numbers (1-based) whose tensors are cloned and compared
RT._dbg_ref = os.environ.get("DET_DUMP_REF") == "1"      # references instead of clones: nothing is added to the step (a few copies after the fork hide the effect)
def grads():
    RT._dbg_dump, RT._dbg_dump_at = ([], set(DUMP)) if DUMP else (None, set())
    model.load_state_dict({**model.state_dict(), **sd})
    RT.manual_seed(5)
    model.store.g_clean = False
    model.store.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        _, _, loss, _ = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), model.store.G.clone()
def dumped():
    return {k: v for k, v in (RT._dbg_dump or [])}
l0, g0 = grads()
d0 = dumped()
bad_runs = 0
for i in range(1, N):
    l, g = grads()
    diff = g != g0
    if l != l0 or diff.any():
        bad_runs += 1
        names = [n for n, p_, o, k, _ in model.store.entries if bool(diff[o:o + k].any())]
        groups = {}
        for n in names:
            k = ".".join(n.split(".")[:3]) if n.startswith("backbone.visual") else n.split(".")[0] + "." + n.split(".")[1]
            groups[k] = groups.get(k, 0) + 1
        print(f"run {i}: loss equal {l == l0}; {int(diff.sum())} elements in {len(names)} parameters differ; by module: {groups}")
        for k, t in sorted(dumped().items()):
            if i == 1:
                print(f"    (conv backward #{k}: " + ", ".join(f"{nm} {tuple(v.shape)}" for nm, v in t.items() if v is not None) + ")")
            for nm in t:
                if t[nm] is not None and k in d0 and not torch.equal(t[nm], d0[k][nm]):
                    a, c = t[nm].reshape(-1, t[nm].shape[-1]), d0[k][nm].reshape(-1, t[nm].shape[-1])
                    rows = (a != c).any(1).nonzero().flatten()
                    cols = (a != c).any(0).nonzero().flatten()
                    print(f"    conv backward #{k}: {nm} differs: {int((a != c).sum())} elements, {rows.numel()} rows [{rows[:6].tolist()} .. {rows[-3:].tolist()}], {cols.numel()} columns [{cols[:6].tolist()} .. {cols[-3:].tolist()}]; max |d| {float((a.float() - c.float()).abs().max()):.3e} of scale {float(c.float().abs().max()):.3e}")
print(f"[{V}] {bad_runs} of {N - 1} runs differ from the first", flush=True)
