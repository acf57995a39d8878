"""Does any kernel of the training step read memory nobody wrote?  (GPU box)  torch.empty / empty_like are patched to hand out NaN-filled
tensors (one stream, no forks: the fills are ordered like the allocations); a NaN in the loss or in a gradient names the reader.
usage: poison_probe.py [B=8] [dropout=0.1] [det=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd.model import build_crog
from crog_amd.runtime import RT, set_deterministic
from crog_amd.testing import make_cfg, synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
det = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
if det:
    set_deterministic(True)
RT.overlap_wgrad = False
torch.manual_seed(0)
cfg = make_cfg(dropout=p)
model, _ = build_crog(cfg); model = model.cuda().prepare(); model.train()
model.overlap_text = False
b = {k: v.cuda() for k, v in synthetic_batch(B, 416, cfg.word_len, cfg.clip_arch["vocab_size"], seed=9).items()}
_empty, _empty_like = torch.empty, torch.empty_like
POISON = float(os.environ.get("POISON", "nan"))
def empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_cuda and t.is_floating_point():
        t.fill_(POISON)
    elif t.is_cuda and t.dtype == torch.uint8:
        t.fill_(0xAA)
    return t
def empty_like(x, **k):
    t = _empty_like(x, **k)
    if t.is_cuda and t.is_floating_point():
        t.fill_(POISON)
    return t
def run(poison):
    torch.empty, torch.empty_like = (empty, empty_like) if poison else (_empty, _empty_like)
    try:
        RT.manual_seed(5)
        model.store.g_clean = False
        model.store.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            _, _, loss, _ = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
        loss.backward()
        torch.cuda.synchronize()
    finally:
        torch.empty, torch.empty_like = _empty, _empty_like
    return float(loss.detach()), model.store.G.clone()
l0, g0 = run(False)
l1, g1 = run(True)
print(f"deterministic={det} B={B}: loss plain {l0:.6f}, poisoned allocations {l1:.6f}")
bad = []
for n, p_, o, k, _ in model.store.entries:
    a, c = g1[o:o + k], g0[o:o + k]
    if not torch.isfinite(a).all() or not torch.equal(a, c):
        bad.append((n, int((~torch.isfinite(a)).sum()), int((a != c).sum()), k))
print(f"{len(bad)} parameters whose gradient changed under poisoned allocations (name, non-finite, differing, size):")
for x in bad[:40]:
    print("   ", x)
