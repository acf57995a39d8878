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
torch.manual_seed(0)
cfg = make_cfg(dropout=p)
model, _ = build_crog(cfg); model = model.cuda().prepare(); model.train()
b = {k: v.cuda() for k, v in synthetic_batch(B, 416, cfg.word_len, cfg.clip_arch["vocab_size"], seed=9).items()}
sd = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
def grads():
    model.load_state_dict({**model.state_dict(), **sd})
    RT.manual_seed(5)
    model.store.g_clean = False
    model.store.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        _, _, loss, _ = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), model.store.G.clone()
l0, g0 = grads()
bad_runs = 0
for i in range(1, N):
    l, g = grads()
    diff = g != g0
    if l != l0 or diff.any():
        bad_runs += 1
        names = [n for n, p_, o, k, _ in model.store.entries if bool(diff[o:o + k].any())]
        print(f"run {i}: loss equal {l == l0}; {int(diff.sum())} elements in {len(names)} parameters differ; last in store order: {names[-3:]}")
print(f"{bad_runs} of {N - 1} runs differ from the first")
