"""Per-parameter gradient comparison HIP vs CPU oracle on the tiny config (debug aid; GPU box)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from crog_amd.testing import seeded_state, synthetic_batch, tiny_cfg
from crog_amd.model import build_crog
from oracle import crog_oracle as O

dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
meta = json.load(open(os.path.join(ROOT, "tests/golden/tiny_crog.json")))
cfg = tiny_cfg()
shapes = {k: tuple(v) for k, v in meta["shapes"].items()}
P = seeded_state(shapes, seed=meta["seed"])
for n in meta["param_names"]:
    P[n].requires_grad_(True)
b = synthetic_batch(meta["B"], cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=1234 + meta["seed"])
out = O.crog_forward(P, b["img"], b["word"], [b[k] for k in ("mask", "qua", "sin", "cos", "wid")], num_head=cfg.num_head)
out["total"].backward()
# fp64 truth
P64 = {k: (v.detach().double() if v.is_floating_point() else v.clone()) for k, v in seeded_state(shapes, seed=meta["seed"]).items()}
for n in meta["param_names"]:
    P64[n].requires_grad_(True)
import oracle.crog_oracle as O2
_f = torch.full
out64 = None
torch.set_default_dtype(torch.float64)
try:
    out64 = O.crog_forward(P64, b["img"].double(), b["word"], [b[k].double() for k in ("mask", "qua", "sin", "cos", "wid")], num_head=cfg.num_head)
    out64["total"].backward()
finally:
    torch.set_default_dtype(torch.float32)
model, _ = build_crog(cfg)
model.load_state_dict(seeded_state(shapes, seed=meta["seed"]))
model = model.cuda(); model.compute_dtype = dtype; model.prepare(); model.train()
bc = {k: v.cuda() for k, v in b.items()}
preds, tgts, loss, ld = model(bc["img"], bc["word"], bc["mask"], bc["qua"], bc["sin"], bc["cos"], bc["wid"])
loss.backward(); torch.cuda.synchronize()
print("loss", float(loss), float(out["total"]))
params = dict(model.named_parameters())
rows = []
for n in meta["param_names"]:
    r = P[n].grad
    if r is None: continue
    a = params[n].grad.detach().cpu().float()
    t = P64[n].grad
    rel = float((a.double() - t).norm() / (t.norm() + 1e-30))
    relc = float((r.double() - t).norm() / (t.norm() + 1e-30))
    rows.append((n, rel, relc, float(t.norm())))
print("pred err vs fp64: hip %.3e cpu32 %.3e" % (max(float((preds[i].cpu().double() - out64["preds"][i]).abs().max()) for i in range(5)),
      max(float((out["preds"][i].double() - out64["preds"][i]).abs().max()) for i in range(5))))
for n, rel, relc, nr in rows:
    flag = "  <<<<" if rel > 4 * relc + 1e-5 and nr > 1e-4 else ""
    print(f"{n:66s} hip_vs_f64 {rel:.2e}  cpu32_vs_f64 {relc:.2e}  |g| {nr:.2e}{flag}")
