"""Spread of tests/test_fulldepth_gpu.py's bf16 metrics (GPU box): one deterministic pass + N default-mode passes of the same step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_fulldepth_gpu as T
from crog_amd.model import build_crog
from crog_amd.runtime import set_deterministic
from crog_amd.testing import make_cfg, seeded_state, synthetic_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
case = "crog_r50_b4_damped"
g32, meta = T.load_case(case)
names = meta["param_names"]
cfg = make_cfg(dropout=0.0)
model, _ = build_crog(cfg)
model.load_state_dict(seeded_state({k: tuple(v) for k, v in meta["shapes"].items()}, seed=meta["seed"], residual_gain=meta["residual_gain"]))
model = model.cuda(); model.compute_dtype = torch.bfloat16; model.prepare().train()
b = {k: v.cuda() for k, v in synthetic_batch(meta["B"], cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=1234 + meta["seed"]).items()}
params = dict(model.named_parameters())
gbf = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(T.GOLD, case + "_bf16ref.npz")).items()}
refd = T.bf16_distances([gbf["pred_" + nm] for nm in T.NAMES], gbf["loss_total"], {n: gbf["grad_norms"][i] for i, n in enumerate(names) if gbf["grad_norms"][i] >= 0}, g32, names)
def one():
    model._store.g_clean = False
    model._store.zero_grad()
    preds, tgts, loss, _ = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
    loss.backward(); torch.cuda.synchronize()
    return T.bf16_distances(preds, loss.detach(), {n: params[n].grad.float().norm() for n in names if params[n].grad is not None}, g32, names)
keys = ["logit_rms", "loss"] + [k for k in refd if k.startswith("gnorm_med")]
print("reference bf16:", {k: round(refd[k], 4) for k in keys})
set_deterministic(True); d = one(); set_deterministic(False)
print("deterministic :", {k: round(d[k], 4) for k in keys})
for i in range(N):
    d = one()
    print(f"default run {i} :", {k: round(d[k], 4) for k in keys}, flush=True)
