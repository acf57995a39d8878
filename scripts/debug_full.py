"""Config-1 (CROG-R50, B=2, 416^2) logits: HIP fp32 / bf16 vs reference fp32 fixture and fp64 truth (GPU box)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_model_gpu import load_case, build, batch_for
from crog_amd.testing import make_cfg
g, meta = load_case("crog_r50_b2"); cfg = make_cfg(dropout=0.0)
t64 = np.load(os.path.join(ROOT, "tests/golden/crog_r50_b2_fp64.npz"))
for dt in (torch.float32, torch.bfloat16):
    model, _ = build(cfg, meta, dtype=dt); b = batch_for(cfg, meta); model.train()
    preds, tgts, loss, ld = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
    torch.cuda.synchronize()
    print(dt, "loss", float(loss.detach()), "ref", float(g["loss_total"]))
    for i, nm in enumerate(["ins", "qua", "sin", "cos", "wid"]):
        p = preds[i].double().cpu(); t = torch.from_numpy(t64["pred_" + nm]); r = g["pred_" + nm].double()
        print(f"  {nm}: vs fp64 max {float((p-t).abs().max()):.3e} rms {float((p-t).pow(2).mean().sqrt()):.3e} | vs ref32 max {float((p-r).abs().max()):.3e} | ref32 vs fp64 max {float((r-t).abs().max()):.3e} rms {float((r-t).pow(2).mean().sqrt()):.3e} | pred rms {float(t.pow(2).mean().sqrt()):.2f}")
    del model
