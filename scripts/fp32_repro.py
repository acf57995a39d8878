"""Is the fp32 forward of config 1 (B = 2, undamped weights) the same bits run after run, in one process and across processes?  (GPU box)
usage: fp32_repro.py [passes=6]   prints one checksum per pass; run it twice to compare processes."""
import os, sys, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_model_gpu import build, batch_for, load_case, NAMES
from crog_amd.testing import make_cfg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
g, meta = load_case("crog_r50_b2")
cfg = make_cfg(dropout=0.0)
model, _ = build(cfg, meta)
b = batch_for(cfg, meta)
model.train()
sd = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
first = None
for i in range(N):
    model.load_state_dict({**model.state_dict(), **sd})
    with torch.no_grad():
        preds, tgts, loss, _ = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
    torch.cuda.synchronize()
    h = hashlib.sha1(b"".join(p.float().cpu().numpy().tobytes() for p in preds)).hexdigest()[:12]
    e = max(float((preds[j].float().cpu() - g["pred_" + nm]).abs().max()) for j, nm in enumerate(NAMES))
    print(f"pass {i}: logits sha1 {h}  max |d| to the fixture {e:.3e}  loss {float(loss):.7f}", flush=True)
