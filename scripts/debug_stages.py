"""Stage-wise forward errors of the tiny model vs the reference fixture (debug aid; GPU box). argv[1]: f32|bf16"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_model_gpu import load_case, build, batch_for, err, nchw
from crog_amd.testing import tiny_cfg
dt = torch.bfloat16 if sys.argv[1:] == ["bf16"] else torch.float32
g, meta = load_case("tiny_crog"); cfg = tiny_cfg()
model, _ = build(cfg, meta, dtype=dt); b = batch_for(cfg, meta); model.train()
def stat(name, a, ref):
    a = a.detach().float().cpu(); d = (a - ref).abs()
    print(f"{name:10s} max {d.max():.3e} mean {d.mean():.3e} ref_rms {ref.pow(2).mean().sqrt():.3e} frac>0.5 {(d>0.5).float().mean():.2e} argmax {np.unravel_index(int(d.argmax()), d.shape)}")
V = model.backbone.visual
x = b["img"]
from crog_amd import functional as Fn
x2, x3, x4 = model.backbone.image_features(b["img"], dt)
stat("x2", nchw(x2), g["x2"]); stat("x3", nchw(x3), g["x3"]); stat("x4", nchw(x4), g["x4"])
wfeat, state = model.backbone.text_features(b["word"], dt)
stat("word_feat", wfeat, g["word_feat"]); stat("state", state, g["state"])
fq = model.neck((x2, x3, x4), state); stat("fq", nchw(fq), g["fq"])
fqd = model.decoder(fq, wfeat, (b["word"] == 0).contiguous()); stat("fq_dec", nchw(fqd).reshape(g["fq_dec"].shape), g["fq_dec"])
pred = model.proj(fqd, state)
for i, nm in enumerate(["ins", "qua", "sin", "cos", "wid"]):
    stat("pred_" + nm, pred[:, i:i+1], g["pred_" + nm])
