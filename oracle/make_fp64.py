"""Exact-arithmetic yardstick for the config-1 fixtures: the CPU oracle evaluated in float64 on the same seeded weights and inputs
(test infrastructure; run here or anywhere, it does not need the reference):   python oracle/make_fp64.py crog_r50_b2_damped
Writes tests/golden/<case>_fp64.npz with the five logit maps.  Tells how far the REFERENCE's own fp32 result sits from the exact
value, which is the floor any fp32 implementation can be asked to meet."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from crog_amd.testing import make_cfg, seeded_state, synthetic_batch  # noqa: E402
from oracle import crog_oracle as O  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "crog_r50_b2_damped"
    meta = json.load(open(os.path.join(GOLD, case + ".json")))
    g = np.load(os.path.join(GOLD, case + ".npz"))
    cfg = make_cfg(dropout=0.0)
    P = seeded_state({k: tuple(v) for k, v in meta["shapes"].items()}, seed=meta["seed"], residual_gain=meta.get("residual_gain", 1.0))
    P = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    b = synthetic_batch(meta["B"], cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=1234 + meta["seed"])
    torch.set_num_threads(8)
    with torch.no_grad():
        out = O.crog_forward(P, b["img"].double(), b["word"], [b[k].double() for k in ("mask", "qua", "sin", "cos", "wid")], num_head=cfg.num_head)
    res = {}
    for i, nm in enumerate(["ins", "qua", "sin", "cos", "wid"]):
        t = out["preds"][i]
        res["pred_" + nm] = t.numpy()
        r = torch.from_numpy(g["pred_" + nm]).double()
        print(f"{nm}: reference fp32 vs float64: max {float((r - t).abs().max()):.3e}  rms {float((r - t).pow(2).mean().sqrt()):.3e}  |logit| max {float(t.abs().max()):.2f}")
    res["loss_total"] = out["total"].numpy()
    print("loss: reference fp32", float(g["loss_total"]), "float64", float(out["total"]))
    np.savez_compressed(os.path.join(GOLD, case + "_fp64.npz"), **res)


if __name__ == "__main__":
    main()
