"""bf16 vs fp32 HIP paths on better-conditioned batches (GPU box)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from crog_amd.testing import seeded_state, synthetic_batch, tiny_cfg, make_cfg
from crog_amd.model import build_crog
def run(cfg, B, size, seed=3, gain=1.0):
    cfg.input_size = size
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        torch.manual_seed(0)
        model, _ = build_crog(cfg)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(seeded_state(shapes, seed=seed, residual_gain=gain))
        model = model.cuda(); model.compute_dtype = dt; model.prepare(); model.train()
        b = {k: v.cuda() for k, v in synthetic_batch(B, size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=77).items()}
        x2, x3, x4 = model.backbone.image_features(b["img"], dt)
        wfeat, state = model.backbone.text_features(b["word"], dt)
        fq = model.neck((x2, x3, x4), state)
        fqd = model.decoder(fq, wfeat, (b["word"] == 0).contiguous())
        pred = model.proj(fqd, state)
        res[dt] = dict(x2=x2.float(), x3=x3.float(), x4=x4.float(), wfeat=wfeat.float(), state=state.float(), fq=fq.float(), fqd=fqd.float(), pred=pred.float())
        del model
    print(f"B={B} size={size} gain={gain}: " + "  ".join(f"{k} {float((res[torch.bfloat16][k]-res[torch.float32][k]).pow(2).mean().sqrt() / res[torch.float32][k].pow(2).mean().sqrt()):.3f}" for k in res[torch.float32]))
run(tiny_cfg(), 16, 160, gain=0.25)
run(make_cfg(dropout=0.0), 8, 416, seed=5, gain=0.25)
run(make_cfg(dropout=0.0), 8, 416, seed=5, gain=0.0)
