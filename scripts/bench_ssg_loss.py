"""Row N4 timing (GPU box): the SSG loss on B = 64 images at ssg_r50.yaml's sizes — the per-image loop (the oracle's restatement of
the reference's compute_loss, run on the GPU as the reference would) vs the batched device implementation (crog_amd/ssg_loss.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from crog_amd.ssg_loss import ssg_loss as batched
from crog_amd.testing import ssg_cfg, synthetic_ssg_predictions, synthetic_ssg_targets
from oracle.ssg_loss_oracle import ssg_loss as looped
from oracle import ssg_oracle as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = ssg_cfg()
anchors = torch.tensor(S.anchors(cfg.aspect_ratios, cfg.img_size, cfg.anchor_strides)).reshape(-1, 4).cuda()
raw = {k: v.requires_grad_(True) for k, v in synthetic_ssg_predictions(B, anchors.shape[0], cfg, 3, device="cuda").items()}
tg = synthetic_ssg_targets(B, cfg.img_size, cfg.num_classes, seed=5, device="cuda")
def run(fn, n=3):
    for i in range(n + 1):
        if i == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for v in raw.values(): v.grad = None
        losses = fn(cfg, anchors, raw, tg, {})
        sum(losses.values()).backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, {k: float(v) for k, v in losses.items()}
tl, ll = run(looped)
tb, lb = run(batched)
print(f"B={B}: per-image loop {tl:.1f} ms, batched {tb:.1f} ms ({tl / tb:.1f}x); max loss difference {max(abs(ll[k] - lb[k]) for k in ll):.2e}")
