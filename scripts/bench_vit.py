"""BASELINE config 4: CLIP ViT-B/16 image tower, 64 x 3 x 224 x 224, forward + backward (encoder-level), bf16. GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd.model.blocks import bind_all
from crog_amd.model.clip import VisionTransformer
from crog_amd.runtime import ParamStore, RT
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dt = torch.bfloat16
torch.manual_seed(0)
vit = VisionTransformer(224, 16, 768, 12, 12, 512)
store = ParamStore(vit, torch.device("cuda")); bind_all(vit, store); vit.train()
img = torch.randn(B, 3, 224, 224, device="cuda")
w = torch.randn(B, 196, 512, device="cuda", dtype=dt) / 512
def step():
    store.invalidate_shadow(); store.relink_grads(); store.zero_grad()
    out = vit(img, dt)
    (out * w).sum().backward()
    RT.join_streams()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 10
for _ in range(N): step()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / N * 1e3
# 12 blocks: 12*W^2 per token in the linears (x2 flop, x3 fwd+bwd) + attention 4*T*W per token
T, W = 197, 768
flops = 3 * 2 * B * T * (12 * 12 * W * W + 12 * 2 * T * W) + 3 * 2 * B * 196 * (768 * 768 + 768 * 512)
print(f"ViT-B/16 B={B} fwd+bwd {ms:.2f} ms  {B / ms * 1e3:.0f} img/s  {flops / ms / 1e9:.0f} TFLOP/s", flush=True)
import json
print(json.dumps({"metric": "encoder images/sec CLIP ViT-B/16 224x224 bs%d/GPU (forward + backward, encoder level: the reference has no end-to-end ViT path)" % B,
                  "value": round(B / ms * 1e3, 1), "unit": "images/sec", "n_gpus": 1, "steps": N, "warmup": 3, "ms_per_step": round(ms, 3),
                  "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                  "config": {"workload": "BASELINE config 4: CLIP ViT-B/16 image tower, fwd + bwd under a linear surrogate loss, eager issue", "global_batch": B, "parallelism": "dp1"},
                  "step_roofline": {"mfma_frac": round(flops / ms / 1e9 / 2500.0, 4), "note": "analytic FLOPs of the 12 blocks + patch / output projections vs 2.5 PFLOP/s"}}), flush=True)
