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
