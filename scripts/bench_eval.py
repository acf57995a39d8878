"""Validation forward of CROG-R50: eval-mode model + crog_eval_maps (sigmoid + bicubic resize of the five maps),
B images of 416x416, fp32 (the reference's eval path, crog_engine.py:166) and bf16 autocast.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from crog_amd.engine import eval_maps
from crog_amd.model import build_crog
from crog_amd.testing import make_cfg, synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
from crog_amd import functional as Fn
if len(sys.argv) > 2 and sys.argv[2] == "nofold": Fn.EVAL_BN_FOLD = False      # A/B: BatchNorm as a separate scale / shift pass
cfg = make_cfg(); torch.manual_seed(0)
model, _ = build_crog(cfg); model = model.cuda().prepare(); model.eval()
batch = synthetic_batch(B, 416, 20, 49408, seed=1, device="cuda")
for name, dt in (("fp32", None), ("bf16", torch.bfloat16)):
    for _ in range(3): eval_maps(model, batch, autocast_dtype=dt)
    torch.cuda.synchronize(); t0 = time.perf_counter(); N = 10
    for _ in range(N): maps, _ = eval_maps(model, batch, autocast_dtype=dt)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / N
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    logits = torch.randn(B, 5, 104, 104, device="cuda")
    from crog_amd import kernels as K
    K.eval_maps(logits, 0b10011, 416, 416); torch.cuda.synchronize()
    e0.record()
    for _ in range(20): K.eval_maps(logits, 0b10011, 416, 416)
    e1.record(); torch.cuda.synchronize()
    tk = e0.elapsed_time(e1) / 20 * 1e-3
    by = B * 5 * (104 * 104 + 416 * 416) * 4
    print(f"eval {name} (BatchNorm {'folded' if Fn.EVAL_BN_FOLD else 'as a pass'}): {t*1e3:.2f} ms / {B} images = {B/t:.0f} img/s; crog_eval_maps alone {tk*1e6:.1f} us = {by/tk/1e9:.0f} GB/s algorithmic")
