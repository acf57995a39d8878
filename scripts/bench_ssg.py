"""BASELINE config 5: SSG-R50 trunk (ssg_r50.yaml widths, 544x544), forward + backward with a linear surrogate loss, bf16. GPU box.
Usage: bench_ssg.py [B] [rgb|rgbd]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd.model.ssg import build_ssg
from crog_amd.runtime import RT
from crog_amd.testing import SSG_OUTPUTS, ssg_cfg, synthetic_ssg_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
depth = (sys.argv[2] if len(sys.argv) > 2 else "rgb") == "rgbd"
cfg = ssg_cfg(with_depth=depth)
torch.manual_seed(0)
model = build_ssg(cfg).cuda().prepare(); model.train()
batch = synthetic_ssg_batch(B, cfg.img_size, depth, device="cuda")
img = torch.cat([batch["rgb"], batch["depth"]], 1) if depth else batch["rgb"]
ws = None
def step():
    global ws
    with torch.autocast("cuda", dtype=torch.bfloat16):
        raw = model.trunk(img)
    if ws is None:
        ws = {k: torch.randn_like(raw[k]) / raw[k].numel() for k in SSG_OUTPUTS}
    loss = sum((raw[k] * ws[k]).sum() for k in SSG_OUTPUTS)
    loss.backward()
    RT.join_streams()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 8
for _ in range(N): step()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / N * 1e3
print(f"SSG-R50 {'RGB-D' if depth else 'RGB'} 544x544 B={B} trunk fwd+bwd {ms:.1f} ms  {B / ms * 1e3:.0f} img/s  "
      f"{360.3e9 * B / ms / 1e9:.0f} TFLOP/s (360.3 GFLOP/img fwd+bwd)  peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
import json
print(json.dumps({"metric": "trunk images/sec SSG-R50 544x544 bs%d/GPU (forward + backward)" % B, "value": round(B / ms * 1e3, 1), "unit": "images/sec",
                  "n_gpus": 1, "steps": N, "warmup": 3, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                  "dtype": "bf16", "data": "synthetic",
                  "config": {"workload": "BASELINE config 5: SSG-R50 (ssg_r50.yaml widths, %s) conv trunk + heads, fwd + bwd under a linear surrogate loss, eager issue" % ("RGB-D" if depth else "RGB"),
                             "global_batch": B, "parallelism": "dp1"},
                  "step_roofline": {"mfma_frac": round(360.3e9 * B / ms / 1e9 / 2500.0, 4), "note": "360.3 GFLOP/img fwd+bwd (BASELINE.md) vs 2.5 PFLOP/s"},
                  "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}), flush=True)
