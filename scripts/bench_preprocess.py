"""Row N3 timing (GPU box): crog_preprocess_u8 on batches of 640 x 480 uint8 samples -> 416 x 416 fp32 tensors, against the CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from crog_amd.data import Preprocessor
from oracle import preprocess_oracle as O
B = 32
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.integers(0, 256, (B, 480, 640, 3), dtype=np.uint8)).cuda()
masks = torch.from_numpy(rng.integers(0, 180, (B, 4, 480, 640), dtype=np.uint8)).cuda()
pre = Preprocessor(416)
for _ in range(3): out = pre(img, masks)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): out = pre(img, masks)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e-3
byt = img.numel() + masks.numel() + 4 * (out["img"].numel() + 5 * out["mask"].numel())
t0 = time.time(); O.preprocess(img[0].cpu().numpy(), *[m.cpu().numpy() for m in masks[0]], 416); tc = time.time() - t0
print(f"preprocess B={B}: {t*1e6:.0f} us/batch = {B/t:.0f} img/s, {byt/t/1e9:.0f} GB/s algorithmic; CPU oracle (numpy, 1 core) {1/tc:.1f} img/s")
