import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import crog_amd.functional as Fn
from crog_amd.model import build_crog
from crog_amd.runtime import RT
from crog_amd.testing import make_cfg
torch.manual_seed(0)
model, _ = build_crog(make_cfg()); model = model.cuda().prepare(); model.train()
img = torch.randn(2, 3, 416, 416, generator=torch.Generator().manual_seed(3)).cuda()
st = model.store
for n, p, o, k, _ in st.entries:
    if n.endswith("bn3.weight"): st.P[o:o + k].fill_(0.5)
st.invalidate_shadow()
names = [(n, o, k) for n, p, o, k, _ in st.entries if n.startswith(("backbone.visual.conv", "backbone.visual.bn", "backbone.visual.layer1", "backbone.visual.layer2"))]
def run(fused):
    Fn.BN_BWD_FUSED = fused
    st.g_clean = False; st.zero_grad(); RT.begin_step(img.device); st.forward_begins()
    x3 = model.backbone.visual(img, torch.bfloat16)[0]
    x3.float().pow(2).mean().backward(); torch.cuda.synchronize()
    return st.G.clone()
g0 = run(False); ga, gb, gc, gd = run(False), run(False), run(True), run(True)
print('run1 vs run2 (plain):', ((g0-ga).norm()/ga.norm()).item(), ' run2 vs run3 (plain):', ((ga-gb).norm()/ga.norm()).item(), ' fused vs fused:', ((gc-gd).norm()/gc.norm()).item(), ' plain vs fused:', ((ga-gc).norm()/ga.norm()).item())
rel = lambda a, b: ((a - b).norm() / a.norm().clamp_min(1e-12)).item()
rows = sorted(((rel(ga[o:o+k], gc[o:o+k]), rel(ga[o:o+k], gb[o:o+k]), n, ga[o:o+k].norm().item()) for n, o, k in names), reverse=True)
for r in rows[:25]: print(f"fused-vs-plain {r[0]:.4f}  noise {r[1]:.4f}  |g| {r[3]:.3e}  {r[2]}")
