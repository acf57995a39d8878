"""Time of the per-step data-gradient weight layout refresh (crog_dgrad_weights over the CROG-R50 store) stand-alone (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
from crog_amd.model import build_crog
from crog_amd.testing import make_cfg
cfg = make_cfg(); torch.manual_seed(0)
model, _ = build_crog(cfg); model = model.cuda().prepare()
st = model.store
src = st.weights(torch.bfloat16); st.ensure_t(torch.bfloat16)
def run(): K.dgrad_weights(src, st.T[torch.bfloat16], st.tr_table, st.tr_count)
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
print(f"grid {os.environ.get('CROG_DGW_GRID', '24')}: {s.elapsed_time(e) / 20 * 1e3:.1f} us for {st.tr_count} matrices, {st.total * 2 / 1e6:.0f} MB store")
