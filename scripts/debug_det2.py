"""First op of the image tower whose output differs between two runs on identical inputs (bf16, training mode)."""
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
v = model.backbone.visual
dt = torch.bfloat16
rel = lambda a, b: ((a.float() - b.float()).norm() / a.float().norm().clamp_min(1e-12)).item()
def twice(name, fn):
    outs = []
    for _ in range(3):
        RT.begin_step(img.device); st.forward_begins()
        with torch.no_grad():
            outs.append(fn().clone())
    torch.cuda.synchronize()
    print(f"{name:34s} equal {torch.equal(outs[0], outs[1])} {torch.equal(outs[1], outs[2])}  rel {rel(outs[0], outs[1]):.2e}  shape {tuple(outs[0].shape)}")
    return outs[0]
x1 = twice("stem conv1+bn+relu", lambda: Fn.conv_bn_act(img, v.conv1.w, v.bn1.buffers_ref(), ksize="s", relu=True, training=True, wpad=(27, 32, 32), dtype=dt))
x2 = twice("stem conv2+bn+relu", lambda: Fn.conv_bn_act(x1, v.conv2.w, v.bn2.buffers_ref(), ksize=3, relu=True, training=True))
x3 = twice("stem conv3+bn+relu", lambda: Fn.conv_bn_act(x2, v.conv3.w, v.bn3.buffers_ref(), ksize=3, relu=True, training=True))
x4 = twice("avgpool", lambda: Fn.avgpool2(x3))
b = v.layer1[0]
y1 = twice("l1.0 conv1", lambda: Fn.conv_bn_act(x4, b.conv1.w, b.bn1.buffers_ref(), ksize=1, relu=True, training=True))
y2 = twice("l1.0 conv2", lambda: Fn.conv_bn_act(y1, b.conv2.w, b.bn2.buffers_ref(), ksize=3, relu=True, training=True))
idn = twice("l1.0 downsample", lambda: Fn.conv_bn_act(x4, b.downsample["0"].w, b.downsample["1"].buffers_ref(), ksize=1, relu=False, training=True))
y3 = twice("l1.0 conv3+res", lambda: Fn.conv_bn_act(y2, b.conv3.w, b.bn3.buffers_ref(), ksize=1, relu=True, res=idn, training=True))
z = twice("layer1 whole", lambda: v.layer1(x4))
z2 = twice("layer2 whole", lambda: v.layer2(z))
