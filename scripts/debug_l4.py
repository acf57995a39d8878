"""Gradient probes inside layer4.0 of the tiny model: HIP fp32 vs fp64 oracle (debug aid; GPU box)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from crog_amd.testing import seeded_state, synthetic_batch, tiny_cfg
from crog_amd.model import build_crog
from crog_amd import functional as Fn
from crog_amd.model import clip as C
from oracle import crog_oracle as O

meta = json.load(open(os.path.join(ROOT, "tests/golden/tiny_crog.json"))); cfg = tiny_cfg()
shapes = {k: tuple(v) for k, v in meta["shapes"].items()}
b = synthetic_batch(meta["B"], cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=1234 + meta["seed"])
probe64, probeh = {}, {}

def o_bottleneck(P, pre, x, stride, training):
    def hk(name, t):
        if pre.endswith("layer4.0"):
            t.register_hook(lambda g: probe64.__setitem__(name, g.detach().clone()))
            probe64["fwd_" + name] = t.detach().clone()
    out = F.relu(O.batchnorm(P, pre + ".bn1", F.conv2d(x, P[pre + ".conv1.weight"]), training)); hk("y1", out)
    out = F.relu(O.batchnorm(P, pre + ".bn2", F.conv2d(out, P[pre + ".conv2.weight"], padding=1), training)); hk("y2", out)
    if stride > 1:
        out = F.avg_pool2d(out, stride); hk("pool", out)
    out = O.batchnorm(P, pre + ".bn3", F.conv2d(out, P[pre + ".conv3.weight"]), training)
    identity = x
    if (pre + ".downsample.0.weight") in P:
        identity = F.avg_pool2d(x, stride) if stride > 1 else x
        identity = O.batchnorm(P, pre + ".downsample.1", F.conv2d(identity, P[pre + ".downsample.0.weight"]), training)
    y = F.relu(out + identity); hk("out", y)
    return y
O.bottleneck = o_bottleneck
P = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in seeded_state(shapes, seed=meta["seed"]).items()}
torch.set_default_dtype(torch.float64)
for n in meta["param_names"]: P[n].requires_grad_(True)
out = O.crog_forward(P, b["img"].double(), b["word"], [b[k].double() for k in ("mask","qua","sin","cos","wid")], num_head=cfg.num_head)
out["total"].backward()
torch.set_default_dtype(torch.float32)

def h_forward(self, x):
    tr = self.training
    def hk(name, t):
        if getattr(self, "_probe", False):
            t.register_hook(lambda g: probeh.__setitem__(name, g.detach().clone()))
            probeh["fwd_" + name] = t.detach().clone()
    out = Fn.conv_bn_act(x, self.conv1.w, self.bn1.buffers_ref(), ksize=1, relu=True, training=tr); hk("y1", out)
    out = Fn.conv_bn_act(out, self.conv2.w, self.bn2.buffers_ref(), ksize=3, relu=True, training=tr); hk("y2", out)
    if self.stride > 1:
        out = Fn.avgpool2(out); hk("pool", out)
    identity = x
    if self.downsample is not None:
        if self.stride > 1:
            identity = Fn.avgpool2(x)
        identity = Fn.conv_bn_act(identity, self.downsample["0"].w, self.downsample["1"].buffers_ref(), ksize=1, relu=False, training=tr)
    y = Fn.conv_bn_act(out, self.conv3.w, self.bn3.buffers_ref(), ksize=1, relu=True, res=identity, training=tr); hk("out", y)
    return y
C.Bottleneck.forward = h_forward
model, _ = build_crog(cfg)
model.load_state_dict(seeded_state(shapes, seed=meta["seed"]))
model = model.cuda(); model.compute_dtype = torch.float32; model.prepare(); model.train()
model.backbone.visual.layer4[0]._probe = True
bc = {k: v.cuda() for k, v in b.items()}
preds, tgts, loss, ld = model(bc["img"], bc["word"], bc["mask"], bc["qua"], bc["sin"], bc["cos"], bc["wid"])
loss.backward(); torch.cuda.synchronize()
def rel(a, b_):
    return float((a.double() - b_).norm() / (b_.norm() + 1e-30))
for k in ("out", "pool", "y2", "y1"):
    gh = probeh[k].permute(0, 3, 1, 2).cpu(); g6 = probe64[k]
    fh = probeh["fwd_" + k].permute(0, 3, 1, 2).cpu(); f6 = probe64["fwd_" + k]
    d = (gh.double() - g6)
    print(f"{k:5s} fwd rel {rel(fh, f6):.2e} | grad rel {rel(gh, g6):.2e} | per-channel mean of grad err / rms grad: {float(d.mean((0,2,3)).abs().mean() / g6.pow(2).mean().sqrt()):.2e}"
          f" | mask mismatch {(int(((fh > 0) != (f6 > 0)).sum()))} of {fh.numel()}")
