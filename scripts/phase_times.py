"""Per-phase GPU time of one training step (serialised: side streams off) (GPU box)."""
import os, sys
os.environ["CROG_OVERLAP_WGRAD"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from crog_amd import functional as Fn
from crog_amd.model import build_crog
from crog_amd.optim import FusedAdam
from crog_amd.testing import make_cfg, synthetic_batch
cfg = make_cfg(); torch.manual_seed(0)
model, groups = build_crog(cfg); model = model.cuda().prepare(); model.overlap_text = False
opt = FusedAdam(groups, lr=1e-4, store=model.store)
b = synthetic_batch(32, 416, 20, 49408, seed=1, device="cuda"); model.train()
dt = torch.bfloat16
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
def run(timed):
    st = model.store; st.invalidate_shadow(); st.relink_grads(); st.zero_grad()
    marks = [("start", ev())]
    st.weights(dt); marks.append(("cast", ev()))
    V = model.backbone.visual
    x = Fn.conv_bn_act(b["img"].float(), V.conv1.w, V.bn1.buffers_ref(), ksize="s", relu=True, training=True, wpad=(27, 32, 32), dtype=dt)
    x = Fn.conv_bn_act(x, V.conv2.w, V.bn2.buffers_ref(), ksize=3, relu=True, training=True)
    x = Fn.conv_bn_act(x, V.conv3.w, V.bn3.buffers_ref(), ksize=3, relu=True, training=True)
    x = Fn.avgpool2(x); marks.append(("stem", ev()))
    x = V.layer1(x); marks.append(("layer1", ev()))
    x2 = V.layer2(x); marks.append(("layer2", ev()))
    x3 = V.layer3(x2); marks.append(("layer3", ev()))
    x4 = V.layer4(x3); marks.append(("layer4", ev()))
    x4 = V.attnpool(x4); marks.append(("attnpool", ev()))
    wfeat, state = model.backbone.text_features(b["word"], dt); marks.append(("text", ev()))
    fq = model.neck((x2, x3, x4), state); marks.append(("neck", ev()))
    fq = model.decoder(fq, wfeat, (b["word"] == 0).contiguous()); marks.append(("decoder", ev()))
    pred = model.proj(fq, state); marks.append(("proj", ev()))
    total, sums, small = Fn.head_loss(pred, [b[k] for k in ("mask", "qua", "sin", "cos", "wid")], True); marks.append(("loss", ev()))
    total.backward(); marks.append(("backward", ev()))
    opt.step(); marks.append(("adam", ev()))
    torch.cuda.synchronize()
    if timed:
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            print(f"{n1:10s} {e0.elapsed_time(e1):7.2f} ms")
        print(f"{'TOTAL':10s} {marks[0][1].elapsed_time(marks[-1][1]):7.2f} ms")
for i in range(3): run(i == 2)
