"""Pin the CPU oracle (oracle/crog_oracle.py) against fixtures captured from the reference itself
(tests/golden/*, written by oracle/make_golden.py which imported /root/reference in the build container).
No GPU needed."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crog_amd.testing import make_cfg, seeded_state, synthetic_batch, tiny_cfg  # noqa: E402
from oracle import crog_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def load_case(name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    return {k: torch.from_numpy(d[k]) for k in d.files}, meta


def oracle_step(cfg, meta, training=True, grads=True):
    P = seeded_state({k: tuple(v) for k, v in meta["shapes"].items()}, seed=meta["seed"])
    for n in meta["param_names"]:
        P[n].requires_grad_(grads)
    batch = synthetic_batch(meta["B"], cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=1234 + meta["seed"])
    tg = [batch[k] for k in ("mask", "qua", "sin", "cos", "wid")]
    out = O.crog_forward(P, batch["img"], batch["word"], tg, num_head=cfg.num_head, training=training,
                         use_contrastive=cfg.use_contrastive, use_grasp_masks=cfg.use_grasp_masks)
    return P, batch, out


def check(a, b, atol=1e-5, rtol=1e-4, what=""):
    a, b = a.detach().float(), b.detach().float()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    lim = atol + rtol * b.abs().max().item()
    assert err <= lim, f"{what}: max err {err:.3e} > {lim:.3e}"


@pytest.mark.parametrize("case,cfgf", [("tiny_crog", lambda: tiny_cfg()), ("tiny_crog_nomask", lambda: tiny_cfg(use_grasp_masks=False))])
def test_oracle_matches_reference_tiny(case, cfgf):
    g, meta = load_case(case)
    cfg = cfgf()
    P, batch, out = oracle_step(cfg, meta)
    out["total"].backward()
    names = ["ins", "qua", "sin", "cos", "wid"][: 5 if cfg.use_grasp_masks else 1]
    for i, nm in enumerate(names):
        check(out["preds"][i], g["pred_" + nm], what="pred_" + nm)
        check(out["targets_small"][i], g["tgt_" + nm], what="tgt_" + nm)
        check(out["losses"][i], g["loss_items"][i], atol=1e-5, what="loss_" + nm)
    check(out["total"], g["loss_total"], what="total")
    if "x2" in g:
        for k, v in zip(("x2", "x3", "x4"), out["vis"]):
            check(v, g[k], what=k)
        check(out["word_feat"], g["word_feat"], what="word_feat")
        check(out["state"], g["state"], what="state")
        check(out["fq"], g["fq"], what="fq")
        check(out["fq_dec"].reshape(g["fq_dec"].shape), g["fq_dec"], what="fq_dec")
    gn = torch.tensor([float(P[n].grad.norm()) if P[n].grad is not None else -1.0 for n in meta["param_names"]])
    ref = g["grad_norms"]
    assert ((gn < 0) == (ref < 0)).all(), "set of parameters without gradient differs"
    # k_proj.bias has a mathematically zero gradient (softmax is shift-invariant): absolute floor 1e-5
    bad = ((gn - ref).abs() > 3e-3 * ref.abs() + 1e-5) & (ref >= 0)
    assert not bad.any(), [(meta["param_names"][i], float(gn[i]), float(ref[i])) for i in bad.nonzero().flatten()[:5]]
    for k in g:
        if k.startswith("grad::"):
            check(P[k[6:]].grad, g[k], atol=1e-5, rtol=5e-3, what=k)
    chk = torch.tensor([float(P[k].double().sum()) for k in meta["bn_keys"]])
    assert torch.allclose(chk, g["bn_running_checksum"].float(), rtol=1e-4, atol=1e-3)
    # eval mode
    P2, _, ev = oracle_step(cfg, meta, training=False, grads=False)
    # eval pass in the fixture ran AFTER the training step: replay the same BN statistics
    for k in meta["bn_keys"]:
        P2[k] = P[k].detach().clone()
    with torch.no_grad():
        ev = O.crog_forward(P2, batch["img"], batch["word"], None, num_head=cfg.num_head, training=False,
                            use_contrastive=cfg.use_contrastive, use_grasp_masks=cfg.use_grasp_masks)
    for i, nm in enumerate(names):
        check(ev["preds"][i], g["eval_pred_" + nm], atol=2e-5, what="eval_pred_" + nm)


def test_param_groups_match_reference():
    _, meta = load_case("tiny_crog")
    g0, g1 = O.param_groups(meta["param_names"])
    assert len(g0) == meta["group_backbone"] and len(g1) == meta["group_head"]
    assert meta["group_lrs"] == pytest.approx([1e-5, 1e-4])


def _ops():
    d = np.load(os.path.join(GOLD, "ops.npz"))
    return {k: torch.from_numpy(d[k]) for k in d.files}


def _w(fx, tag):
    pre = tag + "::w::"
    return {k[len(pre):]: v.clone() for k, v in fx.items() if k.startswith(pre)}


def _run(fx, tag, fn):
    ins = []
    j = 0
    while f"{tag}::in{j}" in fx:
        t = fx[f"{tag}::in{j}"].clone()
        ins.append(t.requires_grad_(True) if t.is_floating_point() else t)
        j += 1
    o = fn(*ins)
    (o * torch.linspace(-1, 1, o.numel()).view_as(o)).sum().backward()
    check(o, fx[tag + "::out"], atol=2e-5, what=tag)
    for j, t in enumerate(ins):
        if f"{tag}::din{j}" in fx:
            check(t.grad, fx[f"{tag}::din{j}"], atol=2e-5, rtol=1e-3, what=f"{tag} din{j}")


def test_oracle_ops_match_reference_classes():
    fx = _ops()
    P = {"b." + k: v for k, v in _w(fx, "bottleneck_s2").items()}
    _run(fx, "bottleneck_s2", lambda x: O.bottleneck(P, "b", x, 2, True))
    P = {"b." + k: v for k, v in _w(fx, "bottleneck_s1").items()}
    _run(fx, "bottleneck_s1", lambda x: O.bottleneck(P, "b", x, 1, True))
    P = {"a." + k: v for k, v in _w(fx, "attnpool").items()}
    _run(fx, "attnpool", lambda x: O.attnpool(P, "a", x, 2, True))
    P = {"r." + k: v for k, v in _w(fx, "resblock").items()}
    mask = torch.full((6, 6), float("-inf")).triu_(1)
    _run(fx, "resblock", lambda x: O.residual_attention_block(P, "r", x, 2, mask))
    P = {"d." + k: v for k, v in _w(fx, "declayer").items()}
    _run(fx, "declayer", lambda v, t, vp, tp, pm: O.decoder_layer(P, "d", v, t, vp.squeeze(1), tp.squeeze(1), pm, 2))
    check(O.pos1d(128, 9), fx["pos1d_128_9"], what="pos1d")
    check(O.pos2d(128, 5, 7), fx["pos2d_128_5_7"], what="pos2d")
    P = {"p." + k: v for k, v in _w(fx, "mtproj").items()}
    _run(fx, "mtproj", lambda x, s: torch.cat(O.projector(P, x, s, True, "p"), 1))
    P = {"p." + k: v for k, v in _w(fx, "proj").items()}
    _run(fx, "proj", lambda x, s: O.projector(P, x, s, True, "p")[0])


def test_coordconv_matches_reference():
    fx = _ops()
    P = {"neck.coordconv.0." + k: v for k, v in _w(fx, "coordconv").items()}
    x = fx["coordconv::in0"]
    b, _, h, w = x.shape
    xs = torch.linspace(-1, 1, w).view(1, 1, 1, w).expand(b, 1, h, w)
    ys = torch.linspace(-1, 1, h).view(1, 1, h, 1).expand(b, 1, h, w)
    out = O.conv_bn_relu(P, "neck.coordconv.0.conv1", torch.cat([x, xs, ys], 1), True)
    check(out, fx["coordconv::out"], atol=2e-5, what="coordconv")


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "crog_r50_b2.npz")), reason="full fixture not generated")
def test_oracle_matches_reference_config1():
    """BASELINE config 1: CROG-R50, 2 x 416x416 RGB + 20 tokens (forward + losses; ~20 s on 8 cores)."""
    g, meta = load_case("crog_r50_b2")
    cfg = make_cfg(dropout=0.0)
    P, batch, out = oracle_step(cfg, meta, grads=False)
    for i, nm in enumerate(["ins", "qua", "sin", "cos", "wid"]):
        check(out["preds"][i], g["pred_" + nm], atol=5e-5, rtol=1e-4, what="pred_" + nm)
    check(out["total"], g["loss_total"], atol=1e-5, what="total")


@pytest.mark.parametrize("shape,size,chans", [((2, 5, 13, 13), (52, 52), (0, 1, 4)), ((1, 1, 26, 26), (104, 104), (0,)),
                                              ((1, 2, 7, 9), (30, 17), ()), ((1, 5, 104, 104), (416, 416), (0, 1, 4))])
def test_oracle_eval_maps_matches_reference_statements(shape, size, chans):
    """engine/crog_engine.py:181-211 verbatim in behaviour: torch.sigmoid on (ins, qua, wid), then F.interpolate(bicubic,
    align_corners=True) — the ATen routine the reference runs — against the oracle's explicit-tap restatement."""
    import torch.nn.functional as F
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(5)) * 3
    ref = x.clone()
    for c in chans:
        ref[:, c] = torch.sigmoid(ref[:, c])
    ref = F.interpolate(ref, size=size, mode="bicubic", align_corners=True)
    assert (O.eval_maps(x, chans, size) - ref).abs().max().item() < 1e-5


def test_oracle_vit_tower_matches_reference():
    """BASELINE config 4 (CLIP ViT tower): encoder-level pin of the oracle against the reference's VisionTransformer
    (clip.py:286-332) on the tiny fixture written by oracle/make_golden.py."""
    d = np.load(os.path.join(GOLD, "vit_tiny.npz"))
    fx = {k: torch.from_numpy(d[k]) for k in d.files}
    P = {"v." + k[3:]: v.clone().requires_grad_(True) for k, v in fx.items() if k.startswith("w::")}
    o = O.encode_image_vit(P, fx["in0"], pre="v")
    (o * torch.linspace(-1, 1, o.numel()).view_as(o)).sum().backward()
    check(o, fx["out"], atol=2e-5, what="vit out")
    for k, v in fx.items():
        if k.startswith("dw::"):
            check(P["v." + k[4:]].grad, v, atol=5e-5, rtol=1e-3, what="vit " + k)


@pytest.mark.parametrize("case", ["ssg_tiny_rgbd", "ssg_tiny_rgb"])
def test_oracle_ssg_trunk_matches_reference(case):
    """BASELINE config 5 (SSG-R50): oracle/ssg_oracle.py against the reference's own SSG modules (fixture written by
    oracle/make_golden.py ssg): raw predictions, surrogate-loss gradients, BN running statistics, anchors, class softmax."""
    from types import SimpleNamespace
    from crog_amd.testing import SSG_OUTPUTS, ssg_surrogate_loss, synthetic_ssg_batch
    from oracle import ssg_oracle as S
    fx, meta = load_case(case)
    cfg = SimpleNamespace(**meta["cfg"])
    P = seeded_state({k: tuple(v) for k, v in meta["shapes"].items()}, seed=meta["seed"])
    for n in meta["param_names"]:
        P[n].requires_grad_(True)
    batch = synthetic_ssg_batch(meta["B"], cfg.img_size, cfg.with_depth, seed=1234 + meta["seed"])
    img = torch.cat([batch["rgb"], batch["depth"]], 1) if cfg.with_depth else batch["rgb"]
    raw = S.ssg_trunk(P, img, cfg.num_classes, cfg.num_protos, training=True)
    for k in SSG_OUTPUTS:
        check(raw[k], fx[k], atol=2e-5, what=f"{case} {k}")
    loss = ssg_surrogate_loss(raw, meta["seed"])
    check(loss, fx["loss"], atol=1e-6, what="loss")
    loss.backward()
    gn = torch.tensor([float(P[n].grad.norm()) for n in meta["param_names"]])
    check(gn, fx["grad_norms"], atol=1e-6, rtol=2e-3, what="grad norms")
    for n in meta["param_names"]:
        check(P[n].grad.flatten()[:64], fx["grad::" + n], atol=2e-6, rtol=2e-3, what="grad " + n)
    bn = torch.tensor([float(P[k].double().sum()) for k in meta["bn_keys"]])
    check(bn, fx["bn_running_checksum"], atol=1e-3, what="bn running stats")
    check(torch.tensor(S.anchors(cfg.aspect_ratios, cfg.img_size, cfg.anchor_strides)), fx["anchors"].flatten(), atol=1e-7, what="anchors")
    with torch.no_grad():
        ev = S.ssg_trunk(P, img, cfg.num_classes, cfg.num_protos, training=False)
    check(torch.softmax(ev["class_pred"], -1), fx["eval_cls_pred"], atol=2e-5, what="eval cls_pred")
    check(ev["box_pred"], fx["eval_box_pred"], atol=2e-5, what="eval box_pred")


@pytest.mark.parametrize("case", ["ssg_tiny_rgbd", "ssg_tiny_rgb"])
def test_ssg_loss_matches_reference(case):
    """the loss oracle (oracle/ssg_loss_oracle.py) on the fixture's raw predictions + synthetic ground truth
    against the reference's own SSG.compute_loss (ssg.py:297-530): eight losses and the gradient of their sum w.r.t. every
    prediction.  (The product's batched device implementation is held to the same fixtures in tests/test_ssg_gpu.py.)"""
    from types import SimpleNamespace
    from oracle.ssg_loss_oracle import ssg_loss
    from crog_amd.testing import SSG_OUTPUTS, synthetic_ssg_targets
    from oracle import ssg_oracle as S
    fx, meta = load_case(case)
    cfg = SimpleNamespace(**meta["cfg"])
    tg = synthetic_ssg_targets(meta["B"], cfg.img_size, cfg.num_classes, seed=1234 + meta["seed"])
    raw = {k: fx[k].clone().requires_grad_(True) for k in SSG_OUTPUTS}
    anchors = torch.tensor(S.anchors(cfg.aspect_ratios, cfg.img_size, cfg.anchor_strides)).reshape(-1, 4)
    out = {}
    losses = ssg_loss(cfg, anchors, raw, tg, out)
    assert list(losses) == ["loss_cls", "loss_box", "loss_ins", "loss_sem", "loss_qua", "loss_sin", "loss_cos", "loss_wid"]
    for k, v in losses.items():
        check(v, fx["S2::" + k], atol=1e-6, rtol=1e-5, what=f"{case} {k}")
    sum(losses.values()).backward()
    for k in SSG_OUTPUTS:
        check(raw[k].grad, fx["S2::d_" + k], atol=1e-7, rtol=1e-4, what=f"{case} d{k}")
    assert out["inter_mask_p"].shape == out["inter_mask_gt"].shape and out["inter_mask_p"].dim() == 3


@pytest.mark.parametrize("case", ["ssg_loss_b8", "ssg_loss_b8_limit"])
def test_ssg_loss_oracle_at_the_full_anchor_set(case):
    """The loss oracle at ssg_r50.yaml's own sizes (18,525 anchors, 136 x 136 prototypes, B = 8 ragged images) against the reference's
    compute_loss on seeded random predictions; `_limit` lowers masks_to_train so that the CPU-randperm subsampling branch runs
    (same generator seed as the fixture script)."""
    from types import SimpleNamespace
    from crog_amd.testing import SSG_OUTPUTS, synthetic_ssg_batch, synthetic_ssg_predictions, synthetic_ssg_targets
    from oracle import ssg_oracle as S
    from oracle.ssg_loss_oracle import ssg_loss
    fx, meta = load_case(case)
    cfg = SimpleNamespace(**meta["cfg"])
    anchors = torch.tensor(S.anchors(cfg.aspect_ratios, cfg.img_size, cfg.anchor_strides)).reshape(-1, 4)
    assert anchors.shape[0] == meta["anchors"]
    raw = {k: v.requires_grad_(True) for k, v in synthetic_ssg_predictions(meta["B"], anchors.shape[0], cfg, meta["seed"]).items()}
    tg = synthetic_ssg_targets(meta["B"], cfg.img_size, cfg.num_classes, seed=1234 + meta["seed"])
    torch.manual_seed(4242 + meta["seed"])
    losses = ssg_loss(cfg, anchors, raw, tg, {})
    for k, v in losses.items():
        check(v, fx["S2::" + k], atol=1e-5, rtol=1e-5, what=f"{case} {k}")
    sum(losses.values()).backward()
    for k in SSG_OUTPUTS:
        g = raw[k].grad.flatten()
        check(g[::meta["stride"]], fx["S2::d_" + k + "::sample"], atol=1e-7, rtol=1e-4, what=f"{case} d{k} samples")
        ref_sums = fx["S2::d_" + k + "::sums"]
        got = torch.stack([g.double().sum(), g.double().abs().sum()])
        assert float((got - ref_sums).abs().max()) <= 1e-4 * float(ref_sums[1]) + 1e-9, (case, k, got, ref_sums)
