"""Generate tests/golden/* by importing the REFERENCE (HilbertXu/CROG) in the build container.

Run here only (needs /root/reference; the GPU box never sees it):   python oracle/make_golden.py
The reference is imported read-only (no bytecode written), with stubs for the packages the image
lacks (loguru, cv2) and torch.jit.load replaced by an object that yields a CLIP state dict of the
wanted architecture (there is no RN50.pt and no network) — recipe from SURVEY.md §8(c).
Only data leaves this script: inputs are re-derivable from seeds, outputs are stored as .npz/.json.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = os.environ.get("CROG_REFERENCE", "/root/reference")

import numpy as np
import torch

from crog_amd.testing import (SSG_OUTPUTS, grad_probe, make_cfg, seeded_cotangent, seeded_state, ssg_surrogate_loss, ssg_tiny_cfg, stage_of,
                              synthetic_batch, synthetic_ssg_batch, synthetic_ssg_targets, tiny_cfg)

GOLD = os.path.join(REPO, "tests", "golden")


def import_reference():
    class _Logger:
        def __getattr__(self, k):
            return lambda *a, **kw: None
    sys.modules["loguru"] = types.SimpleNamespace(logger=_Logger())
    sys.modules["cv2"] = types.ModuleType("cv2")
    sys.path.insert(0, REF)
    import model as ref_model  # noqa
    import model.clip as ref_clip
    import model.crog as ref_crog
    import model.layers as ref_layers
    return ref_model, ref_clip, ref_crog, ref_layers


def build_reference(cfg, ref_model, ref_clip):
    a = cfg.clip_arch
    proto = ref_clip.CLIP(a["embed_dim"], a["image_resolution"], a["vision_layers"], a["vision_width"], a["vision_patch_size"],
                          a["context_length"], cfg.word_len, a["vocab_size"], a["transformer_width"], a["transformer_heads"],
                          a["transformer_layers"])
    sd = proto.state_dict()

    class _Jit:
        def eval(self):
            return self

        def state_dict(self):
            return dict(sd)
    orig = torch.jit.load
    torch.jit.load = lambda *a_, **k_: _Jit()
    try:
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            model, groups = ref_model.build_crog(cfg)
    finally:
        torch.jit.load = orig
    return model, groups


def run_case(name, cfg, B, seed, ref_model, ref_clip, store_intermediates, residual_gain=1.0):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model, groups = build_reference(cfg, ref_model, ref_clip)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(seeded_state(shapes, seed=seed, residual_gain=residual_gain))
    batch = synthetic_batch(B, cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=1234 + seed)
    out = {}
    inter = {}
    if store_intermediates:
        def hook(key):
            def f(_m, _i, o):
                inter[key] = o
            return f
        model.neck.register_forward_hook(hook("fq"))
        model.decoder.register_forward_hook(hook("fq_dec"))
    model.train()
    vis = model.backbone.encode_image(batch["img"])
    wfeat, state = model.backbone.encode_text(batch["word"])
    if store_intermediates:
        out.update(x2=vis[0], x3=vis[1], x4=vis[2], word_feat=wfeat, state=state)
    # the encode_* calls above updated BN running stats once; reload so that the recorded step is step 1
    model.load_state_dict(seeded_state(shapes, seed=seed, residual_gain=residual_gain))
    model.zero_grad()
    preds, tgts, loss, loss_dict = model(batch["img"], batch["word"], batch["mask"], batch["qua"], batch["sin"], batch["cos"], batch["wid"])
    loss.backward()
    for i, nm in enumerate(["ins", "qua", "sin", "cos", "wid"]):
        if preds[i] is not None:
            out["pred_" + nm] = preds[i]
            out["tgt_" + nm] = tgts[i]
    out["loss_total"] = loss.detach()
    out["loss_items"] = torch.tensor([loss_dict[k] for k in ("m_ins", "m_qua", "m_sin", "m_cos", "m_wid")])
    if store_intermediates:
        out["fq"] = inter["fq"]
        out["fq_dec"] = inter["fq_dec"]
    names = [n for n, _ in model.named_parameters()]
    gn = torch.tensor([float(p.grad.norm()) if p.grad is not None else -1.0 for _, p in model.named_parameters()])
    out["grad_norms"] = gn
    # a few full gradients (small tensors) to pin direction, not just norm
    sd_grads = {}
    for n, p in model.named_parameters():
        if p.grad is not None and p.numel() <= 4096 and any(t in n for t in ("bn1.weight", "ln_final", "norm.weight", "txt.bias", "vis.4.bias",
                                                                              "attnpool.c_proj.bias", "norm_layer.0.bias")):
            sd_grads["grad::" + n] = p.grad
    out.update(sd_grads)
    # BN running statistics after this one training step
    bn_sum = {k: float(v.double().sum()) for k, v in model.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    out["bn_running_checksum"] = torch.tensor([bn_sum[k] for k in sorted(bn_sum)])
    # metric (utils/misc.py:115-131) restated inline to avoid importing cv2-dependent utils
    model.eval()
    with torch.no_grad():
        ev = model(batch["img"], batch["word"], batch["mask"], batch["qua"], batch["sin"], batch["cos"], batch["wid"])
    ev_preds = ev[0] if isinstance(ev[0], (tuple, list)) else [ev[0]]
    for i, nm in enumerate(["ins", "qua", "sin", "cos", "wid"]):
        if i < len(ev_preds) and ev_preds[i] is not None:
            out["eval_pred_" + nm] = ev_preds[i]
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **{k: v.detach().numpy() for k, v in out.items()})
    meta = dict(param_names=names, shapes={k: list(v) for k, v in shapes.items()},
                group_backbone=len(groups[0]["params"]), group_head=len(groups[1]["params"]),
                group_lrs=[groups[0]["initial_lr"], groups[1]["initial_lr"]], seed=seed, B=B, residual_gain=residual_gain,
                bn_keys=sorted(bn_sum))
    json.dump(meta, open(os.path.join(GOLD, name + ".json"), "w"))
    print(name, "loss", float(loss.detach()), "items", out["loss_items"].tolist(), flush=True)
    return model


PINNED_GRADS = ("bn1.weight", "ln_final", "norm.weight", "txt.bias", "vis.4.bias", "attnpool.c_proj.bias", "norm_layer.0.bias")


def run_case_bf16(name, cfg, B, seed, ref_model, ref_clip, residual_gain=1.0):
    """The yardstick for the benchmarked dtype: the REFERENCE's own training step under bf16 autocast (crog_engine.py:72-73 runs the
    forward under amp.autocast(); here torch.autocast("cpu", dtype=torch.bfloat16), the only autocast this container can execute) on
    the weights / inputs of the fp32 fixture `name`.  Stored next to it as `name`_bf16ref: logits, losses, gradient norms and the
    pinned small gradients - what a correct bf16 implementation's distance to the fp32 result looks like."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model, groups = build_reference(cfg, ref_model, ref_clip)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(seeded_state(shapes, seed=seed, residual_gain=residual_gain))
    batch = synthetic_batch(B, cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=1234 + seed)
    model.train()
    model.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        preds, tgts, loss, loss_dict = model(batch["img"], batch["word"], batch["mask"], batch["qua"], batch["sin"], batch["cos"], batch["wid"])
    loss.backward()
    out = {}
    for i, nm in enumerate(["ins", "qua", "sin", "cos", "wid"]):
        if preds[i] is not None:
            out["pred_" + nm] = preds[i].float()
    out["loss_total"] = loss.detach().float()
    out["loss_items"] = torch.tensor([loss_dict[k] for k in ("m_ins", "m_qua", "m_sin", "m_cos", "m_wid")])
    out["grad_norms"] = torch.tensor([float(p.grad.float().norm()) if p.grad is not None else -1.0 for _, p in model.named_parameters()])
    for n, p in model.named_parameters():
        if p.grad is not None and p.numel() <= 4096 and any(t in n for t in PINNED_GRADS):
            out["grad::" + n] = p.grad.float()
    np.savez_compressed(os.path.join(GOLD, name + "_bf16ref.npz"), **{k: v.detach().numpy() for k, v in out.items()})
    print(name + "_bf16ref", "loss", float(loss.detach()), "pred dtype", preds[0].dtype, flush=True)


def stage_fixture(name, cfg, B, seed, ref_model, ref_clip, residual_gain, maps, bf16=True):
    """Stage-isolated backward pins.  The reference's full training step is run once with `retain_grad` on the stage
    boundaries - (word_feat, state) out of the text tower (clip.py:439-456), (x2, x3, x4) out of the image tower (clip.py:207-223), fq out of
    the neck (layers.py:371-398), fq_dec out of the decoder (layers.py:243-277) - and each stage is then run ALONE on the recorded boundary
    values with the recorded upstream gradients injected: in fp32 (must reproduce the full step's parameter gradients: asserted here) and
    under bf16 autocast (what a correct bf16 stage costs with its upstream gradient held fixed - the yardstick of the GPU test, free of
    whatever the stages above amplify).  Stored as `name`_stages.npz: per parameter ||g|| and <g, probe(name)> of the fp32 step and of the
    bf16 stage runs; the boundary values and gradients themselves for the text tower (always: 45 K floats) and, with `maps` (the tiny model),
    for every boundary.  Full depth (`maps` False): image tower / neck / decoder boundaries are 23 M floats - instead those stages are pinned on
    SEEDED inputs and cotangents (crog_amd.testing.seeded_cotangent, re-derived on the GPU side), fp32 and bf16 alike."""
    import contextlib
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model, _ = build_reference(cfg, ref_model, ref_clip)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    state0 = seeded_state(shapes, seed=seed, residual_gain=residual_gain)
    model.load_state_dict(state0)
    batch = synthetic_batch(B, cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=1234 + seed)
    names = [n for n, _ in model.named_parameters()]
    params = dict(model.named_parameters())
    probes = {n: grad_probe(n, params[n].numel(), seed) for n in names}
    model.train()
    kept = {}
    bb = model.backbone
    enc_i, enc_t = bb.encode_image, bb.encode_text

    # (x3 is computed from x2, x4 from x3, state from word_feat: a boundary's OWN retained gradient would include what flows back through the
    # later outputs of the same stage.  The stage hands out identity copies; their gradients are what the downstream stages send back.)
    def wrap_i(img):
        vis = tuple(v.clone() for v in enc_i(img))
        for k, v in zip(("x2", "x3", "x4"), vis):
            v.retain_grad()
            kept[k] = v
        return vis

    def wrap_t(w):
        wf, st = (t.clone() for t in enc_t(w))
        wf.retain_grad(), st.retain_grad()
        kept["word_feat"], kept["state"] = wf, st
        return wf, st

    def keep(key):
        def f(_m, _i, o):
            o.retain_grad()
            kept[key] = o
        return f
    bb.encode_image, bb.encode_text = wrap_i, wrap_t
    hooks = [model.neck.register_forward_hook(keep("fq")), model.decoder.register_forward_hook(keep("fq_dec"))]
    model.zero_grad()
    _, _, loss, _ = model(batch["img"], batch["word"], batch["mask"], batch["qua"], batch["sin"], batch["cos"], batch["wid"])
    loss.backward()
    del bb.encode_image, bb.encode_text
    for h in hooks:
        h.remove()
    vals = {k: v.detach().clone() for k, v in kept.items()}
    grads = {k: v.grad.detach().clone() for k, v in kept.items()}
    pad_mask = torch.zeros_like(batch["word"]).masked_fill_(batch["word"] == 0, 1).bool()      # crog.py:55

    def measure():
        gn = torch.full((len(names),), -1.0, dtype=torch.float64)
        gd = torch.zeros(len(names), dtype=torch.float64)
        for i, n in enumerate(names):
            g = params[n].grad
            if g is not None:
                gn[i], gd[i] = float(g.double().norm()), float(g.double().flatten() @ probes[n].double())
        return gn, gd
    full_gn, full_gd = measure()
    out = {"full::gnorm": full_gn, "full::gdot": full_gd}

    def run_stage(stage, ins, cots, autocast, f64=False):
        """-> (parameter gradient norms, probes, input gradients) of `stage` alone: outputs contracted with `cots`.  f64: the same in
        float64 (the exact answer the two fp32 implementations are measured against)."""
        if f64:
            model.double()
        model.load_state_dict(state0)
        model.zero_grad()
        leaves = {k: (v.double() if (f64 and v.is_floating_point()) else v.clone()).requires_grad_(v.is_floating_point()) for k, v in ins.items()}
        with (torch.autocast("cpu", dtype=torch.bfloat16) if autocast else contextlib.nullcontext()):
            if stage == "text":
                outs = list(bb.encode_text(leaves["word"]))
            elif stage == "image":
                outs = list(bb.encode_image(leaves["img"]))
            elif stage == "neck":
                outs = [model.neck((leaves["x2"], leaves["x3"], leaves["x4"]), leaves["state"])]
            else:
                outs = [model.decoder(leaves["fq"], leaves["word_feat"], pad_mask)]
        torch.autograd.backward(outs, [c.to(o.dtype) for c, o in zip(cots, outs)])
        gn, gd = measure()
        res = gn, gd, {k: v.grad.float() for k, v in leaves.items() if v.grad is not None}, [o.detach().float() for o in outs]
        if f64:
            model.float()
            model.load_state_dict(state0)
        return res

    real = {"text": (dict(word=batch["word"]), [grads["word_feat"], grads["state"]]),
            "image": (dict(img=batch["img"]), [grads["x2"], grads["x3"], grads["x4"]]),
            "neck": ({k: vals[k] for k in ("x2", "x3", "x4", "state")}, [grads["fq"]]),
            "decoder": (dict(fq=vals["fq"], word_feat=vals["word_feat"]), [grads["fq_dec"]])}
    meta_shapes = {k: list(v.shape) for k, v in vals.items()}
    for k in ("word_feat", "state"):
        out["val::" + k], out["grad::" + k] = vals[k], grads[k]
    if maps:
        for k in vals:
            out["val::" + k], out["grad::" + k] = vals[k], grads[k]
    stages = ["text", "image", "neck", "decoder"] if maps else ["text"]
    for st in stages:
        ins, cots = real[st]
        gn, gd, din, _ = run_stage(st, ins, cots, False)
        sel = [i for i, n in enumerate(names) if stage_of(n) == st and full_gn[i] > 0]
        # the isolation itself: the stage alone, fed the recorded upstream gradient, IS the full step's backward through that stage
        rel = max(abs(float(gn[i]) - float(full_gn[i])) / (float(full_gn[i]) + 1e-12) for i in sel)
        reld = max(abs(float(gd[i]) - float(full_gd[i])) / (float(full_gn[i]) + 1e-12) for i in sel)
        print(f"{name} stage {st}: fp32 stage-alone vs full step, worst relative norm difference {rel:.2e}, probe {reld:.2e} over {len(sel)} tensors", flush=True)
        assert rel < 1e-4 and reld < 1e-4, (st, rel, reld)
        if st == "neck":
            for k in ("x2", "x3", "x4"):
                assert float((din[k] - grads[k]).abs().max()) <= 1e-5 * float(grads[k].abs().max()) + 1e-9
        if st == "decoder":
            for k in ("fq", "word_feat"):
                assert float((din[k] - grads[k]).abs().max()) <= 1e-4 * float(grads[k].abs().max()) + 1e-9
        if bf16:
            gnb, gdb, _, _ = run_stage(st, ins, cots, True)
            out[f"bf16::{st}::gnorm"], out[f"bf16::{st}::gdot"] = gnb, gdb
    if not maps:
        # full depth: image tower / neck / decoder on seeded inputs and cotangents (NCHW element order), nothing large to commit
        sh = meta_shapes
        syn = {"image": (dict(img=batch["img"]), [seeded_cotangent("d_" + k, tuple(sh[k]), seed) for k in ("x2", "x3", "x4")]),
               "neck": ({**{k: seeded_cotangent(k, tuple(sh[k]), seed, scale=1.0, relu=True) for k in ("x2", "x3", "x4")},
                         "state": seeded_cotangent("state", tuple(sh["state"]), seed, scale=1.0)}, [seeded_cotangent("d_fq", tuple(sh["fq"]), seed)]),
               "decoder": (dict(fq=seeded_cotangent("fq", tuple(sh["fq"]), seed, scale=1.0), word_feat=seeded_cotangent("word_feat", tuple(sh["word_feat"]), seed, scale=1.0)),
                           [seeded_cotangent("d_fq_dec", tuple(sh["fq_dec"]), seed)])}
        for st, (ins, cots) in syn.items():
            gn, gd, din, outs = run_stage(st, ins, cots, False)
            out[f"syn::{st}::gnorm"], out[f"syn::{st}::gdot"] = gn, gd
            for j, o in enumerate(outs):      # forward pins of the seeded stage: sums and a strided sample
                f = o.flatten()
                out[f"syn::{st}::out{j}::sums"] = torch.stack([f.double().sum(), f.double().abs().sum()])
                out[f"syn::{st}::out{j}::sample"] = f[::997].clone()
            for k, g in din.items():
                if k != "img":
                    out[f"syn::{st}::din::{k}::norm"] = g.norm()
                    out[f"syn::{st}::din::{k}::sample"] = g.flatten()[::997].clone()
            # the exact gradients (float64): the seeded inputs put many ReLU pre-activations within fp32 rounding of zero, and a gate that opens on
            # one side only moves the gradients downstream of it - the reference's OWN fp32 distance to float64 is the yardstick of the GPU test
            gn64, gd64, _, _ = run_stage(st, ins, cots, False, f64=True)
            out[f"syn64::{st}::gnorm"], out[f"syn64::{st}::gdot"] = gn64, gd64
            sel = [i for i, n in enumerate(names) if stage_of(n) == st and gn64[i] > 0]
            e = [abs(float(gd[i]) - float(gd64[i])) / float(gn64[i]) for i in sel]
            print(f"{name} seeded stage {st}: fp32 vs float64 probe distance median {np.median(e):.2e} p90 {np.quantile(e, 0.9):.2e} max {max(e):.2e}", flush=True)
            if bf16:
                gnb, gdb, _, _ = run_stage(st, ins, cots, True)
                out[f"synbf16::{st}::gnorm"], out[f"synbf16::{st}::gdot"] = gnb, gdb
                print(f"{name} seeded stage {st}: bf16 autocast done", flush=True)
    np.savez_compressed(os.path.join(GOLD, name + "_stages.npz"), **{k: v.detach().numpy() for k, v in out.items()})
    json.dump(dict(param_names=names, boundary_shapes=meta_shapes, seed=seed, B=B, residual_gain=residual_gain, maps=bool(maps)),
              open(os.path.join(GOLD, name + "_stages.json"), "w"))


def op_fixtures(ref_clip, ref_layers):
    """Per-op pins from the reference classes at small shapes (weights + input + output + input grad)."""
    torch.manual_seed(7)
    fx = {}

    def rec(tag, mod, *inputs, post=lambda o: o):
        mod.train()
        for p in mod.parameters():
            p.data.normal_(0, 0.2)
        for n, b in mod.named_buffers():
            if n.endswith("running_var"):
                b.data.uniform_(0.8, 1.2)
        sd0 = {k: v.clone() for k, v in mod.state_dict().items()}
        ins = [i.clone().requires_grad_(True) if i.is_floating_point() else i for i in inputs]
        o = post(mod(*ins))
        o.sum().backward() if o.dim() == 0 else (o * torch.linspace(-1, 1, o.numel()).view_as(o)).sum().backward()
        fx[tag + "::out"] = o.detach()
        for j, i in enumerate(ins):
            fx[f"{tag}::in{j}"] = i.detach()
            if i.is_floating_point() and i.grad is not None:
                fx[f"{tag}::din{j}"] = i.grad
        for k, v in sd0.items():
            fx[f"{tag}::w::{k}"] = v

    rec("bottleneck_s2", ref_clip.Bottleneck(64, 32, 2), torch.randn(2, 64, 8, 8))
    rec("bottleneck_s1", ref_clip.Bottleneck(128, 32, 1), torch.randn(2, 128, 6, 6))
    rec("attnpool", ref_clip.AttentionPool2d(7, 128, 2, 64), torch.randn(2, 128, 3, 3))
    mask = torch.full((6, 6), float("-inf")).triu_(1)
    rec("resblock", ref_clip.ResidualAttentionBlock(128, 2, mask), torch.randn(6, 3, 128))
    rec("declayer", ref_layers.TransformerDecoderLayer(128, 2, 256, 0.0),
        torch.randn(16, 2, 128), torch.randn(5, 2, 128), ref_layers.TransformerDecoder.pos2d(128, 4, 4),
        ref_layers.TransformerDecoder.pos1d(128, 5), torch.tensor([[False, False, False, True, True], [False] * 5]))
    rec("coordconv", ref_layers.CoordConv(32, 32, 3, 1), torch.randn(2, 32, 5, 7))
    rec("mtproj", ref_layers.MultiTaskProjector(64, 32, 3), torch.randn(2, 64, 4, 4), torch.randn(2, 64),
        post=lambda o: torch.cat(o, 1))
    rec("proj", ref_layers.Projector(64, 32, 3), torch.randn(2, 64, 4, 4), torch.randn(2, 64))
    fx["pos1d_128_9"] = ref_layers.TransformerDecoder.pos1d(128, 9).squeeze(1)
    fx["pos2d_128_5_7"] = ref_layers.TransformerDecoder.pos2d(128, 5, 7).squeeze(1)
    np.savez_compressed(os.path.join(GOLD, "ops.npz"), **{k: v.detach().numpy() for k, v in fx.items()})
    print("ops fixtures:", len(fx), flush=True)


def vit_fixture(ref_clip):
    """BASELINE config 4 (CLIP ViT tower), encoder-level pin at a small shape: VisionTransformer(64, 16, 128, 2 layers, 2 heads, 64)
    -> output, loss-weighted parameter gradients."""
    torch.manual_seed(11)
    m = ref_clip.VisionTransformer(64, 16, 128, 2, 2, 64).train()
    for n, p in m.named_parameters():
        if n.endswith("ln_pre.weight") or n.endswith("ln_post.weight") or ".ln_1.weight" in n or ".ln_2.weight" in n:
            p.data.normal_(1.0, 0.1)
        elif p.dim() == 1:
            p.data.normal_(0, 0.1)
        else:
            p.data.normal_(0, p.shape[-1] ** -0.5 if p.dim() == 2 else 0.04)
    fx = {"w::" + k: v.clone() for k, v in m.state_dict().items()}
    img = torch.randn(2, 3, 64, 64)
    o = m(img)
    (o * torch.linspace(-1, 1, o.numel()).view_as(o)).sum().backward()
    fx["in0"] = img
    fx["out"] = o.detach()
    for n, p in m.named_parameters():
        fx["dw::" + n] = p.grad
    np.savez_compressed(os.path.join(GOLD, "vit_tiny.npz"), **{k: v.detach().numpy() for k, v in fx.items()})
    print("vit fixture: out", tuple(o.shape), "absmax", float(o.abs().max()), flush=True)


def ssg_fixture(name, cfg, B, seed):
    """BASELINE config 5 (SSG-R50), trunk-level pin: the reference's own SSG sub-modules driven exactly as SSG.forward does
    (ssg.py:248-281) on a seeded state and batch; raw predictions, surrogate-loss gradients, BN running statistics, and the
    eval-mode output_dict (anchors, class softmax)."""
    import model.ssg as ref_ssg
    torch.manual_seed(0)
    m = ref_ssg.SSG(cfg)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(seeded_state(shapes, seed=seed))
    batch = synthetic_ssg_batch(B, cfg.img_size, cfg.with_depth, seed=1234 + seed)
    img = torch.cat([batch["rgb"], batch["depth"]], 1) if cfg.with_depth else batch["rgb"]
    m.train()
    x = m.backbone(img)
    x = m.fpn(x[1:4])
    protos = m.proto_net(x[0]).permute(0, 2, 3, 1).contiguous()
    per = [m.prediction_layers(a) for a in x]
    raw = {k: torch.cat([p[i] for p in per], dim=1) for i, k in enumerate(SSG_OUTPUTS[:4])}
    raw["protos"] = protos
    raw["seg_pred"] = m.semantic_seg_conv(x[0])
    loss = ssg_surrogate_loss(raw, seed)
    loss.backward()
    out = {k: v.detach() for k, v in raw.items()}
    out["loss"] = loss.detach()
    names = [n for n, _ in m.named_parameters()]
    out["grad_norms"] = torch.tensor([float(p.grad.norm()) for _, p in m.named_parameters()])
    for n, p in m.named_parameters():   # direction pins: the head of every gradient
        out["grad::" + n] = p.grad.flatten()[:64].clone()
    bn = {k: float(v.double().sum()) for k, v in m.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    out["bn_running_checksum"] = torch.tensor([bn[k] for k in sorted(bn)])
    # the reference's own compute_loss (ssg.py:297-530) on these predictions and synthetic ground truth: eight losses and
    # d(sum of losses)/d(prediction) for the six prediction tensors
    tg = synthetic_ssg_targets(B, cfg.img_size, cfg.num_classes, seed=1234 + seed)
    leaf = {k: raw[k].detach().clone().requires_grad_(True) for k in SSG_OUTPUTS}
    losses = m.compute_loss(leaf["class_pred"], leaf["box_pred"], leaf["ins_coef_pred"], leaf["grasp_coef_pred"], leaf["protos"],
                            leaf["seg_pred"], {**batch, **tg}, {})
    sum(losses.values()).backward()
    for k, v in losses.items():
        out["S2::" + k] = v.detach()
    for k in SSG_OUTPUTS:
        out["S2::d_" + k] = leaf[k].grad
    m.eval()
    with torch.no_grad():
        ev = m(batch)
    out["eval_cls_pred"] = ev["cls_pred"]
    out["eval_box_pred"] = ev["box_pred"]
    out["anchors"] = torch.tensor(ev["anchors"])
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **{k: v.detach().numpy() for k, v in out.items()})
    json.dump(dict(param_names=names, shapes={k: list(v) for k, v in shapes.items()}, seed=seed, B=B, bn_keys=sorted(bn),
                   cfg={k: v for k, v in vars(cfg).items()}), open(os.path.join(GOLD, name + ".json"), "w"))
    print(name, "loss", float(loss), "anchors", len(ev["anchors"]) // 4, flush=True)


def sampled(t, stride):
    """Fixed-stride sample of a large tensor + its sum / abs-sum (fixtures stay small; the test samples the same way)."""
    f = t.detach().flatten()
    return f[::stride].clone(), torch.stack([f.double().sum(), f.double().abs().sum()])


def vit_full_fixture(ref_clip):
    """BASELINE config 4 at full depth: the reference's VisionTransformer as CLIP ViT-B/16 builds it (clip.py:354-361:
    input_resolution 224, patch 16, width 768, 12 layers, 12 heads, output_dim 512), B = 2.  Weights come from the name-seeded
    recipe (86 M values are not committed).  Pinned: the full output, every parameter-gradient norm, the head of every gradient."""
    torch.manual_seed(0)
    m = ref_clip.VisionTransformer(224, 16, 768, 12, 12, 512).train()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    from crog_amd.testing import vit_seeded_state
    m.load_state_dict(vit_seeded_state(shapes, seed=21))
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(2101))
    o = m(img)
    w = torch.linspace(-1, 1, o.numel()).view_as(o)
    (o * w).sum().backward()
    fx = {"out": o.detach()}
    names = [n for n, _ in m.named_parameters()]
    fx["grad_norms"] = torch.tensor([float(p.grad.norm()) for _, p in m.named_parameters()])
    for n, p in m.named_parameters():
        fx["grad::" + n] = p.grad.flatten()[:64].clone()
    np.savez_compressed(os.path.join(GOLD, "vit_b16.npz"), **{k: v.detach().numpy() for k, v in fx.items()})
    json.dump(dict(param_names=names, shapes={k: list(v) for k, v in shapes.items()}, seed=21, img_seed=2101, B=2),
              open(os.path.join(GOLD, "vit_b16.json"), "w"))
    print("vit_b16 fixture: out", tuple(o.shape), "absmax", float(o.abs().max()), flush=True)


def ssg_full_fixture(name, cfg, B, seed, stride=29, residual_gain=0.25):
    """BASELINE config 5 at full depth (ssg_r50.yaml: ResNet-50 [3,4,6,3], 544 x 544, RGB-D): same protocol as ssg_fixture, with
    the large prediction tensors pinned through fixed-stride samples + (sum, abs-sum) instead of in full."""
    import model.ssg as ref_ssg
    torch.manual_seed(0)
    m = ref_ssg.SSG(cfg)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    # last BatchNorm scale of every Bottleneck damped (the usual zero-init-residual conditioning): a 16-block gain-1 random trunk
    # amplifies fp32 rounding past 1e-3 on its own (the reference's CPU result vs float64 included)
    m.load_state_dict(seeded_state(shapes, seed=seed, residual_gain=residual_gain))
    batch = synthetic_ssg_batch(B, cfg.img_size, cfg.with_depth, seed=1234 + seed)
    img = torch.cat([batch["rgb"], batch["depth"]], 1) if cfg.with_depth else batch["rgb"]
    m.train()
    x = m.backbone(img)
    x = m.fpn(x[1:4])
    protos = m.proto_net(x[0]).permute(0, 2, 3, 1).contiguous()
    per = [m.prediction_layers(a) for a in x]
    raw = {k: torch.cat([p[i] for p in per], dim=1) for i, k in enumerate(SSG_OUTPUTS[:4])}
    raw["protos"] = protos
    raw["seg_pred"] = m.semantic_seg_conv(x[0])
    loss = ssg_surrogate_loss(raw, seed)
    loss.backward()
    out = {"loss": loss.detach()}
    for k, v in raw.items():
        out["absmax::" + k] = v.detach().abs().max()
    for k, v in raw.items():
        out["sample::" + k], out["sums::" + k] = sampled(v, stride)
        out["shape::" + k] = torch.tensor(v.shape)
    names = [n for n, _ in m.named_parameters()]
    out["grad_norms"] = torch.tensor([float(p.grad.norm()) for _, p in m.named_parameters()])
    for n, p in m.named_parameters():
        out["grad::" + n] = p.grad.flatten()[:64].clone()
    bn = {k: float(v.double().sum()) for k, v in m.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    out["bn_running_checksum"] = torch.tensor([bn[k] for k in sorted(bn)])
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **{k: v.detach().numpy() for k, v in out.items()})
    json.dump(dict(param_names=names, shapes={k: list(v) for k, v in shapes.items()}, seed=seed, B=B, bn_keys=sorted(bn), stride=stride,
                   residual_gain=residual_gain, cfg={k: v for k, v in vars(cfg).items()}), open(os.path.join(GOLD, name + ".json"), "w"))
    print(name, "loss", float(loss), {k: tuple(v.shape) for k, v in raw.items()}, flush=True)


def clip_load_fixture(ref_clip):
    """Pretrained-CLIP load path (clip.py:477-556, crog.py:20-23): a synthetic checkpoint with CLIP key names (small RN
    architecture, no `attnpool.connect.*` keys, as real CLIP archives) goes through the reference's build_model(load_weights=True)
    and .float().  Recorded per key of the resulting module: 0 = the checkpoint value survived exactly, 1 = it equals the fp16
    round trip of the checkpoint value, 2 = not in the checkpoint (kept its random init) + whether that init is fp16-representable;
    plus the architecture the reference inferred."""
    from crog_amd.testing import clip_load_arch, synthetic_clip_checkpoint
    arch = clip_load_arch()
    sd = synthetic_clip_checkpoint(arch, seed=31)
    torch.manual_seed(5)
    model = ref_clip.build_model(dict(sd), 20, True).float()
    flags = {}
    for k, v in model.state_dict().items():
        if k not in sd:
            flags[k] = [2, bool(torch.equal(v, v.half().float()))]
        elif torch.equal(v, sd[k].float()) and not torch.equal(sd[k].float(), sd[k].half().float()):
            flags[k] = [0, False]
        elif torch.equal(v, sd[k].half().float()):
            flags[k] = [1, True]
        else:
            raise AssertionError(("unexpected transformation", k))
    v = model.visual
    inferred = dict(embed_dim=model.text_projection.shape[1], image_resolution=v.input_resolution,
                    vision_layers=[len(v.layer1), len(v.layer2), len(v.layer3), len(v.layer4)], vision_width=v.layer1[0].conv1.weight.shape[0],
                    context_length=model.context_length, vocab_size=model.vocab_size, transformer_width=model.transformer.width,
                    transformer_heads=model.transformer.resblocks[0].attn.num_heads, transformer_layers=model.transformer.layers,
                    training=model.training)
    json.dump(dict(flags=flags, arch=inferred, seed=31), open(os.path.join(GOLD, "clip_load.json"), "w"))
    print("clip_load fixture:", {f: sum(1 for x in flags.values() if x[0] == f) for f in (0, 1, 2)}, inferred, flush=True)


def ssg_loss_fixture(name, cfg, B, seed, stride=97):
    """Row S2 / N4 at the yaml's anchor set: the reference's own SSG.compute_loss (ssg.py:297-530) on seeded random predictions
    (the loss does not care where they came from) and ragged synthetic ground truth for B images -> eight losses + fixed-stride samples
    and sums of d(sum of losses)/d(prediction).  `cfg.masks_to_train` below the positives per image exercises the randperm branch;
    the CPU generator is seeded right before the call (the test does the same)."""
    import model.ssg as ref_ssg
    from crog_amd.testing import synthetic_ssg_predictions
    torch.manual_seed(0)
    m = ref_ssg.SSG(cfg)
    anchors = torch.tensor(m.anchors).reshape(-1, 4)
    preds = synthetic_ssg_predictions(B, anchors.shape[0], cfg, seed)
    batch = synthetic_ssg_batch(B, cfg.img_size, cfg.with_depth, seed=1234 + seed)
    tg = synthetic_ssg_targets(B, cfg.img_size, cfg.num_classes, seed=1234 + seed)
    leaf = {k: preds[k].clone().requires_grad_(True) for k in SSG_OUTPUTS}
    torch.manual_seed(4242 + seed)
    losses = m.compute_loss(leaf["class_pred"], leaf["box_pred"], leaf["ins_coef_pred"], leaf["grasp_coef_pred"], leaf["protos"],
                            leaf["seg_pred"], {**batch, **tg}, {})
    sum(losses.values()).backward()
    out = {"S2::" + k: v.detach() for k, v in losses.items()}
    for k in SSG_OUTPUTS:
        out["S2::d_" + k + "::sample"], out["S2::d_" + k + "::sums"] = sampled(leaf[k].grad, stride)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **{k: v.detach().numpy() for k, v in out.items()})
    json.dump(dict(seed=seed, B=B, stride=stride, anchors=int(anchors.shape[0]), cfg={k: v for k, v in vars(cfg).items()}),
              open(os.path.join(GOLD, name + ".json"), "w"))
    print(name, {k: round(float(v), 5) for k, v in losses.items()}, flush=True)


def ssg_detect_fixture(name, cfg, seed):
    """Row N4, detection side: the reference's own `fast_nms` and `crop` (utils/grasp_eval.py:54-94, utils/box_utils.py:150-169) driven
    as `ssg_post_processing` drives them (grasp_eval.py:99-150,168-186; batch size 1) on a seeded synthetic output_dict."""
    for mod, attrs in (("skimage", ()), ("skimage.draw", ("polygon",)), ("skimage.filters", ("gaussian",)), ("skimage.feature", ("peak_local_max",))):
        m_ = types.ModuleType(mod)
        for a in attrs:
            setattr(m_, a, None)
        sys.modules.setdefault(mod, m_)
    import utils.grasp_eval as ge
    import model.ssg as ref_ssg
    from crog_amd.testing import synthetic_ssg_output
    torch.manual_seed(0)
    anchors = torch.tensor(ref_ssg.SSG(cfg).anchors).reshape(-1, 4)
    od = synthetic_ssg_output(anchors, cfg, seed)
    protos, cls_pred, box_pred = od["protos"].squeeze(), od["cls_pred"].squeeze().transpose(1, 0).contiguous()[1:], od["box_pred"].squeeze()
    ins_coef, grasp_coef = od["ins_coef_pred"].squeeze(), od["grasp_coef_pred"].squeeze()
    keep = cls_pred.max(dim=0)[0] > cfg.nms_score_thre
    a, b = anchors[keep], box_pred[keep]
    dec = torch.cat((a[:, :2] + b[:, :2] * 0.1 * a[:, 2:], a[:, 2:] * torch.exp(b[:, 2:] * 0.2)), 1)
    dec[:, :2] -= dec[:, 2:] / 2
    dec[:, 2:] += dec[:, :2]
    dec = torch.clip(dec, min=0., max=1.)
    ids, sc, bx, ic, gc = ge.fast_nms(cfg, dec, cls_pred[:, keep], ins_coef[keep], grasp_coef[keep])
    ok = sc > 0.3
    assert bool(ok.any())
    ids, sc, bx, ic, gc = ids[ok], sc[ok], bx[ok], ic[ok], gc[ok]
    maps = dict(ins=torch.sigmoid(protos @ ic.t()), qua=torch.sigmoid(protos @ gc[:, 0].t()), sin=protos @ gc[:, 1].t(),
                cos=protos @ gc[:, 2].t(), wid=torch.sigmoid(protos @ gc[:, 3].t()))
    out = {"cls": ids + 1, "scores": sc, "bboxes": bx, "n_kept_before_nms": torch.tensor(int(keep.sum()))}
    for k, v in maps.items():
        v = ge.crop(v.contiguous(), bx).permute(2, 0, 1)
        out["map_sums::" + k] = v.double().sum((1, 2))
        out["map_sample::" + k] = v.flatten()[::211].clone()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **{k: v.detach().numpy() for k, v in out.items()})
    json.dump(dict(seed=seed, cfg={k: v for k, v in vars(cfg).items()}), open(os.path.join(GOLD, name + ".json"), "w"))
    print(name, "detections", int(ids.numel()), "of", int(keep.sum()), "score-filtered anchors", flush=True)



def rn_wide_fixture(ref_clip):
    """A width-128 ModifiedResNet (the RN50x64 family, clip.py:165-185,517-532: stem 64 / 64 / 128, stages 128 .. 1024 planes, 64 heads) at
    depth (1, 1, 1, 1) and 128 x 128 input, B = 2, training mode, name-seeded weights (not committed).  Pinned: the three returned maps,
    every parameter-gradient norm, the head of every gradient, the BatchNorm running statistics after the forward."""
    torch.manual_seed(0)
    m = ref_clip.ModifiedResNet((1, 1, 1, 1), 512, 64, input_resolution=128, width=128).train()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    from crog_amd.testing import vit_seeded_state
    m.load_state_dict(vit_seeded_state(shapes, seed=33))
    img = torch.randn(2, 3, 128, 128, generator=torch.Generator().manual_seed(3301))
    x2, x3, x4 = m(img)
    loss = sum((o * torch.linspace(-1, 1, o.numel()).view_as(o)).sum() for o in (x2, x3, x4))
    loss.backward()
    fx = {"x2": x2.detach(), "x3": x3.detach(), "x4": x4.detach()}
    names = [n for n, _ in m.named_parameters()]
    fx["grad_norms"] = torch.tensor([float(p.grad.norm()) for _, p in m.named_parameters()])
    for n, p in m.named_parameters():
        fx["grad::" + n] = p.grad.flatten()[:64].clone()
    for k, v in m.state_dict().items():
        if "running_" in k:
            fx["buf::" + k] = v.clone()
    np.savez_compressed(os.path.join(GOLD, "rn_wide.npz"), **{k: v.detach().numpy() for k, v in fx.items()})
    json.dump(dict(param_names=names, shapes={k: list(v) for k, v in shapes.items()}, seed=33, img_seed=3301, B=2),
              open(os.path.join(GOLD, "rn_wide.json"), "w"))
    print("rn_wide fixture: x2", tuple(x2.shape), "x3", tuple(x3.shape), "x4", tuple(x4.shape), "absmax", float(x4.abs().max()), flush=True)

def shapes_only(ref_clip):
    """Parameter names/shapes of the real CLIP RN50 and ViT-B/16 towers + CROG heads (names are the checkpoint contract)."""
    vit = ref_clip.CLIP(512, 224, 12, 768, 16, 77, 20, 49408, 512, 8, 12)
    json.dump({k: list(v.shape) for k, v in vit.state_dict().items()}, open(os.path.join(GOLD, "shapes_clip_vitb16.json"), "w"))


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref_model, ref_clip, ref_crog, ref_layers = import_reference()
    which = sys.argv[1:] or ["tiny", "ops", "vit", "ssg", "shapes", "full", "damped", "damped4", "vitfull", "ssgfull", "ssgloss", "ssgdet", "clipload", "rnwide"]
    if "tiny" in which:
        run_case("tiny_crog", tiny_cfg(), B=4, seed=3, ref_model=ref_model, ref_clip=ref_clip, store_intermediates=True)
        run_case("tiny_crog_nomask", tiny_cfg(use_grasp_masks=False), B=4, seed=4, ref_model=ref_model, ref_clip=ref_clip,
                 store_intermediates=False)
    if "ops" in which:
        op_fixtures(ref_clip, ref_layers)
    if "vit" in which:
        vit_fixture(ref_clip)
    if "ssg" in which:
        ssg_fixture("ssg_tiny_rgbd", ssg_tiny_cfg(), B=2, seed=6)
        ssg_fixture("ssg_tiny_rgb", ssg_tiny_cfg(with_depth=False), B=2, seed=7)
    if "ssgloss" in which:
        from crog_amd.testing import ssg_cfg
        ssg_loss_fixture("ssg_loss_b8", ssg_cfg(), B=8, seed=12)
        ssg_loss_fixture("ssg_loss_b8_limit", ssg_cfg(masks_to_train=12), B=8, seed=13)
    if "ssgdet" in which:
        from crog_amd.testing import ssg_cfg
        ssg_detect_fixture("ssg_detect", ssg_cfg(nms_score_thre=0.05, nms_iou_thre=0.5, top_k=200, max_detections=100), seed=14)
    if "vitfull" in which:
        vit_full_fixture(ref_clip)
    if "rnwide" in which:
        rn_wide_fixture(ref_clip)
    if "ssgfull" in which:
        from crog_amd.testing import ssg_cfg
        ssg_full_fixture("ssg_r50_rgbd", ssg_cfg(), B=2, seed=8)
    if "clipload" in which:
        clip_load_fixture(ref_clip)
    if "damped" in which:
        # BASELINE config 1 on reference-conditioned weights: every Bottleneck's last BatchNorm scale small (clip.py:402-408 zero-inits
        # them; 0.25 keeps the residual branches alive) -> the trunk does not amplify rounding, 1e-3 ABSOLUTE is the test's bound
        run_case("crog_r50_b2_damped", make_cfg(dropout=0.0), B=2, seed=9, ref_model=ref_model, ref_clip=ref_clip,
                 store_intermediates=False, residual_gain=0.25)
    if "damped4" in which:
        # the same model on FOUR samples: config 1's B = 2 puts neck.txt_proj's BatchNorm1d (layers.py:14-16) over two samples, which
        # amplifies fp32 rounding ~60x by itself (the reference's own fp32 logits sit 1.0-1.3e-3 from the float64 value there,
        # oracle/make_fp64.py); with four samples the reference is 1.7e-4 from exact and 1e-3 absolute is a meaningful bound
        run_case("crog_r50_b4_damped", make_cfg(dropout=0.0), B=4, seed=10, ref_model=ref_model, ref_clip=ref_clip,
                 store_intermediates=False, residual_gain=0.25)
    if "bf16ref" in which:
        # (not in the default list: ~10 minutes of emulated bf16 on the build container's CPU)
        run_case_bf16("tiny_crog", tiny_cfg(), B=4, seed=3, ref_model=ref_model, ref_clip=ref_clip)
        run_case_bf16("crog_r50_b4_damped", make_cfg(dropout=0.0), B=4, seed=10, ref_model=ref_model, ref_clip=ref_clip, residual_gain=0.25)
    if "stages" in which:
        # (not in the default list: the full-depth bf16-autocast stage runs take ~15 minutes of emulated bf16 on the build container's CPU)
        stage_fixture("tiny_crog", tiny_cfg(), B=4, seed=3, ref_model=ref_model, ref_clip=ref_clip, residual_gain=1.0, maps=True)
        stage_fixture("crog_r50_b4_damped", make_cfg(dropout=0.0), B=4, seed=10, ref_model=ref_model, ref_clip=ref_clip, residual_gain=0.25, maps=False)
    if "shapes" in which:
        shapes_only(ref_clip)
    if "full" in which:
        # BASELINE config 1: CROG-R50, 2 x 416x416 + 20 tokens, dropout 0 for determinism
        run_case("crog_r50_b2", make_cfg(dropout=0.0), B=2, seed=5, ref_model=ref_model, ref_clip=ref_clip, store_intermediates=False)


if __name__ == "__main__":
    main()
