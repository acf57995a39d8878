"""The step driver itself (engine/crog_engine.py:17-122 -> crog_amd/engine.py) and torch's gradient conventions on the flat store:
GradScaler-enabled + clip_grad_norm_ branch of `train_with_grasp` against torch.optim.Adam + clip_grad_norm_ on the same
gradients; the reference's statement order (forward -> optimizer.zero_grad() -> backward -> step) with a STOCK torch optimizer whose
zero_grad() sets .grad to None; gradient accumulation over two forward/backward pairs."""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crog_amd.testing import seeded_state, synthetic_batch, tiny_cfg  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")


def _meta():
    return json.load(open(os.path.join(GOLD, "tiny_crog.json")))


def _build(cfg, meta, gain=0.25):
    from crog_amd.model import build_crog
    model, groups = build_crog(cfg)
    model.load_state_dict(seeded_state({k: tuple(v) for k, v in meta["shapes"].items()}, seed=meta["seed"], residual_gain=gain))
    model = model.cuda()
    model.compute_dtype = torch.float32      # pinned: the comparison below is between two fp32 runs
    model.prepare().train()
    return model, groups


def _loader(cfg, n):
    """Batches in the collate format train_with_grasp unpacks (crog_engine.py:49-66): CPU tensors, masks [B, H, W]."""
    out = []
    for i in range(n):
        b = synthetic_batch(4, cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=500 + i)
        out.append(dict(img=b["img"], word_vec=b["word"], mask=b["mask"][:, 0],
                        grasp_masks={k: b[k][:, 0] for k in ("qua", "sin", "cos", "wid")}))
    return out


def test_train_with_grasp_gradscaler_and_clipping_match_torch_adam():
    from torch.optim.lr_scheduler import MultiStepLR
    from crog_amd.engine import train_with_grasp
    from crog_amd.optim import FusedAdam
    from crog_amd.runtime import RT
    meta = _meta()
    cfg = tiny_cfg()
    args = SimpleNamespace(print_freq=2, epochs=1, max_norm=1.0)
    loader = _loader(cfg, 3)

    # ---- side A: the driver under test, GradScaler ENABLED, max_norm > 0 -----------------------------------------------
    model, groups = _build(cfg, meta)
    LR = 1e-6     # Adam moves every weight by ~lr per step: small enough that steps 2 and 3 see (almost) the same model on both sides
    opt = FusedAdam(groups, lr=LR, store=model.store)
    sched = MultiStepLR(opt, milestones=[35], gamma=0.1)
    scaler = torch.amp.GradScaler("cuda", enabled=True)
    lines = []
    RT.manual_seed(5)
    train_with_grasp(loader, model, opt, sched, scaler, 1, args, log=lines.append)
    torch.cuda.synchronize()
    assert len(lines) == 2 and "Loss" in lines[0] and "IoU" in lines[0] and "Prec@50" in lines[1]      # print_freq = 2, 3 batches
    assert scaler.get_scale() == 65536.0 and opt._step == 3                                             # no overflow step was skipped
    got = {k: v.detach().clone() for k, v in model.named_parameters()}
    bn_a = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k}

    # ---- side B: same gradients (HIP forward/backward, unscaled), torch's own clip_grad_norm_ + torch.optim.Adam -----------
    model_b, _ = _build(cfg, meta)
    names = [k for k, _ in model_b.named_parameters()]
    ref_p = {k: torch.nn.Parameter(v.detach().clone()) for k, v in model_b.named_parameters()}
    ref_groups = [{"params": [ref_p[k] for k in names if k.startswith("backbone") and "positional_embedding" not in k]},
                  {"params": [ref_p[k] for k in names if not (k.startswith("backbone") and "positional_embedding" not in k)]}]
    ref_opt = torch.optim.Adam(ref_groups, lr=LR)
    norms = []
    for data in loader:
        gm = data["grasp_masks"]
        _, _, loss, _ = model_b(data["img"].cuda(), data["word_vec"].cuda(), data["mask"].cuda().unsqueeze(1), gm["qua"].cuda().unsqueeze(1),
                                gm["sin"].cuda().unsqueeze(1), gm["cos"].cuda().unsqueeze(1), gm["wid"].cuda().unsqueeze(1))
        model_b.zero_grad()             # set_to_none between forward and backward (crog_engine.py:77 with a stock optimizer)
        loss.backward()
        torch.cuda.synchronize()
        cur = dict(model_b.named_parameters())
        for k in names:      # (`logit_scale` is never used by the forward: its .grad stays None, as in torch)
            ref_p[k].grad = None if cur[k].grad is None else cur[k].grad.detach().clone()
        norms.append(float(torch.nn.utils.clip_grad_norm_([q for q in ref_p.values() if q.grad is not None], args.max_norm)))
        ref_opt.step()
        sd = model_b.state_dict()
        sd.update({k: ref_p[k].detach() for k in names})
        model_b.load_state_dict(sd)
    assert max(norms) > args.max_norm, norms            # the clipping branch really clipped
    # Adam's update is scale-free (m / sqrt(v)): an element whose gradient is at the fp32 summation-noise level gets +-lr on either
    # side with a coin flip, so the comparison is on the whole update vector (relative L2) and on the fraction of elements that
    # moved differently, not on the worst element
    init = seeded_state({k: tuple(v) for k, v in meta["shapes"].items()}, seed=meta["seed"], residual_gain=0.25)
    da = torch.cat([(got[k].cpu() - init[k]).flatten() for k in names])
    db = torch.cat([(ref_p[k].detach().cpu() - init[k]).flatten() for k in names])
    rel = float((da - db).norm() / db.norm())
    frac = float(((da - db).abs() > 0.5 * LR).float().mean())
    print(f"train_with_grasp (GradScaler on, max_norm 1.0) vs torch Adam + clip_grad_norm_: update vectors differ by {rel:.2e} (relative L2), "
          f"{100 * frac:.3f} % of elements by more than lr / 2; |update| {float(db.norm()):.2e}; gradient norms before clipping {['%.2f' % n for n in norms]}")
    assert rel < 5e-2 and frac < 1e-2, (rel, frac)
    for k, v in bn_a.items():
        assert torch.allclose(v, model_b.state_dict()[k], rtol=1e-4, atol=1e-5), k


def test_reference_statement_order_with_stock_torch_adam():
    """crog_engine.py:72-84 with torch.optim.Adam: forward -> optimizer.zero_grad() [sets .grad = None] -> backward -> step.
    The flat store must hand the optimizer fresh, linked gradients; the result equals FusedAdam's."""
    from crog_amd.optim import FusedAdam
    meta = _meta()
    cfg = tiny_cfg()
    b = {k: v.cuda() for k, v in synthetic_batch(4, cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=77).items()}

    def steps(make_opt, n=2):
        model, groups = _build(cfg, meta)
        opt = make_opt(model, groups)
        for _ in range(n):
            _, _, loss, _ = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
            opt.zero_grad()
            loss.backward()
            live = [p.grad is not None for _, p in model.named_parameters()]
            opt.step()
        torch.cuda.synchronize()
        return model, live

    m_t, live = steps(lambda m, g: torch.optim.Adam(g, lr=1e-3))
    names = [n for n, _ in m_t.named_parameters()]
    # every parameter the forward uses has a gradient again after backward; `logit_scale` (never used, clip.py:385) stays None, as in torch
    assert [n for n, ok in zip(names, live) if not ok] == ["backbone.logit_scale"]
    init = seeded_state({k: tuple(v) for k, v in meta["shapes"].items()}, seed=meta["seed"], residual_gain=0.25)
    moved = (m_t.state_dict()["neck.f1_v_proj.0.weight"].cpu() - init["neck.f1_v_proj.0.weight"]).abs().max().item()
    assert moved > 1e-4, "torch.optim.Adam silently skipped the parameters"
    m_f, _ = steps(lambda m, g: FusedAdam(g, lr=1e-3, store=m.store))
    da = torch.cat([(p.detach().cpu() - init[n]).flatten() for n, p in m_t.named_parameters()])
    db = torch.cat([(p.detach().cpu() - init[n]).flatten() for n, p in m_f.named_parameters()])
    rel = float((da - db).norm() / db.norm())
    print(f"torch.optim.Adam (set_to_none zero_grad) vs FusedAdam after 2 steps: update vectors differ by {rel:.2e} (relative L2)")
    assert rel < 2e-2


def test_gradients_accumulate_until_zero_grad():
    meta = _meta()
    cfg = tiny_cfg()
    model, _ = _build(cfg, meta)
    b = {k: v.cuda() for k, v in synthetic_batch(4, cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=78).items()}

    def fwd_bwd():
        sd = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
        _, _, loss, _ = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
        loss.backward()
        torch.cuda.synchronize()
        model.load_state_dict({**model.state_dict(), **sd})   # same BatchNorm running statistics for the second pass (not that they matter in train mode)
    model.store.zero_grad()
    fwd_bwd()
    g1 = model.store.G.clone()
    fwd_bwd()
    g2 = model.store.G.clone()
    rel = ((g2 - 2 * g1).norm() / (2 * g1).norm()).item()
    print(f"two forward/backward pairs without zero_grad: |G2 - 2 G1| / |2 G1| = {rel:.2e}")
    assert rel < 1e-4
    for n, p in model.named_parameters():       # every writer accumulates, including the norm layers' (sum, sum) vectors
        if n.endswith(("bn1.weight", "ln_final.weight", "norm.weight", "txt.bias")) and float(p.grad.norm()) > 0:
            o = model.store.off(p)
            assert torch.allclose(g2[o:o + p.numel()], 2 * g1[o:o + p.numel()], rtol=2e-3, atol=1e-6), n
    model.zero_grad()                           # set_to_none: the next backward REPLACES
    fwd_bwd()
    rel1 = ((model.store.G - g1).norm() / g1.norm()).item()
    assert rel1 < 1e-4 and all(p.grad is not None for n, p in model.named_parameters() if n != "backbone.logit_scale")


@pytest.mark.parametrize("capturable", [False, True])
def test_adam_chunks_stepped_inside_backward_see_the_final_gradients(capturable, monkeypatch):
    """FusedAdam.overlap_backward (crog_amd/optim.py; train_crog.py:119-121 + crog_engine.py:77-84 with max_norm 0): chunks of the
    flat buffer are stepped on the weight-gradient stream while backward is still running.  After the step, G still holds the
    gradients of that backward, so the update every parameter received can be recomputed from (P, m, v) before the step and G after
    it: a chunk stepped before its gradient was complete would not match."""
    from crog_amd.engine import train_step
    from crog_amd.optim import FusedAdam
    meta = _meta()
    cfg = tiny_cfg()
    model, groups = _build(cfg, meta)
    monkeypatch.setattr(FusedAdam, "CHUNK_ELEMS", 1 << 13)          # many chunks on the tiny model
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    opt = FusedAdam(groups, lr=lr, betas=(b1, b2), eps=eps, store=model.store, capturable=capturable)
    args = SimpleNamespace(max_norm=0.0)
    b = {k: v.cuda() for k, v in synthetic_batch(4, cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=31).items()}
    st = model.store
    for step in range(1, 5):
        torch.cuda.synchronize()
        P0 = st.P.clone()
        m0 = opt.m.clone() if opt.m is not None else torch.zeros_like(P0)
        v0 = opt.v.clone() if opt.v is not None else torch.zeros_like(P0)
        launched = opt.early_launches
        train_step(model, opt, None, b, args, autocast_dtype=None)
        torch.cuda.synchronize()
        G = st.G
        m1 = b1 * m0 + (1 - b1) * G
        v1 = b2 * v0 + (1 - b2) * G * G
        want = P0 - lr / (1 - b1 ** step) * m1 / ((v1 / (1 - b2 ** step)).sqrt() + eps)
        # the moments are linear / quadratic in the gradient the launch saw (tolerance relative to the two terms, not to their sum:
        # b1 m0 and (1 - b1) G may cancel)
        tol_m = 1e-5 * ((b1 * m0).abs() + ((1 - b1) * G).abs()) + 1e-12
        bad = [n for n, p in model.named_parameters()
               if bool(((opt.m - m1).abs() > tol_m)[st.off(p):st.off(p) + p.numel()].any())]
        assert not bad, (step, bad[:8])
        assert torch.allclose(opt.v.sqrt(), v1.sqrt(), rtol=1e-4, atol=1e-12), step
        err = float(((st.P - want).abs() / (lr + want.abs() * 1e-5)).max())
        assert err < 1e-3, (step, err)
        if step == 1:
            assert opt.early_launches == 0            # the first armed backward only records the announcement counts
        else:
            assert opt.early_launches - launched >= len(opt._chunks) // 2, (opt.early_launches - launched, len(opt._chunks))
    # a second backward before the step (gradient accumulation) is not a static step: refused loudly, not silently wrong
    opt.zero_grad()
    opt.overlap_backward()
    for _ in range(2):
        _, _, loss, _ = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
        loss.backward()
    with pytest.raises(RuntimeError, match="announced more often"):
        opt.step()


def test_overlapped_adam_leaves_the_bits_of_the_update_done_in_step():
    """Full-depth CROG-R50 bf16, deterministic mode (every side stream on), four optimizer steps: the overlapped update (chunks stepped
    inside backward the moment their last gradient is announced) must leave the SAME bits in the parameters and both Adam moments as the
    update done after backward (CROG_ADAM_OVERLAP=0) - a chunk stepped before its gradient was final, or a weight updated under a data
    gradient that still reads it, would show (scripts/adam_late_check.py prints the checksums; engine.train_step, crog_engine.py:77-84)."""
    import subprocess
    out = {}
    for tag, env in (("in step()", {"CROG_ADAM_OVERLAP": "0"}), ("overlapped", {})):
        e = {k: v for k, v in os.environ.items() if k not in ("CROG_ADAM_OVERLAP", "CROG_ADAM_LATE")}
        e.update(env)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "adam_late_check.py")], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("P ")][-1].split()
        out[tag] = (line[1], line[3], line[5], int(line[-1]))
    assert out["in step()"][:3] == out["overlapped"][:3], out
    assert out["in step()"][3] == 0 and out["overlapped"][3] >= 20, out


def test_decoder_return_intermediate_outputs():
    """layers.py:259-274: with return_intermediate the decoder returns the final-norm'd output of EVERY layer; the last entry is what
    the plain call returns.  CROG.forward fails on that list exactly as the reference does (crog.py:69)."""
    from crog_amd.model import build_crog
    meta = _meta()
    cfg = tiny_cfg(num_layers=2, intermediate=True)
    model, _ = build_crog(cfg)
    model = model.cuda()
    model.compute_dtype = torch.float32
    model.prepare().train()
    b = {k: v.cuda() for k, v in synthetic_batch(2, cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=79).items()}
    with pytest.raises(AttributeError):
        model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
    vis = model.backbone.image_features(b["img"], torch.float32)
    wfeat, state = model.backbone.text_features(b["word"], torch.float32)
    fq = model.neck(vis, state)
    pad = (b["word"] == 0).contiguous()
    outs = model.decoder(fq, wfeat, pad)
    assert isinstance(outs, list) and len(outs) == 2 and all(o.shape == fq.shape for o in outs)
    model.decoder.return_intermediate = False
    last = model.decoder(fq, wfeat, pad)
    assert torch.allclose(outs[-1], last, atol=1e-6)
    assert not torch.allclose(outs[0], last, atol=1e-3)
    # the shared final LayerNorm was applied twice: its gradient is the SUM of both uses
    model.decoder.return_intermediate = True
    model.store.zero_grad()
    o2 = model.decoder(fq.detach(), wfeat.detach(), pad)
    (o2[0].float().sum() + 2.0 * o2[1].float().sum()).backward()
    torch.cuda.synchronize()
    assert float(model.decoder.norm.bias.grad.sum()) == pytest.approx(3.0 * o2[0].numel(), rel=1e-4)     # 1 x (first use) + 2 x (second use), per element


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_decoder_residual_gradients_ride_gradslots_into_the_layernorm_backward(dtype):
    """layers.py:313-338: every `vis` of a decoder layer feeds a norm AND the residual add after the sub-layer.  With LN_GRAD_SLOTS the
    residual's gradient is added inside crog_ln_bwd (dxadd) instead of by an autograd accumulation pass: same gradients (fp32: the same
    fp32 sum; bf16: one rounding less), with dropout on (same seeds) and off."""
    import crog_amd.functional as Fn
    from crog_amd.model import build_crog
    from crog_amd.runtime import RT
    for p_drop in (0.0, 0.1):
        cfg = tiny_cfg(num_layers=2, dropout=p_drop)
        model, _ = build_crog(cfg)
        model = model.cuda()
        model.compute_dtype = dtype
        model.prepare().train()
        dec = model.decoder
        torch.manual_seed(5)
        fq0 = torch.randn(2, 6, 6, 512, device="cuda").to(dtype)
        txt0 = torch.randn(2, 12, 512, device="cuda").to(dtype)
        pad = torch.zeros(2, 12, dtype=torch.bool, device="cuda"); pad[:, 9:] = True
        wgt = torch.randn(2, 6, 6, 512, device="cuda")
        runs = {}
        for slots in (True, False):
            Fn.LN_GRAD_SLOTS = slots
            try:
                RT.manual_seed(1234)
                model.store.zero_grad()
                fq, txt = fq0.clone().requires_grad_(True), txt0.clone().requires_grad_(True)
                out = dec(fq, txt, pad)
                (out.float() * wgt).sum().backward()
                torch.cuda.synchronize()
                runs[slots] = (out.detach().float(), fq.grad.float().clone(), txt.grad.float().clone(), model.store.G.clone())
            finally:
                Fn.LN_GRAD_SLOTS = True
        a, b = runs[True], runs[False]
        assert torch.equal(a[0], b[0])
        tol = 1e-5 if dtype == torch.float32 else 3e-2
        for x, y, what in zip(a[1:], b[1:], ("d vis", "d txt", "parameter gradients")):
            rel = float((x - y).norm() / y.norm().clamp_min(1e-12))
            assert rel < tol, (what, p_drop, rel)
        assert float(a[1].abs().max()) > 0


def test_validate_with_grasp_is_a_callable_drop_in():
    """engine.validate_with_grasp (crog_engine.py:125-285): same arguments and return triple.  The device half (eval forward, sigmoid,
    bicubic align_corners resize) is checked against ATen on the model's own eval logits; the host half is injected (this image has no
    cv2 / skimage: in a deployment the reference's own functions are picked up) and the aggregation - per-sample IoU at 0.35,
    Pr@50..90, J@1 / J@5 counting - against a plain numpy restatement of the reference's loop."""
    import numpy as np
    import torch.nn.functional as F
    from types import SimpleNamespace
    from crog_amd.engine import validate_with_grasp
    from crog_amd.model import build_crog
    cfg = tiny_cfg()
    model, _ = build_crog(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(seeded_state(shapes, seed=21, residual_gain=0.25))
    model = model.cuda().prepare()
    model.compute_dtype = torch.float32
    S = cfg.input_size
    batches = []
    for i in range(2):
        b = synthetic_batch(3, S, cfg.word_len, cfg.clip_arch["vocab_size"], seed=300 + i)
        batches.append({"img": b["img"], "word_vec": b["word"], "mask": b["mask"].squeeze(1),
                        "grasp_masks": {k: b[k].squeeze(1) for k in ("qua", "sin", "cos", "wid")},
                        "inverse": [np.eye(2, 3, dtype=np.float32)] * 3, "ori_size": [(S, S)] * 3, "grasps": [[[1.0, 2.0, 30.0, 20.0, 0.0, 0]]] * 3})
    inverse = lambda img, mat, w, h: img                                  # identity warp: ori_size == input size
    detect = lambda q, s_, c, w_, n: ([[float(np.argmax(q) % q.shape[1]), float(np.argmax(q) // q.shape[1]), float(w_.max()) * 100, 20, 0.0]] * n, None)
    jacquard = lambda grasps, targets: int(len(grasps) == 5 or grasps[0][2] > 40.0)      # depends on n and on the width map
    lines = []
    iou, prec, J = validate_with_grasp(batches, model, 3, SimpleNamespace(epochs=50), inverse=inverse, detect=detect, jacquard=jacquard, log=lines.append)
    # restatement with ATen on the model's eval outputs
    model.eval()
    ious, j1 = [], []
    with torch.no_grad():
        for d in batches:
            pred, _ = model(d["img"].cuda(), d["word_vec"].cuda(), d["mask"].cuda().unsqueeze(1), *[d["grasp_masks"][k].cuda().unsqueeze(1) for k in ("qua", "sin", "cos", "wid")])
            ins = F.interpolate(torch.sigmoid(pred[0]), size=(S, S), mode="bicubic", align_corners=True).squeeze(1).cpu().numpy()
            wid = F.interpolate(torch.sigmoid(pred[4]), size=(S, S), mode="bicubic", align_corners=True).squeeze(1).cpu().numpy()
            for k in range(3):
                m, t = ins[k] > 0.35, d["mask"][k].numpy()
                ious.append(np.logical_and(m, t).sum() / (np.logical_or(m, t).sum() + 1e-6))
                j1.append(int(float(wid[k].max()) * 100 > 40.0))
    ious = np.array(ious)
    assert abs(iou - ious.mean()) < 1e-4
    for i, th in enumerate((0.5, 0.6, 0.7, 0.8, 0.9)):
        assert abs(prec[f"Pr@{(5 + i) * 10}"] - float((ious > th).mean())) < 1e-6
    assert J == [sum(j1) / 6, 1.0] and set(prec) == {"Pr@50", "Pr@60", "Pr@70", "Pr@80", "Pr@90"}
    assert lines and lines[0].startswith("Evaluation: Epoch=[3/50]  IoU=") and "J_index@5: 100.00" in lines[0]
    with pytest.raises(ImportError):                                            # no cv2 / reference utils in this image: the defaults say so
        validate_with_grasp(batches, model, 3, SimpleNamespace(epochs=50))
