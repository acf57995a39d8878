"""Whole-path parity on the MI355X: crog_amd.CROG (HIP kernels via the C ABI) against
 (a) golden fixtures captured from the reference itself (tests/golden, see oracle/make_golden.py), and
 (b) the CPU oracle (oracle/crog_oracle.py) run on the same seeded inputs.
Tolerance: 1e-3 absolute on fp32 outputs (BASELINE.json north_star); bf16 compute is checked against a
looser, documented bound."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crog_amd.testing import make_cfg, seeded_state, synthetic_batch, tiny_cfg  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")
NAMES = ["ins", "qua", "sin", "cos", "wid"]


def load_case(name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    return {k: torch.from_numpy(d[k]) for k in d.files}, meta


def build(cfg, meta, dtype=torch.float32):
    from crog_amd.model import build_crog
    model, groups = build_crog(cfg)
    model.load_state_dict(seeded_state({k: tuple(v) for k, v in meta["shapes"].items()}, seed=meta["seed"]))
    model = model.cuda()
    model.compute_dtype = dtype
    model.prepare()
    return model, groups


def batch_for(cfg, meta):
    b = synthetic_batch(meta["B"], cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=1234 + meta["seed"])
    return {k: v.cuda() for k, v in b.items()}


def err(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


def nchw(x):
    return x.permute(0, 3, 1, 2)


def test_tiny_fp32_matches_reference_fixture():
    g, meta = load_case("tiny_crog")
    cfg = tiny_cfg()
    model, _ = build(cfg, meta)
    b = batch_for(cfg, meta)
    model.train()
    # stage-wise (BN running stats get an extra update here, reloaded below — same protocol as the fixture)
    report = {}
    x2, x3, x4 = model.backbone.image_features(b["img"], torch.float32)
    report["x2"], report["x3"], report["x4"] = err(nchw(x2), g["x2"]), err(nchw(x3), g["x3"]), err(nchw(x4), g["x4"])
    wfeat, state = model.backbone.text_features(b["word"], torch.float32)
    report["word_feat"], report["state"] = err(wfeat, g["word_feat"]), err(state, g["state"])
    fq = model.neck((x2, x3, x4), state)
    report["fq"] = err(nchw(fq), g["fq"])
    fqd = model.decoder(fq, wfeat, (b["word"] == 0).contiguous())
    report["fq_dec"] = err(nchw(fqd).reshape(g["fq_dec"].shape), g["fq_dec"])
    print("stage errors:", {k: f"{v:.2e}" for k, v in report.items()})
    for k, v in report.items():
        assert v < 1e-3, (k, v, report)
    model.load_state_dict(seeded_state({k: tuple(v) for k, v in meta["shapes"].items()}, seed=meta["seed"]))
    preds, tgts, loss, loss_dict = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
    loss.backward()
    torch.cuda.synchronize()
    for i, nm in enumerate(NAMES):
        assert err(preds[i], g["pred_" + nm]) < 1e-3, (nm, err(preds[i], g["pred_" + nm]))
        assert err(tgts[i], g["tgt_" + nm]) == 0
    assert abs(float(loss) - float(g["loss_total"])) < 1e-4
    items = [loss_dict[k] for k in ("m_ins", "m_qua", "m_sin", "m_cos", "m_wid")]
    assert np.allclose(items, g["loss_items"].numpy(), atol=1e-4)
    # gradients: norms for every parameter, full tensors for the pinned ones
    params = dict(model.named_parameters())
    gn = torch.tensor([float(params[n].grad.norm()) for n in meta["param_names"]])
    ref = g["grad_norms"]
    ref0 = torch.where(ref < 0, torch.zeros_like(ref), ref)  # reference: None grad (logit_scale) == our zero grad
    # neck.txt_proj's BatchNorm1d normalises over only B = 4 samples: its backward amplifies summation-order noise to ~2 %
    loose = torch.tensor(["txt_proj" in n for n in meta["param_names"]])
    trunk = torch.tensor([n.startswith("backbone.visual") for n in meta["param_names"]])  # upstream of ReLU knife-edge flips
    tol = torch.where(loose, torch.tensor(4e-2), torch.where(trunk, torch.tensor(2e-2), torch.tensor(5e-3)))
    bad = (gn - ref0).abs() > tol * ref0.abs() + 2e-5
    assert not bad.any(), [(meta["param_names"][i], float(gn[i]), float(ref0[i])) for i in bad.nonzero().flatten()[:8]]
    # Full gradient tensors.  fp32 noise floor of this network is ~1e-3 relative (CPU fp32 vs fp64, scripts/debug_grads.py);
    # a single ReLU whose pre-activation sits within 1e-7 of zero flips between implementations and moves every
    # gradient upstream of it by ~1e-2 relative (scripts/debug_l4.py) — hence 3e-2 upstream, 5e-3 for the head-side tensors.
    for k in g:
        if k.startswith("grad::"):
            r = g[k]
            a = params[k[6:]].grad.detach().cpu()
            rel = float((a - r).norm() / r.norm())
            lim = 5e-3 if k[6:].startswith(("proj.", "decoder.")) else 3e-2
            assert rel < lim, (k, rel)
    chk = torch.tensor([float(model.state_dict()[k].double().sum()) for k in meta["bn_keys"]])
    assert torch.allclose(chk, g["bn_running_checksum"].float(), rtol=1e-4, atol=1e-3)
    # eval mode (fp32, no autocast — crog_engine.py:166)
    model.eval()
    with torch.no_grad():
        ev = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
    for i, nm in enumerate(NAMES):
        assert err(ev[0][i], g["eval_pred_" + nm]) < 1e-3, nm
        assert ev[1][i] is b[["mask", "qua", "sin", "cos", "wid"][i]]
    # the eval forward above folds BatchNorm into the convolution weights (functional._conv_bn_act_eval); the unfolded form (conv -> z ->
    # scale / shift pass) must give the same logits up to fp32 rounding, in fp32 and (looser) in bf16
    from crog_amd import functional as Fn
    for dtype, lim in ((torch.float32, 2e-5), (torch.bfloat16, 0.15)):
        model.compute_dtype = dtype
        outs = []
        for fold in (True, False):
            Fn.EVAL_BN_FOLD = fold
            try:
                with torch.no_grad():
                    outs.append(torch.cat([o.float() for o in model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])[0]], 1))
            finally:
                Fn.EVAL_BN_FOLD = True
        d = float((outs[0] - outs[1]).abs().max())
        print(f"eval forward, BatchNorm folded vs unfolded ({dtype}): max |dlogit| = {d:.2e} (logit scale {float(outs[1].abs().max()):.1f})")
        assert d < lim * max(1.0, float(outs[1].abs().max())), (dtype, d)
    model.compute_dtype = None


def test_tiny_nomask_variant():
    g, meta = load_case("tiny_crog_nomask")
    cfg = tiny_cfg(use_grasp_masks=False)
    model, _ = build(cfg, meta)
    b = batch_for(cfg, meta)
    model.train()
    preds, tgts, loss, loss_dict = model(b["img"], b["word"], b["mask"])
    loss.backward()
    assert err(preds[0], g["pred_ins"]) < 1e-3 and preds[1] is None and tgts[1] is None
    assert abs(float(loss) - float(g["loss_total"])) < 1e-4 and loss_dict["m_qua"] == 0
    model.eval()
    with torch.no_grad():
        p, m = model(b["img"], b["word"], b["mask"])
    assert err(p, g["eval_pred_ins"]) < 1e-3 and m is b["mask"]


def test_eval_maps_of_validate_with_grasp():
    """Device part of validate_with_grasp (crog_engine.py:163-211): eval forward + sigmoid + bicubic resize to the input size,
    from the reference's own eval logits (fixture) through the oracle, against engine.eval_maps on the HIP path."""
    from crog_amd.engine import eval_maps, mask_iou
    from oracle import crog_oracle as O
    g, meta = load_case("tiny_crog")
    cfg = tiny_cfg()
    model, _ = build(cfg, meta)
    b = batch_for(cfg, meta)
    model.train()
    with torch.no_grad():   # the fixture's eval logits follow ONE training forward (BatchNorm running statistics updated once)
        model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
    maps, target = eval_maps(model, b)
    assert model.training and len(maps) == 5 and target[0] is b["mask"]
    logits = torch.cat([g["eval_pred_" + nm] for nm in NAMES], 1)
    want = O.eval_maps(logits, (0, 1, 4), tuple(b["img"].shape[-2:]))
    for i, nm in enumerate(NAMES):
        assert tuple(maps[i].shape) == (meta["B"],) + tuple(b["img"].shape[-2:])
        assert err(maps[i], want[:, i]) < 1e-3, nm
    iou = mask_iou(maps[0], b["mask"])
    p = want[:, 0] > 0.35
    t = b["mask"].cpu().reshape(p.shape) > 0.5
    ref_iou = (p & t).flatten(1).sum(1) / ((p | t).flatten(1).sum(1) + 1e-6)
    assert (iou.cpu() - ref_iou).abs().max().item() < 2e-2   # a logit within 1e-3 of the 0.35 threshold may flip a pixel


def test_checkpoint_wire_format_round_trips_with_torch_adam(tmp_path):
    """SURVEY.md §8f N2 (train_crog.py:206-226,245-267): a checkpoint written by crog_amd resumes a torch.optim.Adam /
    MultiStepLR over the same named parameters (what the reference holds), a `module.`-prefixed checkpoint written by that side
    resumes crog_amd, and both continue in lock-step."""
    from torch.optim.lr_scheduler import MultiStepLR
    from crog_amd.checkpoint import KEYS, load_checkpoint, save_checkpoint
    from crog_amd.optim import FusedAdam
    g, meta = load_case("tiny_crog")
    cfg = tiny_cfg()
    b = batch_for(cfg, meta)

    def fresh():
        model, groups = build(cfg, meta)
        model.train()
        opt = FusedAdam(groups, lr=1e-3, weight_decay=0.0, store=model.store)
        return model, groups, opt, MultiStepLR(opt, milestones=[1, 3], gamma=0.1)

    def backward(model):
        _, _, loss, _ = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
        model.zero_grad()       # torch default set_to_none=True, BETWEEN forward and backward as crog_engine.py:77 does
        loss.backward()
        torch.cuda.synchronize()

    model, groups, opt, sched = fresh()
    for _ in range(2):
        backward(model)
        opt.step()
    sched.step()
    path = str(tmp_path / "last_model.pth")
    save_checkpoint(path, model, opt, sched, epoch=1, cur_iou=0.25, best_iou=0.5, best_j_index=0.125, prec={"Pr@50": 0.5}, j_index=[0.1, 0.2])
    blob = torch.load(path, weights_only=False)
    assert tuple(blob.keys()) == KEYS and blob["optimizer"]["param_groups"][0]["initial_lr"] == cfg.lr_multi * cfg.base_lr

    # the reference's side of the wire: plain parameters in build_crog's group order, torch.optim.Adam, MultiStepLR
    names = [k for k, _ in model.named_parameters()]
    ref_p = {k: torch.nn.Parameter(blob["state_dict"][k].clone().cuda()) for k in names}
    ref_groups = [{"params": [ref_p[k] for k in names if k.startswith("backbone") and "positional_embedding" not in k],
                   "initial_lr": cfg.lr_multi * cfg.base_lr},
                  {"params": [ref_p[k] for k in names if not (k.startswith("backbone") and "positional_embedding" not in k)],
                   "initial_lr": cfg.base_lr}]
    ref_opt = torch.optim.Adam(ref_groups, lr=1e-3, weight_decay=0.0)
    ref_sched = MultiStepLR(ref_opt, milestones=[1, 3], gamma=0.1)
    ref_opt.load_state_dict(blob["optimizer"])
    ref_sched.load_state_dict(blob["scheduler"])
    assert [gr["lr"] for gr in ref_opt.param_groups] == [gr["lr"] for gr in opt.param_groups] == [1e-4, 1e-4]
    mine = dict(model.named_parameters())
    for k in names[::37]:
        assert torch.equal(ref_opt.state[ref_p[k]]["exp_avg"], opt.state[mine[k]]["exp_avg"]), k
        assert float(ref_opt.state[ref_p[k]]["step"]) == 2.0

    def lockstep(model, opt):
        backward(model)
        cur_p = dict(model.named_parameters())
        for k in names:      # (`logit_scale` is never used by the forward: its .grad stays None, as in torch)
            ref_p[k].grad = None if cur_p[k].grad is None else cur_p[k].grad.detach().clone()
        opt.step()
        ref_opt.step()
        cur = dict(model.named_parameters())
        worst = max((cur[k].detach() - ref_p[k].detach()).abs().max().item() for k in names)
        assert worst < 1e-6, worst

    lockstep(model, opt)

    # the other direction: a checkpoint as the reference writes it (DDP wrapper -> `module.` keys) resumes a fresh crog_amd model
    path2 = str(tmp_path / "ref_model.pth")
    sd = {"module." + k: v.detach().clone() for k, v in ref_p.items()}
    sd.update({"module." + k: v.clone() for k, v in model.state_dict().items() if k not in ref_p})   # BatchNorm buffers
    torch.save({"epoch": 2, "cur_iou": 0.3, "best_iou": 0.5, "best_j_index": 0.125, "prec": {}, "j_index": [0, 0], "state_dict": sd,
                "optimizer": ref_opt.state_dict(), "scheduler": ref_sched.state_dict()}, path2)
    model2, _, opt2, sched2 = fresh()
    info = load_checkpoint(path2, model2, opt2, sched2, map_location="cuda")
    assert info["epoch"] == 2 and info["best_iou"] == 0.5 and sched2.last_epoch == ref_sched.last_epoch
    assert opt2._step == 3 and [gr["lr"] for gr in opt2.param_groups] == [1e-4, 1e-4]
    for k in names[::37]:
        assert torch.equal(dict(model2.named_parameters())[k].detach(), ref_p[k].detach()), k
    lockstep(model2, opt2)


def test_store_bookkeeping_across_fused_adam_steps():
    """The flat store's two shortcuts stay sound over real steps: FusedAdam refreshes the bf16 shadow in the pass that updates the
    fp32 parameters (no cast launch in the next forward), and zero_grad skips the memset only while the gradient buffer is clean."""
    from crog_amd.optim import FusedAdam
    g, meta = load_case("tiny_crog")
    cfg = tiny_cfg()
    model, groups = build(cfg, meta, dtype=torch.bfloat16)
    model.train()
    b = batch_for(cfg, meta)
    opt = FusedAdam(groups, lr=1e-3, store=model.store)
    st = model.store
    for step in range(3):
        _, _, loss, _ = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
        if step:
            assert not st.g_clean and st.G.abs().max().item() > 0       # the forward does NOT clear gradients (torch semantics)
            assert torch.equal(st.S, st.P.to(torch.bfloat16))          # the shadow this forward used == cast of the stepped parameters
        opt.zero_grad()                                                 # crog_engine.py:77 — the one memset of the step
        assert st.g_clean and st.G.abs().max().item() == 0
        opt.zero_grad()                                                 # clean buffer: nothing launched
        assert st.g_clean
        loss.backward()
        torch.cuda.synchronize()
        assert not st.g_clean and st.G.abs().max().item() > 0
        before = st.P.clone()
        opt.step()
        torch.cuda.synchronize()
        assert not torch.equal(before, st.P) and torch.equal(st.S, st.P.to(torch.bfloat16)) and st.shadow_fresh
    # a foreign parameter edit must be announced: load_state_dict does it, and the next forward re-casts
    model.load_state_dict(seeded_state({k: tuple(v) for k, v in meta["shapes"].items()}, seed=meta["seed"]))
    assert not st.shadow_fresh
    model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
    assert torch.equal(st.S, st.P.to(torch.bfloat16))


def test_text_tower_hip_graph_replays_match_eager(monkeypatch):
    """CROG_TEXT_GRAPH=1: text tower forward/backward as hipGraph replays (crog_amd/graphs.py) == the eager launches,
    step after step (static buffers are overwritten in place), for outputs, loss and every parameter gradient."""
    import crog_amd.model.crog as crog_mod
    g, meta = load_case("tiny_crog")
    cfg = tiny_cfg()
    model, _ = build(cfg, meta)
    model.train()
    batches = [batch_for(cfg, meta)]
    b2 = {k: v.clone() for k, v in batches[0].items()}
    b2["word"] = torch.roll(b2["word"], 1, 0)
    batches.append(b2)

    def run(flag):
        monkeypatch.setattr(crog_mod, "TEXT_GRAPH", flag)
        res = []
        for b in batches + batches[:1]:
            # dropout seeds must line up between the two runs
            from crog_amd.runtime import RT
            RT.manual_seed(7)
            model.store.zero_grad()
            preds, _, loss, _ = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
            loss.backward()
            torch.cuda.synchronize()
            res.append((preds[0].clone(), float(loss), model.store.G.clone()))
        return res
    eager, again, graphed = run(False), run(False), run(True)
    assert "text" in model._graphs
    # the yardstick is the eager path's own run-to-run noise: BatchNorm statistics and split-K sums are fp32 atomics, and the tiny
    # model's 1x1 / 2x2 BatchNorm layers amplify their ordering noise to ~2e-4 on the logits and ~1 % on the stem gradients
    for (p0, l0, g0), (p1, l1, g1), (p2, l2, g2) in zip(eager, again, graphed):
        assert err(p0, p2) < max(1e-3, 4 * err(p0, p1)) and abs(l0 - l2) < 1e-4
        rel = lambda a, b: ((a - b).norm() / a.norm()).item()    # max-norm differences of the amplified noise are heavy-tailed
        assert rel(g0, g2) <= max(4 * rel(g0, g1), 2e-2)
    assert err(graphed[0][0], g["pred_ins"]) < 1e-3


def test_bf16_path_tracks_fp32_path():
    """bf16 storage/compute (the benchmark dtype) against the already-pinned fp32 HIP path on the same inputs.
    The comparison uses a damped-residual trunk (crog_amd.testing.seeded_state docstring): with the chaotic gain-1
    random trunk even fp32 rounding is amplified ~5000x, so bf16 (8-bit mantissa) would measure the weights, not the kernels.
    Measured on MI355X (scripts/debug_bf16.py): trunk features 1-3 %, logits ~9 % relative RMS at B=16, 160x160."""
    from crog_amd.model import build_crog
    cfg = tiny_cfg(input_size=160)
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        model, _ = build_crog(cfg)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(seeded_state(shapes, seed=3, residual_gain=0.25))
        model = model.cuda()
        model.compute_dtype = dt
        model.prepare()
        model.train()
        b = {k: v.cuda() for k, v in synthetic_batch(16, 160, cfg.word_len, cfg.clip_arch["vocab_size"], seed=77).items()}
        preds, tgts, loss, loss_dict = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
        loss.backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        outs[dt] = (torch.cat(preds, 1).float(), float(loss.detach()), grads)
        del model
    p32, l32, g32 = outs[torch.float32]
    p16, l16, g16 = outs[torch.bfloat16]
    rel = float((p16 - p32).pow(2).mean().sqrt() / p32.pow(2).mean().sqrt())
    print("bf16 logits rel rms", rel, "loss", l16, l32)
    assert rel < 0.2
    assert abs(l16 - l32) < 0.03 * abs(l32)
    cos = []
    for n in g32:
        a, b_ = g16[n].flatten().double(), g32[n].flatten().double()
        if b_.norm() > 1e-3:
            cos.append(float(torch.dot(a, b_) / (a.norm() * b_.norm() + 1e-30)))
    cos = torch.tensor(cos)
    print("bf16 grad cosine: min %.4f median %.4f" % (cos.min(), cos.median()))
    # measured: median 0.94, min 0.80 (trunk tensors, which sit behind ~60 bf16 layers); head/decoder tensors are > 0.99
    assert cos.median() > 0.9 and cos.min() > 0.7


def test_autocast_selects_bf16_and_state_dict_roundtrip():
    g, meta = load_case("tiny_crog")
    cfg = tiny_cfg()
    model, groups = build(cfg, meta)
    model.compute_dtype = None
    b = batch_for(cfg, meta)
    model.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
    assert out[0][0].dtype == torch.float32 and model.store.S is not None
    sd = model.state_dict()
    ref = seeded_state({k: tuple(v) for k, v in meta["shapes"].items()}, seed=meta["seed"])
    for k in ("backbone.visual.layer1.0.conv2.weight", "neck.coordconv.0.conv1.0.weight", "proj.txt.bias"):
        assert torch.equal(sd[k].cpu(), ref[k]), k
    assert len(groups[0]["params"]) == meta["group_backbone"] and len(groups[1]["params"]) == meta["group_head"]
    with pytest.raises(RuntimeError):
        model(b["img"], b["word"][:, :5], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])  # token count != word_len


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "crog_r50_b2.npz")), reason="full fixture missing")
def test_config1_crog_r50_fp32_matches_reference():
    """BASELINE config 1: CROG-R50, 2 x 416x416 + 20 tokens; fp32; outputs within 1e-3 of the reference's CPU path."""
    g, meta = load_case("crog_r50_b2")
    cfg = make_cfg(dropout=0.0)
    model, _ = build(cfg, meta)
    b = batch_for(cfg, meta)
    model.train()
    preds, tgts, loss, loss_dict = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
    loss.backward()
    errs = [err(preds[i], g["pred_" + nm]) for i, nm in enumerate(NAMES)]
    mags = [float(g["pred_" + nm].abs().max()) for nm in NAMES]
    print("config-1 pred errs", errs, "max |logit|", mags)
    # The reference's own fp32 CPU logits sit 3.1e-3..3.8e-3 (max abs) from the exact (fp64) value on this input
    # (tests/golden/crog_r50_b2_fp64.npz, oracle in float64): logits reach +-12 and the random gain-1 trunk amplifies
    # rounding ~5000x.  1e-3 is therefore applied relative to the logit scale, and the HIP path must stay within
    # 1.5x of the reference's own distance to the exact result (3x until round 3: the fp32 GEMMs now accumulate k-blocked, partial sums
    # of 128 as a blocked CPU GEMM forms them, instead of one sequential-k chain).
    for e, m in zip(errs, mags):
        assert e < 1e-3 * max(1.0, m), (errs, mags)
    t64 = np.load(os.path.join(GOLD, "crog_r50_b2_fp64.npz"))
    # the conditioning of this input, measured (as in tests/test_fulldepth_gpu.py): what a last-bit perturbation of the images moves the logits by.
    # Until round 5 the fp32 forward was not bit-reproducible on layers of more than 32768 rows (atomic slab reductions) and this test drew
    # 3.7 ... 5.4e-3 for `ins` from run to run against a bound of 5.2e-3; it is reproducible now, and the bound says what it depends on.
    keep = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
    spread = [0.0] * len(NAMES)
    with torch.no_grad():
        for k in range(3):
            gen = torch.Generator(device="cuda").manual_seed(77 + k)
            img_p = b["img"] * (1.0 + 6e-8 * torch.randn(b["img"].shape, device="cuda", generator=gen).sign())
            pp, _, _, _ = model(img_p, b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
            spread = [max(spread[i], err(pp[i], preds[i])) for i in range(len(NAMES))]
    model.load_state_dict({**model.state_dict(), **keep})
    for i, nm in enumerate(NAMES):
        truth = torch.from_numpy(t64["pred_" + nm])
        e_hip = float((preds[i].double().cpu() - truth).abs().max())
        e_ref = float((g["pred_" + nm].double() - truth).abs().max())
        print(f"  {nm}: distance to the float64 result: HIP {e_hip:.2e}, reference fp32 {e_ref:.2e}; a last-bit input perturbation moves the HIP logits by {spread[i]:.2e}")
        assert e_hip < 1.5 * e_ref + 2.0 * spread[i], (nm, e_hip, e_ref, spread[i])
    assert abs(float(loss.detach()) - float(g["loss_total"])) < 1e-3   # logits carry ~6e-3 of amplified fp32 noise (see above)
    params = dict(model.named_parameters())
    gn = torch.tensor([float(params[n].grad.norm()) for n in meta["param_names"]])
    ref = torch.where(g["grad_norms"] < 0, torch.zeros_like(g["grad_norms"]), g["grad_norms"])
    # B = 2 makes BatchNorm1d (neck.txt_proj) backward ill-conditioned (SURVEY §8c note in tests/test_oracle_golden.py):
    # the text-side gradients are compared loosely, the image/neck/head side tightly
    names = meta["param_names"]
    text_side = torch.tensor([("transformer" in n or "token_embedding" in n or "text_projection" in n or "ln_final" in n
                               or n == "backbone.positional_embedding" or "txt_proj" in n) for n in names])
    rel = ((gn - ref).abs() / (ref + 1e-6))[~text_side]
    print("config-1 worst relative gradient-norm error, image / neck / decoder / head side: %.2e" % float(rel.max()))
    tight = ((gn - ref).abs() > 3e-2 * ref + 2e-5) & ~text_side  # ReLU knife-edge flips move trunk gradients by a few % (see tiny test)
    assert not tight.any(), [(names[i], float(gn[i]), float(ref[i])) for i in tight.nonzero().flatten()[:8]]


def test_vit_tower_matches_reference_fixture():
    """SURVEY §8a row V / BASELINE config 4: the CLIP ViT image tower on the HIP path, encoder-level parity (the reference cannot
    run CROG end to end with a ViT backbone).  fp32 output within 1e-3 of the reference fixture; parameter gradients within
    1e-3 relative to each gradient's scale."""
    from crog_amd.model.blocks import bind_all
    from crog_amd.model.clip import VisionTransformer
    from crog_amd.runtime import ParamStore
    d = np.load(os.path.join(GOLD, "vit_tiny.npz"))
    fx = {k: torch.from_numpy(d[k]) for k in d.files}
    vit = VisionTransformer(64, 16, 128, 2, 2, 64)
    vit.load_state_dict({k[3:]: v for k, v in fx.items() if k.startswith("w::")})
    store = ParamStore(vit, torch.device("cuda"))
    bind_all(vit, store)
    vit.train()
    store.zero_grad()
    out = vit(fx["in0"].cuda(), torch.float32)
    assert tuple(out.shape) == tuple(fx["out"].shape)
    e = err(out, fx["out"])
    assert e < 1e-3, f"vit out err {e}"
    w = torch.linspace(-1, 1, out.numel(), device="cuda").view_as(out)
    (out * w).sum().backward()
    torch.cuda.synchronize()
    worst = 0.0
    for n, p in vit.named_parameters():
        ref = fx["dw::" + n]
        scale = ref.abs().max().item() + 1e-6
        worst = max(worst, err(p.grad, ref) / scale)
        assert err(p.grad, ref) / scale < 1e-3, f"vit grad {n}: {err(p.grad, ref)} vs scale {scale}"
    # wrong input resolution: same failure class as the reference's positional add (RuntimeError)
    with pytest.raises(RuntimeError):
        vit(torch.zeros(1, 3, 32, 32, device="cuda"), torch.float32)
    # bf16 compute follows fp32 (cosine of the outputs)
    store.invalidate_shadow()
    ob = vit(fx["in0"].cuda(), torch.bfloat16).float().flatten()
    cos = torch.nn.functional.cosine_similarity(ob, out.detach().flatten(), dim=0).item()
    assert cos > 0.999, cos


def test_batchnorm_backward_first_pass_in_the_dgrad_epilogue_matches_the_two_launch_path(monkeypatch):
    """Fn.BnLink (crog_gemm bwd_z): bn1 / bn2 of every bottleneck and the stem chain get their (sum g, sum g*xhat) from the epilogue
    of the data-gradient GEMM that produces their dy.  Same tower, same input, fusion on vs off: parameter gradients within the
    run-to-run noise of the atomic bf16 path (relative L2 per parameter), on the stem + layer1 of the real RN50 tower at
    2 x 416 x 416 (rows 86528 / 21632, below the row limit, so the fused path really runs — asserted through the launch count).
    The loss sits on layer1's output: the random-init tower amplifies the ~1e-5 jitter of the atomic statistics tenfold per stage
    (gradients from layer2's output differ by 19 % between two identical runs, which would make any comparison vacuous)."""
    import crog_amd.functional as Fn
    from crog_amd import kernels as K
    from crog_amd.model import build_crog
    from crog_amd.runtime import RT
    from crog_amd.testing import make_cfg
    torch.manual_seed(0)
    model, _ = build_crog(make_cfg())
    model = model.cuda().prepare()
    model.train()
    img = torch.randn(2, 3, 416, 416, generator=torch.Generator().manual_seed(3)).cuda()
    st = model.store
    for n, p, o, k, _ in st.entries:
        if n.endswith("bn3.weight"):
            st.P[o:o + k].fill_(0.5)
    st.invalidate_shadow()
    names = [(n, o, k) for n, p, o, k, _ in st.entries
             if n.startswith(("backbone.visual.conv", "backbone.visual.bn", "backbone.visual.layer1"))]
    calls = {"partial": 0}
    tap = {}
    hook = model.backbone.visual.layer1.register_forward_hook(lambda m, i, o: tap.__setitem__("x", o))
    real = K.bn_bwd_partial

    def counted(*a, **kw):
        calls["partial"] += 1
        return real(*a, **kw)
    monkeypatch.setattr(K, "bn_bwd_partial", counted)

    def run(fused):
        monkeypatch.setattr(Fn, "BN_BWD_FUSED", fused)
        calls["partial"] = 0
        st.g_clean = False
        st.zero_grad()
        RT.begin_step(img.device)
        st.forward_begins()
        model.backbone.visual(img, torch.bfloat16)
        tap["x"].float().pow(2).mean().backward()
        torch.cuda.synchronize()
        return st.G.clone(), calls["partial"]

    g_a, n_a = run(False)
    g_b, n_b = run(False)
    g_c, n_c = run(True)
    hook.remove()
    # bn1 + bn2 of layer1's three bottlenecks + two stem layers left the first pass, and bn3 of the first two (the next block's first
    # data gradient does it: Fn.BN_RES_FUSED; the last block's output carries the loss)
    assert n_a == n_b and n_c == n_a - 10, (n_a, n_c)

    def rel(a, b):
        return ((a - b).norm() / a.norm().clamp_min(1e-12)).item()
    worst = 0.0
    for n, o, k in names:
        a, b, c = g_a[o:o + k], g_b[o:o + k], g_c[o:o + k]
        assert a.abs().max().item() > 0, n
        worst = max(worst, rel(a, c))
        assert rel(a, c) <= max(3 * rel(a, b), 2e-2), (n, rel(a, c), rel(a, b))
        assert rel(a, b) < 0.1, (n, rel(a, b))        # the comparison must not be vacuous: a wrong first pass is an O(1) error
    print("fused vs two-launch BatchNorm backward: worst relative L2 over", len(names), "parameters:", worst, "first-pass launches", n_a, "->", n_c)
