"""Deterministic mode (crog_amd.runtime.set_deterministic / CROG_DETERMINISTIC=1; include/crog_hip.h crog_set_deterministic): every sum
whose order would depend on fp32 atomics takes an ordered form, so the training step (crog_engine.py:60-90) gives the SAME BITS run after
run, issued eagerly or replayed from the captured step, in bf16 as well as fp32 - and the same values as the default mode up to the
order of the additions."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from crog_amd.testing import make_cfg, synthetic_batch, tiny_cfg  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture()
def deterministic():
    from crog_amd.runtime import set_deterministic
    set_deterministic(True)
    yield
    set_deterministic(False)


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (a.float().norm() + 1e-30)).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_steps_are_bit_identical_run_to_run_and_replay_equals_eager(deterministic, dtype):
    """Six optimizer steps of the tiny CROG (dropout 0.1, a different batch every step), three times from the same initial state: twice
    issued from Python, once as 3 eager steps + capture + 3 replays.  Per-step statistics, the gradients of the last step, the parameters
    and both Adam moments are bit-identical across all three."""
    import test_graph_step_gpu as T
    from crog_amd.engine import train_step
    from crog_amd.graphs import GraphedTrainStep
    from crog_amd.runtime import RT
    cfg = tiny_cfg(dropout=0.1)
    batches = T._batches(cfg, 6)
    adt = torch.bfloat16 if dtype == torch.bfloat16 else None

    def run(graph):
        model, opt = T._fresh(cfg, dtype)
        RT.manual_seed(21)
        graphed = GraphedTrainStep(model, opt, cfg, adt, warmup=3) if graph else None
        stats = []
        for b in batches:
            st, _ = graphed(b) if graph else train_step(model, opt, None, b, cfg, autocast_dtype=adt)
            stats.append(st.clone())
        torch.cuda.synchronize()
        if graph:
            assert graphed.failed is None and graphed.replays == 3, graphed.failed
        return torch.stack(stats), model.store.G.clone(), model.store.P.clone(), opt.m.clone(), opt.v.clone()

    a, b, c = run(False), run(False), run(True)
    names = ("statistics", "gradients", "parameters", "exp_avg", "exp_avg_sq")
    for n, x, y, z in zip(names, a, b, c):
        assert torch.equal(x, y), f"eager vs eager: {n} differ (max |d| = {float((x - y).abs().max()):.3e})"
        assert torch.equal(x, z), f"replay vs eager: {n} differ (max |d| = {float((x - z).abs().max()):.3e})"
    assert torch.isfinite(a[0]).all()


def test_deterministic_step_matches_the_default_step(deterministic):
    """Same step, default mode against deterministic mode (fp32, tiny model, first step from identical weights): the gradients agree
    within the default mode's own run-to-run noise (the modes differ only in the order of fp32 additions)."""
    import test_graph_step_gpu as T
    from crog_amd.runtime import RT, set_deterministic
    cfg = tiny_cfg(dropout=0.0)
    b = T._batches(cfg, 1)[0]

    def grads():
        model, opt = T._fresh(cfg, torch.float32)
        RT.manual_seed(3)
        _, _, loss, _ = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
        opt.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), model.store.G.clone()
    l_det, g_det = grads()
    l_det2, g_det2 = grads()
    assert l_det == l_det2 and torch.equal(g_det, g_det2)
    set_deterministic(False)
    l0, g0 = grads()
    l1, g1 = grads()
    noise = max(_rel(g0, g1), 1e-6)
    print(f"default vs default {noise:.2e}; deterministic vs default {_rel(g0, g_det):.2e}")
    assert abs(l_det - l0) <= max(4 * abs(l0 - l1), 1e-4) and _rel(g0, g_det) <= max(6 * noise, 2e-2)


@pytest.mark.parametrize("B,dropout", [(2, 0.1), (4, 0.0), (8, 0.1)])
def test_full_depth_bf16_step_is_bit_identical(deterministic, B, dropout):
    """CROG-R50 at full depth, bf16, 416 x 416: forward + backward twice from the same weights - the production kernels (the
    ping-pong GEMMs with slab statistics, split-K slabs + crog_splitk_reduce with empty trailing slices, the ordered BatchNorm /
    LayerNorm reductions, the ordered embedding / head sums) leave bit-identical gradients.  (B = 4 without dropout is the case that
    exposed the run-to-run last-bit differences of LayerNorm backward beside a forked weight-gradient GEMM: runtime.set_deterministic.)"""
    from crog_amd.model import build_crog
    from crog_amd.runtime import RT
    torch.manual_seed(0)
    cfg = make_cfg(dropout=dropout)
    model, _ = build_crog(cfg)
    model = model.cuda().prepare()
    model.train()
    b = {k: v.cuda() for k, v in synthetic_batch(B, 416, cfg.word_len, cfg.clip_arch["vocab_size"], seed=9).items()}
    sd = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}

    def grads():
        model.load_state_dict({**model.state_dict(), **sd})
        RT.manual_seed(5)
        model.store.g_clean = False
        model.store.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            _, _, loss, _ = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), model.store.G.clone()
    l0, g0 = grads()
    l1, g1 = grads()
    assert l0 == l1 and torch.isfinite(g0).all()
    diff = (g0 != g1)
    if diff.any():
        bad = [n for n, p, o, k, _ in model.store.entries if bool(diff[o:o + k].any())]
        raise AssertionError(f"{int(diff.sum())} gradient elements differ between two runs; parameters: {bad[:10]}")
