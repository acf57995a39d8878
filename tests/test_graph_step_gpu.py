"""Whole-step hipGraph (crog_amd/graphs.py::GraphedTrainStep): a replayed step must be the eager step it replaces
(crog_engine.py:60-90) - same forward, same dropout masks, same Adam update, fresh inputs every step - and the device-resident
per-step state it relies on (dropout seed epoch, Adam step count / learning rate) must match the host-scalar form."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crog_amd.testing import seeded_state, synthetic_batch, tiny_cfg  # noqa: E402

pytestmark = pytest.mark.gpu


def _fresh(cfg, dtype, seed=11):
    from crog_amd.model import build_crog
    from crog_amd.optim import FusedAdam
    model, groups = build_crog(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(seeded_state(shapes, seed=seed, residual_gain=0.25))
    model = model.cuda()
    model.compute_dtype = dtype
    model.prepare()
    model.train()
    opt = FusedAdam(groups, lr=cfg.base_lr, weight_decay=1e-4, store=model.store)
    return model, opt


def _batches(cfg, n, B=4):
    return [{k: v.cuda() for k, v in synthetic_batch(B, cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=50 + i).items()}
            for i in range(n)]


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (a.float().norm() + 1e-30)).item()


@pytest.mark.parametrize("executor", ["streams", "hipgraph"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_replayed_steps_equal_eager_steps(dtype, executor):
    """7 optimizer steps with dropout 0.1 and a different batch every step: (a) engine.train_step eagerly, (b) again (the eager
    path's own run-to-run noise: split-K / statistics atomics), (c) GraphedTrainStep = 3 eager + capture + replays, with one
    eager step mixed in after the capture.  Per-step loss / metric and the final parameters of (c) sit within that noise of (a),
    and an lr change between replays reaches the captured Adam kernels."""
    from crog_amd.engine import train_step
    from crog_amd.graphs import GraphedTrainStep
    from crog_amd.runtime import RT
    cfg = tiny_cfg(dropout=0.1)
    batches = _batches(cfg, 7)
    adt = torch.bfloat16 if dtype == torch.bfloat16 else None

    def run(mode):
        model, opt = _fresh(cfg, dtype)
        RT.manual_seed(21)
        graphed = GraphedTrainStep(model, opt, cfg, adt, warmup=3, executor=executor) if mode == "graph" else None
        stats = []
        for i, b in enumerate(batches):
            if i == 5:
                for g in opt.param_groups:          # MultiStepLR milestone (train_crog.py:122,270)
                    g["lr"] = g["lr"] * 0.1
            if graphed is not None:
                st, _ = graphed(b, eager=(i == 5))
            else:
                st, _ = train_step(model, opt, None, b, cfg, autocast_dtype=adt)
            stats.append(st.clone())
        torch.cuda.synchronize()
        if graphed is not None:
            assert graphed.failed is None and graphed.graph is not None, graphed.failed
            assert graphed.replays == 3 and opt._step == 7
            assert float(opt._hyper[0, 3]) == 7.0
            if executor == "streams":     # the capture's stream-ordered chains were recovered: main, weight gradients, text tower (+ aux)
                info = graphed.replay_info
                assert info["chains"] == 3 + (RT.aux_stream is not None) and info["kernels"] > 300 and 0 < info["waits"] < info["nodes"], info
        return torch.stack(stats).cpu(), model.store.P.clone(), (opt.m.clone(), opt.v.clone())

    s0, p0, mv0 = run("eager")
    s1, p1, mv1 = run("eager")
    s3, p3, mv3 = run("eager")
    s2, p2, mv2 = run("graph")
    # The yardstick is the eager path's own run-to-run spread (split-K and statistics atomics, amplified by seven optimizer steps of a
    # chaotic tiny model).  ONE eager pair under-estimates it now and then: with 4 x |s0 - s1| this test failed once in eight runs of a
    # build whose replays were fine (0.229 against 4 x 0.045 on the fp32 loss).  Three eager runs, the largest pairwise distance, factor 6.
    pairs = ((s0, s1), (s0, s3), (s1, s3))
    # column 0 = loss (continuous); columns 1-2 = 100 * IoU / Prec@50 of a thresholded mask: a count, it jumps when one pixel of the
    # chaotic tiny model crosses 0.35, so it is held to the loose bound and the loss to the tight one
    noise_l = max((a[:, 0] - b[:, 0]).abs().max().item() for a, b in pairs)
    assert (s0[:, 0] - s2[:, 0]).abs().max().item() <= max(6 * noise_l, 2e-3 if dtype == torch.float32 else 5e-2), (noise_l, s0, s2)
    noise_m = max((a[:, 1:] - b[:, 1:]).abs().max().item() for a, b in pairs)
    assert (s0[:, 1:] - s2[:, 1:]).abs().max().item() <= max(10 * noise_m, 1.0), (s0, s2)
    noise_p = max(_rel(p0, p1), _rel(p0, p3), _rel(p1, p3))
    assert _rel(p0, p2) <= max(6 * noise_p, 1e-6 if dtype == torch.float32 else 1e-4), (noise_p, _rel(p0, p2))
    noise_mv = max(_rel(mv0[0], mv1[0]), _rel(mv0[0], mv3[0]), _rel(mv1[0], mv3[0]))
    assert _rel(mv0[0], mv2[0]) <= max(6 * noise_mv, 1e-3 if dtype == torch.float32 else 5e-2)


def test_replays_draw_fresh_dropout_masks_and_fresh_inputs():
    """Two replays on the SAME batch differ (the seed epoch advances inside the graph); two replays on different batches follow the
    batch (inputs are copied into the captured step's tensors)."""
    from crog_amd.graphs import GraphedTrainStep
    from crog_amd.runtime import RT
    cfg = tiny_cfg(dropout=0.3)
    model, opt = _fresh(cfg, torch.float32)
    for g in opt.param_groups:
        g["lr"] = 0.0            # freeze the weights: what changes between steps is only the masks / the batch
    RT.manual_seed(5)
    b = _batches(cfg, 2)
    graphed = GraphedTrainStep(model, opt, cfg, None, warmup=2)
    for _ in range(3):
        graphed(b[0])
    assert graphed.graph is not None, graphed.failed
    e0 = int(RT.seed_epoch)
    l1 = float(graphed(b[0])[0][0])
    l2 = float(graphed(b[0])[0][0])
    l3 = float(graphed(b[1])[0][0])
    assert int(RT.seed_epoch) == e0 + 3 * graphed._seeds_per_step and graphed._seeds_per_step > 0
    assert l1 != l2 and abs(l3 - l2) > 1e-4
    # same step eagerly at the same epoch == the replay (up to atomics noise)
    RT.seed_epoch.fill_(e0)
    l1e = float(graphed(b[0], eager=True)[0][0])
    assert abs(l1e - l1) < 2e-4, (l1e, l1)


def test_replay_times_the_selected_launches():
    """profile_key: the replay puts a timing-only event pair around every launch of one GEMM variant (bench.py's roofline leg)."""
    from crog_amd import kernels as K
    from crog_amd.graphs import GraphedTrainStep
    cfg = tiny_cfg(dropout=0.1)
    model, opt = _fresh(cfg, torch.bfloat16)
    b = _batches(cfg, 1)[0]
    graphed = GraphedTrainStep(model, opt, cfg, torch.bfloat16, warmup=2, profile_key=(K.A_IM2COL, K.B_KC))
    for i in range(6):
        graphed(b, profile=(i >= 4))
    assert graphed.failed is None and graphed.replay_handle is not None, graphed.failed
    n = len(graphed.prof_nodes)
    assert n >= 10          # 3x3 forward + data-gradient launches of the tiny model
    recs = graphed.profile_records()
    assert len(recs) == 2 * n
    assert all(0.0005 < ms < 5.0 for ms, _, _ in recs), [ms for ms, _, _ in recs]
    assert all(f > 0 and meta[0] == K.A_IM2COL for _, f, meta in recs)


def test_replay_notices_parameters_changed_behind_its_back():
    """load_state_dict between replays: the captured step reads the bf16 shadow FusedAdam wrote in the previous step, so after an
    external change of the fp32 parameters one step must run eagerly (re-casting the shadow) before replays resume.  Checked on the
    loss of the step right after the load: it must be the loss of the RESTORED weights on that batch (computed by a fresh model), not
    the one the stale shadow - five large Adam steps away - would give."""
    from crog_amd.graphs import GraphedTrainStep
    from crog_amd.runtime import RT
    cfg = tiny_cfg(dropout=0.0)
    batches = _batches(cfg, 7)
    model, opt = _fresh(cfg, torch.bfloat16)
    for g_ in opt.param_groups:
        g_["lr"], g_["weight_decay"] = 3e-3, 0.0           # large steps: a stale shadow is far from the restored weights
    RT.manual_seed(4)
    g = GraphedTrainStep(model, opt, cfg, torch.bfloat16, warmup=2)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    losses = []
    for i, b in enumerate(batches):
        if i == 5:
            model.load_state_dict(sd0)           # back to the initial weights (checkpoint resume)
        st, _ = g(b)
        losses.append(float(st[0]))
    torch.cuda.synchronize()
    assert g.replays == 4 and g.graph is not None        # calls 3, 4, 5 and 7 replayed; call 6 (after the load) went eager
    # what the restored weights give on batch 5, from a fresh model (training-mode forward: batch statistics, no dropout)
    ref, _ = _fresh(cfg, torch.bfloat16)
    ref.load_state_dict(sd0)
    b = batches[5]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        _, _, loss_ref, _ = ref(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
    loss_ref = float(loss_ref.detach())
    # what the weights BEFORE the load (five steps of lr 3e-3 away) give on the same batch: the stale-shadow outcome
    assert abs(losses[5] - loss_ref) < 0.02 * max(1.0, abs(loss_ref)), (losses, loss_ref)
    assert abs(losses[4] - loss_ref) > 5 * abs(losses[5] - loss_ref) or abs(losses[5] - loss_ref) < 1e-3, (losses, loss_ref)


def test_capturable_adam_matches_host_scalar_adam():
    """crog_adam_step_dev + crog_adam_advance (step count / bias corrections / lr in device memory) == crog_adam_step with host
    scalars == torch.optim.Adam, over several steps and an lr change."""
    from crog_amd import kernels as K
    n = 4096 + 24
    torch.manual_seed(0)
    p0 = torch.randn(n, device="cuda")
    ref = torch.nn.Parameter(p0.clone())
    topt = torch.optim.Adam([ref], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    ph, pd = p0.clone(), p0.clone()
    mh, vh, md, vd = (torch.zeros(n, device="cuda") for _ in range(4))
    hyper = torch.zeros(4, device="cuda")
    lr = 1e-2
    for step in range(1, 8):
        g = torch.randn(n, device="cuda") * (1 + step)
        if step == 5:
            lr = 1e-3
            topt.param_groups[0]["lr"] = lr
        ref.grad = g.clone()
        topt.step()
        K.adam_step(ph, g, mh, vh, n, lr, 0.9, 0.999, 1e-8, 1e-2, step)
        hyper[0:1].fill_(lr)
        K.adam_advance(hyper, 0.9, 0.999)
        K.adam_step_dev(pd, g, md, vd, n, hyper, 0.9, 0.999, 1e-8, 1e-2)
    torch.cuda.synchronize()
    assert float(hyper[3]) == 7.0
    assert torch.equal(ph, pd) and torch.equal(mh, md) and torch.equal(vh, vd)       # same arithmetic, bit for bit
    assert (ph - ref.detach()).abs().max().item() < 2e-6


def test_seed_epoch_shifts_every_dropout_kernel_like_a_host_seed():
    """seed s + epoch e (device) drops exactly the elements that seed s + e (host argument) drops."""
    from crog_amd import kernels as K
    from crog_amd.runtime import RT
    x = torch.randn(512, 256, device="cuda", dtype=torch.bfloat16)
    out_a, out_b = torch.empty_like(x), torch.empty_like(x)
    ep = RT.enable_seed_epoch(x.device)
    try:
        ep.fill_(0)
        K.add_dropout(None, x, out_a, 0.25, (7 << 32) | 1234 + 99)
        ep.fill_(99)
        K.add_dropout(None, x, out_b, 0.25, (7 << 32) | 1234)
        torch.cuda.synchronize()
        assert torch.equal(out_a, out_b)
        K.add_dropout(None, x, out_b, 0.25, (7 << 32) | 1235)
        torch.cuda.synchronize()
        assert not torch.equal(out_a, out_b)
    finally:
        ep.fill_(0)


def _forced_ddp_graph_case():
    """Body of test_graphed_step_under_forced_ddp_and_syncbn (runs in a child process, see there)."""
    import socket
    import torch.distributed as dist
    from crog_amd.engine import train_step
    from crog_amd.graphs import GraphedTrainStep
    from crog_amd.parallel import DistributedDataParallel, convert_sync_batchnorm
    from crog_amd.runtime import RT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    cfg = tiny_cfg(dropout=0.1)
    batches = _batches(cfg, 6)
    try:
        def run(graph):
            model, opt = _fresh(cfg, torch.bfloat16)
            convert_sync_batchnorm(model, force=True)
            net = DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True, force=True, bucket_cap_mb=0.25)
            opt.attach(model.store)        # the wrapper re-homed the parameters (prepare): the optimizer follows, as in bench.py
            RT.manual_seed(3)
            # the job picks its transports itself (start-up self-test, collective verdict): at world size 1 both pass, so the
            # statistics go through the C-ABI communicator, the buckets through crog_allreduce_bucket, and the step is capturable
            from crog_amd.parallel import step_is_capturable
            assert RT.comm.kind.startswith("crog_comm:") and "rccl" in RT.comm.kind, RT.comm.kind
            assert net.bucket_comm is not None and step_is_capturable(net)
            graphed = GraphedTrainStep(net, opt, cfg, torch.bfloat16, warmup=3, verify=True) if graph else None
            if graph == "reject":          # a replay that does not reproduce the eager step: the first-replay check must refuse the graph
                real = graphed._replay_once
                graphed._replay_once = lambda b, profile=False: tuple((3.0 * real(b, profile)[0], None))
            if graph == "reject_update":   # ... and one with the right loss whose optimizer update went missing (a dropped bucket
                real = graphed._replay_once      # all-reduce / Adam chunk leaves the forward loss untouched): refused on the post-step state
                def lossy(b, profile=False):
                    r = real(b, profile)
                    opt.m.mul_(0.5)
                    return r
                graphed._replay_once = lossy
            out = []
            for b in batches:
                st, _ = graphed(b) if graph else train_step(net, opt, None, b, cfg, autocast_dtype=torch.bfloat16)
                out.append(st.clone())
            torch.cuda.synchronize()
            assert net.reducer.direct is net.bucket_comm
            if graph in ("reject", "reject_update"):
                assert graphed.verified is False and graphed.failed is not None and graphed.graph is None and graphed.replay_handle is None
                assert graph == "reject" or "exp_avg" in graphed.failed, graphed.failed
            elif graph:
                assert graphed.failed is None and graphed.verified is True and graphed.replays == 3, graphed.failed
                assert graphed.collectives["syncbn"] > 0 and graphed.collectives["buckets"] > 1
            return torch.stack(out).cpu(), model.store.P.clone()
        s0, p0 = run(False)
        s1, p1 = run(False)
        s3, p3 = run(False)
        s2, p2 = run(True)
        s4, p4 = run("reject")
        s5, p5 = run("reject_update")
        # (three eager runs, largest pairwise distance, factor 6: see test_replayed_steps_equal_eager_steps)
        noise_l = max((a[:, 0] - b[:, 0]).abs().max().item() for a, b in ((s0, s1), (s0, s3), (s1, s3)))
        noise_p = max(_rel(p0, p1), _rel(p0, p3), _rel(p1, p3))
        assert (s0[:, 0] - s2[:, 0]).abs().max().item() <= max(6 * noise_l, 5e-2), (noise_l, s0, s2)
        assert _rel(p0, p2) <= max(6 * noise_p, 1e-4)
        # the refused graph left no trace: the state was rewound and every step ran eagerly
        assert (s0[:, 0] - s4[:, 0]).abs().max().item() <= max(6 * noise_l, 5e-2), (noise_l, s0, s4)
        assert _rel(p0, p4) <= max(6 * noise_p, 1e-4)
        assert (s0[:, 0] - s5[:, 0]).abs().max().item() <= max(6 * noise_l, 5e-2), (noise_l, s0, s5)
        assert _rel(p0, p5) <= max(6 * noise_p, 1e-4)
    finally:
        RT.comm = None
        RT.reducer = None
        dist.destroy_process_group()
    print("FORCED_DDP_GRAPH_OK")


def test_graphed_step_under_forced_ddp_and_syncbn():
    """DistributedDataParallel + SyncBatchNorm forced on at world size 1 over RCCL (every statistics exchange and gradient bucket
    really issued): the captured step carries the collectives and replays match the eager DDP steps.
    In a child process: torch's ProcessGroupNCCL watchdog THREAD polls the completion events of collectives, and on this stack
    (torch 2.10 / ROCm 7.0) it was seen - once in a few runs - to query an event that was last recorded inside the stream capture,
    which HIP refuses (hipErrorCapturedEvent) and torch turns into std::terminate of the whole process.  That race is torch's and only
    exists while a capture is open next to a live process group (replay with collectives is opt-in for multi-rank runs, DESIGN.md
    section 6); a run that dies of it is repeated, any other failure fails the test."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_graph_step_gpu as t; t._forced_ddp_graph_case()"
            % (ROOT, os.path.join(ROOT, "tests")))
    last = ""
    for attempt in range(4):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
        last = (r.stdout + r.stderr)[-4000:]
        if r.returncode == 0 and "FORCED_DDP_GRAPH_OK" in r.stdout:
            return
        if "hipErrorCapturedEvent" not in last and "captured" not in last.lower():
            break
        print(f"attempt {attempt}: the process group watchdog hit the capture (hipErrorCapturedEvent); repeating")
    raise AssertionError(last)
