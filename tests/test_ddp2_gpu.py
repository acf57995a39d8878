"""SURVEY §4 tier 3 on the real kernels: a world-size-2 run of DistributedDataParallel + SyncBatchNorm (two fresh child
processes sharing cuda:0, gloo collectives on CUDA tensors) must reproduce the single-process result on the concatenated batch
(train_crog.py:113-114,154-156).  fp32: against the REFERENCE's own B = 4 fixture (its BatchNorm over the whole batch is what
SyncBatchNorm over 2 x 2 samples must equal).  bf16: against a single-process bf16 run — that path exchanges whole statistic
replica buffers and stores BatchNorm parameter gradients as global totals / world, which only a world > 1 run can check."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")
WORKER = os.path.join(ROOT, "tests", "ddp2_worker.py")
# Ranks of the "many processes on one GPU" tests.  BASELINE's configs run 8; this pool's GPU boxes kill a job with more than 6 processes on
# the card (the pytest process holds it as well), so the default is 4 ranks - three peers per exchange, the same slot-parity / sequence
# protocol.  CROG_TEST_WORLD=8 runs the target world size where the box allows nine processes on one GPU (<= 16 = MAX_WORLD of comm.hip).
MANY = int(os.environ.get("CROG_TEST_WORLD", "4"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


_RUNS = {}      # (configuration, replica) -> results: several tests want the SAME run (a default two-rank bf16 step, a one-process step and its twin
#                  for the run-to-run floor): each is executed once per pytest session.  `replica` tells runs of one configuration apart.


def same(a, b, key):
    """Bit-equality of the FULL flat gradient / parameter buffer of two runs: the workers store a SHA-1 of the whole 66.6 M-float buffer (`G_sha`,
    `P_sha`) and every 16th element (`G`, `P`: what norms and cosines are taken on) instead of 266 MB per buffer and rank."""
    return bool(np.array_equal(a[key + "_sha"], b[key + "_sha"])) and bool(np.array_equal(a[key], b[key]))


def _run(world, out_dir, dtype, gain, size=96, B=4, tag="a", extra_env=None, replica=0):
    key = (world, dtype, gain, size, B, tuple(sorted((extra_env or {}).items())), replica)
    if key in _RUNS:
        return _RUNS[key]
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), port, str(out_dir), dtype, str(gain), str(size), str(B), tag], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=420)
            outs.append(o.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} of {world} failed:\n{o[-3000:]}"
    _RUNS[key] = [dict(np.load(os.path.join(out_dir, f"{tag}_rank{r}_of{world}.npz"))) for r in range(world)]
    return _RUNS[key]


def test_two_rank_ddp_syncbn_fp32_equals_reference_on_concatenated_batch(tmp_path):
    g = np.load(os.path.join(GOLD, "tiny_crog.npz"))
    meta = json.load(open(os.path.join(GOLD, "tiny_crog.json")))
    r0, r1 = _run(2, tmp_path, "f32", 1.0)
    assert int(r0["n_buckets"]) > 3                      # several gradient buckets were in flight (bucket_cap_mb = 0.25)
    want = np.concatenate([g["pred_" + n] for n in ("ins", "qua", "sin", "cos", "wid")], 1)
    got = np.concatenate([r0["preds"], r1["preds"]], 0)
    e = float(np.abs(got - want).max())
    print(f"2-rank logits vs reference fixture (B = 4 in one process): max err {e:.2e}")
    assert e < 1e-3
    # each rank's loss is the mean over its half; the reference's is the mean over all four samples
    assert abs(0.5 * (float(r0["loss"]) + float(r1["loss"])) - float(g["loss_total"])) < 1e-4
    # after the bucketed all-reduce both ranks hold the SAME averaged gradient, and it is the full-batch gradient
    assert same(r0, r1, "G")
    ref = np.where(g["grad_norms"] < 0, 0.0, g["grad_norms"])
    names = meta["param_names"]
    loose = np.array(["txt_proj" in n for n in names])
    trunk = np.array([n.startswith("backbone.visual") for n in names])
    tol = np.where(loose, 4e-2, np.where(trunk, 2e-2, 5e-3))
    bad = np.abs(r0["grad_norms"] - ref) > tol * np.abs(ref) + 2e-5
    assert not bad.any(), [(names[i], float(r0["grad_norms"][i]), float(ref[i])) for i in np.nonzero(bad)[0][:8]]
    # SyncBatchNorm: running statistics of the global batch, identical on both ranks
    assert np.array_equal(r0["bn_checksum"], r1["bn_checksum"])
    assert np.allclose(r0["bn_checksum"], g["bn_running_checksum"], rtol=1e-4, atol=1e-3)
    # two optimizer steps later the replicas are still bit-identical (rank 1 STARTED from different weights: broadcast worked)
    assert same(r0, r1, "P") and np.array_equal(r0["bn_final"], r1["bn_final"])


def test_two_rank_ddp_syncbn_bf16_equals_single_process(tmp_path):
    """bf16 (the benchmark dtype): statistics travel as [R][C][2] atomic replica rows, BatchNorm parameter gradients are the
    all-reduced totals / world.  2 ranks x 4 samples against 1 process x 8 samples at 160 x 160 on a damped trunk.  Yardstick: the
    single-process path's own run-to-run noise (its BatchNorm sums are fp32 atomics whose order changes, and every later bf16
    rounding amplifies that) — the 2-rank run must sit within a small multiple of it."""
    meta = json.load(open(os.path.join(GOLD, "tiny_crog.json")))
    kw = dict(size=160, B=8)
    (one,) = _run(1, tmp_path, "bf16", 0.25, tag="one", **kw)
    (two,) = _run(1, tmp_path, "bf16", 0.25, tag="again", replica=1, **kw)
    r0, r1 = _run(2, tmp_path, "bf16", 0.25, tag="ddp", **kw)
    assert same(r0, r1, "G") and same(r0, r1, "P") and np.array_equal(r0["bn_final"], r1["bn_final"])
    got = np.concatenate([r0["preds"], r1["preds"]], 0)
    scale = float(np.abs(one["preds"]).max())

    def rms(a, b):
        return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))
    e, floor = rms(got, one["preds"]), rms(two["preds"], one["preds"])
    print(f"bf16 logits, relative RMS: 2-rank vs 1-process {e:.3e}; 1-process run-to-run {floor:.3e}; logit scale {scale:.2f}")
    assert e < max(4 * floor, 2e-2)
    lm = 0.5 * (float(r0["loss"]) + float(r1["loss"]))
    print(f"loss: 2-rank mean {lm:.5f}, 1-process {float(one['loss']):.5f} / {float(two['loss']):.5f}")
    assert abs(lm - float(one["loss"])) < max(1e-2 * abs(float(one["loss"])), 4 * abs(float(two["loss"]) - float(one["loss"])))
    names = meta["param_names"]
    a, b, c = r0["grad_norms"], one["grad_norms"], two["grad_norms"]
    rel, rel_floor = np.abs(a - b) / (np.abs(b) + 1e-6), np.abs(c - b) / (np.abs(b) + 1e-6)
    bn_params = np.array([(".bn" in n or "downsample.1" in n or "connect.1" in n or "norm_layer" in n or n.endswith(".1.weight") or n.endswith(".1.bias"))
                          for n in names])
    big = b > 1e-4
    print(f"bf16 gradient norms vs 1-process: 2-rank median rel {np.median(rel[big]):.2e} worst {rel[big].max():.2e} "
          f"(BatchNorm parameters worst {rel[big & bn_params].max():.2e}); run-to-run median {np.median(rel_floor[big]):.2e} worst {rel_floor[big].max():.2e}")
    # a wrong 1/world factor on the BatchNorm parameter gradients (or on `count`) would be a factor 2, not a few per cent
    assert rel[big & bn_params].max() < max(0.25, 4 * rel_floor[big & bn_params].max())
    assert np.median(rel[big]) < max(3e-2, 4 * np.median(rel_floor[big]))
    d_bn, f_bn = np.abs(r0["bn_checksum"] - one["bn_checksum"]), np.abs(two["bn_checksum"] - one["bn_checksum"])
    print(f"BatchNorm running-statistic checksums: 2-rank vs 1-process worst {d_bn.max():.3e}; run-to-run worst {f_bn.max():.3e}")
    assert (d_bn <= np.maximum(4 * f_bn.max(), 2e-2 + 2e-3 * np.abs(one["bn_checksum"]))).all()
    ga, gb, gc = r0["G"].astype(np.float64), one["G"].astype(np.float64), two["G"].astype(np.float64)
    cos = lambda x, y: float(x @ y / (np.linalg.norm(x) * np.linalg.norm(y)))
    print(f"flat gradient cosine: 2-rank vs 1-process {cos(ga, gb):.5f}; run-to-run {cos(gc, gb):.5f}")
    assert cos(ga, gb) > min(0.98, 1 - 4 * (1 - cos(gc, gb)))


@pytest.mark.parametrize("world", [2, MANY])
def test_peer_mailbox_allreduce_processes_sharing_one_gpu(tmp_path, world):
    """crog_syncbn_stats through the hipIpc mailboxes (csrc/comm.hip): `world` processes sharing cuda:0 exchange 60 vectors of 2 ... 8192
    floats.  Two ranks: every result equals gloo's all-reduce bit for bit.  MANY ranks (4 on this pool, 8 = the target world size of
    BASELINE's configs with CROG_TEST_WORLD=8; the slot-parity / sequence protocol with several peers per exchange): every result equals
    the rank-order sum of the gathered contributions bit for bit (and gloo's tree sum to rounding).  All ranks hold identical bits,
    nothing timed out."""
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(ROOT, "tests", "peer_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), port, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=420)
            outs.append(o.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
    res = [json.load(open(tmp_path / f"peer_rank{r}.json")) for r in range(world)]
    print("peer mailbox exchange:", res)
    for r in res:
        assert r["created"], r["err"]
        assert "timed_out_at" not in r and r["timed_out"] == 0
        assert r["mismatches"] == 0 and r["identical_across_ranks"] and r["chain_value_ok"]


def test_two_rank_ddp_syncbn_over_the_peer_mailboxes_equals_the_gloo_exchange(tmp_path):
    """The whole 2-rank fp32 step with the BatchNorm statistics exchanged by crog_syncbn_stats (CROG_SYNCBN_DIRECT=peer) instead of
    gloo: a two-rank sum is a + b either way, the fp32 forward is bit-reproducible, so the logits are IDENTICAL to the gloo run (the loss to the last bits);
    the gradient buffers (split-K atomics) agree to run-to-run noise."""
    a0, a1 = _run(2, tmp_path, "f32", 1.0, tag="gloo")
    b0, b1 = _run(2, tmp_path, "f32", 1.0, tag="peer", extra_env={"CROG_SYNCBN_DIRECT": "peer"})
    for a, b in ((a0, b0), (a1, b1)):
        assert np.array_equal(a["preds"], b["preds"])
        assert abs(float(a["loss"]) - float(b["loss"])) < 1e-5        # (the loss kernel adds its partial sums atomically: last-bit noise)
        rel = np.linalg.norm(a["G"] - b["G"]) / np.linalg.norm(a["G"])
        assert rel < 2e-2, rel
        assert np.allclose(a["bn_checksum"], b["bn_checksum"], rtol=1e-6)
    assert same(b0, b1, "P")            # both ranks end on identical parameters


def test_two_rank_deterministic_mode_is_bit_reproducible(tmp_path):
    """Deterministic mode under DistributedDataParallel + SyncBatchNorm, two ranks (bf16, side streams on): the same two-rank run twice
    leaves the same bits - logits, loss, the averaged gradient buffer, the parameters after two optimizer steps, the BatchNorm buffers -
    on both ranks.  (Bit-identity ACROSS world sizes is not defined: two ranks add (sum over half 0) + (sum over half 1) where one
    process adds its tiles in one ordered pass - another association of the same fp32 sums, in the BatchNorm statistics and in every
    gradient.  What can be asserted across world sizes is closeness without the run-to-run noise floor: both runs are noise-free, so the
    bounds below are the reassociation's own, far under the 4x-noise yardsticks of the default-mode test above.)"""
    kw = dict(size=160, B=8, extra_env={"CROG_DETERMINISTIC": "1", **({"CROG_DET_STREAMS": os.environ["TEST_DET_STREAMS"]} if "TEST_DET_STREAMS" in os.environ else {})})
    a0, a1 = _run(2, tmp_path, "bf16", 0.25, tag="det_a", **kw)
    b0, b1 = _run(2, tmp_path, "bf16", 0.25, tag="det_b", replica=1, **kw)
    for x, y in ((a0, b0), (a1, b1)):
        for k in ("preds", "G", "bn_checksum", "grad_norms"):
            assert np.array_equal(x[k], y[k]), k
        assert float(x["loss"]) == float(y["loss"])
        # two optimizer steps later as well.  (Until the library was built without packed-fp32 VALU instructions this was only reported: two
        # PROCESSES time-slicing one GPU run their kernels beside each other, and the other process's MFMA kernels were enough to change a
        # v_pk_fma_f32 result of this one - LAB_NOTES section 10; seen once in three runs of this pair.)
        assert same(x, y, "P") and np.array_equal(x["bn_final"], y["bn_final"])
    assert same(a0, a1, "G") and same(a0, a1, "P")
    (one,) = _run(1, tmp_path, "bf16", 0.25, tag="det_one", **kw)
    (again,) = _run(1, tmp_path, "bf16", 0.25, tag="det_one2", replica=1, **kw)
    assert same(one, again, "G") and np.array_equal(one["preds"], again["preds"])
    got = np.concatenate([a0["preds"], a1["preds"]], 0)
    rms = float(np.sqrt(np.mean((got - one["preds"]) ** 2)) / np.sqrt(np.mean(one["preds"] ** 2)))
    ga, gb = a0["G"].astype(np.float64), one["G"].astype(np.float64)
    cosg = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
    print(f"deterministic mode, 2 ranks vs 1 process (bf16): logits relative RMS {rms:.3e}, flat-gradient cosine {cosg:.6f}")
    assert rms < 0.1 and cosg > 0.95      # (measured 4.9e-2: bf16 at 160 x 160 amplifies the reassociated BatchNorm sums; the default-mode test's run-to-run floor is of the same size)


def test_many_rank_ddp_syncbn_over_the_peer_mailboxes(tmp_path):
    """More than two ranks on one GPU (VERDICT r4 asked for the target world size, 8: CROG_TEST_WORLD=8; 4 by default, see MANY):
    MANY processes sharing cuda:0, one sample each, DistributedDataParallel + SyncBatchNorm (fp32, tiny CROG): every BatchNorm layer's
    statistics cross MANY - 1 peers per exchange through the hipIpc mailboxes (CROG_SYNCBN_DIRECT=peer), the gradient buckets through
    gloo.  All ranks end on identical bits (logit statistics of the global batch, averaged gradients, parameters after two optimizer
    steps), the run agrees with ONE process on the whole batch (what SyncBatchNorm over MANY x 1 samples must equal).

    Gradient buckets of 32 MiB here (10 bucket collectives per step) instead of the two-rank tests' 0.25 MiB (100 per step).  Round 5 ended with
    this test skipped: at 4 ranks it did not finish.  Round 6 traced it (scripts/many_rank_probe.py, one time-stamped line per collective and
    rank; LAB_NOTES section 11): every rank issues the SAME 171 collectives in the SAME order - no bucket-order divergence -, and the run
    stops INSIDE one `dist.all_reduce(async_op=True)` of a gloo bucket on a CUDA tensor (ProcessGroupGloo's issue path: pinned staging
    allocation + device-to-host copy), which on this pool does not return while the device sits in a cross-process mailbox exchange that
    itself waits for this very rank: 108.9 s = seven mailbox time-outs on one rank, 15.4 s = one on another, the other two ranks through
    in 1.2-1.9 s.  Four processes time-slice the one GPU (a mailbox exchange costs ~0.13 s of wall time, a gloo all-reduce 18-44 ms), so
    100 bucket issues per step meet a spinning exchange often; with 10 the same step takes 9-16 s and completes (53 s per run).  A real
    multi-GPU job takes neither path: its buckets are ncclAllReduce calls on a carrier stream (crog_allreduce_bucket), no host staging."""
    kw = dict(size=96, B=MANY)
    big = {"CROG_WORKER_BUCKET_MB": "32"}
    peer = _run(MANY, tmp_path, "f32", 1.0, tag="peer8", extra_env={"CROG_SYNCBN_DIRECT": "peer", **big}, **kw)
    assert int(peer[0]["n_buckets"]) >= 8 and int(peer[0]["bucket_launches"]) == 2 * int(peer[0]["n_buckets"])
    (one,) = _run(1, tmp_path, "f32", 1.0, tag="one8", **kw)
    for r in peer[1:]:
        assert same(r, peer[0], "G") and same(r, peer[0], "P") and np.array_equal(r["bn_final"], peer[0]["bn_final"])
        assert np.array_equal(r["bn_checksum"], peer[0]["bn_checksum"])
    got = np.concatenate([r["preds"] for r in peer], 0)
    e_one = float(np.abs(got - one["preds"]).max())
    print(f"{MANY} ranks on one GPU, statistics over the mailboxes, vs one process on the whole batch: max |dlogit| {e_one:.2e}")
    assert e_one < 1e-3
    lm = float(np.mean([float(r["loss"]) for r in peer]))
    assert abs(lm - float(one["loss"])) < 1e-4
    rel1 = np.linalg.norm(peer[0]["G"] - one["G"]) / np.linalg.norm(one["G"])
    print(f"averaged gradient buffer vs one process (every 16th element): {rel1:.2e}")
    assert rel1 < 2e-2
    assert np.allclose(peer[0]["bn_checksum"], one["bn_checksum"], rtol=1e-4, atol=1e-3)


def test_two_rank_parked_weight_gradients_reach_their_bucket_before_it_is_reduced(tmp_path):
    """Reducer.wait is queued by the first gradient announcement of the pass (the head's), i.e. AHEAD of the runtime's own
    end-of-backward flush - a weight gradient still parked for a grouped launch when backward ends was all-reduced before its GEMM was
    enqueued, and its late announcement re-armed the reducer for a second round of all-reduces.  CROG_GROUP_STALE=1000 keeps every group
    parked until it fills or backward ends (the tiny model's last groups stay parked): every bucket must still be launched exactly once per
    step, both ranks must hold identical bits, and the gradients must agree with the default rule's (bf16 run-to-run noise)."""
    kw = dict(size=160, B=8)
    p0, p1 = _run(2, tmp_path, "bf16", 0.25, tag="parked", extra_env={"CROG_GROUP_STALE": "1000"}, **kw)
    d0, d1 = _run(2, tmp_path, "bf16", 0.25, tag="default", **kw)
    e0, _ = _run(2, tmp_path, "bf16", 0.25, tag="default2", replica=1, **kw)
    for r in (p0, p1, d0, d1):
        assert int(r["bucket_launches"]) == 2 * int(r["n_buckets"]), (int(r["bucket_launches"]), int(r["n_buckets"]))
    assert same(p0, p1, "G") and same(p0, p1, "P")
    ga, gb, gc = (x["G"].astype(np.float64) for x in (p0, d0, e0))
    cos = lambda x, y: float(x @ y / (np.linalg.norm(x) * np.linalg.norm(y)))
    print(f"flat gradient cosine: groups parked to the end vs default rule {cos(ga, gb):.5f}; default rule run-to-run {cos(gc, gb):.5f}; "
          f"norm ratio {np.linalg.norm(ga) / np.linalg.norm(gb):.4f}")
    assert cos(ga, gb) > min(0.98, 1 - 4 * (1 - cos(gc, gb)))
    assert abs(np.linalg.norm(ga) / np.linalg.norm(gb) - 1) < max(0.05, 4 * abs(np.linalg.norm(gc) / np.linalg.norm(gb) - 1))


def test_two_rank_bf16_backward_exchanges_ride_in_the_producing_kernels(tmp_path):
    """Round 5: under SyncBatchNorm the BACKWARD statistics are exchanged by the last block of the kernel that produces them
    (crog_bn_bwd_partial_sync, crog_gemm_desc.stat_sync: comm_dev.h) instead of by a launch of their own.  Two ranks sharing cuda:0,
    bf16 (the path with atomic replica rows), statistics through the peer mailboxes: half of the exchanges of a step are in-kernel, both
    ranks end on identical bits, and the run agrees with the same two ranks exchanging through gloo (launches of their own) and with one
    process on the whole batch within the bf16 path's run-to-run noise."""
    kw = dict(size=160, B=8)
    r0, r1 = _run(2, tmp_path, "bf16", 0.25, tag="fused", extra_env={"CROG_SYNCBN_DIRECT": "peer"}, **kw)
    u0, u1 = _run(2, tmp_path, "bf16", 0.25, tag="unfused", extra_env={"CROG_SYNCBN_DIRECT": "peer", "CROG_SYNCBN_FUSE": "0"}, **kw)
    (one,) = _run(1, tmp_path, "bf16", 0.25, tag="one_f", **kw)
    (two,) = _run(1, tmp_path, "bf16", 0.25, tag="one_f2", replica=1, **kw)
    assert int(r0["syncbn_in_kernel"]) > 0 and int(u0["syncbn_in_kernel"]) == 0
    print(f"exchanges over two steps: fused run {int(r0['syncbn_launches'])} launches + {int(r0['syncbn_in_kernel'])} in-kernel; unfused run {int(u0['syncbn_launches'])} launches")
    assert int(r0["syncbn_launches"]) + int(r0["syncbn_in_kernel"]) == int(u0["syncbn_launches"])
    # every backward exchange, and - end of round 5 - the forward ones too: in the tail of the ping-pong GEMM that accumulates the statistics, or in
    # a single-block launch crog_gemm adds behind any other kernel (counted as in-kernel by SyncBNComm.fuse_ptr: no Python-side exchange call)
    assert int(r0["syncbn_in_kernel"]) >= int(u0["syncbn_launches"]) - 4
    assert same(r0, r1, "G") and same(r0, r1, "P") and np.array_equal(r0["bn_final"], r1["bn_final"])

    def rms(a, b):
        return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))
    got, ref = np.concatenate([r0["preds"], r1["preds"]], 0), np.concatenate([u0["preds"], u1["preds"]], 0)
    floor = rms(two["preds"], one["preds"])
    print(f"bf16 logits relative RMS: fused vs unfused 2-rank {rms(got, ref):.3e}; fused vs 1-process {rms(got, one['preds']):.3e}; 1-process run-to-run {floor:.3e}")
    assert rms(got, ref) < max(4 * floor, 2e-2) and rms(got, one["preds"]) < max(4 * floor, 5e-2)
    ga, gb, gc, gd = (x["G"].astype(np.float64) for x in (r0, u0, one, two))
    cos = lambda x, y: float(x @ y / (np.linalg.norm(x) * np.linalg.norm(y)))
    print(f"flat gradient cosine: fused vs unfused {cos(ga, gb):.5f}; fused vs 1-process {cos(ga, gc):.5f}; run-to-run {cos(gd, gc):.5f}")
    assert cos(ga, gb) > min(0.98, 1 - 4 * (1 - cos(gd, gc))) and cos(ga, gc) > min(0.98, 1 - 4 * (1 - cos(gd, gc)))
