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


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run(world, out_dir, dtype, gain):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), port, str(out_dir), dtype, str(gain)], env=env,
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
    return [dict(np.load(os.path.join(out_dir, f"rank{r}_of{world}.npz"))) for r in range(world)]


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
    assert np.array_equal(r0["G"], r1["G"])
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
    assert np.array_equal(r0["P"], r1["P"]) and np.array_equal(r0["bn_final"], r1["bn_final"])


def test_two_rank_ddp_syncbn_bf16_equals_single_process(tmp_path):
    """bf16 (the benchmark dtype): statistics travel as [R][C][2] atomic replica rows, BatchNorm parameter gradients are the
    all-reduced totals / world.  2 ranks x 2 samples against 1 process x 4 samples, damped trunk (bf16 noise is not amplified)."""
    meta = json.load(open(os.path.join(GOLD, "tiny_crog.json")))
    (one,) = _run(1, tmp_path, "bf16", 0.25)
    r0, r1 = _run(2, tmp_path, "bf16", 0.25)
    assert np.array_equal(r0["G"], r1["G"]) and np.array_equal(r0["P"], r1["P"]) and np.array_equal(r0["bn_final"], r1["bn_final"])
    got = np.concatenate([r0["preds"], r1["preds"]], 0)
    scale = float(np.abs(one["preds"]).max())
    e = float(np.abs(got - one["preds"]).max())
    print(f"bf16 2-rank vs 1-process logits: max err {e:.3e} at logit scale {scale:.2f}")
    assert e < 3e-2 * max(1.0, scale)
    assert abs(0.5 * (float(r0["loss"]) + float(r1["loss"])) - float(one["loss"])) < 1e-2 * abs(float(one["loss"]))
    names = meta["param_names"]
    a, b = r0["grad_norms"], one["grad_norms"]
    rel = np.abs(a - b) / (np.abs(b) + 1e-6)
    bn_params = np.array([(".bn" in n or "downsample.1" in n or "connect.1" in n or "norm_layer" in n or n.endswith(".1.weight") or n.endswith(".1.bias"))
                          for n in names])
    big = b > 1e-4
    print(f"bf16 2-rank vs 1-process gradient norms: median rel {np.median(rel[big]):.2e}, worst {rel[big].max():.2e}; "
          f"BatchNorm-parameter tensors worst {rel[big & bn_params].max():.2e}")
    # a wrong 1/world factor on the BatchNorm parameter gradients (or on `count`) would be a factor 2, not a few per cent
    assert rel[big & bn_params].max() < 0.25 and np.median(rel[big]) < 3e-2
    assert np.allclose(r0["bn_checksum"], one["bn_checksum"], rtol=2e-3, atol=2e-2)
    ga, gb = r0["G"].astype(np.float64), one["G"].astype(np.float64)
    cos = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
    assert cos > 0.98, cos
