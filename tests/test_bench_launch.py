"""bench.py as its own launcher (train_crog.py:67-78 spawns one worker per GPU): `python bench.py --gpus N` with no
torch.distributed.run around it must start N ranks, and a launcher whose WORLD_SIZE disagrees with --gpus must fail instead of
reporting the wrong N.  --dry runs the launcher / rendezvous / max-over-ranks timing protocol on gloo with CPU tensors only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return env


def test_gpus_2_without_a_launcher_spawns_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry", "--steps", "3", "--warmup", "1"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["dry"] is True
    assert out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 64


def test_under_a_launcher_the_ranks_are_not_respawned():
    """torch.distributed.run shape: WORLD_SIZE / RANK come from the environment; each process is one rank."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(_env(), WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--dry", "--steps", "2", "--warmup", "0"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1000:] for o in outs]
    assert json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])["n_gpus"] == 2
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]      # only rank 0 prints


def test_world_size_mismatch_is_an_error_not_a_silent_n1():
    env = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "does not match" in r.stderr


def test_more_gpus_than_devices_is_an_error():
    """No GPU in the build container (and one on the GPU box): --gpus 8 without --dry must exit non-zero before spawning."""
    import torch
    if torch.cuda.device_count() >= 8:
        return
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "1", "--warmup", "0"], env=_env(), capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "visible" in r.stderr
