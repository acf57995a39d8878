#!/usr/bin/env python3
"""Headline benchmark: training images/sec of CROG-R50, 416x416, 20 tokens, batch 32 per GPU (BASELINE.json).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = forward (bf16 autocast) + losses + backward + gradient all-reduce (N > 1) + fused Adam + train metric on one
synthetic batch that is already resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

FLOP_PER_IMG = 413.6e9          # fwd+bwd, 2*MAC (probe of the reference)
BYTES_PER_IMG = 1.58e9          # algorithmic HBM bytes per image, bf16 activations, contraction-level fusion
BYTES_PER_STEP_WEIGHTS = 5.6e9  # per-step weight-side traffic (bf16 weight reads x3, fp32 grad write, Adam 7 streams)
PEAK_HBM_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_MFMA_TF = 2500.0           # dense bf16 MFMA


def _cpu_info():
    """(physical cores usable by this process, logical CPUs, model string) from /proc/cpuinfo + the affinity mask."""
    model, cores, logical = "unknown CPU", set(), 0
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "processor":
                logical += 1
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
                cores.add((phys, core))
    except OSError:
        pass
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    n_phys = len(cores) or allowed
    return max(1, min(n_phys, allowed)), logical or (os.cpu_count() or 1), model


CPU_WARMUP, CPU_TIMED = 5, 5     # >= 5 warm-up steps (the first ones are 3-10x slower), then the mean of >= 5


def cpu_baseline_worker(path, threads):
    """Child process: the CPU oracle's training step under a gloo DistributedDataParallel wrapper at world size 1 (what the
    reference's train_crog.py would be on CPU), appending each step's seconds to `path` as it goes."""
    import socket
    from crog_amd.testing import make_cfg, seeded_state, synthetic_batch
    from crog_amd.model import build_crog
    from oracle import crog_oracle as O
    torch.set_num_threads(threads)
    cfg = make_cfg(dropout=0.0)
    model, _ = build_crog(cfg)  # only for names/shapes (CPU tensors; never run)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    names = [n for n, _ in model.named_parameters()]
    del model
    P = seeded_state(shapes, seed=5, residual_gain=0.25)
    b = synthetic_batch(2, 416, 20, 49408, seed=1)

    class Oracle(torch.nn.Module):       # the functional oracle behind an nn.Module so that DDP can wrap it
        def __init__(self):
            super().__init__()
            self.params = torch.nn.ParameterList([torch.nn.Parameter(P[n]) for n in names])

        def forward(self, img, word, *masks):
            state = dict(P)
            state.update({n: p for n, p in zip(names, self.params)})
            return O.crog_forward(state, img, word, list(masks), num_head=cfg.num_head)["total"]

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    net = torch.nn.parallel.DistributedDataParallel(Oracle(), find_unused_parameters=True)     # train_crog.py:154-156
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    times = []
    for i in range(CPU_WARMUP + CPU_TIMED):
        t0 = time.time()
        loss = net(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
        opt.zero_grad()
        loss.backward()
        opt.step()
        times.append(time.time() - t0)
        json.dump(times, open(path, "w"))
    dist.destroy_process_group()


def cpu_baseline(budget_s=150):
    """The CPU oracle (oracle/crog_oracle.py, a restatement of the reference's PyTorch-CPU path) doing the same training step
    (fwd + losses + bwd + Adam, gloo DDP world size 1) on BASELINE config 1: CROG-R50, B=2, 416x416, fp32, on this box's host
    cores: 5 warm-up + 5 timed steps in a child process (killed at the wall-clock budget; whatever completed is reported)."""
    import subprocess
    import tempfile
    n_phys, logical, model = _cpu_info()
    # B = 2 convolutions scale NEGATIVELY past ~32 threads on the 128-core EPYC hosts of this pool (measured: 6.16 s/step at 128 threads,
    # 1.2 s/step at 32), so the baseline gets the thread count that serves it best, capped by the physical cores present
    threads = min(n_phys, 32)
    path = os.path.join(tempfile.mkdtemp(), "cpu_steps.json")
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", path, "--cpu-threads", str(threads)],
                         stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        p.wait(timeout=budget_s)
    except subprocess.TimeoutExpired:
        p.kill()
        p.wait()
    times = json.load(open(path)) if os.path.exists(path) else []
    where = f"{model}; torch CPU threads = {threads} of {n_phys} physical cores ({logical} logical CPUs)"
    if not times:
        return dict(value=None, unit="images/sec", cores=threads, kind="port", sample=f"no CPU step finished within {budget_s} s; {where}")
    used = times[CPU_WARMUP:] if len(times) > CPU_WARMUP else times[-1:]
    t = sum(used) / len(used)
    return dict(value=round(2.0 / t, 4), unit="images/sec", cores=threads, kind="port",
                sample=f"oracle training step (fwd+loss+bwd+Adam, gloo DDP world size 1), CROG-R50 fp32, B=2, 416x416, 20 tokens; {len(times)} steps "
                       f"in <= {budget_s} s (first {times[0]:.1f} s), mean of the last {len(used)} after {min(CPU_WARMUP, len(times) - len(used))} warm-ups: "
                       f"{t:.2f} s/step; {where}")


def measured_peaks(dev):
    """The box's own stream-copy bandwidth and bf16 MFMA issue rate, next to the vendor peaks the fractions use."""
    from crog_amd import kernels as K
    lib = K.lib()
    out = {}
    n = 1 << 30
    src = torch.empty(n, device=dev, dtype=torch.uint8).random_(0, 255)
    dst = torch.empty_like(src)
    s = torch.cuda.current_stream().cuda_stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    rates = {}
    for mode in range(6):          # V = 1/2/4 vectors per lane, non-temporal (0-2) or plain (3-5) loads/stores: report the best shape
        for _ in range(2):
            K.check(lib.crog_probe_copy(src.data_ptr(), dst.data_ptr(), n, mode, s), "probe_copy")
        ev[0].record()
        for _ in range(10):
            K.check(lib.crog_probe_copy(src.data_ptr(), dst.data_ptr(), n, mode, s), "probe_copy")
        ev[1].record()
        torch.cuda.synchronize()
        rates[mode] = round(10 * 2 * n / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9, 1)
    best = max(rates, key=rates.get)
    out["hbm_copy_GBps"] = rates[best]
    out["hbm_copy_modes_GBps"] = rates
    del src, dst
    blocks, iters = 256 * 2, 4000
    sink = torch.zeros(blocks * 256, device=dev)
    K.check(lib.crog_probe_mfma_bf16(sink.data_ptr(), blocks, 200, s), "probe_mfma")
    ev[0].record()
    K.check(lib.crog_probe_mfma_bf16(sink.data_ptr(), blocks, iters, s), "probe_mfma")
    ev[1].record()
    torch.cuda.synchronize()
    out["mfma_bf16_TFLOPs"] = round(blocks * 4 * iters * 8 * 32768 / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e12, 1)
    out["note"] = ("crog_probe_copy: 1 GiB device-to-device copy, one block per 256*V consecutive 16-byte vectors (no grid-stride loop), read + write bytes / time, best of six shapes; crog_probe_mfma_bf16: 2 blocks x 4 waves per CU issuing "
                   "independent v_mfma_f32_32x32x16_bf16 from registers (no memory traffic): what the chip sustains at the clock it holds")
    return out


def _spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes, one per GPU (train_crog.py:67-78 spawns its
    per-GPU workers the same way), BEFORE this process has made any GPU call - a process that initialised the GPU must never be
    replaced or forked.  Rank 0's JSON line goes straight to our stdout; the exit code is the worst child's."""
    import socket
    import subprocess
    n = args.gpus
    if not args.dry:
        have = torch.cuda.device_count()          # counting devices does not initialise the GPU on this image
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
            return 2
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                r = p.poll()
                if r is None:
                    continue
                pending.remove(p)
                if r != 0:
                    rc = rc or r
                    for q in pending:            # one rank failed: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def _dry_rank(args, world, rank):
    """--dry: the launcher / rendezvous / timing protocol only (gloo, CPU tensors, no model, no GPU): what `tests/test_bench_launch.py`
    runs in the build container."""
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    t0 = time.perf_counter()
    x = torch.ones(4)
    for _ in range(args.steps):
        dist.all_reduce(x)
    dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    ranks = torch.zeros(world, dtype=torch.int64)
    ranks[rank] = 1
    dist.all_reduce(ranks)
    if rank == 0:
        print(json.dumps({"metric": "training images/sec CROG-R50 416x416 bs32/GPU", "value": None, "unit": "images/sec", "dry": True,
                          "n_gpus": world, "ranks_seen": int(ranks.sum()), "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(float(tmax) / max(args.steps, 1) * 1e3, 3), "scaling": "weak",
                          "config": {"workload": "launcher dry run (gloo, no model)", "global_batch": args.batch * world,
                                     "parallelism": f"dp{world}"}}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=8, help=argparse.SUPPRESS)
    ap.add_argument("--roofline-kernel", default="conv3x3_fwd", choices=["conv3x3_fwd", "conv3x3_dgrad", "conv3x3_wgrad", "lin_fwd", "lin_wgrad", "none"])
    ap.add_argument("--dry", action="store_true", help="launcher / rendezvous check only: gloo on CPU tensors, no model, no GPU")
    ap.add_argument("--eager", action="store_true", help="issue every step from Python instead of replaying the captured whole-step hipGraph")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.cpu_baseline_worker, args.cpu_threads)
        return 0

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return _spawn_ranks(args)            # no launcher: become one (the children come back here with WORLD_SIZE set)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} does not match the launcher's WORLD_SIZE={world}", file=sys.stderr)
        return 2
    if args.dry:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        _dry_rank(args, world, rank)
        return 0
    if torch.cuda.device_count() < (world if world > 1 else 1):
        print(f"bench.py: {world} rank(s) but only {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
        return 2
    force_ddp = os.environ.get("CROG_FORCE_DDP") == "1"   # exercise the RCCL path on a single GPU (smoke test of the N > 1 code)
    if world > 1 or force_ddp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        # no `device_id=`: binding the process group to the device at init (eager communicator setup) measured 2.8 ms/step
        # slower for the whole step on this stack (43.6 vs 40.8 ms with no collective issued at all); the device is selected by
        # torch.cuda.set_device above and the barrier below names it explicitly
        dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    from crog_amd import kernels as K
    from crog_amd.engine import train_step
    from crog_amd.model import build_crog
    from crog_amd.optim import FusedAdam
    from crog_amd.parallel import DistributedDataParallel, convert_sync_batchnorm
    from crog_amd.runtime import RT
    from crog_amd.testing import make_cfg, synthetic_batch

    cfg = make_cfg(batch_size=args.batch * world, input_size=args.size)  # crog_multiple_r50.yaml keys, dropout 0.1, sync_bn True
    torch.manual_seed(0)
    model, groups = build_crog(cfg)            # random init of the RN50 architecture (no checkpoint / network here)
    model = model.to(dev)
    model.prepare(dev)
    if os.environ.get("CROG_SINGLE_STREAM") == "1":      # profiling aid: one HIP stream, so that per-kernel durations are not inflated by overlap
        RT.overlap_wgrad = False
        model.overlap_text = False
    net = model
    if world > 1 or force_ddp:
        convert_sync_batchnorm(model, force=force_ddp)
        net = DistributedDataParallel(model, device_ids=[local_rank], find_unused_parameters=True, force=force_ddp)
    opt = FusedAdam(groups, lr=cfg.base_lr, weight_decay=cfg.weight_decay, store=model.store)
    RT.manual_seed(1234 + rank)
    batch = synthetic_batch(args.batch, args.size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=1234 + rank, device=dev)
    net.train()
    adt = torch.bfloat16 if args.dtype == "bf16" else None

    def sync():
        if world > 1 or force_ddp:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    # Default: the step is captured once (ordinary stream capture) after 3 eager steps and then re-issued per step by ONE C call
    # (crog_amd/graphs.py, csrc/replay.hip: same kernels, arguments, streams and cross-stream edges as the eager step) instead of ~1300
    # launches through Python.  --eager (or CROG_STEP_GRAPH=0) issues every step from Python as rounds 1-2 did;
    # CROG_STEP_GRAPH=hipgraph replays with hipGraphLaunch instead.  Multi-GPU runs replay too when every collective of the step goes
    # through the C-ABI communicators (crog_amd.parallel.step_is_capturable: RCCL / mailbox launches on captured streams, chosen by a
    # start-up self-test); the first replay is then checked against an eager step from the same state (GraphedTrainStep verify=) and
    # the run falls back to eager steps in the same process if it does not reproduce it.
    KEYS = {"conv3x3_fwd": (K.A_IM2COL, K.B_KC), "conv3x3_dgrad": (K.A_IM2COL, K.B_NC_DGRAD), "conv3x3_wgrad": (K.A_MC, K.B_NC_IM2COL),
            "lin_fwd": (K.A_KC, K.B_KC), "lin_wgrad": (K.A_MC, K.B_NC), "none": None}
    key = KEYS[args.roofline_kernel]
    # `roofline` is the headline family (--roofline-kernel); `roofline_dominant` is the family of the kernel that takes the most time in the
    # step (profiles/r0N_summary.md: the 3x3 weight gradient, gemm_ppt_kernel<B_NC_IM2COL>) - bracketed in the same sampled steps
    DOMINANT = "conv3x3_wgrad"
    dom_key = KEYS[DOMINANT] if (key is not None and args.roofline_kernel != DOMINANT) else None
    keys = [k for k in (key, dom_key) if k is not None]
    graphed = None
    # (torch's process-group watchdog thread was seen to poll an event recorded inside a capture - hipErrorCapturedEvent, std::terminate -
    # once in a few runs when torch.distributed collectives were captured: with the C-ABI communicators no ProcessGroup work is in the step)
    multi = world > 1 or force_ddp
    from crog_amd.parallel import step_is_capturable
    want = os.environ.get("CROG_STEP_GRAPH", "1" if (not multi or step_is_capturable(net)) else "0")
    if not args.eager and want != "0":
        from crog_amd.graphs import GraphedTrainStep
        graphed = GraphedTrainStep(net, opt, cfg, adt, warmup=3, profile_key=(keys if len(keys) > 1 else key) if rank == 0 else None, verify=multi)

    def step(eager=False, profile=False):
        if graphed is not None:
            return graphed(batch, eager=eager, profile=profile)
        return train_step(net, opt, None, batch, cfg, autocast_dtype=adt)

    for _ in range(args.warmup):
        stats, _ = step()
    while graphed is not None and graphed.graph is None and graphed.failed is None:
        stats, _ = step()          # --warmup < 4: the capture still happens before the timed region
    sync()
    replaying = graphed is not None and graphed.graph is not None
    if replaying:
        batch = graphed.static_batch()     # the synthetic batch is resident in the captured step's own input tensors (no per-step copy, as in the eager loop)
    timers_in_replay = replaying and bool(graphed.prof_nodes)      # the replay brackets the roofline kernel's launches itself
    if key is not None and rank == 0 and not timers_in_replay:
        K.PROF = dict(key=key, keys=set(keys), records=[], on=False)
    coll0 = (RT.comm.calls if RT.comm is not None else 0, net.reducer.launches if (world > 1 or force_ddp) else 0)
    fused0 = getattr(RT.comm, "fused", 0) if RT.comm is not None else 0
    t0 = time.perf_counter()
    # Roofline leg, measured inside the timed region: a timing-only HIP event pair (no system fence) around every launch of the roofline
    # kernel, on the stream it is launched on, in every PROF_EVERY-th timed step (106 event records per bracketed step cost ~0.3 ms:
    # sampling keeps `value` untaxed).  Replayed steps carry the pairs inside the replay (crog_replay_profile_*); eager steps take them
    # in K.gemm; with hipGraphLaunch (no way to read an event out of a graph) steps K/3 and 2K/3 are issued eagerly instead.
    PROF_EVERY = 5
    want_prof = key is not None and rank == 0
    prof_steps = ({i for i in range(args.steps) if i % PROF_EVERY == 0} if (not replaying or timers_in_replay)
                  else {args.steps // 3, (2 * args.steps) // 3})
    sampled = 0
    for i in range(args.steps):
        prof = want_prof and i in prof_steps
        sampled += prof
        if K.PROF is not None:
            K.PROF["on"] = prof
        stats, _ = step(eager=prof and replaying and not timers_in_replay, profile=prof and timers_in_replay)
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    last = stats.tolist()
    if not all(v == v and abs(v) != float("inf") for v in last):
        # a number measured on a step that produces NaN / inf is not a measurement of the training step
        print(f"bench.py: non-finite training statistics after the timed region (loss, IoU, Prec@50 = {last}): refusing to report a throughput", file=sys.stderr)
        return 3

    if rank == 0:
        ms = dt / args.steps * 1e3
        ips = args.batch * world * args.steps / dt
        roof = roof_dom = None
        recs = None
        if timers_in_replay:
            recs = [(ms_ * 1e-3, f, meta) for ms_, f, meta in graphed.profile_records()]
        elif K.PROF is not None and K.PROF["records"]:
            recs = [(e0.elapsed_time(e1) * 1e-3, f, meta) for e0, e1, f, meta in K.PROF["records"]]
            K.PROF = None

        def leg(name, k):
            """Roofline leg of one GEMM family from its bracketed launches: MFMA fraction, and the HBM fraction of its ALGORITHMIC bytes
            (each operand and the output once; a 3x3 form reads its NHWC map once, not nine times) - `bound` names the larger of the two."""
            mine = [(d, f, m) for d, f, m in recs if (m[0], m[1]) == k]
            if not mine:
                return None
            durs, flops = [d for d, _, _ in mine], [f for _, f, _ in mine]
            avg_d, avg_f = sum(durs) / len(durs), sum(flops) / len(flops)

            def alg_bytes(meta):
                al, bl, M, N, Kd, bt, sk = meta
                esz = 2 if args.dtype == "bf16" else 4
                a_el = M * Kd / (9 if al == K.A_IM2COL else 1)
                b_el = N * Kd / (9 if bl == K.B_NC_IM2COL else 1)
                out_b = M * N * (4 if al == K.A_MC else esz)
                return bt * (esz * (a_el + b_el) + out_b)
            alg = sum(alg_bytes(m) for _, _, m in mine) / len(mine)
            # HBM bytes per launch of this kernel: NOT measured in this run (PMC counters need rocprofv3 around the process) but read from
            # the committed PMC passes of the same workload, and labelled as such - it goes stale when the kernels change and the
            # round's profile is not re-taken
            traffic = traffic_source = traffic_stale = None
            try:
                pm = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")))
                if args.dtype == "bf16" and args.batch == 32:
                    traffic = pm[name]["bytes_per_launch"]
                    traffic_source = "committed profile, not this run: profiles/pmc_traffic.json (" + pm[name].get("source", "?") + ")"
                    from crog_amd import _lib
                    took = pm[name].get("source_digest")
                    traffic_stale = None if took is None else (took != _lib.source_digest())
                    if traffic_stale:
                        traffic_source += f"; STALE: counters taken on kernel sources {took}, this run's are {_lib.source_digest()}"
            except (OSError, KeyError, ValueError):
                pass
            mf, hf = avg_f / avg_d / 1e12 / PEAK_MFMA_TF, alg / avg_d / 1e9 / PEAK_HBM_GBS
            return dict(bound="mfma" if mf >= hf else "hbm", achieved=round(avg_f / avg_d / 1e12, 2), peak=PEAK_MFMA_TF, unit="TFLOP/s",
                        frac=round(mf, 4), hbm_frac=round(hf, 4), hbm_achieved_GBps=round(alg / avg_d / 1e9, 1),
                        traffic=traffic, traffic_source=traffic_source, traffic_stale=traffic_stale,
                        traffic_algorithmic=round(alg), traffic_ratio=(round(traffic / alg, 3) if traffic else None),
                        kernel=K.GEMM_SYMBOL[k], launches_per_step=len(mine) // max(sampled, 1), timed_steps_bracketed=sampled,
                        avg_launch_us=round(avg_d * 1e6, 1), avg_gflop_per_launch=round(avg_f / 1e9, 2),
                        share_of_step=round(sum(durs) / max(sampled, 1) / (dt / args.steps), 4))
        if recs and os.environ.get("CROG_BENCH_LAUNCH_TABLE"):
            # per-shape table of the bracketed family: launches per step, average microseconds, algorithmic bytes, TB/s, TFLOP/s
            esz = 2 if args.dtype == "bf16" else 4
            by = {}
            for d_, f_, m_ in recs:
                if (m_[0], m_[1]) == key:
                    by.setdefault(tuple(m_), []).append((d_, f_))
            rows_ = []
            for m_, L_ in by.items():
                al, bl, M_, N_, K_, bt, sk = m_
                alg_ = bt * (esz * (M_ * K_ / (9 if al == K.A_IM2COL else 1) + N_ * K_ / (9 if bl == K.B_NC_IM2COL else 1)) + M_ * N_ * (4 if al == K.A_MC else esz))
                us_ = sum(d for d, _ in L_) / len(L_) * 1e6
                rows_.append((us_ * len(L_) / max(sampled, 1), len(L_) / max(sampled, 1), M_, N_, K_, bt, sk, us_, alg_, alg_ / us_ / 1e6, L_[0][1] / us_ / 1e6))
            with open(os.environ["CROG_BENCH_LAUNCH_TABLE"], "w") as tf:
                tf.write(f"# {args.roofline_kernel} ({K.GEMM_SYMBOL[key]}) launches of one step, {'single stream' if os.environ.get('CROG_SINGLE_STREAM') == '1' else 'in situ'}; "
                         f"bytes = each operand and the output once\n# us/step  launches  M  N  K  batch  splitk  us/launch  MB  TB/s  TFLOP/s\n")
                for r_ in sorted(rows_, reverse=True):
                    tf.write(f"{r_[0]:8.1f} {r_[1]:5.1f} {r_[2]:8d} {r_[3]:6d} {r_[4]:8d} {r_[5]:4d} {r_[6]:3d} {r_[7]:8.1f} {r_[8] / 1e6:8.1f} {r_[9]:6.2f} {r_[10]:7.1f}\n")
                tf.write(f"# total {sum(r[0] for r in rows_):.1f} us per step in {sum(r[1] for r in rows_):.0f} launches, {sum(r[8] * r[1] for r in rows_) / 1e9:.2f} GB\n")
        if recs:
            roof = leg(args.roofline_kernel, key)
            if dom_key is not None:
                roof_dom = leg(DOMINANT, dom_key)
                if roof_dom is not None:
                    roof_dom["why"] = "the family of the kernel with the largest share of the step's kernel time (profiles/r0N_summary.md: gemm_ppt_kernel<B_NC_IM2COL>, the 3x3 weight gradient)"
        out = {
            "metric": "training images/sec CROG-R50 416x416 bs32/GPU", "value": round(ips, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"CROG-R50 {args.dtype} training step (fwd+loss+bwd+{'gradient all-reduce+' if multi else ''}Adam+metric), "
                                   f"{args.size}x{args.size} RGB + 20 tokens, batch {args.batch}/GPU, dropout 0.1, "
                                   f"{'SyncBatchNorm, ' if multi else ''}random-init RN50 architecture",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            "step_roofline": {"hbm_frac": round((BYTES_PER_IMG * args.batch + BYTES_PER_STEP_WEIGHTS) / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                              "mfma_frac": round(FLOP_PER_IMG * args.batch / (ms * 1e-3) / 1e12 / PEAK_MFMA_TF, 4),
                              "note": "per-GPU algorithmic bytes (1.58 GB/img + 5.6 GB/step) and FLOPs (413.6 GFLOP/img) / step time vs 8 TB/s, 2.5 PFLOP/s"},
            "roofline": roof,
            "roofline_dominant": roof_dom,
            "last_step": {"loss": round(last[0], 4), "iou": round(last[1], 3), "prec50": round(last[2], 3)},
        }
        out["step_issue"] = ({"mode": "replay:" + graphed.executor, "replays": graphed.replays, "captured": graphed.replay_info,
                              "eager_steps_in_timed_region": 0 if timers_in_replay else sampled}
                             if replaying else {"mode": "eager", "graph_capture_failed": getattr(graphed, "failed", None)})
        if world > 1 or force_ddp:
            per_step = (lambda now, then: (now - then) // args.steps)
            sb = per_step(RT.comm.calls, coll0[0]) if RT.comm is not None else 0
            gb = per_step(net.reducer.launches, coll0[1])
            fb = (getattr(RT.comm, "fused", 0) - fused0) // args.steps if RT.comm is not None else 0
            if replaying:      # the Python counters only move in eager steps: a replay re-issues what the capture recorded
                sb, gb, fb = graphed.collectives["syncbn"], graphed.collectives["buckets"], graphed.collectives.get("fused", 0)
            if RT.comm is not None:
                RT.comm.check()      # a timed-out mailbox exchange poisons the statistics: never report a throughput measured on it
            out["collectives_per_step"] = {"syncbn_allreduce": sb,
                                           "syncbn_exchanges_inside_or_right_behind_a_producing_kernel": fb,
                                           "gradient_buckets": gb,
                                           "syncbn_transport": RT.comm.kind if RT.comm is not None else None,
                                           "bucket_transport": "crog_comm:rccl" if getattr(net, "bucket_comm", None) is not None else "torch.distributed",
                                           # one ncclAllReduce per bucket or reduce-scatter + all-gather, timed at start-up on this node (rccl.tune_bucket_algo)
                                           "bucket_schedule": getattr(net, "bucket_algo", None),
                                           "bucket_schedule_ms_64MiB": getattr(getattr(net, "bucket_comm", None), "bucket_times_ms", None),
                                           "replay_verified": getattr(graphed, "verified", None),
                                           "note": "BatchNorm statistics on a communicator of their own (crog_amd/parallel.py); metric all-reduce not counted"}
        if world == 1:
            try:
                out["measured_peaks"] = measured_peaks(dev)
            except Exception as e:      # the probes are diagnostics: never lose the bench line over them
                out["measured_peaks"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    # The JSON line must be the LAST thing on stdout.  RCCL prints a version banner through C stdio when a communicator is created; with
    # stdout redirected that sits in a buffer until the process exits - i.e. AFTER a line printed from Python (seen: five banner lines
    # behind the JSON of a forced-DDP run), and with N ranks sharing one stdout in whatever order they exit.  So every rank flushes C
    # stdio before the final barrier, the process group goes away, and only then rank 0 prints.
    import ctypes

    def flush_c_stdio():
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()

    flush_c_stdio()
    if world > 1 or force_ddp:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()
        flush_c_stdio()
    if rank == 0:
        if world > 1:
            time.sleep(0.5)      # the other ranks' last flush (above) may still be on its way to the shared stdout
        probes = [k for k in ("CROG_PROBE_TEXT_FREE",) if os.environ.get(k) == "1"]
        if probes:      # timing probes skip work inside the step: the line is labelled so that it cannot pass for a measurement of the training step
            out["valid"] = False
            out["invalid_because"] = "timing probe(s) in the environment: " + ", ".join(probes)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    sys.exit(main() or 0)
