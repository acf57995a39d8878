"""HIP-graph capture: the whole training step (GraphedTrainStep, the default of bench.py and train_with_grasp), and a fixed-shape,
launch-bound sub-network with its backward (GraphedTower, the CLIP text tower; kept for eager steps).

The CLIP text tower of CROG (reference model/clip.py:334-500, called from model/crog.py:60) works on B x 20 token rows:
about 250 forward and 400 backward launches of a few microseconds each.  Issued eagerly they cost the host ~8 ms per
step - as much as the GPU needs for the whole image stem - and the GPU idles between them whenever the main stream has
run dry.  Both passes are therefore captured once per (batch, length, dtype) into two hipGraphs that share a private
memory pool and are replayed with one host call each.

What makes that legal for this tower: shapes are static, there is no dropout (no per-step seed in a kernel argument),
parameters / bf16 shadow weights / gradient accumulators live at fixed addresses in the ParamStore, no entry point of
the C ABI allocates or synchronises, and parameter gradients are side effects on `G` (the Functions return None for
them), so the captured backward needs no autograd leaves.  Gradient-ready notifications for the DDP reducer cannot fire
from inside a replay; they are recorded during capture and re-issued after each backward replay.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from .runtime import RT


class _ReadyRecorder:
    """Stands in for RT.reducer during capture: remembers which parameters the captured backward produces."""

    def __init__(self):
        self.params = []
        self._seen = set()

    def mark_ready(self, param):
        if id(param) not in self._seen:
            self._seen.add(id(param))
            self.params.append(param)


class GraphedTower:
    """run(tokens) -> tuple of floating-point tensors, captured with its backward on `stream`."""

    def __init__(self, run, tokens: torch.Tensor, stream: "torch.cuda.Stream"):
        self.stream = stream
        self.tokens = tokens.clone()
        with torch.cuda.stream(stream), torch.no_grad():
            run(self.tokens)                     # eager warm-up: lazy initialisation must not happen inside a capture
        torch.cuda.synchronize()
        saved = (RT.reducer, RT.overlap_wgrad, RT.streams)
        rec = _ReadyRecorder()
        # inside the capture: weight gradients stay on the captured stream (a fork to the shared weight-gradient stream would
        # tie the graph to work outside it) and the end-of-backward join has no other stream to wait for
        RT.reducer, RT.overlap_wgrad, RT.streams = rec, False, []
        try:
            self.pool = torch.cuda.graph_pool_handle()
            self.fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.fwd, pool=self.pool, stream=stream, capture_error_mode="thread_local"):
                with torch.enable_grad():
                    outs = run(self.tokens)
            self.outs = tuple(outs)
            self.grads = tuple(torch.zeros_like(o) for o in self.outs)
            self.bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.bwd, pool=self.pool, stream=stream, capture_error_mode="thread_local"):
                torch.autograd.backward(self.outs, self.grads)
        finally:
            RT.reducer, RT.overlap_wgrad, RT.streams = saved
        self.params = rec.params

    def __call__(self, anchor: torch.Tensor, tokens: torch.Tensor):
        """anchor: any parameter of the tower that requires grad (ties the node into the autograd graph)."""
        return _Replay.apply(anchor, self, tokens)


class _Replay(Function):
    @staticmethod
    def forward(ctx, _anchor, tower: GraphedTower, tokens):
        tower.tokens.copy_(tokens)
        tower.fwd.replay()
        ctx.tower = tower
        return tuple(o.detach() for o in tower.outs)   # fresh aliases of the static outputs: each call gets its own autograd history

    @staticmethod
    def backward(ctx, *grads):
        tower = ctx.tower
        for dst, g in zip(tower.grads, grads):
            if g is None:
                dst.zero_()
            else:
                dst.copy_(g)
        tower.bwd.replay()
        if RT.reducer is not None:
            for p in tower.params:
                RT.reducer.mark_ready(p)
        return None, None, None


_FAILED_CAPTURES = []


UPDATE_TOL = 0.05      # first-replay check: relative L2 distance of the replayed step's (dm, dv) from the eager step's, or 4x what two eager steps differ by
UPDATE_TOL_P = 0.9     # ... and of dP (see _verified_first_replay)


class GraphedTrainStep:
    """engine.train_step (crog_engine.py:60-90: autocast forward, zero_grad, backward, optimizer step, train metric, rank-averaged
    scalars) as ONE hipGraph launch per step.

    Why: an eager step is ~1300 kernel launches issued through Python and ctypes, ~30 ms of host time at any batch size, so the
    step could never be shorter than that however fast the kernels are (B = 8: 24 ms for 10 ms of GPU work).  A replay costs the host
    one call.

    What makes the capture legal (and keeps a replay equal to the eager step it replaces):
      * shapes are static (the loader drops the ragged last batch, train_crog.py:192); inputs are copied into the captured step's
        own input tensors (a loader's H2D copy can target them directly: `static_batch()`),
      * no entry point of the C ABI allocates or synchronises; parameters, gradients, bf16 shadow and Adam moments live at fixed
        addresses in the ParamStore; activations come from the graph's private pool,
      * per-step scalars live in device memory: dropout seeds = captured seed + a device epoch that the step itself advances
        (crog_set_seed_epoch), Adam's step count / bias corrections / learning rates (FusedAdam(capturable=True)),
      * the side streams (text tower, weight gradients) fork from and re-join the capture stream inside the step, so their
        overlap is part of the graph; DDP's bucket all-reduces and the SyncBatchNorm exchanges are RCCL launches on captured streams
        and replay with it (the reducer's Python bookkeeping ran once, at capture).
    The first `warmup` calls run eagerly (lazy initialisation, RCCL communicator set-up, the store's steady-state flags); if the
    capture raises, every later call stays eager and `self.failed` says why.  Eager and replayed steps draw the same seed sequence,
    so they can be mixed (bench.py brackets kernels with timers in an occasional eager step)."""

    def __init__(self, model, optimizer, args=None, autocast_dtype=torch.bfloat16, warmup: int = 3, enabled: bool = True,
                 executor: str = None, profile_key=None, verify: bool = False):
        """executor: "streams" (default) = csrc/replay.hip re-issues the captured nodes on this process's own three streams, so the
        weight-gradient / text-tower overlap is the eager step's; "hipgraph" = hipGraphLaunch (ROCm 7.0 re-partitions the branches
        over its own queues: 38.6 instead of 33 ms per CROG-R50 step, kept for A/B).  CROG_STEP_GRAPH=streams|hipgraph overrides.
        profile_key: (a_layout, b_layout) of the GEMM variant whose launches get timer pairs in profiled replays (bench.py)."""
        import os
        # verify: check the FIRST replay against an eager step from the same state (parameters, moments, running statistics, seed
        # epoch are snapshotted and rewound) before trusting the graph; on a mismatch - or any exception - the state is rewound once
        # more, the step is issued eagerly and every later step stays eager (`self.failed` says why).  With several ranks the verdict is
        # a MIN all-reduce, so all ranks replay or none does.  Default for multi-rank jobs (captured collectives), CROG_REPLAY_VERIFY=0/1.
        self.verify = {"0": False, "1": True}.get(os.environ.get("CROG_REPLAY_VERIFY", ""), verify)
        self.verified = None
        self.executor = executor or {"hip": "hipgraph", "hipgraph": "hipgraph"}.get(os.environ.get("CROG_STEP_GRAPH", ""), "streams")
        self.profile_key = profile_key
        self.prof_nodes = []            # (node handle, flops, meta) of the profiled launches, capture order
        self.prof_ms = []               # per profiled replay: list of milliseconds, same order
        self._prof_pending = False
        self.replay_handle = None
        self.replay_info = None
        self.model, self.optimizer, self.args, self.autocast_dtype = model, optimizer, args, autocast_dtype
        self.warmup = max(2, warmup)      # >= 2: the first step casts the bf16 shadow and builds the optimizer state
        self.enabled = enabled and torch.cuda.is_available()
        self.calls = 0
        self.graph = None
        self.failed = None
        self.static = None
        self.replays = 0
        self._seed0 = None
        self._seeds_per_step = None
        if self.enabled and not getattr(optimizer, "capturable", False):
            if not hasattr(optimizer, "sync_lr"):
                raise TypeError("GraphedTrainStep needs crog_amd.optim.FusedAdam (device-resident step state); pass enabled=False for other optimizers")
            optimizer.capturable = True

    # ---- the step itself (what engine.train_step does, minus GradScaler: bf16 needs none) ------------------------------------
    def _body(self, batch):
        from . import functional as Fn
        import torch.distributed as dist
        model, opt = self.model, self.optimizer
        adt = self.autocast_dtype
        from .engine import EARLY_ZERO
        if EARLY_ZERO and getattr(opt, "_store", None) is not None:
            opt._store.request_zero()      # (engine.train_step: the gradient memset may run on the forward's side stream)
        with torch.autocast("cuda", dtype=adt or torch.bfloat16, enabled=adt is not None):
            pred, target, loss, loss_dict = model(batch["img"], batch["word"], batch["mask"], batch["qua"], batch["sin"], batch["cos"], batch["wid"])
        m = Fn.train_metric_beside(pred[0], target[0], 0.35, 0.5)      # (beside the start of backward, not behind the last Adam launch: engine.train_step)
        opt.zero_grad()
        max_norm = getattr(self.args, "max_norm", 0.0) if self.args is not None else 0.0
        from .engine import ADAM_OVERLAP
        if ADAM_OVERLAP and not max_norm and hasattr(opt, "overlap_backward"):
            opt.overlap_backward()
        loss.backward()
        if max_norm:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        opt.step()
        stats = torch.stack([loss.detach().float(), m[0], m[1]])
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from .parallel import all_reduce_mean_
            all_reduce_mean_(stats, model)
        return stats, loss_dict

    # ---- first-replay check -----------------------------------------------------------------------------------------------------
    def _snapshot(self):
        """Everything a training step changes, device tensors cloned and host flags copied."""
        st, opt = self._store(), self.optimizer
        m = self.model.module if hasattr(self.model, "module") and not hasattr(self.model, "store") else self.model
        tens = dict(P=st.P, S=st.S, m=opt.m, v=opt.v, hyper=opt._hyper, epoch=RT.seed_epoch)
        return dict(tens={k: (t, t.clone()) for k, t in tens.items() if t is not None},
                    bufs=[(b, b.clone()) for b in m.buffers()],
                    host=dict(step=opt._step, hyper_lr=list(opt._hyper_lr) if opt._hyper_lr is not None else None,
                              shadow_fresh=st.shadow_fresh, synced=getattr(st, "synced", False), t_fresh=dict(st.t_fresh), g_clean=st.g_clean,
                              written=set(st.written), seed_ctr=RT._seed_ctr))

    def _restore(self, snap):
        st, opt = self._store(), self.optimizer
        for t, c in list(snap["tens"].values()) + snap["bufs"]:
            t.copy_(c)
        h = snap["host"]
        opt._step, opt._hyper_lr = h["step"], (list(h["hyper_lr"]) if h["hyper_lr"] is not None else None)
        st.shadow_fresh, st.synced, st.g_clean, st.written = h["shadow_fresh"], h["synced"], h["g_clean"], set(h["written"])
        st.t_fresh.update(h["t_fresh"])
        RT._seed_ctr = h["seed_ctr"]

    def _verified_first_replay(self, batch):
        """-> (stats, loss_dict) of the step, replayed when the replay reproduces the eager step, eager otherwise."""
        import torch.distributed as dist
        snap = self._snapshot()
        upd = lambda: {k: snap["tens"][k][0] - snap["tens"][k][1] for k in ("P", "m", "v") if k in snap["tens"]}
        # the yardstick: the SAME eager step twice from the same state - how far two correct steps are apart (order of fp32 atomic sums; a
        # bf16 toy model amplifies that to tens of per cent of a moment's update, a full-size one to a fraction of a per cent)
        self._eager(batch)
        torch.cuda.synchronize()
        other = upd()
        self._restore(snap)
        e_stats, _ = self._eager(batch)
        torch.cuda.synchronize()
        ref = e_stats.tolist()
        # what the step did to the parameters and both Adam moments (the loss above is computed BEFORE backward, the gradient all-reduce
        # and the optimizer run: a replay that dropped or mis-ordered a captured bucket all-reduce or an Adam chunk has the same loss)
        ref_upd = upd()
        floor = {k: ((other[k] - ref_upd[k]).double().norm() / ref_upd[k].double().norm().clamp_min(1e-30)).item() for k in ref_upd}
        del other
        self._restore(snap)
        why = None
        try:
            out = self._replay_once(batch)
            torch.cuda.synchronize()
            got = out[0].tolist()
            if not all(v == v and abs(v) != float("inf") for v in got):
                why = f"the replayed step produced non-finite statistics {got}"
            elif abs(got[0] - ref[0]) > 0.05 * abs(ref[0]) + 1e-3:
                # (same state, same seeds, same batch: the two differ only by the order of fp32 atomic sums, ~1 % of the loss at most)
                why = f"the replayed step's loss {got[0]:.5f} differs from the eager step's {ref[0]:.5f} (same state, same seeds)"
            else:
                for k, r in ref_upd.items():
                    t, before = snap["tens"][k]
                    d = ((t - before) - r).double().norm().item()
                    n = r.double().norm().item()
                    # Same gradients up to the order of fp32 atomic sums (~1e-3 ... 1e-2 of a gradient): the moments' updates
                    # (1 - beta) (g - m), (1 - beta2) (g^2 - v) inherit exactly that, a missing all-reduce or Adam chunk changes them by
                    # their own size.  The PARAMETER update is Adam's normalised step: an element whose gradient is noise takes a full
                    # +-lr step of noisy sign (measured 0.27 of the update's norm between two CORRECT steps of the tiny model), so it only
                    # tells whether the optimizer ran at all.
                    tol = UPDATE_TOL_P if k == "P" else max(UPDATE_TOL, 4.0 * floor[k])
                    if not d <= tol * n + 1e-12:
                        why = (f"the replayed step's update of {dict(P='the parameters', m='exp_avg', v='exp_avg_sq')[k]} differs from the eager step's "
                               f"by {d / max(n, 1e-30):.3f} of its norm (tolerance {tol:.3f}; two eager steps differ by {floor[k]:.3f})")
                        break
        except Exception as e:
            why = f"the replay raised {e!r}"
            out = None
        ok = why is None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            flag = torch.tensor([1 if ok else 0], device=e_stats.device, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if ok and not int(flag.item()):
                ok, why = False, "another rank's first replay did not reproduce its eager step"
        self.verified = ok
        if ok:
            return out
        import warnings
        warnings.warn(f"crog_amd: captured training step rejected ({why}); steps stay eager")
        self.failed = why
        self._drop_replay()
        self.graph = None
        self._restore(snap)
        return self._eager(batch)

    def _eager(self, batch):
        """One eager step inside the seed protocol: seeds restart at the same host counter every step and the device epoch moves on by
        the number of seeds a step draws - exactly what a replay does."""
        from . import kernels as K
        if self._seed0 is None:
            self._seed0 = RT._seed_ctr
            RT.enable_seed_epoch(batch["img"].device)
        RT._seed_ctr = self._seed0
        out = self._body(batch)
        n = RT._seed_ctr - self._seed0
        if self._seeds_per_step is None:
            self._seeds_per_step = n
        elif n != self._seeds_per_step:
            raise RuntimeError(f"training step drew {n} dropout seeds, {self._seeds_per_step} before: not a static step")
        if n:
            K.counter_add(RT.seed_epoch, n)
        return out

    def _capture(self, batch):
        from . import kernels as K
        from .model import crog as crog_mod
        # the captured kernels read THESE tensors; later batches are copied into them (clones: the caller's batch is never written)
        self.static = {k: v.clone() for k, v in batch.items()}
        torch.cuda.synchronize()
        saved_text_graph, saved_prof = crog_mod.TEXT_GRAPH, K.PROF
        crog_mod.TEXT_GRAPH, K.PROF = False, None      # no graph replay and no timing events inside a capture
        own = self.executor == "streams"
        g = torch.cuda.CUDAGraph(keep_graph=True) if own else torch.cuda.CUDAGraph()
        if own and self.profile_key is not None:
            # (one (a_layout, b_layout) pair, or a list of pairs: bench.py brackets the headline family and the dominant-by-time one)
            pk = self.profile_key
            many = isinstance(pk[0], (tuple, list))
            K.CAPTURE_NODES = dict(key=None if many else tuple(pk), keys={tuple(k) for k in pk} if many else (), nodes=[])
        if own:
            K.CAPTURE_TAGS = {}
        host_step = getattr(self.optimizer, "_step", None)
        # the learning rates reach the device state BEFORE the capture opens: a fill launched inside it would become a graph node that
        # rewrites the capture-time rate on every replay, freezing any schedule (step() itself never syncs while capturing)
        self.optimizer.sync_lr()
        try:
            RT._seed_ctr = self._seed0
            # with a process group alive, its watchdog thread polls events of earlier collectives while this thread captures: "relaxed"
            # keeps such calls from other threads legal (an order-dependent abort inside the capture was seen with "thread_local")
            import torch.distributed as dist
            mode = "relaxed" if (dist.is_available() and dist.is_initialized()) else "thread_local"
            with torch.cuda.graph(g, capture_error_mode=mode):
                self._capture_main = K.stream()      # (torch captures on a stream of its own: the "caller's stream" of this capture)
                stats, loss_dict = self._body(self.static)
                if self._seeds_per_step:
                    K.counter_add(RT.seed_epoch, self._seeds_per_step)
            if RT._seed_ctr - self._seed0 != self._seeds_per_step:
                raise RuntimeError("seed count changed during capture")
        except BaseException:
            # a CUDAGraph whose capture failed half-way aborts the PROCESS when it is destroyed ("The graph should be registered to the
            # state", seen on torch 2.10 / ROCm 7.0) - also at interpreter exit, after a bench line has been printed.  The caller falls
            # back to eager steps; the broken object is parked where no destructor ever runs.
            _FAILED_CAPTURES.append(g)
            import ctypes
            ctypes.pythonapi.Py_IncRef(ctypes.py_object(g))
            raise
        finally:
            crog_mod.TEXT_GRAPH, K.PROF = saved_text_graph, saved_prof
            captured, K.CAPTURE_NODES = K.CAPTURE_NODES, None
            tags, K.CAPTURE_TAGS = K.CAPTURE_TAGS, None
            if host_step is not None:
                self.optimizer._step = host_step      # the capture ran step()'s Python once without executing anything
        if own:
            self._build_replay(g, captured, tags)
        self.graph, self._stats, self._loss_dict = g, stats, loss_dict
        self._captured_store = self._store()
        self._loss_sums = getattr(loss_dict, "_sums", None)
        self.collectives = dict(syncbn=(RT.comm.calls if RT.comm is not None else 0), buckets=(RT.reducer.launches if RT.reducer is not None else 0),
                                fused=(getattr(RT.comm, "fused", 0) if RT.comm is not None else 0))

    def _build_replay(self, g, captured, tags=None):
        import ctypes
        from . import kernels as K
        lib = K.lib()
        h = ctypes.c_void_p()
        # chain i of the replay = stream i of this list: the caller's stream, then the runtime's side streams, then whatever else the
        # capture launched on (in the order it first did)
        known = [s for s in ((RT._wgrad_stream or []) + [RT.text_stream, RT.aux_stream]) if s is not None]
        tags = tags or {}
        raws = [getattr(self, "_capture_main", None)] + [s.cuda_stream for s in known]
        extra = [r for r in tags if r not in raws]
        used = [r for r in raws if r in tags or r == raws[0]] + extra
        pairs = [(nd, used.index(r)) for r, nodes in tags.items() for nd in nodes]
        if pairs and len(used) <= 8:
            nodes = (ctypes.c_void_p * len(pairs))(*[nd for nd, _ in pairs])
            chains = (ctypes.c_int * len(pairs))(*[c for _, c in pairs])
            K.check(lib.crog_replay_build_tagged(ctypes.c_void_p(g.raw_cuda_graph()), 8, nodes, chains, len(pairs), ctypes.byref(h)), "replay_build_tagged")
            by_raw = {s.cuda_stream: s for s in known}
            self._tagged_streams = [by_raw.get(r) for r in used]
        else:
            K.check(lib.crog_replay_build(ctypes.c_void_p(g.raw_cuda_graph()), 8, ctypes.byref(h)), "replay_build")
            self._tagged_streams = None
        n = [ctypes.c_int() for _ in range(5)]
        sizes = (ctypes.c_int * 8)()
        K.check(lib.crog_replay_info(h, *[ctypes.byref(x) for x in n], sizes, 8), "replay_info")
        self.replay_handle = h
        self.replay_info = dict(nodes=n[0].value, kernels=n[1].value, chains=n[2].value, events=n[3].value, waits=n[4].value,
                                chain_sizes=list(sizes)[:n[2].value])
        # one stream per chain: the caller's current stream, then the runtime's weight-gradient and text streams, then fresh ones
        if self._tagged_streams is not None:
            # (a stream the capture used but the runtime does not own - none in CROG - gets a fresh stand-in)
            side = [s if s is not None else torch.cuda.Stream() for s in self._tagged_streams[1:]]
            while len(side) < n[2].value - 1:
                side.append(torch.cuda.Stream())
            self._replay_side = side[:n[2].value - 1]
        else:
            pool = [s for s in ((RT._wgrad_stream or []) + [RT.text_stream, RT.aux_stream]) if s is not None]
            while len(pool) < n[2].value - 1:
                pool.append(torch.cuda.Stream())
            self._replay_side = pool[:n[2].value - 1]
        if captured is not None and captured["nodes"]:
            self.prof_nodes = captured["nodes"]
            arr = (ctypes.c_void_p * len(self.prof_nodes))(*[nd for nd, _, _ in self.prof_nodes])
            K.check(lib.crog_replay_profile_nodes(h, arr, len(self.prof_nodes)), "replay_profile_nodes")

    def _replay_launch(self, profile: bool):
        import ctypes
        from . import kernels as K
        lib = K.lib()
        if self._prof_pending:
            self._read_profile()
        if self.prof_nodes:
            K.check(lib.crog_replay_profile_enable(self.replay_handle, 1 if profile else 0), "replay_profile_enable")
            self._prof_pending = bool(profile)
        raws = [K.stream()] + [s.cuda_stream for s in self._replay_side]
        arr = (ctypes.c_void_p * len(raws))(*raws)
        K.check(lib.crog_replay_launch(self.replay_handle, arr, len(raws)), "replay_launch")

    def _read_profile(self):
        import ctypes
        from . import kernels as K
        out = (ctypes.c_float * len(self.prof_nodes))()
        K.check(K.lib().crog_replay_profile_read(self.replay_handle, out, len(self.prof_nodes)), "replay_profile_read")
        self.prof_ms.append(list(out))
        self._prof_pending = False

    def profile_records(self):
        """[(milliseconds, flops, meta)] of every profiled launch so far (waits for the last profiled replay)."""
        if self._prof_pending:
            self._read_profile()
        return [(ms, f, meta) for step in self.prof_ms for ms, (_, f, meta) in zip(step, self.prof_nodes)]

    def _drop_replay(self):
        """Destroy the replay object (and its ~200 HIP events) before the graph it walks goes away."""
        h, self.replay_handle = self.replay_handle, None
        if h is not None:
            try:
                from . import kernels as K
                K.lib().crog_replay_destroy(h)
            except Exception:
                pass

    def release(self):
        self._drop_replay()
        self.graph = None
        self.static = None

    def __del__(self):
        self._drop_replay()

    def _store(self):
        m = self.model.module if hasattr(self.model, "module") and not hasattr(self.model, "store") else self.model
        return getattr(m, "store", None)

    def static_batch(self):
        """The captured step's input tensors (None before capture): write the next batch straight into them to skip the copy."""
        return self.static

    def __call__(self, batch, eager: bool = False, profile: bool = False):
        """-> (stats [loss, 100*IoU, 100*Prec@50] as a fresh 3-element device tensor, loss_dict).  eager=True issues this step
        from Python even when a graph exists (same results); profile=True puts timer pairs around the `profile_key` launches of
        this replay (executor "streams" only; read them with profile_records())."""
        self.calls += 1
        if not self.enabled or self.failed is not None:
            return self._eager(batch)
        if self.graph is None:
            if self.calls <= self.warmup:
                return self._eager(batch)
            try:
                n0 = (RT.comm.calls if RT.comm is not None else 0, RT.reducer.launches if RT.reducer is not None else 0,
                      getattr(RT.comm, "fused", 0) if RT.comm is not None else 0)
                self._capture(batch)
                self.collectives = dict(syncbn=self.collectives["syncbn"] - n0[0], buckets=self.collectives["buckets"] - n0[1],
                                        fused=self.collectives.get("fused", 0) - n0[2])
            except Exception as e:          # stay correct: an uncapturable configuration trains eagerly
                import warnings
                self.failed = repr(e)
                self.graph = None
                self._drop_replay()
                torch.cuda.synchronize()
                st = self._store()
                if st is not None:
                    # the aborted capture ran the step's Python (zero_grad's "known clean" flag, the shadow / transposed-weight freshness
                    # flags) without executing a kernel: forget what it claimed before falling back to eager steps
                    st.g_clean = False
                    st.invalidate_shadow()
                warnings.warn(f"crog_amd: whole-step hipGraph capture failed ({e!r}); steps stay eager")
                return self._eager(batch)
        if eager or any(batch[k].shape != t.shape or batch[k].dtype != t.dtype for k, t in self.static.items()):
            return self._eager(batch)       # asked for, or a batch of another shape (a ragged last batch): issue it from Python
        store = self._store()
        if store is not self._captured_store or not store.valid():
            # the parameters were re-materialised (.cuda() / .to() / .float()): the captured addresses are gone - capture again
            self._drop_replay()
            self.graph, self.calls = None, self.warmup
            return self._eager(batch)
        if store.S is not None and not getattr(store, "synced", False):
            # the fp32 parameters changed behind the captured step's back (load_state_dict, an in-place edit): the replay assumes
            # the bf16 shadow FusedAdam wrote last step; one eager step re-casts it and leaves the state a replay expects
            return self._eager(batch)
        if self.verify and self.verified is None:
            return self._verified_first_replay(batch)
        return self._replay_once(batch, profile)

    def _replay_once(self, batch, profile: bool = False):
        for k, dst in self.static.items():
            src = batch[k]
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.optimizer.sync_lr()
        if self.replay_handle is not None:
            self._replay_launch(profile)
        else:
            self.graph.replay()
        self.optimizer.replayed()
        self.replays += 1
        from .model.crog import LossDict
        ld = self._loss_dict
        if self._loss_sums is not None:
            ld = LossDict(self._loss_sums.clone(), ld._heads)
        return self._stats.clone(), ld
