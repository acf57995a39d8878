"""Fused Adam over the flat parameter store (torch.optim.Adam semantics: train_crog.py:119-121).

One kernel launch per contiguous learning-rate segment of the flat buffer (CROG: ~6 launches for 449
tensors / 147 M parameters) instead of a multi-tensor loop; moments live in two flat fp32 buffers whose
per-parameter views are exposed through `optimizer.state`, so `state_dict()` has torch.optim.Adam's layout
(`exp_avg`, `exp_avg_sq`, `step`) and round-trips with reference checkpoints (train_crog.py:213,259).
"""
from __future__ import annotations

import os
from typing import List

import torch

from . import kernels as K
from .runtime import ALIGN, RT, ParamStore

# 0 (default since the end of round 5): a chunk is stepped the moment its last announcement arrives; 1: when the NEXT chunk completes (rounds 4-5:
# a margin for an announcement that came before the last reader of the weight was enqueued - WRef.done() is placed behind that reader everywhere,
# and scripts/adam_late_check.py / tests/test_engine_gpu.py hold the parameters and both moments to the bits of the update done in step()).
# The margin kept the text tower's last 15.8 M-parameter chunk until the end of backward: 83 us of the step's tail.
ADAM_LATE = int(os.environ.get("CROG_ADAM_LATE", "0"))


class _Chunk:
    """A contiguous piece of one learning-rate segment of the flat buffer: the unit of the overlapped update."""
    __slots__ = ("gi", "off", "numel", "params", "expect", "pending", "fired")

    def __init__(self, gi, off):
        self.gi, self.off, self.numel, self.params = gi, off, 0, []
        self.expect = self.pending = 0
        self.fired = False


class FusedAdam(torch.optim.Optimizer):
    # elements per chunk of the overlapped update (16 M floats = 64 MiB of parameters: ~0.1 ms of Adam, nine chunks for CROG-R50)
    CHUNK_ELEMS = 1 << 24
    FIRST_CHUNK_ELEMS = 1 << 20
    # ... and the second one as well (CROG: the rest of layer2 and the first convolutions of layer3).  A chunk is stepped one chunk LATE, so
    # with a 16 M-parameter second chunk the end of backward launched the small first chunk AND 16 M parameters (89 us behind the last
    # weight gradient); now the big one goes when the second completes, ~2 ms before the end
    SECOND_CHUNK_ELEMS = int(os.environ.get("CROG_ADAM_SECOND_CHUNK", str(1 << 21)))

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, store: ParamStore = None, capturable: bool = False):
        """capturable: the step count, its bias corrections and the learning rates live in device memory (one float[4] per
        parameter group, advanced by a one-thread kernel), so `step()` has no per-step scalar in a kernel argument and can be
        captured into a hipGraph and replayed (crog_amd.graphs.GraphedTrainStep switches it on).  Same arithmetic either way."""
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._store = store
        self._segments = None
        self._step = 0
        self.m = self.v = None
        self.capturable = capturable
        self._hyper = None          # [groups, 4] device floats {lr, 1 - beta1^t, sqrt(1 - beta2^t), t}
        self._hyper_lr = None       # the learning rates last written into it
        # overlapped update (overlap_backward): chunks of the flat buffer are stepped inside backward, as their gradients become final
        self._chunks = None         # list of _Chunk, built with the segments
        self._chunk_of = {}         # id(param) -> chunk
        self._done_counts = None    # id(param) -> gradient-ready announcements per backward, learnt from the first armed backward
        self._recording = None
        self._armed = False
        self._ripe = []
        self._advanced = set()      # groups whose device step state has been advanced for the step in flight
        self.early_launches = 0     # Adam launches issued inside backward so far (tests, bench.py)

    def attach(self, store: ParamStore):
        self._store = store
        self._segments = None

    def _build(self):
        store = self._store
        if store is None:
            raise RuntimeError("FusedAdam needs the model's ParamStore: FusedAdam(..., store=model.store) or .attach(model.store)")
        if self.m is None or self.m.numel() != store.total:
            self.m = torch.zeros(store.total, device=store.device, dtype=torch.float32)
            self.v = torch.zeros(store.total, device=store.device, dtype=torch.float32)
        self._segments = []
        for group in self.param_groups:
            ext = []
            for p in group["params"]:
                o = store.off(p)
                n = (p.numel() + ALIGN - 1) // ALIGN * ALIGN
                ext.append((o, n))
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = torch.tensor(float(self._step))
                    st["exp_avg"] = self.m[o:o + p.numel()].view(p.shape) if p.dim() != 4 else self._like(self.m, p, o)
                    st["exp_avg_sq"] = self.v[o:o + p.numel()].view(p.shape) if p.dim() != 4 else self._like(self.v, p, o)
            ext.sort()
            segs: List[List[int]] = []
            for o, n in ext:
                if segs and segs[-1][0] + segs[-1][1] == o:
                    segs[-1][1] += n
                else:
                    segs.append([o, n])
            self._segments.append(segs)
        self._build_chunks()

    def _build_chunks(self):
        """Cut every segment at parameter boundaries into pieces of about CHUNK_ELEMS (the overlapped update's launch units)."""
        store = self._store
        self._chunks, self._chunk_of = [], {}
        self._done_counts = None
        for gi, group in enumerate(self.param_groups):
            cur, made = None, 0
            for o, n, p in sorted((store.off(p), (p.numel() + ALIGN - 1) // ALIGN * ALIGN, p) for p in group["params"]):
                # (the first chunk of a group is kept small: in CROG it holds the stem / layer1 weights, whose gradients are the LAST
                # of the step - what step() still has to update after backward is then ~1 M parameters, not 16 M)
                cap = self.FIRST_CHUNK_ELEMS if made == 1 else self.SECOND_CHUNK_ELEMS if made == 2 else self.CHUNK_ELEMS
                if cur is None or cur.off + cur.numel != o or cur.numel + n > cap:
                    cur = _Chunk(gi, o)
                    self._chunks.append(cur)
                    made += 1
                cur.numel += n
                cur.params.append(p)
                self._chunk_of[id(p)] = cur

    # ---- overlapped update ------------------------------------------------------------------------------------------------
    def overlap_backward(self):
        """Arm the overlapped update for the backward pass that follows (engine.train_step calls this right after zero_grad, when
        nothing stands between backward and step: no gradient clipping, no GradScaler, no gradient all-reduce).  While armed, every
        WRef.done() is counted; a chunk whose announcements are complete is stepped on the weight-gradient stream one chunk LATER
        (when the next chunk completes: by then every kernel of the first chunk's layers - also the data gradients that still read
        its weights - has been enqueued, and the launch waits for all streams), the rest in step().  The announcement counts per
        parameter are learnt from the first armed backward, which updates nothing early; a later backward that announces a
        parameter MORE often than learnt (not a static step) raises in step()."""
        if self._store is None or not self._store.explicit:
            return
        if any(g["weight_decay"] != 0 for g in self.param_groups):
            return      # (weight decay skips parameters without a gradient, _decay_segments: known only once backward is over)
        if RT.reducer is not None:
            self._arm_buckets()
            return
        if self._segments is None:
            self._build()
        if self.capturable and not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        self._armed = True
        self._ripe = []
        self._advanced = set()
        self._home = torch.cuda.current_stream() if torch.cuda.is_available() else None      # the stream forward ran on (the caller's)
        RT.early_adam = self
        if self._done_counts is None:
            self._recording = {}
            return
        for c in self._chunks:
            c.pending, c.fired = c.expect, False

    # ---- the same under DistributedDataParallel: per gradient bucket ------------------------------------------------------------
    def _arm_buckets(self):
        """DDP (crog_amd.parallel.Reducer with the C-ABI communicator): a gradient bucket's all-reduce is enqueued on a carrier stream
        as soon as its last gradient is announced; the bucket's slice of the parameters is stepped on that SAME stream, behind the
        collective - one bucket LATER, when the next bucket is launched (by then every kernel of the first bucket's layers, also the
        data gradients that still read its weights, has been enqueued, and the launch waits for all streams), the rest in step().
        The update no longer waits for the last all-reduce: 0.9 ms of Adam per step leave the end of the step."""
        red = RT.reducer
        if getattr(red, "direct", None) is None or not self._store.G.is_cuda:
            return      # (torch.distributed buckets are async work handles: no stream to ride on)
        if self._segments is None:
            self._build()
        if self.capturable and not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        self._bucket_mode = True
        self._bucket_done = []
        self._bucket_ripe = None
        self._advanced = set()
        self._home = torch.cuda.current_stream()
        red.after_launch = self._on_bucket

    def _on_bucket(self, b, carrier):
        prev, self._bucket_ripe = self._bucket_ripe, (b["start"], b["numel"], carrier)
        if prev is not None:
            self._step_range(*prev)

    def _step_piece(self, gi, off, n):
        store, group = self._store, self.param_groups[gi]
        b1, b2 = group["betas"]
        if self.capturable:
            hyper = self._hyper[gi]
            if gi not in self._advanced:
                self._advanced.add(gi)
                K.adam_advance(hyper, b1, b2)
            K.adam_step_dev(store.P, store.G, self.m, self.v, n, hyper, b1, b2, group["eps"], group["weight_decay"], shadow=store.S, off=off)
        else:
            K.adam_step(store.P, store.G, self.m, self.v, n, group["lr"], b1, b2, group["eps"], group["weight_decay"], self._step + 1,
                        shadow=store.S, off=off)

    def _step_range(self, start, numel, carrier):
        """Adam over [start, start + numel) of the flat buffers (cut along the learning-rate groups), on `carrier` behind everything
        enqueued so far on any stream, or - carrier None - on the current stream."""
        prev_override = K._STREAM_OVERRIDE
        if carrier is not None:
            cur = torch.cuda.current_stream()
            for s in {id(x): x for x in [cur, self._home] + list(RT.streams) if x is not None}.values():
                if s != carrier:
                    carrier.wait_stream(s)
            K.set_stream_override(carrier.cuda_stream)
        try:
            for gi, segs in enumerate(self._segments):
                for o, n in segs:
                    lo, hi = max(o, start), min(o + n, start + numel)
                    if lo < hi:
                        self._step_piece(gi, lo, hi - lo)
        finally:
            if carrier is not None:
                K.set_stream_override(prev_override)
        self._bucket_done.append((start, numel))
        self.early_launches += 1

    def _finish_buckets(self):
        """step() after a backward with per-bucket updates: everything no bucket has stepped yet, on the current stream (which has
        joined the carriers: Reducer.wait)."""
        self._bucket_mode = False
        if RT.reducer is not None:
            RT.reducer.after_launch = None
        done = sorted(self._bucket_done)
        for gi, segs in enumerate(self._segments):
            for o, n in segs:
                pos = o
                for ds, dn in done:
                    if ds + dn <= pos or ds >= o + n:
                        continue
                    if ds > pos:
                        self._step_piece(gi, pos, ds - pos)
                    pos = max(pos, ds + dn)
                if pos < o + n:
                    self._step_piece(gi, pos, o + n - pos)
        self._bucket_done, self._bucket_ripe = [], None

    def mark_ready(self, param):
        """WRef.done(): one of the kernels that write this parameter's gradient has been enqueued."""
        if not self._armed:
            return
        if self._recording is not None:
            self._recording[id(param)] = self._recording.get(id(param), 0) + 1
            return
        c = self._chunk_of.get(id(param))
        if c is None or c.expect == 0:
            return
        c.pending -= 1
        if c.pending == 0:
            if ADAM_LATE == 0:
                self._launch_chunk(c, early=True)
                return
            for r in self._ripe:
                self._launch_chunk(r, early=True)
            self._ripe = [c]

    def _launch_chunk(self, c, early: bool):
        if c.fired:
            return
        c.fired = True
        store, group = self._store, self.param_groups[c.gi]
        b1, b2 = group["betas"]
        side = RT.adam_stream() if early else None
        if side is not None:
            # gradients are written on every stream of the step (main chain: norm scales / biases; text tower: its own stream; weight
            # gradients: `side` itself): the launch waits for all of them as they stand now.  The announcement that completed a chunk can
            # come from a backward node running on ANY of these streams, so "the current stream" alone is not enough.
            cur = torch.cuda.current_stream()
            for s in {id(x): x for x in [cur, self._home] + list(RT.streams) if x is not None}.values():
                if s != side:
                    side.wait_stream(s)
            prev_override = K._STREAM_OVERRIDE      # (an announcement can arrive from inside a weight-gradient closure: Runtime.flush_group)
            K.set_stream_override(side.cuda_stream)
        try:
            if self.capturable:
                hyper = self._hyper[c.gi]
                if c.gi not in self._advanced:
                    self._advanced.add(c.gi)
                    K.adam_advance(hyper, b1, b2)
                K.adam_step_dev(store.P, store.G, self.m, self.v, c.numel, hyper, b1, b2, group["eps"], group["weight_decay"],
                                shadow=store.S, off=c.off)
            else:
                K.adam_step(store.P, store.G, self.m, self.v, c.numel, group["lr"], b1, b2, group["eps"], group["weight_decay"],
                            self._step + 1, shadow=store.S, off=c.off)
        finally:
            if side is not None:
                K.set_stream_override(prev_override)
        if early:
            self.early_launches += 1

    def _finish_overlapped(self):
        """step() of an armed backward: close the recording, or launch what backward left (the last ripe chunk, chunks of
        parameters that were not announced) on the current stream."""
        self._armed = False
        RT.early_adam = None
        store = self._store
        if self._recording is not None:
            counts, self._recording = self._recording, None
            self._done_counts = counts
            for c in self._chunks:
                c.expect = sum(counts.get(id(p), 0) for p in c.params)
            return False
        over = [c for c in self._chunks if c.pending < 0]
        if over:
            raise RuntimeError("FusedAdam.overlap_backward: a parameter's gradient was announced more often than in the recorded step "
                               "(gradient accumulation or a changing graph): parameters may have been updated before their gradient was "
                               "complete.  Do not arm the overlapped update for such steps.")
        return True

    def _decay_segments(self, group, segs):
        """torch.optim.Adam skips parameters whose gradient is None (CROG: `logit_scale`, never used by the forward): no decay,
        no moment update.  Here such a parameter has a zero gradient in the flat buffer, which only differs from being skipped
        when weight decay is on, so with decay the segments leave out every parameter no kernel has written a gradient for SINCE
        THE LAST zero_grad (`store.written`; advisor, round 2: "ever written" also decayed a parameter that is used in some steps only).
        Known deviation: all parameters share one step count for the bias correction; a parameter first used at step k is corrected
        as at step k, torch would start it at 1 (CROG has no such parameter)."""
        store = self._store
        idle = frozenset(id(p) for p in group["params"] if id(p) not in store.written)
        if not idle:
            return segs
        cache = self.__dict__.setdefault("_decay_cache", {})
        key = (id(group), idle)
        if key not in cache:
            ext = sorted((store.off(p), (p.numel() + ALIGN - 1) // ALIGN * ALIGN) for p in group["params"] if id(p) not in idle)
            out = []
            for o, n in ext:
                if out and out[-1][0] + out[-1][1] == o:
                    out[-1][1] += n
                else:
                    out.append([o, n])
            cache[key] = out
        return cache[key]

    # ---- device-resident step state (capturable mode) ---------------------------------------------------------------
    def _ensure_hyper(self):
        if self._hyper is None:
            self._hyper = torch.zeros(len(self.param_groups), 4, device=self._store.device, dtype=torch.float32)
            self._hyper[:, 3] = float(self._step)
            self._hyper_lr = [None] * len(self.param_groups)
        return self._hyper

    def sync_lr(self):
        """Write the groups' learning rates into the device state when a scheduler changed them (MultiStepLR: a few times per
        run).  Launches fill kernels, so it runs OUTSIDE a captured step: GraphedTrainStep calls it before every replay."""
        if not self.capturable or self._store is None:
            return
        hyper = self._ensure_hyper()
        for i, group in enumerate(self.param_groups):
            if self._hyper_lr[i] != group["lr"]:
                hyper[i, 0:1].fill_(float(group["lr"]))
                self._hyper_lr[i] = group["lr"]

    def replayed(self, steps: int = 1):
        """A captured step() was replayed `steps` times: the device count advanced by itself, the host mirror follows."""
        self._step += steps

    @staticmethod
    def _like(buf, p, o):
        co, ci, kh, kw = p.shape
        if kh * kw == 1:
            return buf[o:o + p.numel()].view(p.shape)
        return buf[o:o + p.numel()].view(co, kh, kw, ci).permute(0, 3, 1, 2)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        RT.join_streams()                       # weight-gradient / text-tower side streams must have landed in G
        if RT.reducer is not None:
            RT.reducer.wait()
        if self._store is not None and not self._store.valid():
            # .cuda() / .to() / prepare() after the optimizer was built moved the parameters into ANOTHER flat store: updating this one
            # would train nothing, silently
            raise RuntimeError("FusedAdam is attached to a ParamStore its parameters no longer live in (the model was moved or prepared again "
                               "after the optimizer was built): call optimizer.attach(model.store)")
        if self._segments is None:
            self._build()
        if self.capturable and not torch.cuda.is_current_stream_capturing():
            # (creates the device state from the host step count: before that count moves on.  Never inside a capture: a fill launched
            # there would be replayed with the capture-time learning rate every step - GraphedTrainStep syncs before capture / replay)
            self.sync_lr()
        store = self._store
        if getattr(self, "_bucket_mode", False):
            self._finish_buckets()
            self._step += 1
            if store.S is not None:
                store.shadow_written()
            else:
                store.invalidate_shadow()
            store.g_clean = False
            return loss
        if self._armed and self._finish_overlapped():
            # overlapped update: most chunks were stepped inside backward; the rest (and every chunk of a learning-rate group whose
            # parameters never announce a gradient) go now, on the current stream, which has joined the side streams above
            for c in self._chunks:
                self._launch_chunk(c, early=False)
            self._step += 1
            if store.S is not None:
                store.shadow_written()
            else:
                store.invalidate_shadow()
            store.g_clean = False
            return loss
        self._step += 1
        shadow = store.S      # bf16 compute copy (None until a bf16 forward has run): refreshed by the same pass that updates P
        for gi, (group, segs) in enumerate(zip(self.param_groups, self._segments)):
            b1, b2 = group["betas"]
            if group["weight_decay"] != 0 and store.explicit:
                segs = self._decay_segments(group, segs)
            if self.capturable:
                hyper = self._hyper[gi]
                K.adam_advance(hyper, b1, b2)
                for o, n in segs:
                    K.adam_step_dev(store.P, store.G, self.m, self.v, n, hyper, b1, b2, group["eps"], group["weight_decay"],
                                    shadow=shadow, off=o)
                continue
            for o, n in segs:
                K.adam_step(store.P, store.G, self.m, self.v, n, group["lr"], b1, b2, group["eps"], group["weight_decay"], self._step,
                            shadow=shadow, off=o)
        if shadow is not None:
            store.shadow_written()
        else:
            store.invalidate_shadow()
        store.g_clean = False           # the gradients were consumed; the next zero_grad clears them
        return loss

    def zero_grad(self, set_to_none: bool = False):
        """Gradients alias the flat buffer: one memset, never set to None."""
        if self._store is not None:
            self._store.zero_grad()
        else:
            super().zero_grad(set_to_none=False)

    def state_dict(self):
        for st in self.state.values():
            if "step" in st:
                st["step"].fill_(float(self._step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # re-home the loaded moments into the flat buffers
        loaded = {p: dict(st) for p, st in self.state.items()}
        steps = [int(st["step"]) for st in loaded.values() if "step" in st]
        self._step = max(steps) if steps else 0
        if self._hyper is not None:           # keep the tensor (a captured graph holds its address); rewrite its contents
            self._hyper[:, 3] = float(self._step)
            self._hyper_lr = [None] * len(self.param_groups)
        self.state.clear()
        self._segments = None
        if self._store is not None:
            self._build()
            for p, st in loaded.items():
                if "exp_avg" in st:
                    self.state[p]["exp_avg"].copy_(st["exp_avg"])
                    self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
