"""Fused Adam over the flat parameter store (torch.optim.Adam semantics: train_crog.py:119-121).

One kernel launch per contiguous learning-rate segment of the flat buffer (CROG: ~6 launches for 449
tensors / 147 M parameters) instead of a multi-tensor loop; moments live in two flat fp32 buffers whose
per-parameter views are exposed through `optimizer.state`, so `state_dict()` has torch.optim.Adam's layout
(`exp_avg`, `exp_avg_sq`, `step`) and round-trips with reference checkpoints (train_crog.py:213,259).
"""
from __future__ import annotations

from typing import List

import torch

from . import kernels as K
from .runtime import ALIGN, RT, ParamStore


class FusedAdam(torch.optim.Optimizer):
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
        if self.capturable:
            self.sync_lr()        # (creates the device state from the host step count: before that count moves on)
        self._step += 1
        store = self._store
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
