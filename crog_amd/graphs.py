"""HIP-graph capture of a fixed-shape, launch-bound sub-network (forward AND backward).

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
