"""Process-wide runtime state for the HIP path: flat parameter/gradient storage, the bf16 shadow
copy of the weights, the SyncBatchNorm communicator hook and the dropout seed stream.

Memory plan (per GPU, 288 GB HBM3E): all 147 M parameters live in ONE fp32 buffer `P`, their
gradients in ONE fp32 buffer `G` (so DDP buckets are plain slices of `G`, Adam is one launch per
learning-rate segment, and zeroing the gradients is a single memset), and the bf16 compute copy in
ONE buffer `S` refreshed by a single cast kernel.  3x3 convolution weights are stored physically
as [Cout][ky][kx][Cin] (torch channels_last) so the implicit-GEMM kernels read them in place;
the nn.Parameter keeps the reference's logical [Cout,Cin,3,3] shape, so state_dicts round-trip.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import kernels as K

ALIGN = 8  # elements; keeps every parameter 16-byte aligned in the bf16 shadow too


class Runtime:
    def __init__(self):
        self.comm = None       # object with .world_size and .all_reduce_sum(tensor) for SyncBatchNorm
        self.reducer = None    # gradient reducer (crog_amd.parallel); .mark_ready(param)
        self.streams = []      # HIP streams the model's kernels run on (main + text-tower side stream)
        self.seed_base = 0x5EED
        self._seed_ctr = 0

    def next_seed(self) -> int:
        self._seed_ctr += 1
        return ((self.seed_base & 0xFFFFFFFF) << 32) | (self._seed_ctr & 0xFFFFFFFF)

    def manual_seed(self, seed: int):
        self.seed_base = int(seed)
        self._seed_ctr = 0


RT = Runtime()


class ParamStore:
    """Flat storage for a module's parameters."""

    def __init__(self, module: torch.nn.Module, device: torch.device):
        self.device = device
        self.entries: List = []  # (name, param, offset, numel, grad_view)
        off = 0
        plan = []
        for name, p in module.named_parameters():
            plan.append((name, p, off))
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.P = torch.zeros(off, device=device, dtype=torch.float32)
        self.G = torch.zeros(off, device=device, dtype=torch.float32)
        self.S: Optional[torch.Tensor] = None
        self.shadow_fresh = False
        self.by_param: Dict[int, int] = {}
        for name, p, o in plan:
            n = p.numel()
            if p.dim() == 4 and p.shape[2] * p.shape[3] > 1:
                co, ci, kh, kw = p.shape
                v = self.P[o:o + n].view(co, kh, kw, ci).permute(0, 3, 1, 2)
                g = self.G[o:o + n].view(co, kh, kw, ci).permute(0, 3, 1, 2)
            else:
                v = self.P[o:o + n].view(p.shape)
                g = self.G[o:o + n].view(p.shape)
            with torch.no_grad():
                v.copy_(p.data.to(device=device, dtype=torch.float32))
            p.data = v
            p.grad = g
            self.entries.append((name, p, o, n, g))
            self.by_param[id(p)] = o

    # -- lookups ---------------------------------------------------------------------------------
    def off(self, p) -> int:
        return self.by_param[id(p)]

    def weights(self, dtype: torch.dtype) -> torch.Tensor:
        """Flat buffer holding the weights in the compute dtype."""
        if dtype == torch.float32:
            return self.P
        if self.S is None:
            self.S = torch.empty(self.total, device=self.device, dtype=torch.bfloat16)
            self.shadow_fresh = False
        if not self.shadow_fresh:
            K.cast_f32_to_bf16(self.P, self.S, self.total)
            self.shadow_fresh = True
        return self.S

    def valid(self) -> bool:
        name, p, o, n, g = self.entries[0]
        return p.data_ptr() == self.P.data_ptr() + 4 * o

    def relink_grads(self):
        """Make sure every p.grad is (still) the view into G (optimizer.zero_grad(set_to_none=True) drops it)."""
        for name, p, o, n, g in self.entries:
            if p.grad is not g:
                p.grad = g

    def zero_grad(self):
        self.G.zero_()

    def invalidate_shadow(self):
        self.shadow_fresh = False
