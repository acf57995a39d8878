"""Parameter containers that reproduce the reference's state_dict names/shapes, plus the glue that
binds them to the flat ParamStore.  They hold no compute of their own: forward code in
clip.py / layers.py / crog.py calls crog_amd.functional with WRefs taken from these containers.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from .. import functional as Fn
from ..functional import BnBuffers, WRef


class Bound(nn.Module):
    """Module whose parameters live in a ParamStore once `bind_all(root, store)` has run."""

    def _make_refs(self, store):
        pass


def bind_all(root: nn.Module, store):
    for m in root.modules():
        if isinstance(m, Bound):
            m._store = store
            m._make_refs(store)


class Conv2d(Bound):
    def __init__(self, cin, cout, k, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(cin * k * k)
            nn.init.uniform_(self.bias, -bound, bound)
        self.k = k

    def _make_refs(self, store):
        self.w = WRef(store, self.weight)
        self.b = WRef(store, self.bias) if self.bias is not None else None


class Linear(Bound):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(cin)
            nn.init.uniform_(self.bias, -bound, bound)

    def _make_refs(self, store):
        self.w = WRef(store, self.weight)
        self.b = WRef(store, self.bias) if self.bias is not None else None


class BatchNorm(Bound):
    """nn.BatchNorm1d/2d state (eps 1e-5, momentum 0.1, affine, running stats)."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps, self.momentum = 1e-5, 0.1

    def _make_refs(self, store):
        self.g = WRef(store, self.weight, cols=1)
        self.b = WRef(store, self.bias, cols=1)

    def buffers_ref(self) -> BnBuffers:
        return BnBuffers(self.g, self.b, self.running_mean, self.running_var, self.momentum, self.eps)


class LayerNorm(Bound):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.eps = 1e-5

    def _make_refs(self, store):
        self.g = WRef(store, self.weight, cols=1)
        self.b = WRef(store, self.bias, cols=1)

    def __call__(self, x, **kw):
        return Fn.layernorm(x, self.g, self.b, eps=self.eps, **kw)


class MultiheadAttention(Bound):
    """nn.MultiheadAttention parameter layout: packed in_proj_weight [3E, E], in_proj_bias [3E], out_proj.{weight,bias}."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def _make_refs(self, store):
        E = self.embed_dim
        self.wq, self.wk, self.wv = (WRef(store, self.in_proj_weight, i * E, E) for i in range(3))
        self.bq, self.bk, self.bv = (WRef(store, self.in_proj_bias, i * E, E, cols=1) for i in range(3))

    def project_kv(self, xk, xv):
        """The key / value projections alone (for a caller that computes them on another stream): -> what `kv=` of __call__ takes."""
        return Fn.kv_proj(xk, xv, self.wk, self.wv, self.bk, self.bv)

    def __call__(self, xq, xk, xv, *, B, causal=False, kpm=None, res=None, training=True, res_slot=None, kv=None):
        p = self.dropout if training else 0.0
        return Fn.mha(xq, xk, xv, self.wq, self.wk, self.wv, self.bq, self.bk, self.bv, self.out_proj.w, self.out_proj.b, B=B,
                      heads=self.num_heads, causal=causal, kpm=kpm, p_drop=p, res=res, res_slot=res_slot, kv=kv)


class ConvBN(Bound):
    """(conv | linear) -> BatchNorm [-> ReLU]: reference `conv_layer` / `linear_layer` Sequentials and the
    conv+bn pairs of Bottleneck.  Children are registered under the reference's names by the owner."""

    def __init__(self, conv: Optional[nn.Module], bn: BatchNorm, names=("0", "1")):
        super().__init__()
        if conv is not None:
            self.add_module(names[0], conv)
        self.add_module(names[1], bn)
        self._names = names
        self._has_conv = conv is not None

    @property
    def conv(self):
        return getattr(self, self._names[0]) if self._has_conv else None

    @property
    def bn(self):
        return getattr(self, self._names[1])

    def run(self, x, *, ksize=None, relu=True, res=None, out=None, wpad=None, dtype=None, stat_out=None, stat_in=None, dx_slot=None):
        conv = self.conv
        if ksize is None:
            ksize = 0 if conv is None else (conv.k if isinstance(conv, Conv2d) else 1)
        o = Fn.OutRef(out) if out is not None else None
        return Fn.conv_bn_act(x, conv.w if conv is not None else None, self.bn.buffers_ref(), ksize=ksize, relu=relu, res=res,
                              training=self.bn.training, out=o, wpad=wpad, dtype=dtype, stat_out=stat_out, stat_in=stat_in, dx_slot=dx_slot)


def conv_layer(cin, cout, k=1):
    """layers.py:8-11 (padding k//2, stride 1, no bias)."""
    return ConvBN(Conv2d(cin, cout, k), BatchNorm(cout))


def linear_layer(cin, cout):
    """layers.py:14-16 (bias-free Linear + BatchNorm1d + ReLU)."""
    return ConvBN(Linear(cin, cout, bias=False), BatchNorm(cout))
