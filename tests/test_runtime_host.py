"""Host-side bookkeeping of the runtime (no GPU): gradient-ready announcements of parameters that several launches write."""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_packed_parameter_is_announced_once_after_its_last_parked_block():
    """A packed in_proj_weight (clip.py:246, layers.py:291-296) gets its gradient from three row-block GEMMs.  Any block's announcement marks the
    WHOLE parameter ready for the DDP bucket / Adam chunk, so the announcement must wait for the last block - also when the blocks are parked in
    DIFFERENT grouped launches (the decoder's cross-attention: q over pixel rows, k / v over token rows).  Round 6: functional.done_joint."""
    from crog_amd.functional import WRef, done_joint
    from crog_amd.runtime import RT, ParamStore
    m = torch.nn.Module()
    m.w = torch.nn.Parameter(torch.zeros(12, 4))
    m.b = torch.nn.Parameter(torch.zeros(12))
    st = ParamStore(m, torch.device("cpu"))
    wq, wk, wv = (WRef(st, m.w, i * 4, 4) for i in range(3))
    bq, bk, bv = (WRef(st, m.b, i * 4, 4, cols=1) for i in range(3))
    calls = []
    RT.reducer = SimpleNamespace(mark_ready=lambda p: calls.append(p))
    old_groups = RT._groups
    try:
        # the k block's GEMM (and its bias sum riding along) are parked in a group of their own
        desc = SimpleNamespace(C=st.G.data_ptr() + 4 * wk.off, a_sum=st.G.data_ptr() + 4 * bk.off)
        RT._groups = {None: {640: dict(descs=[desc], keep=[], n=1, K=640, done=[])}}
        done_joint([wq, wk, wv, bq, bk, bv])
        assert calls == [], "announced while the k block was still parked"
        g = RT._groups[None].pop(640)
        assert len(g["done"]) == 2
        for d in g["done"]:      # what Runtime.flush_group does once the grouped launch is enqueued
            d()
        assert len(calls) == 2 and {id(p) for p in calls} == {id(m.w), id(m.b)}      # each parameter exactly once
        # nothing parked: announced at once, once per parameter
        calls.clear()
        RT._groups = {}
        done_joint([wq, wk, wv, bq, bk, bv])
        assert len(calls) == 2
        # a single reference (an unpacked parameter) goes the plain way
        calls.clear()
        done_joint([WRef(st, m.w)])
        assert calls == [m.w]
    finally:
        RT._groups = old_groups
        RT.reducer = None
