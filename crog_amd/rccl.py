"""A direct RCCL communicator for the SyncBatchNorm statistics exchange (`crog_comm_init` / `crog_syncbn_stats`).

Why not `torch.distributed.all_reduce` for this one: ProcessGroupNCCL runs every collective on a stream of its own and fences it
with two event hops (compute stream -> communicator stream -> compute stream) plus ~25 us of host work.  The CROG step issues 142
statistics exchanges of 2·C floats each, ALL on the critical path (BatchNorm cannot apply before the global sums exist): measured at
world size 1, where the collective itself is a no-op, that machinery alone costs 2.2 ms of a 37 ms step.  Here the exchange is one
`ncclAllReduce` enqueued IN ORDER on the stream the BatchNorm kernels run on — no events, one ctypes call.

STATUS: opt-in (`CROG_SYNCBN_DIRECT=1`).  Measured at world size 1 (the only size this build has hardware for): the direct call costs
1.2 us of host time and nothing on the GPU (an in-place single-rank all-reduce is a no-op) where torch's costs 7.7 us + a 9.5 us stream
round trip (`scripts/probe_allreduce_host.py`), the BatchNorm-backward partial -> apply gap shrinks from 12.8 us to 0
(`scripts/_ddp_gaps.sh`), and the forced-DDP step is 35.1-35.6 ms against 35.8-36.2 ms with torch's process groups.  Opt-in only
because it has never run with real peers (a set-up problem falls back to torch; a hang would not).

RCCL is the library PyTorch-ROCm already has resident (`torch/lib/librccl.so`); the unique id is created on rank 0 and handed to the
other ranks through the existing `torch.distributed` group, which is also what the gradient buckets keep using (they are large,
asynchronous and belong on a side stream).  If anything in the set-up fails ON ANY RANK, every rank learns it (RcclComm.create) and
`crog_amd.parallel` falls back to a torch process group on all ranks alike.
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.distributed as dist

from . import kernels as K

NCCL_FLOAT32, NCCL_SUM = 7, 0          # ncclDataType_t / ncclRedOp_t (nccl.h; RCCL keeps NCCL's values)


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]


_lib = None


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        lib = ctypes.CDLL(path)
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        for fn in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllReduce, lib.ncclCommDestroy):
            fn.restype = ctypes.c_int
        _lib = lib
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"RCCL {what} failed: {_load().ncclGetErrorString(rc).decode()} ({rc})")


def _all_agree(ok: bool, group, dev) -> bool:
    """True only when EVERY rank of the group reports ok (a MIN all-reduce over the torch group that carries the set-up)."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend(group) == "nccl":
        flag = flag.to(dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


class RcclComm:
    """ncclComm over the ranks of a torch.distributed group (default: all ranks).  Construction is a collective call; use
    `RcclComm.create`, whose outcome is the SAME on every rank: a communicator everywhere, or None everywhere."""

    @classmethod
    def create(cls, group=None, device=None):
        """Set-up with a collective verdict.  Every step that can fail on a subset of the ranks (loading librccl, ncclGetUniqueId
        on rank 0, ncclCommInitRank) is followed by a MIN all-reduce of an ok flag over the torch group, and no rank enters the next
        collective step unless all ranks passed the previous one - so a one-sided failure can neither leave some ranks waiting in a
        broadcast / ncclCommInitRank the others never enter, nor make the ranks disagree on which communicator BatchNorm uses."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self = cls.__new__(cls)
        self.rank, self.world_size = dist.get_rank(group), dist.get_world_size(group)
        self._comm = ctypes.c_void_p()
        self._lib = None
        err = None
        uid = _UniqueId()
        try:
            self._lib = _load()
            if self.rank == 0:
                _check(self._lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        except Exception as e:
            err = e
        if not _all_agree(err is None, group, dev):
            return None, err or RuntimeError("direct RCCL set-up failed on another rank")
        # every rank has the library and rank 0 an id: hand the id out over the existing group (always entered by all ranks)
        wire = torch.tensor(list(bytes(uid)), dtype=torch.uint8)
        if dist.get_backend(group) == "nccl":
            wire = wire.to(dev)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(wire, src=src, group=group)
        ctypes.memmove(ctypes.byref(uid), bytes(wire.cpu().tolist()), 128)
        try:
            import contextlib
            with (torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()):
                _check(self._lib.ncclCommInitRank(ctypes.byref(self._comm), self.world_size, uid, self.rank), "ncclCommInitRank")
        except Exception as e:
            err = e
        if not _all_agree(err is None, group, dev):
            self.close()
            return None, err or RuntimeError("ncclCommInitRank failed on another rank")
        return self, None

    def __init__(self, group=None, device=None):
        comm, err = RcclComm.create(group, device)
        if comm is None:
            raise err
        self.__dict__.update(comm.__dict__)

    def all_reduce_sum(self, t: torch.Tensor):
        """In-place fp32 sum over the ranks, enqueued on the stream the caller's kernels run on."""
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise TypeError("RcclComm.all_reduce_sum expects a contiguous fp32 GPU tensor")
        _check(self._lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), NCCL_FLOAT32, NCCL_SUM, self._comm, K.stream()), "ncclAllReduce")

    def close(self):
        if self._comm and self._lib is not None:
            self._lib.ncclCommDestroy(self._comm)
            self._comm = ctypes.c_void_p()
