"""Data parallelism for the HIP path: one process per MI355X, RCCL (torch.distributed backend "nccl") over xGMI.

Re-expresses the reference's DistributedDataParallel + SyncBatchNorm (train_crog.py:113-114,154-156):
  * gradients: the flat gradient buffer is cut into contiguous buckets in reverse parameter order (the order
    in which backward finishes them); a bucket's all-reduce is issued asynchronously the moment its last
    parameter gradient has been written, so communication overlaps the rest of backward.  Buckets are slices
    of the flat buffer — no packing copies.  xGMI is point-to-point (7 links x ~153 GB/s per GPU), so buckets
    are large (64 MiB default) to keep each collective bandwidth-bound rather than latency-bound.
  * BatchNorm statistics: per-layer (sum, sum^2) / (sum g, sum g*xhat) pairs all-reduced on the compute stream
    (crog_amd.functional.ConvBnAct), i.e. SyncBatchNorm semantics with equal per-rank batch sizes.
  * the reference's find_unused_parameters round trip (only `logit_scale` is unused) is not needed: unused
    parameters keep a zero gradient in the flat buffer and their bucket is flushed at the end of backward.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from .runtime import ALIGN, RT


import os as _os
_DRY = False   # diagnostics (set by scripts): run the reducer's bookkeeping without issuing collectives


def _direct_mode(direct):
    """direct / CROG_SYNCBN_DIRECT -> (rccl, peer): which transports of the C-ABI communicator (crog_amd/rccl.py) to build.
    "peer": the one-shot hipIpc mailbox exchange only (any torch backend carries the set-up: two test ranks on one GPU use gloo);
    "rccl" / "1" / True: ncclAllReduce on the compute stream; "both": mailbox for what fits a slot, RCCL for the rest."""
    if direct is None:
        direct = _os.environ.get("CROG_SYNCBN_DIRECT", "auto")
    if direct in (False, "0", "", "off", "torch"):
        return False, False
    if direct == "auto":      # build both and let the start-up self-test decide (DirectComm.create(selftest=True))
        return True, True
    if direct == "peer":
        return False, True
    if direct == "both":
        return True, True
    return True, False


class SyncBNComm:
    """Communicator of the cross-replica BatchNorm statistics (train_crog.py:113-114).  The exchanges are on the critical path (142
    per CROG step, 2·C floats each), so they avoid both the gradient buckets' communicator — collectives of one communicator execute in
    issue order, and a statistics exchange would queue behind whatever 64 MiB bucket is in flight — and, with `direct`, torch.distributed's
    stream/event fencing: `direct` = the C-ABI communicator (include/crog_hip.h crog_comm_*, crog_amd/rccl.py) whose exchange is
    enqueued on the compute stream itself, either as ONE single-block kernel per rank (peer writes into hipIpc mailboxes: one hop
    instead of a ring) or as an ncclAllReduce.  Chosen by the job itself since round 4 (CROG_SYNCBN_DIRECT unset = "auto": both transports
    are built and self-tested at start-up with a collective verdict, rccl.DirectComm.create(selftest=True); "torch" / "0" forces the
    process group, peer | rccl | both force a transport).  Measured at world size
    1, the only size a box of this pool has: the direct RCCL calls cost 1.2 us of host time and no GPU work per call against 7.7 us + a
    9.5 us stream round trip for torch's, and the forced-DDP step is 35.1-35.6 ms with them against 35.8-36.2 ms on torch's groups.  The
    mailbox exchange is validated with two processes sharing one GPU (tests/test_ddp2_gpu.py); neither has run across xGMI.
    Fallback (self-test failed anywhere, or a gloo group): a torch process group of its own."""

    def __init__(self, group=None, direct=None):
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.force = False
        self.calls = 0          # collectives issued (bench.py reports the per-step count)
        self.direct = None
        self.kind = "torch.distributed"
        rccl, peer = _direct_mode(direct)
        rccl = rccl and dist.get_backend(group) == "nccl"
        if (rccl or peer) and torch.cuda.is_available():
            # DirectComm.create is collective and its verdict is the same on every rank (a MIN all-reduce after each step that can fail
            # one-sidedly): either all ranks get the direct communicator or all of them fall back, so the ranks can never disagree
            # about the collectives that follow (the dedicated torch group below is created by all of them or by none)
            from .rccl import DirectComm
            self.direct, err = DirectComm.create(group, rccl=rccl, peer=peer, selftest=True)
            if self.direct is not None:
                self.kind = "crog_comm:" + "+".join(k for k, on in (("rccl", self.direct.has_rccl), ("peer", self.direct.has_peer)) if on)
            else:
                import warnings
                warnings.warn(f"crog_amd: direct communicator unavailable ({err!r}); SyncBatchNorm uses torch.distributed on every rank")

    def fuse_ptr(self, n: int):
        """The kernel-tail form of a BACKWARD statistics exchange of n floats (round 5: the kernel that produces the sums - the first
        BatchNorm-backward pass or a data-gradient GEMM's bwd_z epilogue - exchanges them in its last block, comm_dev.h): the device
        block to hand to that kernel, or None when the exchange has to be a launch of its own (no mailboxes, too large for a slot,
        CROG_SYNCBN_FUSE=0).  Counted in `fused` (bench.py reports both counts)."""
        if self.direct is None or not self.direct.has_peer or n > self._slot() or _os.environ.get("CROG_SYNCBN_FUSE", "1") == "0":
            return None
        if not getattr(self.direct, "tail_ok", False):
            return None      # the tail form has not passed its start-up self-test on every rank (failed, or a communicator built without one): launches of their own
        self.fused = getattr(self, "fused", 0) + 1
        return self.direct.sync_block()

    def all_reduce_sum(self, t: torch.Tensor):
        self.calls += 1
        if self.direct is not None and t.is_cuda and (self.direct.has_rccl or t.numel() <= self._slot()):
            self.direct.all_reduce_sum(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def check(self):
        """Raise if an exchange of the direct communicator ever gave up waiting for a peer (the kernel then returns NaN statistics on
        that and every later exchange, csrc/comm.hip): called where the host already synchronises - engine.train_with_grasp's print
        window, the end of bench.py - so that a run cannot continue on diverged BatchNorm statistics.  Synchronises the device."""
        if self.direct is not None:
            seq = self.direct.timed_out()
            if seq:
                raise RuntimeError(f"crog_amd: SyncBatchNorm exchange #{seq} timed out waiting for a peer rank (CROG_COMM_TIMEOUT_S); "
                                   "the communicator is dead and the BatchNorm statistics since then are NaN - restart from the last checkpoint")

    @staticmethod
    def _slot():
        from .rccl import SLOT_FLOATS
        return SLOT_FLOATS


def convert_sync_batchnorm(model, process_group=None, force=False, dedicated_group=True, direct=None):
    """nn.SyncBatchNorm.convert_sync_batchnorm equivalent: BatchNorm statistics of the HIP path become cross-replica.
    `force` installs the communicator even at world_size 1 (single-GPU smoke test of the collective path).  A collective call:
    every rank converts its model (as with the reference's conversion), because the statistics get a communicator of their own —
    the C-ABI one when asked for (`direct`, CROG_SYNCBN_DIRECT), otherwise a dedicated torch process group (world size > 1; measured
    at world size 1: 36.0 vs 35.8-36.2 ms on the default group, i.e. free once the stream creation order is fixed).  Collectives of one communicator run in issue order, so on the default group every statistics exchange
    issued while a 64 MiB gradient bucket is in flight (nine per step, ~0.4 ms each at 8 GPUs) would wait for it on the critical path."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(process_group) > 1 or force):
        RT.ensure_streams()       # the side streams take their hardware queues before the communicator's streams exist
        rccl, peer = _direct_mode(direct)
        use_direct = ((rccl and dist.get_backend(process_group) == "nccl") or peer) and torch.cuda.is_available()
        group = process_group
        need_group = dedicated_group and dist.get_world_size() > 1
        if group is None and need_group and not (use_direct and rccl):
            # (a peer-only communicator still needs a torch group for exchanges larger than a mailbox slot)
            group = dist.new_group()
        RT.comm = SyncBNComm(group, direct=direct if use_direct else False)
        if RT.comm.direct is None and use_direct and rccl and group is None and need_group:
            RT.comm.group = dist.new_group()     # the direct set-up failed, on every rank alike (DirectComm.create): a dedicated torch group
        RT.comm.force = force
    return model


def all_reduce_mean_(t: torch.Tensor, net=None):
    """In-place rank mean of a small fp32 tensor (the step's (loss, IoU, Prec@50) triple, crog_engine.py:88-93) on the current stream:
    through the gradient-bucket communicator of `net` (crog_amd DistributedDataParallel) when it has one - a plain RCCL call that a
    captured step can hold - otherwise torch.distributed."""
    world = dist.get_world_size()
    comm = getattr(net, "bucket_comm", None)
    if comm is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
        comm.all_reduce_bucket(t, average=True)
        return t
    dist.all_reduce(t)
    t.div_(world)
    return t


def step_is_capturable(net) -> bool:
    """True when every collective of a training step goes through the C-ABI communicators (BatchNorm statistics: RT.comm.direct with
    RCCL behind it; gradient buckets and the metric: net.bucket_comm): no torch ProcessGroup work inside the step, so nothing of it
    records events a capture must not see (the watchdog thread polling a captured event aborted one run in a few in round 3)."""
    if getattr(net, "bucket_comm", None) is None:
        return False
    if RT.comm is None:
        # plain per-rank BatchNorm (sync_bn: False): DistributedDataParallel._broadcast_buffers then runs a torch.distributed broadcast
        # inside every training forward - ProcessGroup work a capture must not hold (the configuration that aborted in round 3)
        return not (getattr(net, "broadcast_buffers", False) and dist.is_initialized() and dist.get_world_size(getattr(net, "process_group", None)) > 1)
    return RT.comm.direct is not None and RT.comm.direct.has_rccl


class Reducer:
    """Bucketed, overlapped mean all-reduce of a flat gradient buffer."""

    def __init__(self, flat_grad: torch.Tensor, entries, group=None, bucket_cap_mb: float = 64.0, payload_dtype=None, direct=None):
        """entries: iterable of (param, offset, numel) in registration order.
        payload_dtype: None = exchange the fp32 gradients as they are (the reference's DDP, bit-compatible); torch.bfloat16 = each
        bucket is rounded to bf16 for the wire (294 instead of 588 MB per step for CROG-R50), averaged, and widened back — an opt-in
        that changes the numerics (`DistributedDataParallel(..., gradient_payload=torch.bfloat16)` or CROG_GRAD_PAYLOAD=bf16)."""
        self.G = flat_grad
        self.direct = direct      # rccl.DirectComm of the gradient buckets (crog_allreduce_bucket: one ncclAllReduce on the carrier stream,
                                  # no ProcessGroup hop, capturable), or None: torch.distributed
        self.payload_dtype = payload_dtype
        self._wire = {}           # bucket index -> bf16 staging buffer
        self.group = group
        self.world = dist.get_world_size(group)
        cap = int(bucket_cap_mb * (1 << 20) / 4)
        ents = sorted(((o, (n + ALIGN - 1) // ALIGN * ALIGN, p) for p, o, n in entries), key=lambda e: e[0])
        self.buckets: List[dict] = []
        cur = None
        for o, n, p in reversed(ents):  # backward produces gradients from the end of the buffer
            if cur is None or cur["numel"] + n > cap or o + n != cur["start"]:
                cur = dict(start=o + n, end=o + n, numel=0, params=[], pending=0, work=None, launched=False)
                self.buckets.append(cur)
            cur["start"] = o
            cur["numel"] += n
            cur["params"].append(p)
        self.bucket_of = {}
        for i, b in enumerate(self.buckets):
            for p in b["params"]:
                self.bucket_of[id(p)] = i
        self._armed = False
        self._use_avg = dist.get_backend(group) == "nccl"
        self.launches = 0       # bucket all-reduces issued (bench.py reports the per-step count)
        self.after_launch = None   # callable(bucket, carrier stream): FusedAdam's per-bucket update rides the carrier (optim.py: _arm_buckets)
        self.reset()

    def reset(self):
        for b in self.buckets:
            b["pending"] = len(b["params"])
            b["work"] = None
            b["launched"] = False
            b["seen"] = set()
        self._armed = False
        self._done = True

    def _launch(self, b):
        if b["launched"]:
            return
        b["launched"] = True
        self.launches += 1
        # a bucket can hold gradients written on different streams (image tower: main, text tower: side stream):
        # make the launching stream wait for the others before RCCL's stream takes its dependency on it
        view = self.G[b["start"]:b["start"] + b["numel"]]
        op = dist.ReduceOp.AVG if self._use_avg else dist.ReduceOp.SUM
        if _DRY:
            b["work"] = None
            return
        b["wire"] = None
        if self.payload_dtype is not None and self.payload_dtype != view.dtype:
            i = self.buckets.index(b)
            if i not in self._wire:
                self._wire[i] = torch.empty(b["numel"], device=view.device, dtype=self.payload_dtype)
            b["wire"] = self._wire[i]
        if self.G.is_cuda and self.direct is not None:
            # the C-ABI path: the collective is enqueued on the carrier stream itself (after it has waited for every writer), in
            # bucket order - the same order on every rank, as RCCL requires of one communicator
            from . import kernels as K
            cur = torch.cuda.current_stream()
            side = [s for s in RT.streams if s != cur]
            carrier = side[-1] if side else cur
            if side:
                carrier.wait_stream(cur)
                for s in side[:-1]:
                    carrier.wait_stream(s)
            with torch.cuda.stream(carrier):
                t = self._to_wire(b, view)
                self.direct.all_reduce_bucket(t, average=True)
                if b["wire"] is not None:
                    view.copy_(b["wire"])          # widen the averaged payload back into the fp32 gradient buffer
            b["carrier"], b["work"], b["direct"] = carrier, None, True
            if self.after_launch is not None:
                self.after_launch(b, carrier)
            return
        b["direct"] = False
        if self.G.is_cuda:
            cur = torch.cuda.current_stream()
            side = [s for s in RT.streams if s != cur]
            if side:
                # Issue the collective from a side stream that has waited for every other writer: RCCL's stream then depends
                # on all of them, while the main stream (the backward critical path) is NOT stalled behind the weight-gradient
                # stream at every bucket boundary.
                carrier = side[-1]
                carrier.wait_stream(cur)
                for s in side[:-1]:
                    carrier.wait_stream(s)
                with torch.cuda.stream(carrier):
                    b["work"] = dist.all_reduce(self._to_wire(b, view), op=op, group=self.group, async_op=True)
                b["carrier"] = carrier
                return
        b["carrier"] = None
        b["work"] = dist.all_reduce(self._to_wire(b, view), op=op, group=self.group, async_op=True)

    @staticmethod
    def _to_wire(b, view):
        """The tensor that goes on the wire: the gradient slice itself, or its rounded copy (made on the launching stream)."""
        if b["wire"] is None:
            return view
        b["wire"].copy_(view)
        return b["wire"]

    def mark_ready(self, param):
        i = self.bucket_of.get(id(param))
        if i is None:
            return
        if not self._armed:
            self._armed = True
            self._done = False
            torch.autograd.Variable._execution_engine.queue_callback(self.wait)
        b = self.buckets[i]
        if id(param) in b["seen"]:
            return
        b["seen"].add(id(param))
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def wait(self):
        """Flush buckets holding unused parameters, wait for all collectives, finish the mean."""
        if self._done:
            return
        # Everything the runtime still holds back goes FIRST: this callback is queued by the first gradient announcement of the
        # pass (the head's, before any weight gradient has armed Runtime._end_of_backward), so it runs ahead of the runtime's own
        # end-of-backward flush.  A weight gradient still parked for a grouped launch (or deferred behind a data gradient) would be
        # all-reduced before its GEMM is enqueued, and its late announcement would hit the reset reducer and start a second round of
        # all-reduces.  The flush enqueues those launches and lets their announcements through (which may launch buckets from here).
        RT.flush_wgrad()
        RT.flush_group()
        for b in self.buckets:
            self._launch(b)
        for b in self.buckets:
            if b.get("direct"):
                if b["carrier"] is not None and b["carrier"] != torch.cuda.current_stream():
                    torch.cuda.current_stream().wait_stream(b["carrier"])
                continue                        # averaged (and widened) on the carrier stream already
            if b["work"] is not None:
                b["work"].wait()
            view = self.G[b["start"]:b["start"] + b["numel"]]
            if b.get("wire") is not None and b["work"] is not None:
                view.copy_(b["wire"])          # widen the averaged payload back into the fp32 gradient buffer
            if not self._use_avg:
                view.div_(self.world)
        self.reset()


class DistributedDataParallel(torch.nn.Module):
    """Drop-in for torch.nn.parallel.DistributedDataParallel(model.cuda(), device_ids=[gpu], find_unused_parameters=True)
    as used at train_crog.py:154-156, for crog_amd models (or any module whose parameters live in a ParamStore)."""

    def __init__(self, module, device_ids=None, find_unused_parameters=False, process_group=None, bucket_cap_mb: float = 64.0,
                 broadcast_buffers: bool = True, force: bool = False, gradient_payload=None):
        super().__init__()
        self.module = module
        if gradient_payload is None and _os.environ.get("CROG_GRAD_PAYLOAD", "").lower() in ("bf16", "bfloat16"):
            gradient_payload = torch.bfloat16
        self.gradient_payload = gradient_payload
        self.force = force
        self.process_group = process_group
        self.broadcast_buffers = broadcast_buffers
        self.bucket_cap_mb = bucket_cap_mb
        self.reducer: Optional[Reducer] = None
        self.bucket_comm = None      # rccl.DirectComm for the gradient buckets (None: torch.distributed)
        self._hooks = []
        if hasattr(module, "prepare"):
            dev = torch.device("cuda", device_ids[0]) if device_ids else next(module.parameters()).device
            if dev.type == "cuda":
                RT.ensure_streams(dev)
            if hasattr(module, "_ensure"):
                module._ensure(dev)       # keep a flat store that is already in place: an optimizer may be attached to it
            else:
                module.prepare(dev)
        self._sync_initial_state()
        self._make_bucket_comm()

    def _make_bucket_comm(self):
        """The gradient buckets get an RCCL communicator of their own behind the C ABI (crog_allreduce_bucket): collectives of one
        communicator run in issue order, so it is not the BatchNorm statistics' communicator.  Collective set-up with a collective
        verdict and a self-test (rccl.DirectComm.create); CROG_DDP_DIRECT=0 keeps torch.distributed.  A gloo group (CPU tests) has none."""
        if _os.environ.get("CROG_DDP_DIRECT", "1") == "0" or not torch.cuda.is_available():
            return
        if dist.get_world_size(self.process_group) == 1 and not self.force:
            return
        if dist.get_backend(self.process_group) != "nccl":
            return
        from .rccl import DirectComm
        RT.ensure_streams()
        self.bucket_comm, err = DirectComm.create(self.process_group, rccl=True, peer=False, selftest=True)
        if self.bucket_comm is None:
            import warnings
            warnings.warn(f"crog_amd: direct gradient-bucket communicator unavailable ({err!r}); buckets use torch.distributed on every rank")
        else:
            # one ncclAllReduce per bucket, or reduce-scatter + all-gather: timed on this node at start-up, the same choice on every rank
            self.bucket_algo = self.bucket_comm.tune_bucket_algo(torch.device("cuda", torch.cuda.current_device()))

    def _sync_initial_state(self):
        """DDP broadcasts rank 0's parameters and buffers at construction."""
        if dist.get_world_size(self.process_group) == 1 and not self.force:
            return
        store = getattr(self.module, "store", None)
        if store is not None:
            dist.broadcast(store.P, 0, group=self.process_group)
        else:
            for p in self.module.parameters():
                dist.broadcast(p.data, 0, group=self.process_group)
        for b in self.module.buffers():
            dist.broadcast(b, 0, group=self.process_group)

    def _broadcast_buffers(self):
        """torch DDP's broadcast_buffers=True (the reference's default, train_crog.py:154-156): rank 0's buffers - BatchNorm running
        statistics - replace every rank's before each training forward.  Under SyncBatchNorm (RT.comm installed, the reference's
        configuration) all ranks compute the same running statistics from the same global sums and the broadcast would change nothing,
        so it is skipped; with plain per-rank BatchNorm (sync_bn: False) the ranks' statistics drift apart and this keeps them rank
        0's, as torch does.  One coalesced broadcast per dtype."""
        if not self.broadcast_buffers or RT.comm is not None:
            return
        by_dtype = {}
        for b in self.module.buffers():
            if b.numel():
                by_dtype.setdefault(b.dtype, []).append(b)
        for bufs in by_dtype.values():
            flat = torch.cat([b.reshape(-1) for b in bufs])
            dist.broadcast(flat, 0, group=self.process_group)
            off = 0
            for b in bufs:
                b.copy_(flat[off:off + b.numel()].view_as(b))
                off += b.numel()

    def _ensure_reducer(self):
        store = self.module.store
        if self.reducer is None or self.reducer.G is not store.G:
            self.reducer = Reducer(store.G, [(p, o, n) for _, p, o, n, _ in store.entries], self.process_group, self.bucket_cap_mb,
                                   payload_dtype=self.gradient_payload, direct=self.bucket_comm)
            for h in self._hooks:
                h.remove()
            self._hooks = []
            if not getattr(self.module, "explicit_grad_ready", False):
                for _, p, _, _, _ in store.entries:
                    self._hooks.append(p.register_post_accumulate_grad_hook(lambda p_, r=self.reducer: r.mark_ready(p_)))

    def forward(self, *args, **kwargs):
        out_is_training = self.module.training and torch.is_grad_enabled()
        if hasattr(self.module, "_ensure") and args:
            self.module._ensure(args[0].device)
        if out_is_training and (dist.get_world_size(self.process_group) > 1 or self.force):
            self._broadcast_buffers()
            self._ensure_reducer()
            self.reducer.reset()
            RT.reducer = self.reducer
        else:
            RT.reducer = None
        return self.module(*args, **kwargs)

    def state_dict(self, *a, **kw):
        sd = self.module.state_dict(*a, **kw)
        return type(sd)(("module." + k, v) for k, v in sd.items())

    def load_state_dict(self, sd, strict=True):
        return self.module.load_state_dict({k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}, strict)
