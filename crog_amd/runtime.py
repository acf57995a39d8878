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

import os
from typing import Dict, List, Optional

import torch

from . import kernels as K

ALIGN = 8  # elements; keeps every parameter 16-byte aligned in the bf16 shadow too


_DBG_GROUP = os.environ.get("CROG_DBG_GROUP") == "1"      # print every grouped weight-gradient launch (what was parked together)


# Under DistributedDataParallel, park image-tower weight gradients for grouped launches too (until the end of round 5 only the text tower's: a
# parked gradient announces itself - and starts its bucket's all-reduce - when its group goes out).  With the stale-group rule (note_wgrad) that
# is at most a few layers late, and the last bucket's layers (stem, layer1) are never parked: forced DDP 30.6 -> 30.0-30.5 ms.  CROG_DDP_PARK_ALL=0
# restores the old rule.
DDP_PARK_ALL = os.environ.get("CROG_DDP_PARK_ALL", "1") == "1"


def _xcd_masked_stream(xcds, dev):
    """A/B aid (CROG_WGRAD_XCDS): a stream whose kernels may only use the CUs of the given XCDs (hipExtStreamCreateWithCUMask;
    the driver deals mask bit i to XCD i mod 8, CU i div 8 of that XCD), wrapped for torch.  Round 3 measured low-N-bit masks (N / 8 CUs of
    EVERY XCD) as the weight-gradient stream: 54-64 ms against 33.4 (LAB_NOTES section 4); this is the XCD-granular form of the same question."""
    import ctypes
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    bits = 0
    for i in range(256):
        if (i % 8) in xcds:
            bits |= 1 << i
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(8), words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return torch.cuda.ExternalStream(h.value, device=dev)


def _low_priority_stream(dev):
    """A/B aid (CROG_WGRAD_PRIO=low): a stream of the LEAST priority the device offers (torch.cuda.Stream clamps positive priorities to normal),
    wrapped for torch: with short weight-gradient blocks the dispatcher would hand a freed CU to the main chain's waiting blocks first."""
    import ctypes
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    least, greatest = ctypes.c_int(), ctypes.c_int()
    if hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) != 0:
        raise RuntimeError("hipDeviceGetStreamPriorityRange failed")
    h = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint32(1), ctypes.c_int(least.value))      # 1 = hipStreamNonBlocking
    if rc != 0:
        raise RuntimeError(f"hipStreamCreateWithPriority({least.value}) failed ({rc})")
    if os.environ.get("CROG_DBG_GROUP") == "1":
        print(f"[streams] weight-gradient stream at priority {least.value} (range least {least.value} .. greatest {greatest.value})", flush=True)
    return torch.cuda.ExternalStream(h.value, device=dev)


class Runtime:
    def __init__(self):
        self.comm = None       # object with .world_size and .all_reduce_sum(tensor) for SyncBatchNorm
        self.reducer = None    # gradient reducer (crog_amd.parallel); .mark_ready(param)
        self.early_adam = None # FusedAdam with an armed overlapped update (optim.py: overlap_backward); .mark_ready(param)
        self.streams = []      # HIP streams the model's kernels run on (main + text-tower side stream + wgrad stream)
        self.overlap_wgrad = True     # weight gradients on a side stream (bench.py's CROG_SINGLE_STREAM profile mode switches it off)
        # A/B (round 3): weight gradients start after the layer's data gradient: "all" layers, or only the "big" ones whose data gradient
        # runs on the 256 x 256 tile (one block per CU: it needs CUs free of weight-gradient blocks)
        self.defer_wgrad = {"1": "all", "all": "all", "big": "big"}.get(os.environ.get("CROG_DEFER_WGRAD", "0"), "")
        self._pending_wgrad = []
        self._pending_done = []
        # small weight gradients parked for ONE grouped launch (functional.wgrad_gemm -> park_wgrad; kernels.gemm_group): descriptors,
        # the tensors they read (kept alive until the launch is enqueued), blocks so far, the stream they were ordered on.  Standalone
        # the grouped launch halves their time (scripts/ab_group.py: 12 x [512 x 512 x 21632] 572 -> 206 us); in the overlapped step it
        # is neutral (29.86 vs 29.93 ms, 3 x 80 steps each, box noise +-0.3): the weight-gradient stream has slack, the step follows
        # the main chain.  CROG_WGRAD_GROUP=0 switches it off.
        self.group_wgrad = os.environ.get("CROG_WGRAD_GROUP", "1") != "0"
        self.group_blocks = int(os.environ.get("CROG_WGRAD_GROUP_BLOCKS", "144"))
        # one parked group per side stream (round 5: the text tower's weight gradients go to the aux stream and are grouped there, beside the
        # image tower's on the weight-gradient stream): stream -> dict(descs, keep, n, K, done)
        self._groups = {}
        self._override = None        # the side stream a weight-gradient closure is being launched on (None: inline)
        self._in_wgrad = False
        self._slots = []       # GradSlots filled during the backward pass in flight (functional.GradSlot): all must be empty when it ends
        self._wgrad_stream = None
        self.text_stream = None
        # fourth stream (single-process jobs by default, CROG_AUX_STREAM): the text tower's weight gradients and the Adam chunks stepped during
        # backward.  On the weight-gradient stream both WAIT for the text chain (~200 latency-bound launches) and hold up the image tower's
        # weight gradients queued behind them.  Alone the stream changes nothing (30.00 vs 30.00 ms; 28.34 / 28.42 vs 28.34); together with a
        # high-priority text stream it is worth 0.1-0.6 ms (ensure_streams)
        self.aux_stream = None
        self._streams_ready = False
        self._join_armed = False
        # deterministic mode (set_deterministic below; CROG_DETERMINISTIC=1): every sum whose order would depend on atomics takes its
        # ordered form - same inputs, same bits, run after run, eager or replayed
        self.deterministic = False
        # which side streams deterministic mode keeps (set_deterministic): "all" (default since the packed-fp32 finding of round 5: every
        # stream of the default mode), "text" (only the text tower beside the image tower), "0" (one stream, rounds 3-4)
        self.det_streams = {"1": "all", "all": "all", "0": "0", "none": "0", "text": "text"}.get(os.environ.get("CROG_DET_STREAMS", "all"), "all")
        self.no_fork = set()   # probe only: on_wgrad_stream tags ("conv", "linear", "mha", "ln") whose launches stay on the caller's stream
        self.seed_base = 0x5EED
        self._seed_ctr = 0
        self.seed_epoch = None   # device int64 added to every dropout seed inside the kernels (enable_seed_epoch)

    # ---- weight-gradient side stream -----------------------------------------------------------------
    # Weight gradients are only consumed by the optimizer (or the gradient all-reduce), never by the rest of backward,
    # so their GEMMs run on a second stream and fill the CUs that the skinny data-gradient / BatchNorm kernels of the
    # dependency chain leave idle.
    # ---- stream creation order ---------------------------------------------------------------------------
    # HIP multiplexes its streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in the order the streams are first
    # used, and streams that land on the SAME hardware queue execute strictly in order.  Measured at world size 1 with
    # DistributedDataParallel + SyncBatchNorm forced on: when a communicator's internal stream is created between the main stream and
    # the weight-gradient stream, the latter wraps around onto the main stream's queue and the overlap is gone (+1.5 ms per step for
    # one extra communicator, +2.9 ms for two; 2 queues instead of 4 costs the plain step the same 1.5 ms; 8 queues make the
    # cross-queue waits of torch's collectives so slow that the step takes 54 ms).  So the side streams of the dependency chain are
    # created AND touched before anything else can create a stream: main -> queue 0, weight gradients -> 1, text tower -> 2.
    def ensure_streams(self, device=None):
        if not torch.cuda.is_available() or getattr(self, "_streams_ready", False):
            return
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type != "cuda":
            return
        with torch.cuda.device(dev):
            main = torch.cuda.current_stream()
            order = []
            if self.overlap_wgrad:
                # ONE weight-gradient stream at default priority.  Measured alternatives: two or three streams
                # round-robin (1 % slower: the small atomic-bound launches contend with each other), stream priorities either way (no
                # change), a CU-masked stream of 64 / 96 / 128 / 192 CUs (64.2 / 57.5 / 56.9 / 53.9 ms per step against 33.4: the weight
                # gradients need half of the chip-time of a step and cannot finish on a slice of it).
                nx = int(os.environ.get("CROG_WGRAD_XCDS", "0"))      # A/B only: the weight-gradient stream on the LAST nx XCDs
                if 0 < nx < 8:
                    self._wgrad_stream = [_xcd_masked_stream(set(range(8 - nx, 8)), dev)]
                elif os.environ.get("CROG_WGRAD_PRIO") == "low":      # A/B only
                    self._wgrad_stream = [_low_priority_stream(dev)]
                else:
                    self._wgrad_stream = [torch.cuda.Stream()]
                order += self._wgrad_stream
            import torch.distributed as dist
            multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            # Fourth stream (the text tower's weight gradients and the Adam chunks stepped during backward) + a HIGH-priority text stream:
            # together 28.33 / 28.56 / 29.05 / 28.89 -> 28.27 / 28.29 / 28.85 / 28.27 ms and 29.08 / 28.60 / 28.62 -> 28.47 / 28.50 / 28.51
            # (two boxes, alternating passes): the text tower's ~200 latency-bound backward launches no longer wait for free CUs behind the
            # image tower's kernels, and nothing queues behind them.  Each alone: the aux stream neutral (28.34 / 28.42 vs 28.34), the
            # priority a loss (28.76 / 28.97).  Default in a single-process job; with torch.distributed initialised (a communicator's internal
            # stream takes the fourth hardware queue under DDP: see the note above) both stay off unless the environment asks for them.
            solo = not (dist.is_available() and dist.is_initialized())
            want_aux = self.overlap_wgrad and not multi and os.environ.get("CROG_AUX_STREAM", "1" if solo else "0") == "1"
            text_prio = os.environ.get("CROG_TEXT_PRIO", "1" if want_aux else "0") == "1"
            self.text_stream = torch.cuda.Stream(device=dev, priority=-1 if text_prio else 0)
            order.append(self.text_stream)
            if want_aux:
                self.aux_stream = torch.cuda.Stream(device=dev)
                order.append(self.aux_stream)
            touch = torch.zeros(8, device=dev)
            for s in order:                     # first USE binds the stream to its hardware queue
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    touch.add_(1.0)
                main.wait_stream(s)
        self._streams_ready = True

    def wgrad_stream(self):
        """The weight-gradient side stream (one: every reduction into a parameter gradient is ordered on it).  None in deterministic
        mode: weight gradients then stay on the stream of the layer's backward (see set_deterministic)."""
        if not self.overlap_wgrad or not torch.cuda.is_available() or (self.deterministic and self.det_streams != "all"):
            return None
        if self._wgrad_stream is None:
            self.ensure_streams()
        if self._wgrad_stream is None:
            self._wgrad_stream = [torch.cuda.Stream()]
        s = self._wgrad_stream[0]
        if s not in self.streams:
            self.streams.append(s)
        return s

    def on_wgrad_stream(self, fn, *tensors, defer=None, tag=None):
        """Run fn() (kernel launches only) on the weight-gradient stream, after everything enqueued so far on the current
        stream; `tensors` are the operands it reads (kept alive for that stream).
        With `defer_wgrad` the fork is taken one launch LATER: the request is parked and issued by the next `flush_wgrad()` - which
        ConvBnAct.backward calls right after it has enqueued the layer's data-gradient GEMM - so a weight gradient starts when that
        data gradient has finished instead of next to it (both are MFMA-bound: side by side the one on the critical path takes up
        to 2.5x its own time)."""
        if tag is not None and tag in self.no_fork:
            fn()
            return
        if tag == "conv" and getattr(self, "fork_range", None) is not None:      # probe only (scripts/det_stress.py): fork the i-th conv weight gradient of the pass only for lo <= i < hi
            i = self._fork_ctr = getattr(self, "_fork_ctr", 0) + 1
            if not (self.fork_range[0] <= i - 1 < self.fork_range[1]):
                fn()
                return
            dummy = getattr(self, "fork_dummy", None)
            if dummy:      # the real launches stay on the caller's stream; the side stream gets a stand-in (a sleep, or a copy that streams HBM)
                fn()
                s = self.wgrad_stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    if dummy == "sleep":
                        torch.cuda._sleep(600000)
                    else:
                        if getattr(self, "_dummy_buf", None) is None:
                            self._dummy_buf = torch.empty(2, 1 << 28, device="cuda", dtype=torch.uint8)
                        for _ in range(4):
                            self._dummy_buf[1].copy_(self._dummy_buf[0])
                return
        want = self.defer_wgrad == "all" or (self.defer_wgrad == "big" and defer)
        if want and self.overlap_wgrad and torch.cuda.is_available():
            self.flush_wgrad()
            self._pending_wgrad.append((fn, tensors))
            if not self._join_armed:
                try:
                    torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
                    self._join_armed = True
                except RuntimeError:
                    self.flush_wgrad()
            return
        self._issue_wgrad(fn, tensors)

    def flush_wgrad(self):
        pend, self._pending_wgrad = self._pending_wgrad, []
        for fn, tensors in pend:
            self._issue_wgrad(fn, tensors)
        done, self._pending_done = self._pending_done, []
        for d in done:
            d()

    # ---- grouped weight gradients ---------------------------------------------------------------------------------------
    def can_park(self):
        """Inside on_wgrad_stream's closure: on a side stream, or - one-stream runs (bench.py's CROG_SINGLE_STREAM profile mode) - inline, the
        group then goes to the caller's stream when it is flushed (same kernels as the default path, so the single-stream profile shows them)."""
        # (not under DDP: a parked gradient announces itself late, and the bucket all-reduce it completes would start late)
        if not (self.group_wgrad and not self.deterministic and getattr(self, "_in_wgrad", False)):
            return False
        return getattr(self, "_override", None) is not None or not self.overlap_wgrad

    def can_park_K(self, Kd: int) -> bool:
        """Which reduction lengths may be parked under DistributedDataParallel: all (DDP_PARK_ALL above), or only the text tower's 640 token
        rows (its buckets complete when its latency-bound chain does, grouped or not)."""
        return self.reducer is None or Kd < 4096 or DDP_PARK_ALL

    @property
    def _group(self):
        """Anything parked on any stream?"""
        return any(g["descs"] for gs in self._groups.values() for g in gs.values())

    GROUPS_PER_STREAM = 2      # open groups per stream (one per reduction length): the decoder alternates 21632-pixel and 640-token reductions

    def park_wgrad(self, desc, blocks, keep, Kd):
        """Park one weight gradient for a grouped launch on the current weight-gradient stream.  Groups are kept per reduction length
        (round 5: until then a change of length flushed what was parked, and the decoder, whose cross-attention alternates pixel and
        token rows, went out in launches of 12-36 blocks); a THIRD length on a stream flushes the oldest group - the backward pass has
        moved on to layers of another resolution, what is parked goes now, not at the end."""
        s = getattr(self, "_override", None)
        gs = self._groups.setdefault(s, {})
        if Kd not in gs:
            while len(gs) >= self.GROUPS_PER_STREAM:
                self.flush_group(s, next(iter(gs)), all_streams=False)
            gs[Kd] = dict(descs=[], keep=[], n=0, K=Kd, done=[])
        g = gs[Kd]
        g["last"] = getattr(self, "_wgrad_calls", 0)
        g["descs"].append(desc)
        g["keep"].append(keep)
        g["n"] += blocks
        self._arm_end_of_backward()
        if g["n"] >= self.group_blocks or len(g["descs"]) >= 32:
            self.flush_group(s, Kd, all_streams=False)

    GROUP_STALE = int(os.environ.get("CROG_GROUP_STALE", "8"))

    def note_wgrad(self):
        """Every weight gradient passes here (functional.wgrad_gemm), parked or not: a group that nothing has joined for GROUP_STALE weight
        gradients goes now - the backward pass has left its layers.  (Without this the remainders of layer3 and layer2 - 99 and 143 blocks,
        just under the launch threshold - waited for the end of backward, where they ran alone behind the last weight gradient: 0.6 ms of
        the step's tail in profiles/r05_kernel_trace; layer1's and the stem's weight gradients are never parked, so no third reduction
        length arrived to push them out.  Flushing on every other length instead cut the neck's groups into launches of 16-36 blocks.)"""
        self._wgrad_calls = n = getattr(self, "_wgrad_calls", 0) + 1
        for s, gs in list(self._groups.items()):
            for k in [k for k, g in gs.items() if g["descs"] and n - g.get("last", n) >= self.GROUP_STALE]:
                self.flush_group(s, k, all_streams=False)

    def flush_short_groups(self, below: int = 4096):
        """Launch the parked groups of token-row reductions (the text tower's: K = batch x context length) on every stream."""
        for s, gs in list(self._groups.items()):
            for k in [k for k in gs if k < below]:
                self.flush_group(s, k, all_streams=False)

    def defer_done(self, lo: int, hi: int, fn) -> bool:
        """WRef.done(): is a gradient in the address range [lo, hi) still parked?  Then its announcement waits for that group's launch."""
        for gs in self._groups.values():
            for g in gs.values():
                for d in g["descs"]:
                    c, a = d.C or 0, d.a_sum or 0
                    if lo <= c < hi or lo <= a < hi:
                        g["done"].append(fn)
                        return True
        return False

    def flush_group(self, stream=None, Kd=None, all_streams=True):
        """Launch what is parked on `stream` (None with all_streams: on every stream; Kd: only the group of that reduction length) - each
        stream was ordered behind a request's producers when it was parked - then let the announcements that waited for it through
        (WRef.done)."""
        streams = list(self._groups) if (stream is None and all_streams) else [stream]
        for s in streams:
            gs = self._groups.get(s)
            if not gs:
                continue
            for k in ([Kd] if Kd is not None else list(gs)):
                g = gs.pop(k, None)
                if not g or not g["descs"]:
                    continue
                descs, keep, done = g["descs"], g["keep"], g["done"]
                if _DBG_GROUP:
                    print(f"[group] stream {'caller' if s is None else hex(s.cuda_stream)} blocks {g['n']} K {g['K']}: " + ", ".join(f"{d.M}x{d.N}/{d.splitk}" for d in descs), flush=True)
                prev = K._STREAM_OVERRIDE
                K.set_stream_override(s.cuda_stream if s is not None else None)      # (None: parked inline, launched on the caller's stream)
                try:
                    K.gemm_group(descs)
                finally:
                    K.set_stream_override(prev)
                del keep
                for d in done:
                    d()

    def _issue_wgrad(self, fn, tensors):
        s = self.wgrad_stream()
        if s is None:
            self._in_wgrad = True
            try:
                fn()
            finally:
                self._in_wgrad = False
            return
        if not self._join_armed:
            # at the end of this backward pass the caller's stream waits for the side streams, so that whatever the user
            # does next with the gradients (any optimizer, .norm(), clipping, ...) is ordered after them
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
                self._join_armed = True
            except RuntimeError:
                pass
        cur = torch.cuda.current_stream()
        if self.aux_stream is not None and self.text_stream is not None and cur == self.text_stream:
            s = self.adam_stream()
        s.wait_stream(cur)
        self._override = s
        # fn() only launches kernels of this library: point them at the side stream directly instead of paying torch's
        # stream-context manager (~25 us of Python) ~150 times per step
        K.set_stream_override(s.cuda_stream)
        self._in_wgrad = True
        try:
            fn()
        finally:
            self._in_wgrad = False
            K.set_stream_override(None)
            self._override = None
        for t in tensors:
            if t is not None:
                t.record_stream(s)
        dbg = os.environ.get("CROG_DBG_FORK", "")      # probe only (scripts/det_stress.py)
        if dbg:
            if "keep" in dbg:
                self._dbg_keep = getattr(self, "_dbg_keep", [])
                self._dbg_keep.append(tensors)
                self._arm_end_of_backward()
            if "serial" in dbg:
                cur.wait_stream(s)

    def adam_stream(self):
        """Where FusedAdam steps its chunks during backward: the aux stream, else the weight-gradient stream (None: no side streams)."""
        if self.aux_stream is None or (self.deterministic and self.det_streams != "all") or not self.overlap_wgrad:
            return self.wgrad_stream()
        if self.aux_stream not in self.streams:
            self.streams.append(self.aux_stream)
        return self.aux_stream

    def _arm_end_of_backward(self):
        if not self._join_armed:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
                self._join_armed = True
            except RuntimeError:      # not inside a backward pass
                pass

    def watch_slot(self, slot):
        self._slots.append(slot)
        self._arm_end_of_backward()

    def _end_of_backward(self):
        self._join_armed = False
        self._dbg_keep = []
        self.flush_wgrad()
        self.flush_group()
        self.join_streams()
        slots, self._slots = self._slots, []
        if any(s.t is not None for s in slots):
            raise RuntimeError("crog_amd GradSlot: a residual / downsample gradient was parked for an op whose backward never took it "
                               "(it would have been dropped silently): the autograd order the slot relies on did not hold for this backward")

    def join_streams(self):
        """Make the current stream wait for every side stream (before the optimizer step / gradient zeroing)."""
        if not torch.cuda.is_available():
            return
        self.flush_wgrad()
        self.flush_group()
        cur = torch.cuda.current_stream()
        for s in self.streams:
            if s != cur:
                cur.wait_stream(s)

    # ---- pre-zeroed fp32 scratch: one memset per step instead of one per BatchNorm reduction ---------------------------
    ZERO_POOL = 1 << 23   # floats (32 MiB): BatchNorm statistic replicas, reduction targets and column-sum workspaces of one step

    def begin_step(self, device):
        """Called once per training forward: clears the zero pool and rewinds its bump pointer."""
        if not getattr(self, "_env_checked", False):      # CROG_DETERMINISTIC=1: the bit-reproducible mode from the first step on
            self._env_checked = True
            if os.environ.get("CROG_DETERMINISTIC") == "1" and not self.deterministic and not torch.cuda.is_current_stream_capturing():
                set_deterministic(True)
        if getattr(self, "_zpool", None) is None or self._zpool.device != device:
            self._zpool = torch.zeros(self.ZERO_POOL, device=device, dtype=torch.float32)
        else:
            self._zpool.zero_()
        self._zptr = 0
        self._fork_ctr = 0

    def zeros(self, n: int, device):
        """n pre-zeroed floats valid until the next begin_step() -> (tensor, True), or a fresh torch.zeros -> (tensor, True)."""
        pool = getattr(self, "_zpool", None)
        n8 = (n + 7) // 8 * 8
        if pool is None or pool.device != device or self._zptr + n8 > pool.numel():
            t = torch.zeros(n, device=device, dtype=torch.float32)
            if K._STREAM_OVERRIDE is not None and getattr(self, "_override", None) is not None:
                # the caller launches on a side stream (on_wgrad_stream): order that stream behind the fill above and keep
                # the block from being handed out again while it is still read there
                s = self._override
                s.wait_stream(torch.cuda.current_stream())
                t.record_stream(s)
            return t
        t = pool[self._zptr:self._zptr + n]
        self._zptr += n8
        return t

    def enable_seed_epoch(self, device):
        """Dropout seeds = host seed + a device-resident epoch (crog_set_seed_epoch): a captured step advances the epoch itself, so
        every replay drops different elements (crog_amd.graphs.GraphedTrainStep)."""
        if getattr(self, "seed_epoch", None) is None or self.seed_epoch.device != torch.device(device):
            self.seed_epoch = torch.zeros(1, device=device, dtype=torch.int64)
            K.set_seed_epoch(self.seed_epoch)
        return self.seed_epoch

    def next_seed(self) -> int:
        self._seed_ctr += 1
        return ((self.seed_base & 0xFFFFFFFF) << 32) | (self._seed_ctr & 0xFFFFFFFF)

    def manual_seed(self, seed: int):
        self.seed_base = int(seed)
        self._seed_ctr = 0
        if getattr(self, "seed_epoch", None) is not None:
            self.seed_epoch.zero_()


RT = Runtime()


def set_deterministic(on: bool = True):
    """Switch the bit-reproducible mode on or off, for the kernel library (crog_set_deterministic: ordered forms of the embedding
    scatter, the head's bias / tap sums, the loss sums and the long-slab reductions) and for the launch policy of this package
    (functional.py: BatchNorm / LayerNorm statistics as per-tile slabs + ordered reduction - the fp32 parity mode's path - for bf16
    too, no BatchNorm-backward statistics in GEMM epilogues, split-K weight gradients as slabs + crog_splitk_reduce, bias gradients
    by the two-pass column sum instead of a_sum, no grouped weight-gradient launches).
    Streams: all of the default mode's (RT.det_streams = "all"; CROG_DET_STREAMS=text keeps only the text tower's, =0 none).  Rounds 3-4 had to
    keep the whole mode on one stream: beside a forked stream a LayerNorm backward returned rows that differed in the last bf16 bit from run
    to run with identical operands, and round 5 found the bilinear x2 backward doing the same beside a forked weight gradient.  Root cause
    (scripts/det_probe.py, det_stress.py with reference dumps, pk_probe.py; LAB_NOTES section 10): the PACKED-FP32 VALU instructions
    (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, which the compiler's SLP vectoriser forms from adjacent scalar fp32 operations) return wrong
    results in lanes 48-63 of a wave while an MFMA kernel of another stream shares the SIMD: stand-alone, 2997 of 3000 launches of the
    bilinear backward beside a 3x3 weight-gradient GEMM differ from the serial result, 0 of 3000 once the library is built without those
    instructions (crog_amd/_lib.py NO_PACKED_F32).  With that build 0 of 23 + 23 + 7 full-depth passes differ with every side stream on
    (11 of 11 before).  (A first reading of round 5 blamed ds_bpermute_b32 - the LayerNorm backward's two row sums were a v_pk_add_f32
    pair behind each shuffle; with packed fp32 off the shuffle build is clean too.  The DPP / v_readlane reductions that replaced the
    shuffles stay: they are what the library uses.)
    Call it before the first step and outside a capture (it allocates the library's scratch once)."""
    if on:
        from ._lib import packed_f32_off
        if not packed_f32_off(K.lib()):
            raise RuntimeError("crog_amd: deterministic mode needs a library built without packed-fp32 VALU instructions "
                               f"(crog_build_flags: {K.lib().crog_build_flags().decode()}); rebuild with crog_amd._lib.build()")
    K.check(K.lib().crog_set_deterministic(1 if on else 0), "set_deterministic")
    RT.deterministic = bool(on)


def slab_scratch(n: int, device) -> torch.Tensor:
    """n fp32 elements of scratch for a launch that may run on the weight-gradient stream (split-K slabs): the block is kept from
    being handed out again while that stream still uses it."""
    t = torch.empty(n, device=device, dtype=torch.float32)
    if "keepws" in os.environ.get("CROG_DBG_FORK", ""):
        RT._dbg_keep = getattr(RT, "_dbg_keep", [])
        RT._dbg_keep.append(t)
        RT._arm_end_of_backward()
    if K._STREAM_OVERRIDE is not None:
        # (the stream the closure is really launched on: _issue_wgrad redirects the text tower's weight gradients to the aux stream)
        s = getattr(RT, "_override", None) or (RT._wgrad_stream[0] if RT._wgrad_stream else None)
        if s is not None:
            t.record_stream(s)
    return t


_LEGACY_SYNC = False   # (round-1 A/B: always memset G in zero_grad, always re-cast the shadow per forward)


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
        self.T: Dict[torch.dtype, torch.Tensor] = {}   # data-gradient layout of the 3x3 conv weights, per compute dtype
        self.t_fresh: Dict[torch.dtype, bool] = {}
        self.by_param: Dict[int, int] = {}
        conv3, tr = [], []
        self.lin_t = set()       # ids of 2-D / 1x1 parameters that have a transposed copy in `weights_t` (K-contiguous data gradients)
        for name, p, o in plan:
            n = p.numel()
            if (p.dim() == 2 or (p.dim() == 4 and p.shape[2] * p.shape[3] == 1)) and p.shape[0] % 8 == 0 and p.shape[1] % 8 == 0 and n <= (1 << 23):
                # ([rows, cols] with both sides whole 16-byte vectors; embedding tables - 25 M elements, no data gradient - stay out)
                tr += [o, p.shape[0], p.shape[1], 1]
                self.lin_t.add(id(p))
            if p.dim() == 4 and p.shape[2] * p.shape[3] > 1:
                co, ci, kh, kw = p.shape
                if kh == 3 and kw == 3:
                    conv3 += [o, co, ci]
                    tr += [o, co, ci, 9]
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
        self.explicit = False    # set by crog_amd models: all their gradients are announced through WRef.done()
        self.g_clean = False
        self.touched = set()     # ids of parameters whose gradient has been written at least once
        self.written = set()     # ... since the last zero_grad (FusedAdam: weight decay skips the others, as torch skips grad-None parameters)
        self.gview = {id(p): g for _, p, _, _, g in self.entries}
        self.conv3_count = len(conv3) // 3
        self.conv3_table = torch.tensor(conv3, dtype=torch.int64, device=device) if conv3 else None
        self.tr_count = len(tr) // 4
        self.tr_table = torch.tensor(tr, dtype=torch.int64, device=device) if (tr and device.type == "cuda") else None

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

    def weights_t(self, dtype: torch.dtype) -> torch.Tensor:
        """Flat buffer (same offsets as `weights`) holding the data-gradient layout of the weights: 3x3 convolutions as
        [Cin][8 - tap][Cout], 1x1 convolutions / linears (`lin_t`) as [cols][rows]; refreshed by one launch per step, on first use after
        the weights changed.  Models whose backward runs on several streams refresh it at the start of the forward, before the
        streams fork (CROG.forward), so that every later reader is ordered behind the launch."""
        src = self.weights(dtype)
        self.ensure_t(dtype)
        if not self.t_fresh[dtype]:
            if self.tr_table is not None:
                K.dgrad_weights(src, self.T[dtype], self.tr_table, self.tr_count)
            self.t_fresh[dtype] = True
        return self.T[dtype]

    def ensure_t(self, dtype: torch.dtype):
        """Allocate (and zero) the data-gradient weight buffer on torch's CURRENT stream.  Callers that launch the refresh on another
        stream call this first, on the stream the other one then waits for: a buffer first created inside the side-stream launch would
        be zero-filled by the current stream, unordered against the refresh."""
        if dtype not in self.T:
            self.T[dtype] = torch.zeros(self.total, device=self.device, dtype=dtype)
            self.t_fresh[dtype] = False

    def valid(self) -> bool:
        name, p, o, n, g = self.entries[0]
        return p.data_ptr() == self.P.data_ptr() + 4 * o

    def relink_grads(self):
        """Make sure every p.grad is (still) the view into G (optimizer.zero_grad(set_to_none=True) drops it)."""
        for name, p, o, n, g in self.entries:
            if p.grad is not g:
                p.grad = g

    def grads_dropped(self) -> bool:
        """True when somebody set a parameter's .grad to None (torch's default `zero_grad(set_to_none=True)`)."""
        for name, p, o, n, g in self.entries:
            if p.grad is None:
                return True
        return False

    def request_zero(self):
        """engine.train_step, before the forward: the caller WILL call zero_grad() between this forward and its backward, so the forward may
        do the memset early on a side stream it joins before it returns (`zero_early`).  Forgotten by the next zero_grad() either way."""
        self._zero_req = True

    def zero_early(self):
        """CROG.forward, on the weight-gradient stream (idle during the forward): the memset zero_grad() would do on the main stream."""
        if not getattr(self, "_zero_req", False):
            return
        self._zero_req = False
        if _LEGACY_SYNC or not getattr(self, "explicit", False) or getattr(self, "g_clean", False):
            return
        RT._issue_wgrad(lambda: K.zero_f32(self.G), ())      # (behind everything enqueued on the caller's stream so far)
        self.g_clean = True

    def zero_grad(self):
        """One memset of the flat gradient buffer, skipped while the buffer is known to be all zeros.  "Known" needs the owner's
        promise (`explicit`, set by the crog_amd models) that every gradient it produces is written by a kernel path ending in
        WRef.done(), which marks the buffer dirty; a generic module whose gradients arrive through autograd's AccumulateGrad gets
        the memset every time.  Also restores dropped .grad links."""
        self._zero_req = False
        if _LEGACY_SYNC or not (getattr(self, "explicit", False) and getattr(self, "g_clean", False)):
            self.G.zero_()
            self.g_clean = getattr(self, "explicit", False) and not _LEGACY_SYNC
        self.written = set()
        self.relink_grads()

    def fresh_grads_if_dropped(self):
        """torch semantics of `p.grad = None` (model.zero_grad(), a stock optimizer's zero_grad()): the next backward REPLACES the
        gradient.  The kernels always accumulate into G, so a dropped link means: clear that parameter's slice of G (one memset
        when every link is gone); the link itself comes back in WRef.done() when a kernel writes the gradient, so a parameter the
        forward never uses (`logit_scale`) keeps .grad = None, as in torch.  Called at the start of a training forward and again by
        the first node of backward (the reference zeroes between the two, crog_engine.py:77).  Gradients that are still linked are
        left alone: forward/backward twice without zero_grad accumulates, as in torch."""
        # decided by IDENTITY (advisor, round 2): only parameters a kernel has written (`touched`) can hold anything in G, so
        #  * all of those dropped -> one memset of G (the usual case: a stock optimizer's zero_grad drops every link);
        #  * some of them dropped -> clear exactly their slices; gradients that are still linked keep accumulating;
        #  * a dropped parameter no kernel ever wrote (`logit_scale`) has a zero slice already: nothing to launch for it.
        touched = [(p, o, n) for name, p, o, n, g in self.entries if (not self.explicit) or id(p) in self.touched]
        dropped = [(o, n) for p, o, n in touched if p.grad is None]
        if not dropped:
            return
        if len(dropped) == len(touched):
            if _LEGACY_SYNC or not (self.explicit and self.g_clean):
                self.G.zero_()
                self.g_clean = self.explicit and not _LEGACY_SYNC
        else:
            for o, n in dropped:
                self.G[o:o + n].zero_()

    def invalidate_shadow(self):
        """The fp32 parameters changed behind the store's back (load_state_dict, a foreign optimizer, in-place edits)."""
        self.shadow_fresh = False
        self.synced = False
        for k in self.t_fresh:
            self.t_fresh[k] = False

    def forward_begins(self):
        """Called by every forward.  FusedAdam writes the bf16 shadow itself while it updates the parameters (`shadow_written`), so
        the first forward after its step needs no cast pass; anything else falls back to re-casting."""
        if getattr(self, "synced", False) and not _LEGACY_SYNC:
            self.synced = False            # consumed: a later forward without an optimizer step in between re-casts
        else:
            self.invalidate_shadow()

    def shadow_written(self):
        self.shadow_fresh = True
        self.synced = True
        for k in self.t_fresh:
            self.t_fresh[k] = False        # the data-gradient weight layout is derived from the shadow
