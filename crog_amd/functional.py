"""Autograd-visible operators of the HIP path.

Every operator is a torch.autograd.Function whose forward and backward enqueue HIP kernels from
libcrog_hip.so through crog_amd.kernels.  Autograd is used only to order the backward launches;
no ATen kernel computes anything here.

Conventions
  * activations are channels-last row matrices [..., C] in the compute dtype (bf16 or fp32)
  * weights are read from the flat store in the compute dtype (`WRef.w()`), weight gradients are
    accumulated with fp32 atomics straight into the flat gradient buffer (`WRef.g()`); backward
    therefore returns None for parameter inputs (p.grad already aliases that buffer)
  * token tensors are batch-first: row = b * L + l
"""
from __future__ import annotations

import math
import contextlib
import os
from typing import List, Optional, Tuple

import torch
from torch.autograd import Function

from . import kernels as K
from .runtime import RT, ParamStore, slab_scratch


def _cdt(t: torch.Tensor) -> int:
    return K.dcode(t)


def _bk(dt: int) -> int:
    return 32 if dt == K.BF16 else 16


def _vec(dt: int) -> int:
    return 8 if dt == K.BF16 else 4


def _pad(n: int, m: int) -> int:
    return (n + m - 1) // m * m


# Policy constants of the operator layer.  They were environment switches while rounds 1-2 measured them (DESIGN.md section 4 has the
# numbers); the measured winners are now fixed, and tests that compare two forms set the module attribute.
# BatchNorm slab reductions up to this many partial rows are folded into the finalize launch (ordered: same bits run after run).  256 until
# round 5; longer slabs then went through crog_reduce_pairs' atomic adds, and the fp32 parity path was NOT bit-reproducible on layers of more
# than 32768 rows (scripts/fp32_repro.py: config 1's logits differed in every pass).  The ordered reduction walks 64 rows at a time now
# (norm.hip reduce_slab): every slab takes it.
FUSED_REDUCE_MAX_PARTS = 1 << 30
FLASH_ATTN = True                # fused attention kernels (csrc/attn.hip) where they apply; tests compare with the unfused path
FLASH_MIN_KEYS = 64
FLASH_CAUSAL = os.environ.get("CROG_FLASH_CAUSAL", "1") != "0"      # causal self-attention (the CLIP text tower) through the fused kernels too
FLASH_CROSS = os.environ.get("CROG_FLASH_CROSS", "1") != "0"        # the decoder's vision-to-text cross-attention (20 keys, key padding mask) through the fused kernels:
#                                                                     one launch forward and two backward instead of three and five
FLASH_CROSS_MIN_QUERIES = 128
FLASH_KEEP = os.environ.get("CROG_FLASH_KEEP", "1") != "0"          # the fused forward leaves its dropout decisions as a bit map for the two backward kernels
FWD_STAT_SYNC = os.environ.get("CROG_SYNCBN_FUSE_FWD", "1") != "0"      # SyncBatchNorm forward exchanges in / behind the statistics GEMM (round 5)
BN_ATOMIC_STATS = True           # BN statistics: atomic replicas in the GEMM epilogue + in-kernel finalize (bf16)
RELU_BITMASK = True              # residual+ReLU layers keep a bit mask of y for backward (1/16 of y's bytes)
LN_BWD_ATOMIC = False            # LayerNorm parameter gradients through atomics in ln_bwd itself: measured 0.5 % SLOWER (every block adds into the same 2 C floats)
LN_REDUCE_SIDE = True            # LayerNorm parameter-gradient reduction on the weight-gradient stream
CONV3_SLAB_MAX = 147456          # ... and so do the other 3x3 weight gradients with outputs up to 128 x 1152
SW_SLABS = True                  # sliding-window 3x3 weight gradients (stem, layer1) meet in slabs + one ordered reduction, not in atomic adds
FAN_SLOTS = True                 # a map with a consumer outside its block (layer2 / layer3 -> neck, layer4 -> attention tokens): that gradient rides a GradSlot too
LN_GRAD_SLOTS = True             # decoder: a residual's gradient is added inside the LayerNorm backward of the same tensor (GradSlot -> crog_ln_bwd dxadd)
BN_BWD_ATOMIC = True             # backward partial sums through coalesced atomics (bf16)
DGRAD_T = True                   # 3x3 data gradients on the transposed weight copy (forward-shaped GEMM)
LIN_DGRAD_T = True               # ... and the 1x1 / linear ones (round 3); tests compare with the transposed-read form
FUSED_HEAD = True                # fold vis.4 into the dynamic head (no groups*C-channel map)
GROUP_BIAS = os.environ.get("CROG_GROUP_BIAS", "1") != "0"        # grouped weight gradients may carry their bias gradient (crog_gemm_group: a_sum blocks)
GROUP_MIN_K = int(os.environ.get("CROG_GROUP_MIN_K", "512"))      # shortest reduction that is parked for a grouped launch (4096 until round 5)
GROUP_MAX_TILES = int(os.environ.get("CROG_GROUP_MAX_TILES", "36"))
# ... and outputs of at most this many 256 x 256 tiles that crog_gemm would give the ping-pong tile ALONE (split 9-16 ways to reach 144 blocks:
# 9.4 M atomic adds per launch whatever the output size) join a group too, at ~84 k-tiles per block
GROUP_BIG_TILES = int(os.environ.get("CROG_GROUP_BIG_TILES", "16"))
GROUP_KT = int(os.environ.get("CROG_GROUP_KT", "120"))      # k-tiles (of 64 rows) per block of a grouped launch


class OutRef:
    """Hides a preallocated destination view (a channel slice of a concat buffer) from autograd's argument scan."""

    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        self.t = t


def _dest(out: Optional["OutRef"]):
    return None if out is None else out.t.view(out.t.shape)


class WRef:
    """A block of `rows` consecutive rows (each `cols` long) of a parameter in the flat store."""

    __slots__ = ("store", "param", "off", "rows", "cols")

    def __init__(self, store: ParamStore, param, row0: int = 0, rows: Optional[int] = None, cols: Optional[int] = None):
        self.store = store
        self.param = param
        total_rows = param.shape[0]
        self.cols = cols if cols is not None else param.numel() // total_rows
        self.rows = rows if rows is not None else total_rows - row0
        self.off = store.off(param) + row0 * self.cols

    def w(self, dtype: torch.dtype) -> torch.Tensor:
        return self.store.weights(dtype)

    @property
    def G(self) -> torch.Tensor:
        return self.store.G

    @property
    def P(self) -> torch.Tensor:
        return self.store.P

    def master(self) -> torch.Tensor:
        """fp32 view of the block (biases, norm scales)."""
        return self.store.P[self.off:self.off + self.rows * self.cols]

    def grad(self) -> torch.Tensor:
        return self.store.G[self.off:self.off + self.rows * self.cols]

    def done(self, then=None):
        """The kernels that write this parameter's gradient are enqueued (or parked: Runtime.defer_wgrad - then the announcement waits
        for the flush that really enqueues them, so a DDP bucket can never be launched ahead of its last gradient).
        then: what to call instead of the announcement itself (done_joint: one of several row blocks of a packed parameter)."""
        fn = then if then is not None else self._done_now
        if RT._pending_wgrad:
            # (re-evaluated by flush_wgrad once the deferred closure has run: if that closure PARKED the gradient for a grouped launch the
            # announcement moves on to the group - ADVICE r5: _done_now here let FusedAdam / a DDP bucket go ahead of the grouped GEMM)
            RT._pending_done.append(self.done if then is None else (lambda: self.done(then)))
            return
        if RT._groups:      # parked for a grouped launch (Runtime.park_wgrad): announced when that launch is enqueued
            lo = self.store.G.data_ptr() + 4 * self.off
            if RT.defer_done(lo, lo + 4 * self.rows * self.cols, fn):
                return
        fn()

    def _done_now(self):
        st, p = self.store, self.param
        st.g_clean = False
        st.touched.add(id(p))
        st.written.add(id(p))
        if p.grad is None:      # dropped by a set_to_none zero_grad between forward and backward: G was cleared by BackwardBegin
            p.grad = st.gview[id(p)]
        if RT.reducer is not None:
            RT.reducer.mark_ready(p)
        if RT.early_adam is not None:
            RT.early_adam.mark_ready(p)


def done_joint(refs):
    """Announce parameters whose gradient is written by SEVERAL launches - the q / k / v row blocks of a packed in_proj_weight / in_proj_bias
    (clip.py:246, layers.py:291-296) - exactly once each, after the LAST of those launches is really enqueued.  Round 6: the blocks were
    announced one by one, and any one of them marks the whole parameter ready; with the blocks parked in different grouped launches (the
    decoder's cross-attention: q reduces over 21632 pixel rows, k and v over 640 token rows - two groups) the first group to go out announced
    the parameter while another block's GEMM was still parked, i.e. a DDP bucket could be reduced (or an Adam chunk stepped) ahead of it."""
    by_param = {}
    for r in refs:
        if r is not None:
            by_param.setdefault(id(r.param), []).append(r)
    for rs in by_param.values():
        if len(rs) == 1:
            rs[0].done()
            continue
        left = [len(rs)]

        def arrive(left=left, first=rs[0]):
            left[0] -= 1
            if left[0] == 0:
                first._done_now()
        for r in rs:
            r.done(arrive)


# ------------------------------------------------------------------------------------------------
# GEMM helpers on row matrices
# ------------------------------------------------------------------------------------------------
def lin_fwd(x, w: WRef, out, *, bias: Optional[WRef] = None, act=K.ACT_NONE, res=None, stats=None, out_mode=K.OUT_T, alpha=1.0,
            c_off=0, ldc=None, N=None):
    """out[M, N] = act(alpha * x[M, K] @ W[N, K]^T + bias) + res"""
    M, Kd, lda = K.mat(x)
    dt = _cdt(x)
    n = w.rows if N is None else N
    K.gemm(dt, K.A_KC, K.B_KC, x, w.w(x.dtype), out, M, n, Kd, lda, w.cols, ldc if ldc is not None else K.mat(out)[2],
           b_off=w.off, bias=bias.master() if bias is not None else None, act=act, R=res, ldr=K.mat(res)[2] if res is not None else 0,
           col_stats=stats, out_mode=out_mode, alpha=alpha, c_off=c_off)


def lin_dgrad(dy, w: WRef, dx, *, accumulate_into: Optional[torch.Tensor] = None, a_off=0, lda=None, N=None):
    """dx[M, K] = dy[M, N] @ W[N, K]  (+ accumulate_into)"""
    M, n, ld = K.mat(dy)
    if N is not None:
        n = N
    dt = _cdt(dx)
    R, ldr = accumulate_into, K.mat(accumulate_into)[2] if accumulate_into is not None else 0
    if LIN_DGRAD_T and id(w.param) in w.store.lin_t and n % _vec(dt) == 0:
        # forward-shaped GEMM on the transposed copy of the weight ([K][total rows]: this block's rows are its columns row0 .. row0 + N)
        poff = w.store.off(w.param)
        total_rows = w.param.numel() // w.cols
        K.gemm(dt, K.A_KC, K.B_KC, dy, w.store.weights_t(dx.dtype), dx, M, w.cols, n, lda if lda is not None else ld, total_rows, K.mat(dx)[2],
               a_off=a_off, b_off=poff + (w.off - poff) // w.cols, R=R, ldr=ldr)
        return
    K.gemm(dt, K.A_KC, K.B_NC, dy, w.w(dx.dtype), dx, M, w.cols, n, lda if lda is not None else ld, w.cols, K.mat(dx)[2],
           a_off=a_off, b_off=w.off, R=R, ldr=ldr)


def wgrad_gemm(dt, b_layout, dy, x, G, M, N, Kd, lda, ldb, ldc, *, a_off=0, c_off=0, conv=(0, 0, 0), a_sum=None, a_sum_off=0, park=True):
    """G[c_off + m * ldc + n] += sum_k dy[k][m] * xcol[k][n]: every weight gradient of the package.  Split over the reduction; the
    slices meet through fp32 atomic adds, or - deterministic mode - as slabs of a [splitk][M][N] workspace that crog_splitk_reduce adds
    onto G in slice order (a_sum, an atomic sum as well, is refused there: the caller takes the two-pass column sum).
    park=False: the caller reads the result on the same stream right after this call (a padded scratch gradient that is stripped into the
    real one): the product must not wait for a grouped launch."""
    conv3 = b_layout == K.B_NC_IM2COL
    RT.note_wgrad()
    if (park and (a_sum is None or (GROUP_BIAS and not conv3)) and dt == K.BF16 and RT.can_park() and RT.can_park_K(Kd) and M >= 256 and N >= 256 and M % 8 == 0
            and N % 8 == 0 and Kd >= GROUP_MIN_K and a_off % 8 == 0 and lda % 8 == 0 and ldb % 8 == 0 and (Kd + 64) * max(lda, ldb) * 2 < 2 ** 31
            and M * ldc < 2 ** 31):
        # a small output (a few 256 x 256 tiles): parked, and launched together with its neighbours' by crog_gemm_group at ~84 k-tiles per
        # block - alone it would be split 16-fold to fill the chip (Runtime.park_wgrad).  Round 5: a bias gradient (a_sum) rides along as
        # extra blocks of the same launch, and short reductions qualify too (the text tower: 48 linears over 640 token rows, each a
        # latency-bound launch of its own before)
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        if tiles <= GROUP_MAX_TILES and (tiles <= GROUP_BIG_TILES or K.lib().crog_gemm_wgrad_tile(dt, K.A_MC, b_layout, M, N, Kd) != 256):
            gsk = max(1, min(16, round(Kd / 64 / GROUP_KT)))
            K.GROUP_SINK = sink = []
            try:
                K.gemm(dt, K.A_MC, b_layout, dy, x, G, M, N, Kd, lda, ldb, ldc, a_off=a_off, c_off=c_off, conv=conv, splitk=gsk,
                       out_mode=K.OUT_F32_ATOMIC, a_sum=a_sum, a_sum_off=a_sum_off)
            finally:
                K.GROUP_SINK = None
            RT.park_wgrad(sink[0], (tiles + (((M + 255) // 256) if a_sum is not None else 0)) * gsk, (dy, x, G, a_sum), Kd)
            return
    if conv3 and not RT.deterministic and a_sum is None and dt == K.BF16 and a_off == 0 and lda == M and ldb == conv[2] and SW_SLABS:
        # stem / layer1: the sliding-window kernel, one strip of image rows per workgroup, meeting in slabs rather than 9.4 M atomic adds
        slabs = K.lib().crog_wgrad_sw_slabs(int(M), int(conv[0]), int(conv[1]), int(conv[2]), int(Kd))
        if slabs:
            ws = slab_scratch(slabs * M * N, dy.device)
            K.gemm(dt, K.A_MC, b_layout, dy, x, ws, M, N, Kd, lda, ldb, N, conv=conv, splitk=slabs, out_mode=K.OUT_F32)
            K.splitk_reduce(ws, slabs, M, N, N, G, c_off, ldc, accumulate=True)
            return
    sk = K.pick_splitk(M, N, Kd, _bk(dt), conv=conv3)
    if (conv3 and not RT.deterministic and a_sum is None and dt == K.BF16 and SW_SLABS and M * N <= CONV3_SLAB_MAX and sk >= 16
            and K.lib().crog_gemm_wgrad_tile(dt, K.A_MC, b_layout, M, N, Kd) == 128):
        # small 3x3 outputs (layer2: 128 x 1152) split 56-113 ways: slabs at two workgroups per CU + the ordered reduction beat the atomic
        # adds (86528 pixels 78-86 -> 55 us, 346112 pixels 229 -> 170; scripts/ab_wgrad_slab.py)
        sk = min(sk, 512 // (((M + 127) // 128) * ((N + 127) // 128)))
        ws = slab_scratch(sk * M * N, dy.device)
        K.gemm(dt, K.A_MC, b_layout, dy, x, ws, M, N, Kd, lda, ldb, N, a_off=a_off, conv=conv, splitk=sk, out_mode=K.OUT_F32)
        K.splitk_reduce(ws, sk, M, N, N, G, c_off, ldc, accumulate=True)
        return
    if not RT.deterministic:
        K.gemm(dt, K.A_MC, b_layout, dy, x, G, M, N, Kd, lda, ldb, ldc, a_off=a_off, c_off=c_off, conv=conv, splitk=sk,
               out_mode=K.OUT_F32_ATOMIC, a_sum=a_sum, a_sum_off=a_sum_off)
        return
    assert a_sum is None, "deterministic mode: bias gradients go through bias_grad (crog_colsum), not a_sum"
    ldws = _pad(N, 4)
    ws = slab_scratch(sk * M * ldws, dy.device)
    K.gemm(dt, K.A_MC, b_layout, dy, x, ws, M, N, Kd, lda, ldb, ldws, a_off=a_off, conv=conv, splitk=sk, out_mode=K.OUT_F32)
    K.splitk_reduce(ws, sk, M, N, ldws, G, c_off, ldc, accumulate=True)


def lin_wgrad(dy, x, w: WRef, *, a_off=0, lda=None, N=None, bias: Optional[WRef] = None):
    """G[w] += dy[M, N]^T @ x[M, K];  with `bias`: G[bias] += column sums of dy, folded into the same launch (a_sum)."""
    M, n, ld = K.mat(dy)
    if N is not None:
        n = N
    _, Kd, ldx = K.mat(x)
    dt = _cdt(x)
    if bias is not None and (RT.deterministic or (a_off == 0 and K.lib().crog_gemm_wgrad_tile(dt, K.A_MC, K.B_NC, n, Kd, M) == 256)):
        # the 256 x 256 weight-gradient tile has no a_sum path, and a_sum is an atomic sum (deterministic mode): the bias gradient is
        # its own two-pass column sum
        bias_grad(dy, bias, a_off, n, pooled=True)
        bias = None
    wgrad_gemm(dt, K.B_NC, dy, x, w.G, n, Kd, M, lda if lda is not None else ld, ldx, w.cols, a_off=a_off, c_off=w.off,
               a_sum=bias.G if bias is not None else None, a_sum_off=bias.off if bias is not None else 0)


def bias_grad(dy, b: WRef, col0=0, n=None, pooled=False):
    """pooled: take the reduction workspace from the step's pre-zeroed pool (valid until the next step, never handed out twice):
    needed when the launch goes to the weight-gradient stream, where a fresh torch allocation would not be ordered."""
    M, C, ld = K.mat(dy)
    view = dy if (col0 == 0 and (n is None or n == C)) else dy[..., col0:col0 + n]
    ws = RT.zeros(K.colsum_workspace(M, K.mat(view)[1]), dy.device) if pooled else None
    K.colsum(view, b.G, b.off, ws=ws)


# ------------------------------------------------------------------------------------------------
# Conv (1x1 / 3x3 / stem / none) + BatchNorm(train|eval, cross-replica) + residual + ReLU
# ------------------------------------------------------------------------------------------------
class BnBuffers:
    """What conv_bn_act needs from a BatchNorm module."""

    def __init__(self, gamma: WRef, beta: WRef, running_mean, running_var, momentum, eps):
        self.gamma, self.beta = gamma, beta
        self.running_mean, self.running_var = running_mean, running_var
        self.momentum, self.eps = momentum, eps


class GradSlot:
    """Side channel that carries a residual branch's gradient from the op that consumes the identity (`res_slot`) to the op
    whose data gradient it must be added to (`grad_slot`), so the sum happens in that GEMM's residual epilogue instead of a
    separate autograd accumulation pass (Bottleneck, clip.py:44-57: grad(x) = dgrad(conv1) + grad(identity); with a downsample branch
    the producer is that branch's last backward op - AvgPool2Fn or the 1x1 ConvBnAct, `dx_slot` - instead of conv3's residual)."""

    __slots__ = ("t", "consumed")

    def __init__(self):
        self.t = None
        self.consumed = False

    # The contract rests on autograd's ordering (the producer's backward node runs before the consumer's: it was recorded later), not
    # on a data dependency.  Activation checkpointing, a partial torch.autograd.grad or a forward built on another thread can break
    # that order; the slot then fails loudly instead of dropping a gradient: a producer that arrives after its consumer raises, and so
    # does a gradient still parked in a slot when backward ends (Runtime._end_of_backward).
    def put(self, t):
        if self.consumed:
            raise RuntimeError("crog_amd GradSlot: the producer's backward ran AFTER the consumer's - the residual / downsample gradient "
                               "would have been dropped (autograd order violated: checkpointing or a partial backward over this block?)")
        self.t = t
        RT.watch_slot(self)

    def take(self):
        """The consumer's backward: the parked gradient (or None when the producer's branch carried no gradient)."""
        t, self.t = self.t, None
        self.consumed = True
        return t


class BnLink:
    """Side channel between two consecutive ConvBnAct layers whose only connection is y_L -> x_{L+1} (bn1 -> conv2, bn2 -> conv3 of
    a Bottleneck, the stem convolutions): layer L publishes its pre-normalisation activation and ReLU gate in the forward
    (`stat_out`); in backward layer L+1, whose data gradient IS layer L's dy, lets the GEMM epilogue gate that gradient and
    accumulate (sum g, sum g*z) (`stat_in`, crog_gemm bwd_z) — layer L then skips the first pass of BatchNorm backward, which
    would have re-read dy and z from HBM.  Only valid when NOTHING else consumes y_L: the model code that wires it vouches."""

    __slots__ = ("z", "ss", "C", "M", "sums", "R", "mask", "dx_ptr", "synced")

    def __init__(self):
        self.z = self.ss = self.sums = self.mask = None
        self.C = self.M = self.R = self.dx_ptr = 0
        self.synced = False      # the GEMM that left `sums` also exchanged them across the ranks (stat_sync): totals behind the R rows


BN_BWD_FUSED = True   # BnLink fusion on (bf16, atomic statistics path); tests compare with the two-launch form
# ... for layers of at most this many rows.  Measured per layer, HBM-cold (scripts/bench_bwd_fused.py): the gated-statistics epilogue
# adds 5-7 us to a 21632- / 86528-row data gradient and saves a 16-23 us first pass; on the 346112-row layers of layer1 it costs what it
# saves (+17 us vs 19-21 us: bn_bwd_partial streams at 4.2 TB/s, the epilogue gathers z in 4-byte pieces and its atomics collide on 64
# columns), and on the 1.38 M-row stem it LOSES 13-19 us.
BN_BWD_FUSED_MAX_ROWS = int(os.environ.get("CROG_BN_BWD_FUSED_MAX_ROWS", "131072"))
# The same between two Bottlenecks (a BnLink as `res_out` of block b-1's conv3 / bn3 layer and `res_in` of block b's conv1): the gradient
# of out_{b-1} = relu(bn3(z) + identity) is conv1's data gradient plus block b's identity gradient, both already joined in that GEMM
# (R) - its epilogue also gates the sum with the forward's bit mask and accumulates (sum g, sum g*z): block b-1 skips bn3's first
# backward pass over the widest tensors of the stage, and the stored g is at once bn3's dy and block b-1's identity gradient (no dres
# pass).  CROG_BN_RES_FUSED=0 switches it off; CROG_BN_RES_FUSED_MAX_ROWS limits the layer size.
BN_RES_FUSED = os.environ.get("CROG_BN_RES_FUSED", "1") != "0"
LN_RELU_FUSED = os.environ.get("CROG_LN_RELU_FUSED", "1") != "0"     # decoder FFN: ReLU backward inside the LayerNorm backward
POOL_FUSED = os.environ.get("CROG_POOL_FUSED", "1") != "0"      # average pooling inside the BatchNorm apply / backward passes (conv_bn_act pool=True)
BN_RES_FUSED_MAX_ROWS = int(os.environ.get("CROG_BN_RES_FUSED_MAX_ROWS", str(1 << 30)))


def stat_replicas(slabs: int, C: int) -> int:
    """Replica rows for the atomic BatchNorm statistics: at most ~256 adds land on one address (every tile of a layer adds to the
    same 2C floats: with ONE row the 346112 x 64 layer1 convolution takes 159 us instead of 68, the 1.4 M-row stem 540 instead of
    130), bounded so that the apply kernel's per-block sum over the rows stays a few KB."""
    r = 1
    while r * 256 < slabs and r < 32 and 4 * r * C <= 4096:
        r *= 2
    return r


class ConvBnAct(Function):
    """y = [relu]( BN( conv_k(x) ) [+ res] ).   ksize: 0 (no conv), 1, 3, or 's' (stem 3x3/s2 on an NCHW fp32 image).
    Reference: Bottleneck clip.py:44-57, stem clip.py:208-213, conv_layer layers.py:8-11, linear_layer layers.py:14-16,
    attnpool.connect clip.py:76-78, norm_layer layers.py:351,379.  Training-mode statistics are per-channel
    (sum, sum^2) pairs produced by the GEMM epilogue, optionally all-reduced across replicas (SyncBatchNorm,
    train_crog.py:113-114)."""

    @staticmethod
    def forward(ctx, x, res, _wp, _gp, _bp, w: Optional[WRef], bn: BnBuffers, ksize, relu: bool, training: bool, out, wpad, dtype,
                grad_slot=None, res_slot=None, stat_out=None, stat_in=None, dx_slot=None, res_out=None, res_in=None, pool=False):
        dev = x.device
        ctx.pool = None
        ctx.slots = (grad_slot, res_slot)
        ctx.dx_slot = dx_slot
        ctx.links = (stat_out, stat_in)
        ctx.res_links = (res_out, res_in)
        if ksize == "s":
            B, _, Hi, Wi = x.shape
            H, W = Hi // 2, Wi // 2
            lead = (B, H, W)
            M = B * H * W
            cin = 27
        else:
            lead = tuple(x.shape[:-1])
            M, cin, _ = K.mat(x)
            if ksize == 3:
                B, H, W = x.shape[0], x.shape[1], x.shape[2]
        dt = K.dcode(dtype)
        C = bn.gamma.rows * bn.gamma.cols
        wt = wbuf_off = None
        if ksize != 0:
            wt, wbuf_off, wcols = w.w(dtype), w.off, w.cols
            if wpad is not None:  # ragged Cin (stem 27 -> 32, CoordConv 514 -> 544): zero-padded compute copy of the weight
                src_cols, dst_cols, rows = wpad
                wt = torch.empty(rows * dst_cols, device=dev, dtype=dtype)
                K.cast_pad2d(w.P, src_cols, src_cols, wt, dst_cols, dst_cols, rows, src_off=w.off)
                wbuf_off = 0
                wcols = dst_cols * (9 if ksize == 3 else 1)
        stats = None
        stat_R = 0
        fwd_sync = None
        if ksize == 0:
            z = x
        else:
            z = torch.empty(lead + (C,), device=dev, dtype=dtype)
            if training:
                slabs = K.stat_tiles(M)
                comm_on = RT.comm is not None and (RT.comm.world_size > 1 or RT.comm.force)
                if BN_ATOMIC_STATS and dtype != torch.float32 and not RT.deterministic:
                    # (fp32 is the parity mode: it keeps the per-tile slab + ordered reduction, so the forward is bit-reproducible
                    # run to run; atomic accumulation order jitters the statistics by ~1e-7, which tiny BatchNorm layers amplify)
                    # statistics accumulate atomically into R pre-zeroed [C][2] rows in the GEMM epilogue and are finalised
                    # inside bn_apply: GEMM -> (all-reduce of all R rows) -> apply, no reduction / finalize launches in between
                    stat_R = stat_replicas(slabs, C)
                    # SyncBatchNorm: the exchange of the (sum x, sum x^2) totals rides in the tail of the GEMM that accumulates them (the
                    # ping-pong tile) or in a single-block launch crog_gemm adds behind any other kernel: [R rows][totals][ticket]
                    fwd_sync = RT.comm.fuse_ptr(2 * C) if (comm_on and FWD_STAT_SYNC) else None
                    stats = RT.zeros(stat_R * C * 2 + (2 * C + 8 if fwd_sync is not None else 0), dev)
                else:
                    stats = torch.empty(slabs, C, 2, device=dev, dtype=torch.float32)
            if ksize == "s":
                patches = torch.empty(M, 32, device=dev, dtype=dtype)
                K.stem_im2col(x, patches)
                K.gemm(dt, K.A_KC, K.B_KC, patches, wt, z, M, C, 32, 32, 32, C, b_off=wbuf_off, col_stats=stats, stat_replicas=stat_R, stat_sync=fwd_sync)
            elif ksize == 1:
                K.gemm(dt, K.A_KC, K.B_KC, x, wt, z, M, C, cin, K.mat(x)[2], wcols, C, b_off=wbuf_off, col_stats=stats, stat_replicas=stat_R, stat_sync=fwd_sync)
            else:
                K.gemm(dt, K.A_IM2COL, K.B_KC, x, wt, z, M, C, 9 * cin, K.mat(x)[2], wcols, C, b_off=wbuf_off, conv=(H, W, cin),
                       col_stats=stats, stat_replicas=stat_R, stat_sync=fwd_sync)
        ss = torch.empty(C, 2, device=dev, dtype=torch.float32)
        mi = None
        count = float(M)
        if pool:
            # (conv_bn_act only asks for this on the bf16 training path with in-kernel statistics, ReLU, no residual: POOL_FUSED)
            assert training and stat_R > 0 and relu and res is None and out is None and len(lead) == 3 and lead[1] % 2 == 0 and lead[2] % 2 == 0
            ctx.pool = (lead[1], lead[2])
            y = torch.empty((lead[0], lead[1] // 2, lead[2] // 2, C), device=dev, dtype=dtype)
        else:
            y = _dest(out) if out is not None else torch.empty(lead + (C,), device=dev, dtype=dtype)
        applied = False
        # residual + ReLU layers: backward needs sign(y); one bit per element is kept instead of re-reading y twice
        rmask = K.relu_mask_like(y) if (RELU_BITMASK and training and relu and res is not None) else None
        if training and stat_R > 0:
            mi = torch.empty(C, 2, device=dev, dtype=torch.float32)
            sums_in, rows_in = stats, stat_R
            if RT.comm is not None and (RT.comm.world_size > 1 or RT.comm.force):
                if fwd_sync is not None:      # the totals are already global: one row behind the replica rows
                    sums_in, rows_in = stats[stat_R * 2 * C:(stat_R + 1) * 2 * C], 1
                else:
                    RT.comm.all_reduce_sum(stats)
                count = float(M * RT.comm.world_size)
            K.bn_apply_stats(z, sums_in, rows_in, count, bn.gamma.master(), bn.beta.master(), bn.running_mean, bn.running_var, bn.momentum,
                             bn.eps, ss, mi, res, relu, y, relu_mask=rmask, pool=ctx.pool)
            applied = True
        elif training:
            if stats is None:
                rpb = K.bn_rows_per_block(M)
                nparts = (M + rpb - 1) // rpb
                stats = torch.empty(nparts, C, 2, device=dev, dtype=torch.float32)
                K.bn_partial_stats(z, stats, rpb)
            mi = torch.empty(C, 2, device=dev, dtype=torch.float32)
            comm_on = RT.comm is not None and (RT.comm.world_size > 1 or RT.comm.force)
            if M <= 64 and not comm_on and stats.shape[0] <= 8:
                # BatchNorm1d over the rows of a batch (linear_layer, layers.py:14-16): the in-kernel finalisation reads the few rows twice
                # (mean, then squared deviations) instead of E[x^2] - mean^2, which cancels to the last bits at B = 2 (crog_bn_apply_stats)
                K.bn_apply_stats(z, stats, stats.shape[0], count, bn.gamma.master(), bn.beta.master(), bn.running_mean, bn.running_var, bn.momentum,
                                 bn.eps, ss, mi, res, relu, y, relu_mask=rmask)
                applied = True
            elif comm_on:
                sums = RT.zeros(2 * C, dev).view(C, 2)
                K.reduce_pairs(stats, stats.shape[0], C, sums, zeroed=True)
                RT.comm.all_reduce_sum(sums)
                count = float(M * RT.comm.world_size)
                K.bn_finalize(sums, count, bn.gamma.master(), bn.beta.master(), bn.running_mean, bn.running_var, bn.momentum, bn.eps, C, ss, mi)
            elif stats.shape[0] <= FUSED_REDUCE_MAX_PARTS or RT.deterministic:   # single replica, short slab: statistics -> scale/shift in one launch
                K.bn_reduce_finalize(stats, stats.shape[0], count, bn.gamma.master(), bn.beta.master(), bn.running_mean, bn.running_var,
                                     bn.momentum, bn.eps, C, ss, mi)
            else:                                            # long slab (>= 100k rows): split reduction across blocks first
                sums = RT.zeros(2 * C, dev).view(C, 2)
                K.reduce_pairs(stats, stats.shape[0], C, sums, zeroed=True)
                K.bn_finalize(sums, count, bn.gamma.master(), bn.beta.master(), bn.running_mean, bn.running_var, bn.momentum, bn.eps, C, ss, mi)
        else:
            K.bn_eval_scale(bn.gamma.master(), bn.beta.master(), bn.running_mean, bn.running_var, bn.eps, C, ss)
        if not applied:
            K.bn_apply(z, ss, res, relu, y, relu_mask=rmask)
        ctx.rmask = rmask
        ctx.cfg = (ksize, relu, training, w, bn, wpad, count, cin, C, lead, dtype)
        ctx.has_res = res is not None
        ctx.relu_ss = ss if (relu and res is None and training) else None   # ReLU mask can be recomputed from z: y is not re-read in backward
        ctx.x_needs = ksize != "s" and x.requires_grad
        if stat_out is not None:
            # publish what the NEXT layer's data-gradient epilogue needs to do this layer's first backward pass (BnLink)
            # (no torch.is_grad_enabled() here: it is always False inside Function.forward)
            ok = (BN_BWD_FUSED and BN_BWD_ATOMIC and training and dtype == torch.bfloat16 and res is None and ksize != 0
                  and M <= BN_BWD_FUSED_MAX_ROWS and not RT.deterministic)      # (the epilogue's statistics are atomic adds)
            stat_out.z, stat_out.ss, stat_out.C, stat_out.M, stat_out.sums = (z if ok else None), (ss if relu else None), C, M, None
        if res_out is not None:
            # ... and what the NEXT BLOCK's first data gradient needs to do this residual layer's first backward pass (BN_RES_FUSED)
            ok = (BN_BWD_FUSED and BN_RES_FUSED and BN_BWD_ATOMIC and training and dtype == torch.bfloat16 and res is not None and relu
                  and rmask is not None and ksize != 0 and M <= BN_RES_FUSED_MAX_ROWS and C % 8 == 0 and not RT.deterministic)
            res_out.z, res_out.mask, res_out.C, res_out.M, res_out.sums, res_out.dx_ptr = (z if ok else None), rmask, C, M, None, 0
        ctx.wt = (wt, wbuf_off) if wpad is not None else None
        if ksize == "s":
            ctx.save_for_backward(patches, z, y, mi)
        else:
            ctx.save_for_backward(x, z, y, mi)
        return y

    @staticmethod
    def backward(ctx, dy):
        ksize, relu, training, w, bn, wpad, count, cin, C, lead, dtype = ctx.cfg
        if not training:
            raise NotImplementedError("crog_amd: backward through eval-mode BatchNorm is not supported (reference trains with model.train())")
        x, z, y, mi = ctx.saved_tensors
        dev = dy.device
        dt = K.dcode(dtype)
        dy = K.as_mat(dy)
        M = z.numel() // C
        rpb = K.bn_rows_per_block(M)
        nparts = (M + rpb - 1) // rpb
        relu_ss = ctx.relu_ss
        rmask = ctx.rmask
        ymask = y if (relu and relu_ss is None and rmask is None) else None
        comm_on = RT.comm is not None and (RT.comm.world_size > 1 or RT.comm.force)
        dz = torch.empty(lead + (C,), device=dev, dtype=dtype)
        stat_out, stat_in = ctx.links
        res_out, res_in = ctx.res_links
        # (the pointer test: the sums describe the tensor the next block's GEMM stored; any other dy - a hook, a second consumer that
        # autograd summed in - takes the two-pass path, which is still right: that tensor is gated already and gating is idempotent)
        fused_res = res_out is not None and res_out.sums is not None and res_out.dx_ptr == dy.data_ptr()
        if res_out is not None and not fused_res:
            res_out.sums = res_out.z = None
        dres = torch.empty(lead + (C,), device=dev, dtype=dtype) if (ctx.has_res and not fused_res) else None
        fused = stat_out is not None and stat_out.sums is not None
        if fused_res:
            # the next block's first data gradient added its identity gradient, gated the sum with this layer's ReLU bits and left
            # (sum g, sum g*z): no first pass, and dy IS the gradient of this block's identity path
            sums, R = res_out.sums, res_out.R
            res_out.sums = res_out.z = None
            scale = 1.0
            if comm_on:
                if res_out.synced:      # the GEMM's last block exchanged the sums: one row of global totals behind the R rows
                    sums, R = sums[R * 2 * C:(R + 1) * 2 * C], 1
                else:
                    RT.comm.all_reduce_sum(sums)
                scale = 1.0 / RT.comm.world_size
            K.bn_bwd_apply(dy, None, z, mi, bn.gamma.master(), sums, count, dz, None, None, sum_rows=-R,
                           dgamma=bn.gamma.grad(), dbeta=bn.beta.grad(), relu_mask=None, param_grad_scale=scale)
            bn.beta.done()
            bn.gamma.done()
            dres = dy.view(lead + (C,))
        elif fused:
            # the layer that consumed y already gated dy and left (sum g, sum g*z) in R replica rows (BnLink): no first pass
            sums, R = stat_out.sums, stat_out.R
            stat_out.sums = stat_out.z = None
            scale = 1.0
            if comm_on:
                if stat_out.synced:
                    sums, R = sums[R * 2 * C:(R + 1) * 2 * C], 1
                else:
                    RT.comm.all_reduce_sum(sums)
                scale = 1.0 / RT.comm.world_size
            K.bn_bwd_apply(dy, None, z, mi, bn.gamma.master(), sums, count, dz, dres, relu_ss, sum_rows=-R,
                           dgamma=bn.gamma.grad(), dbeta=bn.beta.grad(), relu_mask=None, param_grad_scale=scale)
            bn.beta.done()
            bn.gamma.done()
        elif BN_BWD_ATOMIC and dtype != torch.float32 and not RT.deterministic:
            # (sum g, sum g*xhat) accumulate atomically into R pre-zeroed [C][2] rows - each block parks its sums in LDS and adds them
            # as 256-byte runs - and the apply kernel adds the rows up itself and stores dbeta / dgamma: two launches, no slab, no
            # reduction kernel.  Under SyncBatchNorm the R rows are all-reduced in between; the totals are then global, and storing
            # them divided by the world size equals what DDP's averaging makes of the local sums, so no local-sum pass is needed.
            # Not in fp32: the parity mode keeps the ordered slab reduction below.
            R = stat_replicas(nparts, C)
            sync = RT.comm.fuse_ptr(2 * C) if comm_on else None
            scale = 1.0
            if sync is not None:
                # SyncBatchNorm: the first pass's last block adds its R rows up, exchanges the 2 C sums through the peer mailboxes and
                # stores the global totals behind the rows - no exchange launch between the two passes
                buf = RT.zeros(R * 2 * C + 2 * C + 8, dev)
                K.bn_bwd_partial(dy, ymask, z, mi, rpb, buf, relu_ss, replicas=R, relu_mask=rmask, pool=ctx.pool, stat_sync=sync, tail=True)
                sums, R = buf[R * 2 * C:(R + 1) * 2 * C], 1
                scale = 1.0 / RT.comm.world_size
            else:
                sums = RT.zeros(R * 2 * C, dev)
                K.bn_bwd_partial(dy, ymask, z, mi, rpb, sums, relu_ss, replicas=R, relu_mask=rmask, pool=ctx.pool)
                if comm_on:
                    RT.comm.all_reduce_sum(sums)
                    scale = 1.0 / RT.comm.world_size
            K.bn_bwd_apply(dy, ymask, z, mi, bn.gamma.master(), sums, count, dz, dres, relu_ss, sum_rows=R,
                           dgamma=bn.gamma.grad(), dbeta=bn.beta.grad(), relu_mask=rmask, param_grad_scale=scale, pool=ctx.pool)
            bn.beta.done()
            bn.gamma.done()
        elif ctx.pool is not None:
            raise RuntimeError("crog_amd: the pooled BatchNorm backward only exists on the atomic-statistics path (the mode changed between forward and backward)")
        elif BN_ATOMIC_STATS:
            # per-block slab -> one reduction launch -> apply kernel that stages the totals in LDS and (block 0) stores dbeta / dgamma
            partial = torch.empty(nparts, C, 2, device=dev, dtype=torch.float32)
            K.bn_bwd_partial(dy, ymask, z, mi, rpb, partial, relu_ss, relu_mask=rmask)
            fused = nparts <= FUSED_REDUCE_MAX_PARTS or RT.deterministic
            sums = torch.empty(2 * C, device=dev, dtype=torch.float32) if fused else RT.zeros(2 * C, dev)
            local_grads = fused or comm_on      # the parameter gradients are LOCAL sums: written before the all-reduce
            if fused:
                K.reduce_split(partial, nparts, C, sums, bn.beta.grad(), bn.gamma.grad())
            else:
                K.reduce_pairs(partial, nparts, C, sums, zeroed=True)
                if comm_on:
                    K.split_pairs(sums, C, bn.beta.grad(), bn.gamma.grad())
            if comm_on:
                RT.comm.all_reduce_sum(sums)
            K.bn_bwd_apply(dy, ymask, z, mi, bn.gamma.master(), sums, count, dz, dres, relu_ss, sum_rows=1,
                           dgamma=None if local_grads else bn.gamma.grad(), dbeta=None if local_grads else bn.beta.grad(), relu_mask=rmask)
            bn.beta.done()
            bn.gamma.done()
        else:
            partial = torch.empty(nparts, C, 2, device=dev, dtype=torch.float32)
            K.bn_bwd_partial(dy, ymask, z, mi, rpb, partial, relu_ss, relu_mask=rmask)
            sums = RT.zeros(2 * C, dev).view(C, 2) if nparts > FUSED_REDUCE_MAX_PARTS else torch.empty(C, 2, device=dev, dtype=torch.float32)
            # local sums: (sum g -> dbeta, sum g*xhat -> dgamma) and the pair vector for the second pass
            if nparts <= FUSED_REDUCE_MAX_PARTS:
                K.reduce_split(partial, nparts, C, sums, bn.beta.grad(), bn.gamma.grad())
            else:
                K.reduce_pairs(partial, nparts, C, sums, zeroed=True)
                K.split_pairs(sums, C, bn.beta.grad(), bn.gamma.grad())
            bn.beta.done()
            bn.gamma.done()
            if comm_on:
                RT.comm.all_reduce_sum(sums)
            K.bn_bwd_apply(dy, ymask, z, mi, bn.gamma.master(), sums, count, dz, dres, relu_ss, relu_mask=rmask)
        grad_slot, res_slot = ctx.slots
        if res_slot is not None and dres is not None:   # hand the identity's gradient to the block's first convolution
            res_slot.put(dres)
            dres = None
        extra = None
        if grad_slot is not None and ksize == 1 and ctx.x_needs:
            extra = grad_slot.take()
        dx = None
        if ksize == 0:
            dx = dz
        else:
            if wpad is not None:
                wt, woff = ctx.wt
                src_cols, dst_cols, rows = wpad
                wcols = dst_cols * (9 if ksize == 3 else 1)
                gscratch = torch.zeros(rows * dst_cols, device=dev, dtype=torch.float32)
                gt, goff = gscratch, 0
            else:
                wt, woff, wcols = w.w(dtype), w.off, w.cols
                gt, goff = w.G, w.off
            def wgrad():
                if ksize == "s":
                    wgrad_gemm(dt, K.B_NC, dz, x, gt, C, 32, M, C, 32, 32, c_off=goff, park=wpad is None)
                elif ksize == 1:
                    wgrad_gemm(dt, K.B_NC, dz, x, gt, C, cin, M, C, K.mat(x)[2], wcols, c_off=goff, park=wpad is None)
                else:
                    wgrad_gemm(dt, K.B_NC_IM2COL, dz, x, gt, C, 9 * cin, M, C, K.mat(x)[2], wcols, c_off=goff, conv=(lead[1], lead[2], cin),
                               park=wpad is None)
                if wpad is not None:  # strip the zero padding back out into the real gradient
                    K.add_pad2d(gscratch, dst_cols, w.G, src_cols, src_cols, rows, dst_off=w.off)
            # (a data gradient that will take the one-block-per-CU 256 x 256 tile: csrc/gemm.hip dispatch_shape)
            big = ksize == 3 and ctx.x_needs and wpad is None and dtype == torch.bfloat16 and cin % 256 == 0 and ((M + 255) // 256) * (cin // 256) >= 160
            RT.on_wgrad_stream(wgrad, dz, x, gt if wpad is not None else None, defer=big, tag="conv")
            # BnLink: this data gradient is the previous layer's dy -> its epilogue does that layer's first BatchNorm-backward pass
            bwd = {}
            # (the last two terms mirror crog_gemm's own eligibility test for bwd_z - operands the LDS-DMA path can address with 32-bit
            # byte offsets, C-ABI crog_gemm_supports_bwd_z - so that a launch that would be refused is never armed: the producer layer
            # has not committed to anything yet, `sums` stays None and it runs its own first pass)
            if (stat_in is not None and stat_in.z is not None and ctx.x_needs and extra is None and dtype == torch.bfloat16
                    and stat_in.C == cin and stat_in.M == M and cin % 8 == 0 and C % 8 == 0
                    and 2 * M * max(C, cin) < 2 ** 31):
                stat_in.R = stat_replicas(K.stat_tiles(M), cin)
                sync = RT.comm.fuse_ptr(2 * cin) if comm_on else None      # SyncBatchNorm: this GEMM's last block also exchanges the sums
                stat_in.synced = sync is not None
                stat_in.sums = RT.zeros(stat_in.R * 2 * cin + (2 * cin + 8 if sync is not None else 0), dev)
                bwd = dict(col_stats=stat_in.sums, stat_replicas=stat_in.R, bwd_z=stat_in.z, bwd_ss=stat_in.ss, stat_sync=sync)
            if (res_in is not None and res_in.z is not None and not bwd and ksize == 1 and ctx.x_needs and extra is not None
                    and dtype == torch.bfloat16 and res_in.C == cin and res_in.M == M and cin % 8 == 0 and C % 8 == 0
                    and 2 * M * max(C, cin) < 2 ** 31 and K.mat(extra)[2] % 2 == 0):
                res_in.R = stat_replicas(K.stat_tiles(M), cin)
                sync = RT.comm.fuse_ptr(2 * cin) if comm_on else None
                res_in.synced = sync is not None
                res_in.sums = RT.zeros(res_in.R * 2 * cin + (2 * cin + 8 if sync is not None else 0), dev)
                bwd = dict(col_stats=res_in.sums, stat_replicas=res_in.R, bwd_z=res_in.z, bwd_mask=res_in.mask, stat_sync=sync)
            if ksize == 1 and ctx.x_needs:
                dx = torch.empty(lead + (cin,), device=dev, dtype=dtype)
                if bwd and res_in is not None and res_in.sums is bwd.get("col_stats"):
                    res_in.dx_ptr = dx.data_ptr()
                if LIN_DGRAD_T and wpad is None and id(w.param) in w.store.lin_t:
                    # forward-shaped GEMM on the [Cin][Cout] copy of the weight: both operands K-contiguous
                    K.gemm(dt, K.A_KC, K.B_KC, dz, w.store.weights_t(dtype), dx, M, cin, C, C, C, cin, b_off=woff, R=extra,
                           ldr=K.mat(extra)[2] if extra is not None else 0, **bwd)
                else:
                    K.gemm(dt, K.A_KC, K.B_NC, dz, wt, dx, M, cin, C, C, wcols, cin, b_off=woff, R=extra,
                           ldr=K.mat(extra)[2] if extra is not None else 0, **bwd)
            elif ksize == 3 and ctx.x_needs:
                B, H, W = lead
                dx = torch.empty(lead + (cin,), device=dev, dtype=dtype)
                if DGRAD_T and wpad is None and cin % 8 == 0:
                    # data gradient as a forward-shaped implicit GEMM on the [Cin][flipped tap][Cout] copy of the weight
                    K.gemm(dt, K.A_IM2COL, K.B_KC, dz, w.store.weights_t(dtype), dx, M, cin, 9 * C, C, 9 * C, cin, b_off=woff, conv=(H, W, C), **bwd)
                else:
                    K.gemm(dt, K.A_IM2COL, K.B_NC_DGRAD, dz, wt, dx, M, cin, 9 * C, C, cin, cin, b_off=woff, conv=(H, W, C), **bwd)
            elif stat_in is not None:
                stat_in.sums = None
            RT.flush_wgrad()       # (defer_wgrad: the weight gradient parked above forks here, behind the data gradient just enqueued)
            w.done()
        if getattr(RT, "_dbg_dump", None) is not None and getattr(RT, "_fork_ctr", 0) in RT._dbg_dump_at:      # probe only (scripts/det_stress.py)
            # (_dbg_ref: keep REFERENCES - no launch is added to the step, whose timing is what the probe is about; they are compared after the pass)
            cl = (lambda t: t) if getattr(RT, "_dbg_ref", False) else (lambda t: t.clone())
            loc = locals()
            RT._dbg_dump.append((RT._fork_ctr, {k_: (cl(v_) if v_ is not None else None) for k_, v_ in
                                                dict(dy=dy, dz=dz, dx=dx, x=x, z=z, partial=loc.get("partial"), sums=loc.get("sums"), dres=loc.get("dres")).items()}))
        if ctx.dx_slot is not None and dx is not None:      # (see avgpool2: x's other consumer adds this gradient in its own epilogue)
            ctx.dx_slot.put(dx)
            dx = None
        return (dx, dres) + (None,) * 19


EVAL_BN_FOLD = True


def _conv_bn_act_eval(x, w: WRef, bn: BnBuffers, ksize, relu, res, out, wpad, dtype):
    """Inference form of conv -> BatchNorm (-> + identity) (-> ReLU) (validate_with_grasp runs under model.eval() + torch.no_grad(),
    crog_engine.py:133-135): with running statistics BatchNorm is an affine map per output channel, so it is folded into the weights
    (crog_bn_fold_weights: fp32 master weights x scale -> compute dtype, one tiny launch per layer) and the shift, the residual and
    the ReLU go into the GEMM epilogue - the pre-normalisation map z is never written or read (training needs z for the batch
    statistics and the backward pass; inference does not)."""
    dev = x.device
    dt = K.dcode(dtype)
    C = bn.gamma.rows * bn.gamma.cols
    if ksize == "s":
        B, _, Hi, Wi = x.shape
        H, W = Hi // 2, Wi // 2
        lead, M, cin = (B, H, W), B * (Hi // 2) * (Wi // 2), 27
    else:
        lead = tuple(x.shape[:-1])
        M, cin, _ = K.mat(x)
        if ksize == 3:
            H, W = x.shape[1], x.shape[2]
    if wpad is not None:
        src_cols, dst_cols, rows = wpad
    else:
        rows, src_cols, dst_cols = w.rows, w.cols, w.cols
    wf = torch.empty(rows * dst_cols, device=dev, dtype=dtype)
    shift = torch.empty(C, device=dev, dtype=torch.float32)
    K.bn_fold_weights(w.P, w.off, src_cols, rows // C, bn.gamma.master(), bn.beta.master(), bn.running_mean, bn.running_var, bn.eps, wf,
                      dst_cols, rows, shift)
    wcols = dst_cols * (rows // C)
    y = _dest(out) if out is not None else torch.empty(lead + (C,), device=dev, dtype=dtype)
    ldy = K.mat(y)[2]
    act = (K.ACT_RELU_POST if res is not None else K.ACT_RELU) if relu else K.ACT_NONE
    ldr = K.mat(res)[2] if res is not None else 0
    if ksize == "s":
        patches = torch.empty(M, 32, device=dev, dtype=dtype)
        K.stem_im2col(x, patches)
        K.gemm(dt, K.A_KC, K.B_KC, patches, wf, y, M, C, 32, 32, 32, ldy, bias=shift, act=act, R=res, ldr=ldr)
    elif ksize == 1:
        K.gemm(dt, K.A_KC, K.B_KC, x, wf, y, M, C, cin, K.mat(x)[2], wcols, ldy, bias=shift, act=act, R=res, ldr=ldr)
    else:
        K.gemm(dt, K.A_IM2COL, K.B_KC, x, wf, y, M, C, 9 * cin, K.mat(x)[2], wcols, ldy, conv=(H, W, cin), bias=shift, act=act, R=res, ldr=ldr)
    return y


def conv_bn_act(x, w: Optional[WRef], bn: BnBuffers, *, ksize, relu=True, res=None, training=True, out=None, wpad=None, dtype=None,
                grad_slot=None, res_slot=None, stat_out=None, stat_in=None, dx_slot=None, res_out=None, res_in=None, pool=False):
    """stat_out / stat_in: a BnLink shared by two layers with y_L -> x_{L+1} and no other consumer of y_L (see BnLink).
    res_out / res_in: a BnLink between a residual layer and the first convolution of the block its output feeds (BN_RES_FUSED).
    pool: follow the layer by the 2 x 2 average pooling (clip.py:49-50, 213-214) - inside the BatchNorm passes on the bf16 training path
    (POOL_FUSED: the full-resolution activation and its gradient never reach HBM), as a separate avgpool2 otherwise."""
    dtype = dtype if dtype is not None else x.dtype
    wp = w.param if w is not None else None
    if EVAL_BN_FOLD and not training and ksize != 0 and not torch.is_grad_enabled():
        y = _conv_bn_act_eval(x, w, bn, ksize, relu, res, out, wpad, dtype)
        return avgpool2(y) if pool else y
    # (odd maps - 400 x 400 inputs reach 25 x 25 in layer4 - take the separate avgpool2, which floors like nn.AvgPool2d)
    even = ksize == "s" or (x.dim() == 4 and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0)
    if ksize == "s":
        even = (x.shape[2] // 2) % 2 == 0 and (x.shape[3] // 2) % 2 == 0
    fuse = (pool and POOL_FUSED and even and training and relu and res is None and out is None and ksize != 0 and dtype == torch.bfloat16
            and BN_ATOMIC_STATS and BN_BWD_ATOMIC and not RT.deterministic and stat_out is None and res_out is None)
    y = ConvBnAct.apply(x, res, wp, bn.gamma.param, bn.beta.param, w, bn, ksize, relu, training, out, wpad, dtype, grad_slot, res_slot,
                        stat_out, stat_in, dx_slot, res_out, res_in, fuse)
    return avgpool2(y) if (pool and not fuse) else y


# ------------------------------------------------------------------------------------------------
# Linear with fused bias / activation / residual epilogue
# ------------------------------------------------------------------------------------------------
class LinearFn(Function):
    """y = act(x W^T + b) + res.  act in {none, relu, tanh}.  nn.Linear call sites: clip.py:249-251 (c_fc/c_proj),
    layers.py:298-301 (ffn), proj.vis.4 1x1 conv with bias (layers.py:58); SSG's bias convolutions on im2col / 1x1 rows
    (ssg.py:123-133,163-164,178,188-191,230).  An output width that is not a multiple of the 16-byte vector (SSG's 12 box
    channels) lives in a padded buffer; the Function hands out the [.., :N] view."""

    @staticmethod
    def forward(ctx, x, res, _wp, _bp, w: WRef, b: Optional[WRef], act, out, grad_gated=False, res_slot=None):
        ctx.res_slot = res_slot          # the residual's gradient goes to the LayerNorm backward of the same tensor (GradSlot), not to autograd
        ctx.grad_gated = grad_gated      # the consumer's backward already gated dy by y > 0 (layernorm(relu_in=True)): no activation-backward pass
        M, Kd, _ = K.mat(x)
        N = w.rows
        Np = _pad(N, _vec(_cdt(x)))
        if out is not None:
            y = _dest(out)
        elif Np != N:
            y = torch.zeros(tuple(x.shape[:-1]) + (Np,), device=x.device, dtype=x.dtype)[..., :N]
        else:
            y = torch.empty(tuple(x.shape[:-1]) + (N,), device=x.device, dtype=x.dtype)
        lin_fwd(x, w, y, bias=b, act=act, res=res)
        ctx.cfg = (w, b, act, res is not None)
        ctx.save_for_backward(x, y if act in (K.ACT_RELU, K.ACT_TANH) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        w, b, act, has_res = ctx.cfg
        x, y = ctx.saved_tensors
        N = w.rows
        Np = _pad(N, _vec(_cdt(x)))
        if Np != N:   # ragged width: re-pad the incoming gradient so that its rows are 16-byte vectors
            dyp = torch.zeros(tuple(dy.shape[:-1]) + (Np,), device=dy.device, dtype=dy.dtype)
            dyp[..., :N].copy_(dy)
            dy = dyp[..., :N]
        else:
            dy = K.as_mat(dy)
        dres = dy if has_res else None
        g = dy
        if act in (K.ACT_RELU, K.ACT_TANH) and not (ctx.grad_gated and act == K.ACT_RELU):
            if has_res:
                raise NotImplementedError("activation + residual epilogue backward")
            if Np != N:
                g = torch.zeros(tuple(dy.shape[:-1]) + (Np,), device=dy.device, dtype=dy.dtype)
                K.act_bwd(dyp, y_full(y, Np), g, 0 if act == K.ACT_RELU else 2)
                g = g[..., :N]
            else:
                g = torch.empty(dy.shape, device=dy.device, dtype=dy.dtype)
                K.act_bwd(dy, y, g, 0 if act == K.ACT_RELU else 2)
        def wgrad():
            lin_wgrad(g, x, w, bias=b)
        if has_res and g is dy and ctx.res_slot is None:
            wgrad()   # dy is handed on as the residual's gradient and autograd may accumulate into it IN PLACE: keep the read ordered
        else:
            RT.on_wgrad_stream(wgrad, g, x, tag="linear")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x.shape, device=x.device, dtype=x.dtype)
            lin_dgrad(g, w, dx)
        # announced AFTER the data gradient that reads the weight is enqueued: an optimizer chunk / DDP bucket this completes may be
        # stepped as soon as the NEXT one completes (optim.py: overlap_backward), which can be the bias a line below
        w.done()
        if b is not None:
            b.done()
        if ctx.res_slot is not None and dres is not None:
            ctx.res_slot.put(dres)
            dres = None
        return dx, dres, None, None, None, None, None, None, None, None


def y_full(y: torch.Tensor, Np: int) -> torch.Tensor:
    """The padded buffer behind a [.., :N] view handed out by LinearFn.forward."""
    return y.as_strided(tuple(y.shape[:-1]) + (Np,), y.stride(), y.storage_offset())


def linear(x, w: WRef, b: Optional[WRef] = None, *, act=K.ACT_NONE, res=None, out=None, grad_gated=False, res_slot=None):
    """grad_gated (act = ReLU only): the ONLY consumer of the output is a layernorm(relu_in=True), whose backward hands back the gradient
    of the ReLU's input.  res_slot: `res` is also the input of a layernorm(add_slot=...) whose backward adds the residual's gradient."""
    return LinearFn.apply(x, res, w.param, b.param if b is not None else None, w, b, act, out, bool(grad_gated), res_slot)


class QuickGeluFn(Function):
    """x * sigmoid(1.702 x)  (clip.py:234-236)."""

    @staticmethod
    def forward(ctx, u):
        a = torch.empty_like(u)
        K.quickgelu_fwd(u, a)
        ctx.save_for_backward(u)
        return a

    @staticmethod
    def backward(ctx, da):
        (u,) = ctx.saved_tensors
        du = torch.empty_like(u)
        K.act_bwd(K.as_mat(da), u, du, 1)
        return du


def quickgelu(u):
    return QuickGeluFn.apply(u)


# ------------------------------------------------------------------------------------------------
# LayerNorm with fused dropout / residual / positional add
# ------------------------------------------------------------------------------------------------
class LayerNormFn(Function):
    """out = res + dropout_out(LN(dropout_in(x)));  out2 = out + pos (optional second output).
    clip.py:226-231; layers.py:288-305,313-339.
    GradSlots for the decoder's `vis -> norm(vis)` / `vis -> ... + vis` pairs (layers.py:313-338): the norm that consumes the identity
    (`res_slot`) leaves the residual's gradient in the slot instead of returning it, and the norm applied to the same tensor
    (`add_slot`, whose backward runs later: its output feeds the other one) adds it to its dx inside crog_ln_bwd."""

    @staticmethod
    def forward(ctx, x, res, _gp, _bp, gamma: WRef, beta: WRef, eps, pos, p_in, p_out, want_out2, res_slot=None, add_slot=None, relu_in=False):
        ctx.set_materialize_grads(False)
        ctx.relu_in = relu_in      # x is the output of a ReLU whose backward this norm's backward does (crog_ln_bwd_relu; linear(grad_gated=True))
        M, C, _ = K.mat(x)
        out = torch.empty(x.shape, device=x.device, dtype=x.dtype)
        out2 = torch.empty(x.shape, device=x.device, dtype=x.dtype) if want_out2 else None
        stats = torch.empty(M, 2, device=x.device, dtype=torch.float32)
        seed_in = RT.next_seed() if p_in > 0 else 0
        seed_out = RT.next_seed() if p_out > 0 else 0
        K.ln_fwd(x, gamma.master(), beta.master(), eps, out, stats, res=res, out2=out2, pos=pos, p_in=p_in, seed_in=seed_in, p_out=p_out,
                 seed_out=seed_out)
        ctx.cfg = (gamma, beta, p_in, seed_in, p_out, seed_out, res is not None, want_out2)
        ctx.slots = (res_slot, add_slot)
        ctx.save_for_backward(x, stats)
        if want_out2:
            return out, out2
        return out

    @staticmethod
    def backward(ctx, dout, dout2=None):
        gamma, beta, p_in, seed_in, p_out, seed_out, has_res, want_out2 = ctx.cfg
        x, stats = ctx.saved_tensors
        M, C, _ = K.mat(x)
        res_slot, add_slot = ctx.slots
        parked = add_slot.take() if add_slot is not None else None
        if dout is None and dout2 is None:
            return (parked,) + (None,) * 13      # no gradient through the norm itself: x's gradient is what the residual branch parked
        if dout is None:
            dout, dout2 = dout2, None
        dout = K.as_mat(dout)
        dout2 = K.as_mat(dout2) if dout2 is not None else None
        dxadd = K.as_mat(parked) if parked is not None else None
        dx = torch.empty(x.shape, device=x.device, dtype=x.dtype)
        rpb = K.ln_bwd_rows_per_block(M)
        nb = (M + rpb - 1) // rpb
        if LN_BWD_ATOMIC and x.dtype != torch.float32 and not RT.deterministic:
            # the blocks add their (dgamma, dbeta) sums straight into the gradient vectors: no slab, no reduction launch (85 launches
            # per CROG step).  fp32, the parity mode, keeps the ordered reduction (bit-reproducible run to run)
            K.ln_bwd(dout, dout2, x, gamma.master(), stats, dx, None, rpb, p_in=p_in, seed_in=seed_in, p_out=p_out, seed_out=seed_out,
                     dgamma=gamma.grad(), dbeta=beta.grad(), dxadd=dxadd, relu_in=ctx.relu_in)
        else:
            partial = torch.empty(nb, C, 2, device=x.device, dtype=torch.float32)
            K.ln_bwd(dout, dout2, x, gamma.master(), stats, dx, partial, rpb, p_in=p_in, seed_in=seed_in, p_out=p_out, seed_out=seed_out,
                     dxadd=dxadd, relu_in=ctx.relu_in)
            # the parameter gradients are consumed by the optimizer only: their reduction leaves the dependency chain for the
            # weight-gradient stream (41 small launches per CROG step that sat between dependent kernels of the main stream)
            if LN_REDUCE_SIDE:
                RT.on_wgrad_stream(lambda: K.reduce_split(partial, nb, C, None, gamma.grad(), beta.grad()), partial, tag="ln")
            else:
                K.reduce_split(partial, nb, C, None, gamma.grad(), beta.grad())
        gamma.done()
        beta.done()
        dres = None
        if has_res:
            if dout2 is None:
                dres = dout
            else:
                dres = torch.empty(x.shape, device=x.device, dtype=x.dtype)
                K.add_rows(dout, dout2, dres)
            if res_slot is not None:
                res_slot.put(dres)
                dres = None
        return (dx, dres) + (None,) * 12


def layernorm(x, gamma: WRef, beta: WRef, *, eps=1e-5, res=None, pos=None, p_in=0.0, p_out=0.0, want_out2=False, res_slot=None, add_slot=None,
              relu_in=False):
    """relu_in: x is `linear(..., act=ReLU, grad_gated=True)` and has no other consumer - this norm's backward also does that ReLU's."""
    return LayerNormFn.apply(x, res, gamma.param, beta.param, gamma, beta, eps, pos, float(p_in), float(p_out), want_out2, res_slot, add_slot,
                             bool(relu_in))


# ------------------------------------------------------------------------------------------------
# Multi-head attention (projections + scores + masked softmax + PV + out-proj [+ residual])
# ------------------------------------------------------------------------------------------------
class JointDone:
    """done_joint across autograd nodes: a packed parameter whose row blocks are written by DIFFERENT backward nodes (q in MhaFn, k / v in
    KvProjFn) is announced by whichever node's last block arrives, once `n` blocks have."""

    def __init__(self, n: int):
        self.left, self.first = n, None

    def add(self, ref):
        if ref is None:
            return
        if self.first is None:
            self.first = ref
        ref.done(self._arrive)

    def _arrive(self):
        self.left -= 1
        if self.left == 0:
            self.first._done_now()


class KvProjFn(Function):
    """The key / value projections of a cross-attention (layers.py:329-332: nn.MultiheadAttention's packed in_proj rows E..3E) as a node of their
    own: they depend on the text side alone, so CROG.forward runs them on the text stream behind the tower, and autograd replays their backward
    (two 80-block data gradients per decoder layer, latency-bound) there instead of on the main chain.  -> [B*Lk, 2E]: K | V."""

    @staticmethod
    def forward(ctx, xk, xv, _p1, _p2, wk: WRef, wv: WRef, bk, bv, jw, jb):
        E = wk.rows
        kv = torch.empty(xk.shape[0], 2 * E, device=xk.device, dtype=xk.dtype)
        lin_fwd(xk, wk, kv, bias=bk, c_off=0, ldc=2 * E)
        lin_fwd(xv, wv, kv, bias=bv, c_off=E, ldc=2 * E)
        ctx.cfg = (wk, wv, bk, bv, jw, jb, E)
        ctx.save_for_backward(xk, xv)
        return kv

    @staticmethod
    def backward(ctx, dkv):
        wk, wv, bk, bv, jw, jb, E = ctx.cfg
        xk, xv = ctx.saved_tensors
        dkv = dkv.contiguous()
        out = []
        for x_, w_, b_, col in ((xk, wk, bk, 0), (xv, wv, bv, E)):
            def wgrad_proj(x_=x_, w_=w_, b_=b_, col=col):
                lin_wgrad(dkv, x_, w_, a_off=col, lda=2 * E, N=E, bias=b_)
            RT.on_wgrad_stream(wgrad_proj, dkv, x_, tag="mha")
            dx = torch.empty(x_.shape, device=x_.device, dtype=x_.dtype)
            lin_dgrad(dkv, w_, dx, a_off=col, lda=2 * E, N=E)
            out.append(dx)
        for r in (wk, wv):
            jw.add(r)
        for r in (bk, bv):
            jb.add(r)
        return out[0], out[1], None, None, None, None, None, None, None, None


def kv_proj(xk, xv, wk: WRef, wv: WRef, bk, bv):
    """-> (K | V buffer, (joint announcement of the packed weight, of the packed bias)) for mha(..., kv=)."""
    jw, jb = JointDone(3), JointDone(3)
    kv = KvProjFn.apply(xk, xv, wk.param, bk.param if bk is not None else None, wk, wv, bk, bv, jw, jb)
    return kv, (jw, jb)


class MhaFn(Function):
    """F.multi_head_attention_forward restated on row matrices (clip.py:119-139,246-260; layers.py:291-296,324,329-332).
    xq: [B*Lq, E]; xk, xv: [B*Lk, E].  Weight rows come as WRefs so packed in_proj slices and separate q/k/v
    projections share one code path.  Scores are materialised per (batch, head) in the compute dtype with the row
    padded to a multiple of 8 so that P feeds the P.V MFMA GEMM directly."""

    @staticmethod
    def forward(ctx, xq, xk, xv, res, kv, *args):
        """kv: the K | V projections [B*Lk, 2E] when the caller has computed them already (KvProjFn, with `joint` = its JointDone pair):
        this node then projects the queries only and hands dK | dV back as kv's gradient."""
        (wq, wk, wv, bq, bk, bv, wo, bo, B, heads, causal, kpm, p_drop, res_slot, joint) = args[-15:]
        ctx.res_slot = res_slot
        ctx.joint = joint if kv is not None else None
        dev, dtype = xq.device, xq.dtype
        dt = K.dcode(dtype)
        E = wq.rows
        dh = E // heads
        Lq, Lk = xq.shape[0] // B, xk.shape[0] // B
        same_qk, same_kv = xq is xk, xk is xv
        # ---- projections ----
        self_attn = Lq == Lk and (same_qk or same_kv)
        if Lq == Lk:
            qkv = torch.empty(B * Lq, 3 * E, device=dev, dtype=dtype)
            qb, kb, vb = (qkv, 0, 3 * E), (qkv, E, 3 * E), (qkv, 2 * E, 3 * E)
        else:
            qbuf = torch.empty(B * Lq, E, device=dev, dtype=dtype)
            kvbuf = kv if kv is not None else torch.empty(B * Lk, 2 * E, device=dev, dtype=dtype)
            qb, kb, vb = (qbuf, 0, E), (kvbuf, 0, 2 * E), (kvbuf, E, 2 * E)
        if kv is not None and Lq == Lk:
            raise ValueError("mha(kv=): precomputed key / value projections are for cross-attention (Lq != Lk)")
        jobs = [[xq, wq, bq, qb]] if kv is not None else [[xq, wq, bq, qb], [xk, wk, bk, kb], [xv, wv, bv, vb]]
        merged = []
        for j in jobs:
            if merged:
                p = merged[-1]
                contiguous_w = p[1].param is j[1].param and p[1].off + p[1].rows * p[1].cols == j[1].off
                contiguous_b = (p[2] is None and j[2] is None) or (p[2] is not None and j[2] is not None and p[2].param is j[2].param
                                                                    and p[2].off + p[2].rows == j[2].off)
                contiguous_o = p[3][0] is j[3][0] and p[3][1] + p[1].rows == j[3][1]
                if p[0] is j[0] and contiguous_w and contiguous_b and contiguous_o:
                    p[1] = _merge_w(p[1], j[1])
                    p[2] = _merge_w(p[2], j[2]) if p[2] is not None else None
                    continue
            merged.append(list(j))
        for x_, w_, b_, (buf, col, ld) in merged:
            lin_fwd(x_, w_, buf, bias=b_, c_off=col, ldc=ld)
        Lkp = _pad(Lk, 8)
        scale = dh ** -0.5
        # ---- fused attention: no score matrix in HBM (unmasked bf16, head_dim 64, long key rows) ----
        # (... and - FLASH_CAUSAL - the text tower's causal 20-token blocks: one launch instead of three, two instead of five backward, on a
        # stream whose ~150 backward launches are latency-bound)
        cross = FLASH_CROSS and not causal and Lq >= FLASH_CROSS_MIN_QUERIES and (kpm is not None or Lk < FLASH_MIN_KEYS)
        if FLASH_ATTN and dtype == torch.bfloat16 and dh == 64 and (cross or (kpm is None and (
                (not causal and Lk >= FLASH_MIN_KEYS) or (causal and FLASH_CAUSAL and Lq == Lk)))):
            seed = RT.next_seed() if p_drop > 0 else 0
            O = torch.empty(B * Lq, E, device=dev, dtype=dtype)
            lse = torch.empty(B * heads * Lq, device=dev, dtype=torch.float32)
            keep = torch.empty(K.flash_keep_words(B, heads, Lq, Lk), device=dev, dtype=torch.int32) if (p_drop > 0 and FLASH_KEEP) else None
            if kpm is not None and kpm.dtype not in (torch.bool, torch.uint8):
                raise TypeError("key_padding_mask must be a bool / uint8 tensor")
            K.flash_attn_fwd(qb, kb, vb, (O, 0, E), lse, B, heads, Lq, Lk, dh, scale, p_drop, seed, Lkp, causal=causal, keep=keep, kpm=kpm)
            ctx.causal = bool(causal)
            ctx.kpm = kpm
            out = torch.empty(B * Lq, wo.rows, device=dev, dtype=dtype)
            lin_fwd(O, wo, out, bias=bo, res=res)
            ctx.cfg = (merged, qb, kb, vb, wo, bo, B, heads, Lq, Lk, Lkp, E, dh, scale, p_drop, seed, res is not None, same_qk, same_kv)
            ctx.flash = True
            ctx.save_for_backward(xq, xk, xv, lse, keep, O)
            return out
        ctx.flash = False
        # ---- scores, softmax ----
        S = torch.empty(B * heads, Lq, Lkp, device=dev, dtype=dtype)
        K.gemm(dt, K.A_KC, K.B_KC, qb[0], kb[0], S, Lq, Lk, dh, qb[2], kb[2], Lkp, batch=B * heads, batch_inner=heads,
               sA=(Lq * qb[2], dh), sB=(Lk * kb[2], dh), sC=(heads * Lq * Lkp, Lq * Lkp), a_off=qb[1], b_off=kb[1], alpha=scale)
        Pd = None
        seed = 0
        if p_drop > 0:
            Pd = torch.empty_like(S)
            seed = RT.next_seed()
        K.softmax_fwd(S, B * heads * Lq, Lq, Lk, Lkp, heads, causal, kpm, Pd, p_drop, seed)
        Pm = Pd if Pd is not None else S
        # ---- O = P V, out projection ----
        O = torch.empty(B * Lq, E, device=dev, dtype=dtype)
        K.gemm(dt, K.A_KC, K.B_NC, Pm, vb[0], O, Lq, dh, Lk, Lkp, vb[2], E, batch=B * heads, batch_inner=heads,
               sA=(heads * Lq * Lkp, Lq * Lkp), sB=(Lk * vb[2], dh), sC=(Lq * E, dh), b_off=vb[1])
        out = torch.empty(B * Lq, wo.rows, device=dev, dtype=dtype)
        lin_fwd(O, wo, out, bias=bo, res=res)
        ctx.cfg = (merged, qb, kb, vb, wo, bo, B, heads, Lq, Lk, Lkp, E, dh, scale, p_drop, seed, res is not None, same_qk, same_kv)
        ctx.save_for_backward(xq, xk, xv, S, Pd, O)
        return out

    @staticmethod
    def backward(ctx, dout):
        merged, qb, kb, vb, wo, bo, B, heads, Lq, Lk, Lkp, E, dh, scale, p_drop, seed, has_res, same_qk, same_kv = ctx.cfg
        xq, xk, xv, S, Pd, O = ctx.saved_tensors
        dev, dtype = dout.device, xq.dtype
        dt = K.dcode(dtype)
        dout = K.as_mat(dout)
        # out projection
        def wgrad_out():
            lin_wgrad(dout, O, wo, bias=bo)
        if has_res and ctx.res_slot is None:
            wgrad_out()   # dout doubles as the residual's gradient (possible in-place accumulation by autograd): stay on this stream
        else:
            RT.on_wgrad_stream(wgrad_out, dout, O, tag="mha")
        dO = torch.empty(B * Lq, E, device=dev, dtype=dtype)
        lin_dgrad(dout, wo, dO)
        wo.done()      # (after the data gradient that reads it: see LinearFn.backward)
        if bo is not None:
            bo.done()
        # gradient buffers mirror the projection buffers
        if qb[0] is kb[0]:
            dqkv = torch.empty_like(qb[0])
            dqb, dkb, dvb = (dqkv, 0, 3 * E), (dqkv, E, 3 * E), (dqkv, 2 * E, 3 * E)
        else:
            dq_ = torch.empty_like(qb[0])
            dkv_ = torch.empty_like(kb[0])
            dqb, dkb, dvb = (dq_, 0, E), (dkv_, 0, 2 * E), (dkv_, E, 2 * E)
        if ctx.flash:
            lse = S
            D = torch.empty_like(lse)
            K.flash_attn_bwd(qb, kb, vb, (O, 0, E), (dO, 0, E), lse, D, dqb, dkb, dvb, B, heads, Lq, Lk, dh, scale, p_drop, seed, Lkp,
                             causal=ctx.causal, keep=Pd, kpm=ctx.kpm)      # (Pd: the forward's keep-bit map in the fused case)
            return MhaFn._proj_backward(ctx, merged, qb, dqb, dkb, xq, xk, xv, dout, has_res, same_qk, same_kv, dev, dtype)
        Pm = Pd if Pd is not None else S
        bh = B * heads
        # dV[b,h] = Pm^T dO
        K.gemm(dt, K.A_MC, K.B_NC, Pm, dO, dvb[0], Lk, dh, Lq, Lkp, E, dvb[2], batch=bh, batch_inner=heads,
               sA=(heads * Lq * Lkp, Lq * Lkp), sB=(Lq * E, dh), sC=(Lk * dvb[2], dh), c_off=dvb[1])
        # dPd = dO V^T
        dP = torch.empty(bh, Lq, Lkp, device=dev, dtype=dtype)
        K.gemm(dt, K.A_KC, K.B_KC, dO, vb[0], dP, Lq, Lk, dh, E, vb[2], Lkp, batch=bh, batch_inner=heads,
               sA=(Lq * E, dh), sB=(Lk * vb[2], dh), sC=(heads * Lq * Lkp, Lq * Lkp), b_off=vb[1])
        K.softmax_bwd(S, dP, bh * Lq, Lk, Lkp, p_drop, seed)
        # dQ = scale * dS K ;  dK = scale * dS^T Q
        K.gemm(dt, K.A_KC, K.B_NC, dP, kb[0], dqb[0], Lq, dh, Lk, Lkp, kb[2], dqb[2], batch=bh, batch_inner=heads,
               sA=(heads * Lq * Lkp, Lq * Lkp), sB=(Lk * kb[2], dh), sC=(Lq * dqb[2], dh), b_off=kb[1], c_off=dqb[1], alpha=scale)
        K.gemm(dt, K.A_MC, K.B_NC, dP, qb[0], dkb[0], Lk, dh, Lq, Lkp, qb[2], dkb[2], batch=bh, batch_inner=heads,
               sA=(heads * Lq * Lkp, Lq * Lkp), sB=(Lq * qb[2], dh), sC=(Lk * dkb[2], dh), b_off=qb[1], c_off=dkb[1], alpha=scale)
        return MhaFn._proj_backward(ctx, merged, qb, dqb, dkb, xq, xk, xv, dout, has_res, same_qk, same_kv, dev, dtype)

    @staticmethod
    def _proj_backward(ctx, merged, qb, dqb, dkb, xq, xk, xv, dout, has_res, same_qk, same_kv, dev, dtype):
        """Backward of the q/k/v input projections from the gradient buffers that mirror the projection buffers."""
        grads = {}
        for x_, w_, b_, (buf, col, ld) in merged:
            dbuf = dqb[0] if buf is qb[0] else dkb[0]
            def wgrad_proj(dbuf=dbuf, x_=x_, w_=w_, b_=b_, col=col, ld=ld):
                lin_wgrad(dbuf, x_, w_, a_off=col, lda=ld, N=w_.rows, bias=b_)
            RT.on_wgrad_stream(wgrad_proj, dbuf, x_, tag="mha")
            prev = grads.get(id(x_))
            dx = prev if prev is not None else torch.empty(x_.shape, device=dev, dtype=dtype)
            lin_dgrad(dbuf, w_, dx, accumulate_into=prev, a_off=col, lda=ld, N=w_.rows)
            grads[id(x_)] = dx
        # a packed in_proj parameter is written by up to three row-block GEMMs: it is ready (DDP bucket bookkeeping) only once
        # ALL of them are enqueued, so the marks come after the loop, one per parameter
        dkv = None
        if ctx.joint is not None:
            # the key / value projections belong to KvProjFn: dK | dV goes back as the gradient of its output, and the packed parameters are
            # announced by whichever of the two nodes enqueues its last block
            jw, jb = ctx.joint
            for _, w_, b_, _ in merged:
                jw.add(w_)
                jb.add(b_)
            dkv = dkb[0]
        else:
            done_joint([w_ for _, w_, _, _ in merged] + [b_ for _, _, b_, _ in merged])
        dxq = grads.get(id(xq))
        dxk = None if (same_qk or dkv is not None) else grads.get(id(xk))
        dxv = None if (same_kv or xv is xq or dkv is not None) else grads.get(id(xv))
        dres = dout if has_res else None
        if ctx.res_slot is not None and dres is not None:      # (LinearFn: the LayerNorm backward of the same tensor adds it)
            ctx.res_slot.put(dres)
            dres = None
        n_extra = 15 + 8
        return (dxq, dxk, dxv, dres, dkv) + (None,) * n_extra


def _merge_w(a: WRef, b: WRef) -> WRef:
    m = WRef.__new__(WRef)
    m.store, m.param, m.off, m.cols = a.store, a.param, a.off, a.cols
    m.rows = a.rows + b.rows
    return m


def mha(xq, xk, xv, wq: WRef, wk: WRef, wv: WRef, bq, bk, bv, wo: WRef, bo, *, B, heads, causal=False, kpm=None, p_drop=0.0, res=None,
        res_slot=None, kv=None):
    """kv: (K | V buffer, joint) from kv_proj(xk, xv, ...) when the key / value projections were computed elsewhere (another stream)."""
    params = [r.param for r in (wq, wk, wv, bq, bk, bv, wo, bo) if r is not None]
    params = params + [None] * (8 - len(params))
    kvbuf, joint = kv if kv is not None else (None, None)
    return MhaFn.apply(xq, xk, xv, res, kvbuf, *params, wq, wk, wv, bq, bk, bv, wo, bo, B, heads, causal, kpm, float(p_drop), res_slot, joint)


# ------------------------------------------------------------------------------------------------
# pooling / upsampling / joins / broadcasts
# ------------------------------------------------------------------------------------------------
class AvgPool2Fn(Function):
    @staticmethod
    def forward(ctx, x, out, grad_slot=None, add_slot=None):
        B, H, W, C = x.shape
        y = _dest(out) if out is not None else torch.empty(B, H // 2, W // 2, C, device=x.device, dtype=x.dtype)
        K.avgpool2_fwd(x, y)
        ctx.shape = x.shape
        ctx.grad_slot, ctx.add_slot = grad_slot, add_slot
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = torch.empty(ctx.shape, device=dy.device, dtype=dy.dtype)
        parked = ctx.add_slot.take() if ctx.add_slot is not None else None
        K.avgpool2_bwd(K.as_mat(dy), dx, add=parked)
        if ctx.grad_slot is not None:      # the other consumer of x adds this branch's gradient in its data-gradient epilogue (GradSlot)
            ctx.grad_slot.put(dx)
            return None, None, None, None
        return dx, None, None, None


def avgpool2(x, out=None, grad_slot=None, add_slot=None):
    """grad_slot: hand the gradient w.r.t. x to a GradSlot instead of autograd (the caller vouches that the slot's consumer - another
    consumer of the same x - runs its backward AFTER this op: Bottleneck's downsample branch, created after conv1 / conv2).
    add_slot: a third consumer of x whose backward ran BEFORE this one (the neck, for the outputs of layer2 / layer3) left its gradient
    there; the pool backward adds it while it writes dx."""
    return AvgPool2Fn.apply(x, out, grad_slot, add_slot)


class Upsample2Fn(Function):
    @staticmethod
    def forward(ctx, x, out):
        B, H, W, C = x.shape
        y = _dest(out) if out is not None else torch.empty(B, 2 * H, 2 * W, C, device=x.device, dtype=x.dtype)
        K.upsample2_fwd(x, y)
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = torch.empty(ctx.shape, device=dy.device, dtype=dy.dtype)
        K.upsample2_bwd(K.as_mat(dy), dx)
        return dx, None


def upsample2(x, out=None):
    return Upsample2Fn.apply(x, out)


class JoinFn(Function):
    """Channel concatenation without a copy: the producers already wrote into channel slices of `buf`
    (torch.cat call sites layers.py:38,383,387,394).  Backward hands each producer its slice of the gradient."""

    @staticmethod
    def forward(ctx, buf: "OutRef", widths, *parts):
        ctx.widths = widths
        return buf.t.view(buf.t.shape)

    @staticmethod
    def backward(ctx, dbuf):
        outs, c = [], 0
        for w in ctx.widths:
            outs.append(dbuf[..., c:c + w])
            c += w
        return (None, None) + tuple(outs)


def join(buf: torch.Tensor, parts):
    return JoinFn.apply(OutRef(buf), tuple(p.shape[-1] for p in parts), *parts)


class AddReluFn(Function):
    """relu(a + b)   (attnpool tail clip.py:141-142)."""

    @staticmethod
    def forward(ctx, a, b):
        C = a.shape[-1]
        ident = torch.zeros(C, 2, device=a.device, dtype=torch.float32)
        ident[:, 0] = 1.0
        y = torch.empty_like(a)
        K.bn_apply(a, ident, b, True, y)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        g = torch.empty_like(y)
        K.act_bwd(K.as_mat(dy), y, g, 0)
        return g, g


def add_relu(a, b):
    return AddReluFn.apply(a, b)


class MulBcastFn(Function):
    """z[b,p,:] = x[b,p,:] * s[b,:]   (layers.py:379)."""

    @staticmethod
    def forward(ctx, x, s):
        B = s.shape[0]
        P = x.numel() // (B * x.shape[-1])
        z = torch.empty_like(x)
        K.mul_bcast_fwd(x, s, z, B, P)
        ctx.save_for_backward(x, s)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, s = ctx.saved_tensors
        B = s.shape[0]
        P = x.numel() // (B * x.shape[-1])
        dx, ds = torch.empty_like(x), torch.empty_like(s)
        K.mul_bcast_bwd(K.as_mat(dz), x, s, dx, ds, B, P)
        return dx, ds


def mul_bcast(x, s):
    return MulBcastFn.apply(x, s)


class AddRowsFn(Function):
    """out = a + table[row % rows(table)]; the table is a constant (sin/cos encodings) or carries its own grad path."""

    @staticmethod
    def forward(ctx, a, table, grad_slot=None):
        out = torch.empty_like(a)
        K.add_rows(a, table, out)
        ctx.tshape = table.shape
        ctx.tgrad = table.requires_grad
        ctx.grad_slot = grad_slot      # a's other consumer (a convolution created BEFORE this op) adds this gradient in its dgrad epilogue
        return out

    @staticmethod
    def backward(ctx, dout):
        dt_ = None
        if ctx.tgrad:
            dout_m = K.as_mat(dout)
            rows = 1
            for s in ctx.tshape[:-1]:
                rows *= s
            acc = torch.empty(rows, ctx.tshape[-1], device=dout.device, dtype=torch.float32)
            K.sum_over_batch(dout_m.reshape(-1, ctx.tshape[-1]), acc, dout_m.numel() // (rows * ctx.tshape[-1]))
            dt_ = acc.to(dout.dtype).view(ctx.tshape) if dout.dtype != torch.float32 else acc.view(ctx.tshape)
        if ctx.grad_slot is not None:
            ctx.grad_slot.put(dout)
            return None, dt_, None
        return dout, dt_, None


def add_rows(a, table, grad_slot=None):
    return AddRowsFn.apply(a, table, grad_slot)


class AddDropoutFn(Function):
    """out = a + dropout(b)   (layers.py:338 with p > 0)."""

    @staticmethod
    def forward(ctx, a, b, p, res_slot=None):
        out = torch.empty_like(b)
        seed = RT.next_seed() if p > 0 else 0
        K.add_dropout(a, b, out, p, seed)
        ctx.cfg = (p, seed, res_slot)
        return out

    @staticmethod
    def backward(ctx, dout):
        p, seed, res_slot = ctx.cfg
        dout = K.as_mat(dout)
        db = torch.empty_like(dout)
        K.add_dropout(None, dout, db, p, seed)
        if res_slot is not None:      # `a`'s gradient rides the slot into the LayerNorm backward that also differentiates `a` (LayerNormFn)
            res_slot.put(dout)
            return None, db, None, None
        return dout, db, None, None


def add_dropout(a, b, p, res_slot=None):
    return AddDropoutFn.apply(a, b, float(p), res_slot)


# ------------------------------------------------------------------------------------------------
# SSG trunk pieces (model/ssg.py): explicit im2col, 3x3 bias convolution, max pool, align_corners bilinear
# ------------------------------------------------------------------------------------------------
class Im2colFn(Function):
    """x [B, H, W, C] -> patch rows [B, OH, OW, kh*kw*C]; the transpose (col2im) is the backward.  Strided convolutions are
    `linear(im2col(x), w, b)` / `conv_bn_act(im2col(x), w, bn, ksize=1)` with the weight's physical [Cout][ky][kx][ci] rows."""

    @staticmethod
    def forward(ctx, x, kh, kw, stride, pad):
        B, H, W, C = x.shape
        OH, OW = K.conv_out(H, kh, stride, pad), K.conv_out(W, kw, stride, pad)
        col = torch.empty(B, OH, OW, kh * kw * C, device=x.device, dtype=x.dtype)
        K.im2col_nhwc(x, col, kh, kw, stride, pad)
        ctx.cfg = (tuple(x.shape), kh, kw, stride, pad)
        return col

    @staticmethod
    def backward(ctx, dcol):
        shape, kh, kw, stride, pad = ctx.cfg
        dx = torch.empty(shape, device=dcol.device, dtype=dcol.dtype)
        K.col2im_nhwc(K.as_mat(dcol), dx, kh, kw, stride, pad)
        return dx, None, None, None, None


def im2col(x, kh, kw, stride, pad):
    return Im2colFn.apply(x, kh, kw, stride, pad)


class Conv3BiasActFn(Function):
    """y = act(conv3x3_s1_p1(x) + b) as an implicit GEMM (ssg.py:123-133,151-164,178-181), act in {none, relu, tanh}."""

    @staticmethod
    def forward(ctx, x, _wp, _bp, w: WRef, b: Optional[WRef], act):
        B, H, W, cin = x.shape
        C = w.rows
        dt = _cdt(x)
        y = torch.empty(B, H, W, C, device=x.device, dtype=x.dtype)
        K.gemm(dt, K.A_IM2COL, K.B_KC, x, w.w(x.dtype), y, B * H * W, C, 9 * cin, K.mat(x)[2], w.cols, C, b_off=w.off, conv=(H, W, cin),
               bias=b.master() if b is not None else None, act=act)
        ctx.cfg = (w, b, act)
        ctx.save_for_backward(x, y if act != K.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        w, b, act = ctx.cfg
        x, y = ctx.saved_tensors
        B, H, W, cin = x.shape
        C = w.rows
        dt = _cdt(x)
        g = K.as_mat(dy)
        if act != K.ACT_NONE:
            g = torch.empty(B, H, W, C, device=dy.device, dtype=dy.dtype)
            K.act_bwd(K.as_mat(dy), y, g, 0 if act == K.ACT_RELU else 2)
        M = B * H * W
        def wgrad():
            bb = b
            if bb is not None and RT.deterministic:
                bias_grad(g, bb, 0, C, pooled=True)
                bb = None
            wgrad_gemm(dt, K.B_NC_IM2COL, g, x, w.G, C, 9 * cin, M, C, K.mat(x)[2], w.cols, c_off=w.off, conv=(H, W, cin),
                       a_sum=bb.G if bb is not None else None, a_sum_off=bb.off if bb is not None else 0)
        RT.on_wgrad_stream(wgrad, g, x)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x.shape, device=x.device, dtype=x.dtype)
            if DGRAD_T:
                K.gemm(dt, K.A_IM2COL, K.B_KC, g, w.store.weights_t(x.dtype), dx, M, cin, 9 * C, C, 9 * C, cin, b_off=w.off, conv=(H, W, C))
            else:
                K.gemm(dt, K.A_IM2COL, K.B_NC_DGRAD, g, w.w(x.dtype), dx, M, cin, 9 * C, C, cin, cin, b_off=w.off, conv=(H, W, C))
        w.done()       # (after the data gradient that reads it: see LinearFn.backward)
        if b is not None:
            b.done()
        return dx, None, None, None, None, None


def conv_bias_act(x, w: WRef, b: Optional[WRef], *, ksize, stride=1, act=K.ACT_NONE):
    """Bias convolution of the SSG heads.  3x3/s1 with vector-friendly channel counts -> implicit GEMM; 1x1/s1 -> plain GEMM;
    anything else (stride 2, 12 box channels) -> explicit im2col rows + GEMM."""
    cin, cout = x.shape[-1], w.rows
    if ksize == 1 and stride == 1:
        return linear(x, w, b, act=act)
    if ksize == 3 and stride == 1 and cin % 32 == 0 and cout % 32 == 0:
        return Conv3BiasActFn.apply(x, w.param, b.param if b is not None else None, w, b, act)
    return linear(im2col(x, ksize, ksize, stride, ksize // 2), w, b, act=act)


class MaxPool3s2Fn(Function):
    """nn.MaxPool2d(3, 2, 1) (ssg.py:66) on channels-last maps."""

    @staticmethod
    def forward(ctx, x):
        B, H, W, C = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(B, OH, OW, C, device=x.device, dtype=x.dtype)
        arg = torch.empty(B, OH, OW, C, device=x.device, dtype=torch.uint8)
        K.maxpool3s2_fwd(x, y, arg)
        ctx.shape = tuple(x.shape)
        ctx.save_for_backward(arg)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        dx = torch.empty(ctx.shape, device=dy.device, dtype=dy.dtype)
        K.maxpool3s2_bwd(K.as_mat(dy), arg, dx)
        return dx


def maxpool3s2(x):
    return MaxPool3s2Fn.apply(x)


class Upsample2AcFn(Function):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (ssg.py:159)."""

    @staticmethod
    def forward(ctx, x):
        B, H, W, C = x.shape
        y = torch.empty(B, 2 * H, 2 * W, C, device=x.device, dtype=x.dtype)
        K.upsample2ac_fwd(x, y)
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = torch.empty(ctx.shape, device=dy.device, dtype=dy.dtype)
        K.upsample2ac_bwd(K.as_mat(dy), dx)
        return dx


def upsample2ac(x):
    return Upsample2AcFn.apply(x)


class AddFn(Function):
    """a + b (FPN top-down merge, ssg.py:196,199) — one streaming kernel (add_dropout with p = 0)."""

    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty_like(b)
        K.add_dropout(a, b, out, 0.0, 0)
        return out

    @staticmethod
    def backward(ctx, dout):
        return dout, dout


def add(a, b):
    return AddFn.apply(a, b)


# ------------------------------------------------------------------------------------------------
# text front end
# ------------------------------------------------------------------------------------------------
class EmbeddingFn(Function):
    """token_embedding(text) + positional_embedding[:L]   (clip.py:440-443)."""

    @staticmethod
    def forward(ctx, word, _tp, _pp, tok: WRef, pos: WRef, dtype):
        B, L = word.shape
        C = tok.cols
        out = torch.empty(B * L, C, device=word.device, dtype=dtype)
        wbuf = tok.w(dtype)
        esz = 2 if dtype == torch.bfloat16 else 4
        K.check(K.lib().crog_embedding_fwd(K.dcode(dtype), K.ptr(word), K.ptr(wbuf) + esz * tok.off, K.ptr(wbuf) + esz * pos.off, K.ptr(out),
                                           B * L, L, C, tok.rows, K.stream()), "embedding_fwd")
        ctx.cfg = (tok, pos, L)
        ctx.save_for_backward(word)
        return out

    @staticmethod
    def backward(ctx, dout):
        tok, pos, L = ctx.cfg
        (word,) = ctx.saved_tensors
        dout = K.as_mat(dout)
        C = tok.cols
        K.check(K.lib().crog_embedding_bwd(K.dcode(dout), K.ptr(word), K.ptr(dout), K.ptr(tok.G) + 4 * tok.off, K.ptr(pos.G) + 4 * pos.off,
                                           dout.shape[0], L, C, tok.rows, K.stream()), "embedding_bwd")
        tok.done()
        pos.done()
        # the text tower's backward ends here: whatever it still has parked (its first block's projections: 24 blocks) goes now, not at the
        # end of the image tower's backward 3 ms later, where it sat between the last weight gradient and the last Adam launch (79 us)
        RT.flush_short_groups()
        return (None,) * 6


def embedding(word, tok: WRef, pos: WRef, dtype):
    return EmbeddingFn.apply(word, tok.param, pos.param, tok, pos, dtype)


class VitTokensFn(Function):
    """[class_embedding; patch rows] + positional_embedding  (clip.py:313-320) -> token rows [B*T, C], batch-major."""

    @staticmethod
    def forward(ctx, y, _cp, _pp, cls: WRef, pos: WRef, B):
        T, C = pos.rows, pos.cols
        _, _, ldy = K.mat(y)
        out = torch.empty(B * T, C, device=y.device, dtype=y.dtype)
        wbuf = cls.w(y.dtype)
        esz = 2 if y.dtype == torch.bfloat16 else 4
        K.check(K.lib().crog_vit_tokens_fwd(K.dcode(y), K.ptr(y), ldy, K.ptr(wbuf) + esz * cls.off, K.ptr(wbuf) + esz * pos.off, K.ptr(out),
                                            B, T, C, K.stream()), "vit_tokens_fwd")
        ctx.cfg = (cls, pos, B)
        return out

    @staticmethod
    def backward(ctx, dtok):
        cls, pos, B = ctx.cfg
        dtok = K.as_mat(dtok)
        T, C = pos.rows, pos.cols
        dy = torch.empty(B * (T - 1), C, device=dtok.device, dtype=dtok.dtype)
        K.check(K.lib().crog_vit_tokens_bwd(K.dcode(dtok), K.ptr(dtok), K.ptr(dy), C, K.ptr(cls.G) + 4 * cls.off, K.ptr(pos.G) + 4 * pos.off,
                                            B, T, C, K.stream()), "vit_tokens_bwd")
        cls.done()
        pos.done()
        return dy, None, None, None, None, None


def vit_tokens(y, cls: WRef, pos: WRef, B):
    return VitTokensFn.apply(y, cls.param, pos.param, cls, pos, B)


class GatherRowsFn(Function):
    """x[idx] over rows (EOT token select, clip.py:451-452)."""

    @staticmethod
    def forward(ctx, x, idx):
        out = torch.empty(idx.shape[0], x.shape[-1], device=x.device, dtype=x.dtype)
        K.gather_rows(x, idx, out)
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dx = torch.zeros(ctx.shape, device=dout.device, dtype=dout.dtype)
        K.scatter_rows(K.as_mat(dout), idx, dx)
        return dx, None


def gather_rows(x, idx):
    return GatherRowsFn.apply(x, idx)


class TableMatmulFn(Function):
    """out[R, C] = A[R, Kp] @ table[K, C] for a constant A and a parameter table: the bicubic 7x7 -> HxW resize of
    attnpool.positional_embedding[1:] written as its (fixed) interpolation matrix (clip.py:95-108), and
    `x @ text_projection` / `x @ proj` style right-multiplications by a [K, N] parameter (clip.py:452,330)."""

    @staticmethod
    def forward(ctx, A, _tp, table: WRef, dtype):
        R, Kp = A.shape
        C = table.cols
        out = torch.empty(R, C, device=A.device, dtype=dtype)
        K.gemm(K.dcode(dtype), K.A_KC, K.B_NC, A, table.w(dtype), out, R, C, table.rows, Kp, C, C, b_off=table.off)
        ctx.cfg = (table, A.requires_grad)
        ctx.save_for_backward(A)
        return out

    @staticmethod
    def backward(ctx, dout):
        table, a_grad = ctx.cfg
        (A,) = ctx.saved_tensors
        dout = K.as_mat(dout)
        R, Kp = A.shape
        C = table.cols
        dt = K.dcode(dout)
        wgrad_gemm(dt, K.B_NC, A, dout, table.G, table.rows, C, R, Kp, K.mat(dout)[2], C, c_off=table.off)
        dA = None
        if a_grad:
            dA = torch.empty_like(A)
            # dA[R, K] = dout[R, C] @ table[K, C]^T : table rows are K-contiguous over C
            K.gemm(dt, K.A_KC, K.B_KC, dout, table.w(dout.dtype), dA, R, table.rows, C, K.mat(dout)[2], C, Kp, b_off=table.off)
            if Kp > table.rows:
                dA[:, table.rows:].zero_()
        table.done()
        return dA, None, None, None


def table_matmul(A, table: WRef, dtype):
    return TableMatmulFn.apply(A, table.param, table, dtype)


# ------------------------------------------------------------------------------------------------
# dynamic-conv head + losses
# ------------------------------------------------------------------------------------------------
class DynHeadFn(Function):
    """Projector / MultiTaskProjector tail (layers.py:90-132,161-173): txt Linear(state) -> per-sample 3x3 kernel + bias,
    applied to each C-channel group of x5.  Output: fp32 logits [B, groups, H, W]."""

    @staticmethod
    def forward(ctx, x5, state, _wp, _bp, tw: WRef, tb: WRef, C):
        B, H, W, CT = x5.shape
        groups = CT // C
        dev, dtype = x5.device, x5.dtype
        dt = K.dcode(dtype)
        nout = C * 9 + 1
        ldw = _pad(nout, 8)
        word = torch.empty(B, ldw, device=dev, dtype=torch.float32)
        lin_fwd(state, tw, word, bias=tb, out_mode=K.OUT_F32, ldc=ldw)
        wpad = torch.empty(B, C, 16, device=dev, dtype=dtype)
        K.head_pack_weights(word, wpad, B, C)
        P = H * W
        t = torch.empty(B * P * groups, 16, device=dev, dtype=torch.float32)
        K.gemm(dt, K.A_KC, K.B_NC, x5, wpad, t, P * groups, 16, C, C, 16, 16, batch=B, sA=(P * groups * C, 0), sB=(C * 16, 0),
               sC=(P * groups * 16, 0), out_mode=K.OUT_F32)
        out = torch.empty(B, groups, H, W, device=dev, dtype=torch.float32)
        K.head_stencil_fwd(t, word, C * 9, out, B, groups, H, W)
        ctx.cfg = (tw, tb, C, groups, ldw)
        ctx.save_for_backward(x5, state, wpad)
        return out

    @staticmethod
    def backward(ctx, dout):
        tw, tb, C, groups, ldw = ctx.cfg
        x5, state, wpad = ctx.saved_tensors
        B, H, W, CT = x5.shape
        dev, dtype = x5.device, x5.dtype
        dt = K.dcode(dtype)
        P = H * W
        dout = dout.contiguous()
        dtb = torch.empty(B * P * groups, 16, device=dev, dtype=dtype)
        dbias = torch.empty(B, device=dev, dtype=torch.float32)
        K.head_stencil_bwd(dout, dtb, dbias, B, groups, H, W)
        # dx5[b,(p,g),c] = sum_tap dt[b,(p,g),tap] * w[b,c,tap]
        dx5 = torch.empty_like(x5)
        K.gemm(dt, K.A_KC, K.B_KC, dtb, wpad, dx5, P * groups, C, 16, 16, 16, C, batch=B, sA=(P * groups * 16, 0), sB=(C * 16, 0),
               sC=(P * groups * C, 0))
        # dw[b,c,tap] = sum_(p,g) x5[b,(p,g),c] * dt[b,(p,g),tap]
        dwpad = torch.zeros(B, C, 16, device=dev, dtype=torch.float32)
        sk = 1 if RT.deterministic else max(1, min(16, (P * groups) // 2048))      # (one slice per sample: plain stores, one summation order)
        K.gemm(dt, K.A_MC, K.B_NC, x5, dtb, dwpad, C, 16, P * groups, C, 16, 16, batch=B, sA=(P * groups * C, 0), sB=(P * groups * 16, 0),
               sC=(C * 16, 0), splitk=sk, out_mode=K.OUT_F32 if RT.deterministic else K.OUT_F32_ATOMIC)
        dword = torch.empty(B, ldw, device=dev, dtype=dtype)
        K.head_unpack_wgrad(dwpad, dbias, dword, B, C)
        nout = C * 9 + 1
        lin_wgrad(dword, state, tw, N=nout)
        bias_grad(dword[:, :nout], tb)
        dstate = torch.empty_like(state)
        lin_dgrad(dword, tw, dstate, N=nout)
        tw.done()      # (after the data gradient that reads it: see LinearFn.backward)
        tb.done()
        return dx5, dstate, None, None, None, None, None


def dyn_head(x5, state, tw: WRef, tb: WRef, C):
    return DynHeadFn.apply(x5, state, tw.param, tb.param, tw, tb, C)


class FusedHeadFn(Function):
    """vis.4 (1x1 conv in_dim -> groups*in_dim with bias, layers.py:58) folded into the dynamic per-sample 3x3 head
    (layers.py:90-132): per sample the two linear maps compose into Wf[g][tap][k] = sum_c W5[g*C+c][k] * w_b[c][tap], so the taps
    come straight from x4 (t = x4 . Wf^T) and the groups*C-channel map (0.89 GB at B = 32) is neither written nor re-read, in
    either direction.  The conv bias reaches the logits as per-sample constants cb[b][g][tap] (csrc/head.hip)."""

    @staticmethod
    def forward(ctx, x4, word, _p1, _p2, w5: WRef, b5: WRef, C, groups, tail_stream=None):
        """word: fp32 [B, pad8(9 C + 1)] = txt(state), layers.py:90-91 - HeadWordFn (its own node since round 6: it depends on the text tower
        alone, so CROG.forward runs it on the text stream, and its backward - a 32-row weight gradient, a bias sum and a 16-block data
        gradient, 0.08 ms of latency-bound launches - leaves the main chain with it)."""
        B, H, W, Cx = x4.shape
        dev, dtype = x4.device, x4.dtype
        dt = K.dcode(dtype)
        P, g = H * W, groups
        ldw = word.shape[1]
        # the "compose" half depends on word and the weights alone: with the word on another stream (tail_stream) it is enqueued there - the
        # host reaches this point long before the main chain's GPU work does, so it runs beside the projector's convolutions.  Its buffers are
        # allocated IN that stream's context (a block just freed on this stream may still be in use by queued work here).
        cur = torch.cuda.current_stream()
        side = tail_stream if (tail_stream is not None and tail_stream != cur and K._STREAM_OVERRIDE is None) else None
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            wpad = torch.empty(B, C, 16, device=dev, dtype=dtype)
            Wf = torch.empty(B, g * 16, C, device=dev, dtype=dtype)
            cb = torch.empty(B, g, 16, device=dev, dtype=torch.float32)
            K.head_pack_weights(word, wpad, B, C)
            # Wf[b][g][tap][k] = sum_c wpad[b][c][tap] * W5[g*C + c][k]
            K.gemm(dt, K.A_MC, K.B_NC, wpad, w5.w(dtype), Wf, 16, C, C, 16, w5.cols, C, batch=B * g, batch_inner=g,
                   sA=(C * 16, 0), sB=(0, C * w5.cols), sC=(g * 16 * C, 16 * C), b_off=w5.off)
            K.head_cb_fwd(b5.P, b5.off, wpad, cb, B, g, C)
        if side is not None:
            cur.wait_stream(side)
            for t_ in (wpad, Wf, cb):
                t_.record_stream(cur)
        ldx = K.mat(x4)[2]
        t = torch.empty(B * P * g, 16, device=dev, dtype=torch.float32)
        K.gemm(dt, K.A_KC, K.B_KC, x4, Wf, t, P, g * 16, C, ldx, C, g * 16, batch=B, sA=(P * ldx, 0), sB=(g * 16 * C, 0),
               sC=(P * g * 16, 0), out_mode=K.OUT_F32)
        out = torch.empty(B, g, H, W, device=dev, dtype=torch.float32)
        K.head_stencil_fwd(t, word, C * 9, out, B, g, H, W, tbias=cb)
        ctx.cfg = (w5, b5, C, g, ldw)
        ctx.tail_stream = tail_stream      # the stream `word` was produced on (None: this one)
        ctx.save_for_backward(x4, wpad, Wf)
        return out

    @staticmethod
    def backward(ctx, dout):
        w5, b5, C, g, ldw = ctx.cfg
        x4, wpad, Wf = ctx.saved_tensors
        B, H, W, _ = x4.shape
        dev, dtype = x4.device, x4.dtype
        dt = K.dcode(dtype)
        P = H * W
        ldx = K.mat(x4)[2]
        dout = dout.contiguous()
        dtb = torch.empty(B * P * g, 16, device=dev, dtype=dtype)
        dbias = torch.empty(B, device=dev, dtype=torch.float32)
        K.head_stencil_bwd(dout, dtb, dbias, B, g, H, W)
        dcb = torch.empty(B, g, 16, device=dev, dtype=torch.float32)
        K.head_tap_sums(dtb, dcb, B, g, P)
        # dx4[b][p][k] = sum_(g,tap) dt[b][p][(g,tap)] * Wf[b][(g,tap)][k]
        dx4 = torch.empty(B, H, W, C, device=dev, dtype=dtype)
        K.gemm(dt, K.A_KC, K.B_NC, dtb, Wf, dx4, P, C, g * 16, g * 16, C, C, batch=B, sA=(P * g * 16, 0), sB=(g * 16 * C, 0), sC=(P * C, 0))
        # Everything below feeds the gradient of `word` and two parameter gradients - nothing the main chain waits for.  When `word` came from
        # another stream (CROG.forward computes txt(state) on the text stream, `tail_stream`), the whole tail is enqueued THERE: its only consumer,
        # HeadWordFn.backward, runs on that stream too (autograd replays a node on its forward stream), so stream order carries the dependency and
        # the main chain goes on with dx4 at once (0.15 ms of launches at the very start of backward).  torch's fills run on the current stream:
        # they are issued first, then the tail stream waits for this one.
        dWf = torch.zeros(B, g * 16, C, device=dev, dtype=torch.float32)
        dwpad = torch.zeros(B, C, 16, device=dev, dtype=torch.float32)
        dWf_c = dWf if dtype == torch.float32 else torch.empty(B, g * 16, C, device=dev, dtype=dtype)
        dword = torch.empty(B, ldw, device=dev, dtype=torch.float32)      # (the unpack kernel writes the padding columns too)
        side = ctx.tail_stream if (ctx.tail_stream is not None and not RT.deterministic and K._STREAM_OVERRIDE is None) else None
        cur = torch.cuda.current_stream()
        if side is not None and side != cur:
            side.wait_stream(cur)
            K.set_stream_override(side.cuda_stream)
        else:
            side = None
        try:
            # dWf[b][(g,tap)][k] = sum_p dt[b][p][(g,tap)] * x4[b][p][k]
            sk = 1 if RT.deterministic else max(1, min(24, P // 1024))
            K.gemm(dt, K.A_MC, K.B_NC, dtb, x4, dWf, g * 16, C, P, g * 16, ldx, C, batch=B, sA=(P * g * 16, 0), sB=(P * ldx, 0), sC=(g * 16 * C, 0),
                   splitk=sk, out_mode=K.OUT_F32 if RT.deterministic else K.OUT_F32_ATOMIC)
            if dtype != torch.float32:
                K.cast_pad2d(dWf, C, C, dWf_c, C, C, B * g * 16)
            # dW5[g*C + c][k] += sum_b sum_tap wpad[b][c][tap] * dWf[b][g][tap][k]
            if RT.deterministic:
                # the B samples add into the same rows of dW5: one launch per sample (its g groups write disjoint rows), in stream order
                for bi in range(B):
                    K.gemm(dt, K.A_KC, K.B_NC, wpad, dWf_c, w5.G, C, C, 16, 16, C, w5.cols, batch=g, batch_inner=g, sA=(0, 0),
                           sB=(0, 16 * C), sC=(0, C * w5.cols), a_off=bi * C * 16, b_off=bi * g * 16 * C, c_off=w5.off, out_mode=K.OUT_F32_ATOMIC)
            else:
                def dw5():
                    K.gemm(dt, K.A_KC, K.B_NC, wpad, dWf_c, w5.G, C, C, 16, 16, C, w5.cols, batch=B * g, batch_inner=g, sA=(C * 16, 0),
                           sB=(g * 16 * C, 16 * C), sC=(0, C * w5.cols), c_off=w5.off, out_mode=K.OUT_F32_ATOMIC)
                if side is not None:
                    dw5()          # (already off the main chain)
                else:
                    RT.on_wgrad_stream(dw5, wpad, dWf_c, tag="linear")      # a parameter gradient only: the weight-gradient stream
            # dwpad[b][c][tap] = sum_g sum_k W5[g*C + c][k] * dWf[b][g][tap][k]   (+ the cb share below)
            if RT.deterministic:
                # the g groups add into the same dwpad[b]: one launch per group (its B samples write disjoint blocks), in stream order
                for gi in range(g):
                    K.gemm(dt, K.A_KC, K.B_KC, w5.w(dtype), dWf_c, dwpad, C, 16, C, w5.cols, C, 16, batch=B, batch_inner=1, sA=(0, 0),
                           sB=(g * 16 * C, 0), sC=(C * 16, 0), a_off=w5.off + gi * C * w5.cols, b_off=gi * 16 * C, out_mode=K.OUT_F32_ATOMIC)
            else:
                K.gemm(dt, K.A_KC, K.B_KC, w5.w(dtype), dWf_c, dwpad, C, 16, C, w5.cols, C, 16, batch=B * g, batch_inner=g, sA=(0, C * w5.cols),
                       sB=(g * 16 * C, 16 * C), sC=(C * 16, 0), a_off=w5.off, out_mode=K.OUT_F32_ATOMIC)
            K.head_cb_bwd(b5.P, b5.off, wpad, dcb, b5.G, b5.off, dwpad, B, g, C)
            # the gradient of `word` (fp32, as the forward's word is): HeadWordFn rounds it to the compute dtype exactly as this kernel did when
            # it wrote the compute dtype itself
            K.head_unpack_wgrad(dwpad, dbias, dword, B, C)
        finally:
            if side is not None:
                K.set_stream_override(None)
        if side is not None:
            for t in (dtb, x4, dWf, dWf_c, dwpad, dword, dcb, dbias, wpad):
                t.record_stream(side)
        # (announced after the last kernel that reads them is enqueued: see LinearFn.backward)
        w5.done()
        b5.done()
        return (dx4, dword) + (None,) * 7


class HeadWordFn(Function):
    """word = txt(state) (layers.py:90-91: Linear word_dim -> 9 C + 1) as fp32 rows padded to 8 columns: the per-sample 3x3 kernel and bias of
    the dynamic head.  A node of its own so that forward AND backward run on the stream of the text tower (CROG.forward)."""

    @staticmethod
    def forward(ctx, state, _p1, _p2, tw: WRef, tb: WRef, C):
        B = state.shape[0]
        nout = C * 9 + 1
        ldw = _pad(nout, 8)
        word = torch.empty(B, ldw, device=state.device, dtype=torch.float32)
        lin_fwd(state, tw, word, bias=tb, out_mode=K.OUT_F32, ldc=ldw)
        ctx.cfg = (tw, tb, nout)
        ctx.save_for_backward(state)
        return word

    @staticmethod
    def backward(ctx, dword32):
        tw, tb, nout = ctx.cfg
        (state,) = ctx.saved_tensors
        dword = dword32.contiguous() if state.dtype == torch.float32 else dword32.to(state.dtype)
        lin_wgrad(dword, state, tw, N=nout)
        bias_grad(dword[:, :nout], tb)
        dstate = torch.empty_like(state)
        lin_dgrad(dword, tw, dstate, N=nout)
        tw.done()
        tb.done()
        return dstate, None, None, None, None, None


def head_word(state, tw: WRef, tb: WRef, C):
    return HeadWordFn.apply(state, tw.param, tb.param, tw, tb, C)


def fused_head(x4, word, w5: WRef, b5: WRef, C, groups, tail_stream=None):
    return FusedHeadFn.apply(x4, word, w5.param, b5.param, w5, b5, C, groups, tail_stream)


class BackwardBeginFn(Function):
    """Identity on the model's differentiable outputs; its backward is the first node of the model's backward pass and gives
    the flat store the chance to honour a `zero_grad(set_to_none=True)` issued between forward and backward
    (crog_engine.py:77: forward -> optimizer.zero_grad() -> backward)."""

    @staticmethod
    def forward(ctx, store, *outs):
        ctx.store = store
        return tuple(o.view_as(o) for o in outs)

    @staticmethod
    def backward(ctx, *grads):
        ctx.store.fresh_grads_if_dropped()
        return (None,) + tuple(grads)


def backward_begin(store: ParamStore, *outs):
    res = BackwardBeginFn.apply(store, *outs)
    return res[0] if len(outs) == 1 else res


class LossFn(Function):
    """crog.py:76-99 / 119-124 in one pass: nearest target resize, weighted BCE + 4 smooth-L1, d(loss)/d(pred)."""

    @staticmethod
    def forward(ctx, pred, targets, weighted):
        B, heads, H, W = pred.shape
        Hin, Win = targets[0].shape[-2:]
        tgt_small = torch.empty(heads, B, 1, H, W, device=pred.device, dtype=torch.float32)
        sums = torch.empty(5, device=pred.device, dtype=torch.float32)
        dpred = torch.empty_like(pred)
        K.head_loss(pred, [t.contiguous() for t in targets], Hin, Win, weighted, tgt_small, sums, dpred)
        ctx.save_for_backward(dpred)
        ctx.mark_non_differentiable(tgt_small, sums)
        total = sums[:heads].sum()
        return total, sums, tgt_small

    @staticmethod
    def backward(ctx, dtotal, _a, _b):
        (dpred,) = ctx.saved_tensors
        return dpred * dtotal, None, None


def head_loss(pred, targets, weighted):
    return LossFn.apply(pred, targets, weighted)


def train_metric_beside(pred_ins, target, threshold=0.35, pr_iou=0.5):
    """train_metric on the text stream when the step has one (round 6): the two launches (27 us) leave the main chain, which goes straight
    into backward; the result is read after the end-of-backward join of the side streams (engine.train_step, graphs.GraphedTrainStep)."""
    side = RT.text_stream
    cur = torch.cuda.current_stream() if torch.cuda.is_available() else None
    if side is None or cur is None or side not in RT.streams or side == cur or not pred_ins.is_cuda:
        return train_metric(pred_ins, target, threshold, pr_iou)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        out = train_metric(pred_ins, target, threshold, pr_iou)
    pred_ins.record_stream(side)
    target.record_stream(side)
    out.record_stream(cur)
    return out


def train_metric(pred_ins, target, threshold=0.35, pr_iou=0.5):
    """utils/misc.py:115-131 on device: returns (100*IoU, 100*Prec@pr_iou) as a 2-element fp32 tensor."""
    B = pred_ins.shape[0]
    P = pred_ins[0].numel()
    if pred_ins.dtype != torch.float32:
        raise TypeError("train_metric expects fp32 logits")
    counts = torch.empty(B, 2, device=pred_ins.device, dtype=torch.float32)
    out2 = torch.empty(2, device=pred_ins.device, dtype=torch.float32)
    K.train_metric(pred_ins, pred_ins.stride(0), target.contiguous(), B, P, threshold, pr_iou, counts, out2)
    return out2
