"""Tensor-level wrappers over the C ABI (include/crog_hip.h).

Each function takes torch tensors that live on the GPU, checks layout, and enqueues the HIP
kernel on torch's current stream.  PyTorch is only the allocator / stream owner here; no torch
operator computes anything on this path.  Nothing in this file falls back to CPU or to ATen.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib
from ._lib import GemmDesc, check as _check

F32, BF16 = 0, 1
A_KC, A_IM2COL, A_MC = 0, 1, 2
B_KC, B_NC, B_NC_DGRAD, B_NC_IM2COL = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_QUICKGELU, ACT_TANH, ACT_RELU_POST = 0, 1, 2, 3, 4
OUT_T, OUT_F32, OUT_F32_ATOMIC = 0, 1, 2


# Optional per-launch timing of ONE GEMM variant (bench.py's roofline leg): {"key": (a_layout, b_layout), "records": []}
PROF = None
# While a training step is being captured for replay (crog_amd/graphs.py): {"key": (a_layout, b_layout), "nodes": []} collects the
# graph node of every launch of that variant, so that the replay can put a timer pair around exactly those launches
CAPTURE_NODES = None
# ... and {raw stream: [graph nodes]}: after every launch of this library the node it created is noted under the stream it went to, so that
# the replay re-issues each node on the stream it was captured on (crog_replay_build_tagged)
CAPTURE_TAGS = None
# a list: gemm() appends its descriptor here instead of launching it (crog_amd.runtime parks small weight gradients for crog_gemm_group)
GROUP_SINK = None
DEBUG_FLAGS = 0  # ablation / A-B bits of crog_gemm_desc.debug (set by tests and scripts/ablate_gemm.py); 0 in production
GEMM_SYMBOL = {
    (A_KC, B_KC): "gemm_pp_kernel<A_KC, 4|3|2, 5> (>= 150 tiles of 256 x 256) / gemm_dma16_kernel<A_KC, 128x128> / gemm_dma_kernel<T, A_KC, B_KC> / "
                  "gemm_skinny32_kernel (the stem's im2col GEMM)  "
                  "(1x1 conv / linear forward and data gradient on the transposed weight copy, Q.K^T)",
    (A_IM2COL, B_KC): "gemm_pp_kernel<A_IM2COL, 4|3, 4> (ping-pong 256 / 192 x 256 x 64: launches of >= 150 tiles) / gemm_dma16_kernel<A_IM2COL, 128x128> / "
                      "gemm_dma_kernel<T, A_IM2COL, B_KC> (smaller launches) / conv_sw_kernel<CI, CO, G> (32 / 64 channels on >= 64 K pixels: stem, layer1)  "
                      "(3x3 conv forward and data gradient)",
    (A_KC, B_NC): "gemm_dma_kernel<T, A_KC, B_NC>  (1x1 / linear dgrad, P.V)",
    (A_IM2COL, B_NC_DGRAD): "gemm_dma_kernel<T, A_IM2COL, B_NC_DGRAD>  (3x3 conv dgrad)",
    (A_MC, B_NC): "gemm_ppt_kernel<B_NC> (outputs >= 1 M, both sides multiples of 256) / gemm_dma_kernel<T, A_MC, B_NC>  (1x1 / linear wgrad)",
    (A_MC, B_NC_IM2COL): "gemm_ppt_kernel<B_NC_IM2COL> (outputs >= 512 K, both sides multiples of 256) / gemm_dma_kernel<T, A_MC, B_NC_IM2COL> / "
                         "wgrad_sw_kernel<CI, CO> (32 / 64 channels: stem, layer1)  (3x3 conv wgrad)",
    (A_MC, B_KC): "gemm_dma_kernel<T, A_MC, B_KC>",
}


def lib():
    return _lib.load()


def check(rc: int, what: str = ""):
    _check(rc, what)
    if CAPTURE_TAGS is not None:
        _tag_last_launch()


class Timer:
    """Timing-only HIP event without the system-scope fence of a default event (crog_timer_*, include/crog_hip.h): a pair around
    one launch inside a running step measures that launch, not the cache write-back a fencing event puts between kernels."""
    __slots__ = ("h",)

    def __init__(self):
        self.h = ctypes.c_void_p()
        check(lib().crog_timer_create(ctypes.byref(self.h)), "timer_create")

    def record(self, raw_stream=None):
        check(lib().crog_timer_record(self.h, stream() if raw_stream is None else raw_stream), "timer_record")

    def elapsed_time(self, stop: "Timer") -> float:
        """Milliseconds from this timer to `stop` (waits for `stop`), same call shape as torch.cuda.Event.elapsed_time."""
        ms = ctypes.c_float()
        check(lib().crog_timer_elapsed_ms(self.h, stop.h, ctypes.byref(ms)), "timer_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            if self.h:
                lib().crog_timer_destroy(self.h)
        except Exception:
            pass


def dcode(t_or_dtype) -> int:
    dt = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise TypeError(f"crog_amd kernels support float32 / bfloat16 activations, got {dt}")


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_DEV_INDEX = None
_STREAM_OVERRIDE = None   # raw handle set by Runtime.on_wgrad_stream while it enqueues weight-gradient kernels on the side stream


def set_stream_override(raw):
    global _STREAM_OVERRIDE
    _STREAM_OVERRIDE = raw


def stream() -> int:
    """Raw hipStream_t of torch's current stream.  torch.cuda.current_stream() costs ~8 us of Python per call (device-index
    plumbing) and is hit once per kernel launch (~1300 per step); the raw accessor is ~0.3 us."""
    global _DEV_INDEX, _LAST_STREAM
    if _STREAM_OVERRIDE is not None:
        _LAST_STREAM = _STREAM_OVERRIDE
        return _STREAM_OVERRIDE
    if _RAW_STREAM is None:
        _LAST_STREAM = torch.cuda.current_stream().cuda_stream
        return _LAST_STREAM
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()
    _LAST_STREAM = _RAW_STREAM(_DEV_INDEX)
    return _LAST_STREAM


_LAST_STREAM = None


def _tag_last_launch():
    """Capture only: note the graph node at the tail of the stream the last launch used."""
    node = ctypes.c_void_p()
    if _LAST_STREAM is not None and lib().crog_capture_last_node(_LAST_STREAM, ctypes.byref(node)) == 0 and node.value:
        CAPTURE_TAGS.setdefault(_LAST_STREAM, []).append(node.value)


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("crog_amd kernels need GPU tensors (there is no CPU path)")
    return t.data_ptr()


def mat(t: torch.Tensor):
    """View a tensor as a row matrix: returns (rows, cols, ld). Last dim must be unit-stride and
    the leading dims must collapse to a single row index with constant stride ld."""
    nd = t.dim()
    if nd == 2:   # fast path: the common case on the hot path
        r, c = t.shape
        s0, s1 = t.stride()
        if (s1 == 1 or c == 1) and s0 >= c:
            return r, c, s0
    if nd == 1:
        if t.stride(0) != 1 and t.numel() > 1:
            raise ValueError("1-D tensor must be contiguous")
        return 1, t.shape[0], t.shape[0]
    if t.stride(-1) != 1 and t.shape[-1] != 1:
        raise ValueError(f"last dim must be contiguous, strides={t.stride()}")
    ld = t.stride(-2)
    rows = 1
    for i in range(nd - 1):
        rows *= t.shape[i]
    for i in range(nd - 2):
        if t.shape[i] != 1 and t.stride(i) != t.stride(i + 1) * t.shape[i + 1]:
            raise ValueError(f"leading dims do not collapse: shape={tuple(t.shape)} strides={t.stride()}")
    if ld < t.shape[-1]:
        raise ValueError("row stride smaller than row length")
    return rows, t.shape[-1], ld


def is_mat(t: torch.Tensor) -> bool:
    try:
        mat(t)
        return True
    except ValueError:
        return False


def as_mat(t: torch.Tensor) -> torch.Tensor:
    """Return `t` if it is a valid row matrix view, else a contiguous copy (incoming autograd grads)."""
    return t if is_mat(t) else t.contiguous()


# --------------------------------------------------------------------------------------------
# GEMM
# --------------------------------------------------------------------------------------------
def gemm(dtype: int, a_layout: int, b_layout: int, A, B, C, M, N, K, lda, ldb, ldc, *, batch=1, batch_inner=1,
         sA=(0, 0), sB=(0, 0), sC=(0, 0), splitk=1, conv=(0, 0, 0), alpha=1.0, bias=None, act=ACT_NONE, R=None,
         ldr=0, out_mode=OUT_T, col_stats=None, a_off=0, b_off=0, c_off=0, a_sum=None, a_sum_off=0, stat_replicas=0, bwd_mask=None, bwd_z=None,
         bwd_ss=None, stat_sync=None):
    """Raw descriptor launch. A/B/C are tensors (or ints = device addresses); *_off are element offsets."""
    esz = 2 if dtype == BF16 else 4
    csz = esz if out_mode == OUT_T else 4
    # one positional constructor call (field order of crog_gemm_desc) instead of ~35 attribute stores: this wrapper runs ~550
    # times per step and was the largest single item of host time
    if A.__class__ is not int and not A.is_cuda:
        raise RuntimeError("crog_amd kernels need GPU tensors (there is no CPU path)")
    pa = (A if A.__class__ is int else A.data_ptr()) + a_off * esz
    pb = (B if B.__class__ is int else B.data_ptr()) + b_off * esz
    pc = (C if C.__class__ is int else C.data_ptr()) + c_off * csz
    d = GemmDesc(dtype, a_layout, b_layout, pa, pb, pc, M, N, K, lda, ldb, ldc, batch, batch_inner,
                 sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], splitk, conv[0], conv[1], conv[2], alpha,
                 None if bias is None else bias.data_ptr(), act, None if R is None else R.data_ptr(), ldr, out_mode, DEBUG_FLAGS,
                 None if col_stats is None else col_stats.data_ptr(), stat_replicas,
                 None if a_sum is None else a_sum.data_ptr() + 4 * a_sum_off,
                 None if bwd_z is None else bwd_z.data_ptr(), 0 if bwd_z is None else mat(bwd_z)[2], None if bwd_ss is None else bwd_ss.data_ptr(),
                 None if bwd_mask is None else bwd_mask.data_ptr(), stat_sync)
    if GROUP_SINK is not None:
        GROUP_SINK.append(d)
        return
    if PROF is not None and PROF.get("on", True) and (PROF["key"] is None or PROF["key"] == (a_layout, b_layout) or (a_layout, b_layout) in PROF.get("keys", ())):
        # timers go on the stream the kernel is actually launched on (the weight-gradient side stream while it is overridden)
        raw = stream()
        e0, e1 = Timer(), Timer()
        e0.record(raw)
        check(lib().crog_gemm(ctypes.byref(d), raw), "crog_gemm")
        e1.record(raw)
        PROF["records"].append((e0, e1, 2.0 * M * N * K * batch, (a_layout, b_layout, M, N, K, batch, splitk)))
        if "descs" in PROF:      # scripts/profile_gemms.py replays the launch on the same memory after the step
            PROF["descs"].append(d)
        return
    if CAPTURE_NODES is not None and (CAPTURE_NODES["key"] == (a_layout, b_layout) or (a_layout, b_layout) in CAPTURE_NODES.get("keys", ())):
        raw = stream()
        check(lib().crog_gemm(ctypes.byref(d), raw), "crog_gemm")
        node = ctypes.c_void_p()
        check(lib().crog_capture_last_node(raw, ctypes.byref(node)), "capture_last_node")
        if node.value:
            CAPTURE_NODES["nodes"].append((node.value, 2.0 * M * N * K * batch, (a_layout, b_layout, M, N, K, batch, splitk)))
        return
    check(lib().crog_gemm(ctypes.byref(d), stream()), "crog_gemm")


def gemm_group(descs):
    """One launch for a list of weight-gradient descriptors (crog_gemm_group; built by gemm() with GROUP_SINK set)."""
    n = len(descs)
    arr = (GemmDesc * n)(*descs)
    check(lib().crog_gemm_group(arr, n, stream()), "crog_gemm_group")


def splitk_reduce(ws: torch.Tensor, splits: int, M: int, N: int, ldws: int, out: torch.Tensor, out_off: int, ldo: int, accumulate: bool = True):
    """out[out_off + m * ldo + n] (+)= sum_z ws[z][m][n]: second stage of a split-K weight gradient written as slabs (OUT_F32, splitk > 1)."""
    check(lib().crog_splitk_reduce(ptr(ws), splits, M, N, ldws, ptr(out) + 4 * out_off, ldo, int(accumulate), stream()), "splitk_reduce")


def stat_tiles(M: int) -> int:
    return (M + 127) // 128


def pick_splitk(M: int, N: int, K: int, bk: int, conv: bool = False) -> int:
    """Split the reduction of a weight-gradient GEMM (small MxN, huge K).  The policy lives next to the tile selection in
    csrc/gemm.hip (crog_gemm_splitk_hint): enough blocks to fill 256 CUs, but >= 16-24 k-tiles per block, because every split pays
    one fp32 atomic epilogue and fp32 atomics run at ~1.3 TB/s chip-wide."""
    return lib().crog_gemm_splitk_hint(BF16 if bk >= 32 else F32, A_MC, B_NC_IM2COL if conv else B_NC, int(M), int(N), int(K))


# --------------------------------------------------------------------------------------------
# BatchNorm pieces
# --------------------------------------------------------------------------------------------
def bn_rows_per_block(M: int) -> int:
    return max(32, (M + 1023) // 1024)


def bn_partial_stats(x: torch.Tensor, partial: torch.Tensor, rows_per_block: int):
    M, C, ld = mat(x)
    check(lib().crog_bn_partial_stats(dcode(x), ptr(x), M, C, ld, rows_per_block, ptr(partial), stream()), "bn_partial_stats")


def reduce_pairs(partial: torch.Tensor, nparts: int, C: int, sums: torch.Tensor, zeroed: bool = False):
    check(lib().crog_reduce_pairs(ptr(partial), nparts, C, ptr(sums), 1 if zeroed else 0, stream()), "reduce_pairs")


def split_pairs(sums: torch.Tensor, C: int, a: torch.Tensor, b: torch.Tensor):
    check(lib().crog_split_pairs(ptr(sums), C, ptr(a), ptr(b), stream()), "split_pairs")


def bn_finalize(sums, count, gamma, beta, running_mean, running_var, momentum, eps, C, scale_shift, mean_invstd):
    check(lib().crog_bn_finalize(ptr(sums), float(count), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                 float(momentum), float(eps), C, ptr(scale_shift), ptr(mean_invstd), stream()), "bn_finalize")


def bn_eval_scale(gamma, beta, running_mean, running_var, eps, C, scale_shift):
    check(lib().crog_bn_eval_scale(ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps), C,
                                   ptr(scale_shift), stream()), "bn_eval_scale")


def bn_fold_weights(w_master, w_off, cols_src, rows_per_channel, gamma, beta, running_mean, running_var, eps, w_dst, cols_dst, rows, bias_dst):
    """Eval-mode BatchNorm folded into the convolution weights (crog_bn_fold_weights): fp32 master rows -> scaled compute-dtype rows + bias."""
    check(lib().crog_bn_fold_weights(dcode(w_dst), ptr(w_master) + 4 * w_off, cols_src, cols_src, rows_per_channel, ptr(gamma), ptr(beta),
                                     ptr(running_mean), ptr(running_var), float(eps), ptr(w_dst), cols_dst, cols_dst, rows, ptr(bias_dst),
                                     stream()), "bn_fold_weights")


def relu_mask_like(y: torch.Tensor) -> torch.Tensor:
    """uint8 [M, C / vec]: one byte of sign bits per 16-byte vector of y (vec = 8 for bf16, 4 for fp32)."""
    M, C, _ = mat(y)
    return torch.empty(M, C // (8 if y.dtype == torch.bfloat16 else 4), device=y.device, dtype=torch.uint8)


def bn_apply(z, scale_shift, res, relu: bool, y, relu_mask=None):
    M, C, ldz = mat(z)
    _, _, ldy = mat(y)
    ldr = mat(res)[2] if res is not None else 0
    check(lib().crog_bn_apply(dcode(z), ptr(z), ldz, ptr(scale_shift), ptr(res), ldr, int(relu), ptr(y), ldy, M, C, ptr(relu_mask),
                              stream()), "bn_apply")


def bn_apply_stats(z, sums, replicas, count, gamma, beta, running_mean, running_var, momentum, eps, scale_shift, mean_invstd, res, relu, y,
                   relu_mask=None, pool=None):
    """pool=(H, W): y is the 2 x 2-average-pooled map (crog_bn_apply_stats_pool)."""
    M, C, ldz = mat(z)
    if pool is not None:
        check(lib().crog_bn_apply_stats_pool(dcode(z), ptr(z), ldz, ptr(sums), replicas, float(count), ptr(gamma), ptr(beta), ptr(running_mean),
                                             ptr(running_var), float(momentum), float(eps), ptr(scale_shift), ptr(mean_invstd), int(relu), ptr(y),
                                             mat(y)[2], M, C, pool[0], pool[1], stream()), "bn_apply_stats_pool")
        return
    ldr = mat(res)[2] if res is not None else 0
    check(lib().crog_bn_apply_stats(dcode(z), ptr(z), ldz, ptr(sums), replicas, float(count), ptr(gamma), ptr(beta), ptr(running_mean),
                                    ptr(running_var), float(momentum), float(eps), ptr(scale_shift), ptr(mean_invstd), ptr(res), ldr,
                                    int(relu), ptr(y), mat(y)[2], M, C, ptr(relu_mask), stream()), "bn_apply_stats")


def bn_bwd_partial(dy, y, z, mean_invstd, rows_per_block, partial, relu_ss=None, replicas=0, relu_mask=None, pool=None, stat_sync=None, tail=False):
    """pool=(H, W): dy is the gradient of the pooled map (crog_bn_bwd_partial_pool).
    tail: `partial` is [R][C][2] rows + [C][2] totals + a counter word; the last block stores the totals, after exchanging them with the other
    ranks when stat_sync (rccl.DirectComm.sync_block()) is given (crog_bn_bwd_partial_sync)."""
    M, C, lddy = mat(dy)
    if tail:
        Mz = mat(z)[0] if pool is not None else M
        ldy = mat(y)[2] if y is not None else 0
        check(lib().crog_bn_bwd_partial_sync(dcode(dy), ptr(dy), lddy, ptr(y), ldy, ptr(z), mat(z)[2], ptr(mean_invstd), ptr(relu_ss), Mz, C,
                                             rows_per_block, ptr(partial), replicas, ptr(relu_mask), pool[0] if pool is not None else 0,
                                             pool[1] if pool is not None else 0, stat_sync, stream()), "bn_bwd_partial_sync")
        return
    if pool is not None:
        M = mat(z)[0]
        check(lib().crog_bn_bwd_partial_pool(dcode(dy), ptr(dy), lddy, ptr(z), mat(z)[2], ptr(mean_invstd), ptr(relu_ss), M, C, rows_per_block,
                                             ptr(partial), replicas, pool[0], pool[1], stream()), "bn_bwd_partial_pool")
        return
    ldy = mat(y)[2] if y is not None else 0
    check(lib().crog_bn_bwd_partial(dcode(dy), ptr(dy), lddy, ptr(y), ldy, ptr(z), mat(z)[2], ptr(mean_invstd), ptr(relu_ss), M, C,
                                    rows_per_block, ptr(partial), replicas, ptr(relu_mask), stream()), "bn_bwd_partial")


def bn_reduce_finalize(partial, nparts, count, gamma, beta, running_mean, running_var, momentum, eps, C, scale_shift, mean_invstd):
    check(lib().crog_bn_reduce_finalize(ptr(partial), nparts, float(count), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                        float(momentum), float(eps), C, ptr(scale_shift), ptr(mean_invstd), stream()), "bn_reduce_finalize")


def reduce_split(partial, nparts, C, sums, a, b):
    check(lib().crog_reduce_split(ptr(partial), nparts, C, ptr(sums), ptr(a), ptr(b), stream()), "reduce_split")


def bn_bwd_apply(dy, y, z, mean_invstd, gamma, sums, count, dz, dres, relu_ss=None, sum_rows=0, dgamma=None, dbeta=None, relu_mask=None,
                 param_grad_scale=1.0, pool=None):
    M, C, lddy = mat(dy)
    if pool is not None:
        M = mat(z)[0]
        check(lib().crog_bn_bwd_apply_pool(dcode(dy), ptr(dy), lddy, ptr(z), mat(z)[2], ptr(mean_invstd), ptr(gamma), ptr(sums), float(count),
                                           ptr(relu_ss), ptr(dz), mat(dz)[2], M, C, sum_rows, ptr(dgamma), ptr(dbeta), float(param_grad_scale),
                                           pool[0], pool[1], stream()), "bn_bwd_apply_pool")
        return
    ldy = mat(y)[2] if y is not None else 0
    lddres = mat(dres)[2] if dres is not None else 0
    check(lib().crog_bn_bwd_apply(dcode(dy), ptr(dy), lddy, ptr(y), ldy, ptr(z), mat(z)[2], ptr(mean_invstd), ptr(gamma),
                                  ptr(sums), float(count), ptr(relu_ss), ptr(dz), mat(dz)[2], ptr(dres), lddres, M, C, sum_rows,
                                  ptr(dgamma), ptr(dbeta), float(param_grad_scale), ptr(relu_mask), stream()), "bn_bwd_apply")


# --------------------------------------------------------------------------------------------
# LayerNorm / softmax
# --------------------------------------------------------------------------------------------
def ln_fwd(x, gamma, beta, eps, out, stats, res=None, out2=None, pos=None, p_in=0.0, seed_in=0, p_out=0.0, seed_out=0):
    M, C, ldx = mat(x)
    pos_rows, ldp = (0, 0)
    if pos is not None:
        pos_rows, _, ldp = mat(pos)
    check(lib().crog_ln_fwd(dcode(x), ptr(x), ldx, ptr(gamma), ptr(beta), float(eps), M, C, ptr(out), mat(out)[2], ptr(stats),
                            ptr(res), mat(res)[2] if res is not None else 0, ptr(out2), mat(out2)[2] if out2 is not None else 0,
                            ptr(pos), pos_rows, ldp, float(p_in), int(seed_in), float(p_out), int(seed_out), stream()), "ln_fwd")


def ln_bwd_rows_per_block(M: int) -> int:
    return max(8, (M + 1023) // 1024)


def ln_bwd(dout, dout2, x, gamma, stats, dx, partial, rows_per_block, p_in=0.0, seed_in=0, p_out=0.0, seed_out=0, dgamma=None, dbeta=None,
           dxadd=None, relu_in=False):
    """partial: per-block slab for an ordered reduction (crog_reduce_split), or None with dgamma / dbeta: atomic adds into the gradients.
    dxadd: gradient of the residual branch around the norm, added to dx inside the kernel."""
    M, C, ldx = mat(x)
    fn = lib().crog_ln_bwd_relu if relu_in else lib().crog_ln_bwd      # relu_in: x is a ReLU output, dx is also gated by x > 0
    check(fn(dcode(x), ptr(dout), mat(dout)[2], ptr(dout2), mat(dout2)[2] if dout2 is not None else 0, ptr(x), ldx,
                            ptr(gamma), ptr(stats), M, C, ptr(dx), mat(dx)[2], ptr(partial), rows_per_block, float(p_in),
                            int(seed_in), float(p_out), int(seed_out), ptr(dgamma), ptr(dbeta), ptr(dxadd), mat(dxadd)[2] if dxadd is not None else 0,
                            stream()), "ln_bwd")


def softmax_fwd(S, rows, Lq, Lk, ldp, heads, causal, kpm, Pd, p_drop, seed):
    check(lib().crog_softmax_fwd(dcode(S), ptr(S), rows, Lq, Lk, ldp, heads, int(causal), ptr(kpm), ptr(Pd), float(p_drop),
                                 int(seed), stream()), "softmax_fwd")


def softmax_bwd(P, dPd, rows, Lk, ldp, p_drop, seed):
    check(lib().crog_softmax_bwd(dcode(P), ptr(P), ptr(dPd), rows, Lk, ldp, float(p_drop), int(seed), stream()), "softmax_bwd")


def flash_keep_words(B, heads, Lq, Lk):
    """int32 words of the fused attention's dropout keep-bit map (include/crog_hip.h: crog_flash_attn_fwd_bits)."""
    return B * heads * ((Lk + 31) // 32) * Lq


def _keep_ptr(keep, B, heads, Lq, Lk):
    if keep is None:
        return None
    assert keep.dtype == torch.int32 and keep.is_contiguous() and keep.numel() >= flash_keep_words(B, heads, Lq, Lk), "flash_attn: keep-bit buffer"
    return ptr(keep)


def _kpm_ptr(kpm, B, Lk):
    if kpm is None:
        return None
    assert kpm.dtype in (torch.bool, torch.uint8) and kpm.is_contiguous() and kpm.numel() == B * Lk, "flash_attn: key padding mask must be [B, Lk] bytes"
    return ptr(kpm)


def flash_attn_fwd(q, k, v, o, lse, B, heads, Lq, Lk, dh, scale, p_drop, seed, ldp, causal=False, keep=None, kpm=None):
    """q/k/v/o: (tensor, column offset, row stride) triples addressing [B*L, heads*dh] column slices (bf16).  keep: int32 buffer of
    flash_keep_words() words that receives the dropout decisions (p_drop > 0) for flash_attn_bwd.  kpm: [B, Lk] bool / uint8, non-zero =
    padding key."""
    (qt, qc, ldq), (kt, kc, ldk), (vt, vc, ldv), (ot, oc, ldo) = q, k, v, o
    check(lib().crog_flash_attn_fwd_bits(ptr(qt) + 2 * qc, ldq, ptr(kt) + 2 * kc, ldk, ptr(vt) + 2 * vc, ldv, ptr(ot) + 2 * oc, ldo, ptr(lse),
                                         B, heads, Lq, Lk, dh, float(scale), float(p_drop), int(seed), int(ldp), int(bool(causal)),
                                         _kpm_ptr(kpm, B, Lk), _keep_ptr(keep, B, heads, Lq, Lk), stream()), "flash_attn_fwd")


def flash_attn_bwd(q, k, v, o, do, lse, D, dq, dk, dv, B, heads, Lq, Lk, dh, scale, p_drop, seed, ldp, causal=False, keep=None, kpm=None):
    (qt, qc, ldq), (kt, kc, ldk), (vt, vc, ldv), (ot, oc, ldo), (gt, gc, ldg) = q, k, v, o, do
    (dqt, dqc, lddq), (dkt, dkc, lddk), (dvt, dvc, lddv) = dq, dk, dv
    check(lib().crog_flash_attn_bwd_bits(ptr(qt) + 2 * qc, ldq, ptr(kt) + 2 * kc, ldk, ptr(vt) + 2 * vc, ldv, ptr(ot) + 2 * oc, ldo,
                                         ptr(gt) + 2 * gc, ldg, ptr(lse), ptr(D), ptr(dqt) + 2 * dqc, lddq, ptr(dkt) + 2 * dkc, lddk,
                                         ptr(dvt) + 2 * dvc, lddv, B, heads, Lq, Lk, dh, float(scale), float(p_drop), int(seed), int(ldp),
                                         int(bool(causal)), _kpm_ptr(kpm, B, Lk), _keep_ptr(keep, B, heads, Lq, Lk), stream()), "flash_attn_bwd")


# --------------------------------------------------------------------------------------------
# streaming ops
# --------------------------------------------------------------------------------------------
def _nhwc(x):
    B, H, W, C = x.shape
    _, _, ld = mat(x)
    return B, H, W, C, ld


def avgpool2_fwd(x, y):
    B, H, W, C, ld = _nhwc(x)
    check(lib().crog_avgpool2_fwd(dcode(x), ptr(x), ld, ptr(y), mat(y)[2], B, H, W, C, stream()), "avgpool2_fwd")


def avgpool2_bwd(dy, dx, add=None):
    """add: a gradient of the same map another branch already produced (dx = pool-backward(dy) + add)."""
    B, H, W, C, ld = _nhwc(dx)
    check(lib().crog_avgpool2_bwd_add(dcode(dy), ptr(dy), mat(dy)[2], ptr(add) if add is not None else None, mat(add)[2] if add is not None else 0,
                                      ptr(dx), ld, B, H, W, C, stream()), "avgpool2_bwd")


def upsample2_fwd(x, y):
    B, H, W, C, ld = _nhwc(x)
    check(lib().crog_upsample2_fwd(dcode(x), ptr(x), ld, ptr(y), mat(y)[2], B, H, W, C, stream()), "upsample2_fwd")


def upsample2_bwd(dy, dx):
    B, H, W, C, ld = _nhwc(dx)
    check(lib().crog_upsample2_bwd(dcode(dy), ptr(dy), mat(dy)[2], ptr(dx), ld, B, H, W, C, stream()), "upsample2_bwd")


def embedding_fwd(word, tok, pos, out, L, vocab):
    rows, C, _ = mat(out)
    check(lib().crog_embedding_fwd(dcode(out), ptr(word), ptr(tok), ptr(pos), ptr(out), rows, L, C, vocab, stream()), "embedding_fwd")


def embedding_bwd(word, dout, dtok, dpos, L, vocab):
    rows, C, _ = mat(dout)
    check(lib().crog_embedding_bwd(dcode(dout), ptr(word), ptr(dout), ptr(dtok), ptr(dpos), rows, L, C, vocab, stream()),
          "embedding_bwd")


def gather_rows(x, idx, out):
    _, C, ldx = mat(x)
    n, _, ldo = mat(out)
    check(lib().crog_gather_rows(dcode(x), ptr(x), ldx, ptr(idx), ptr(out), ldo, n, C, stream()), "gather_rows")


def scatter_rows(dout, idx, dx):
    n, C, lddo = mat(dout)
    check(lib().crog_scatter_rows(dcode(dout), ptr(dout), lddo, ptr(idx), ptr(dx), mat(dx)[2], n, C, stream()), "scatter_rows")


def mul_bcast_fwd(x, s, z, B, P):
    _, C, ldx = mat(x)
    check(lib().crog_mul_bcast_fwd(dcode(x), ptr(x), ldx, ptr(s), mat(s)[2], ptr(z), mat(z)[2], B, P, C, stream()), "mul_bcast_fwd")


def mul_bcast_bwd(dz, x, s, dx, ds, B, P):
    _, C, ldx = mat(x)
    check(lib().crog_mul_bcast_bwd(dcode(x), ptr(dz), mat(dz)[2], ptr(x), ldx, ptr(s), mat(s)[2], ptr(dx), mat(dx)[2], ptr(ds),
                                   mat(ds)[2], B, P, C, stream()), "mul_bcast_bwd")


def add_rows(a, b, out):
    M, C, lda = mat(a)
    brows, _, ldb = mat(b)
    check(lib().crog_add_rows(dcode(a), ptr(a), lda, ptr(b), ldb, brows, ptr(out), mat(out)[2], M, C, stream()), "add_rows")


def sum_over_batch(x, out, B, accumulate=False):
    M, C, ldx = mat(x)
    R = M // B
    check(lib().crog_sum_over_batch(dcode(x), ptr(x), ldx, ptr(out), mat(out)[2], B, R, C, int(accumulate), stream()), "sum_over_batch")


def add_dropout(a, b, out, p, seed):
    M, C, ldb = mat(b)
    check(lib().crog_add_dropout(dcode(b), ptr(a), mat(a)[2] if a is not None else 0, ptr(b), ldb, ptr(out), mat(out)[2], M, C,
                                 float(p), int(seed), stream()), "add_dropout")


def act_bwd(dy, y, dx, mode):
    M, C, lddy = mat(dy)
    check(lib().crog_act_bwd(dcode(dy), ptr(dy), lddy, ptr(y), mat(y)[2], ptr(dx), mat(dx)[2], M, C, mode, stream()), "act_bwd")


def quickgelu_fwd(u, out):
    M, C, ldu = mat(u)
    check(lib().crog_quickgelu_fwd(dcode(u), ptr(u), ldu, ptr(out), mat(out)[2], M, C, stream()), "quickgelu_fwd")


def stem_im2col(img, out):
    B, _, H, W = img.shape
    check(lib().crog_stem_im2col(dcode(out), ptr(img), ptr(out), B, H, W, stream()), "stem_im2col")


def conv_out(n: int, k: int, s: int, p: int) -> int:
    return (n + 2 * p - k) // s + 1


def im2col_nhwc(x, col, kh, kw, stride, pad):
    """x [B, H, W, C] -> col [B, OH, OW, kh*kw*C]"""
    B, H, W, C = x.shape
    OH, OW = conv_out(H, kh, stride, pad), conv_out(W, kw, stride, pad)
    check(lib().crog_im2col_nhwc(dcode(x), ptr(x), mat(x)[2], ptr(col), mat(col)[2], B, H, W, C, kh, kw, stride, pad, OH, OW, stream()), "im2col_nhwc")


def col2im_nhwc(dcol, dx, kh, kw, stride, pad):
    B, H, W, C = dx.shape
    OH, OW = conv_out(H, kh, stride, pad), conv_out(W, kw, stride, pad)
    check(lib().crog_col2im_nhwc(dcode(dx), ptr(dcol), mat(dcol)[2], ptr(dx), mat(dx)[2], B, H, W, C, kh, kw, stride, pad, OH, OW, stream()), "col2im_nhwc")


def im2col_image(img, col, kh, kw, stride, pad):
    """NCHW fp32 image -> col [B, OH, OW, ldo] (zero beyond kh*kw*C)"""
    B, C, H, W = img.shape
    OH, OW = conv_out(H, kh, stride, pad), conv_out(W, kw, stride, pad)
    check(lib().crog_im2col_image(dcode(col), ptr(img), ptr(col), col.shape[-1], B, C, H, W, kh, kw, stride, pad, OH, OW, stream()), "im2col_image")


def maxpool3s2_fwd(x, y, arg):
    B, H, W, C = x.shape
    check(lib().crog_maxpool3s2_fwd(dcode(x), ptr(x), mat(x)[2], ptr(y), mat(y)[2], ptr(arg), B, H, W, C, stream()), "maxpool3s2_fwd")


def maxpool3s2_bwd(dy, arg, dx):
    B, H, W, C = dx.shape
    check(lib().crog_maxpool3s2_bwd(dcode(dx), ptr(dy), mat(dy)[2], ptr(arg), ptr(dx), mat(dx)[2], B, H, W, C, stream()), "maxpool3s2_bwd")


def upsample2ac_fwd(x, y):
    B, H, W, C = x.shape
    check(lib().crog_upsample2ac_fwd(dcode(x), ptr(x), mat(x)[2], ptr(y), mat(y)[2], B, H, W, C, stream()), "upsample2ac_fwd")


def upsample2ac_bwd(dy, dx):
    B, H, W, C = dx.shape
    check(lib().crog_upsample2ac_bwd(dcode(dx), ptr(dy), mat(dy)[2], ptr(dx), mat(dx)[2], B, H, W, C, stream()), "upsample2ac_bwd")


def patchify(img, out, patch: int):
    B, _, H, W = img.shape
    check(lib().crog_patchify(dcode(out), ptr(img), ptr(out), B, H, W, patch, stream()), "patchify")


def eval_maps(x: torch.Tensor, sigmoid_mask: int, H: int, W: int) -> torch.Tensor:
    """x: fp32 [B, G, h, w] logits -> fp32 [B, G, H, W] (sigmoid on the channels in `sigmoid_mask`, bicubic align_corners=True)."""
    B, G, h, w = x.shape
    x = x.contiguous()
    y = torch.empty(B, G, H, W, device=x.device, dtype=torch.float32)
    check(lib().crog_eval_maps(ptr(x), B, G, h, w, sigmoid_mask, ptr(y), H, W, stream()), "eval_maps")
    return y


def conv3_dgrad_weights(src, dst, table, count):
    check(lib().crog_conv3_dgrad_weights(dcode(src), ptr(src), ptr(dst), ptr(table), count, stream()), "conv3_dgrad_weights")


def dgrad_weights(src, dst, table, count):
    check(lib().crog_dgrad_weights(dcode(src), ptr(src), ptr(dst), ptr(table), count, stream()), "dgrad_weights")


def cast_pad2d(src, lds, cols_src, dst, ldd, cols_dst, rows, src_off=0, dst_off=0):
    dt = dcode(dst)
    sz = 2 if dt == BF16 else 4
    check(lib().crog_cast_pad2d(dt, ptr(src) + 4 * src_off, lds, cols_src, ptr(dst) + sz * dst_off, ldd, cols_dst, rows, stream()),
          "cast_pad2d")


def add_pad2d(src, lds, dst, ldd, cols, rows, dst_off=0):
    check(lib().crog_add_pad2d(ptr(src), lds, ptr(dst) + 4 * dst_off, ldd, cols, rows, stream()), "add_pad2d")


def zero_f32(t):
    check(lib().crog_zero_f32(ptr(t), t.numel(), stream()), "zero_f32")


def cast_f32_to_bf16(src, dst, n):
    check(lib().crog_cast_f32_to_bf16(ptr(src), ptr(dst), n, stream()), "cast_f32_to_bf16")


def cast_to_f32(src, dst):
    M, C, lds = mat(src)
    check(lib().crog_cast_to_f32(dcode(src), ptr(src), lds, ptr(dst), mat(dst)[2], M, C, stream()), "cast_to_f32")


def coord_fill(buf, c0, cend):
    B, H, W, _, ld = _nhwc(buf)
    check(lib().crog_coord_fill(dcode(buf), ptr(buf), ld, B, H, W, c0, cend, stream()), "coord_fill")


def colsum_workspace(M: int, C: int) -> int:
    rpb = max(8, (M + 127) // 128)
    return ((M + rpb - 1) // rpb) * C


def colsum(x, out, out_off=0, ws=None):
    """ws: fp32 workspace of colsum_workspace(M, C) floats.  REQUIRED when the launch goes to another stream than torch's current one
    (Runtime.on_wgrad_stream overrides the launch stream): a torch.empty here would belong to the current stream and could be handed
    out again while the side stream still reads it."""
    M, C, ldx = mat(x)
    rpb = max(8, (M + 127) // 128)
    if ws is None:
        if _STREAM_OVERRIDE is not None:
            raise RuntimeError("colsum on an overridden stream needs a caller-owned workspace")
        ws = torch.empty(((M + rpb - 1) // rpb) * C, device=x.device, dtype=torch.float32)
    check(lib().crog_colsum(dcode(x), ptr(x), ldx, M, C, rpb, ptr(ws), ptr(out) + 4 * out_off, stream()), "colsum")


def adam_step(p, g, m, v, n, lr, beta1, beta2, eps, wd, step, shadow=None, off=0):
    sh = None if shadow is None else ptr(shadow) + 2 * off
    check(lib().crog_adam_step(ptr(p) + 4 * off, ptr(g) + 4 * off, ptr(m) + 4 * off, ptr(v) + 4 * off, n, float(lr), float(beta1),
                               float(beta2), float(eps), float(wd), int(step), sh, stream()), "adam_step")


def adam_advance(hyper, beta1, beta2):
    """hyper: device float[4] {lr, 1 - beta1^t, sqrt(1 - beta2^t), t} -> t += 1 and its two corrections."""
    check(lib().crog_adam_advance(ptr(hyper), float(beta1), float(beta2), stream()), "adam_advance")


def adam_step_dev(p, g, m, v, n, hyper, beta1, beta2, eps, wd, shadow=None, off=0):
    sh = None if shadow is None else ptr(shadow) + 2 * off
    check(lib().crog_adam_step_dev(ptr(p) + 4 * off, ptr(g) + 4 * off, ptr(m) + 4 * off, ptr(v) + 4 * off, n, ptr(hyper), float(beta1),
                                   float(beta2), float(eps), float(wd), sh, stream()), "adam_step_dev")


def set_seed_epoch(epoch: Optional[torch.Tensor]):
    """Install (None: remove) the device uint64 every dropout kernel adds to its seeds (include/crog_hip.h)."""
    if epoch is not None and (epoch.dtype != torch.int64 or epoch.numel() != 1 or not epoch.is_cuda):
        raise TypeError("seed epoch must be a one-element int64 GPU tensor")
    check(lib().crog_set_seed_epoch(None if epoch is None else epoch.data_ptr()), "set_seed_epoch")


def counter_add(counter: torch.Tensor, inc: int):
    check(lib().crog_counter_add(ptr(counter), int(inc), stream()), "counter_add")


# --------------------------------------------------------------------------------------------
# head
# --------------------------------------------------------------------------------------------
def head_pack_weights(word, wpad, B, C):
    check(lib().crog_head_pack_weights(dcode(wpad), ptr(word), mat(word)[2], ptr(wpad), B, C, stream()), "head_pack_weights")


def head_unpack_wgrad(dwpad, dbias, dword, B, C):
    check(lib().crog_head_unpack_wgrad(dcode(dword), ptr(dwpad), ptr(dbias), ptr(dword), mat(dword)[2], B, C, stream()),
          "head_unpack_wgrad")


def head_stencil_fwd(t, word, bias_col, out, B, heads, H, W, tbias=None):
    check(lib().crog_head_stencil_fwd(ptr(t), ptr(word), mat(word)[2], bias_col, ptr(tbias), ptr(out), B, heads, H, W, stream()),
          "head_stencil_fwd")


def head_cb_fwd(b5, b5_off, wpad, cb, B, heads, C):
    check(lib().crog_head_cb_fwd(dcode(wpad), ptr(b5) + 4 * b5_off, ptr(wpad), ptr(cb), B, heads, C, stream()), "head_cb_fwd")


def head_tap_sums(dt, dcb, B, heads, P):
    check(lib().crog_head_tap_sums(dcode(dt), ptr(dt), ptr(dcb), B, heads, P, stream()), "head_tap_sums")


def head_cb_bwd(b5, b5_off, wpad, dcb, db5, db5_off, dwpad, B, heads, C):
    check(lib().crog_head_cb_bwd(dcode(wpad), ptr(b5) + 4 * b5_off, ptr(wpad), ptr(dcb), ptr(db5) + 4 * db5_off, ptr(dwpad), B, heads, C,
                                 stream()), "head_cb_bwd")


def head_stencil_bwd(dout, dt, dbias, B, heads, H, W):
    check(lib().crog_head_stencil_bwd(dcode(dt), ptr(dout), ptr(dt), ptr(dbias), B, heads, H, W, stream()), "head_stencil_bwd")


def head_loss(pred, targets, Hin, Win, weighted, tgt_small, loss_sums, dpred):
    B, heads, H, W = pred.shape
    arr = (ctypes.c_void_p * 5)(*[ptr(t) for t in targets] + [None] * (5 - len(targets)))
    check(lib().crog_head_loss(ptr(pred), ctypes.cast(arr, ctypes.c_void_p), B, heads, H, W, Hin, Win, int(weighted), ptr(tgt_small),
                               ptr(loss_sums), ptr(dpred), stream()), "head_loss")


def train_metric(pred, pred_bstride, tgt, B, P, threshold, pr_iou, counts, out2):
    check(lib().crog_train_metric(ptr(pred), pred_bstride, ptr(tgt), B, P, float(threshold), float(pr_iou), ptr(counts), ptr(out2),
                                  stream()), "train_metric")


# --------------------------------------------------------------------------------------------
# SSG target assignment (csrc/ssg.hip)
# --------------------------------------------------------------------------------------------
def ssg_match(anchors: torch.Tensor, gt: torch.Tensor, ng: torch.Tensor, pos_thr: float, neg_thr: float):
    """anchors [A,4] fp32 (cx,cy,w,h); gt [B,Gmax,5] fp32; ng [B] int32 -> (offsets [B,A,4], labels [B,A] i64, matched box [B,A,4],
    matched index [B,A] i64) for the whole batch in two launches."""
    A, (B, Gmax) = anchors.shape[0], gt.shape[:2]
    dev = gt.device
    claim = torch.empty(B, Gmax, device=dev, dtype=torch.int32)
    offsets = torch.empty(B, A, 4, device=dev, dtype=torch.float32)
    labels = torch.empty(B, A, device=dev, dtype=torch.int64)
    mbox = torch.empty(B, A, 4, device=dev, dtype=torch.float32)
    midx = torch.empty(B, A, device=dev, dtype=torch.int64)
    check(lib().crog_ssg_match(ptr(anchors), A, ptr(gt), ptr(ng), B, Gmax, float(pos_thr), float(neg_thr), ptr(claim), ptr(offsets), ptr(labels),
                               ptr(mbox), ptr(midx), stream()), "ssg_match")
    return offsets, labels, mbox, midx
