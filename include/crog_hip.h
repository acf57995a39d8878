/*
 * crog_hip.h — C ABI of libcrog_hip.so, the MI355X (gfx950) kernel library behind
 * crog_amd's drop-in CROG module.
 *
 * The reference (HilbertXu/CROG) has no FFI layer: its hot path calls torch.nn / ATen ops
 * from Python.  Each entry point below therefore names the reference
 * call site(s) (file:line under the reference tree) whose ATen op it replaces.  A reference
 * maintainer binds these with ctypes exactly as crog_amd/_lib.py does.
 *
 * Conventions
 *  - every function returns 0 (CROG_OK) or a negative crog_status; crog_last_error() gives text
 *  - nothing here allocates, frees or synchronises: all buffers are caller-owned device
 *    memory, work is enqueued on the hipStream_t passed as `stream`, and every launch is
 *    legal inside hipGraph stream capture
 *  - `dtype` selects the activation/compute storage type: CROG_F32 or CROG_BF16
 *    (accumulation, statistics and reductions are always fp32)
 *  - activations are channels-last: a feature map is [B*H*W, C] row-major with an explicit
 *    row stride `ld` (elements), so channel concatenation is a pointer offset
 */
#ifndef CROG_HIP_H
#define CROG_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* crog_stream_t; /* hipStream_t */

enum crog_dtype { CROG_F32 = 0, CROG_BF16 = 1 };
enum crog_status { CROG_OK = 0, CROG_ERR_ARG = -1, CROG_ERR_LAUNCH = -2 };

int crog_hip_version(void);
const char* crog_last_error(void);
/* Compile-time facts of the device code, as "key=value" words: "arch=gfx950 packed-fp32-ops=off|on".  The bit-reproducibility guarantee of
 * crog_set_deterministic (and the default mode's correctness beside a second stream's MFMA kernel) holds only for a library built WITHOUT
 * the packed-fp32 VALU instructions (crog_amd/_lib.py NO_PACKED_F32); a binding checks this word before it relies on either. */
const char* crog_build_flags(void);
/* Peak probes for the measurement harness (bench.py `measured_peaks`; SURVEY.md §8d asks for the box's own stream-copy and
 * MFMA rates beside the vendor figures).  crog_probe_mfma_bf16: `blocks` x 4 waves each issue iters x 8 independent
 * v_mfma_f32_32x32x16_bf16 on register operands (FLOP = blocks * 4 * iters * 8 * 32768); `sink` needs blocks * 256 floats and is
 * never written.  crog_probe_copy: dst[0:bytes] = src[0:bytes], 16 bytes per lane, one block per 256 * V consecutive vectors, no
 * grid-stride loop (traffic = 2 * bytes); mode 0/1/2 = V 1/2/4 with non-temporal loads and stores, 3/4/5 = the same with plain ones.
 * Timing is the caller's. */
int crog_probe_mfma_bf16(float* sink, int blocks, int iters, crog_stream_t stream);
int crog_probe_copy(const void* src, void* dst, int64_t bytes, int mode, crog_stream_t stream);

/* Per-step launch state in DEVICE memory, so that a hipGraph captured over a whole training step (crog_engine.py:60-90 as one
 * graph launch; crog_amd/graphs.py) replays with fresh values instead of the scalars baked into its kernel arguments.
 * crog_set_seed_epoch installs (NULL: removes) a device uint64 that every dropout-carrying entry point below ADDS to its `seed`
 * arguments inside the kernel (crog_ln_fwd/bwd, crog_softmax_fwd/bwd, crog_add_dropout, crog_flash_attn_fwd/bwd); it is process-wide
 * launch state like the current device: set once, before the first launch that should see it.  crog_counter_add: *counter += inc
 * on `stream` (the captured step advances the epoch by the number of seeds one step draws). */
int crog_set_seed_epoch(const uint64_t* epoch_dev);
/* Deterministic mode: process-wide launch state (like the seed epoch).  on != 0: every entry point whose result would otherwise depend
 * on the arrival order of fp32 atomic adds launches an order-independent form instead - crog_reduce_pairs (ordered slab reduction),
 * crog_embedding_bwd (one gatherer per destination row), crog_head_tap_sums / crog_head_cb_bwd / crog_head_loss (per-block partials
 * summed in block order; the scratch they need is allocated by THIS call, never by a launch).  The caller's side of the mode lives in
 * crog_amd/runtime.py set_deterministic: BatchNorm / LayerNorm statistics as slabs (stat_replicas 0, partial != NULL), split-K weight
 * gradients as CROG_OUT_F32 slabs + crog_splitk_reduce, no a_sum.  Same inputs, same bits, run after run (crog_engine.py:72-84). */
int crog_set_deterministic(int on);
int crog_counter_add(uint64_t* counter_dev, uint64_t inc, crog_stream_t stream);

/* Stream-faithful replay of a captured training step (csrc/replay.hip; replaces the ~1300 Python -> ctypes launches per step of
 * the loop at crog_engine.py:45-104 by one call).  crog_replay_build walks a hipGraph_t produced by ordinary stream capture
 * (kernel / memset / 1-D device memcpy / empty nodes; anything else -> CROG_ERR_ARG and the caller keeps issuing eagerly), recovers
 * the stream-ordered chains of the capture (at most max_chains) and the edges between them; crog_replay_launch re-issues every node in
 * capture order on streams[chain] (streams[0] = the stream the caller orders against: every other stream first waits for it and is
 * joined back into it at the end), with an event record / stream wait per cross-chain edge.  The graph must outlive the replay
 * object (kernel arguments are the node-owned copies).  crog_replay_info: node / kernel / chain / event / wait counts and chain sizes.
 * Per-launch timing inside a replay: crog_capture_last_node (during capture: the node the last launch on `stream` created) gives
 * handles; crog_replay_profile_nodes selects them, _enable switches the timer pairs (timing-only events, no system fence) on for the
 * following launches, _read waits and returns the milliseconds of the LAST profiled launch, in the order the handles were given.
 * crog_replay_build_tagged: the same, with the chain (= index into crog_replay_launch's streams) of n_tags nodes given by the caller, who
 * saw which stream each launch went to while capturing (crog_capture_last_node after the launch); untagged nodes (ATen kernels,
 * memsets) are placed by the topological rule of crog_replay_build.  Topology alone cannot tell a fork from a continuation at the first node of a side stream:
 * with four streams it merged two chains and split the main one (round 4); tags make the replay use the streams of the capture. */
int crog_capture_last_node(crog_stream_t stream, void** node_out);
int crog_replay_build(void* hip_graph, int max_chains, void** replay_out);
int crog_replay_build_tagged(void* hip_graph, int max_chains, void* const* nodes, const int* chains, int n_tags, void** replay_out);
int crog_replay_info(void* replay, int* n_nodes, int* n_kernels, int* n_chains, int* n_events, int* n_waits, int* chain_sizes,
                     int chain_sizes_cap);
int crog_replay_launch(void* replay, const crog_stream_t* streams, int n_streams);
int crog_replay_profile_nodes(void* replay, void* const* nodes, int n);
int crog_replay_profile_enable(void* replay, int on);
int crog_replay_profile_read(void* replay, float* ms_out, int cap);
int crog_replay_destroy(void* replay);

/* Timing-only HIP events for per-launch measurements inside a running step (bench.py `roofline`, scripts/profile_gemms.py).
 * They are created with hipEventDisableSystemFence: a default event performs a system-scope release when it completes (L2
 * write-back between every pair of kernels), which made the kernels BETWEEN two events measure 1.5-2.6x their rocprofv3
 * kernel-trace duration.  crog_timer_create writes an opaque handle; record enqueues it on `stream`; elapsed waits for `stop`
 * and writes the milliseconds between the two. */
int crog_timer_create(void** timer);
int crog_timer_record(void* timer, crog_stream_t stream);
int crog_timer_elapsed_ms(void* start, void* stop, float* ms);
int crog_timer_destroy(void* timer);

/* ------------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM convolution family (MFMA 32x32x16 bf16, 32x32x2 f32).
 *   C[z][m][n] = epilogue( alpha * sum_k Aop(z,m,k) * Bop(z,n,k) )
 * Replaces: F.conv2d 1x1 / 3x3 (clip.py:17-25,47-50; layers.py:8-11,55-58), nn.Linear
 * (clip.py:70-73,249-251; layers.py:14-16,62,298-301), the packed in_proj / out_proj GEMMs and
 * the QK^T / PV batched products inside F.multi_head_attention_forward (clip.py:119-139,
 * clip.py:259; layers.py:324,329), and their autograd backward (dgrad / wgrad).
 * ---------------------------------------------------------------------------------------- */
enum crog_a_layout {
  CROG_A_KC = 0,     /* A[m][k], k contiguous, row stride lda                                   */
  CROG_A_IM2COL = 1, /* A = 3x3/pad1/stride1 patches of an NHWC map [B,H,W,convC] (row stride
                        lda); k = tap*convC + c, tap = ky*3+kx; M = B*H*W                        */
  CROG_A_MC = 2      /* A stored transposed: A_mem[k][m], m contiguous, row stride lda           */
};
enum crog_b_layout {
  CROG_B_KC = 0,        /* B[n][k], k contiguous, row stride ldb (torch Linear / KRSC conv weight) */
  CROG_B_NC = 1,        /* B_mem[k][n], n contiguous, row stride ldb                              */
  CROG_B_NC_DGRAD = 2,  /* conv3x3 data-gradient weights read in place from KRSC storage
                           W[co][tap][ci] (row stride ldb = Cin): k = tap'*convC + co reads
                           W[co][8-tap'][n]                                                      */
  CROG_B_NC_IM2COL = 3  /* wgrad of conv3x3: B_mem[k = pixel][n = tap*convC + c] gathered from an
                           NHWC map [B,H,W,convC] (row stride ldb); K = B*H*W                    */
};
enum crog_act { CROG_ACT_NONE = 0, CROG_ACT_RELU = 1, CROG_ACT_QUICKGELU = 2, CROG_ACT_TANH = 3,
                CROG_ACT_RELU_POST = 4 /* relu(alpha*acc + bias + R): the ReLU AFTER the residual (eval-mode bn3 + identity of a Bottleneck, clip.py:55-56, with BatchNorm folded into the weights) */ };
enum crog_out_mode {
  CROG_OUT_T = 0,         /* store as dtype                                                      */
  CROG_OUT_F32 = 1,       /* store fp32; with splitk > 1: reduction slice z stores its partial result
                             into slab z of a [splitk][M][ldc] fp32 workspace at C (no atomics: the sum
                             over the slabs, crog_splitk_reduce, is bit-reproducible)            */
  CROG_OUT_F32_ATOMIC = 2 /* atomicAdd fp32 (required when splitk > 1; C must be pre-zeroed or
                             hold the value to accumulate onto)                                  */
};

typedef struct crog_gemm_desc {
  int dtype;     /* crog_dtype of A, B, R and (for CROG_OUT_T) C */
  int a_layout;  /* crog_a_layout */
  int b_layout;  /* crog_b_layout */
  const void* A;
  const void* B;
  void* C;
  int M, N, K;
  int64_t lda, ldb, ldc; /* row strides in elements */
  /* batching: z in [0,batch); zo = z / batch_inner, zi = z % batch_inner;
     X += zo*sXo + zi*sXi (elements).  batch_inner >= 1. */
  int batch, batch_inner;
  int64_t sAo, sAi, sBo, sBi, sCo, sCi;
  int splitk; /* >= 1; K range split across blocks (needs CROG_OUT_F32_ATOMIC if > 1) */
  /* 3x3 geometry for the IM2COL / DGRAD layouts */
  int convH, convW, convC;
  /* epilogue */
  float alpha;
  const float* bias; /* [N] fp32 or NULL, added before act */
  int act;           /* crog_act */
  const void* R;     /* residual [M][N] (dtype), added after act (before the ReLU of CROG_ACT_RELU_POST), or NULL (unbatched only) */
  int64_t ldr;
  int out_mode;      /* crog_out_mode */
  int debug;         /* 0 in production.  Timing-only ablations of the LDS-DMA kernel (results are wrong): bit 0 skips the
                        global->LDS loads of the main loop, bit 1 skips the fragment reads + MFMAs (non-pipelined build only); bit 2
                        (results stay right) sends bf16 outputs through the LDS-staged epilogue instead of the direct pair stores, bit 3
                        the same for launches with a residual only; bit 5 (timing only, results wrong) skips the fp32 atomic adds of a
                        split-K launch (scripts/ablate_wgrad.py); bits 6 / 7 / 8 (results stay right) choose the MFMA shape of that launch:
                        6 = v_mfma_f32_16x16x32 on the 256 x 256 tile, 6 + 7 = also on lean 128 x 128 launches, 8 = 32x32x16 everywhere
                        (tests/test_kernels_gpu.py, scripts/ab_mfma16.py; the default is CROG_MFMA16 / 2) */
  float* col_stats;  /* NULL, or [ceil(M/128)][N][2] fp32 partial (sum, sum of squares) over the
                        rows of each 128-row tile of v = alpha*acc + bias (BatchNorm statistics,
                        clip.py:18,21,26; layers.py:11).  batch must be 1, splitk 1. */
  int stat_replicas; /* 0: col_stats is the [ceil(M/128)][N][2] slab above (plain stores, deterministic).  R > 0: col_stats is a
                        PRE-ZEROED [R][N][2] buffer; the slab row s is added atomically into row s mod R (the consumer sums the
                        R rows: crog_bn_apply_stats) — no reduction launch between the GEMM and the normalisation. */
  float* a_sum;      /* NULL, or fp32 [M]: a_sum[m] += sum_k A(m, k), accumulated atomically by the blocks of the first
                        N-tile (every split adds its share).  With A = dy^T (CROG_A_MC) this is the bias gradient of the
                        nn.Linear / bias-conv whose weight gradient the GEMM computes (clip.py:249-251, layers.py:58,
                        298-301, ssg.py:123-133): no separate column-sum pass over dy.  batch must be 1. */
  const void* bwd_z; /* NULL, or [M][N] (dtype, row stride ldz): BatchNorm-BACKWARD statistics mode of col_stats, for a data-gradient
                        GEMM whose output is the gradient dy of a BatchNorm(+ReLU) layer with pre-normalisation activation z
                        (clip.py:44-57: conv1 -> bn1 -> relu -> conv2: conv2's data gradient IS bn1's dy).  The epilogue masks the
                        gradient, g = relu_gate(z) ? v : 0 with relu_gate(z) = z * bwd_ss[n][0] + bwd_ss[n][1] > 0 (all ones when
                        bwd_ss is NULL), STORES g instead of v, and accumulates (sum g, sum g * z) per column into col_stats —
                        the first pass of BatchNorm backward without re-reading dy (crog_bn_bwd_apply takes the raw z-moments
                        with a negative sum_rows).  Needs dtype output, stat_replicas > 0, even N; a residual R only as described under bwd_mask. */
  int64_t ldz;
  const float* bwd_ss; /* NULL, or fp32 [N][2]: (scale, shift) of the layer's normalisation, to recompute the ReLU gate from z */
  const unsigned char* bwd_mask; /* NULL, or (with bwd_z) the ReLU bit mask crog_bn_apply wrote in the forward of a RESIDUAL layer
                        (clip.py:52-57: out = relu(bn3(z) + identity)): [M][N / 8] bytes, bit e of byte (m, n / 8) = column 8 (n / 8) + e
                        passed.  The gate is then these bits (z alone cannot tell), and a residual R - the identity path's gradient of the
                        NEXT block, whose first convolution this data gradient belongs to - is added BEFORE the gate: the stored
                        g = mask ? acc + R : 0 is at once the layer's gated dy and the gradient of its own identity path.  N % 8 == 0. */
  const void* stat_sync; /* NULL, or (with bwd_z, stat_replicas = R > 0) the device block of a communicator (crog_comm_sync_block): col_stats is
                        then [R][N][2] replica rows FOLLOWED by [N][2] totals and one counter word (all zero before the launch), and the block that
                        finishes last adds the rows up, exchanges the 2 N sums with the other ranks through the peer mailboxes and stores the
                        totals - the SyncBatchNorm backward exchange (train_crog.py:113-114) without a launch of its own.  The consumer
                        (crog_bn_bwd_apply) is handed col_stats + R * 2 N with sum_rows = -1. */
} crog_gemm_desc;

int crog_gemm(const crog_gemm_desc* d, crog_stream_t stream);
/* number of 128-row tiles (= rows of the col_stats slab) for a given M */
int crog_gemm_stat_tiles(int M);
/* Split count for a weight-gradient GEMM (out_mode CROG_OUT_F32_ATOMIC) of logical size M x N over K, matched to the tile
 * shape crog_gemm selects for it (wgrad call sites: every conv / linear backward, e.g. clip.py:44-57, layers.py:298-301). */
int crog_gemm_splitk_hint(int dtype, int a_layout, int b_layout, int M, int N, int K);
/* out[m][n] (+)= sum over z < splits of ws[z][m][n]: the second stage of a split-K weight gradient launched with CROG_OUT_F32
 * (slabs of M rows x ldws floats; out has row stride ldo; accumulate != 0 adds onto out, as gradient accumulation does).  The slabs are
 * summed in a fixed order (one thread per four columns; many slabs of a small output: a run of slabs per wave, runs added in order): the same bits on every run, unlike the atomic form (CROG_DETERMINISTIC,
 * crog_engine.py:72-84 run twice gives the same loss curve).  N, ldws, ldo multiples of 4; ws and out 16-byte aligned. */
int crog_splitk_reduce(const float* ws, int splits, int M, int N, int64_t ldws, float* out, int64_t ldo, int accumulate,
                       crog_stream_t stream);
/* Tile edge (64 / 128 / 256) crog_gemm takes for that weight gradient.  The 256 x 256 tile has an atomic-only epilogue: a launch that
 * also asks for a_sum stays on 128 x 128, so a caller that wants the wide tile sums the bias gradient with crog_colsum instead. */
int crog_gemm_wgrad_tile(int dtype, int a_layout, int b_layout, int M, int N, int K);
/* Small-channel 3x3 weight gradients (stem, layer1: clip.py:16-19,156-161 backward): the number of slabs the sliding-window kernel
 * wants for dW[M][9 convC] over K pixels of convH x convW maps with dense [pixels][channels] bf16 operands, or 0 when crog_gemm would
 * not take that kernel.  A caller that gets n > 0 launches crog_gemm with out_mode CROG_OUT_F32, splitk = n into an [n][M][9 convC]
 * workspace and adds the slabs with crog_splitk_reduce: 37.7 MB of plain stores and one reduction launch instead of 9.4 M atomic adds
 * (64 -> 64 over 346112 pixels: 55 us against 89). */
int crog_wgrad_sw_slabs(int M, int convH, int convW, int convC, int K);
/* 1 when crog_gemm can run this descriptor with the BatchNorm-backward statistics epilogue (bwd_z): bf16, one of the three
 * data-gradient layouts, plain epilogue, operands addressable by the LDS-DMA path (32-bit byte offsets).  For callers that must
 * decide before the producing layer skips its own first pass. */
int crog_gemm_supports_bwd_z(const crog_gemm_desc* d);
/* n (1 .. 32) independent weight gradients in ONE launch of the ping-pong weight-gradient kernel (csrc/gemm_ppt.hip): every descriptor
 * bf16, CROG_A_MC x CROG_B_NC / CROG_B_NC_IM2COL, CROG_OUT_F32_ATOMIC, M and N multiples of 8, K >= 128, no bias / residual / a_sum /
 * statistics, its own splitk.  The blocks of all problems run side by side: the small weight gradients of consecutive layers (a 512 x 512
 * output is four tiles) fill the chip together at a split of 4 instead of each at a split of 16 (model/layers.py, clip.py: what autograd
 * does one layer at a time at crog_engine.py:87).  Same results as n crog_gemm calls up to the order of the fp32 atomic adds. */
int crog_gemm_group(const crog_gemm_desc* descs, int n, crog_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm, training mode with optional cross-replica statistics (nn.BatchNorm2d/1d under
 * SyncBatchNorm: clip.py:18,21,26,42,78,171-183; layers.py:11,16,351; train_crog.py:113-114).
 * Statistics travel as per-channel pairs [C][2]:
 *   forward  pairs = (sum x, sum x^2)       backward pairs = (sum g, sum g*xhat)
 * so that the caller can all-reduce `sums` across ranks between reduce and finalize/apply.
 * ---------------------------------------------------------------------------------------- */
int crog_bn_stat_blocks(int64_t M, int rows_per_block);
int crog_bn_partial_stats(int dtype, const void* x, int64_t M, int C, int64_t ld, int rows_per_block,
                          float* partial, crog_stream_t stream);
/* sums_is_zero != 0: `sums` was zeroed by the caller (the reduction accumulates with atomics) */
/* bn_apply with the statistics finalised inside the kernel: `sums` = [replicas][C][2] accumulated by a crog_gemm epilogue
 * (stat_replicas) or the all-reduced [C][2] totals of SyncBatchNorm; also stores scale/shift and (mean, invstd) for the backward
 * pass and updates the running statistics (nn.BatchNorm2d training forward, clip.py:18-26, layers.py:8-16). */
int crog_bn_apply_stats(int dtype, const void* z, int64_t ldz, const float* sums, int replicas, float count,
                        const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                        float eps, float* scale_shift, float* mean_invstd, const void* res, int64_t ldr, int relu, void* y,
                        int64_t ldy, int64_t M, int C, void* relu_mask, crog_stream_t stream);
int crog_reduce_pairs(const float* partial, int nparts, int C, float* sums, int sums_is_zero, crog_stream_t stream);
/* a[c] += sums[c][0]; b[c] += sums[c][1]: a and b are parameter-gradient vectors, and gradients accumulate until zero_grad */
int crog_split_pairs(const float* sums, int C, float* a, float* b, crog_stream_t stream);
int crog_bn_finalize(const float* sums, float count, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, int C,
                     float* scale_shift, float* mean_invstd, crog_stream_t stream);
int crog_bn_eval_scale(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int C, float* scale_shift,
                       crog_stream_t stream);
/* y = [relu](z*scale + shift [+ res])  — bn + residual add + ReLU of Bottleneck (clip.py:47-56).
 * relu_mask (optional, with relu): one byte per 16-byte vector of y, [M][C / (16 / sizeof(element))], bit e = element e > 0 —
 * what `out = self.relu(out)` (clip.py:56) keeps for backward, at 1/16 of y's bytes; crog_bn_apply_stats takes the same. */
int crog_bn_apply(int dtype, const void* z, int64_t ldz, const float* scale_shift, const void* res,
                  int64_t ldr, int relu, void* y, int64_t ldy, int64_t M, int C, void* relu_mask, crog_stream_t stream);
/* g = dy * mask;  mask = (y > 0) when y != NULL, the forward's bit mask when relu_mask != NULL, or (z*scale+shift > 0) when
 * relu_scale_shift != NULL (ReLU without a residual: the output is not re-read), else 1;
 * partial[block][C][2] = (sum g, sum g*xhat) */
int crog_bn_bwd_partial(int dtype, const void* dy, int64_t lddy, const void* y, int64_t ldy, const void* z,
                        int64_t ldz, const float* mean_invstd, const float* relu_scale_shift, int64_t M, int C,
                        int rows_per_block, float* partial, int replicas, const void* relu_mask, crog_stream_t stream);
/* The same pass (replicas = R > 0 only; H = W = 0, or the pooled form's map size) whose LAST block adds the R rows up, exchanges the 2 C
 * sums with the other ranks (stat_sync = crog_comm_sync_block's device block; NULL: no exchange, totals only) and stores them behind
 * the rows: partial = [R][C][2] rows, [C][2] totals, one counter word, all zero before the launch.  crog_bn_bwd_apply then takes
 * partial + R * 2 C with sum_rows = 1. */
int crog_bn_bwd_partial_sync(int dtype, const void* dy, int64_t lddy, const void* y, int64_t ldy, const void* z,
                             int64_t ldz, const float* mean_invstd, const float* relu_scale_shift, int64_t M, int C,
                             int rows_per_block, float* partial, int replicas, const void* relu_mask, int H, int W,
                             const void* stat_sync, crog_stream_t stream);
/* dz = gamma*invstd*(g - sums.g/count - xhat*sums.gx/count);  dres = g when dres != NULL.
 * replicas (partial) / sum_rows (apply) = 0: `partial` is the per-block slab [blocks][C][2] and `sums` the reduced [C][2].
 * replicas = R > 0: the partial kernel adds atomically into a PRE-ZEROED [R][C][2]; the apply kernel is handed the same buffer
 * with sum_rows = R, adds the rows up itself and (dgamma/dbeta != NULL) ADDS the parameter gradients times param_grad_scale to
 * dgamma/dbeta (gradient buffers accumulate until zero_grad, as torch's .grad does) — no reduction launch.  Under SyncBatchNorm the rows are all-reduced first, so the totals are GLOBAL sums: storing them with
 * param_grad_scale = 1/world gives every rank the value DDP's gradient averaging would have produced from the local sums
 * (mean over ranks of the local sums == global sum / world), and no separate local-sum pass is needed.
 * sum_rows < 0: |sum_rows| rows of RAW z-moments (sum g, sum g*z), as a data-gradient GEMM's epilogue accumulates them
 * (crog_gemm_desc.bwd_z); the kernel forms sum g*zhat = invstd * (sum g*z - mean * sum g) itself. */
int crog_bn_bwd_apply(int dtype, const void* dy, int64_t lddy, const void* y, int64_t ldy, const void* z,
                      int64_t ldz, const float* mean_invstd, const float* gamma, const float* sums,
                      float count, const float* relu_scale_shift, void* dz, int64_t lddz, void* dres, int64_t lddres,
                      int64_t M, int C, int sum_rows, float* dgamma, float* dbeta, float param_grad_scale,
                      const void* relu_mask, crog_stream_t stream);
/* BatchNorm + ReLU followed by the 2 x 2 average pooling of the reference's strided layers (clip.py:49-50: avgpool after bn2 + relu of a
 * strided Bottleneck; clip.py:213-214: the stem's) as ONE pass each way: the forward writes the POOLED map y [B][H/2][W/2][C] from z
 * [B][H][W][C] (M = B H W rows) - the full-resolution activation is never stored -, the two backward passes take the gradient of the
 * pooled map (each pixel's share is a quarter of its cell's, the ReLU gate is recomputed from z with relu_scale_shift) - no
 * pooling-backward launch and no full-resolution gradient in HBM.  Arguments as in the plain forms (no residual, no mask, no y). */
int crog_bn_apply_stats_pool(int dtype, const void* z, int64_t ldz, const float* sums, int replicas, float count, const float* gamma,
                             const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                             float* scale_shift, float* mean_invstd, int relu, void* y, int64_t ldy, int64_t M, int C, int H, int W,
                             crog_stream_t stream);
int crog_bn_bwd_partial_pool(int dtype, const void* dy_pooled, int64_t lddy, const void* z, int64_t ldz, const float* mean_invstd,
                             const float* relu_scale_shift, int64_t M, int C, int rows_per_block, float* partial, int replicas,
                             int H, int W, crog_stream_t stream);
int crog_bn_bwd_apply_pool(int dtype, const void* dy_pooled, int64_t lddy, const void* z, int64_t ldz, const float* mean_invstd,
                           const float* gamma, const float* sums, float count, const float* relu_scale_shift, void* dz, int64_t lddz,
                           int64_t M, int C, int sum_rows, float* dgamma, float* dbeta, float param_grad_scale, int H, int W,
                           crog_stream_t stream);
/* single-replica fast paths: slab [nparts][C][2] -> (reduce + finalize) / (reduce + split) in one launch; crog_reduce_split
 * stores the reduced pairs to `sums` (optional) and ADDS the two halves to the gradient vectors a / b (optional) */
int crog_bn_reduce_finalize(const float* partial, int nparts, float count, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, int C,
                            float* scale_shift, float* mean_invstd, crog_stream_t stream);
int crog_reduce_split(const float* partial, int nparts, int C, float* sums, float* a, float* b,
                      crog_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm (fp32 math; clip.py:226-231, layers.py:192,288-305) with the decoder's surrounding
 * dropout / residual / positional adds fused (layers.py:310-338):
 *   y = LN(dropout_in(x));  out = res + dropout_out(y);  out2 = out + pos[row % pos_rows]
 * res / out2 / pos may be NULL; stats[row] = (mean, rstd) is kept for backward.
 * ---------------------------------------------------------------------------------------- */
int crog_ln_fwd(int dtype, const void* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                int64_t M, int C, void* out, int64_t ldo, float* stats, const void* res, int64_t ldr,
                void* out2, int64_t ldo2, const void* pos, int pos_rows, int64_t ldp, float p_in,
                uint64_t seed_in, float p_out, uint64_t seed_out, crog_stream_t stream);
int crog_ln_bwd_blocks(int64_t M, int rows_per_block);
/* dx from g = dout (+ dout2).  Parameter gradients, one of two forms: partial[block][C][2] = per-block (dgamma, dbeta) sums for an
 * ordered reduction by crog_reduce_split (dgamma = dbeta = NULL; the bit-reproducible fp32 parity mode), or partial = NULL and the
 * blocks ADD their sums atomically into the gradient vectors dgamma[C] / dbeta[C] (no reduction launch).
 * dxadd (or NULL): the gradient of a residual branch that by-passes the norm (x -> LN(x) and x -> ... + x, clip.py:262-264,
 * layers.py:313-338); it is added to dx in fp32 here, which replaces autograd's accumulation pass over the two gradients of x. */
int crog_ln_bwd(int dtype, const void* dout, int64_t lddo, const void* dout2, int64_t lddo2, const void* x,
                int64_t ldx, const float* gamma, const float* stats, int64_t M, int C, void* dx,
                int64_t lddx, float* partial, int rows_per_block, float p_in, uint64_t seed_in,
                float p_out, uint64_t seed_out, float* dgamma, float* dbeta, const void* dxadd, int64_t lddxa,
                crog_stream_t stream);
/* The same for a LayerNorm whose input x is a ReLU output (layers.py:298-300: Linear -> ReLU -> Dropout -> LayerNorm in the decoder's
 * FFN): dx is also gated by x > 0, i.e. it is the gradient of the ReLU's INPUT - the producing Linear skips its activation-backward
 * pass (21632 x 2048: one read of dx and y and one write less per layer). */
int crog_ln_bwd_relu(int dtype, const void* dout, int64_t lddo, const void* dout2, int64_t lddo2, const void* x,
                int64_t ldx, const float* gamma, const float* stats, int64_t M, int C, void* dx,
                int64_t lddx, float* partial, int rows_per_block, float p_in, uint64_t seed_in,
                float p_out, uint64_t seed_out, float* dgamma, float* dbeta, const void* dxadd, int64_t lddxa,
                crog_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Masked softmax over attention scores S[batch*heads*Lq][ldp] in place (the softmax inside
 * F.multi_head_attention_forward: causal text mask clip.py:424-430, key_padding_mask
 * layers.py:332 / crog.py:55, attention dropout layers.py:291-296).  Pd = dropout(P) or NULL.
 * ---------------------------------------------------------------------------------------- */
int crog_softmax_fwd(int dtype, void* S, int64_t rows, int Lq, int Lk, int ldp, int heads, int causal,
                     const uint8_t* key_padding_mask, void* Pd, float p_drop, uint64_t seed,
                     crog_stream_t stream);
int crog_softmax_bwd(int dtype, const void* P, void* dPd, int64_t rows, int Lk, int ldp, float p_drop,
                     uint64_t seed, crog_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Streaming ops on channels-last maps [B,H,W,C] / row matrices [M,C]
 * ---------------------------------------------------------------------------------------- */
/* nn.AvgPool2d(2): clip.py:23,35,184; F.avg_pool2d layers.py:386 */
int crog_avgpool2_fwd(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int H, int W, int C,
                      crog_stream_t stream);
int crog_avgpool2_bwd(int dtype, const void* dy, int64_t lddy, void* dx, int64_t lddx, int B, int H, int W,
                      int C, crog_stream_t stream);
/* the same with dx = pool-backward(dy) + add (add may be NULL): the pooled map's input also feeds another branch whose gradient is
 * already there (layer2 / layer3 outputs feed the next stage's downsample AvgPool AND the neck, clip.py:219-222 + layers.py:373-386) */
int crog_avgpool2_bwd_add(int dtype, const void* dy, int64_t lddy, const void* add, int64_t ldadd, void* dx, int64_t lddx,
                          int B, int H, int W, int C, crog_stream_t stream);
/* bilinear x2, align_corners=False: layers.py:54,56,382,393.  (H, W) is the INPUT size. */
int crog_upsample2_fwd(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int H, int W, int C,
                       crog_stream_t stream);
int crog_upsample2_bwd(int dtype, const void* dy, int64_t lddy, void* dx, int64_t lddx, int B, int H, int W,
                       int C, crog_stream_t stream);
/* token_embedding(text) + positional_embedding[:L]: clip.py:440-443 */
int crog_embedding_fwd(int dtype, const int64_t* word, const void* tok, const void* pos, void* out,
                       int64_t rows, int L, int C, int vocab, crog_stream_t stream);
int crog_embedding_bwd(int dtype, const int64_t* word, const void* dout, float* dtok, float* dpos,
                       int64_t rows, int L, int C, int vocab, crog_stream_t stream);
/* x[arange(B), text.argmax(-1)]: clip.py:451-452 */
int crog_gather_rows(int dtype, const void* x, int64_t ldx, const int64_t* idx, void* out, int64_t ldo,
                     int64_t n, int C, crog_stream_t stream);
int crog_scatter_rows(int dtype, const void* dout, int64_t lddo, const int64_t* idx, void* dx, int64_t lddx,
                      int64_t n, int C, crog_stream_t stream);
/* f5 * state broadcast over pixels: layers.py:379 */
int crog_mul_bcast_fwd(int dtype, const void* x, int64_t ldx, const void* s, int64_t lds, void* z, int64_t ldz,
                       int B, int P, int C, crog_stream_t stream);
int crog_mul_bcast_bwd(int dtype, const void* dz, int64_t lddz, const void* x, int64_t ldx, const void* s,
                       int64_t lds, void* dx, int64_t lddx, void* ds, int64_t ldds, int B, int P, int C,
                       crog_stream_t stream);
/* out = a + b[row % brows]: residual adds and positional broadcasts (clip.py:117,443; layers.py:311) */
int crog_add_rows(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, int64_t brows, void* out,
                  int64_t ldo, int64_t M, int C, crog_stream_t stream);
/* out[r] (fp32) (+)= sum_b x[b*R + r]: gradient of a batch-broadcast positional table */
int crog_sum_over_batch(int dtype, const void* x, int64_t ldx, float* out, int64_t ldo, int B, int64_t R, int C,
                        int accumulate, crog_stream_t stream);
/* out = a + dropout(b) (a may be NULL: out = dropout(b)): layers.py:326,334,338 and their backward */
int crog_add_dropout(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo,
                     int64_t M, int C, float p, uint64_t seed, crog_stream_t stream);
/* mode 0: dx = dy*(y>0) (ReLU, y = output);  mode 1: QuickGELU backward, y = pre-activation */
int crog_act_bwd(int dtype, const void* dy, int64_t lddy, const void* y, int64_t ldy, void* dx, int64_t lddx,
                 int64_t M, int C, int mode, crog_stream_t stream);
/* Fused attention (no score matrix in HBM) for the unmasked bf16, head_dim = 64 case: decoder self-attention layers.py:291-296,324,
 * ViT blocks clip.py:246-260, attention pooling clip.py:119-139.  Element (b, l, head, d) of X lives at X + (b*L + l)*ldx + head*64 + d
 * (so Q/K/V may be column slices of one packed projection buffer).  lse[(b*heads + head)*Lq + q] = log-sum-exp of the scaled
 * scores; attention dropout uses the hash of crog_softmax_fwd with index row*ldp + key (ldp = the unfused path's padded row).
 * The backward needs O, dO, lse and a float workspace D of B*heads*Lq entries. */
int crog_flash_attn_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, void* O,
                        int64_t ldo, float* lse, int B, int heads, int Lq, int Lk, int head_dim, float scale,
                        float p_drop, uint64_t seed, int ldp, crog_stream_t stream);
int crog_flash_attn_bwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, const void* O,
                        int64_t ldo, const void* dO, int64_t lddo, const float* lse, float* D, void* dQ, int64_t lddq,
                        void* dK, int64_t lddk, void* dV, int64_t lddv, int B, int heads, int Lq, int Lk, int head_dim,
                        float scale, float p_drop, uint64_t seed, int ldp, crog_stream_t stream);
/* The same with the causal mask of the CLIP text transformer (clip.py:446-452 build_attention_mask; causal != 0: key k reaches query q only
 * if k <= q, self-attention only): the text tower's 20-token attention as one launch forward and two backward instead of three and five. */
int crog_flash_attn_fwd_masked(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, void* O,
                               int64_t ldo, float* lse, int B, int heads, int Lq, int Lk, int head_dim, float scale,
                               float p_drop, uint64_t seed, int ldp, int causal, crog_stream_t stream);
int crog_flash_attn_bwd_masked(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, const void* O,
                               int64_t ldo, const void* dO, int64_t lddo, const float* lse, float* D, void* dQ, int64_t lddq,
                               void* dK, int64_t lddk, void* dV, int64_t lddv, int B, int heads, int Lq, int Lk, int head_dim,
                               float scale, float p_drop, uint64_t seed, int ldp, int causal, crog_stream_t stream);
/* The same again with (a) a key padding mask - key_padding_mask (NULL = none): B*Lk bytes, non-zero = key (b, k) is padding and reaches no
 * query: the decoder's vision-to-text cross-attention layers.py:292-296,329-332 with crog.py:55's pad_mask (not together with causal) - and
 * (b) the forward's dropout decisions kept as a bit map: keep_bits (NULL = as above) is a caller-owned buffer of
 * B*heads*ceil(Lk/32)*Lq 32-bit words (4-byte aligned) that crog_flash_attn_fwd_bits fills when p_drop > 0 and crog_flash_attn_bwd_bits
 * reads instead of hashing (seed, index) again - same decisions, bit-identical gradients (layers.py:291-296: nn.MultiheadAttention(dropout=)
 * keeps its mask for the backward as well).  Word ((b*heads + head)*ceil(Lk/32) + t)*Lq + q: bit 16*h + r <-> key 32*t + (r&3) + 8*(r>>2) + 4*h. */
int crog_flash_attn_fwd_bits(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, void* O,
                             int64_t ldo, float* lse, int B, int heads, int Lq, int Lk, int head_dim, float scale,
                             float p_drop, uint64_t seed, int ldp, int causal, const void* key_padding_mask, void* keep_bits,
                             crog_stream_t stream);
int crog_flash_attn_bwd_bits(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, const void* O,
                             int64_t ldo, const void* dO, int64_t lddo, const float* lse, float* D, void* dQ, int64_t lddq,
                             void* dK, int64_t lddk, void* dV, int64_t lddv, int B, int heads, int Lq, int Lk, int head_dim,
                             float scale, float p_drop, uint64_t seed, int ldp, int causal, const void* key_padding_mask,
                             const void* keep_bits, crog_stream_t stream);
/* QuickGELU x*sigmoid(1.702x): clip.py:234-236 */
int crog_quickgelu_fwd(int dtype, const void* u, int64_t ldu, void* out, int64_t ldo, int64_t M, int C,
                       crog_stream_t stream);
/* stem conv1 (3->32, 3x3, stride 2, pad 1; clip.py:165-170): NCHW fp32 image -> patch rows
 * [B*(H/2)*(W/2)][32], column (ky*3+kx)*3+ci, columns 27..31 zero; the conv is then crog_gemm */
int crog_stem_im2col(int dtype, const float* img, void* out, int B, int H, int W, crog_stream_t stream);
/* ViT patch embedding front end (VisionTransformer.forward clip.py:309-321; conv1 = clip.py:290-294).
 * crog_patchify: NCHW fp32 image -> rows [B*(H/P)*(W/P)][P*P*3], column (ky*P+kx)*3+ci; conv1 is then crog_gemm.
 * crog_vit_tokens_fwd: out[b][0] = class_embedding + pos[0], out[b][1+i] = y[b*G+i] + pos[1+i]  (T = G+1 tokens).
 * crog_vit_tokens_bwd: dy[b*G+i] = dtok[b][1+i]; gpos[t] += sum_b dtok[b][t]; gcls += sum_b dtok[b][0] (fp32). */
int crog_patchify(int dtype, const float* img, void* out, int B, int H, int W, int P, crog_stream_t stream);
int crog_vit_tokens_fwd(int dtype, const void* y, int64_t ldy, const void* cls, const void* pos, void* out,
                        int B, int T, int C, crog_stream_t stream);
int crog_vit_tokens_bwd(int dtype, const void* dtok, void* dy, int64_t lddy, float* gcls, float* gpos,
                        int B, int T, int C, crog_stream_t stream);
/* ---- SSG trunk staging (BASELINE config 5; model/ssg.py) -------------------------------------------------------------
 * Strided / large-window convolutions (7x7 s2 stem ssg.py:63, 3x3 s2 ssg.py:22,188-191, 1x1 s2 ssg.py:79) run as
 * im2col rows -> crog_gemm (K = KH*KW*C, weights physically [Cout][ky][kx][ci]); the data gradient is crog_gemm
 * (dcol = dz W) -> col2im.  Column order (ky*KW + kx)*C + c; OH = floor((H + 2P - KH)/S) + 1. */
int crog_im2col_nhwc(int dtype, const void* x, int64_t ldx, void* col, int64_t ldo, int B, int H, int W, int C,
                     int KH, int KW, int S, int P, int OH, int OW, crog_stream_t stream);
int crog_col2im_nhwc(int dtype, const void* dcol, int64_t ldc, void* dx, int64_t lddx, int B, int H, int W, int C,
                     int KH, int KW, int S, int P, int OH, int OW, crog_stream_t stream);
/* NCHW fp32 image (C = 3 RGB or 4 RGB-D, ssg.py:217-222,250-253) -> patch rows, columns >= KH*KW*C zero up to ldo */
int crog_im2col_image(int dtype, const float* img, void* col, int64_t ldo, int B, int C, int H, int W,
                      int KH, int KW, int S, int P, int OH, int OW, crog_stream_t stream);
/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) ssg.py:66; argmax: one byte per output element (window position) */
int crog_maxpool3s2_fwd(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, void* argmax,
                        int B, int H, int W, int C, crog_stream_t stream);
int crog_maxpool3s2_bwd(int dtype, const void* dy, int64_t lddy, const void* argmax, void* dx, int64_t lddx,
                        int B, int H, int W, int C, crog_stream_t stream);
/* nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) ssg.py:159.  (H, W) is the INPUT size. */
int crog_upsample2ac_fwd(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int H, int W, int C,
                         crog_stream_t stream);
int crog_upsample2ac_bwd(int dtype, const void* dy, int64_t lddy, void* dx, int64_t lddx, int B, int H, int W,
                         int C, crog_stream_t stream);
/* Evaluation maps, engine/crog_engine.py:181-211 (validate_with_grasp) / :356-361 (validate_without_grasp): planar fp32 logits
 * x[B][G][h][w] -> y[B][G][H][W] = F.interpolate(mode='bicubic', align_corners=True)(sigmoid(x) where bit g of sigmoid_mask is
 * set, x elsewhere) — the reference applies sigmoid to the mask / quality / width maps and leaves sin / cos raw. */
int crog_eval_maps(const float* x, int B, int G, int h, int w, int sigmoid_mask, float* y, int H, int W,
                   crog_stream_t stream);
/* Data-gradient layout of 3x3 convolution weights, all convolutions of a model in one launch: for each table entry
 * (element offset, Cout, Cin) dst[off + (ci*9 + 8-tap)*Cout + co] = src[off + (co*9 + tap)*Cin + ci].  The data gradient of
 * F.conv2d(k=3, s=1, p=1) (clip.py:21,166-170; layers.py:8-11) is then crog_gemm(CROG_A_IM2COL, CROG_B_KC) on dy and this copy. */
int crog_conv3_dgrad_weights(int dtype, const void* src, void* dst, const int64_t* table, int count,
                             crog_stream_t stream);
/* The same for every weight of a model, 3x3 and 1x1 / linear alike: table entries (element offset, rows, cols, taps), taps = 9
 * (as above) or 1 - a plain transpose dst[off + c*rows + r] = src[off + r*cols + c], so that the data gradient of a Linear /
 * 1x1 convolution (dx = dy W) is a forward-shaped crog_gemm(CROG_A_KC, CROG_B_KC) on dy and the copy: both operands K-contiguous
 * (20-46 % faster on the text tower's 640-row launches, 0-12 % elsewhere, than reading W transposed out of LDS). */
int crog_dgrad_weights(int dtype, const void* src, void* dst, const int64_t* table, int count, crog_stream_t stream);
/* Eval-mode BatchNorm folded into the preceding convolution (validate_with_grasp runs model.eval(), crog_engine.py:133: BatchNorm
 * is the affine map y = z * sc + sh with sc = gamma / sqrt(running_var + eps), sh = beta - running_mean * sc, clip.py:18-26):
 *   w_dst[r][c] = (c < cols_src ? w_src[r][c] : 0) * sc[r / rows_per_channel]   for c < cols_dst   (fp32 master weights in, compute dtype out)
 *   bias_dst[ch] = sh[ch]
 * so that conv + BatchNorm (+ ReLU, + residual) is ONE crog_gemm with bias / act / R and the pre-normalisation map is never written.
 * rows_per_channel = 9 for a 3x3 weight stored [Cout][3][3][Cin] whose Cin is being zero-padded (rows = 9 * Cout), 1 otherwise. */
int crog_bn_fold_weights(int dtype_dst, const float* w_src, int64_t lds, int cols_src, int rows_per_channel, const float* gamma,
                         const float* beta, const float* running_mean, const float* running_var, float eps, void* w_dst, int64_t ldd,
                         int cols_dst, int64_t rows, float* bias_dst, crog_stream_t stream);
/* dst[r][c] = c < cols_src ? src[r][c] : 0 for c < cols_dst (fp32 source) */
int crog_cast_pad2d(int dtype_dst, const float* src, int64_t lds, int cols_src, void* dst, int64_t ldd,
                    int cols_dst, int64_t rows, crog_stream_t stream);
/* dst[r][c] += src[r][c] for c < cols (fp32): the gradient of a zero-padded compute copy of a ragged-Cin weight (stem 27 -> 32
 * columns, CoordConv 514 -> 544 channels; clip.py:165, layers.py:38-41) added back into the parameter's gradient. */
int crog_add_pad2d(const float* src, int64_t lds, float* dst, int64_t ldd, int cols, int64_t rows,
                   crog_stream_t stream);
int crog_cast_f32_to_bf16(const float* src, void* dst, int64_t n, crog_stream_t stream);
/* p[0 .. n) = 0 (fp32, 16-byte aligned): optimizer.zero_grad() on the flat gradient buffer (crog_engine.py:77) on the stream the caller
 * names - the weight-gradient stream during the forward pass, where the 588 MB of writes cost the main chain nothing */
int crog_zero_f32(float* p, int64_t n, crog_stream_t stream);
int crog_cast_to_f32(int dtype, const void* src, int64_t lds, float* dst, int64_t ldd, int64_t M, int C,
                     crog_stream_t stream);
/* CoordConv coordinate channels: layers.py:30-39 */
int crog_coord_fill(int dtype, void* buf, int64_t ld, int B, int H, int W, int c0, int cend,
                    crog_stream_t stream);
/* out[c] += sum_r x[r][c]: bias gradients of nn.Linear / nn.Conv2d(bias=True).  partial: fp32 workspace of
 * ceil(M / rows_per_block) * C floats (two-pass reduction, no contended atomics) */
int crog_colsum(int dtype, const void* x, int64_t ldx, int64_t M, int C, int rows_per_block, float* partial,
                float* out, crog_stream_t stream);
/* torch.optim.Adam step over one flat fp32 segment (train_crog.py:119-121, crog_engine.py:83) */
int crog_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, void* bf16_shadow, crog_stream_t stream);
/* The same step with its per-step scalars in device memory: hyper_dev = float[4] {lr, 1 - beta1^t, sqrt(1 - beta2^t), t}.
 * crog_adam_advance: t += 1 and the two corrections of the new t (one launch per parameter group, before its segments);
 * the host writes lr (MultiStepLR, train_crog.py:122,270) and, on load_state_dict, t. */
int crog_adam_advance(float* hyper_dev, float beta1, float beta2, crog_stream_t stream);
int crog_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev, float beta1,
                       float beta2, float eps, float weight_decay, void* bf16_shadow, crog_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Collectives of the data-parallel step (csrc/comm.hip).  Replace torch.distributed / NCCL at
 * train_crog.py:96-99 (init_process_group), :113-114 (SyncBatchNorm statistics), :154-156 (DDP gradient all-reduce) and
 * crog_engine.py:88-90 (metric all-reduce).  One process per GPU; every call below is issued by all ranks in the same order.
 *   crog_comm_unique_id  rank 0: 128 opaque bytes (ncclGetUniqueId) to hand to the other ranks by any side channel
 *   crog_comm_init       RCCL communicator over `world` ranks on the current device; id128 == NULL creates a communicator without
 *                        RCCL (peer mailbox only)
 *   crog_comm_peer_handle / _connect   the one-shot peer-write all-reduce for BatchNorm statistics: each rank allocates a mailbox
 *                        (uncached device memory, 2 x world slots of slot_floats floats) and gets a 64-byte hipIpc handle; after the
 *                        handles of all ranks (world x 64 bytes, rank order) were exchanged, _connect maps the peers' mailboxes
 *   crog_syncbn_stats    in-place fp32 SUM of ptr[0:count] over the ranks, asynchronous on `stream`: one single-block kernel per rank
 *                        (peer writes + flag poll, sums formed in rank order: bit-identical on all ranks) when count fits a mailbox
 *                        slot, ncclAllReduce otherwise.  Capture-safe (the exchange counter lives in the mailbox)
 *   crog_allreduce_bucket  in-place ncclAllReduce (SUM, or AVG when `average`) of a gradient bucket, fp32 or bf16, on `stream`
 *   crog_comm_status     *timed_out_seq != 0: an exchange gave up waiting for a peer (120 s by default, CROG_COMM_TIMEOUT_S) - the training state
 *                        is invalid: that exchange and every later one of the communicator return NaN statistics instead of local sums
 * RCCL is bound at run time (dlopen of the librccl.so already resident in the process; CROG_RCCL_LIB overrides): the library has no
 * link-time dependency on it. */
int crog_comm_unique_id(void* id128);
int crog_comm_init(int rank, int world, const void* id128, void** comm_out);
int crog_comm_peer_handle(void* comm, int slot_floats, void* handle64);
int crog_comm_peer_connect(void* comm, const void* handles);
/* Device-resident block (mailbox pointers, rank, world size, slot size, wait bound) that lets a kernel run an exchange in its own tail:
 * crog_gemm_desc.stat_sync, crog_bn_bwd_partial_sync.  Needs the peer mailboxes (crog_comm_peer_connect); owned by the communicator. */
int crog_comm_sync_block(void* comm, void** dev_block);
int crog_comm_status(void* comm, int* timed_out_seq);
int crog_syncbn_stats(void* comm, float* ptr, int64_t count, crog_stream_t stream);
int crog_allreduce_bucket(void* comm, void* ptr, int64_t count, int dtype, int average, crog_stream_t stream);
/* Schedule of crog_allreduce_bucket: 0 = one ncclAllReduce (default), 1 = ncclReduceScatter + ncclAllGather in place (the remainder of count
 * modulo the world size through a small ncclAllReduce).  Set the same value on every rank (crog_amd/rccl.py times both at start-up and
 * agrees on the faster one: SURVEY.md section 2c C1 asks for the two-phase form where a ring would be bound by one xGMI link). */
int crog_comm_set_bucket_algo(void* comm, int algo);
int crog_comm_destroy(void* comm);

/* ------------------------------------------------------------------------------------------
 * Text-conditioned dynamic conv head, losses, metric
 * (layers.py:64-132,152-173; crog.py:76-111,119-131; utils/misc.py:115-131)
 * ---------------------------------------------------------------------------------------- */
int crog_head_pack_weights(int dtype, const float* word, int64_t ldw, void* wpad, int B, int C,
                           crog_stream_t stream);
int crog_head_unpack_wgrad(int dtype, const float* dwpad, const float* dbias, void* dword, int64_t ldd, int B,
                           int C, crog_stream_t stream);
/* tbias: NULL, or fp32 [B][heads][16] constants added per valid source pixel and tap (the folded vis.4 bias below) */
int crog_head_stencil_fwd(const float* t, const float* word, int64_t ldw, int bias_col, const float* tbias, float* out,
                          int B, int heads, int H, int W, crog_stream_t stream);
/* Folding the 1x1 conv vis.4 (layers.py:58: in_dim -> heads*in_dim, bias b5) into the dynamic 3x3 head (layers.py:116-128): the
 * 1280-channel map is never materialised.  Per sample: Wf[h][tap][k] = sum_c W5[h*C+c][k] * w_b[c][tap] (a crog_gemm), then
 * t = x4 . Wf^T (a crog_gemm), and the conv bias enters as cb[b][h][tap] = sum_c b5[h*C+c] * w_b[c][tap] (crog_head_cb_fwd),
 * added by crog_head_stencil_fwd.  Backward: crog_head_tap_sums gives dcb[b][h][tap] = sum_pixels dt; crog_head_cb_bwd
 * accumulates db5 (atomic) and the cb share of dwpad. */
int crog_head_cb_fwd(int dtype, const float* b5, const void* wpad, float* cb, int B, int heads, int C,
                     crog_stream_t stream);
int crog_head_tap_sums(int dtype, const void* dt, float* dcb, int B, int heads, int64_t P, crog_stream_t stream);
int crog_head_cb_bwd(int dtype, const float* b5, const void* wpad, const float* dcb, float* db5, float* dwpad, int B,
                     int heads, int C, crog_stream_t stream);
int crog_head_stencil_bwd(int dtype, const float* dout, void* dt, float* dbias, int B, int heads, int H, int W,
                          crog_stream_t stream);
/* targets: host array of `heads` device pointers, each fp32 [B][1][Hin][Win] */
int crog_head_loss(const float* pred, const float* const* targets, int B, int heads, int H, int W, int Hin,
                   int Win, int weighted, float* tgt_small, float* loss_sums, float* dpred,
                   crog_stream_t stream);
int crog_train_metric(const float* pred, int64_t pred_bstride, const float* tgt, int B, int64_t P,
                      float threshold, float pr_iou, float* counts, float* out2, crog_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * SSG target assignment for a whole batch (model/ssg.py:317-321 looping utils/box_utils.py:57-117 `match` + `encode` per
 * image): anchors [A][4] (cx, cy, w, h), gt [B][Gmax][5] (corner box + class, rows >= ng[b] ignored), ng [B] >= 1.
 * Outputs per (image, anchor): SSD-encoded offsets [B][A][4], labels [B][A] (class of the matched box; -1 neutral below
 * pos_iou_thre; 0 background below neg_iou_thre), the matched box [B][A][4] and its index [B][A].  `claim` [B][Gmax] is
 * scratch (each box's best anchor).  Every box keeps its best anchor; a later box wins a contested anchor. */
int crog_ssg_match(const float* anchors, int A, const float* gt, const int* ng, int B, int Gmax, float pos_iou_thre,
                   float neg_iou_thre, int* claim, float* offsets, int64_t* labels, float* matched_box,
                   int64_t* matched_idx, crog_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training-input preprocessing (OCIDVLGDataset.preprocess, utils/dataset.py:843-914) for B same-sized uint8 samples:
 * img [B][H][W][3], masks [B][4][H][W] (instance, quality, angle in degrees, width) -> out_img [B][3][S][S] fp32 (letterbox
 * warp with cv2.warpAffine's INTER_CUBIC arithmetic, border = the CLIP mean, then (v / 255 - mean) / std) and out_masks
 * [B][5][S][S] fp32 (INTER_LINEAR warp, border 0: mask / 255, quality / 255, sin(2 theta), cos(2 theta), width / 255).
 * Host-side constants (HOST pointers, read before the launch): minv[6] = the inverted 2 x 3 letterbox matrix in double, mean[3],
 * stdv[3], border[3] (8-bit border value per channel).  tab_cubic [1024][16] / tab_linear [1024][4] are DEVICE int16 weight tables
 * (15-bit fixed point, OpenCV's initInterTab2D). */
int crog_preprocess_u8(const uint8_t* img, const uint8_t* masks, int B, int H, int W, const double* minv,
                       const int16_t* tab_cubic, const int16_t* tab_linear, int S, const float* mean, const float* stdv,
                       const int* border, float* out_img, float* out_masks, crog_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CROG_HIP_H */
