// BatchNorm (training, cross-replica capable), LayerNorm and masked softmax kernels.
// All are HBM-bound streaming kernels: 16-byte vector loads, fp32 math, wave-shuffle reductions.
#include "comm_dev.h"
#include <algorithm>

// A/B (round 6, scripts/build_variant.py bnprio -DCROG_CHAIN_PRIO=3): the streaming kernels of the main chain raise their wave priority, so that on
// a CU they share with a weight-gradient GEMM block of the side stream (which runs its MFMA phases at s_setprio 1) their loads issue first.
#ifdef CROG_CHAIN_PRIO
#define CHAIN_PRIO() __builtin_amdgcn_s_setprio(CROG_CHAIN_PRIO)
#else
#define CHAIN_PRIO()
#endif

namespace {

constexpr int NT = 256;

// =============================================================================================
// BatchNorm statistics.  x is [M][C] (row stride ld).  Each block reduces ROWS rows for all
// channels and writes one partial slab row: partial[block][C][2] = (sum, sum of squares).
// =============================================================================================
template <typename T>
__global__ void __launch_bounds__(NT) bn_partial_stats_kernel(const T* __restrict__ x, long M, int C, long ld,
                                                               int rows_per_block, float* __restrict__ partial) {
  constexpr int VEC = Elem<T>::VEC;
  __shared__ float red[NT][2 * VEC + 1];
  const int cvec = C / VEC;
  const int cw = cvec < NT ? cvec : NT;  // threads across channel chunks
  const int tx = threadIdx.x % cw, ty = threadIdx.x / cw, nty = NT / cw;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(r0 + rows_per_block, M);
  for (int cg = 0; cg < cvec; cg += cw) {
    const int c = (cg + tx) * VEC;
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) s1[e] = s2[e] = 0.f;
    for (long r = r0 + ty; r < r1; r += nty) {
      Vec16<T> v = ldg16(x + r * ld + c);
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        float f = Elem<T>::to_f(v.v[e]);
        s1[e] += f;
        s2[e] += f * f;
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; e++) {
      red[threadIdx.x][e] = s1[e];
      red[threadIdx.x][VEC + e] = s2[e];
    }
    __syncthreads();
    if (ty == 0) {
      for (int j = 1; j < nty; j++)
#pragma unroll
        for (int e = 0; e < VEC; e++) {
          s1[e] += red[j * cw + tx][e];
          s2[e] += red[j * cw + tx][VEC + e];
        }
      float* dst = partial + ((long)blockIdx.x * C + c) * 2;
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        dst[2 * e] = s1[e];
        dst[2 * e + 1] = s2[e];
      }
    }
    __syncthreads();
  }
}

// sums[c][2] = sum over parts of partial[part][c][2]  (any pair of per-channel quantities).
// Block = 32 channels (64 consecutive floats of a slab row) x 4 part-lanes; grid.y splits the part range;
// partial results meet in `sums` through fp32 atomics (sums is zeroed by the launcher).
__global__ void __launch_bounds__(NT) reduce_pairs_kernel(const float* __restrict__ partial, int nparts, int C, float* __restrict__ sums) {
  __shared__ float red[4][64];
  const int f = threadIdx.x & 63, lane_p = threadIdx.x >> 6;   // float index inside the 32-channel group, part lane
  const int col = blockIdx.x * 64 + f;                           // float column in a slab row of 2*C floats
  const int per = (nparts + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = min(p0 + per, nparts);
  float acc = 0.f;
  if (col < 2 * C)
    for (int p = p0 + lane_p; p < p1; p += 4) acc += partial[(long)p * 2 * C + col];
  red[lane_p][f] = acc;
  __syncthreads();
  if (lane_p == 0 && col < 2 * C) atomicAdd(sums + col, red[0][f] + red[1][f] + red[2][f] + red[3][f]);
}

// Single-launch reductions of a partial slab [nparts][C][2] (no atomics, no memset): block = SLAB_CH channels (2 SLAB_CH floats of a slab
// row) x SLAB_LANES part lanes; every lane adds its rows p = lane, lane + SLAB_LANES, ... in order, thread f < 2 SLAB_CH then adds the lanes'
// sums in lane order: a FIXED association, so the result is the same bits run after run.  Round 5: 2 channels x 64 lanes instead of 8 x 16
// - the launch sits between a convolution and its BatchNorm apply on the critical chain, and with 2704 slab rows (346112 pixels) and a grid
// of 8 blocks it took 21-23 us of pure latency, 140 launches per deterministic step.
constexpr int SLAB_CH = 2, SLAB_F = 2 * SLAB_CH, SLAB_LANES = NT / SLAB_F;
__device__ inline void reduce_slab(const float* __restrict__ partial, int nparts, int C, float (&red)[SLAB_LANES][SLAB_F + 1], float& out) {
  const int f = threadIdx.x % SLAB_F, lane_p = threadIdx.x / SLAB_F;
  const int col = blockIdx.x * SLAB_F + f;
  float acc = 0.f;
  if (col < 2 * C) {
    int p = lane_p;
    for (; p + 3 * SLAB_LANES < nparts; p += 4 * SLAB_LANES) {      // four independent loads in flight, added in row order
      const float v0 = partial[(long)p * 2 * C + col], v1 = partial[(long)(p + SLAB_LANES) * 2 * C + col];
      const float v2 = partial[(long)(p + 2 * SLAB_LANES) * 2 * C + col], v3 = partial[(long)(p + 3 * SLAB_LANES) * 2 * C + col];
      acc += v0;
      acc += v1;
      acc += v2;
      acc += v3;
    }
    for (; p < nparts; p += SLAB_LANES) acc += partial[(long)p * 2 * C + col];
  }
  red[lane_p][f] = acc;
  __syncthreads();
  out = 0.f;
  if (lane_p == 0) {
#pragma unroll 8
    for (int q = 0; q < SLAB_LANES; q++) out += red[q][f];
    red[0][f] = out;
  }
  __syncthreads();
}
// BN forward (single replica): slab -> sums -> scale/shift, mean/invstd, running statistics
// (this file is compiled with -ffp-contract=off: crog_amd/_lib.py EXTRA_FLAGS says why - y = z * scale + shift wants ROUNDED products)
__global__ void __launch_bounds__(NT) bn_reduce_finalize_kernel(const float* __restrict__ partial, int nparts, float count,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                                                                float eps, int C, float* __restrict__ scale_shift, float* __restrict__ mean_invstd) {
  __shared__ float red[SLAB_LANES][SLAB_F + 1];
  float dummy;
  reduce_slab(partial, nparts, C, red, dummy);
  const int t = threadIdx.x;
  const int c = blockIdx.x * SLAB_CH + t;
  if (t < SLAB_CH && c < C) {
    const float mean = red[0][2 * t] / count;
    float var = red[0][2 * t + 1] / count - mean * mean;
    var = fmaxf(var, 0.f);
    const float invstd = rsqrtf(var + eps);
    const float sc = gamma[c] * invstd;
    scale_shift[2 * c] = sc;
    scale_shift[2 * c + 1] = beta[c] - mean * sc;
    mean_invstd[2 * c] = mean;
    mean_invstd[2 * c + 1] = invstd;
    if (running_mean) {
      const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  }
}
// backward-side: slab -> sums[C][2] (optional) and the two per-channel vectors a[c] = sum of .x, b[c] = sum of .y
__global__ void __launch_bounds__(NT) reduce_split_kernel(const float* __restrict__ partial, int nparts, int C, float* __restrict__ sums,
                                                          float* __restrict__ a, float* __restrict__ b) {
  __shared__ float red[SLAB_LANES][SLAB_F + 1];
  float dummy;
  reduce_slab(partial, nparts, C, red, dummy);
  const int t = threadIdx.x;
  const int col = blockIdx.x * SLAB_F + t;
  if (t < SLAB_F && col < 2 * C) {
    const float v = red[0][t];
    if (sums) sums[col] = v;
    // a / b are gradient vectors of the flat buffer: ADD (torch accumulates gradients until zero_grad; a shared norm that is
    // applied several times per forward sums its uses)
    if (col & 1) { if (b) b[col >> 1] += v; }
    else { if (a) a[col >> 1] += v; }
  }
}

// Finalize training-mode BN from (global) sums: scale/shift for the apply pass, mean/invstd for
// backward, running-stat update with momentum (torch semantics: unbiased var in running_var).
__global__ void bn_finalize_kernel(const float* __restrict__ sums, float count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float momentum, float eps, int C,
                                   float* __restrict__ scale_shift, float* __restrict__ mean_invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mean = sums[2 * c] / count;
  float var = sums[2 * c + 1] / count - mean * mean;
  var = fmaxf(var, 0.f);
  const float invstd = rsqrtf(var + eps);
  const float sc = gamma[c] * invstd;
  scale_shift[2 * c] = sc;
  scale_shift[2 * c + 1] = beta[c] - mean * sc;
  mean_invstd[2 * c] = mean;
  mean_invstd[2 * c + 1] = invstd;
  if (running_mean) {
    const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// Eval-mode BN: scale/shift straight from running statistics.
__global__ void bn_eval_scale_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                     float eps, int C, float* __restrict__ scale_shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] * rsqrtf(running_var[c] + eps);
  scale_shift[2 * c] = sc;
  scale_shift[2 * c + 1] = beta[c] - running_mean[c] * sc;
}

// Eval-mode BN folded into the convolution weights (crog_bn_fold_weights): one thread per destination element
template <typename TD>
__global__ void __launch_bounds__(NT) bn_fold_weights_kernel(const float* __restrict__ src, long lds_, int cols_src, int rpc,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                                             float eps, TD* __restrict__ dst, long ldd, int cols_dst, long rows,
                                                             float* __restrict__ bias_dst) {
  const long total = rows * cols_dst;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const long r = i / cols_dst;
    const int c = (int)(i - r * cols_dst);
    const int ch = (int)(r / rpc);
    const float sc = gamma[ch] * rsqrtf(running_var[ch] + eps);
    const float v = c < cols_src ? src[r * lds_ + c] * sc : 0.f;
    dst[r * ldd + c] = Elem<TD>::from_f(v);
    if (c == 0 && r == (long)ch * rpc) bias_dst[ch] = beta[ch] - running_mean[ch] * sc;
  }
}

template <int VEC>
__device__ inline void ld_f32v(const float* __restrict__ p, float (&o)[VEC]) {   // p is 16-byte aligned (parameter blocks are)
#pragma unroll
  for (int i = 0; i < VEC / 4; i++) {
    const float4 t = *reinterpret_cast<const float4*>(p + 4 * i);
    o[4 * i] = t.x; o[4 * i + 1] = t.y; o[4 * i + 2] = t.z; o[4 * i + 3] = t.w;
  }
}

// interleaved per-channel pairs p[2*c], p[2*c+1] for VEC consecutive channels -> two arrays (32-byte aligned source)
template <int VEC>
__device__ inline void ld_pairs(const float* __restrict__ p, float (&a)[VEC], float (&b)[VEC]) {
#pragma unroll
  for (int i = 0; i < VEC / 2; i++) {
    const float4 t = *reinterpret_cast<const float4*>(p + 4 * i);
    a[2 * i] = t.x; b[2 * i] = t.y; a[2 * i + 1] = t.z; b[2 * i + 1] = t.w;
  }
}

// Row of a 2 x 2-average-pooled [B][H/2][W/2] map that pixel row r of the [B][H][W] map falls into (H, W even; r < 2^31)
__device__ inline long pooled_row(long r, int H, int W) {
  const unsigned rr = (unsigned)r, x = rr % (unsigned)W, t = rr / (unsigned)W, y = t % (unsigned)H, b = t / (unsigned)H;
  return (long)(b * (unsigned)(H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1);
}

// y = [relu]( z*scale + shift [+ res] )
template <typename T>
__global__ void __launch_bounds__(NT) bn_apply_kernel(const T* __restrict__ z, long ldz, const float* __restrict__ scale_shift,
                                                      const T* __restrict__ res, long ldr, int relu, T* __restrict__ y,
                                                      long ldy, long M, int C, unsigned char* __restrict__ relu_mask) {
  CHAIN_PRIO();
  constexpr int VEC = Elem<T>::VEC;
  const int cvec = C / VEC;
  const long total = M * cvec;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const long r = i / cvec;
    const int c = (int)(i % cvec) * VEC;
    Vec16<T> v = ldg16(z + r * ldz + c);
    Vec16<T> rv;
    if (res) rv = ldg16(res + r * ldr + c);
    Vec16<T> o;
    float sc[VEC], sh[VEC], f[VEC];
    ld_pairs<VEC>(scale_shift + 2 * c, sc, sh);
#pragma unroll
    for (int e = 0; e < VEC; e++) f[e] = Elem<T>::to_f(v.v[e]) * sc[e] + sh[e];
    if (res) {   // wave-uniform flags are tested once per vector, not per element
#pragma unroll
      for (int e = 0; e < VEC; e++) f[e] += Elem<T>::to_f(rv.v[e]);
    }
    if (relu) {
      if (relu_mask) {   // one byte per 16-byte vector: bit e = (pre-activation e > 0); backward reads this instead of y
        unsigned bits = 0;
#pragma unroll
        for (int e = 0; e < VEC; e++) bits |= (f[e] > 0.f ? 1u : 0u) << e;
        relu_mask[i] = (unsigned char)bits;
      }
#pragma unroll
      for (int e = 0; e < VEC; e++) f[e] = fmaxf(f[e], 0.f);
    }
#pragma unroll
    for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(f[e]);
    stg16(y + r * ldy + c, o);
  }
}

// Same, with the statistics finalised in the kernel: `sums` is the [R][C][2] (sum x, sum x^2) buffer a GEMM epilogue accumulated
// (crog_gemm_desc.stat_replicas) or, under SyncBatchNorm, the all-reduced [C][2] totals.  Every block derives scale/shift for
// all C channels into LDS (R*2C floats from L2); block 0 also stores scale/shift and (mean, invstd) for the backward pass and
// updates the running statistics.  Replaces reduce -> finalize -> apply (three launches) by one.
template <typename T>
__global__ void __launch_bounds__(NT) bn_apply_stats_kernel(const T* __restrict__ z, long ldz, const float* __restrict__ sums, int R, float count,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                                                            float eps, float* __restrict__ scale_shift, float* __restrict__ mean_invstd,
                                                            const T* __restrict__ res, long ldr, int relu, T* __restrict__ y, long ldy, long M,
                                                            int C, unsigned char* __restrict__ relu_mask, int poolH, int poolW) {
  CHAIN_PRIO();
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) float ss[];   // [C][2]
  const bool tiny = M <= 64 && count == (float)M && !poolW;      // local statistics (not SyncBatchNorm totals) over a handful of rows
  for (int c = threadIdx.x; c < C; c += NT) {
    float s = 0.f, q = 0.f;
    for (int r = 0; r < R; r++) {
      s += sums[((long)r * C + c) * 2];
      q += sums[((long)r * C + c) * 2 + 1];
    }
    float mean = s / count;
    float var = fmaxf(q / count - mean * mean, 0.f);
    if (tiny) {
      // BatchNorm1d over the rows of a batch (linear_layer, layers.py:14-16: M = B = 2 in config 1): E[x^2] - mean^2 cancels to the last bits
      // when a channel's two values are close, and the normalisation then amplifies that rounding ~60x (tests/test_fulldepth_gpu.py, B = 2).
      // With at most 64 rows the channel is simply read twice: mean, then the sum of squared deviations - what torch's two-pass kernel does.
      // (round 6: the rows are fetched eight at a time and then added in the SAME order - one dependent load per addition made this launch
      // 209 us for a 32 x 1024 tensor, on the forward's critical chain: every block derives all C channels, four per thread, two passes)
      float m = 0.f;
      for (long r0 = 0; r0 < M; r0 += 8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) t[j] = r0 + j < M ? Elem<T>::to_f(z[(r0 + j) * ldz + c]) : 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) m += t[j];
      }
      m /= (float)M;
      float v = 0.f;
      for (long r0 = 0; r0 < M; r0 += 8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) t[j] = r0 + j < M ? Elem<T>::to_f(z[(r0 + j) * ldz + c]) : m;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float d = t[j] - m;
          v += d * d;
        }
      }
      mean = m;
      var = v / (float)M;
    }
    const float invstd = rsqrtf(var + eps);
    const float sc = gamma[c] * invstd, sh = beta[c] - mean * sc;
    ss[2 * c] = sc;
    ss[2 * c + 1] = sh;
    if (tiny) {      // (the launcher sized the LDS for it) the apply loop below subtracts the mean FIRST: see there
      ss[2 * C + 2 * c] = mean;
      ss[2 * C + 2 * c + 1] = beta[c];
    }
    if (blockIdx.x == 0) {
      scale_shift[2 * c] = sc;
      scale_shift[2 * c + 1] = sh;
      mean_invstd[2 * c] = mean;
      mean_invstd[2 * c + 1] = invstd;
      if (running_mean) {
        const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
      }
    }
  }
  __syncthreads();
  const int cvec = C / VEC;
  if (poolW) {
    // ... followed by the 2 x 2 average pooling of the reference's strided layers (clip.py:49-50 avgpool after bn2 + relu, 213-214 the
    // stem's): y is the POOLED [B][H/2][W/2][C] map, one thread per output vector, its four pixels normalised (and rectified) in
    // registers - the full-resolution activation is never written or read back (the backward pass regates from z)
    const int Wo = poolW >> 1, Ho = poolH >> 1;
    const long total = (M >> 2) * cvec;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
      const unsigned ro = (unsigned)(i / cvec);
      const int c = (int)(i % cvec) * VEC;
      const unsigned xo = ro % (unsigned)Wo, t = ro / (unsigned)Wo, yo = t % (unsigned)Ho, b = t / (unsigned)Ho;
      const long r00 = ((long)(b * (unsigned)poolH + 2 * yo)) * poolW + 2 * xo;
      const Vec16<T> v0 = ldg16(z + r00 * ldz + c), v1 = ldg16(z + (r00 + 1) * ldz + c);
      const Vec16<T> v2 = ldg16(z + (r00 + poolW) * ldz + c), v3 = ldg16(z + (r00 + poolW + 1) * ldz + c);
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        const float sc = ss[2 * (c + e)], sh = ss[2 * (c + e) + 1];
        float a0 = Elem<T>::to_f(v0.v[e]) * sc + sh, a1 = Elem<T>::to_f(v1.v[e]) * sc + sh;
        float a2 = Elem<T>::to_f(v2.v[e]) * sc + sh, a3 = Elem<T>::to_f(v3.v[e]) * sc + sh;
        if (relu) { a0 = fmaxf(a0, 0.f); a1 = fmaxf(a1, 0.f); a2 = fmaxf(a2, 0.f); a3 = fmaxf(a3, 0.f); }
        o.v[e] = Elem<T>::from_f(0.25f * ((a0 + a1) + (a2 + a3)));
      }
      stg16(y + (long)ro * ldy + c, o);
    }
    return;
  }
  const long total = M * cvec;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const long r = i / cvec;
    const int c = (int)(i % cvec) * VEC;
    Vec16<T> v = ldg16(z + r * ldz + c);
    Vec16<T> rv;
    if (res) rv = ldg16(res + r * ldr + c);
    Vec16<T> o;
    float f[VEC];
    if (tiny) {
      // (z - mean) * scale + beta, the subtraction first (exact when z is close to the mean): over two samples every z IS close to the mean,
      // and z * scale + (beta - mean * scale) then rounds at the size of mean * scale, not of the deviation - config 1's B = 2 logits sat
      // 2.6e-3 from the float64 result with that form against the reference's 1.3e-3
#pragma unroll
      for (int e = 0; e < VEC; e++) f[e] = (Elem<T>::to_f(v.v[e]) - ss[2 * C + 2 * (c + e)]) * ss[2 * (c + e)] + ss[2 * C + 2 * (c + e) + 1];
    } else {
#pragma unroll
      for (int e = 0; e < VEC; e++) f[e] = Elem<T>::to_f(v.v[e]) * ss[2 * (c + e)] + ss[2 * (c + e) + 1];
    }
    if (res) {
#pragma unroll
      for (int e = 0; e < VEC; e++) f[e] += Elem<T>::to_f(rv.v[e]);
    }
    if (relu) {
      if (relu_mask) {   // one byte per 16-byte vector: bit e = (pre-activation e > 0); backward reads this instead of y
        unsigned bits = 0;
#pragma unroll
        for (int e = 0; e < VEC; e++) bits |= (f[e] > 0.f ? 1u : 0u) << e;
        relu_mask[i] = (unsigned char)bits;
      }
#pragma unroll
      for (int e = 0; e < VEC; e++) f[e] = fmaxf(f[e], 0.f);
    }
#pragma unroll
    for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(f[e]);
    stg16(y + r * ldy + c, o);
  }
}

// Backward pass 1: g = dy * (y > 0 if relu);  partial[block][C][2] = (sum g, sum g*zhat)
template <typename T>
__global__ void __launch_bounds__(NT) bn_bwd_partial_kernel(const T* __restrict__ dy, long lddy, const T* __restrict__ y, long ldy,
                                                            const T* __restrict__ z, long ldz, const float* __restrict__ mean_invstd,
                                                            const float* __restrict__ relu_ss, long M, int C, int rows_per_block,
                                                            float* __restrict__ partial, int replicas, const unsigned char* __restrict__ relu_mask,
                                                            int poolH, int poolW, const CrogSyncBlock* __restrict__ sync, int tail) {
  CHAIN_PRIO();
  // poolW != 0: dy is the gradient of the 2 x 2-average-POOLED output ([B][poolH/2][poolW/2][C]); pixel r takes a quarter of its cell's
  constexpr int VEC = Elem<T>::VEC;
  __shared__ float red[NT][2 * VEC + 1];
  const int cvec = C / VEC;
  const int cw = cvec < NT ? cvec : NT;
  const int tx = threadIdx.x % cw, ty = threadIdx.x / cw, nty = NT / cw;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(r0 + rows_per_block, M);
  for (int cg = 0; cg < cvec; cg += cw) {
    const int c = (cg + tx) * VEC;
    float s1[VEC], s2[VEC], mu[VEC], is[VEC], rsc[VEC], rsh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) s1[e] = s2[e] = rsc[e] = rsh[e] = 0.f;
    ld_pairs<VEC>(mean_invstd + 2 * c, mu, is);
    if (relu_ss) ld_pairs<VEC>(relu_ss + 2 * c, rsc, rsh);
    // (not unrolled: two / four rows in flight per thread cost a wave of occupancy - 135 / 205 VGPRs - and run the launch at 2.8-3.3 instead
    // of 3.7-5.1 TB/s; LAB_NOTES section 10)
    for (long r = r0 + ty; r < r1; r += nty) {
      Vec16<T> g = ldg16(dy + (poolW ? pooled_row(r, poolH, poolW) : r) * lddy + c);
      Vec16<T> zz = ldg16(z + r * ldz + c);
      Vec16<T> yy;
      if (y) yy = ldg16(y + r * ldy + c);
      float gf[VEC], zf[VEC];
#pragma unroll
      for (int e = 0; e < VEC; e++) { gf[e] = Elem<T>::to_f(g.v[e]); zf[e] = Elem<T>::to_f(zz.v[e]); }
      if (poolW) {
#pragma unroll
        for (int e = 0; e < VEC; e++) gf[e] *= 0.25f;
      }
      if (y) {
#pragma unroll
        for (int e = 0; e < VEC; e++) gf[e] = Elem<T>::to_f(yy.v[e]) > 0.f ? gf[e] : 0.f;
      }
      if (relu_mask) {   // residual layers: the forward's bit mask (1/16 of y's bytes)
        const unsigned bits = relu_mask[r * cvec + cg + tx];
#pragma unroll
        for (int e = 0; e < VEC; e++) gf[e] = ((bits >> e) & 1u) ? gf[e] : 0.f;
      }
      if (relu_ss) {   // ReLU mask recomputed from z (no residual): y is not read
#pragma unroll
        for (int e = 0; e < VEC; e++) gf[e] = zf[e] * rsc[e] + rsh[e] > 0.f ? gf[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        s1[e] += gf[e];
        s2[e] += gf[e] * ((zf[e] - mu[e]) * is[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; e++) {
      red[threadIdx.x][e] = s1[e];
      red[threadIdx.x][VEC + e] = s2[e];
    }
    __syncthreads();
    if (ty == 0) {
      for (int j = 1; j < nty; j++)
#pragma unroll
        for (int e = 0; e < VEC; e++) {
          s1[e] += red[j * cw + tx][e];
          s2[e] += red[j * cw + tx][VEC + e];
        }
      // park the block's sums in row tx of `red` (only this thread reads that row); they leave the block coalesced below
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        red[tx][e] = s1[e];
        red[tx][VEC + e] = s2[e];
      }
    }
    __syncthreads();
    {
      // consecutive lanes handle consecutive floats of the [C][2] row, i.e. 256 contiguous bytes per wave instruction: plain stores
      // into this block's slab row, or (replicas > 0) atomic adds into one of `replicas` pre-zeroed rows that the consumer sums
      // in-kernel - the shape the memory-side atomic units run at full rate
      const long row = replicas > 0 ? (long)(blockIdx.x % replicas) : (long)blockIdx.x;
      float* base = partial + (row * C + (long)cg * VEC) * 2;
      const int nf = min(cw, cvec - cg) * 2 * VEC;
      for (int k = threadIdx.x; k < nf; k += NT) {
        const int v = k / (2 * VEC), q = k % (2 * VEC);
        const float val = red[v][(q & 1) ? VEC + (q >> 1) : (q >> 1)];
        if (replicas > 0) atomicAdd(base + k, val);
        else base[k] = val;
      }
      __syncthreads();
    }
  }
  // tail != 0 (replicas > 0): `partial` carries 2 C floats of totals and a counter behind its rows - the block that finishes last adds
  // the rows up, exchanges the sums with the other ranks (sync != NULL: SyncBatchNorm) and stores the totals (comm_dev.h)
  if (tail) crog_stat_sync_tail(sync, partial, replicas, 2 * C, gridDim.x);
}

// dgamma/dbeta from the (local) sums: dbeta = sum g, dgamma = sum g*zhat.  accumulate flag adds.
__global__ void bn_param_grads_kernel(const float* __restrict__ sums, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] += sums[2 * c];
  dgamma[c] += sums[2 * c + 1];
}

// Backward pass 2: dz = gamma*invstd * (g - sum_g/count - zhat * sum_gz/count);  dres = g (optional)
template <typename T>
__global__ void __launch_bounds__(NT) bn_bwd_apply_kernel(const T* __restrict__ dy, long lddy, const T* __restrict__ y, long ldy,
                                                          const T* __restrict__ z, long ldz, const float* __restrict__ mean_invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ sums, float count,
                                                          const float* __restrict__ relu_ss, T* __restrict__ dz, long lddz, T* __restrict__ dres,
                                                          long lddres, long M, int C, int sum_rows, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, const unsigned char* __restrict__ relu_mask, float pgrad_scale,
                                                          int poolH, int poolW) {
  CHAIN_PRIO();
  constexpr int VEC = Elem<T>::VEC;
  const float gscale = poolW ? 0.25f : 1.f;      // pooled dy (see bn_bwd_partial_kernel): each pixel takes a quarter of its cell's gradient
  // sum_rows > 0: `sums` is [sum_rows][C][2] (atomic replicas of bn_bwd_partial, or the all-reduced totals): every block adds the
  // rows up into LDS once; block 0 also stores the parameter gradients (dbeta = sum g, dgamma = sum g*zhat) when asked to.
  // sum_rows < 0: |sum_rows| rows of RAW z-moments (sum g, sum g*z) from a data-gradient GEMM's epilogue (crog_gemm bwd_z):
  // sum g*zhat = invstd * (sum g*z - mean * sum g).
  extern __shared__ __attribute__((aligned(16))) float tot[];
  const bool raw = sum_rows < 0;
  if (raw) sum_rows = -sum_rows;
  if (sum_rows > 0) {
    for (int c = threadIdx.x; c < C; c += NT) {
      float a = 0.f, b = 0.f;
      for (int r = 0; r < sum_rows; r++) {
        a += sums[((long)r * C + c) * 2];
        b += sums[((long)r * C + c) * 2 + 1];
      }
      if (raw) b = mean_invstd[2 * c + 1] * (b - mean_invstd[2 * c] * a);
      tot[2 * c] = a;
      tot[2 * c + 1] = b;
      if (blockIdx.x == 0 && dgamma) {   // pgrad_scale = 1/world under SyncBatchNorm: the totals are global there (see header)
        dbeta[c] += a * pgrad_scale;     // gradient buffers accumulate (cleared by zero_grad)
        dgamma[c] += b * pgrad_scale;
      }
    }
    __syncthreads();
  }
  const int cvec = C / VEC;
  const long total = M * cvec;
  const float inv_count = 1.f / count;
  const long G = (long)gridDim.x * NT;
  if (!dres && (cvec & (cvec - 1)) == 0 && (G & (cvec - 1)) == 0) {
    // Streaming form for layers without a residual output (every BatchNorm of the ResNet trunks has a power-of-two C / VEC that
    // divides the thread count): a thread keeps ONE channel group for the whole launch, so its per-channel constants (mean, invstd,
    // gamma * invstd, sum_g / n, sum_gzhat / n, the ReLU gate's scale / shift) are registers and no 64-bit division is left per
    // vector: 346112 x 64: 37.8 -> 32.8 us, 1384448 x 32: 67.6 -> 58.6 us (scripts/bench_bn.py).  With the residual output (four
    // streams) the extra registers cost more occupancy than the arithmetic saves (126.9 -> 145.9 us): those keep the loop below,
    // as does bn_apply_stats (99.8 -> 99.1 us: not worth a second code path).
    const int lg = __ffs(cvec) - 1;
    const long i0 = (long)blockIdx.x * NT + threadIdx.x;
    const int c = (int)(i0 & (cvec - 1)) * VEC;
    float mu[VEC], is[VEC], sg[VEC], sgz[VEC], gm[VEC], rsc[VEC], rsh[VEC];
    ld_pairs<VEC>(mean_invstd + 2 * c, mu, is);
    if (sum_rows > 0) {
#pragma unroll
      for (int e = 0; e < VEC; e++) { sg[e] = tot[2 * (c + e)]; sgz[e] = tot[2 * (c + e) + 1]; }
    } else {
      ld_pairs<VEC>(sums + 2 * c, sg, sgz);
    }
    ld_f32v<VEC>(gamma + c, gm);
#pragma unroll
    for (int e = 0; e < VEC; e++) { sg[e] *= inv_count; sgz[e] *= inv_count; gm[e] *= is[e]; rsc[e] = 0.f; rsh[e] = 1.f; }
    if (relu_ss) ld_pairs<VEC>(relu_ss + 2 * c, rsc, rsh);
    const long rstep = G >> lg;
    auto one = [&](const Vec16<T>& g, const Vec16<T>& zz, const Vec16<T>& yy, unsigned bits, long r) {
      float gf[VEC], zf[VEC];
#pragma unroll
      for (int e = 0; e < VEC; e++) { gf[e] = Elem<T>::to_f(g.v[e]) * gscale; zf[e] = Elem<T>::to_f(zz.v[e]); }
      if (y) {
#pragma unroll
        for (int e = 0; e < VEC; e++) gf[e] = Elem<T>::to_f(yy.v[e]) > 0.f ? gf[e] : 0.f;
      }
      if (relu_mask) {
#pragma unroll
        for (int e = 0; e < VEC; e++) gf[e] = ((bits >> e) & 1u) ? gf[e] : 0.f;
      }
      if (relu_ss) {
#pragma unroll
        for (int e = 0; e < VEC; e++) gf[e] = zf[e] * rsc[e] + rsh[e] > 0.f ? gf[e] : 0.f;
      }
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        const float zh = (zf[e] - mu[e]) * is[e];
        o.v[e] = Elem<T>::from_f(gm[e] * (gf[e] - sg[e] - zh * sgz[e]));
      }
      stg16(dz + r * lddz + c, o);
    };
    long r = i0 >> lg, i = i0;
    for (; r < M; r += rstep, i += G) {
      // last readers of dy and z: non-temporal (a pooled dy row is read by four pixels: plain)
      Vec16<T> g = poolW ? ldg16(dy + pooled_row(r, poolH, poolW) * lddy + c) : ldg16_nt(dy + r * lddy + c), zz = ldg16_nt(z + r * ldz + c), yy;
      if (y) yy = ldg16(y + r * ldy + c);
      one(g, zz, yy, relu_mask ? relu_mask[i] : 0u, r);
    }
    return;
  }
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const long r = i / cvec;
    const int c = (int)(i % cvec) * VEC;
    Vec16<T> g = poolW ? ldg16(dy + pooled_row(r, poolH, poolW) * lddy + c) : ldg16_nt(dy + r * lddy + c);      // last readers of dy and z: non-temporal
    Vec16<T> zz = ldg16_nt(z + r * ldz + c);
    Vec16<T> yy;
    if (y) yy = ldg16(y + r * ldy + c);
    Vec16<T> o, gr;
    float gf[VEC], zf[VEC], mu[VEC], is[VEC], sg[VEC], sgz[VEC], gm[VEC];
    ld_pairs<VEC>(mean_invstd + 2 * c, mu, is);
    if (sum_rows > 0) {
#pragma unroll
      for (int e = 0; e < VEC; e++) { sg[e] = tot[2 * (c + e)]; sgz[e] = tot[2 * (c + e) + 1]; }
    } else {
      ld_pairs<VEC>(sums + 2 * c, sg, sgz);
    }
    ld_f32v<VEC>(gamma + c, gm);
#pragma unroll
    for (int e = 0; e < VEC; e++) { gf[e] = Elem<T>::to_f(g.v[e]) * gscale; zf[e] = Elem<T>::to_f(zz.v[e]); }
    if (y) {
#pragma unroll
      for (int e = 0; e < VEC; e++) gf[e] = Elem<T>::to_f(yy.v[e]) > 0.f ? gf[e] : 0.f;
    }
    if (relu_mask) {
      const unsigned bits = relu_mask[i];
#pragma unroll
      for (int e = 0; e < VEC; e++) gf[e] = ((bits >> e) & 1u) ? gf[e] : 0.f;
    }
    if (relu_ss) {
      float rsc[VEC], rsh[VEC];
      ld_pairs<VEC>(relu_ss + 2 * c, rsc, rsh);
#pragma unroll
      for (int e = 0; e < VEC; e++) gf[e] = zf[e] * rsc[e] + rsh[e] > 0.f ? gf[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < VEC; e++) {
      const float zh = (zf[e] - mu[e]) * is[e];
      o.v[e] = Elem<T>::from_f(gm[e] * is[e] * (gf[e] - sg[e] * inv_count - zh * sgz[e] * inv_count));
      gr.v[e] = Elem<T>::from_f(gf[e]);
    }
    stg16(dz + r * lddz + c, o);
    if (dres) stg16(dres + r * lddres + c, gr);
  }
}

// =============================================================================================
// LayerNorm: one wave per row.  Optional fused pieces (all follow layers.py:313-339 ordering):
//   xin  = dropout_in(x)                                   (ffn Dropout before LayerNorm, layers.py:299-300)
//   y    = LN(xin) * gamma + beta
//   out  = res + dropout_out(y)                            (vis + dropout(norm(attn)), layers.py:325-326)
//   out2 = out + pos[row % pos_rows]                       (with_pos_embed, layers.py:310-311,323)
// stats[row] = (mean, rstd) for backward.
// =============================================================================================
// NV = 16-byte vectors per lane (C <= 64 * VEC * NV): registers and instruction count scale with the row width instead of
// the 2048-wide worst case.  Each wave walks rows with a grid stride and issues the NEXT row's loads before it reduces the
// current one, so two rows of loads are always in flight per wave (a lone 16-byte load per lane is latency-, not
// bandwidth-bound: the single-row version measured 0.9 TB/s).
template <typename T, int NV>
__global__ void __launch_bounds__(NT) ln_fwd_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float eps, long M, int C, T* __restrict__ out,
                                                    long ldo, float* __restrict__ stats, const T* __restrict__ res, long ldr,
                                                    T* __restrict__ out2, long ldo2, const T* __restrict__ pos, int pos_rows, long ldp,
                                                    float p_in, uint64_t seed_in, float p_out, uint64_t seed_out,
                                                    const uint64_t* __restrict__ epoch) {
  CHAIN_PRIO();
  constexpr int VEC = Elem<T>::VEC;
  if (epoch) { const uint64_t e = *epoch; seed_in += e; seed_out += e; }     // crog_set_seed_epoch: per-step offset from device memory
  const int lane = threadIdx.x & 63;
  const long stride = (long)gridDim.x * (NT / 64);
  long row = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int cvec = C / VEC;
  const uint32_t thr_in = attn_thr16(p_in), thr_out = attn_thr16(p_out);      // 16-bit thresholds of the pair hash (dropout_apply, common.h)
  const float sc_in = p_in > 0.f ? 1.f / (1.f - p_in) : 1.f, sc_out = p_out > 0.f ? 1.f / (1.f - p_out) : 1.f;
  const float invC = 1.f / C;
  constexpr bool PF = NV <= 2;   // wide rows already keep >= 4 loads per lane in flight; prefetching them only costs registers
  constexpr int NP = PF ? NV : 1;
  Vec16<T> cur[NV], nxt[NP];
  if (PF) {
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int cv = lane + j * 64;
      if (cv < cvec) cur[j] = ldg16(x + row * ldx + cv * VEC);
    }
  }
  for (; row < M; row += stride) {
    const long nrow = row + stride;
    if (!PF) {
#pragma unroll
      for (int j = 0; j < NV; j++) {
        const int cv = lane + j * 64;
        if (cv < cvec) cur[j] = ldg16(x + row * ldx + cv * VEC);
      }
    }
    if (PF && nrow < M) {
#pragma unroll
      for (int j = 0; j < NV; j++) {
        const int cv = lane + j * 64;
        if (cv < cvec) nxt[j % NP] = ldg16(x + nrow * ldx + cv * VEC);
      }
    }
    Vec16<T> rv[NV], pv[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int cv = lane + j * 64;
      if (cv < cvec) {
        if (res) rv[j] = ldg16(res + row * ldr + cv * VEC);
        if (out2 && pos) pv[j] = ldg16(pos + (row % pos_rows) * ldp + cv * VEC);
      }
    }
    float vals[NV][VEC];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int cv = lane + j * 64;
      if (cv < cvec) {
#pragma unroll
        for (int e = 0; e < VEC; e++) vals[j][e] = Elem<T>::to_f(cur[j].v[e]);
        if (p_in > 0.f) {   // wave-uniform: hoisted around the whole vector so the common p = 0 path stays branch-free
          dropout_apply<VEC>(vals[j], seed_in, (uint64_t)row * C + cv * VEC, thr_in, sc_in);
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) s += vals[j][e];
      }
    }
    const float mean = wave_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int cv = lane + j * 64;
      if (cv < cvec)
#pragma unroll
        for (int e = 0; e < VEC; e++) {
          const float d = vals[j][e] - mean;
          q += d * d;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) * invC + eps);
    if (lane == 0 && stats) {
      stats[2 * row] = mean;
      stats[2 * row + 1] = rstd;
    }
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int cv = lane + j * 64;
      if (cv < cvec) {
        const int c = cv * VEC;
        Vec16<T> o;
        float gv[VEC], bv[VEC], f[VEC];
        ld_f32v<VEC>(gamma + c, gv);
        ld_f32v<VEC>(beta + c, bv);
#pragma unroll
        for (int e = 0; e < VEC; e++) f[e] = (vals[j][e] - mean) * rstd * gv[e] + bv[e];
        if (p_out > 0.f) {
          dropout_apply<VEC>(f, seed_out, (uint64_t)row * C + c, thr_out, sc_out);
        }
        if (res) {
#pragma unroll
          for (int e = 0; e < VEC; e++) f[e] += Elem<T>::to_f(rv[j].v[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(f[e]);
        stg16(out + row * ldo + c, o);
        if (out2) {
          Vec16<T> o2;
#pragma unroll
          for (int e = 0; e < VEC; e++) o2.v[e] = Elem<T>::from_f(Elem<T>::to_f(o.v[e]) + (pos ? Elem<T>::to_f(pv[j].v[e]) : 0.f));
          stg16(out2 + row * ldo2 + c, o2);
        }
      }
    }
    if (PF) {
#pragma unroll
      for (int j = 0; j < NV; j++) cur[j] = nxt[j % NP];
    }
  }
}

// LayerNorm backward.  g = dout (+ dout2);  dy = dropout_out_bwd(g);  dxin via LN backward;
// dx = dropout_in_bwd(dxin) (+ dxadd: the gradient of a residual branch that by-passes the norm, summed in fp32 here instead of by
// an autograd accumulation pass).  Per-block partial (dgamma, dbeta) rows go to partial[block][C][2].
template <typename T, int NV>
__global__ void __launch_bounds__(NT) ln_bwd_kernel(const T* __restrict__ dout, long lddo, const T* __restrict__ dout2, long lddo2,
                                                    const T* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                    const float* __restrict__ stats, long M, int C, T* __restrict__ dx, long lddx,
                                                    float* __restrict__ partial, int rows_per_block, float p_in, uint64_t seed_in,
                                                    float p_out, uint64_t seed_out, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                    const uint64_t* __restrict__ epoch, const T* __restrict__ dxadd, long lddxa, int relu_in) {
  CHAIN_PRIO();
  constexpr int VEC = Elem<T>::VEC;
  if (epoch) { const uint64_t e = *epoch; seed_in += e; seed_out += e; }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cvec = C / VEC;
  const uint32_t thr_in = attn_thr16(p_in), thr_out = attn_thr16(p_out);      // 16-bit thresholds of the pair hash (dropout_apply, common.h)
  const float sc_in = p_in > 0.f ? 1.f / (1.f - p_in) : 1.f, sc_out = p_out > 0.f ? 1.f / (1.f - p_out) : 1.f;
  const float invC = 1.f / C;
  constexpr bool PF = NV <= 2;   // see ln_fwd_kernel
  constexpr int NP = PF ? NV : 1;
  float dg[NV][VEC], db[NV][VEC], gm[NP][VEC];
#pragma unroll
  for (int j = 0; j < NV; j++) {
    const int cv = lane + j * 64;
#pragma unroll
    for (int e = 0; e < VEC; e++) dg[j][e] = db[j][e] = 0.f;
    if (PF && cv < cvec) ld_f32v<VEC>(gamma + cv * VEC, gm[j % NP]);
  }
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(r0 + rows_per_block, M);
  long row = r0 + wv;
  Vec16<T> cg[NV], cg2[NV], cx[NV], ca[NV], ng[NP], ng2[NP], nx[NP], na[NP];
  if (PF && row < r1) {
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int cv = lane + j * 64;
      if (cv < cvec) {
        cg[j] = ldg16(dout + row * lddo + cv * VEC);
        if (dout2) cg2[j] = ldg16(dout2 + row * lddo2 + cv * VEC);
        cx[j] = ldg16(x + row * ldx + cv * VEC);
        if (dxadd) ca[j] = ldg16(dxadd + row * lddxa + cv * VEC);
      }
    }
  }
  for (; row < r1; row += NT / 64) {
    const long nrow = row + NT / 64;
    if (!PF) {
#pragma unroll
      for (int j = 0; j < NV; j++) {
        const int cv = lane + j * 64;
        if (cv < cvec) {
          cg[j] = ldg16(dout + row * lddo + cv * VEC);
          if (dout2) cg2[j] = ldg16(dout2 + row * lddo2 + cv * VEC);
          cx[j] = ldg16(x + row * ldx + cv * VEC);
          if (dxadd) ca[j] = ldg16(dxadd + row * lddxa + cv * VEC);
        }
      }
    }
    if (PF && nrow < r1) {
#pragma unroll
      for (int j = 0; j < NV; j++) {
        const int cv = lane + j * 64;
        if (cv < cvec) {
          ng[j % NP] = ldg16(dout + nrow * lddo + cv * VEC);
          if (dout2) ng2[j % NP] = ldg16(dout2 + nrow * lddo2 + cv * VEC);
          nx[j % NP] = ldg16(x + nrow * ldx + cv * VEC);
          if (dxadd) na[j % NP] = ldg16(dxadd + nrow * lddxa + cv * VEC);
        }
      }
    }
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float xh[NV][VEC], gy[NV][VEC];
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int cv = lane + j * 64;
      if (cv < cvec) {
        const int c = cv * VEC;
        float gf[VEC], xf[VEC], gmv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e++) { gf[e] = Elem<T>::to_f(cg[j].v[e]); xf[e] = Elem<T>::to_f(cx[j].v[e]); }
        if (dout2) {
#pragma unroll
          for (int e = 0; e < VEC; e++) gf[e] += Elem<T>::to_f(cg2[j].v[e]);
        }
        if (p_out > 0.f) {
          dropout_apply<VEC>(gf, seed_out, (uint64_t)row * C + c, thr_out, sc_out);
        }
        if (p_in > 0.f) {
          dropout_apply<VEC>(xf, seed_in, (uint64_t)row * C + c, thr_in, sc_in);
        }
        if (PF) {
#pragma unroll
          for (int e = 0; e < VEC; e++) gmv[e] = gm[j % NP][e];
        } else {
          ld_f32v<VEC>(gamma + c, gmv);
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) {
          const float h = (xf[e] - mean) * rstd;
          xh[j][e] = h;
          dg[j][e] += gf[e] * h;
          db[j][e] += gf[e];
          const float gyv = gf[e] * gmv[e];
          gy[j][e] = gyv;
          a += gyv;
          b += gyv * h;
        }
      }
    }
    a = wave_sum(a) * invC;
    b = wave_sum(b) * invC;
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int cv = lane + j * 64;
      if (cv < cvec) {
        const int c = cv * VEC;
        Vec16<T> o;
        float d[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e++) d[e] = rstd * (gy[j][e] - a - xh[j][e] * b);
        if (p_in > 0.f) {
          dropout_apply<VEC>(d, seed_in, (uint64_t)row * C + c, thr_in, sc_in);
        }
        if (relu_in) {      // x is a ReLU output (layers.py:298-300: Linear -> ReLU -> Dropout -> LayerNorm): its backward rides along
#pragma unroll
          for (int e = 0; e < VEC; e++) d[e] = Elem<T>::to_f(cx[j].v[e]) > 0.f ? d[e] : 0.f;
        }
        if (dxadd) {
#pragma unroll
          for (int e = 0; e < VEC; e++) d[e] += Elem<T>::to_f(ca[j].v[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(d[e]);
        stg16(dx + row * lddx + c, o);
      }
    }
    if (PF) {
#pragma unroll
      for (int j = 0; j < NV; j++) { cg[j] = ng[j % NP]; cg2[j] = ng2[j % NP]; cx[j] = nx[j % NP]; ca[j] = na[j % NP]; }
    }
  }
  // combine the block's 4 waves through LDS, then one plain store per value into this block's partial row — or, with dgamma / dbeta
  // given, add the block's sums straight into the gradient vectors (a wave's 64 atomics are two runs of 32 consecutive floats: the
  // full-rate shape), which saves the reduction launch that followed every LayerNorm backward
  __shared__ float red[NT / 64][2 * 2048 / 8 + 1];
  float* dst = partial ? partial + (long)blockIdx.x * C * 2 : nullptr;
  for (int cg_ = 0; cg_ < 2 * C; cg_ += 2 * 2048 / 8) {  // 512 floats (256 channels) per round keeps LDS at 8 KB
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int cv = lane + j * 64;
      if (cv < cvec)
#pragma unroll
        for (int e = 0; e < VEC; e++) {
          const int f = 2 * (cv * VEC + e) - cg_;
          if (f >= 0 && f < 2 * 2048 / 8) { red[wv][f] = dg[j][e]; red[wv][f + 1] = db[j][e]; }
        }
    }
    __syncthreads();
    if (dgamma) {
      // lanes 0..31 of a wave take the dgamma halves of 32 consecutive channels, lanes 32..63 the dbeta halves
      for (int q = threadIdx.x; q < 2 * 2048 / 8; q += NT) {
        const int half = (q >> 5) & 1, ch = ((q >> 6) << 5) | (q & 31);      // channel inside this round, 32-channel groups
        const int f = 2 * ch + half;
        if (cg_ + 2 * ch < 2 * C) {
          const float v = red[0][f] + red[1][f] + red[2][f] + red[3][f];
          atomicAdd((half ? dbeta : dgamma) + (cg_ >> 1) + ch, v);
        }
      }
    } else {
      for (int f = threadIdx.x; f < 2 * 2048 / 8 && cg_ + f < 2 * C; f += NT)
        dst[cg_ + f] = red[0][f] + red[1][f] + red[2][f] + red[3][f];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over WIDE rows (more than 128 16-byte vectors: the decoder's LayerNorm(2048) inside the FFN, layers.py:299-300): one row
// per BLOCK iteration, thread t owns the vectors t + 256 j.  The wave-per-row kernels above need 184 (forward) / 268 (backward)
// VGPRs at this width - one wave per SIMD, nothing to hide the load latency behind: 21632 x 2048 bf16 ran at 1.8 / 1.07 TB/s.  Here a
// thread holds one vector per tensor (and the next row's, prefetched), the row's two sums cross the four waves through LDS - one
// barrier per row, slots alternating by row parity - and each thread owns its columns' (dgamma, dbeta) for the whole block.
// Same arguments and results as ln_fwd_kernel / ln_bwd_kernel (the order of the fp32 row sums differs).
// ---------------------------------------------------------------------------------------------
template <typename T, int NVT>
__global__ void __launch_bounds__(NT) ln_fwd_row_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, long M, int C, T* __restrict__ out,
                                                        long ldo, float* __restrict__ stats, const T* __restrict__ res, long ldr,
                                                        T* __restrict__ out2, long ldo2, const T* __restrict__ pos, int pos_rows, long ldp,
                                                        float p_in, uint64_t seed_in, float p_out, uint64_t seed_out,
                                                        const uint64_t* __restrict__ epoch) {
  CHAIN_PRIO();
  constexpr int VEC = Elem<T>::VEC, NW = NT / 64;
  if (epoch) { const uint64_t e = *epoch; seed_in += e; seed_out += e; }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cvec = C / VEC;
  const uint32_t thr_in = attn_thr16(p_in), thr_out = attn_thr16(p_out);
  const float sc_in = p_in > 0.f ? 1.f / (1.f - p_in) : 1.f, sc_out = p_out > 0.f ? 1.f / (1.f - p_out) : 1.f;
  const float invC = 1.f / C;
  __shared__ float red[2][NW][2];
  float nel[NW];      // elements of the row each wave owns (a row narrower than 256 NVT vectors leaves the last waves short)
#pragma unroll
  for (int w = 0; w < NW; w++) {
    int n = 0;
#pragma unroll
    for (int j = 0; j < NVT; j++) n += min(max(cvec - (w * 64 + j * NT), 0), 64);
    nel[w] = (float)(n * VEC);
  }
  long row = blockIdx.x;
  Vec16<T> cur[NVT], nxt[NVT];
#pragma unroll
  for (int j = 0; j < NVT; j++) {
    const int cv = threadIdx.x + j * NT;
    if (row < M && cv < cvec) cur[j] = ldg16(x + row * ldx + cv * VEC);
  }
  for (int par = 0; row < M; row += gridDim.x, par ^= 1) {
    const long nrow = row + gridDim.x;
    Vec16<T> rv[NVT], pv[NVT];
#pragma unroll
    for (int j = 0; j < NVT; j++) {
      const int cv = threadIdx.x + j * NT;
      if (cv < cvec) {
        if (nrow < M) nxt[j] = ldg16(x + nrow * ldx + cv * VEC);
        if (res) rv[j] = ldg16(res + row * ldr + cv * VEC);
        if (out2 && pos) pv[j] = ldg16(pos + (row % pos_rows) * ldp + cv * VEC);
      }
    }
    float vals[NVT][VEC];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NVT; j++) {
      const int cv = threadIdx.x + j * NT;
      if (cv < cvec) {
#pragma unroll
        for (int e = 0; e < VEC; e++) vals[j][e] = Elem<T>::to_f(cur[j].v[e]);
        if (p_in > 0.f) dropout_apply<VEC>(vals[j], seed_in, (uint64_t)row * C + cv * VEC, thr_in, sc_in);
#pragma unroll
        for (int e = 0; e < VEC; e++) s += vals[j][e];
      }
    }
    // per-wave (sum, centred sum of squares), combined exactly: Q = sum_w [ q_w + n_w (m_w - mean)^2 ]
    const float sw = wave_sum(s);
    const float mw = nel[wv] > 0.f ? sw / nel[wv] : 0.f;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NVT; j++) {
      const int cv = threadIdx.x + j * NT;
      if (cv < cvec)
#pragma unroll
        for (int e = 0; e < VEC; e++) {
          const float d = vals[j][e] - mw;
          q += d * d;
        }
    }
    const float qw = wave_sum(q);
    if (lane == 0) { red[par][wv][0] = sw; red[par][wv][1] = qw; }
    __syncthreads();
    float S = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) S += red[par][w][0];
    const float mean = S * invC;
    float Q = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const float d = (nel[w] > 0.f ? red[par][w][0] / nel[w] : mean) - mean;
      Q += red[par][w][1] + nel[w] * d * d;
    }
    const float rstd = rsqrtf(Q * invC + eps);
    if (threadIdx.x == 0 && stats) {
      stats[2 * row] = mean;
      stats[2 * row + 1] = rstd;
    }
#pragma unroll
    for (int j = 0; j < NVT; j++) {
      const int cv = threadIdx.x + j * NT;
      if (cv < cvec) {
        const int c = cv * VEC;
        Vec16<T> o;
        float gv[VEC], bv[VEC], f[VEC];
        ld_f32v<VEC>(gamma + c, gv);
        ld_f32v<VEC>(beta + c, bv);
#pragma unroll
        for (int e = 0; e < VEC; e++) f[e] = (vals[j][e] - mean) * rstd * gv[e] + bv[e];
        if (p_out > 0.f) dropout_apply<VEC>(f, seed_out, (uint64_t)row * C + c, thr_out, sc_out);
        if (res) {
#pragma unroll
          for (int e = 0; e < VEC; e++) f[e] += Elem<T>::to_f(rv[j].v[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(f[e]);
        stg16(out + row * ldo + c, o);
        if (out2) {
          Vec16<T> o2;
#pragma unroll
          for (int e = 0; e < VEC; e++) o2.v[e] = Elem<T>::from_f(Elem<T>::to_f(o.v[e]) + (pos ? Elem<T>::to_f(pv[j].v[e]) : 0.f));
          stg16(out2 + row * ldo2 + c, o2);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NVT; j++) cur[j] = nxt[j];
  }
}

template <typename T, int NVT>
__global__ void __launch_bounds__(NT) ln_bwd_row_kernel(const T* __restrict__ dout, long lddo, const T* __restrict__ dout2, long lddo2,
                                                        const T* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                        const float* __restrict__ stats, long M, int C, T* __restrict__ dx, long lddx,
                                                        float* __restrict__ partial, int rows_per_block, float p_in, uint64_t seed_in,
                                                        float p_out, uint64_t seed_out, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                        const uint64_t* __restrict__ epoch, const T* __restrict__ dxadd, long lddxa, int relu_in) {
  CHAIN_PRIO();
  constexpr int VEC = Elem<T>::VEC, NW = NT / 64;
  if (epoch) { const uint64_t e = *epoch; seed_in += e; seed_out += e; }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cvec = C / VEC;
  const uint32_t thr_in = attn_thr16(p_in), thr_out = attn_thr16(p_out);
  const float sc_in = p_in > 0.f ? 1.f / (1.f - p_in) : 1.f, sc_out = p_out > 0.f ? 1.f / (1.f - p_out) : 1.f;
  const float invC = 1.f / C;
  __shared__ float red[2][NW][2];
  __shared__ float colsum[2][NVT * NT * VEC];      // (dgamma, dbeta) of the block, staged for the contiguous atomic adds
  float dg[NVT][VEC], db[NVT][VEC], gm[NVT][VEC];
#pragma unroll
  for (int j = 0; j < NVT; j++) {
    const int cv = threadIdx.x + j * NT;
#pragma unroll
    for (int e = 0; e < VEC; e++) dg[j][e] = db[j][e] = gm[j][e] = 0.f;
    if (cv < cvec) ld_f32v<VEC>(gamma + cv * VEC, gm[j]);
  }
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(r0 + rows_per_block, M);
  Vec16<T> cg[NVT], cg2[NVT], cx[NVT], ca[NVT], ng[NVT], ng2[NVT], nx[NVT], na[NVT];
#pragma unroll
  for (int j = 0; j < NVT; j++) {
    const int cv = threadIdx.x + j * NT;
    if (r0 < r1 && cv < cvec) {
      cg[j] = ldg16(dout + r0 * lddo + cv * VEC);
      if (dout2) cg2[j] = ldg16(dout2 + r0 * lddo2 + cv * VEC);
      cx[j] = ldg16(x + r0 * ldx + cv * VEC);
      if (dxadd) ca[j] = ldg16(dxadd + r0 * lddxa + cv * VEC);
    }
  }
  int par = 0;
  for (long row = r0; row < r1; row++, par ^= 1) {
    const long nrow = row + 1;
    if (nrow < r1) {
#pragma unroll
      for (int j = 0; j < NVT; j++) {
        const int cv = threadIdx.x + j * NT;
        if (cv < cvec) {
          ng[j] = ldg16(dout + nrow * lddo + cv * VEC);
          if (dout2) ng2[j] = ldg16(dout2 + nrow * lddo2 + cv * VEC);
          nx[j] = ldg16(x + nrow * ldx + cv * VEC);
          if (dxadd) na[j] = ldg16(dxadd + nrow * lddxa + cv * VEC);
        }
      }
    }
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float xh[NVT][VEC], gy[NVT][VEC];
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < NVT; j++) {
      const int cv = threadIdx.x + j * NT;
      if (cv < cvec) {
        const int c = cv * VEC;
        float gf[VEC], xf[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e++) { gf[e] = Elem<T>::to_f(cg[j].v[e]); xf[e] = Elem<T>::to_f(cx[j].v[e]); }
        if (dout2) {
#pragma unroll
          for (int e = 0; e < VEC; e++) gf[e] += Elem<T>::to_f(cg2[j].v[e]);
        }
        if (p_out > 0.f) dropout_apply<VEC>(gf, seed_out, (uint64_t)row * C + c, thr_out, sc_out);
        if (p_in > 0.f) dropout_apply<VEC>(xf, seed_in, (uint64_t)row * C + c, thr_in, sc_in);
#pragma unroll
        for (int e = 0; e < VEC; e++) {
          const float h = (xf[e] - mean) * rstd;
          xh[j][e] = h;
          dg[j][e] += gf[e] * h;
          db[j][e] += gf[e];
          const float gyv = gf[e] * gm[j][e];
          gy[j][e] = gyv;
          a += gyv;
          b += gyv * h;
        }
      }
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) { red[par][wv][0] = a; red[par][wv][1] = b; }
    __syncthreads();
    a = b = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) { a += red[par][w][0]; b += red[par][w][1]; }
    a *= invC;
    b *= invC;
#pragma unroll
    for (int j = 0; j < NVT; j++) {
      const int cv = threadIdx.x + j * NT;
      if (cv < cvec) {
        const int c = cv * VEC;
        Vec16<T> o;
        float d[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e++) d[e] = rstd * (gy[j][e] - a - xh[j][e] * b);
        if (p_in > 0.f) dropout_apply<VEC>(d, seed_in, (uint64_t)row * C + c, thr_in, sc_in);
        if (relu_in) {      // (see ln_bwd_kernel)
#pragma unroll
          for (int e = 0; e < VEC; e++) d[e] = Elem<T>::to_f(cx[j].v[e]) > 0.f ? d[e] : 0.f;
        }
        if (dxadd) {
#pragma unroll
          for (int e = 0; e < VEC; e++) d[e] += Elem<T>::to_f(ca[j].v[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(d[e]);
        stg16(dx + row * lddx + c, o);
      }
    }
#pragma unroll
    for (int j = 0; j < NVT; j++) { cg[j] = ng[j]; cg2[j] = ng2[j]; cx[j] = nx[j]; ca[j] = na[j]; }
  }
  if (partial) {      // this block's row of the slab: every thread stores its own columns
    float* dst = partial + (long)blockIdx.x * C * 2;
#pragma unroll
    for (int j = 0; j < NVT; j++) {
      const int cv = threadIdx.x + j * NT;
      if (cv < cvec)
#pragma unroll
        for (int e = 0; e < VEC; e++) {
          dst[2 * (cv * VEC + e)] = dg[j][e];
          dst[2 * (cv * VEC + e) + 1] = db[j][e];
        }
    }
  } else {            // ... or adds them to the gradient vectors, consecutive lanes on consecutive addresses (the full-rate atomic shape)
#pragma unroll
    for (int j = 0; j < NVT; j++) {
      const int cv = threadIdx.x + j * NT;
      if (cv < cvec)
#pragma unroll
        for (int e = 0; e < VEC; e++) {
          colsum[0][cv * VEC + e] = dg[j][e];
          colsum[1][cv * VEC + e] = db[j][e];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += NT) {
      atomicAdd(dgamma + c, colsum[0][c]);
      atomicAdd(dbeta + c, colsum[1][c]);
    }
  }
}

// pairs [C][2] -> ADDED to two separate fp32 gradient vectors (dgamma, dbeta)
__global__ void split_pairs_kernel(const float* __restrict__ sums, int C, float* __restrict__ a, float* __restrict__ b) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  a[c] += sums[2 * c];
  b[c] += sums[2 * c + 1];
}

// =============================================================================================
// Masked row softmax over scores S[rows][ldp] (already scaled), one wave per row.
//   rows = batch*heads*Lq, row -> (b = row / (heads*Lq), q = row % Lq)
//   causal: key j > q masked (clip.py:424-430);  kpm[b][j] != 0 masked (layers.py:332, crog.py:55)
//   columns >= Lk (row padding up to ldp) are written as 0 so they can feed the P.V GEMM.
//   P  = softmax probabilities;  Pd (optional) = dropout(P)  (nn.MultiheadAttention dropout, layers.py:291)
// =============================================================================================
// LPR lanes share one row (64 / LPR rows per wave), each lane owns NJ 16-byte vectors: (LPR, NJ) = (4, 1) covers the
// 20-key rows of the text tower / cross attention (16 rows per wave instead of one), (64, 2) the 676-key decoder rows.
// (group_sum<LPR> / group_max<LPR>: common.h - DPP / permlane swaps, nothing through the LDS unit)

template <typename T, int LPR, int NJ>
__global__ void __launch_bounds__(NT) softmax_fwd_kernel(T* __restrict__ S, long rows, int Lq, int Lk, int ldp, int heads,
                                                         int causal, const uint8_t* __restrict__ kpm, T* __restrict__ Pd,
                                                         float p_drop, uint64_t seed, const uint64_t* __restrict__ epoch) {
  constexpr int VEC = Elem<T>::VEC, RPW = 64 / LPR;
  if (epoch) seed += *epoch;
  const int lane = threadIdx.x & 63, sub = lane % LPR;
  const long row = ((long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool live = row < rows;
  const long rr = live ? row : rows - 1;          // dead lanes shadow the last row: the group shuffles stay convergent
  const int q = (int)(rr % Lq);
  const long b = rr / ((long)heads * Lq);
  T* s = S + rr * ldp;
  const int nvec = ldp / VEC;
  float v[NJ][VEC];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int k0 = (sub + j * LPR) * VEC;
#pragma unroll
    for (int e = 0; e < VEC; e++) v[j][e] = -INFINITY;
    if (sub + j * LPR < nvec) {
      const Vec16<T> t = ldg16(s + k0);
#pragma unroll
      for (int e = 0; e < VEC; e++)
        if (k0 + e < Lk) v[j][e] = Elem<T>::to_f(t.v[e]);
      if (causal) {
#pragma unroll
        for (int e = 0; e < VEC; e++)
          if (k0 + e > q) v[j][e] = -INFINITY;
      }
      if (kpm) {
#pragma unroll
        for (int e = 0; e < VEC; e++)
          if (k0 + e < Lk && kpm[b * Lk + k0 + e]) v[j][e] = -INFINITY;
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; e++) mx = fmaxf(mx, v[j][e]);
  }
  mx = group_max<LPR>(mx);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int e = 0; e < VEC; e++) {
      const float ex = (v[j][e] == -INFINITY) ? 0.f : expf(v[j][e] - mx);
      v[j][e] = ex;
      sum += ex;
    }
  sum = group_sum<LPR>(sum);
  const float inv = 1.f / sum;  // a fully masked row gives NaN exactly as torch does
  const uint32_t thr16 = attn_thr16(p_drop);
  const float sc = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  if (!live) return;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int k0 = (sub + j * LPR) * VEC;
    if (sub + j * LPR < nvec) {
      Vec16<T> o;
      float pv[VEC];
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        pv[e] = (k0 + e < Lk) ? v[j][e] * inv : 0.f;     // padding columns are written as 0: they feed the P.V GEMM
        o.v[e] = Elem<T>::from_f(pv[e]);
      }
      stg16(s + k0, o);
      if (Pd) {
        if (p_drop > 0.f) {
#pragma unroll
          for (int e = 0; e < VEC; e += 2) {      // one hash per pair of keys (attn_hash, common.h); k0 is even
            const uint32_t hh = attn_hash(seed, (uint64_t)row * (uint64_t)((ldp + 1) >> 1) + (uint64_t)((k0 + e) >> 1));
            o.v[e] = Elem<T>::from_f(attn_keep_lo(hh, thr16) ? pv[e] * sc : 0.f);
            o.v[e + 1] = Elem<T>::from_f(attn_keep_hi(hh, thr16) ? pv[e + 1] * sc : 0.f);
          }
        }
        stg16(Pd + row * ldp + k0, o);
      }
    }
  }
}

// dS = P * (dP - sum_k dP*P) with dP = dropout_bwd(dPd); written in place over dPd.
template <typename T, int LPR, int NJ>
__global__ void __launch_bounds__(NT) softmax_bwd_kernel(const T* __restrict__ P, T* __restrict__ dPd, long rows, int Lk, int ldp,
                                                         float p_drop, uint64_t seed, const uint64_t* __restrict__ epoch) {
  constexpr int VEC = Elem<T>::VEC, RPW = 64 / LPR;
  if (epoch) seed += *epoch;
  const int lane = threadIdx.x & 63, sub = lane % LPR;
  const long row = ((long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool live = row < rows;
  const long rr = live ? row : rows - 1;
  const uint32_t thr16 = attn_thr16(p_drop);
  const float sc = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const int nvec = ldp / VEC;
  float pv[NJ][VEC], dp[NJ][VEC];
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int k0 = (sub + j * LPR) * VEC;
#pragma unroll
    for (int e = 0; e < VEC; e++) pv[j][e] = dp[j][e] = 0.f;
    if (sub + j * LPR < nvec) {
      const Vec16<T> a = ldg16(P + rr * ldp + k0), g = ldg16(dPd + rr * ldp + k0);
#pragma unroll
      for (int e = 0; e < VEC; e++)
        if (k0 + e < Lk) { pv[j][e] = Elem<T>::to_f(a.v[e]); dp[j][e] = Elem<T>::to_f(g.v[e]); }
      if (p_drop > 0.f) {
#pragma unroll
        for (int e = 0; e < VEC; e += 2) {
          const uint32_t hh = attn_hash(seed, (uint64_t)rr * (uint64_t)((ldp + 1) >> 1) + (uint64_t)((k0 + e) >> 1));
          dp[j][e] = attn_keep_lo(hh, thr16) ? dp[j][e] * sc : 0.f;
          dp[j][e + 1] = attn_keep_hi(hh, thr16) ? dp[j][e + 1] * sc : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < VEC; e++) dot += dp[j][e] * pv[j][e];
    }
  }
  dot = group_sum<LPR>(dot);
  if (!live) return;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int k0 = (sub + j * LPR) * VEC;
    if (sub + j * LPR < nvec) {
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(k0 + e < Lk ? pv[j][e] * (dp[j][e] - dot) : 0.f);
      stg16(dPd + row * ldp + k0, o);
    }
  }
}

inline int stream_grid(long work_items) {
  // (round 6 A/B: caps of 4096 / 8192 / 16384 blocks - more blocks than are resident at once, so that the dispatcher deals rows to whichever CU is
  // free of the side stream's GEMM blocks - made the step SLOWER: 27.8 / 28.0-28.9 / 29.0 ms against 27.5-27.6; every block re-derives its
  // per-channel constants, and the kernels are not tail-bound)
  long g = (work_items + NT - 1) / NT;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

#define DISPATCH_T(dtype, ...)                                   \
  do {                                                           \
    if ((dtype) == CROG_BF16) { using T = bf16; __VA_ARGS__; }   \
    else if ((dtype) == CROG_F32) { using T = float; __VA_ARGS__; } \
    else { crog_set_error("bad dtype %d", (int)(dtype)); return CROG_ERR_ARG; } \
  } while (0)

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" int crog_bn_stat_blocks(int64_t M, int rows_per_block) { return cdiv(M, rows_per_block); }

extern "C" int crog_bn_partial_stats(int dtype, const void* x, int64_t M, int C, int64_t ld, int rows_per_block,
                                     float* partial, crog_stream_t stream) {
  const int vec = dtype == CROG_BF16 ? 8 : 4;
  CROG_CHECK_ARG(C % vec == 0 && pow2(C / vec) && ld % vec == 0, "bn_partial_stats: C/vec must be a power of two (C=%d)", C);
  CROG_CHECK_ARG(rows_per_block > 0 && M > 0, "bn_partial_stats: bad sizes");
  const int blocks = cdiv(M, rows_per_block);
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_partial_stats_kernel<T>), dim3(blocks), dim3(NT), 0, (hipStream_t)stream,
                                       (const T*)x, (long)M, C, (long)ld, rows_per_block, partial));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

extern "C" int crog_reduce_pairs(const float* partial, int nparts, int C, float* sums, int sums_is_zero, crog_stream_t stream) {
  CROG_CHECK_ARG(nparts > 0 && C > 0, "reduce_pairs: bad sizes");
  if (crog_deterministic()) {      // one block per 8 channels walks the slab in order and STORES the sums: no atomics, no memset needed
    hipLaunchKernelGGL(reduce_split_kernel, dim3(cdiv(C, SLAB_CH)), dim3(NT), 0, (hipStream_t)stream, partial, nparts, C, sums, (float*)nullptr, (float*)nullptr);
    CROG_LAUNCH_CHECK();
    return CROG_OK;
  }
  if (!sums_is_zero) {   // callers that hand out pre-zeroed scratch (one memset per step for all layers) skip this launch
    hipError_t e = hipMemsetAsync(sums, 0, (size_t)C * 2 * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) { crog_set_error("reduce_pairs: memset failed"); return CROG_ERR_LAUNCH; }
  }
  int splits = nparts / 16;
  if (splits < 1) splits = 1;
  if (splits > 128) splits = 128;
  hipLaunchKernelGGL(reduce_pairs_kernel, dim3(cdiv(2 * C, 64), splits), dim3(NT), 0, (hipStream_t)stream, partial, nparts, C, sums);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

extern "C" int crog_split_pairs(const float* sums, int C, float* a, float* b, crog_stream_t stream) {
  hipLaunchKernelGGL(split_pairs_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, sums, C, a, b);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

extern "C" int crog_bn_finalize(const float* sums, float count, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, float momentum, float eps, int C, float* scale_shift, float* mean_invstd,
                                crog_stream_t stream) {
  CROG_CHECK_ARG(count > 0 && C > 0, "bn_finalize: bad sizes");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, sums, count, gamma, beta,
                     running_mean, running_var, momentum, eps, C, scale_shift, mean_invstd);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

extern "C" int crog_bn_eval_scale(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                  float eps, int C, float* scale_shift, crog_stream_t stream) {
  hipLaunchKernelGGL(bn_eval_scale_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, gamma, beta, running_mean,
                     running_var, eps, C, scale_shift);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

extern "C" int crog_bn_fold_weights(int dtype_dst, const float* w_src, int64_t lds_, int cols_src, int rows_per_channel, const float* gamma,
                                    const float* beta, const float* running_mean, const float* running_var, float eps, void* w_dst,
                                    int64_t ldd, int cols_dst, int64_t rows, float* bias_dst, crog_stream_t stream) {
  CROG_CHECK_ARG(w_src && w_dst && bias_dst && gamma && beta && running_mean && running_var, "bn_fold_weights: null pointer");
  CROG_CHECK_ARG(rows > 0 && cols_dst > 0 && cols_src > 0 && cols_src <= cols_dst && rows_per_channel >= 1 && rows % rows_per_channel == 0,
                 "bn_fold_weights: bad sizes rows=%ld cols %d -> %d rows_per_channel=%d", (long)rows, cols_src, cols_dst, rows_per_channel);
  DISPATCH_T(dtype_dst, hipLaunchKernelGGL((bn_fold_weights_kernel<T>), dim3(stream_grid(rows * cols_dst)), dim3(NT), 0, (hipStream_t)stream,
                                           w_src, (long)lds_, cols_src, rows_per_channel, gamma, beta, running_mean, running_var, eps, (T*)w_dst,
                                           (long)ldd, cols_dst, (long)rows, bias_dst));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

extern "C" int crog_bn_apply(int dtype, const void* z, int64_t ldz, const float* scale_shift, const void* res, int64_t ldr,
                             int relu, void* y, int64_t ldy, int64_t M, int C, void* relu_mask, crog_stream_t stream) {
  const int vec = dtype == CROG_BF16 ? 8 : 4;
  CROG_CHECK_ARG(C % vec == 0 && ldz % vec == 0 && ldy % vec == 0 && (!res || ldr % vec == 0), "bn_apply: C/ld must be multiples of %d", vec);
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(stream_grid(M * (C / vec))), dim3(NT), 0, (hipStream_t)stream,
                                       (const T*)z, (long)ldz, scale_shift, (const T*)res, (long)ldr, relu, (T*)y, (long)ldy,
                                       (long)M, C, (unsigned char*)relu_mask));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

static int bn_apply_stats_impl(int dtype, const void* z, int64_t ldz, const float* sums, int replicas, float count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               float* scale_shift, float* mean_invstd, const void* res, int64_t ldr, int relu, void* y, int64_t ldy,
                               int64_t M, int C, void* relu_mask, int poolH, int poolW, crog_stream_t stream) {
  const int vec = dtype == CROG_BF16 ? 8 : 4;
  CROG_CHECK_ARG(poolW == 0 || (poolH > 0 && poolW > 0 && poolH % 2 == 0 && poolW % 2 == 0 && M % ((int64_t)poolH * poolW) == 0 && M < 0x7fffffffL &&
                                !res && !relu_mask),
                 "bn_apply_stats: pooled form needs even H, W, M = B * H * W < 2^31, no residual");
  CROG_CHECK_ARG(C % vec == 0 && ldz % vec == 0 && ldy % vec == 0 && (!res || ldr % vec == 0), "bn_apply_stats: C/ld must be multiples of %d", vec);
  CROG_CHECK_ARG(sums && replicas >= 1 && count > 0 && scale_shift && mean_invstd && C <= 8192, "bn_apply_stats: bad arguments");
  const int grid = std::min(stream_grid(M * (C / vec)), 1024);   // every block re-derives the C scale/shift pairs: keep the grid modest
  const bool tiny = M <= 64 && count == (float)M && !poolW;      // (the kernel's own test: it then keeps (mean, beta) per channel in LDS as well)
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_apply_stats_kernel<T>), dim3(grid), dim3(NT), (size_t)C * (tiny ? 4 : 2) * sizeof(float), (hipStream_t)stream,
                                       (const T*)z, (long)ldz, sums, replicas, count, gamma, beta, running_mean, running_var, momentum, eps,
                                       scale_shift, mean_invstd, (const T*)res, (long)ldr, relu, (T*)y, (long)ldy, (long)M, C, (unsigned char*)relu_mask,
                                       poolH, poolW));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_bn_apply_stats(int dtype, const void* z, int64_t ldz, const float* sums, int replicas, float count, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                   float* scale_shift, float* mean_invstd, const void* res, int64_t ldr, int relu, void* y, int64_t ldy,
                                   int64_t M, int C, void* relu_mask, crog_stream_t stream) {
  return bn_apply_stats_impl(dtype, z, ldz, sums, replicas, count, gamma, beta, running_mean, running_var, momentum, eps, scale_shift, mean_invstd,
                             res, ldr, relu, y, ldy, M, C, relu_mask, 0, 0, stream);
}
extern "C" int crog_bn_apply_stats_pool(int dtype, const void* z, int64_t ldz, const float* sums, int replicas, float count, const float* gamma,
                                        const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                        float* scale_shift, float* mean_invstd, int relu, void* y, int64_t ldy, int64_t M, int C, int H, int W,
                                        crog_stream_t stream) {
  return bn_apply_stats_impl(dtype, z, ldz, sums, replicas, count, gamma, beta, running_mean, running_var, momentum, eps, scale_shift, mean_invstd,
                             nullptr, 0, relu, y, ldy, M, C, nullptr, H, W, stream);
}

static int bn_bwd_partial_impl(int dtype, const void* dy, int64_t lddy, const void* y, int64_t ldy, const void* z, int64_t ldz,
                               const float* mean_invstd, const float* relu_scale_shift, int64_t M, int C, int rows_per_block,
                               float* partial, int replicas, const void* relu_mask, int poolH, int poolW, crog_stream_t stream,
                               const void* sync = nullptr, int tail = 0) {
  const int vec = dtype == CROG_BF16 ? 8 : 4;
  CROG_CHECK_ARG(poolW == 0 || (poolH > 0 && poolH % 2 == 0 && poolW % 2 == 0 && M % ((int64_t)poolH * poolW) == 0 && M < 0x7fffffffL),
                 "bn_bwd_partial: pooled form needs even H, W and M = B * H * W < 2^31");
  CROG_CHECK_ARG(C % vec == 0 && pow2(C / vec), "bn_bwd_partial: C/vec must be a power of two (C=%d)", C);
  CROG_CHECK_ARG(replicas >= 0, "bn_bwd_partial: replicas must be >= 0");
  const int blocks = cdiv(M, rows_per_block);
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_partial_kernel<T>), dim3(blocks), dim3(NT), 0, (hipStream_t)stream, (const T*)dy,
                                       (long)lddy, (const T*)y, (long)ldy, (const T*)z, (long)ldz, mean_invstd, relu_scale_shift, (long)M, C,
                                       rows_per_block, partial, replicas, (const unsigned char*)relu_mask, poolH, poolW, (const CrogSyncBlock*)sync, tail));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_bn_bwd_partial_sync(int dtype, const void* dy, int64_t lddy, const void* y, int64_t ldy, const void* z, int64_t ldz,
                                        const float* mean_invstd, const float* relu_scale_shift, int64_t M, int C, int rows_per_block,
                                        float* partial, int replicas, const void* relu_mask, int H, int W, const void* stat_sync, crog_stream_t stream) {
  CROG_CHECK_ARG(replicas > 0, "bn_bwd_partial_sync: the totals tail needs the atomic replica form (replicas > 0)");
  return bn_bwd_partial_impl(dtype, dy, lddy, y, ldy, z, ldz, mean_invstd, relu_scale_shift, M, C, rows_per_block, partial, replicas, relu_mask, H, W, stream,
                             stat_sync, 1);
}
extern "C" int crog_bn_bwd_partial(int dtype, const void* dy, int64_t lddy, const void* y, int64_t ldy, const void* z, int64_t ldz,
                                   const float* mean_invstd, const float* relu_scale_shift, int64_t M, int C, int rows_per_block,
                                   float* partial, int replicas, const void* relu_mask, crog_stream_t stream) {
  return bn_bwd_partial_impl(dtype, dy, lddy, y, ldy, z, ldz, mean_invstd, relu_scale_shift, M, C, rows_per_block, partial, replicas, relu_mask, 0, 0, stream);
}
extern "C" int crog_bn_bwd_partial_pool(int dtype, const void* dy_pooled, int64_t lddy, const void* z, int64_t ldz, const float* mean_invstd,
                                        const float* relu_scale_shift, int64_t M, int C, int rows_per_block, float* partial, int replicas,
                                        int H, int W, crog_stream_t stream) {
  return bn_bwd_partial_impl(dtype, dy_pooled, lddy, nullptr, 0, z, ldz, mean_invstd, relu_scale_shift, M, C, rows_per_block, partial, replicas, nullptr,
                             H, W, stream);
}

static int bn_bwd_apply_impl(int dtype, const void* dy, int64_t lddy, const void* y, int64_t ldy, const void* z, int64_t ldz,
                             const float* mean_invstd, const float* gamma, const float* sums, float count,
                             const float* relu_scale_shift, void* dz, int64_t lddz, void* dres, int64_t lddres, int64_t M, int C,
                             int sum_rows, float* dgamma, float* dbeta, float param_grad_scale, const void* relu_mask, int poolH, int poolW,
                             crog_stream_t stream) {
  const int vec = dtype == CROG_BF16 ? 8 : 4;
  CROG_CHECK_ARG(poolW == 0 || (poolH > 0 && poolH % 2 == 0 && poolW % 2 == 0 && M % ((int64_t)poolH * poolW) == 0 && M < 0x7fffffffL && !dres),
                 "bn_bwd_apply: pooled form needs even H, W, M = B * H * W < 2^31, no residual output");
  CROG_CHECK_ARG(C % vec == 0, "bn_bwd_apply: C %% %d != 0", vec);
  CROG_CHECK_ARG(C <= 8192 && (!dgamma || (dbeta && sum_rows != 0)), "bn_bwd_apply: bad sum_rows / parameter-gradient outputs");
  int grid = stream_grid(M * (C / vec));
  if (sum_rows != 0) grid = std::min(grid, 1024);   // every block adds up the |sum_rows| x 2C partials once
  const size_t lds = sum_rows != 0 ? (size_t)C * 2 * sizeof(float) : 0;
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(grid), dim3(NT), lds, (hipStream_t)stream, (const T*)dy, (long)lddy,
                                       (const T*)y, (long)ldy, (const T*)z, (long)ldz, mean_invstd, gamma, sums, count, relu_scale_shift,
                                       (T*)dz, (long)lddz, (T*)dres, (long)lddres, (long)M, C, sum_rows, dgamma, dbeta, (const unsigned char*)relu_mask,
                                       param_grad_scale, poolH, poolW));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_bn_bwd_apply(int dtype, const void* dy, int64_t lddy, const void* y, int64_t ldy, const void* z, int64_t ldz,
                                 const float* mean_invstd, const float* gamma, const float* sums, float count,
                                 const float* relu_scale_shift, void* dz, int64_t lddz, void* dres, int64_t lddres, int64_t M, int C,
                                 int sum_rows, float* dgamma, float* dbeta, float param_grad_scale, const void* relu_mask,
                                 crog_stream_t stream) {
  return bn_bwd_apply_impl(dtype, dy, lddy, y, ldy, z, ldz, mean_invstd, gamma, sums, count, relu_scale_shift, dz, lddz, dres, lddres, M, C, sum_rows,
                           dgamma, dbeta, param_grad_scale, relu_mask, 0, 0, stream);
}
extern "C" int crog_bn_bwd_apply_pool(int dtype, const void* dy_pooled, int64_t lddy, const void* z, int64_t ldz, const float* mean_invstd,
                                      const float* gamma, const float* sums, float count, const float* relu_scale_shift, void* dz, int64_t lddz,
                                      int64_t M, int C, int sum_rows, float* dgamma, float* dbeta, float param_grad_scale, int H, int W,
                                      crog_stream_t stream) {
  return bn_bwd_apply_impl(dtype, dy_pooled, lddy, nullptr, 0, z, ldz, mean_invstd, gamma, sums, count, relu_scale_shift, dz, lddz, nullptr, 0, M, C,
                           sum_rows, dgamma, dbeta, param_grad_scale, nullptr, H, W, stream);
}

extern "C" int crog_ln_fwd(int dtype, const void* x, int64_t ldx, const float* gamma, const float* beta, float eps, int64_t M, int C,
                           void* out, int64_t ldo, float* stats, const void* res, int64_t ldr, void* out2, int64_t ldo2,
                           const void* pos, int pos_rows, int64_t ldp, float p_in, uint64_t seed_in, float p_out, uint64_t seed_out,
                           crog_stream_t stream) {
  const int vec = dtype == CROG_BF16 ? 8 : 4;
  CROG_CHECK_ARG(C % vec == 0 && C <= 2048, "ln_fwd: C=%d must be a multiple of %d and <= 2048", C, vec);
  CROG_CHECK_ARG(!pos || pos_rows > 0, "ln_fwd: pos_rows");
  const int nv = cdiv(C / vec, 64);
  const int blocks = (int)std::min<long>(cdiv(M, NT / 64), 256 * 8);   // 8 blocks (32 waves) per CU, rows by grid stride
#define CROG_LN_FWD(NV)                                                                                                          \
  DISPATCH_T(dtype, hipLaunchKernelGGL((ln_fwd_kernel<T, NV>), dim3(blocks), dim3(NT), 0, (hipStream_t)stream, (const T*)x,     \
                                       (long)ldx, gamma, beta, eps, (long)M, C, (T*)out, (long)ldo, stats, (const T*)res, (long)ldr, \
                                       (T*)out2, (long)ldo2, (const T*)pos, pos_rows, (long)ldp, p_in, seed_in, p_out, seed_out, crog_seed_epoch()))
#define CROG_LN_FWD_ROW(NVT)                                                                                                      \
  DISPATCH_T(dtype, hipLaunchKernelGGL((ln_fwd_row_kernel<T, NVT>), dim3((int)std::min<long>(M, 256 * 8)), dim3(NT), 0, (hipStream_t)stream, (const T*)x, \
                                       (long)ldx, gamma, beta, eps, (long)M, C, (T*)out, (long)ldo, stats, (const T*)res, (long)ldr, \
                                       (T*)out2, (long)ldo2, (const T*)pos, pos_rows, (long)ldp, p_in, seed_in, p_out, seed_out, crog_seed_epoch()))
  if (nv <= 1) CROG_LN_FWD(1);
  else if (nv <= 2) CROG_LN_FWD(2);
  else if (C / vec <= NT) CROG_LN_FWD_ROW(1);      // wide rows: one row per block iteration (ln_fwd_row_kernel)
  else CROG_LN_FWD_ROW(2);
#undef CROG_LN_FWD_ROW
#undef CROG_LN_FWD
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

extern "C" int crog_ln_bwd_blocks(int64_t M, int rows_per_block) { return cdiv(M, rows_per_block); }

static int ln_bwd_impl(int dtype, const void* dout, int64_t lddo, const void* dout2, int64_t lddo2, const void* x, int64_t ldx,
                       const float* gamma, const float* stats, int64_t M, int C, void* dx, int64_t lddx, float* partial,
                       int rows_per_block, float p_in, uint64_t seed_in, float p_out, uint64_t seed_out, float* dgamma, float* dbeta,
                       const void* dxadd, int64_t lddxa, int relu_in, crog_stream_t stream) {
  const int vec = dtype == CROG_BF16 ? 8 : 4;
  CROG_CHECK_ARG(!dxadd || (lddxa >= C && lddxa % vec == 0 && ((uintptr_t)dxadd % 16) == 0), "ln_bwd: dxadd rows must be 16-byte aligned (ld=%lld)", (long long)lddxa);
  CROG_CHECK_ARG(C % vec == 0 && C <= 2048, "ln_bwd: C=%d must be a multiple of %d and <= 2048", C, vec);
  CROG_CHECK_ARG((dgamma != nullptr) == (dbeta != nullptr) && (partial != nullptr) != (dgamma != nullptr),
                 "ln_bwd: give either the partial slab or both gradient vectors");
  const int blocks = cdiv(M, rows_per_block);
  const int nv = cdiv(C / vec, 64);
#define CROG_LN_BWD(NV)                                                                                                           \
  DISPATCH_T(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T, NV>), dim3(blocks), dim3(NT), 0, (hipStream_t)stream, (const T*)dout, (long)lddo, \
                                       (const T*)dout2, (long)lddo2, (const T*)x, (long)ldx, gamma, stats, (long)M, C, (T*)dx,       \
                                       (long)lddx, partial, rows_per_block, p_in, seed_in, p_out, seed_out, dgamma, dbeta, crog_seed_epoch(),   \
                                       (const T*)dxadd, (long)lddxa, relu_in))
#define CROG_LN_BWD_ROW(NVT)                                                                                                       \
  DISPATCH_T(dtype, hipLaunchKernelGGL((ln_bwd_row_kernel<T, NVT>), dim3(blocks), dim3(NT), 0, (hipStream_t)stream, (const T*)dout, (long)lddo, \
                                       (const T*)dout2, (long)lddo2, (const T*)x, (long)ldx, gamma, stats, (long)M, C, (T*)dx,       \
                                       (long)lddx, partial, rows_per_block, p_in, seed_in, p_out, seed_out, dgamma, dbeta, crog_seed_epoch(),   \
                                       (const T*)dxadd, (long)lddxa, relu_in))
  if (nv <= 1) CROG_LN_BWD(1);
  else if (nv <= 2) CROG_LN_BWD(2);
  else if (C / vec <= NT) CROG_LN_BWD_ROW(1);      // wide rows: one row per block iteration (ln_bwd_row_kernel)
  else CROG_LN_BWD_ROW(2);
#undef CROG_LN_BWD_ROW
#undef CROG_LN_BWD
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_ln_bwd(int dtype, const void* dout, int64_t lddo, const void* dout2, int64_t lddo2, const void* x, int64_t ldx,
                           const float* gamma, const float* stats, int64_t M, int C, void* dx, int64_t lddx, float* partial,
                           int rows_per_block, float p_in, uint64_t seed_in, float p_out, uint64_t seed_out, float* dgamma, float* dbeta,
                           const void* dxadd, int64_t lddxa, crog_stream_t stream) {
  return ln_bwd_impl(dtype, dout, lddo, dout2, lddo2, x, ldx, gamma, stats, M, C, dx, lddx, partial, rows_per_block, p_in, seed_in, p_out, seed_out,
                     dgamma, dbeta, dxadd, lddxa, 0, stream);
}
extern "C" int crog_ln_bwd_relu(int dtype, const void* dout, int64_t lddo, const void* dout2, int64_t lddo2, const void* x, int64_t ldx,
                                const float* gamma, const float* stats, int64_t M, int C, void* dx, int64_t lddx, float* partial,
                                int rows_per_block, float p_in, uint64_t seed_in, float p_out, uint64_t seed_out, float* dgamma, float* dbeta,
                                const void* dxadd, int64_t lddxa, crog_stream_t stream) {
  return ln_bwd_impl(dtype, dout, lddo, dout2, lddo2, x, ldx, gamma, stats, M, C, dx, lddx, partial, rows_per_block, p_in, seed_in, p_out, seed_out,
                     dgamma, dbeta, dxadd, lddxa, 1, stream);
}

extern "C" int crog_softmax_fwd(int dtype, void* S, int64_t rows, int Lq, int Lk, int ldp, int heads, int causal,
                                const uint8_t* key_padding_mask, void* Pd, float p_drop, uint64_t seed, crog_stream_t stream) {
  CROG_CHECK_ARG(Lk > 0 && Lk <= ldp && ldp <= 768, "softmax_fwd: need Lk <= ldp <= 768 (Lk=%d ldp=%d)", Lk, ldp);
  const int vec = dtype == CROG_BF16 ? 8 : 4;
  CROG_CHECK_ARG(ldp % vec == 0 && ((uintptr_t)S % 16) == 0 && (!Pd || ((uintptr_t)Pd % 16) == 0), "softmax_fwd: ldp %% %d == 0 and 16-byte aligned rows required", vec);
  const int nvec = ldp / vec;
#define CROG_SM_FWD(LPR, NJ)                                                                                                       \
  DISPATCH_T(dtype, hipLaunchKernelGGL((softmax_fwd_kernel<T, LPR, NJ>), dim3(cdiv(rows, (NT / 64) * (64 / LPR))), dim3(NT), 0,    \
                                       (hipStream_t)stream, (T*)S, (long)rows, Lq, Lk, ldp, heads, causal, key_padding_mask, (T*)Pd, \
                                       p_drop, seed, crog_seed_epoch()))
  if (nvec <= 4) CROG_SM_FWD(4, 1);
  else if (nvec <= 16) CROG_SM_FWD(16, 1);
  else if (nvec <= 64) CROG_SM_FWD(64, 1);
  else if (nvec <= 128) CROG_SM_FWD(64, 2);
  else CROG_SM_FWD(64, 3);
#undef CROG_SM_FWD
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

extern "C" int crog_softmax_bwd(int dtype, const void* P, void* dPd, int64_t rows, int Lk, int ldp, float p_drop, uint64_t seed,
                                crog_stream_t stream) {
  CROG_CHECK_ARG(Lk > 0 && Lk <= ldp && ldp <= 768, "softmax_bwd: need Lk <= ldp <= 768");
  const int vec = dtype == CROG_BF16 ? 8 : 4;
  CROG_CHECK_ARG(ldp % vec == 0 && ((uintptr_t)P % 16) == 0 && ((uintptr_t)dPd % 16) == 0, "softmax_bwd: ldp %% %d == 0 and 16-byte aligned rows required", vec);
  const int nvec = ldp / vec;
#define CROG_SM_BWD(LPR, NJ)                                                                                                       \
  DISPATCH_T(dtype, hipLaunchKernelGGL((softmax_bwd_kernel<T, LPR, NJ>), dim3(cdiv(rows, (NT / 64) * (64 / LPR))), dim3(NT), 0,    \
                                       (hipStream_t)stream, (const T*)P, (T*)dPd, (long)rows, Lk, ldp, p_drop, seed, crog_seed_epoch()))
  if (nvec <= 4) CROG_SM_BWD(4, 1);
  else if (nvec <= 16) CROG_SM_BWD(16, 1);
  else if (nvec <= 64) CROG_SM_BWD(64, 1);
  else if (nvec <= 128) CROG_SM_BWD(64, 2);
  else CROG_SM_BWD(64, 3);
#undef CROG_SM_BWD
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

extern "C" int crog_bn_reduce_finalize(const float* partial, int nparts, float count, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, float momentum, float eps, int C, float* scale_shift,
                                       float* mean_invstd, crog_stream_t stream) {
  CROG_CHECK_ARG(nparts > 0 && C > 0 && count > 0, "bn_reduce_finalize: bad sizes");
  hipLaunchKernelGGL(bn_reduce_finalize_kernel, dim3(cdiv(C, SLAB_CH)), dim3(NT), 0, (hipStream_t)stream, partial, nparts, count, gamma, beta,
                     running_mean, running_var, momentum, eps, C, scale_shift, mean_invstd);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

extern "C" int crog_reduce_split(const float* partial, int nparts, int C, float* sums, float* a, float* b, crog_stream_t stream) {
  CROG_CHECK_ARG(nparts > 0 && C > 0, "reduce_split: bad sizes");
  hipLaunchKernelGGL(reduce_split_kernel, dim3(cdiv(C, SLAB_CH)), dim3(NT), 0, (hipStream_t)stream, partial, nparts, C, sums, a, b);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
