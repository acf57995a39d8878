// Text-conditioned dynamic-conv head, losses and the training metric
// (layers.py:64-132 MultiTaskProjector / :152-173 Projector; crog.py:76-111; utils/misc.py:115-131).
//
// The per-sample grouped 3x3 conv  out[b,h] = conv3x3(x5[b, h*C:(h+1)*C], w[b]) + bias[b]  is evaluated as
//   t[b,p,h,tap] = sum_c x5[b,p,h*C+c] * w[b,c,tap]        (batched MFMA GEMM, reads x5 exactly once)
//   out[b,h,y,x] = bias[b] + sum_tap t[b,(y+dy,x+dx),h,tap] (9-point stencil on a 16-wide fp32 map)
// which is the same sum re-associated; the stencil, the nearest-neighbour target resize, the five
// losses and d(loss)/d(pred) are one pass each over the small maps.
#include "common.h"
#include <algorithm>

namespace {

constexpr int NT = 256;
inline int stream_grid(long work_items) {
  long g = (work_items + NT - 1) / NT;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}
#define GRID_STRIDE(i, total) for (long i = (long)blockIdx.x * NT + threadIdx.x; i < (total); i += (long)gridDim.x * NT)

// word (fp32 [B][ldw], txt Linear output: C*9 weights then 1 bias) -> wpad[b][c][16] (taps 9..15 zero)
template <typename T>
__global__ void __launch_bounds__(NT) head_pack_weights_kernel(const float* __restrict__ word, long ldw, T* __restrict__ wpad, int B, int C) {
  GRID_STRIDE(i, (long)B * C * 16) {
    const int tap = (int)(i & 15);
    const long bc = i >> 4;
    const int c = (int)(bc % C);
    const long b = bc / C;
    wpad[i] = Elem<T>::from_f(tap < 9 ? word[b * ldw + c * 9 + tap] : 0.f);
  }
}
// dwpad (fp32 [B][C][16]) + dbias[b] -> dword (T [B][ldd]): cols [0,C*9) weights, col C*9 bias, rest 0
template <typename T>
__global__ void __launch_bounds__(NT) head_unpack_wgrad_kernel(const float* __restrict__ dwpad, const float* __restrict__ dbias, T* __restrict__ dword,
                                                               long ldd, int B, int C) {
  GRID_STRIDE(i, (long)B * ldd) {
    const int col = (int)(i % ldd);
    const long b = i / ldd;
    float v = 0.f;
    if (col < C * 9) v = dwpad[(b * C + col / 9) * 16 + col % 9];
    else if (col == C * 9) v = dbias[b];
    dword[i] = Elem<T>::from_f(v);
  }
}

// out[b,h,y,x] = bias[b] + sum_tap (t[((b*P + p')*heads + h)*16 + tap] + tbias[(b*heads + h)*16 + tap])   (tbias optional)
__global__ void __launch_bounds__(NT) head_stencil_fwd_kernel(const float* __restrict__ t, const float* __restrict__ word, long ldw, int bias_col,
                                                              const float* __restrict__ tbias, float* __restrict__ out, int B, int heads, int H,
                                                              int W) {
  const long P = (long)H * W;
  GRID_STRIDE(i, (long)B * heads * P) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const int h = (int)((i / P) % heads);
    const long b = i / (P * heads);
    float acc = word[b * ldw + bias_col];
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
      const int sy = y + tap / 3 - 1, sx = x + tap % 3 - 1;
      if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
        acc += t[((b * P + (long)sy * W + sx) * heads + h) * 16 + tap];
        if (tbias) acc += tbias[(b * heads + h) * 16 + tap];
      }
    }
    out[i] = acc;
  }
}
// Same stencil, row-band form: a block owns HS_R output rows of one sample; the t rows it needs (HS_R + 2, taps 0..8 of every
// head) are read ONCE as whole 48-byte runs and parked in LDS, and the nine taps of an output come from LDS.  The gather form
// above pulls a different 64-byte sector for every 4 bytes it uses (measured 1.8 GB of traffic per launch for a 110 MB tensor).
constexpr int HS_R = 2;
__global__ void __launch_bounds__(NT) head_stencil_fwd_rows_kernel(const float* __restrict__ t, const float* __restrict__ word, long ldw, int bias_col,
                                                                   const float* __restrict__ tbias, float* __restrict__ out, int heads, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float tl[];   // [(HS_R + 2)][W][heads * 9]
  const int b = blockIdx.y, y0 = blockIdx.x * HS_R;
  const long P = (long)H * W;
  const int hp = heads * 9;
  const int npair = (HS_R + 2) * W * heads;
  for (int j = threadIdx.x; j < npair; j += NT) {
    const int h = j % heads, px = (j / heads) % W, r = j / (heads * W);
    const int sy = y0 - 1 + r;
    if (sy < 0 || sy >= H) continue;
    const float4* src = reinterpret_cast<const float4*>(t + ((b * P + (long)sy * W + px) * heads + h) * 16);
    const float4 v0 = src[0], v1 = src[1];
    const float v8 = reinterpret_cast<const float*>(src)[8];
    float* dst = tl + ((long)r * W + px) * hp + h * 9;
    dst[0] = v0.x; dst[1] = v0.y; dst[2] = v0.z; dst[3] = v0.w;
    dst[4] = v1.x; dst[5] = v1.y; dst[6] = v1.z; dst[7] = v1.w;
    dst[8] = v8;
  }
  __syncthreads();
  const float bias = word[(long)b * ldw + bias_col];
  const int nout = HS_R * W * heads;
  for (int o = threadIdx.x; o < nout; o += NT) {
    const int x = o % W, ry = (o / W) % HS_R, h = o / (W * HS_R);
    const int y = y0 + ry;
    if (y >= H) continue;
    float acc = bias;
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const int sy = y + dy, sx = x + dx;
      if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
        acc += tl[((long)(ry + 1 + dy) * W + sx) * hp + h * 9 + tap];
        if (tbias) acc += tbias[((long)b * heads + h) * 16 + tap];
      }
    }
    out[(((long)b * heads + h) * H + y) * W + x] = acc;
  }
}
// dt[((b*P + p)*heads + h)*16 + tap] = dout[b,h,p - off(tap)]
template <typename T>
__global__ void __launch_bounds__(NT) head_stencil_bwd_kernel(const float* __restrict__ dout, T* __restrict__ dt, int B, int heads, int H, int W) {
  const long P = (long)H * W;
  GRID_STRIDE(i, (long)B * P * heads) {
    const int h = (int)(i % heads);
    const long bp = i / heads;
    const int x = (int)(bp % W), y = (int)((bp / W) % H);
    const long b = bp / P;
    Vec16<T> o[16 / Elem<T>::VEC];
#pragma unroll
    for (int tap = 0; tap < 16; tap++) {
      float v = 0.f;
      if (tap < 9) {
        const int oy = y - (tap / 3 - 1), ox = x - (tap % 3 - 1);
        if (oy >= 0 && oy < H && ox >= 0 && ox < W) v = dout[((b * heads + h) * H + oy) * W + ox];
      }
      o[tap / Elem<T>::VEC].v[tap % Elem<T>::VEC] = Elem<T>::from_f(v);
    }
#pragma unroll
    for (int v = 0; v < 16 / Elem<T>::VEC; v++) stg16(dt + i * 16 + v * Elem<T>::VEC, o[v]);
  }
}
// ---- folded 1x1 conv (vis.4, layers.py:58) + dynamic 3x3 head: the conv's bias reaches the logits through the per-sample
// constants cb[b][h][tap] = sum_c b5[h*C + c] * w_b[c][tap] (one per source pixel and tap) ------------------------------
template <typename T>
__global__ void __launch_bounds__(NT) head_cb_fwd_kernel(const float* __restrict__ b5, const T* __restrict__ wpad, float* __restrict__ cb, int B,
                                                         int heads, int C) {
  GRID_STRIDE(i, (long)B * heads * 16) {
    const int tap = (int)(i & 15);
    const int h = (int)((i >> 4) % heads);
    const long b = (i >> 4) / heads;
    float acc = 0.f;
    for (int c = 0; c < C; c++) acc += b5[h * C + c] * Elem<T>::to_f(wpad[(b * C + c) * 16 + tap]);
    cb[i] = acc;
  }
}
// dcb[(b*heads + h)*16 + tap] += sum_p dt[((b*P + p)*heads + h)*16 + tap]      grid (chunks, B), dcb zeroed by the launcher
template <typename T>
__global__ void __launch_bounds__(NT) head_tap_sums_kernel(const T* __restrict__ dt, float* __restrict__ dcb, long P, int heads, float* __restrict__ part) {
  const int cols = heads * 16;
  const int groups = NT / cols;
  const int col = threadIdx.x % cols, grp = threadIdx.x / cols;
  if (grp >= groups) return;
  const long b = blockIdx.y;
  const long per = (P + gridDim.x - 1) / gridDim.x;
  const long p0 = blockIdx.x * per, p1 = min(p0 + per, P);
  float acc = 0.f;
  for (long p = p0 + grp; p < p1; p += groups) acc += Elem<T>::to_f(dt[(b * P + p) * cols + col]);
  if (part) {      // deterministic form: the groups' sums meet in LDS and are added in group order; the block's sums go to its row of `part`
    __shared__ float red[NT];      // ([B][chunks][cols]), which head_tap_sums_final_kernel adds in chunk order
    red[threadIdx.x] = acc;
    __syncthreads();
    if (grp == 0) {
      float s = 0.f;
      for (int q = 0; q < groups; q++) s += red[q * cols + col];
      part[(b * gridDim.x + blockIdx.x) * cols + col] = s;
    }
    return;
  }
  atomicAdd(dcb + b * cols + col, acc);
}
__global__ void __launch_bounds__(NT) head_tap_sums_final_kernel(const float* __restrict__ part, float* __restrict__ dcb, int chunks, int cols, int B) {
  GRID_STRIDE(i, (long)B * cols) {
    const long b = i / cols;
    const int col = (int)(i % cols);
    float s = 0.f;
    for (int c = 0; c < chunks; c++) s += part[(b * chunks + c) * cols + col];
    dcb[i] = s;
  }
}
// db5[h*C + c] += sum_{b,tap} dcb[b][h][tap] * w_b[c][tap];   dwpad[b][c][tap] += sum_h b5[h*C + c] * dcb[b][h][tap]
template <typename T>
__global__ void __launch_bounds__(NT) head_cb_bwd_kernel(const float* __restrict__ b5, const T* __restrict__ wpad, const float* __restrict__ dcb,
                                                         float* __restrict__ db5, float* __restrict__ dwpad, int B, int heads, int C, int add_db5) {
  GRID_STRIDE(i, (long)B * C) {
    const int c = (int)(i % C);
    const long b = i / C;
    float w[16], dw[16];
#pragma unroll
    for (int t = 0; t < 16; t++) { w[t] = Elem<T>::to_f(wpad[i * 16 + t]); dw[t] = 0.f; }
    for (int h = 0; h < heads; h++) {
      const float* d = dcb + (b * heads + h) * 16;
      const float bb = b5[h * C + c];
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < 16; t++) { s += d[t] * w[t]; dw[t] += bb * d[t]; }
      if (add_db5) atomicAdd(db5 + h * C + c, s);
    }
#pragma unroll
    for (int t = 0; t < 9; t++) atomicAdd(dwpad + i * 16 + t, dw[t]);
  }
}

// deterministic db5: one thread per (head, channel) sums the samples in order
template <typename T>
__global__ void __launch_bounds__(NT) head_db5_det_kernel(const T* __restrict__ wpad, const float* __restrict__ dcb, float* __restrict__ db5, int B, int heads, int C) {
  GRID_STRIDE(i, (long)heads * C) {
    const int c = (int)(i % C), h = (int)(i / C);
    float acc = 0.f;
    for (long b = 0; b < B; b++) {
      const float* d = dcb + (b * heads + h) * 16;
      const T* w = wpad + (b * C + c) * 16;
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < 16; t++) s += d[t] * Elem<T>::to_f(w[t]);
      acc += s;
    }
    db5[i] += acc;
  }
}

// dbias[b] = sum over heads, pixels of dout[b]
__global__ void __launch_bounds__(NT) head_bias_grad_kernel(const float* __restrict__ dout, float* __restrict__ dbias, long per_b) {
  __shared__ float red[NT / 64];
  const long b = blockIdx.x;
  float s = 0.f;
  const float* src = dout + b * per_b;
  if ((per_b & 3) == 0 && ((uintptr_t)src & 15) == 0) {
    // 16-byte loads, four in flight per thread (round 6: one dependent 4-byte load per iteration made this 52 us for 54080 floats per
    // sample - the first kernel of backward, on the critical chain)
    const f32x4* p4 = reinterpret_cast<const f32x4*>(src);
    const long n4 = per_b >> 2;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    long i = threadIdx.x;
    for (; i + 3 * NT < n4; i += 4 * NT) {
      const f32x4 a = p4[i], c = p4[i + NT], d = p4[i + 2 * NT], e = p4[i + 3 * NT];
      s0 += (a[0] + a[1]) + (a[2] + a[3]);
      s1 += (c[0] + c[1]) + (c[2] + c[3]);
      s2 += (d[0] + d[1]) + (d[2] + d[3]);
      s3 += (e[0] + e[1]) + (e[2] + e[3]);
    }
    for (; i < n4; i += NT) {
      const f32x4 a = p4[i];
      s0 += (a[0] + a[1]) + (a[2] + a[3]);
    }
    s = (s0 + s1) + (s2 + s3);
  } else {
    for (long i = threadIdx.x; i < per_b; i += NT) s += src[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dbias[b] = red[0] + red[1] + red[2] + red[3];
}

// ---- losses (crog.py:76-99) -----------------------------------------------------------------------------
// pred fp32 [B][heads][H][W];  tgt[h] fp32 [B][1][Hin][Win] -> nearest resize to HxW (crog.py:78-83)
// head 0: BCE-with-logits, weight = mask*0.5+1 when weighted (crog.py:90-92) else plain (crog.py:124)
// heads 1..4: smooth-L1, beta = 1 (crog.py:93-96).  All means over N = B*H*W; total = sum of means.
struct LossTargets { const float* t[5]; };
__global__ void __launch_bounds__(NT) head_loss_kernel(const float* __restrict__ pred, LossTargets tg, int B, int heads, int H, int W, int Hin, int Win,
                                                       int weighted, float* __restrict__ tgt_small, float* __restrict__ loss_sums,
                                                       float* __restrict__ dpred, float* __restrict__ part) {
  __shared__ float red[NT / 64][5];
  const long P = (long)H * W, N = (long)B * P;
  const float invN = 1.f / (float)N;
  const float sy = (float)Hin / H, sx = (float)Win / W;
  float ls[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  GRID_STRIDE(i, N) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long b = i / P;
    const int iy = min((int)floorf(y * sy), Hin - 1), ix = min((int)floorf(x * sx), Win - 1);
    for (int h = 0; h < heads; h++) {
      const float t = tg.t[h][(b * Hin + iy) * Win + ix];
      const float p = pred[((b * heads + h) * H + y) * W + x];
      tgt_small[(long)h * N + i] = t;
      float l, g;
      if (h == 0) {
        const float w = weighted ? t * 0.5f + 1.f : 1.f;
        l = w * (fmaxf(p, 0.f) - p * t + log1pf(expf(-fabsf(p))));
        g = w * (1.f / (1.f + expf(-p)) - t);
      } else {
        const float d = p - t, ad = fabsf(d);
        l = ad < 1.f ? 0.5f * d * d : ad - 0.5f;
        g = ad < 1.f ? d : (d > 0.f ? 1.f : -1.f);
      }
      ls[h] += l;
      dpred[((b * heads + h) * H + y) * W + x] = g * invN;
    }
  }
#pragma unroll
  for (int h = 0; h < 5; h++) {
    const float s = wave_sum(ls[h]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][h] = s;
  }
  __syncthreads();
  if (threadIdx.x < heads) {
    float s = 0.f;
    for (int w = 0; w < NT / 64; w++) s += red[w][threadIdx.x];
    if (part) part[blockIdx.x * 8 + threadIdx.x] = s * invN;      // deterministic form: head_loss_finalize_kernel adds the blocks in order
    else atomicAdd(loss_sums + threadIdx.x, s * invN);
  }
}
__global__ void __launch_bounds__(64) head_loss_finalize_kernel(const float* __restrict__ part, int nblocks, int heads, float* __restrict__ loss_sums) {
  if ((int)threadIdx.x < heads) {
    float s = 0.f;
    for (int b = 0; b < nblocks; b++) s += part[b * 8 + threadIdx.x];
    loss_sums[threadIdx.x] = s;
  }
}

// ---- trainMetricGPU (utils/misc.py:115-131): sigmoid, threshold, per-sample IoU, Prec@pr_iou ---------------
__global__ void __launch_bounds__(NT) metric_counts_kernel(const float* __restrict__ pred, long pred_bstride, const float* __restrict__ tgt, long P,
                                                           float thr, float* __restrict__ counts) {
  __shared__ float red[NT / 64][2];
  const long b = blockIdx.x;
  float inter = 0.f, uni = 0.f;
  for (long i = threadIdx.x; i < P; i += NT) {
    const float s = 1.f / (1.f + expf(-pred[b * pred_bstride + i]));
    const bool o = s >= thr, t = tgt[b * P + i] != 0.f;
    inter += (o && t) ? 1.f : 0.f;
    uni += (o || t) ? 1.f : 0.f;
  }
  inter = wave_sum(inter);
  uni = wave_sum(uni);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = inter; red[threadIdx.x >> 6][1] = uni; }
  __syncthreads();
  if (threadIdx.x == 0) {
    counts[2 * b] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    counts[2 * b + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
  }
}
__global__ void metric_finalize_kernel(const float* __restrict__ counts, int B, float pr_iou, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float iou = 0.f, prec = 0.f;
  for (int b = 0; b < B; b++) {
    const float v = counts[2 * b] / (counts[2 * b + 1] + 1e-6f);
    iou += v;
    prec += v > pr_iou ? 1.f : 0.f;
  }
  out[0] = 100.f * iou / B;
  out[1] = 100.f * prec / B;
}

}  // namespace

#define DISPATCH_T(dtype, ...)                                   \
  do {                                                           \
    if ((dtype) == CROG_BF16) { using T = bf16; __VA_ARGS__; }   \
    else if ((dtype) == CROG_F32) { using T = float; __VA_ARGS__; } \
    else { crog_set_error("bad dtype %d", (int)(dtype)); return CROG_ERR_ARG; } \
  } while (0)
#define LAUNCH(kern, work, stream, ...)                                                            \
  hipLaunchKernelGGL(kern, dim3(stream_grid(work)), dim3(NT), 0, (hipStream_t)(stream), __VA_ARGS__)

extern "C" int crog_head_pack_weights(int dtype, const float* word, int64_t ldw, void* wpad, int B, int C, crog_stream_t s) {
  DISPATCH_T(dtype, LAUNCH((head_pack_weights_kernel<T>), (long)B * C * 16, s, word, (long)ldw, (T*)wpad, B, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_head_unpack_wgrad(int dtype, const float* dwpad, const float* dbias, void* dword, int64_t ldd, int B, int C, crog_stream_t s) {
  CROG_CHECK_ARG(ldd >= C * 9 + 1, "head_unpack_wgrad: ldd too small");
  DISPATCH_T(dtype, LAUNCH((head_unpack_wgrad_kernel<T>), (long)B * ldd, s, dwpad, dbias, (T*)dword, (long)ldd, B, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_head_cb_fwd(int dtype, const float* b5, const void* wpad, float* cb, int B, int heads, int C, crog_stream_t s) {
  DISPATCH_T(dtype, LAUNCH((head_cb_fwd_kernel<T>), (long)B * heads * 16, s, b5, (const T*)wpad, cb, B, heads, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_head_tap_sums(int dtype, const void* dt, float* dcb, int B, int heads, int64_t P, crog_stream_t s) {
  CROG_CHECK_ARG(heads >= 1 && heads * 16 <= NT, "head_tap_sums: heads must be in [1, %d]", NT / 16);
  hipError_t e = hipMemsetAsync(dcb, 0, (size_t)B * heads * 16 * sizeof(float), (hipStream_t)s);
  if (e != hipSuccess) { crog_set_error("head_tap_sums: memset failed"); return CROG_ERR_LAUNCH; }
  if (crog_deterministic()) {
    // per-chunk sums into the library's scratch, added in chunk order by a second launch (round 5; one block per sample walked its 10816
    // pixels alone before: 0.89 ms on the critical chain of the deterministic step)
    const int cols = heads * 16;
    const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(32, CROG_DET_TAP_FLOATS / ((int64_t)B * cols)));
    float* part = crog_det_scratch() + CROG_DET_LOSS_FLOATS;
    CROG_CHECK_ARG((int64_t)B * cols <= CROG_DET_TAP_FLOATS, "head_tap_sums (deterministic): B * heads * 16 = %ld exceeds the scratch", (long)B * cols);
    DISPATCH_T(dtype, hipLaunchKernelGGL((head_tap_sums_kernel<T>), dim3(chunks, B), dim3(NT), 0, (hipStream_t)s, (const T*)dt, dcb, (long)P, heads, part));
    CROG_LAUNCH_CHECK();
    hipLaunchKernelGGL(head_tap_sums_final_kernel, dim3(cdiv((long)B * cols, NT)), dim3(NT), 0, (hipStream_t)s, part, dcb, chunks, cols, B);
    CROG_LAUNCH_CHECK();
    return CROG_OK;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL((head_tap_sums_kernel<T>), dim3(32, B), dim3(NT), 0, (hipStream_t)s, (const T*)dt, dcb, (long)P, heads, (float*)nullptr));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_head_cb_bwd(int dtype, const float* b5, const void* wpad, const float* dcb, float* db5, float* dwpad, int B, int heads, int C,
                                crog_stream_t s) {
  const int det = crog_deterministic() ? 1 : 0;
  DISPATCH_T(dtype, LAUNCH((head_cb_bwd_kernel<T>), (long)B * C, s, b5, (const T*)wpad, dcb, db5, dwpad, B, heads, C, det ? 0 : 1));
  CROG_LAUNCH_CHECK();
  if (det) {
    DISPATCH_T(dtype, LAUNCH((head_db5_det_kernel<T>), (long)heads * C, s, (const T*)wpad, dcb, db5, B, heads, C));
    CROG_LAUNCH_CHECK();
  }
  return CROG_OK;
}
extern "C" int crog_head_stencil_fwd(const float* t, const float* word, int64_t ldw, int bias_col, const float* tbias, float* out, int B, int heads,
                                     int H, int W, crog_stream_t s) {
  const size_t lds = (size_t)(HS_R + 2) * W * heads * 9 * sizeof(float);
  if (lds <= 96 * 1024 && B <= 65535) {      // LDS row-band form; the gather form below serves maps too wide for it
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(head_stencil_fwd_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      if (e != hipSuccess) { crog_set_error("head_stencil_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return CROG_ERR_LAUNCH; }
      attr_set = true;
    }
    hipLaunchKernelGGL(head_stencil_fwd_rows_kernel, dim3(cdiv(H, HS_R), B), dim3(NT), lds, (hipStream_t)s, t, word, (long)ldw, bias_col, tbias, out,
                       heads, H, W);
  } else {
    LAUNCH(head_stencil_fwd_kernel, (long)B * heads * H * W, s, t, word, (long)ldw, bias_col, tbias, out, B, heads, H, W);
  }
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_head_stencil_bwd(int dtype, const float* dout, void* dt, float* dbias, int B, int heads, int H, int W, crog_stream_t s) {
  DISPATCH_T(dtype, LAUNCH((head_stencil_bwd_kernel<T>), (long)B * heads * H * W, s, dout, (T*)dt, B, heads, H, W));
  CROG_LAUNCH_CHECK();
  hipLaunchKernelGGL(head_bias_grad_kernel, dim3(B), dim3(NT), 0, (hipStream_t)s, dout, dbias, (long)heads * H * W);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_head_loss(const float* pred, const float* const* targets, int B, int heads, int H, int W, int Hin, int Win, int weighted,
                              float* tgt_small, float* loss_sums, float* dpred, crog_stream_t s) {
  CROG_CHECK_ARG(heads >= 1 && heads <= 5, "head_loss: heads must be in [1,5]");
  LossTargets tg;
  for (int h = 0; h < 5; h++) tg.t[h] = h < heads ? targets[h] : nullptr;
  hipError_t e = hipMemsetAsync(loss_sums, 0, 5 * sizeof(float), (hipStream_t)s);
  if (e != hipSuccess) { crog_set_error("head_loss: memset failed"); return CROG_ERR_LAUNCH; }
  if (crog_deterministic()) {
    const int nb = std::min(stream_grid((long)B * H * W), CROG_DET_LOSS_FLOATS / 8);
    hipLaunchKernelGGL(head_loss_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)s, pred, tg, B, heads, H, W, Hin, Win, weighted, tgt_small, loss_sums, dpred,
                       crog_det_scratch());
    CROG_LAUNCH_CHECK();
    hipLaunchKernelGGL(head_loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, crog_det_scratch(), nb, heads, loss_sums);
    CROG_LAUNCH_CHECK();
    return CROG_OK;
  }
  LAUNCH(head_loss_kernel, (long)B * H * W, s, pred, tg, B, heads, H, W, Hin, Win, weighted, tgt_small, loss_sums, dpred, (float*)nullptr);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_train_metric(const float* pred, int64_t pred_bstride, const float* tgt, int B, int64_t P, float threshold, float pr_iou,
                                 float* counts, float* out2, crog_stream_t s) {
  hipLaunchKernelGGL(metric_counts_kernel, dim3(B), dim3(NT), 0, (hipStream_t)s, pred, (long)pred_bstride, tgt, (long)P, threshold, counts);
  CROG_LAUNCH_CHECK();
  hipLaunchKernelGGL(metric_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, counts, B, pr_iou, out2);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
