// Staging kernels for the convolutions that the implicit-GEMM forms (1x1, 3x3 stride 1) do not cover — the SSG trunk's
// 7x7/s2 stem, 3x3/s2 and 1x1/s2 convolutions (ssg.py:22,63,79,188-191) — plus MaxPool2d(3,2,1) (ssg.py:66), the
// align_corners=True bilinear x2 of ProtoNet (ssg.py:159) and tanh (ssg.py:128,133).  All HBM-bound, 16-byte vectors on
// channels-last rows.  A strided convolution is: im2col rows -> crog_gemm (K = KH*KW*C) ; its data gradient is
// crog_gemm (dcol = dz W) -> col2im, written in gather form (no atomics, deterministic).
#include "common.h"

namespace {

constexpr int NT = 256;

inline int stream_grid(long work_items) {
  long g = (work_items + NT - 1) / NT;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

#define GRID_STRIDE(i, total) for (long i = (long)blockIdx.x * NT + threadIdx.x; i < (total); i += (long)gridDim.x * NT)

struct ConvWin { int H, W, C, KH, KW, S, P, OH, OW; };

// col[(b,oy,ox)][(ky*KW + kx)*C + c] = x[b][oy*S - P + ky][ox*S - P + kx][c]   (0 outside the map)
template <typename T>
__global__ void __launch_bounds__(NT) im2col_nhwc_kernel(const T* __restrict__ x, long ldx, T* __restrict__ col, long ldo, int B, ConvWin g) {
  constexpr int VEC = Elem<T>::VEC;
  const int cvec = g.C / VEC;
  const long total = (long)B * g.OH * g.OW * g.KH * g.KW * cvec;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % cvec) * VEC;
    long p = i / cvec;
    const int kx = (int)(p % g.KW);
    p /= g.KW;
    const int ky = (int)(p % g.KH);
    p /= g.KH;
    const long row = p;
    const int ox = (int)(p % g.OW);
    p /= g.OW;
    const int oy = (int)(p % g.OH);
    const long b = p / g.OH;
    const int iy = oy * g.S - g.P + ky, ix = ox * g.S - g.P + kx;
    Vec16<T> v;
    if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) {
      v = ldg16(x + ((b * g.H + iy) * g.W + ix) * ldx + c);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; e++) v.v[e] = Elem<T>::from_f(0.f);
    }
    stg16(col + row * ldo + (long)(ky * g.KW + kx) * g.C + c, v);
  }
}

// dx[b][y][x][c] = sum over the windows (oy, ox, ky, kx) that read pixel (y, x) of dcol[(b,oy,ox)][(ky*KW + kx)*C + c]
template <typename T>
__global__ void __launch_bounds__(NT) col2im_nhwc_kernel(const T* __restrict__ dcol, long ldc, T* __restrict__ dx, long lddx, int B, ConvWin g) {
  constexpr int VEC = Elem<T>::VEC;
  const int cvec = g.C / VEC;
  const long total = (long)B * g.H * g.W * cvec;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % cvec) * VEC;
    long p = i / cvec;
    const int ix = (int)(p % g.W);
    p /= g.W;
    const int iy = (int)(p % g.H);
    const long b = p / g.H;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) acc[e] = 0.f;
    for (int ky = 0; ky < g.KH; ky++) {
      const int ty = iy + g.P - ky;
      if (ty < 0 || ty % g.S != 0) continue;
      const int oy = ty / g.S;
      if (oy >= g.OH) continue;
      for (int kx = 0; kx < g.KW; kx++) {
        const int tx = ix + g.P - kx;
        if (tx < 0 || tx % g.S != 0) continue;
        const int ox = tx / g.S;
        if (ox >= g.OW) continue;
        const Vec16<T> v = ldg16(dcol + ((b * g.OH + oy) * g.OW + ox) * ldc + (long)(ky * g.KW + kx) * g.C + c);
#pragma unroll
        for (int e = 0; e < VEC; e++) acc[e] += Elem<T>::to_f(v.v[e]);
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(acc[e]);
    stg16(dx + ((b * g.H + iy) * g.W + ix) * lddx + c, o);
  }
}

// NCHW fp32 image (any C: 3 RGB, 4 RGB-D) -> rows [B*OH*OW][ldo], column (ky*KW + kx)*C + c for < KH*KW*C, zeros up to ldo
template <typename T>
__global__ void __launch_bounds__(NT) im2col_image_kernel(const float* __restrict__ img, T* __restrict__ col, long ldo, int B, ConvWin g) {
  constexpr int VEC = Elem<T>::VEC;
  const int chunks = (int)(ldo / VEC), kcols = g.KH * g.KW * g.C;
  const long total = (long)B * g.OH * g.OW * chunks;
  GRID_STRIDE(i, total) {
    const int j0 = (int)(i % chunks) * VEC;
    long p = i / chunks;
    const long row = p;
    const int ox = (int)(p % g.OW);
    p /= g.OW;
    const int oy = (int)(p % g.OH);
    const long b = p / g.OH;
    Vec16<T> v;
#pragma unroll
    for (int e = 0; e < VEC; e++) {
      const int j = j0 + e;
      float f = 0.f;
      if (j < kcols) {
        const int c = j % g.C, t = j / g.C;
        const int kx = t % g.KW, ky = t / g.KW;
        const int iy = oy * g.S - g.P + ky, ix = ox * g.S - g.P + kx;
        if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) f = img[((b * g.C + c) * g.H + iy) * g.W + ix];
      }
      v.v[e] = Elem<T>::from_f(f);
    }
    stg16(col + row * ldo + j0, v);
  }
}

// MaxPool2d(kernel 3, stride 2, padding 1).  arg[...] = window position ky*3+kx of the maximum (first one in scan order, as ATen)
template <typename T>
__global__ void __launch_bounds__(NT) maxpool3s2_fwd_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, unsigned char* __restrict__ arg,
                                                            int B, int H, int W, int C, int OH, int OW) {
  constexpr int VEC = Elem<T>::VEC;
  const int cvec = C / VEC;
  const long total = (long)B * OH * OW * cvec;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % cvec) * VEC;
    long p = i / cvec;
    const long orow = p;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const long b = p / OH;
    float best[VEC];
    unsigned char bi[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) { best[e] = -INFINITY; bi[e] = 0; }
    bool first = true;
    for (int ky = 0; ky < 3; ky++) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; kx++) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const Vec16<T> v = ldg16(x + ((b * H + iy) * W + ix) * ldx + c);
#pragma unroll
        for (int e = 0; e < VEC; e++) {
          const float f = Elem<T>::to_f(v.v[e]);
          if (first || f > best[e] || f != f) { best[e] = f; bi[e] = (unsigned char)(ky * 3 + kx); }
        }
        first = false;
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(best[e]);
    stg16(y + orow * ldy + c, o);
#pragma unroll
    for (int e = 0; e < VEC; e++) arg[orow * C + c + e] = bi[e];
  }
}
template <typename T>
__global__ void __launch_bounds__(NT) maxpool3s2_bwd_kernel(const T* __restrict__ dy, long lddy, const unsigned char* __restrict__ arg, T* __restrict__ dx,
                                                            long lddx, int B, int H, int W, int C, int OH, int OW) {
  constexpr int VEC = Elem<T>::VEC;
  const int cvec = C / VEC;
  const long total = (long)B * H * W * cvec;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % cvec) * VEC;
    long p = i / cvec;
    const int ix = (int)(p % W);
    p /= W;
    const int iy = (int)(p % H);
    const long b = p / H;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) acc[e] = 0.f;
    for (int oy = iy / 2; oy <= (iy + 1) / 2; oy++) {       // windows with 2*oy - 1 <= iy <= 2*oy + 1
      if (oy >= OH) continue;
      const int ky = iy - (2 * oy - 1);
      for (int ox = ix / 2; ox <= (ix + 1) / 2; ox++) {
        if (ox >= OW) continue;
        const int k = ky * 3 + (ix - (2 * ox - 1));
        const long orow = (b * OH + oy) * OW + ox;
        const Vec16<T> g = ldg16(dy + orow * lddy + c);
#pragma unroll
        for (int e = 0; e < VEC; e++)
          if (arg[orow * C + c + e] == k) acc[e] += Elem<T>::to_f(g.v[e]);
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(acc[e]);
    stg16(dx + ((b * H + iy) * W + ix) * lddx + c, o);
  }
}

// bilinear x2, align_corners=True: source coordinate s(d) = d * (n_in - 1) / (n_out - 1)
template <typename T>
__global__ void __launch_bounds__(NT) upsample2ac_fwd_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, int B, int H, int W, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int OH = 2 * H, OW = 2 * W, cvec = C / VEC;
  const float sy = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f, sx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const long total = (long)B * OH * OW * cvec;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % cvec) * VEC;
    long p = i / cvec;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const long b = p / OH;
    const float fy = oy * sy, fx = ox * sx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float wy1 = fy - y0, wx1 = fx - x0, wy0 = 1.f - wy1, wx0 = 1.f - wx1;
    const T* xb = x + (b * H * W) * ldx + c;
    const Vec16<T> v00 = ldg16(xb + ((long)y0 * W + x0) * ldx), v01 = ldg16(xb + ((long)y0 * W + x1) * ldx);
    const Vec16<T> v10 = ldg16(xb + ((long)y1 * W + x0) * ldx), v11 = ldg16(xb + ((long)y1 * W + x1) * ldx);
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; e++)
      o.v[e] = Elem<T>::from_f(wy0 * (wx0 * Elem<T>::to_f(v00.v[e]) + wx1 * Elem<T>::to_f(v01.v[e])) +
                               wy1 * (wx0 * Elem<T>::to_f(v10.v[e]) + wx1 * Elem<T>::to_f(v11.v[e])));
    stg16(y + ((b * OH + oy) * OW + ox) * ldy + c, o);
  }
}
// gather form of the transpose: the weight of input i in output d is the tent max(0, 1 - |s(d) - i|)
template <typename T>
__global__ void __launch_bounds__(NT) upsample2ac_bwd_kernel(const T* __restrict__ dy, long lddy, T* __restrict__ dx, long lddx, int B, int H, int W, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int OH = 2 * H, OW = 2 * W, cvec = C / VEC;
  const float sy = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f, sx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const long total = (long)B * H * W * cvec;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % cvec) * VEC;
    long p = i / cvec;
    const int ix = (int)(p % W);
    p /= W;
    const int iy = (int)(p % H);
    const long b = p / H;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) acc[e] = 0.f;
    const int dy0 = max(0, 2 * iy - 3), dy1 = min(OH - 1, 2 * iy + 4), dx0 = max(0, 2 * ix - 3), dx1 = min(OW - 1, 2 * ix + 4);
    for (int oy = dy0; oy <= dy1; oy++) {
      // same arithmetic as the forward: floor + fractional weight, so the two passes are exact transposes
      const float fy = oy * sy;
      const int y0 = (int)fy;
      const int y1 = y0 + (y0 < H - 1 ? 1 : 0);
      const float wy1 = fy - y0;
      const float wy = (iy == y0 ? 1.f - wy1 : 0.f) + (iy == y1 ? wy1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = dx0; ox <= dx1; ox++) {
        const float fx = ox * sx;
        const int x0 = (int)fx;
        const int x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float wx1 = fx - x0;
        const float wx = (ix == x0 ? 1.f - wx1 : 0.f) + (ix == x1 ? wx1 : 0.f);
        if (wx == 0.f) continue;
        const Vec16<T> g = ldg16(dy + ((b * OH + oy) * OW + ox) * lddy + c);
        const float w = wy * wx;
#pragma unroll
        for (int e = 0; e < VEC; e++) acc[e] += w * Elem<T>::to_f(g.v[e]);
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(acc[e]);
    stg16(dx + ((b * H + iy) * W + ix) * lddx + c, o);
  }
}

#define DISPATCH_T(dtype, ...)                                   \
  do {                                                           \
    if ((dtype) == CROG_BF16) { using T = bf16; __VA_ARGS__; }   \
    else if ((dtype) == CROG_F32) { using T = float; __VA_ARGS__; } \
    else { crog_set_error("bad dtype %d", (int)(dtype)); return CROG_ERR_ARG; } \
  } while (0)
#define VECOF(dtype) ((dtype) == CROG_BF16 ? 8 : 4)
#define LAUNCH(kern, work, stream, ...)                                                            \
  hipLaunchKernelGGL(kern, dim3(stream_grid(work)), dim3(NT), 0, (hipStream_t)(stream), __VA_ARGS__)

inline bool win_ok(int H, int W, int KH, int KW, int S, int P, int OH, int OW) {
  return H > 0 && W > 0 && KH > 0 && KW > 0 && S > 0 && P >= 0 && OH == (H + 2 * P - KH) / S + 1 && OW == (W + 2 * P - KW) / S + 1;
}

// Evaluation maps (crog_engine.py:181-211): out[b][g] = bicubic(align_corners=True, A = -0.75)( g in sigmoid_mask ? sigmoid(x) : x ).
// Planar fp32 [B*G][h][w] -> [B*G][H][W]; one thread per 4 consecutive output columns (one 16-byte store: the 16x larger output is
// the HBM traffic, the input plane set stays in L2).  Taps are clamped to the plane like ATen's upsample_get_value_bounded.
__device__ inline void cubic_taps(float t, float w[4]) {
  constexpr float A = -0.75f;
  const float t1 = t + 1.f, u = 1.f - t, u1 = 2.f - t;
  w[0] = ((A * t1 - 5.f * A) * t1 + 8.f * A) * t1 - 4.f * A;
  w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
  w[2] = ((A + 2.f) * u - (A + 3.f)) * u * u + 1.f;
  w[3] = ((A * u1 - 5.f * A) * u1 + 8.f * A) * u1 - 4.f * A;
}
template <bool SIG>
__device__ inline float eval_tap(const float* __restrict__ p) {
  const float v = *p;
  return SIG ? 1.f / (1.f + __expf(-v)) : v;
}
template <bool SIG>
__device__ inline void eval_bicubic4(const float* __restrict__ plane, int h, int w, float sy, float sx, int oy, int ox0, int W, float out[4]) {
#pragma clang fp contract(off)   // t = fy - y0 must see the ROUNDED product sy*oy, as ATen forms it: fused, the phase moves by the product's
                                 // rounding error (4e-6 near source index 100) and a +-10 logit map by 1e-4
  const float fy = sy * (float)oy;
  const int y0 = (int)floorf(fy);
  float wy[4];
  cubic_taps(fy - y0, wy);
  const float* rows[4];
#pragma unroll
  for (int j = 0; j < 4; j++) rows[j] = plane + (long)min(max(y0 - 1 + j, 0), h - 1) * w;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int ox = min(ox0 + e, W - 1);
    const float fx = sx * (float)ox;
    const int x0 = (int)floorf(fx);
    float wx[4];
    cubic_taps(fx - x0, wx);
    int xs[4];
#pragma unroll
    for (int i = 0; i < 4; i++) xs[i] = min(max(x0 - 1 + i, 0), w - 1);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float r = 0.f;
#pragma unroll
      for (int i = 0; i < 4; i++) r += wx[i] * eval_tap<SIG>(rows[j] + xs[i]);
      acc += wy[j] * r;
    }
    out[e] = acc;
  }
}
// Tiled form for the usual 4x enlargement: a block produces 16 rows x 64 columns of one plane; the few input rows / columns those
// outputs touch (8 x 20 for 104 -> 416) are staged in LDS once, with the sigmoid already applied (once per input value instead of
// once per tap: 16x fewer exp / rcp), and every thread gathers its 4 x 16 taps from LDS instead of global memory.
constexpr int EV_TR = 12, EV_TC = 28;   // LDS tile bounds (input rows x columns); the launcher checks that the scale fits
__global__ void __launch_bounds__(NT) eval_maps_tiled_kernel(const float* __restrict__ x, int G, int h, int w, unsigned sigmoid_mask,
                                                             float* __restrict__ y, int H, int W, float sy, float sx) {
#pragma clang fp contract(off)   // source coordinates as ATen forms them (see eval_bicubic4)
  __shared__ float tile[EV_TR][EV_TC + 1];
  const int pl = blockIdx.z, oy0 = blockIdx.y * 16, ox0 = blockIdx.x * 64;
  const int oy1 = min(oy0 + 15, H - 1), ox1 = min(ox0 + 63, W - 1);
  const int ylo = min(max((int)floorf(sy * (float)oy0) - 1, 0), h - 1), yhi = min(max((int)floorf(sy * (float)oy1) + 2, 0), h - 1);
  const int xlo = min(max((int)floorf(sx * (float)ox0) - 1, 0), w - 1), xhi = min(max((int)floorf(sx * (float)ox1) + 2, 0), w - 1);
  const int tr = yhi - ylo + 1, tc = xhi - xlo + 1;
  const bool sig = (sigmoid_mask >> (pl % G)) & 1u;
  const float* plane = x + (long)pl * h * w;
  for (int i = threadIdx.x; i < tr * tc; i += NT) {
    const int r = i / tc, c = i - r * tc;
    const float v = plane[(long)(ylo + r) * w + xlo + c];
    tile[r][c] = sig ? 1.f / (1.f + __expf(-v)) : v;
  }
  __syncthreads();
  const int oy = oy0 + (int)(threadIdx.x >> 4), oxq = ox0 + (int)(threadIdx.x & 15) * 4;
  if (oy >= H || oxq >= W) return;
  const float fy = sy * (float)oy;
  const int y0 = (int)floorf(fy);
  float wy[4];
  cubic_taps(fy - y0, wy);
  int rr[4];
#pragma unroll
  for (int j = 0; j < 4; j++) rr[j] = min(max(y0 - 1 + j, 0), h - 1) - ylo;
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int ox = min(oxq + e, W - 1);
    const float fx = sx * (float)ox;
    const int x0 = (int)floorf(fx);
    float wx[4];
    cubic_taps(fx - x0, wx);
    int cc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) cc[i] = min(max(x0 - 1 + i, 0), w - 1) - xlo;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float r = 0.f;
#pragma unroll
      for (int i = 0; i < 4; i++) r += wx[i] * tile[rr[j]][cc[i]];
      acc += wy[j] * r;
    }
    o[e] = acc;
  }
  float* dst = y + ((long)pl * H + oy) * W + oxq;
  if (oxq + 4 <= W && (W & 3) == 0) {
    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
    for (int e = 0; e < 4 && oxq + e < W; e++) dst[e] = o[e];
  }
}

__global__ void __launch_bounds__(NT) eval_maps_kernel(const float* __restrict__ x, int planes, int G, int h, int w, unsigned sigmoid_mask,
                                                       float* __restrict__ y, int H, int W, float sy, float sx) {
  const int wq = (W + 3) / 4;
  const long total = (long)planes * H * wq;
  GRID_STRIDE(i, total) {
    const int q = (int)(i % wq);
    long p = i / wq;
    const int oy = (int)(p % H);
    const int pl = (int)(p / H);
    const float* plane = x + (long)pl * h * w;
    float o[4];
    if ((sigmoid_mask >> (pl % G)) & 1u) eval_bicubic4<true>(plane, h, w, sy, sx, oy, q * 4, W, o);
    else eval_bicubic4<false>(plane, h, w, sy, sx, oy, q * 4, W, o);
    float* dst = y + ((long)pl * H + oy) * W + q * 4;
    if (q * 4 + 4 <= W && (W & 3) == 0) {
      *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
      for (int e = 0; e < 4 && q * 4 + e < W; e++) dst[e] = o[e];
    }
  }
}

}  // namespace

extern "C" int crog_im2col_nhwc(int dtype, const void* x, int64_t ldx, void* col, int64_t ldo, int B, int H, int W, int C, int KH, int KW, int S,
                                int P, int OH, int OW, crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(win_ok(H, W, KH, KW, S, P, OH, OW), "im2col: output size does not match floor((H + 2P - K)/S) + 1");
  CROG_CHECK_ARG(C % vec == 0 && ldx % vec == 0 && ldo % vec == 0 && ldo >= (int64_t)KH * KW * C, "im2col: C %% %d == 0 and ldo >= KH*KW*C required", vec);
  const ConvWin g{H, W, C, KH, KW, S, P, OH, OW};
  DISPATCH_T(dtype, LAUNCH((im2col_nhwc_kernel<T>), (long)B * OH * OW * KH * KW * (C / vec), s, (const T*)x, (long)ldx, (T*)col, (long)ldo, B, g));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_col2im_nhwc(int dtype, const void* dcol, int64_t ldc, void* dx, int64_t lddx, int B, int H, int W, int C, int KH, int KW, int S,
                                int P, int OH, int OW, crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(win_ok(H, W, KH, KW, S, P, OH, OW), "col2im: output size does not match floor((H + 2P - K)/S) + 1");
  CROG_CHECK_ARG(C % vec == 0 && ldc % vec == 0 && lddx % vec == 0, "col2im: C %% %d == 0 required", vec);
  const ConvWin g{H, W, C, KH, KW, S, P, OH, OW};
  DISPATCH_T(dtype, LAUNCH((col2im_nhwc_kernel<T>), (long)B * H * W * (C / vec), s, (const T*)dcol, (long)ldc, (T*)dx, (long)lddx, B, g));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_im2col_image(int dtype, const float* img, void* col, int64_t ldo, int B, int C, int H, int W, int KH, int KW, int S, int P,
                                 int OH, int OW, crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(win_ok(H, W, KH, KW, S, P, OH, OW), "im2col_image: output size does not match floor((H + 2P - K)/S) + 1");
  CROG_CHECK_ARG(C > 0 && ldo % vec == 0 && ldo >= (int64_t)KH * KW * C, "im2col_image: ldo must be a multiple of %d and >= KH*KW*C", vec);
  const ConvWin g{H, W, C, KH, KW, S, P, OH, OW};
  DISPATCH_T(dtype, LAUNCH((im2col_image_kernel<T>), (long)B * OH * OW * (ldo / vec), s, img, (T*)col, (long)ldo, B, g));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_maxpool3s2_fwd(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, void* argmax, int B, int H, int W, int C,
                                   crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && ldx % vec == 0 && ldy % vec == 0 && H > 0 && W > 0, "maxpool3s2: C %% %d == 0 required", vec);
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  DISPATCH_T(dtype, LAUNCH((maxpool3s2_fwd_kernel<T>), (long)B * OH * OW * (C / vec), s, (const T*)x, (long)ldx, (T*)y, (long)ldy,
                           (unsigned char*)argmax, B, H, W, C, OH, OW));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_maxpool3s2_bwd(int dtype, const void* dy, int64_t lddy, const void* argmax, void* dx, int64_t lddx, int B, int H, int W, int C,
                                   crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && lddy % vec == 0 && lddx % vec == 0 && H > 0 && W > 0, "maxpool3s2_bwd: C %% %d == 0 required", vec);
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  DISPATCH_T(dtype, LAUNCH((maxpool3s2_bwd_kernel<T>), (long)B * H * W * (C / vec), s, (const T*)dy, (long)lddy, (const unsigned char*)argmax,
                           (T*)dx, (long)lddx, B, H, W, C, OH, OW));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_eval_maps(const float* x, int B, int G, int h, int w, int sigmoid_mask, float* y, int H, int W, crog_stream_t s) {
  CROG_CHECK_ARG(B >= 0 && G >= 1 && G <= 32 && h >= 1 && w >= 1 && H >= 1 && W >= 1, "eval_maps: bad shape");
  if (B == 0) return CROG_OK;
  // align_corners=True source scale (n_in - 1) / (n_out - 1), formed once on the host in fp32 as ATen does
  const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  // the LDS-tiled form needs the inputs of a 16 x 64 output tile to fit EV_TR x EV_TC (any enlargement >= ~2.7x does)
  const bool tiled = (int)(sy * 15.f) + 5 <= EV_TR && (int)(sx * 63.f) + 5 <= EV_TC && (long)B * G <= 65535;
  if (tiled) {
    hipLaunchKernelGGL(eval_maps_tiled_kernel, dim3(cdiv(W, 64), cdiv(H, 16), B * G), dim3(NT), 0, (hipStream_t)s, x, G, h, w,
                       (unsigned)sigmoid_mask, y, H, W, sy, sx);
  } else {
    LAUNCH(eval_maps_kernel, (long)B * G * H * ((W + 3) / 4), s, x, B * G, G, h, w, (unsigned)sigmoid_mask, y, H, W, sy, sx);
  }
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_upsample2ac_fwd(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int H, int W, int C, crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && ldx % vec == 0 && ldy % vec == 0, "upsample2ac: C %% %d == 0 required", vec);
  DISPATCH_T(dtype, LAUNCH((upsample2ac_fwd_kernel<T>), (long)B * 4 * H * W * (C / vec), s, (const T*)x, (long)ldx, (T*)y, (long)ldy, B, H, W, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_upsample2ac_bwd(int dtype, const void* dy, int64_t lddy, void* dx, int64_t lddx, int B, int H, int W, int C, crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && lddy % vec == 0 && lddx % vec == 0, "upsample2ac_bwd: C %% %d == 0 required", vec);
  DISPATCH_T(dtype, LAUNCH((upsample2ac_bwd_kernel<T>), (long)B * H * W * (C / vec), s, (const T*)dy, (long)lddy, (T*)dx, (long)lddx, B, H, W, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
