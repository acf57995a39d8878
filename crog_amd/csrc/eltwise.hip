// Streaming (HBM-bound) kernels: pooling, bilinear x2, embedding, broadcasts, dropout/activation
// backward, layout staging for the stem, casts and the fused Adam step.  16-byte vectors everywhere.
#include "common.h"

namespace {

constexpr int NT = 256;

inline int stream_grid(long work_items) {
  long g = (work_items + NT - 1) / NT;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

#define GRID_STRIDE(i, total) for (long i = (long)blockIdx.x * NT + threadIdx.x; i < (total); i += (long)gridDim.x * NT)

// ---- AvgPool2d(2)  (clip.py:23,35,184; layers.py:386) -------------------------------------------
template <typename T>
__global__ void __launch_bounds__(NT) avgpool2_fwd_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, int B, int H,
                                                          int W, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int OH = H / 2, OW = W / 2, cvec = C / VEC;
  const long total = (long)B * OH * OW * cvec;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % cvec) * VEC;
    long p = i / cvec;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const long b = p / OH;
    const long base = ((b * H + 2 * oy) * W + 2 * ox);
    Vec16<T> a = ldg16(x + base * ldx + c), bb = ldg16(x + (base + 1) * ldx + c);
    Vec16<T> cc = ldg16(x + (base + W) * ldx + c), d = ldg16(x + (base + W + 1) * ldx + c);
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; e++)
      o.v[e] = Elem<T>::from_f(0.25f * (Elem<T>::to_f(a.v[e]) + Elem<T>::to_f(bb.v[e]) + Elem<T>::to_f(cc.v[e]) + Elem<T>::to_f(d.v[e])));
    stg16(y + ((b * OH + oy) * OW + ox) * ldy + c, o);
  }
}
template <typename T>
__global__ void __launch_bounds__(NT) avgpool2_bwd_kernel(const T* __restrict__ dy, long lddy, const T* __restrict__ add, long ldadd,
                                                          T* __restrict__ dx, long lddx, int B, int H, int W, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int OH = H / 2, OW = W / 2, cvec = C / VEC;
  const long total = (long)B * OH * OW * cvec;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % cvec) * VEC;
    long p = i / cvec;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const long b = p / OH;
    Vec16<T> g = ldg16(dy + ((b * OH + oy) * OW + ox) * lddy + c);
#pragma unroll
    for (int e = 0; e < VEC; e++) g.v[e] = Elem<T>::from_f(0.25f * Elem<T>::to_f(g.v[e]));
    const long base = ((b * H + 2 * oy) * W + 2 * ox);
    if (add == nullptr) {
      stg16(dx + base * lddx + c, g);
      stg16(dx + (base + 1) * lddx + c, g);
      stg16(dx + (base + W) * lddx + c, g);
      stg16(dx + (base + W + 1) * lddx + c, g);
    } else {      // + the gradient another consumer of the pooled map's input left behind (a pyramid level that also feeds the neck)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const long px = base + (q >> 1) * W + (q & 1);
        Vec16<T> a = ldg16(add + px * ldadd + c), o;
#pragma unroll
        for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(Elem<T>::to_f(g.v[e]) + Elem<T>::to_f(a.v[e]));
        stg16(dx + px * lddx + c, o);
      }
    }
  }
}

// ---- bilinear x2, align_corners=False (nn.Upsample / F.interpolate: layers.py:54,56,382,393) -----
struct Lin2 { int i0, i1; float w0, w1; };
__device__ inline Lin2 src_index_x2(int d, int n_in) {
  float s = (d + 0.5f) * 0.5f - 0.5f;
  if (s < 0.f) s = 0.f;
  Lin2 r;
  r.i0 = (int)s;
  r.i1 = r.i0 + (r.i0 < n_in - 1 ? 1 : 0);
  r.w1 = s - r.i0;
  r.w0 = 1.f - r.w1;
  return r;
}
template <typename T>
__global__ void __launch_bounds__(NT) upsample2_fwd_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, int B, int H,
                                                           int W, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int OH = 2 * H, OW = 2 * W, cvec = C / VEC;
  const long total = (long)B * OH * OW * cvec;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % cvec) * VEC;
    long p = i / cvec;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const long b = p / OH;
    const Lin2 ly = src_index_x2(oy, H), lx = src_index_x2(ox, W);
    const T* xb = x + (b * H * W) * ldx + c;
    Vec16<T> v00 = ldg16(xb + ((long)ly.i0 * W + lx.i0) * ldx), v01 = ldg16(xb + ((long)ly.i0 * W + lx.i1) * ldx);
    Vec16<T> v10 = ldg16(xb + ((long)ly.i1 * W + lx.i0) * ldx), v11 = ldg16(xb + ((long)ly.i1 * W + lx.i1) * ldx);
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; e++) {
      const float top = lx.w0 * Elem<T>::to_f(v00.v[e]) + lx.w1 * Elem<T>::to_f(v01.v[e]);
      const float bot = lx.w0 * Elem<T>::to_f(v10.v[e]) + lx.w1 * Elem<T>::to_f(v11.v[e]);
      o.v[e] = Elem<T>::from_f(ly.w0 * top + ly.w1 * bot);
    }
    stg16(y + ((b * OH + oy) * OW + ox) * ldy + c, o);
  }
}
// Workgroup bid of an n-block launch runs on XCD bid % 8: hand each XCD a CONTIGUOUS band of the logical block order, so that blocks whose
// windows overlap (neighbouring image rows) meet in one L2 instead of fetching the shared rows into two.
__device__ inline unsigned xcd_band(unsigned bid, unsigned n) {
  const unsigned x = bid & 7, per = n >> 3, rem = n & 7;
  return x * per + min(x, rem) + (bid >> 3);
}
// The two kernels the launchers use.  The per-output forms above / below spend ~150 (forward) / ~500-1000 (backward: a 6 x 6 candidate loop
// with branches around every load) instructions per 16-byte vector and ran at 2.8 TB/s on the projector's 346112 x 512 map; these are
// organised around what neighbouring outputs share.
// Forward, by PATCH: the 2 x 2 outputs (2 py + 1 .. 2 py + 2) x (2 px + 1 .. 2 px + 2) all interpolate the same four inputs (py, py + 1) x
// (px, px + 1) with weights {0.75, 0.25}; patches py = -1 .. H - 1 with clamped input indices cover every output, borders included (an
// index clamped onto its neighbour IS align_corners = False's border rule).  Four loads, four stores, one index computation.
// grid B * (H + 1) * chunks, chunks = ceil((W + 1) * C / VEC / NT) column chunks per patch row, taken in xcd_band order
template <typename T>
__global__ void __launch_bounds__(NT) upsample2_fwd_patch_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, int H, int W, int C,
                                                                 int cshift, int chunks) {
  constexpr int VEC = Elem<T>::VEC;
  const int cvec = C / VEC, OH = 2 * H, OW = 2 * W;
  const unsigned lb = xcd_band(blockIdx.x, gridDim.x), rowi = lb / (unsigned)chunks;
  const int b = rowi / (H + 1), py = (int)(rowi - (unsigned)b * (H + 1)) - 1;
  const int j = (int)(lb - rowi * (unsigned)chunks) * NT + threadIdx.x;
  if (j >= (W + 1) * cvec) return;
  const int pxi = cshift >= 0 ? j >> cshift : j / cvec;
  const int c = (j - pxi * cvec) * VEC, px = pxi - 1;
  const int y0 = max(py, 0), y1 = min(py + 1, H - 1), x0 = max(px, 0), x1 = min(px + 1, W - 1);
  const T* xb = x + ((long)b * H * W) * ldx + c;
  const Vec16<T> v00 = ldg16(xb + ((long)y0 * W + x0) * ldx), v01 = ldg16(xb + ((long)y0 * W + x1) * ldx);
  const Vec16<T> v10 = ldg16(xb + ((long)y1 * W + x0) * ldx), v11 = ldg16(xb + ((long)y1 * W + x1) * ldx);
  Vec16<T> o[2][2];
#pragma unroll
  for (int e = 0; e < VEC; e++) {
    const float a = Elem<T>::to_f(v00.v[e]), bb = Elem<T>::to_f(v01.v[e]), cc = Elem<T>::to_f(v10.v[e]), d = Elem<T>::to_f(v11.v[e]);
    const float t0 = 0.75f * a + 0.25f * bb, t1 = 0.25f * a + 0.75f * bb;       // row y0 at columns 2 px + 1, 2 px + 2
    const float u0 = 0.75f * cc + 0.25f * d, u1 = 0.25f * cc + 0.75f * d;      // row y1
    o[0][0].v[e] = Elem<T>::from_f(0.75f * t0 + 0.25f * u0);
    o[0][1].v[e] = Elem<T>::from_f(0.75f * t1 + 0.25f * u1);
    o[1][0].v[e] = Elem<T>::from_f(0.25f * t0 + 0.75f * u0);
    o[1][1].v[e] = Elem<T>::from_f(0.25f * t1 + 0.75f * u1);
  }
  T* yb = y + ((long)b * OH * OW) * ldy + c;
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int oy = 2 * py + 1 + r;
    if (oy < 0 || oy >= OH) continue;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int ox = 2 * px + 1 + q;
      if (ox >= 0 && ox < OW) stg16(yb + ((long)oy * OW + ox) * ldy, o[r][q]);
    }
  }
}
// Backward (the transpose, gather form), a 2 x 2 block of inputs per thread: input i receives outputs 2 i - 1 .. 2 i + 2 with weights
// {0.25, 0.75, 0.75, 0.25} (border: the missing output's weight goes to its neighbour - {-, 1, 0.75, 0.25} at i = 0, {0.25, 0.75, 1, -} at
// i = n - 1), separably.  The block shares a 6 x 6 window of outputs: 36 unconditional loads (clamped addresses; an absent output enters as
// zero bits, never as 0 x inf) for four inputs instead of 64, each row reduced horizontally as it arrives (two 8-float sums per row, not a
// 6-column register tile).  grid B * ceil(H / 2) * chunks, chunks = ceil(ceil(W / 2) * C / VEC / NT), in xcd_band order: vertically
// neighbouring blocks share two of their six rows
template <typename T>
__global__ void __launch_bounds__(NT) upsample2_bwd_quad_kernel(const T* __restrict__ dy, long lddy, T* __restrict__ dx, long lddx, int H, int W, int C,
                                                                int cshift, int chunks) {
  constexpr int VEC = Elem<T>::VEC;
  const int cvec = C / VEC, OH = 2 * H, OW = 2 * W, pairs = (W + 1) >> 1, rpairs = (H + 1) >> 1;
  const unsigned lb = xcd_band(blockIdx.x, gridDim.x), rowi = lb / (unsigned)chunks;
  const int b = rowi / rpairs, iy0 = 2 * (int)(rowi - (unsigned)b * rpairs), iy1 = iy0 + 1;
  const int j = (int)(lb - rowi * (unsigned)chunks) * NT + threadIdx.x;
  if (j >= pairs * cvec) return;
  const int pj = cshift >= 0 ? j >> cshift : j / cvec;
  const int c = (j - pj * cvec) * VEC, ix0 = 2 * pj, ix1 = ix0 + 1;
  float wa[4] = {0.25f, 0.75f, 0.75f, 0.25f}, wb[4] = {0.25f, 0.75f, 0.75f, 0.25f};      // columns 2 ix0 - 1 + k: input ix0 takes k = 0..3, ix1 k = 2..5
  if (ix0 == 0) wa[1] = 1.f;
  if (ix0 == W - 1) wa[2] = 1.f;
  if (ix1 == W - 1) wb[2] = 1.f;
  float va[4] = {0.25f, 0.75f, 0.75f, 0.25f}, vb[4] = {0.25f, 0.75f, 0.75f, 0.25f};      // rows 2 iy0 - 1 + r: input iy0 takes r = 0..3, iy1 r = 2..5
  if (iy0 == 0) va[1] = 1.f;
  if (iy0 == H - 1) va[2] = 1.f;
  if (iy1 == H - 1) vb[2] = 1.f;
  const T* gb = dy + ((long)b * OH * OW) * lddy + c;
  long coff[6];
  bool cok[6];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const int ox = 2 * ix0 - 1 + k;
    cok[k] = ox >= 0 && ox < OW;
    coff[k] = (long)min(max(ox, 0), OW - 1) * lddy;
  }
  float acc[2][2][VEC];
#pragma unroll
  for (int e = 0; e < VEC; e++) acc[0][0][e] = acc[0][1][e] = acc[1][0][e] = acc[1][1][e] = 0.f;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const int oy = 2 * iy0 - 1 + r;
    const bool rok = oy >= 0 && oy < OH;
    const T* row = gb + ((long)min(max(oy, 0), OH - 1) * OW) * lddy;
    Vec16<T> g[6];
#pragma unroll
    for (int k = 0; k < 6; k++) g[k] = ldg16(row + coff[k]);
    float ha[VEC], hb[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) {
      float f[6];
#pragma unroll
      for (int k = 0; k < 6; k++) f[k] = (rok && cok[k]) ? Elem<T>::to_f(g[k].v[e]) : 0.f;
      ha[e] = wa[0] * f[0] + wa[1] * f[1] + wa[2] * f[2] + wa[3] * f[3];
      hb[e] = wb[0] * f[2] + wb[1] * f[3] + wb[2] * f[4] + wb[3] * f[5];
    }
    if (r < 4) {
#pragma unroll
      for (int e = 0; e < VEC; e++) { acc[0][0][e] += va[r] * ha[e]; acc[0][1][e] += va[r] * hb[e]; }
    }
    if (r >= 2) {
#pragma unroll
      for (int e = 0; e < VEC; e++) { acc[1][0][e] += vb[r - 2] * ha[e]; acc[1][1][e] += vb[r - 2] * hb[e]; }
    }
  }
#pragma unroll
  for (int rr = 0; rr < 2; rr++) {
    if (iy0 + rr >= H) continue;
    T* ob = dx + (((long)b * H + iy0 + rr) * W) * lddx + c;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      if (ix0 + q >= W) continue;
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(acc[rr][q][e]);
      stg16(ob + (long)(ix0 + q) * lddx, o);
    }
  }
}
// gather form of the transpose: input pixel i receives from outputs d in [2i-2, 2i+3]
template <typename T>
__global__ void __launch_bounds__(NT) upsample2_bwd_kernel(const T* __restrict__ dy, long lddy, T* __restrict__ dx, long lddx, int B, int H,
                                                           int W, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int OH = 2 * H, OW = 2 * W, cvec = C / VEC;
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
    for (int oy = max(0, 2 * iy - 2); oy <= min(OH - 1, 2 * iy + 3); oy++) {
      const Lin2 ly = src_index_x2(oy, H);
      const float wy = (ly.i0 == iy ? ly.w0 : 0.f) + (ly.i1 == iy ? ly.w1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = max(0, 2 * ix - 2); ox <= min(OW - 1, 2 * ix + 3); ox++) {
        const Lin2 lx = src_index_x2(ox, W);
        const float wx = (lx.i0 == ix ? lx.w0 : 0.f) + (lx.i1 == ix ? lx.w1 : 0.f);
        if (wx == 0.f) continue;
        Vec16<T> g = ldg16(dy + ((b * OH + oy) * OW + ox) * lddy + c);
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

// ---- token embedding + learned positions (clip.py:440-443) ----------------------------------------
template <typename T>
__global__ void __launch_bounds__(NT) embedding_fwd_kernel(const int64_t* __restrict__ word, const T* __restrict__ tok, const T* __restrict__ pos,
                                                           T* __restrict__ out, long rows, int L, int C, int vocab) {
  constexpr int VEC = Elem<T>::VEC;
  const int cvec = C / VEC;
  GRID_STRIDE(i, rows * cvec) {
    const long r = i / cvec;
    const int c = (int)(i % cvec) * VEC;
    long id = word[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    Vec16<T> a = ldg16(tok + id * C + c), p = ldg16(pos + (r % L) * C + c), o;
#pragma unroll
    for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(Elem<T>::to_f(a.v[e]) + Elem<T>::to_f(p.v[e]));
    stg16(out + r * C + c, o);
  }
}
template <typename T>
__global__ void __launch_bounds__(NT) embedding_bwd_kernel(const int64_t* __restrict__ word, const T* __restrict__ dout, float* __restrict__ dtok,
                                                           float* __restrict__ dpos, long rows, int L, int C, int vocab) {
  GRID_STRIDE(i, rows * C) {
    const long r = i / C;
    const int c = (int)(i % C);
    long id = word[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float g = Elem<T>::to_f(dout[r * C + c]);
    atomicAdd(dtok + id * C + c, g);
    atomicAdd(dpos + (r % L) * C + c, g);
  }
}

// Deterministic form: blocks [0, rows) own the token rows - the block of the FIRST row that carries a token id sums every row with that
// id, in row order, the others leave at once - and blocks [rows, rows + L) own the position rows (sum over the batch, in order).
template <typename T>
__global__ void __launch_bounds__(NT) embedding_bwd_det_kernel(const int64_t* __restrict__ word, const T* __restrict__ dout, float* __restrict__ dtok,
                                                               float* __restrict__ dpos, long rows, int L, int C, int vocab) {
  auto clampid = [&](long id) { return id < 0 ? 0L : (id >= vocab ? (long)vocab - 1 : id); };
  if ((long)blockIdx.x < rows) {
    const long r = blockIdx.x;
    const long id = clampid(word[r]);
    __shared__ int first;
    if (threadIdx.x == 0) {
      int f = 1;
      for (long q = 0; q < r; q++)
        if (clampid(word[q]) == id) { f = 0; break; }
      first = f;
    }
    __syncthreads();
    if (!first) return;
    for (int c = threadIdx.x; c < C; c += NT) {
      float acc = 0.f;
      for (long q = r; q < rows; q++)
        if (clampid(word[q]) == id) acc += Elem<T>::to_f(dout[q * C + c]);
      dtok[id * C + c] += acc;
    }
  } else {
    const long l = (long)blockIdx.x - rows;
    for (int c = threadIdx.x; c < C; c += NT) {
      float acc = 0.f;
      for (long q = l; q < rows; q += L) acc += Elem<T>::to_f(dout[q * C + c]);
      dpos[l * C + c] += acc;
    }
  }
}

// ---- row gather / scatter (EOT token select, clip.py:451-452) ---------------------------------------
template <typename T>
__global__ void __launch_bounds__(NT) gather_rows_kernel(const T* __restrict__ x, long ldx, const int64_t* __restrict__ idx, T* __restrict__ out,
                                                         long ldo, long n, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int cvec = C / VEC;
  GRID_STRIDE(i, n * cvec) {
    const long r = i / cvec;
    const int c = (int)(i % cvec) * VEC;
    stg16(out + r * ldo + c, ldg16(x + idx[r] * ldx + c));
  }
}
template <typename T>
__global__ void __launch_bounds__(NT) scatter_rows_kernel(const T* __restrict__ dout, long lddo, const int64_t* __restrict__ idx, T* __restrict__ dx,
                                                          long lddx, long n, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int cvec = C / VEC;
  GRID_STRIDE(i, n * cvec) {
    const long r = i / cvec;
    const int c = (int)(i % cvec) * VEC;
    stg16(dx + idx[r] * lddx + c, ldg16(dout + r * lddo + c));
  }
}

// ---- z[b,p,:] = x[b,p,:] * s[b,:]   (f5 * state, layers.py:379) -------------------------------------
template <typename T>
__global__ void __launch_bounds__(NT) mul_bcast_fwd_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ s, long lds_, T* __restrict__ z,
                                                           long ldz, int B, int P, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int cvec = C / VEC;
  GRID_STRIDE(i, (long)B * P * cvec) {
    const long r = i / cvec;
    const int c = (int)(i % cvec) * VEC;
    const long b = r / P;
    Vec16<T> a = ldg16(x + r * ldx + c), sv = ldg16(s + b * lds_ + c), o;
#pragma unroll
    for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(Elem<T>::to_f(a.v[e]) * Elem<T>::to_f(sv.v[e]));
    stg16(z + r * ldz + c, o);
  }
}
// dx = dz * s ; ds[b,:] = sum_p dz * x.  Block = (sample, 16 channel vectors) x 16 pixel lanes: the pixel loop is split 16 ways and
// joined through LDS (one thread per (b, vector) walking all P pixels was latency-bound: 102 us for 33 MB at the FPN's 13 x 13 map).
template <typename T>
__global__ void __launch_bounds__(NT) mul_bcast_bwd_kernel(const T* __restrict__ dz, long lddz, const T* __restrict__ x, long ldx,
                                                           const T* __restrict__ s, long lds_, T* __restrict__ dx, long lddx, T* __restrict__ ds,
                                                           long ldds, int B, int P, int C) {
  constexpr int VEC = Elem<T>::VEC, CW = 16, PL = NT / CW;
  __shared__ float red[PL][CW][VEC + 1];
  const int cvec = C / VEC;
  const int cg = (cvec + CW - 1) / CW;
  const int tx = threadIdx.x % CW, tp = threadIdx.x / CW;
  for (int blk = blockIdx.x; blk < B * cg; blk += gridDim.x) {
    const long b = blk / cg;
    const int cv = (blk % cg) * CW + tx;
    const int c = cv * VEC;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) acc[e] = 0.f;
    if (cv < cvec) {
      const Vec16<T> sv = ldg16(s + b * lds_ + c);
      for (int p = tp; p < P; p += PL) {
        const long r = b * P + p;
        Vec16<T> g = ldg16(dz + r * lddz + c), xv = ldg16(x + r * ldx + c), o;
#pragma unroll
        for (int e = 0; e < VEC; e++) {
          const float gf = Elem<T>::to_f(g.v[e]);
          acc[e] += gf * Elem<T>::to_f(xv.v[e]);
          o.v[e] = Elem<T>::from_f(gf * Elem<T>::to_f(sv.v[e]));
        }
        stg16(dx + r * lddx + c, o);
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; e++) red[tp][tx][e] = acc[e];
    __syncthreads();
    if (tp == 0 && cv < cvec) {
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < PL; q++) t += red[q][tx][e];
        o.v[e] = Elem<T>::from_f(t);
      }
      stg16(ds + b * ldds + c, o);
    }
    __syncthreads();
  }
}

// ---- out = a + b[row % brows]  (residual adds, positional broadcasts) ---------------------------------
template <typename T>
__global__ void __launch_bounds__(NT) add_rows_kernel(const T* __restrict__ a, long lda, const T* __restrict__ b, long ldb, long brows,
                                                      T* __restrict__ out, long ldo, long M, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int cvec = C / VEC;
  GRID_STRIDE(i, M * cvec) {
    const long r = i / cvec;
    const int c = (int)(i % cvec) * VEC;
    Vec16<T> x = ldg16(a + r * lda + c), y = ldg16(b + (r % brows) * ldb + c), o;
#pragma unroll
    for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(Elem<T>::to_f(x.v[e]) + Elem<T>::to_f(y.v[e]));
    stg16(out + r * ldo + c, o);
  }
}
// out[r,:] (fp32) = sum_b x[b*R + r,:]
template <typename T>
__global__ void __launch_bounds__(NT) sum_over_batch_kernel(const T* __restrict__ x, long ldx, float* __restrict__ out, long ldo, int B, long R,
                                                            int C, int accumulate) {
  constexpr int VEC = Elem<T>::VEC;
  const int cvec = C / VEC;
  GRID_STRIDE(i, R * cvec) {
    const long r = i / cvec;
    const int c = (int)(i % cvec) * VEC;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) acc[e] = accumulate ? out[r * ldo + c + e] : 0.f;
    for (int b = 0; b < B; b++) {
      Vec16<T> v = ldg16(x + ((long)b * R + r) * ldx + c);
#pragma unroll
      for (int e = 0; e < VEC; e++) acc[e] += Elem<T>::to_f(v.v[e]);
    }
#pragma unroll
    for (int e = 0; e < VEC; e++) out[r * ldo + c + e] = acc[e];
  }
}

// ---- out = a + dropout(b)  /  db = dropout_bwd(dout)   (layers.py:326,334,338) ------------------------
template <typename T>
__global__ void __launch_bounds__(NT) add_dropout_kernel(const T* __restrict__ a, long lda, const T* __restrict__ b, long ldb, T* __restrict__ out,
                                                         long ldo, long M, int C, float p, uint64_t seed, const uint64_t* __restrict__ epoch) {
  constexpr int VEC = Elem<T>::VEC;
  if (epoch) seed += *epoch;                 // crog_set_seed_epoch: per-step offset from device memory
  const int cvec = C / VEC;
  const uint32_t thr = attn_thr16(p);      // 16-bit threshold of the pair hash (dropout_apply, common.h)
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  GRID_STRIDE(i, M * cvec) {
    const long r = i / cvec;
    const int c = (int)(i % cvec) * VEC;
    Vec16<T> y = ldg16(b + r * ldb + c), x, o;
    if (a) x = ldg16(a + r * lda + c);
    float f[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) f[e] = Elem<T>::to_f(y.v[e]);
    if (p > 0.f) {   // launch-uniform flags: tested once per vector
      dropout_apply<VEC>(f, seed, (uint64_t)r * C + c, thr, sc);
    }
    if (a) {
#pragma unroll
      for (int e = 0; e < VEC; e++) f[e] += Elem<T>::to_f(x.v[e]);
    }
#pragma unroll
    for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(f[e]);
    stg16(out + r * ldo + c, o);
  }
}

// ---- activation backward ------------------------------------------------------------------------------
// mode 0: dx = dy * (y > 0)            (ReLU, y = saved output)
// mode 1: dx = dy * d/du quickgelu(u)  (QuickGELU, y = saved pre-activation u; clip.py:234-236)
template <typename T>
__global__ void __launch_bounds__(NT) act_bwd_kernel(const T* __restrict__ dy, long lddy, const T* __restrict__ y, long ldy, T* __restrict__ dx,
                                                     long lddx, long M, int C, int mode) {
  constexpr int VEC = Elem<T>::VEC;
  const int cvec = C / VEC;
  GRID_STRIDE(i, M * cvec) {
    const long r = i / cvec;
    const int c = (int)(i % cvec) * VEC;
    Vec16<T> g = ldg16(dy + r * lddy + c), yv = ldg16(y + r * ldy + c), o;
    float gf[VEC], u[VEC], d[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) { gf[e] = Elem<T>::to_f(g.v[e]); u[e] = Elem<T>::to_f(yv.v[e]); }
    if (mode == 0) {
#pragma unroll
      for (int e = 0; e < VEC; e++) d[e] = u[e] > 0.f ? gf[e] : 0.f;
    } else if (mode == 2) {   // tanh, from the saved OUTPUT (ssg.py:128,133)
#pragma unroll
      for (int e = 0; e < VEC; e++) d[e] = gf[e] * (1.f - u[e] * u[e]);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        const float sg = 1.f / (1.f + expf(-1.702f * u[e]));
        d[e] = gf[e] * sg * (1.f + 1.702f * u[e] * (1.f - sg));
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; e++) o.v[e] = Elem<T>::from_f(d[e]);
    stg16(dx + r * lddx + c, o);
  }
}
template <typename T>
__global__ void __launch_bounds__(NT) quickgelu_fwd_kernel(const T* __restrict__ u, long ldu, T* __restrict__ out, long ldo, long M, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int cvec = C / VEC;
  GRID_STRIDE(i, M * cvec) {
    const long r = i / cvec;
    const int c = (int)(i % cvec) * VEC;
    Vec16<T> v = ldg16(u + r * ldu + c), o;
#pragma unroll
    for (int e = 0; e < VEC; e++) {
      const float f = Elem<T>::to_f(v.v[e]);
      o.v[e] = Elem<T>::from_f(f / (1.f + expf(-1.702f * f)));
    }
    stg16(out + r * ldo + c, o);
  }
}

// ---- stem: 3x3 stride-2 pad-1 patches of an NCHW fp32 image -> [B*OH*OW][32] (27 used; clip.py:165-170) ----
// column k = (ky*3 + kx)*3 + ci matches the channels-last (KRSC) weight row.
template <typename T>
__global__ void __launch_bounds__(NT) stem_im2col_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int H, int W) {
  const int OH = H / 2, OW = W / 2;
  const long total = (long)B * OH * OW;
  GRID_STRIDE(i, total) {
    const int ox = (int)(i % OW);
    const int oy = (int)((i / OW) % OH);
    const long b = i / ((long)OW * OH);
    T row[32];
#pragma unroll
    for (int k = 0; k < 32; k++) row[k] = Elem<T>::from_f(0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W)
#pragma unroll
          for (int ci = 0; ci < 3; ci++)
            row[(ky * 3 + kx) * 3 + ci] = Elem<T>::from_f(img[((b * 3 + ci) * H + iy) * W + ix]);
      }
    T* dst = out + i * 32;
#pragma unroll
    for (int v = 0; v < 32 / Elem<T>::VEC; v++) {
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < Elem<T>::VEC; e++) o.v[e] = row[v * Elem<T>::VEC + e];
      stg16(dst + v * Elem<T>::VEC, o);
    }
  }
}

// ---- ViT patch embedding front end (clip.py:290-296,309-321) ----------------------------------------------------------
// Non-overlapping PxP patches of an NCHW fp32 image -> rows [B*(H/P)*(W/P)][P*P*3], column (ky*P + kx)*3 + ci (the
// channels-last KRSC weight row), so conv1 (kernel = stride = P, no bias) is one crog_gemm.
template <typename T>
__global__ void __launch_bounds__(NT) patchify_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int H, int W, int P) {
  const int GH = H / P, GW = W / P;
  const long total = (long)B * GH * GW * P * P;   // one thread per (patch, ky, kx): 3 channels
  GRID_STRIDE(i, total) {
    const int kx = (int)(i % P);
    const int ky = (int)((i / P) % P);
    const long patch = i / ((long)P * P);
    const int gx = (int)(patch % GW), gy = (int)((patch / GW) % GH);
    const long b = patch / ((long)GW * GH);
    const long src = ((b * 3) * H + (long)gy * P + ky) * W + (long)gx * P + kx;
    T* dst = out + (patch * P * P + (long)ky * P + kx) * 3;
#pragma unroll
    for (int ci = 0; ci < 3; ci++) dst[ci] = Elem<T>::from_f(img[src + (long)ci * H * W]);
  }
}

// tokens[b][0] = class_embedding + pos[0];  tokens[b][1 + i] = y[b*G + i] + pos[1 + i]      (T = G + 1 tokens)
template <typename T>
__global__ void __launch_bounds__(NT) vit_tokens_fwd_kernel(const T* __restrict__ y, long ldy, const T* __restrict__ cls, const T* __restrict__ pos,
                                                            T* __restrict__ out, long rows, int Tn, int C) {
  const int cv = C / Elem<T>::VEC;
  GRID_STRIDE(i, rows * cv) {
    const long r = i / cv;
    const int c = (int)(i % cv) * Elem<T>::VEC;
    const int t = (int)(r % Tn);
    const long b = r / Tn;
    const Vec16<T> a = t == 0 ? ldg16(cls + c) : ldg16(y + (b * (Tn - 1) + t - 1) * ldy + c);
    const Vec16<T> pp = ldg16(pos + (long)t * C + c);
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < Elem<T>::VEC; e++) o.v[e] = Elem<T>::from_f(Elem<T>::to_f(a.v[e]) + Elem<T>::to_f(pp.v[e]));
    stg16(out + r * C + c, o);
  }
}

// dy[b*G + i] = dtok[b][1 + i];  gpos[t] += sum_b dtok[b][t];  gcls += sum_b dtok[b][0]   (fp32 gradient buffers)
template <typename T>
__global__ void __launch_bounds__(NT) vit_tokens_bwd_kernel(const T* __restrict__ dtok, T* __restrict__ dy, long lddy, float* __restrict__ gcls,
                                                            float* __restrict__ gpos, int B, int Tn, int C) {
  GRID_STRIDE(i, (long)Tn * C) {
    const int t = (int)(i / C), c = (int)(i % C);
    float acc = 0.f;
    for (int b = 0; b < B; b++) {
      const T v = dtok[((long)b * Tn + t) * C + c];
      acc += Elem<T>::to_f(v);
      if (t > 0) dy[((long)b * (Tn - 1) + t - 1) * lddy + c] = v;
    }
    gpos[i] += acc;
    if (t == 0) gcls[c] += acc;
  }
}

// ---- data-gradient layout of the 3x3 convolution weights: dst[ci][8 - tap][co] = src[co][tap][ci] for every listed conv ----
// With this copy the data gradient of a 3x3 convolution is the same K-contiguous implicit GEMM as its forward (both operands
// read with ds_read_b128) instead of reading the forward layout transposed out of LDS: measured 707 -> 811 TFLOP/s at 512 ch.
// table[i] = (element offset in the flat buffers, Cout, Cin); one launch per step covers all of a model's 3x3 convs.
// Per (convolution, tap) this is a [Cout][Cin] -> [Cin][Cout] transpose: 64 x 64 tiles go through LDS so that both the reads (runs
// along Cin) and the writes (runs along Cout) are 16-byte vectors (the element-wise form read one 64-byte sector per 2 bytes used:
// 465 us and 1 GB of traffic per step for 80 MB of weights).  Layers whose channel counts are not multiples of the vector (the
// 3-channel stem) take the element-wise loop.
// TSTRIDE: longs per table entry: 3 = (offset, Cout, Cin) of a 3x3 convolution; 4 = (offset, rows, cols, taps) with taps 9 or 1 - a
// plain [rows][cols] -> [cols][rows] transpose for the 1x1 / linear weights, whose data gradients then read a K-contiguous operand too
// (crog_dgrad_weights).
template <typename T, int TSTRIDE>
__global__ void __launch_bounds__(NT) conv3_dgrad_weights_kernel(const T* __restrict__ src, T* __restrict__ dst, const long* __restrict__ table) {
  constexpr int VEC = Elem<T>::VEC, TS = 64, VPR = TS / VEC;
  __shared__ __attribute__((aligned(16))) T tile[TS][TS + VEC];
  const long off = table[TSTRIDE * blockIdx.y], co_n = table[TSTRIDE * blockIdx.y + 1], ci_n = table[TSTRIDE * blockIdx.y + 2];
  const int TAPS = TSTRIDE == 4 ? (int)table[TSTRIDE * blockIdx.y + 3] : 9;
  if (co_n % VEC != 0 || ci_n % VEC != 0) {
    const long total = co_n * TAPS * ci_n;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
      const long co = i % co_n;              // destination index i = (ci * TAPS + t) * Cout + co
      const long t = (i / co_n) % TAPS;
      const long ci = i / (co_n * TAPS);
      dst[off + i] = src[off + (co * TAPS + (TAPS - 1 - t)) * ci_n + ci];
    }
    return;
  }
  const int tco = (int)((co_n + TS - 1) / TS), tci = (int)((ci_n + TS - 1) / TS);
  const int ntiles = TAPS * tco * tci;
  for (int id = blockIdx.x; id < ntiles; id += gridDim.x) {
    const int t = id % TAPS, rest = id / TAPS;
    const long ci0 = (long)(rest % tci) * TS, co0 = (long)(rest / tci) * TS;
    for (int v = threadIdx.x; v < TS * VPR; v += NT) {
      const int r = v / VPR, c = (v % VPR) * VEC;
      Vec16<T> x;
#pragma unroll
      for (int e = 0; e < VEC; e++) x.v[e] = Elem<T>::from_f(0.f);
      if (co0 + r < co_n && ci0 + c < ci_n) x = ldg16(src + off + ((co0 + r) * TAPS + t) * ci_n + ci0 + c);
      *reinterpret_cast<Vec16<T>*>(&tile[r][c]) = x;
    }
    __syncthreads();
    for (int v = threadIdx.x; v < TS * VPR; v += NT) {
      const int r = v / VPR, c = (v % VPR) * VEC;      // r: input channel inside the tile, c: first of VEC output channels
      if (ci0 + r < ci_n && co0 + c < co_n) {
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < VEC; e++) o.v[e] = tile[c + e][r];
        stg16(dst + off + ((ci0 + r) * TAPS + (TAPS - 1 - t)) * co_n + co0 + c, o);
      }
    }
    __syncthreads();
  }
}

// ---- dst[r][c] = c < cols_src ? src[r][c] : 0, c < cols_dst   (fp32 source; T or fp32 destination) ------
template <typename TD>
__global__ void __launch_bounds__(NT) cast_pad2d_kernel(const float* __restrict__ src, long lds_, int cols_src, TD* __restrict__ dst, long ldd,
                                                        int cols_dst, long rows) {
  GRID_STRIDE(i, rows * cols_dst) {
    const long r = i / cols_dst;
    const int c = (int)(i % cols_dst);
    const float v = c < cols_src ? src[r * lds_ + c] : 0.f;
    dst[r * ldd + c] = Elem<TD>::from_f(v);
  }
}

// ---- dst[r][c] += src[r][c], c < cols: strips the zero padding of a ragged-Cin weight gradient back into the flat buffer ------
__global__ void __launch_bounds__(NT) add_pad2d_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst, long ldd, int cols,
                                                       long rows) {
  GRID_STRIDE(i, rows * cols) {
    const long r = i / cols;
    const int c = (int)(i % cols);
    dst[r * ldd + c] += src[r * lds_ + c];
  }
}

// ---- flat casts ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT) cast_f32_to_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long n) {
  const long nv = n / 8;
  GRID_STRIDE(i, nv) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + i * 8), b = *reinterpret_cast<const f32x4*>(src + i * 8 + 4);
    Vec16<bf16> o;
    o.v[0] = (bf16)a[0]; o.v[1] = (bf16)a[1]; o.v[2] = (bf16)a[2]; o.v[3] = (bf16)a[3];
    o.v[4] = (bf16)b[0]; o.v[5] = (bf16)b[1]; o.v[6] = (bf16)b[2]; o.v[7] = (bf16)b[3];
    stg16(dst + i * 8, o);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[nv * 8 + threadIdx.x] = (bf16)src[nv * 8 + threadIdx.x];
}
template <typename T>
__global__ void __launch_bounds__(NT) cast_to_f32_kernel(const T* __restrict__ src, long lds_, float* __restrict__ dst, long ldd, long M, int C) {
  GRID_STRIDE(i, M * C) {
    const long r = i / C;
    const int c = (int)(i % C);
    dst[r * ldd + c] = Elem<T>::to_f(src[r * lds_ + c]);
  }
}

// ---- CoordConv coordinate channels (layers.py:30-39): buf[b,y,x,c0] = x in [-1,1], [c0+1] = y, rest 0 ------
template <typename T>
__global__ void __launch_bounds__(NT) coord_fill_kernel(T* __restrict__ buf, long ld, int B, int H, int W, int c0, int cend) {
  GRID_STRIDE(i, (long)B * H * W) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const float fx = W > 1 ? -1.f + 2.f * x / (W - 1) : -1.f;
    const float fy = H > 1 ? -1.f + 2.f * y / (H - 1) : -1.f;
    T* p = buf + i * ld;
    p[c0] = Elem<T>::from_f(fx);
    p[c0 + 1] = Elem<T>::from_f(fy);
    for (int c = c0 + 2; c < cend; c++) p[c] = Elem<T>::from_f(0.f);
  }
}

// ---- fused Adam (torch.optim.Adam semantics, train_crog.py:119-121) + optional bf16 shadow copy ---------------
__global__ void __launch_bounds__(NT) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                  long n, float lr, float beta1, float beta2, float eps, float weight_decay, float bc1,
                                                  float bc2_sqrt, bf16* __restrict__ shadow, const float* __restrict__ hyper) {
  if (hyper) {           // crog_adam_step_dev: learning rate and bias corrections of THIS step from device memory (graph replays)
    lr = hyper[0];
    bc1 = hyper[1];
    bc2_sqrt = hyper[2];
  }
  const float step_size = lr / bc1;
  // One 16-byte vector of each stream per thread, no grid-stride loop, non-temporal accesses for everything that is not read again
  // before the next optimizer step (g, m, v, the fp32 parameters): the launch shape of the copy probe (api.hip), which streams at
  // 6.5 TB/s where a 4096-block grid-stride loop reached 4.3-4.6.  The bf16 shadow is stored normally: the next forward reads it.
  {
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i >= (n + 3) / 4) return;
    const long o = i * 4;
    if (o + 4 <= n) {
      f32x4 pv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + o)), gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + o));
      f32x4 mv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + o)), vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + o));
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float gg = gv[e] + weight_decay * pv[e];
        mv[e] = beta1 * mv[e] + (1.f - beta1) * gg;
        vv[e] = beta2 * vv[e] + (1.f - beta2) * gg * gg;
        const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
        pv[e] -= step_size * mv[e] / denom;
      }
      __builtin_nontemporal_store(pv, reinterpret_cast<f32x4*>(p + o));
      __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(m + o));
      __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v + o));
      if (shadow) {
        bf16x4 s;
#pragma unroll
        for (int e = 0; e < 4; e++) s[e] = (bf16)pv[e];
        *reinterpret_cast<bf16x4*>(shadow + o) = s;
      }
    } else {
      for (long j = o; j < n; j++) {
        float gg = g[j] + weight_decay * p[j];
        m[j] = beta1 * m[j] + (1.f - beta1) * gg;
        v[j] = beta2 * v[j] + (1.f - beta2) * gg * gg;
        p[j] -= step_size * m[j] / (sqrtf(v[j]) / bc2_sqrt + eps);
        if (shadow) shadow[j] = (bf16)p[j];
      }
    }
  }
}

// ---- out[c] += sum_r x[r][c]  (bias gradients), two passes without contended atomics -----------------------------
// pass 1: block (4 row lanes x 64 chunk lanes) reduces `rows_per_block` rows of a 64-chunk column group -> partial[rb][c]
template <typename T>
__global__ void __launch_bounds__(NT) colsum_partial_kernel(const T* __restrict__ x, long ldx, long M, int C, int rows_per_block,
                                                           float* __restrict__ partial) {
  constexpr int VEC = Elem<T>::VEC;
  __shared__ float red[4][64][VEC + 1];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = (blockIdx.y * 64 + tx) * VEC;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, M);
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; e++) acc[e] = 0.f;
  if (c + VEC <= C) {
    for (long r = r0 + ty; r < r1; r += 4) {
      Vec16<T> v = ldg16(x + r * ldx + c);
#pragma unroll
      for (int e = 0; e < VEC; e++) acc[e] += Elem<T>::to_f(v.v[e]);
    }
  } else if (c < C) {
    for (long r = r0 + ty; r < r1; r += 4)
      for (int e = 0; e < VEC && c + e < C; e++) acc[e] += Elem<T>::to_f(x[r * ldx + c + e]);
  }
#pragma unroll
  for (int e = 0; e < VEC; e++) red[ty][tx][e] = acc[e];
  __syncthreads();
  if (ty == 0 && c < C) {
#pragma unroll
    for (int e = 0; e < VEC; e++)
      if (c + e < C) partial[(long)blockIdx.x * C + c + e] = red[0][tx][e] + red[1][tx][e] + red[2][tx][e] + red[3][tx][e];
  }
}
// pass 2: out[c] += sum_rb partial[rb][c]   (block = 64 channels x 4 row-block lanes)
__global__ void __launch_bounds__(NT) colsum_final_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float a = 0.f;
  if (c < C)
    for (int b = ty; b < nblk; b += 4) a += partial[(long)b * C + c];
  red[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && c < C) out[c] += red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}

}  // namespace

#define DISPATCH_T(dtype, ...)                                   \
  do {                                                           \
    if ((dtype) == CROG_BF16) { using T = bf16; __VA_ARGS__; }   \
    else if ((dtype) == CROG_F32) { using T = float; __VA_ARGS__; } \
    else { crog_set_error("bad dtype %d", (int)(dtype)); return CROG_ERR_ARG; } \
  } while (0)
#define VECOF(dtype) ((dtype) == CROG_BF16 ? 8 : 4)
#define LAUNCH(kern, work, stream, ...)                                                            \
  hipLaunchKernelGGL(kern, dim3(stream_grid(work)), dim3(NT), 0, (hipStream_t)(stream), __VA_ARGS__)

extern "C" int crog_avgpool2_fwd(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int H, int W, int C, crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && C % vec == 0 && ldx % vec == 0 && ldy % vec == 0, "avgpool2: H,W even and C %% %d == 0 required", vec);
  DISPATCH_T(dtype, LAUNCH((avgpool2_fwd_kernel<T>), (long)B * (H / 2) * (W / 2) * (C / vec), s, (const T*)x, (long)ldx, (T*)y, (long)ldy, B, H, W, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_avgpool2_bwd_add(int dtype, const void* dy, int64_t lddy, const void* add, int64_t ldadd, void* dx, int64_t lddx, int B, int H,
                                    int W, int C, crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && C % vec == 0 && lddy % vec == 0 && lddx % vec == 0 && (add == nullptr || ldadd % vec == 0),
                 "avgpool2_bwd: bad shape");
  DISPATCH_T(dtype, LAUNCH((avgpool2_bwd_kernel<T>), (long)B * (H / 2) * (W / 2) * (C / vec), s, (const T*)dy, (long)lddy, (const T*)add,
                           (long)ldadd, (T*)dx, (long)lddx, B, H, W, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_avgpool2_bwd(int dtype, const void* dy, int64_t lddy, void* dx, int64_t lddx, int B, int H, int W, int C, crog_stream_t s) {
  return crog_avgpool2_bwd_add(dtype, dy, lddy, nullptr, 0, dx, lddx, B, H, W, C, s);
}
extern "C" int crog_upsample2_fwd(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int H, int W, int C, crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && ldx % vec == 0 && ldy % vec == 0, "upsample2: C %% %d == 0 required", vec);
  static const bool per_output = [] { const char* e = getenv("CROG_UPSAMPLE_OLD"); return e && atoi(e) != 0; }();
  const int cvec = C / vec, cshift = (cvec & (cvec - 1)) == 0 ? __builtin_ctz(cvec) : -1;
  const int fchunks = cdiv((long)(W + 1) * cvec, NT);
  if (per_output || (long)B * (H + 1) * fchunks >= (1L << 31)) {
    DISPATCH_T(dtype, LAUNCH((upsample2_fwd_kernel<T>), (long)B * 4 * H * W * (C / vec), s, (const T*)x, (long)ldx, (T*)y, (long)ldy, B, H, W, C));
  } else {
    const dim3 grid((unsigned)((long)B * (H + 1) * fchunks));
    DISPATCH_T(dtype, hipLaunchKernelGGL((upsample2_fwd_patch_kernel<T>), grid, dim3(NT), 0, (hipStream_t)s, (const T*)x, (long)ldx, (T*)y, (long)ldy, H, W, C,
                                         cshift, fchunks));
  }
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_upsample2_bwd(int dtype, const void* dy, int64_t lddy, void* dx, int64_t lddx, int B, int H, int W, int C, crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && lddy % vec == 0 && lddx % vec == 0, "upsample2_bwd: C %% %d == 0 required", vec);
  static const bool per_output = [] { const char* e = getenv("CROG_UPSAMPLE_OLD"); return e && atoi(e) != 0; }();
  const int cvec = C / vec, cshift = (cvec & (cvec - 1)) == 0 ? __builtin_ctz(cvec) : -1;
  const int bchunks = cdiv((long)((W + 1) / 2) * cvec, NT);
  if (per_output || (long)B * ((H + 1) / 2) * bchunks >= (1L << 31)) {
    DISPATCH_T(dtype, LAUNCH((upsample2_bwd_kernel<T>), (long)B * H * W * (C / vec), s, (const T*)dy, (long)lddy, (T*)dx, (long)lddx, B, H, W, C));
  } else {
    const dim3 grid((unsigned)((long)B * ((H + 1) / 2) * bchunks));
    DISPATCH_T(dtype, hipLaunchKernelGGL((upsample2_bwd_quad_kernel<T>), grid, dim3(NT), 0, (hipStream_t)s, (const T*)dy, (long)lddy, (T*)dx, (long)lddx, H, W, C,
                                         cshift, bchunks));
  }
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_embedding_fwd(int dtype, const int64_t* word, const void* tok, const void* pos, void* out, int64_t rows, int L, int C, int vocab,
                                  crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && L > 0, "embedding: C %% %d == 0 required", vec);
  DISPATCH_T(dtype, LAUNCH((embedding_fwd_kernel<T>), rows * (C / vec), s, word, (const T*)tok, (const T*)pos, (T*)out, (long)rows, L, C, vocab));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_embedding_bwd(int dtype, const int64_t* word, const void* dout, float* dtok, float* dpos, int64_t rows, int L, int C, int vocab,
                                  crog_stream_t s) {
  if (crog_deterministic()) {
    CROG_CHECK_ARG(rows % L == 0 && rows + L < (1L << 30), "embedding_bwd (deterministic): rows must be whole sequences");
    DISPATCH_T(dtype, hipLaunchKernelGGL((embedding_bwd_det_kernel<T>), dim3((unsigned)(rows + L)), dim3(NT), 0, (hipStream_t)s, word, (const T*)dout, dtok,
                                         dpos, (long)rows, L, C, vocab));
    CROG_LAUNCH_CHECK();
    return CROG_OK;
  }
  DISPATCH_T(dtype, LAUNCH((embedding_bwd_kernel<T>), rows * C, s, word, (const T*)dout, dtok, dpos, (long)rows, L, C, vocab));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_gather_rows(int dtype, const void* x, int64_t ldx, const int64_t* idx, void* out, int64_t ldo, int64_t n, int C, crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && ldx % vec == 0 && ldo % vec == 0, "gather_rows: C %% %d == 0 required", vec);
  DISPATCH_T(dtype, LAUNCH((gather_rows_kernel<T>), n * (C / vec), s, (const T*)x, (long)ldx, idx, (T*)out, (long)ldo, (long)n, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_scatter_rows(int dtype, const void* dout, int64_t lddo, const int64_t* idx, void* dx, int64_t lddx, int64_t n, int C, crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && lddo % vec == 0 && lddx % vec == 0, "scatter_rows: C %% %d == 0 required", vec);
  DISPATCH_T(dtype, LAUNCH((scatter_rows_kernel<T>), n * (C / vec), s, (const T*)dout, (long)lddo, idx, (T*)dx, (long)lddx, (long)n, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_mul_bcast_fwd(int dtype, const void* x, int64_t ldx, const void* sv, int64_t lds_, void* z, int64_t ldz, int B, int P, int C,
                                  crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && ldx % vec == 0 && lds_ % vec == 0 && ldz % vec == 0, "mul_bcast: C %% %d == 0 required", vec);
  DISPATCH_T(dtype, LAUNCH((mul_bcast_fwd_kernel<T>), (long)B * P * (C / vec), s, (const T*)x, (long)ldx, (const T*)sv, (long)lds_, (T*)z, (long)ldz, B, P, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_mul_bcast_bwd(int dtype, const void* dz, int64_t lddz, const void* x, int64_t ldx, const void* sv, int64_t lds_, void* dx,
                                  int64_t lddx, void* ds, int64_t ldds, int B, int P, int C, crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0, "mul_bcast_bwd: C %% %d == 0 required", vec);
  const int blocks = std::min(B * cdiv(C / vec, 16), 4096);
  DISPATCH_T(dtype, hipLaunchKernelGGL((mul_bcast_bwd_kernel<T>), dim3(blocks), dim3(NT), 0, (hipStream_t)s, (const T*)dz, (long)lddz, (const T*)x,
                                       (long)ldx, (const T*)sv, (long)lds_, (T*)dx, (long)lddx, (T*)ds, (long)ldds, B, P, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_add_rows(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, int64_t brows, void* out, int64_t ldo, int64_t M,
                             int C, crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && lda % vec == 0 && ldb % vec == 0 && ldo % vec == 0 && brows > 0, "add_rows: C %% %d == 0 required", vec);
  DISPATCH_T(dtype, LAUNCH((add_rows_kernel<T>), M * (C / vec), s, (const T*)a, (long)lda, (const T*)b, (long)ldb, (long)brows, (T*)out, (long)ldo, (long)M, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_sum_over_batch(int dtype, const void* x, int64_t ldx, float* out, int64_t ldo, int B, int64_t R, int C, int accumulate,
                                   crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && ldx % vec == 0, "sum_over_batch: C %% %d == 0 required", vec);
  DISPATCH_T(dtype, LAUNCH((sum_over_batch_kernel<T>), R * (C / vec), s, (const T*)x, (long)ldx, out, (long)ldo, B, (long)R, C, accumulate));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_add_dropout(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t M, int C, float p,
                                uint64_t seed, crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && ldb % vec == 0 && ldo % vec == 0 && p >= 0.f && p < 1.f, "add_dropout: bad args");
  DISPATCH_T(dtype, LAUNCH((add_dropout_kernel<T>), M * (C / vec), s, (const T*)a, (long)lda, (const T*)b, (long)ldb, (T*)out, (long)ldo, (long)M, C, p, seed, crog_seed_epoch()));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_act_bwd(int dtype, const void* dy, int64_t lddy, const void* y, int64_t ldy, void* dx, int64_t lddx, int64_t M, int C, int mode,
                            crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && lddy % vec == 0 && ldy % vec == 0 && lddx % vec == 0, "act_bwd: C %% %d == 0 required", vec);
  DISPATCH_T(dtype, LAUNCH((act_bwd_kernel<T>), M * (C / vec), s, (const T*)dy, (long)lddy, (const T*)y, (long)ldy, (T*)dx, (long)lddx, (long)M, C, mode));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_quickgelu_fwd(int dtype, const void* u, int64_t ldu, void* out, int64_t ldo, int64_t M, int C, crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && ldu % vec == 0 && ldo % vec == 0, "quickgelu: C %% %d == 0 required", vec);
  DISPATCH_T(dtype, LAUNCH((quickgelu_fwd_kernel<T>), M * (C / vec), s, (const T*)u, (long)ldu, (T*)out, (long)ldo, (long)M, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_stem_im2col(int dtype, const float* img, void* out, int B, int H, int W, crog_stream_t s) {
  CROG_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "stem_im2col: H, W must be even");
  DISPATCH_T(dtype, LAUNCH((stem_im2col_kernel<T>), (long)B * (H / 2) * (W / 2), s, img, (T*)out, B, H, W));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_patchify(int dtype, const float* img, void* out, int B, int H, int W, int P, crog_stream_t s) {
  CROG_CHECK_ARG(P > 0 && H % P == 0 && W % P == 0, "patchify: H, W must be multiples of the patch size");
  DISPATCH_T(dtype, LAUNCH((patchify_kernel<T>), (long)B * H * W, s, img, (T*)out, B, H, W, P));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_vit_tokens_fwd(int dtype, const void* y, int64_t ldy, const void* cls, const void* pos, void* out, int B, int T_, int C,
                                   crog_stream_t s) {
  CROG_CHECK_ARG(T_ >= 2, "vit_tokens: needs at least one patch token");
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(C % vec == 0 && ldy % vec == 0, "vit_tokens: C and ldy must be multiples of %d", vec);
  CROG_CHECK_ARG(((uintptr_t)y % 16) == 0 && ((uintptr_t)cls % 16) == 0 && ((uintptr_t)pos % 16) == 0 && ((uintptr_t)out % 16) == 0,
                 "vit_tokens: pointers must be 16-byte aligned");
  DISPATCH_T(dtype, LAUNCH((vit_tokens_fwd_kernel<T>), (long)B * T_ * (C / vec), s, (const T*)y, (long)ldy, (const T*)cls, (const T*)pos, (T*)out,
                           (long)B * T_, T_, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_vit_tokens_bwd(int dtype, const void* dtok, void* dy, int64_t lddy, float* gcls, float* gpos, int B, int T_, int C,
                                   crog_stream_t s) {
  CROG_CHECK_ARG(T_ >= 2, "vit_tokens: needs at least one patch token");
  DISPATCH_T(dtype, LAUNCH((vit_tokens_bwd_kernel<T>), (long)T_ * C, s, (const T*)dtok, (T*)dy, (long)lddy, gcls, gpos, B, T_, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_conv3_dgrad_weights(int dtype, const void* src, void* dst, const int64_t* table, int count, crog_stream_t s) {
  if (count <= 0) return CROG_OK;
  CROG_CHECK_ARG(src && dst && table, "conv3_dgrad_weights: null pointer");
  static_assert(sizeof(long) == sizeof(int64_t), "table entries are 64-bit");
  DISPATCH_T(dtype, hipLaunchKernelGGL((conv3_dgrad_weights_kernel<T, 3>), dim3(96, count), dim3(NT), 0, (hipStream_t)s, (const T*)src, (T*)dst,
                                       reinterpret_cast<const long*>(table)));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_dgrad_weights(int dtype, const void* src, void* dst, const int64_t* table, int count, crog_stream_t s) {
  if (count <= 0) return CROG_OK;
  CROG_CHECK_ARG(src && dst && table && count <= 65535, "dgrad_weights: null pointer or more than 65535 entries");
  static const int gx = [] { const char* e = getenv("CROG_DGW_GRID"); return e ? atoi(e) : 384; }();
  // (blocks per table entry.  The matrices range from one 64 x 64 tile to 1152: with 24 blocks per entry the largest took 48 tiles per
  // block and the launch 287 us for 450 MB of traffic; 96: 170, 192: 155, 384: 126, 768: 123 us - most blocks of a small entry exit at once)
  DISPATCH_T(dtype, hipLaunchKernelGGL((conv3_dgrad_weights_kernel<T, 4>), dim3(gx, count), dim3(NT), 0, (hipStream_t)s, (const T*)src, (T*)dst,
                                       reinterpret_cast<const long*>(table)));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_add_pad2d(const float* src, int64_t lds_, float* dst, int64_t ldd, int cols, int64_t rows, crog_stream_t s) {
  CROG_CHECK_ARG(src && dst && cols > 0 && rows >= 0, "add_pad2d: bad arguments");
  if (rows == 0) return CROG_OK;
  LAUNCH(add_pad2d_kernel, rows * cols, s, src, (long)lds_, dst, (long)ldd, cols, (long)rows);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_cast_pad2d(int dtype_dst, const float* src, int64_t lds_, int cols_src, void* dst, int64_t ldd, int cols_dst, int64_t rows,
                               crog_stream_t s) {
  DISPATCH_T(dtype_dst, LAUNCH((cast_pad2d_kernel<T>), rows * cols_dst, s, src, (long)lds_, cols_src, (T*)dst, (long)ldd, cols_dst, (long)rows));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
// optimizer.zero_grad() on the flat gradient buffer (crog_engine.py:77) as a launch of this library: it can go to any stream the caller names
// (torch's fill runs on torch's current stream), e.g. the weight-gradient stream while the forward pass keeps the main one busy
__global__ void __launch_bounds__(NT) zero_f32_kernel(float* __restrict__ p, long n4, long n) {
  GRID_STRIDE(i, n4) { reinterpret_cast<f32x4*>(p)[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[4 * n4 + threadIdx.x] = 0.f;
}
extern "C" int crog_zero_f32(float* p, int64_t n, crog_stream_t s) {
  CROG_CHECK_ARG(p && n >= 0 && ((uintptr_t)p % 16) == 0, "zero_f32: 16-byte aligned buffer");
  if (n == 0) return CROG_OK;
  LAUNCH(zero_f32_kernel, n / 4 + 1, s, p, (long)(n / 4), (long)n);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_cast_f32_to_bf16(const float* src, void* dst, int64_t n, crog_stream_t s) {
  CROG_CHECK_ARG(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "cast: pointers must be 16-byte aligned");
  LAUNCH(cast_f32_to_bf16_kernel, n / 8 + 1, s, src, (bf16*)dst, (long)n);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_cast_to_f32(int dtype, const void* src, int64_t lds_, float* dst, int64_t ldd, int64_t M, int C, crog_stream_t s) {
  DISPATCH_T(dtype, LAUNCH((cast_to_f32_kernel<T>), M * C, s, (const T*)src, (long)lds_, dst, (long)ldd, (long)M, C));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_coord_fill(int dtype, void* buf, int64_t ld, int B, int H, int W, int c0, int cend, crog_stream_t s) {
  CROG_CHECK_ARG(c0 + 2 <= cend && cend <= ld, "coord_fill: bad channel range");
  DISPATCH_T(dtype, LAUNCH((coord_fill_kernel<T>), (long)B * H * W, s, (T*)buf, (long)ld, B, H, W, c0, cend));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, void* bf16_shadow, crog_stream_t s) {
  CROG_CHECK_ARG(step >= 1 && n >= 0, "adam: step must be >= 1");
  CROG_CHECK_ARG(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
                 "adam: buffers must be 16-byte aligned");
  if (n == 0) return CROG_OK;
  // bias corrections in double, rounded once: the device-side form (adam_advance_kernel) computes the same expression
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3(cdiv((n + 3) / 4, NT)), dim3(NT), 0, (hipStream_t)s, p, g, m, v, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s,
                     (bf16*)bf16_shadow, (const float*)nullptr);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

namespace {
// hyper = {lr, 1 - beta1^t, sqrt(1 - beta2^t), t}: t += 1 and the corrections of the new t
__global__ void adam_advance_kernel(float* __restrict__ hyper, float beta1, float beta2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double t = (double)hyper[3] + 1.0;
  hyper[3] = (float)t;
  hyper[1] = (float)(1.0 - pow((double)beta1, t));
  hyper[2] = (float)sqrt(1.0 - pow((double)beta2, t));
}
}  // namespace

extern "C" int crog_adam_advance(float* hyper_dev, float beta1, float beta2, crog_stream_t s) {
  CROG_CHECK_ARG(hyper_dev != nullptr && ((uintptr_t)hyper_dev % 16) == 0, "adam_advance: hyper must be a 16-byte aligned device float[4]");
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, hyper_dev, beta1, beta2);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

extern "C" int crog_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev, float beta1, float beta2,
                                  float eps, float weight_decay, void* bf16_shadow, crog_stream_t s) {
  CROG_CHECK_ARG(hyper_dev != nullptr && n >= 0, "adam_step_dev: hyper (device float[4]: lr, bc1, bc2_sqrt, t) is required");
  CROG_CHECK_ARG(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
                 "adam: buffers must be 16-byte aligned");
  if (n == 0) return CROG_OK;
  hipLaunchKernelGGL(adam_kernel, dim3(cdiv((n + 3) / 4, NT)), dim3(NT), 0, (hipStream_t)s, p, g, m, v, (long)n, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f,
                     (bf16*)bf16_shadow, hyper_dev);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

extern "C" int crog_colsum(int dtype, const void* x, int64_t ldx, int64_t M, int C, int rows_per_block, float* partial, float* out,
                           crog_stream_t s) {
  const int vec = VECOF(dtype);
  CROG_CHECK_ARG(ldx % vec == 0 && rows_per_block > 0 && partial != nullptr, "colsum: ldx %% %d == 0 and a partial workspace are required", vec);
  const int nblk = cdiv(M, rows_per_block);
  dim3 grid(nblk, cdiv(cdiv(C, vec), 64));
  DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_partial_kernel<T>), grid, dim3(NT), 0, (hipStream_t)s, (const T*)x, (long)ldx, (long)M, C, rows_per_block, partial));
  CROG_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(C, 64)), dim3(NT), 0, (hipStream_t)s, partial, nblk, C, out);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
