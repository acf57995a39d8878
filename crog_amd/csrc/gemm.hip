// MFMA GEMM / implicit-GEMM conv family for gfx950 (see include/crog_hip.h, crog_gemm).
//
// One 256-thread workgroup (4 waves, 2x2) computes a 128x128 tile of C; each wave owns a
// 64x64 sub-tile as 2x2 MFMA 32x32 accumulators (64 fp32 registers).  Operands are staged
// global -> registers -> LDS with a two-deep LDS ring (loads for k-tile t+1 are issued before
// the MFMAs of tile t and written to LDS after them), one barrier per k-tile.
//
//   bf16: v_mfma_f32_32x32x16_bf16, BK = 32        f32: v_mfma_f32_32x32x2_f32 (exact f32), BK = 16
//
// Operand layouts (crog_a_layout / crog_b_layout):
//   K-contiguous operands land in LDS as [128][BK+pad] and fragments are single ds_read_b128.
//   Transposed operands (reduction index is the slow memory index: dgrad weights, wgrad, P.V)
//   are copied row-major into LDS as [BK][128+pad] with full 16-byte coalesced loads and are
//   transposed on the READ side: ds_read_b64_tr_b16 for bf16, ds_read_b32 for f32.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int OP_BYTES = 10240;  // LDS bytes per operand per ring slot
constexpr int NTHREADS = 256;

template <typename T> struct TileCfg;
template <> struct TileCfg<bf16> {
  static constexpr int VEC = 8, BK = 32;
  static constexpr int KC_ROW = 40;   // elements; 80-byte rows  -> conflict-free ds_read_b128
  static constexpr int TR_ROW = 160;  // elements; 320-byte rows -> conflict-free ds_read_b64_tr_b16
};
template <> struct TileCfg<float> {
  static constexpr int VEC = 4, BK = 16;
  static constexpr int KC_ROW = 20;   // 80-byte rows
  static constexpr int TR_ROW = 132;  // 528-byte rows
};

template <typename T> struct Frag;
template <> struct Frag<bf16> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

// One K=16 step of a 32x32 tile.  Lane (r = lane&31, h = lane>>5) supplies A[r][8h+j] and
// B[8h+j][r], j = 0..7 (for f32 instruction j contracts k in {j, 8+j}: same sum, same lanes).
__device__ inline void mma16(const Frag<bf16>& a, const Frag<bf16>& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
}
__device__ inline void mma16(const Frag<float>& a, const Frag<float>& b, f32x16& c) {
#pragma unroll
  for (int j = 0; j < 8; j++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], c, 0, 0, 0);
}

// ---- fragment reads -------------------------------------------------------------------------
// K-contiguous LDS tile [rows][KC_ROW]
__device__ inline Frag<bf16> frag_kc(const bf16* tile, int row, int ks, int h) {
  Frag<bf16> f;
  f.v = *reinterpret_cast<const bf16x8*>(tile + row * TileCfg<bf16>::KC_ROW + ks * 16 + h * 8);
  return f;
}
__device__ inline Frag<float> frag_kc(const float* tile, int row, int ks, int h) {
  Frag<float> f;
  const f32x4* p = reinterpret_cast<const f32x4*>(tile + row * TileCfg<float>::KC_ROW + ks * 16 + h * 8);
  f32x4 a = p[0], b = p[1];
  f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
  f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
  return f;
}
// Row-major-copy LDS tile [BK][TR_ROW]; wave's 32 columns start at `cbase`.
template <bool HWTR>
__device__ inline Frag<bf16> frag_tr(const bf16* tile, int cbase, int ks, int lane) {
  constexpr int ROW = TileCfg<bf16>::TR_ROW;
  const int h = lane >> 5;
  Frag<bf16> f;
  if constexpr (HWTR) {
    // ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-col block is delivered column-major;
    // lane 4q+p of the group supplies the address of row q, columns 4p..4p+3.
    const int i = lane & 15, q = i >> 2, pp = i & 3;
    const int col = cbase + 16 * ((lane >> 4) & 1) + 4 * pp;
    const int row0 = ks * 16 + 8 * h + q;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + row0 * ROW + col));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + (row0 + 4) * ROW + col));
    f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  } else {
    const int col = cbase + (lane & 31);
#pragma unroll
    for (int j = 0; j < 8; j++) f.v[j] = tile[(ks * 16 + 8 * h + j) * ROW + col];
  }
  return f;
}
template <bool HWTR>
__device__ inline Frag<float> frag_tr(const float* tile, int cbase, int ks, int lane) {
  constexpr int ROW = TileCfg<float>::TR_ROW;
  const int h = lane >> 5, col = cbase + (lane & 31);
  Frag<float> f;
#pragma unroll
  for (int j = 0; j < 8; j++) f.v[j] = tile[(ks * 16 + 8 * h + j) * ROW + col];
  return f;
}

struct ConvGeom { int H, W, C; };

// ---- K-contiguous loader: 128 rows x BK, thread -> rows {tid/4, tid/4+64}, 16-byte chunk tid%4 ----
template <typename T, bool IM2COL>
struct KcLoader {
  static constexpr int VEC = TileCfg<T>::VEC, BK = TileCfg<T>::BK, ROW = TileCfg<T>::KC_ROW;
  const T* base;
  int64_t ld;
  int K;
  int kv;            // element offset of this thread's chunk inside the k-tile
  int lrow[2];       // tile-local rows
  long grow[2];      // global row (pixel) index, -1 when out of range
  int py[2], px[2];  // im2col: pixel coordinates
  ConvGeom g;
  Vec16<T> reg[2];

  __device__ void init(const T* b, int64_t ld_, int rows_total, int K_, int row0, ConvGeom g_, int tid) {
    base = b; ld = ld_; K = K_; g = g_;
    kv = (tid & 3) * VEC;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      lrow[i] = (tid >> 2) + i * 64;
      long m = (long)row0 + lrow[i];
      grow[i] = (m < rows_total) ? m : -1;
      if constexpr (IM2COL) {
        long rem = m % ((long)g.H * g.W);
        py[i] = (int)(rem / g.W);
        px[i] = (int)(rem % g.W);
      }
    }
  }
  __device__ void load(int k0) {
    const int k = k0 + kv;
    if constexpr (IM2COL) {
      const int tap = k0 / g.C;  // k-tile never straddles a tap (convC % BK == 0)
      const int c = k - tap * g.C;
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int sy = py[i] + dy, sx = px[i] + dx;
        const bool ok = grow[i] >= 0 && k < K && sy >= 0 && sy < g.H && sx >= 0 && sx < g.W;
        reg[i] = ok ? ldg16(base + (grow[i] + dy * g.W + dx) * ld + c) : zero16<T>();
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const bool ok = grow[i] >= 0 && k < K;
        if (ok && k + VEC > K) {  // ragged K tail (K % VEC != 0): element-wise, zero filled
          reg[i] = zero16<T>();
          const T* src = base + grow[i] * ld + k;
#pragma unroll
          for (int e = 0; e < VEC; e++)
            if (k + e < K) reg[i].v[e] = src[e];
        } else {
          reg[i] = ok ? ldg16(base + grow[i] * ld + k) : zero16<T>();
        }
      }
    }
  }
  __device__ void store(T* tile) const {
#pragma unroll
    for (int i = 0; i < 2; i++) stg16(tile + lrow[i] * ROW + kv, reg[i]);
  }
};

// ---- transposed-operand loader: row-major copy of [BK] reduction rows x 128 columns ------------
// MODE 0: dense  mem[k][col]            (row stride ld)
// MODE 1: conv3x3 dgrad weights: k = tap'*C + co -> W[co][8-tap'][col]   (row stride ld)
// MODE 2: wgrad im2col: k = pixel, col = tap*C + c -> X[pixel shifted by tap][c] (row stride ld)
template <typename T, int MODE>
struct TrLoader {
  static constexpr int VEC = TileCfg<T>::VEC, BK = TileCfg<T>::BK, ROW = TileCfg<T>::TR_ROW;
  static constexpr int CPR = 128 / VEC;        // chunks per row
  static constexpr int RSTEP = NTHREADS / CPR;  // row step between the thread's two chunks
  const T* base;
  int64_t ld;
  int K, ncols;  // reduction length, number of valid columns
  int col;       // global column of this thread's chunk
  int lcol, lrow0;
  ConvGeom g;
  int tdy, tdx, tc;  // MODE 2: tap offset and channel of this thread's chunk
  Vec16<T> reg[2];

  __device__ void init(const T* b, int64_t ld_, int ncols_, int K_, int col0, ConvGeom g_, int tid) {
    base = b; ld = ld_; K = K_; ncols = ncols_; g = g_;
    lcol = (tid % CPR) * VEC;
    lrow0 = tid / CPR;
    col = col0 + lcol;
    if constexpr (MODE == 2) {
      const int tap = col / g.C;
      tc = col - tap * g.C;
      tdy = tap / 3 - 1;
      tdx = tap % 3 - 1;
    }
  }
  __device__ Vec16<T> guarded(const T* src) const {
    if (col + VEC <= ncols) return ldg16(src);
    Vec16<T> v = zero16<T>();
#pragma unroll
    for (int e = 0; e < VEC; e++)
      if (col + e < ncols) v.v[e] = src[e];
    return v;
  }
  __device__ void load(int k0) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const long k = (long)k0 + lrow0 + i * RSTEP;
      if (k >= K || col >= ncols) { reg[i] = zero16<T>(); continue; }
      if constexpr (MODE == 0) {
        reg[i] = guarded(base + k * ld + col);
      } else if constexpr (MODE == 1) {
        const int tap = (int)(k / g.C), co = (int)(k - (long)tap * g.C);
        reg[i] = guarded(base + ((long)co * 9 + (8 - tap)) * ld + col);
      } else {
        const int x = (int)(k % g.W);
        const int y = (int)((k / g.W) % g.H);
        const int sy = y + tdy, sx = x + tdx;
        const bool ok = sy >= 0 && sy < g.H && sx >= 0 && sx < g.W;
        reg[i] = ok ? ldg16(base + (k + tdy * g.W + tdx) * ld + tc) : zero16<T>();
      }
    }
  }
  __device__ void store(T* tile) const {
#pragma unroll
    for (int i = 0; i < 2; i++) stg16(tile + (lrow0 + i * RSTEP) * ROW + lcol, reg[i]);
  }
};

template <typename T, int AL> struct ALoaderSel;
template <typename T> struct ALoaderSel<T, CROG_A_KC> { using type = KcLoader<T, false>; static constexpr bool TR = false; };
template <typename T> struct ALoaderSel<T, CROG_A_IM2COL> { using type = KcLoader<T, true>; static constexpr bool TR = false; };
template <typename T> struct ALoaderSel<T, CROG_A_MC> { using type = TrLoader<T, 0>; static constexpr bool TR = true; };
template <typename T, int BL> struct BLoaderSel;
template <typename T> struct BLoaderSel<T, CROG_B_KC> { using type = KcLoader<T, false>; static constexpr bool TR = false; };
template <typename T> struct BLoaderSel<T, CROG_B_NC> { using type = TrLoader<T, 0>; static constexpr bool TR = true; };
template <typename T> struct BLoaderSel<T, CROG_B_NC_DGRAD> { using type = TrLoader<T, 1>; static constexpr bool TR = true; };
template <typename T> struct BLoaderSel<T, CROG_B_NC_IM2COL> { using type = TrLoader<T, 2>; static constexpr bool TR = true; };

__device__ inline float apply_act(float v, int act) {
  if (act == CROG_ACT_RELU) return fmaxf(v, 0.f);
  if (act == CROG_ACT_QUICKGELU) return v / (1.f + expf(-1.702f * v));
  return v;
}

template <typename T, int AL, int BL, bool HWTR>
__global__ void __launch_bounds__(NTHREADS, 2) gemm_kernel(const crog_gemm_desc p) {
  using Cfg = TileCfg<T>;
  constexpr int BK = Cfg::BK;
  using ALd = typename ALoaderSel<T, AL>::type;
  using BLd = typename BLoaderSel<T, BL>::type;
  constexpr bool ATR = ALoaderSel<T, AL>::TR, BTR = BLoaderSel<T, BL>::TR;

  __shared__ __attribute__((aligned(16))) char smem[4 * OP_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // XCD-aware tile order: blocks that share an XCD (id % 8) get a contiguous run of tiles.
  const int tilesN = (p.N + BN - 1) / BN, tilesM = (p.M + BM - 1) / BM;
  const int nwg = tilesM * tilesN;
  int id = blockIdx.x;
  {
    const int q = nwg >> 3, rr = nwg & 7, xcd = id & 7, loc = id >> 3;
    id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  }
  const int tm = id / tilesN, tn = id % tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  const int z = blockIdx.y;
  const int zb = z / p.splitk, zs = z % p.splitk;
  const int zo = zb / p.batch_inner, zi = zb % p.batch_inner;
  const T* A = reinterpret_cast<const T*>(p.A) + zo * p.sAo + zi * p.sAi;
  const T* B = reinterpret_cast<const T*>(p.B) + zo * p.sBo + zi * p.sBi;
  const int64_t coff = zo * p.sCo + zi * p.sCi;

  const int ktiles = (p.K + BK - 1) / BK;
  const int per = (ktiles + p.splitk - 1) / p.splitk;
  const int kt0 = zs * per;
  const int kt1 = min(kt0 + per, ktiles);
  if (kt0 >= kt1) return;

  const ConvGeom g{p.convH, p.convW, p.convC};
  ALd la;
  BLd lb;
  la.init(A, p.lda, p.M, p.K, m0, g, tid);
  lb.init(B, p.ldb, p.N, p.K, n0, g, tid);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  // LDS ring: A slots at [0, 2*OP_BYTES), B slots at [2*OP_BYTES, 4*OP_BYTES)
  la.load(kt0 * BK);
  lb.load(kt0 * BK);
  la.store(reinterpret_cast<T*>(smem));
  lb.store(reinterpret_cast<T*>(smem + 2 * OP_BYTES));
  __syncthreads();

  int cur = 0;
  for (int kt = kt0; kt < kt1; kt++) {
    const bool more = kt + 1 < kt1;
    if (more) {
      la.load((kt + 1) * BK);
      lb.load((kt + 1) * BK);
    }
    const T* at = reinterpret_cast<const T*>(smem + cur * OP_BYTES);
    const T* bt = reinterpret_cast<const T*>(smem + (2 + cur) * OP_BYTES);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      Frag<T> fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        if constexpr (ATR) fa[i] = frag_tr<HWTR>(at, wr * 64 + i * 32, ks, lane);
        else fa[i] = frag_kc(at, wr * 64 + i * 32 + r, ks, h);
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if constexpr (BTR) fb[j] = frag_tr<HWTR>(bt, wc * 64 + j * 32, ks, lane);
        else fb[j] = frag_kc(bt, wc * 64 + j * 32 + r, ks, h);
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) mma16(fa[i], fb[j], acc[i][j]);
    }
    if (more) {
      la.store(reinterpret_cast<T*>(smem + (cur ^ 1) * OP_BYTES));
      lb.store(reinterpret_cast<T*>(smem + (2 + (cur ^ 1)) * OP_BYTES));
    }
    __syncthreads();
    cur ^= 1;
  }

  // ------------------------------------------ epilogue ------------------------------------------
  const float alpha = p.alpha;
  const float* bias = (zs == 0) ? p.bias : nullptr;
  float bcol[2];
  int ncol[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    ncol[j] = n0 + wc * 64 + j * 32 + r;
    bcol[j] = (bias && ncol[j] < p.N) ? bias[ncol[j]] : 0.f;
  }
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        float v = alpha * acc[i][j][e] + bcol[j];
        if (p.col_stats) {
          const int m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (m < p.M) { s1[j] += v; s2[j] += v * v; }
        }
        acc[i][j][e] = apply_act(v, p.act);
      }

  if (p.col_stats) {  // block-uniform branch
    float* red = reinterpret_cast<float*>(smem);  // [2 wr][128][2]; the k-loop's last barrier has passed
#pragma unroll
    for (int j = 0; j < 2; j++) {
      s1[j] += __shfl_xor(s1[j], 32, 64);
      s2[j] += __shfl_xor(s2[j], 32, 64);
      if (h == 0) {
        const int c = wc * 64 + j * 32 + r;
        red[(wr * 128 + c) * 2 + 0] = s1[j];
        red[(wr * 128 + c) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    if (tid < 128 && n0 + tid < p.N) {
      float* dst = p.col_stats + ((int64_t)tm * p.N + n0 + tid) * 2;
      dst[0] = red[tid * 2] + red[(128 + tid) * 2];
      dst[1] = red[tid * 2 + 1] + red[(128 + tid) * 2 + 1];
    }
    __syncthreads();
  }

  const T* R = reinterpret_cast<const T*>(p.R);
  if (p.out_mode != CROG_OUT_T || sizeof(T) == 4) {
    // direct stores from the accumulator layout: a register covers 2 rows x 32 consecutive columns
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const int n = ncol[j];
          if (m < p.M && n < p.N) {
            float v = acc[i][j][e];
            if (R) v += Elem<T>::to_f(R[(int64_t)m * p.ldr + n]);
            const int64_t o = coff + (int64_t)m * p.ldc + n;
            if (p.out_mode == CROG_OUT_F32_ATOMIC) atomicAdd(reinterpret_cast<float*>(p.C) + o, v);
            else if (p.out_mode == CROG_OUT_F32) reinterpret_cast<float*>(p.C)[o] = v;
            else reinterpret_cast<T*>(p.C)[o] = Elem<T>::from_f(v);
          }
        }
  } else {
    // 2-byte output: stage the tile in LDS, then residual-add and store 16 bytes per lane
    if constexpr (sizeof(T) == 2) {
      constexpr int CROW = 136;
      T* Cs = reinterpret_cast<T*>(smem);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int e = 0; e < 16; e++) {
            const int lr = wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            Cs[lr * CROW + wc * 64 + j * 32 + r] = (T)acc[i][j][e];
          }
      __syncthreads();
      T* C = reinterpret_cast<T*>(p.C) + coff;
#pragma unroll
      for (int it = 0; it < 8; it++) {
        const int v = tid + it * NTHREADS;
        const int lr = v >> 4, cv = (v & 15) * 8;
        const int m = m0 + lr, n = n0 + cv;
        if (m < p.M && n < p.N) {
          Vec16<T> o = *reinterpret_cast<const Vec16<T>*>(Cs + lr * CROW + cv);
          if (n + 8 <= p.N) {
            if (R) {
              Vec16<T> rv = ldg16(R + (int64_t)m * p.ldr + n);
#pragma unroll
              for (int e = 0; e < 8; e++) o.v[e] = (T)((float)o.v[e] + (float)rv.v[e]);
            }
            stg16(C + (int64_t)m * p.ldc + n, o);
          } else {
            for (int e = 0; e < 8 && n + e < p.N; e++) {
              float f = (float)o.v[e];
              if (R) f += (float)R[(int64_t)m * p.ldr + n + e];
              C[(int64_t)m * p.ldc + n + e] = (T)f;
            }
          }
        }
      }
    }
  }
}

template <typename T, int AL, int BL, bool HWTR>
int launch(const crog_gemm_desc& d, hipStream_t s) {
  const int tilesM = cdiv(d.M, BM), tilesN = cdiv(d.N, BN);
  dim3 grid(tilesM * tilesN, d.batch * d.splitk, 1);
  hipLaunchKernelGGL((gemm_kernel<T, AL, BL, HWTR>), grid, dim3(NTHREADS), 0, s, d);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

template <typename T, bool HWTR>
int dispatch_layout(const crog_gemm_desc& d, hipStream_t s) {
  const int a = d.a_layout, b = d.b_layout;
  if (a == CROG_A_KC && b == CROG_B_KC) return launch<T, CROG_A_KC, CROG_B_KC, true>(d, s);
  if (a == CROG_A_IM2COL && b == CROG_B_KC) return launch<T, CROG_A_IM2COL, CROG_B_KC, true>(d, s);
  if (a == CROG_A_KC && b == CROG_B_NC) return launch<T, CROG_A_KC, CROG_B_NC, HWTR>(d, s);
  if (a == CROG_A_IM2COL && b == CROG_B_NC_DGRAD) return launch<T, CROG_A_IM2COL, CROG_B_NC_DGRAD, HWTR>(d, s);
  if (a == CROG_A_MC && b == CROG_B_NC) return launch<T, CROG_A_MC, CROG_B_NC, HWTR>(d, s);
  if (a == CROG_A_MC && b == CROG_B_NC_IM2COL) return launch<T, CROG_A_MC, CROG_B_NC_IM2COL, HWTR>(d, s);
  if (a == CROG_A_MC && b == CROG_B_KC) return launch<T, CROG_A_MC, CROG_B_KC, HWTR>(d, s);
  crog_set_error("crog_gemm: unsupported layout combination a=%d b=%d", a, b);
  return CROG_ERR_ARG;
}

bool hwtr_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CROG_GEMM_NO_HWTR");
    v = (e && e[0] == '1') ? 0 : 1;
  }
  return v == 1;
}

}  // namespace

extern "C" int crog_gemm_stat_tiles(int M) { return cdiv(M, BM); }

extern "C" int crog_gemm(const crog_gemm_desc* dp, crog_stream_t stream) {
  CROG_CHECK_ARG(dp != nullptr, "crog_gemm: null descriptor");
  crog_gemm_desc d = *dp;
  CROG_CHECK_ARG(d.dtype == CROG_F32 || d.dtype == CROG_BF16, "crog_gemm: bad dtype %d", d.dtype);
  const int vec = d.dtype == CROG_BF16 ? 8 : 4;
  const size_t esz = d.dtype == CROG_BF16 ? 2 : 4;
  if (d.M == 0 || d.N == 0 || d.batch == 0) return CROG_OK;
  CROG_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0 && d.batch > 0, "crog_gemm: bad sizes M=%d N=%d K=%d batch=%d", d.M, d.N, d.K, d.batch);
  if (d.batch_inner < 1) d.batch_inner = 1;
  if (d.splitk < 1) d.splitk = 1;
  CROG_CHECK_ARG(d.batch % d.batch_inner == 0, "crog_gemm: batch %% batch_inner != 0");
  CROG_CHECK_ARG(d.A && d.B && d.C, "crog_gemm: null operand");
  CROG_CHECK_ARG(d.lda % vec == 0 && d.ldb % vec == 0, "crog_gemm: lda/ldb must be multiples of %d elements", vec);
  CROG_CHECK_ARG(((uintptr_t)d.A % 16) == 0 && ((uintptr_t)d.B % 16) == 0, "crog_gemm: A/B must be 16-byte aligned");
  CROG_CHECK_ARG((d.sAo * esz) % 16 == 0 && (d.sAi * esz) % 16 == 0 && (d.sBo * esz) % 16 == 0 && (d.sBi * esz) % 16 == 0,
                 "crog_gemm: batch strides of A/B must keep 16-byte alignment");
  if (d.a_layout == CROG_A_IM2COL || d.b_layout == CROG_B_NC_DGRAD || d.b_layout == CROG_B_NC_IM2COL) {
    CROG_CHECK_ARG(d.convH > 0 && d.convW > 0 && d.convC > 0 && d.convC % 32 == 0,
                   "crog_gemm: conv geometry needs convC %% 32 == 0 (H=%d W=%d C=%d)", d.convH, d.convW, d.convC);
    if (d.a_layout == CROG_A_IM2COL)
      CROG_CHECK_ARG(d.K == 9 * d.convC && d.M % (d.convH * d.convW) == 0, "crog_gemm: im2col needs K == 9*convC and M %% (H*W) == 0");
    if (d.b_layout == CROG_B_NC_IM2COL)
      CROG_CHECK_ARG(d.N == 9 * d.convC && d.K % (d.convH * d.convW) == 0, "crog_gemm: wgrad im2col needs N == 9*convC and K %% (H*W) == 0");
  }
  if (d.out_mode == CROG_OUT_T) {
    CROG_CHECK_ARG(d.ldc % vec == 0 && ((uintptr_t)d.C % 16) == 0 && (d.sCo * esz) % 16 == 0 && (d.sCi * esz) % 16 == 0,
                   "crog_gemm: C (dtype output) must be 16-byte aligned with ldc %% %d == 0", vec);
    if (d.R) CROG_CHECK_ARG(d.ldr % vec == 0 && ((uintptr_t)d.R % 16) == 0, "crog_gemm: R must be 16-byte aligned");
  }
  CROG_CHECK_ARG(d.splitk == 1 || (d.out_mode == CROG_OUT_F32_ATOMIC && d.act == CROG_ACT_NONE && !d.R),
                 "crog_gemm: splitk > 1 needs atomic fp32 output, no activation, no residual");
  CROG_CHECK_ARG(!d.R || d.batch == 1, "crog_gemm: residual only for unbatched GEMM");
  CROG_CHECK_ARG(!d.col_stats || (d.batch == 1 && d.splitk == 1), "crog_gemm: col_stats needs batch == 1 and splitk == 1");
  CROG_CHECK_ARG((long)d.batch * d.splitk <= 65535, "crog_gemm: batch*splitk too large");
  hipStream_t s = (hipStream_t)stream;
  const bool hw = hwtr_enabled();
  if (d.dtype == CROG_BF16) return hw ? dispatch_layout<bf16, true>(d, s) : dispatch_layout<bf16, false>(d, s);
  return dispatch_layout<float, true>(d, s);
}
