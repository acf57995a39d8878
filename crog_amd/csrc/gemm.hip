// MFMA GEMM / implicit-GEMM conv family for gfx950 (see include/crog_hip.h, crog_gemm).
//
// A workgroup of WVM x WVN waves computes a (32*WM*WVM) x (32*WN*WVN) tile of C; each wave owns WM x WN MFMA
// 32x32 accumulators.  Three tile shapes are instantiated and chosen per problem on the host:
//     BIG   256x128, 4 waves (2x2), 128 acc regs/lane  — large convolutions: 85 FLOP per byte staged into the CU
//     MID   128x128, 4 waves (2x2)                      — default
//     SMALL  64x64,  4 waves (2x2)                      — latency-bound small GEMMs (text transformer, attention)
// Operands are staged global -> registers -> LDS with a two-deep LDS ring (loads for k-tile t+1 are issued before
// the MFMAs of tile t and written to LDS after them), one barrier per k-tile.
//
//   bf16: v_mfma_f32_32x32x16_bf16, BK = 32        f32: v_mfma_f32_32x32x2_f32 (exact f32), BK = 16
//
// Operand layouts (crog_a_layout / crog_b_layout):
//   K-contiguous operands land in LDS as [rows][BK+pad] and fragments are single ds_read_b128.
//   Transposed operands (reduction index is the slow memory index: dgrad weights, wgrad, P.V)
//   are copied row-major into LDS as [BK][cols+pad] with full 16-byte coalesced loads and are
//   transposed on the READ side: ds_read_b64_tr_b16 for bf16, ds_read_b32 for f32.
#include "common.h"
#include "gemm_dma.h"
#include "comm_dev.h"
#include <type_traits>
#include <stdlib.h>
#include <algorithm>

// the ping-pong 256 x 256 x 64 kernel (gemm_pp.hip)
bool crog_gemm_pp_eligible(const crog_gemm_desc& d, int rows);
int crog_gemm_pp_launch(const crog_gemm_desc& d, int rows, int dist, hipStream_t s, int full = 0);
bool crog_gemm_pp_bwdz_ok(const crog_gemm_desc& d);
bool crog_gemm_pp_full_epilogue_ok(const crog_gemm_desc& d);
bool crog_conv_sw_eligible(const crog_gemm_desc& d);      // conv_sw.hip: sliding-window 3x3 convolution for 32 / 64 channels
int crog_conv_sw_launch(const crog_gemm_desc& d, hipStream_t s);
bool crog_gemm_skinny_eligible(const crog_gemm_desc& d);  // gemm_skinny.hip: [M][32] x [32][32] streamed without an LDS stage (the stem's first convolution)
int crog_gemm_skinny_launch(const crog_gemm_desc& d, hipStream_t s);
bool crog_wgrad_sw_eligible(const crog_gemm_desc& d);     // wgrad_sw.hip: sliding-window 3x3 weight gradient for 32 / 64 channels
int crog_wgrad_sw_launch(const crog_gemm_desc& d, hipStream_t s);
// the ping-pong weight-gradient kernel (gemm_ppt.hip)
bool crog_gemm_ppt_eligible(const crog_gemm_desc& d);
int crog_gemm_ppt_launch(const crog_gemm_desc& d, int dist, hipStream_t s);

namespace {

template <typename T> struct TileCfg;
template <> struct TileCfg<bf16> {
  static constexpr int VEC = 8, BK = 32;
  static constexpr int KC_ROW = 40;   // elements; 80-byte rows -> conflict-free ds_read_b128
  static constexpr int TR_PAD = 32;   // elements; row bytes = 2*cols + 64 -> (bytes/4) % 64 == 16: conflict-free ds_read_b64_tr_b16
};
template <> struct TileCfg<float> {
  static constexpr int VEC = 4, BK = 16;
  static constexpr int KC_ROW = 20;   // 80-byte rows
  static constexpr int TR_PAD = 4;
};

template <typename T, int ROWS> constexpr int op_bytes() {
  constexpr int kc = ROWS * TileCfg<T>::KC_ROW * (int)sizeof(T);
  constexpr int tr = TileCfg<T>::BK * (ROWS + TileCfg<T>::TR_PAD) * (int)sizeof(T);
  return (kc > tr ? kc : tr + 15) / 16 * 16;
}

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
// Row-major-copy LDS tile [BK][ROW]; wave's 32 columns start at `cbase`.
template <bool HWTR, int ROW>
__device__ inline Frag<bf16> frag_tr(const bf16* tile, int cbase, int ks, int lane) {
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
template <bool HWTR, int ROW>
__device__ inline Frag<float> frag_tr(const float* tile, int cbase, int ks, int lane) {
  const int h = lane >> 5, col = cbase + (lane & 31);
  Frag<float> f;
#pragma unroll
  for (int j = 0; j < 8; j++) f.v[j] = tile[(ks * 16 + 8 * h + j) * ROW + col];
  return f;
}

struct ConvGeom { int H, W, C; };

// fp32 (the parity mode) accumulates k-blocked: every KBLOCK_F32 k-tiles (128 reduction indices) the running MFMA accumulators are
// folded into a second set and restarted.  One long sequential-k chain is the noisier summation - on BASELINE config 1 it put the
// HIP logits 2.5x as far from the float64 result as the reference's blocked CPU GEMM (tests/test_fulldepth_gpu.py); partial sums
// of 128 bound the chain length the way a blocked GEMM does.  bf16 keeps one chain (its operands carry 8 bits; and no registers to spare).
constexpr int KBLOCK_F32 = 8;
template <int WM, int WN>
__device__ __attribute__((always_inline)) inline void kblock_fold(f32x16 (&acc)[WM][WN], f32x16 (&tot)[WM][WN]) {
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        tot[i][j][e] += acc[i][j][e];
        acc[i][j][e] = 0.f;
      }
}
template <int WM, int WN>
__device__ __attribute__((always_inline)) inline void kblock_finish(f32x16 (&acc)[WM][WN], const f32x16 (&tot)[WM][WN]) {
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] += tot[i][j][e];
}

// Reduction order of the 3x3 forward / data-gradient forms: k-tile index -> (tap, 32-channel chunk).
// Chunk-PAIR major, tap, then the two chunks of the pair: consecutive k-tiles read the two 64-byte halves of the same
// 128-byte lines of a pixel (full-line use of L2), and the 9 taps of a chunk pair are 18 consecutive k-tiles (L2/L1 reuse
// of the shifted pixels).  Falls back to chunk-major / tap-minor when the number of chunks is odd (Cin = 32).
__device__ inline void conv_ktile(int kt, int nchunks, int& tap, int& chunk) {
  // both forms are a handful of scalar instructions: computed unconditionally and selected, so that the caller's loop body stays
  // one basic block
  const int pair = kt / 18, r = kt - pair * 18;
  const int tap_e = r >> 1, chunk_e = pair * 2 + (r & 1);
  const int chunk_o = kt / 9, tap_o = kt - chunk_o * 9;
  const bool even = (nchunks & 1) == 0;
  tap = even ? tap_e : tap_o;
  chunk = even ? chunk_e : chunk_o;
}

// ---- K-contiguous loader: ROWS rows x BK, 4 16-byte chunks per row, thread -> rows {tid/4 + i*NT/4} ----
template <typename T, bool IM2COL, int ROWS, int NT>
struct KcLoader {
  static constexpr int VEC = TileCfg<T>::VEC, BK = TileCfg<T>::BK, ROW = TileCfg<T>::KC_ROW;
  static constexpr int NV = ROWS * 4 / NT;
  static_assert(NV >= 1 && ROWS * 4 % NT == 0, "tile/threads mismatch");
  const T* base;
  int64_t ld;
  int K;
  int kv;             // element offset of this thread's chunk inside the k-tile
  int lrow[NV];       // tile-local rows
  long grow[NV];      // global row (pixel) index, -1 when out of range
  int py[NV], px[NV]; // im2col: pixel coordinates
  ConvGeom g;
  Vec16<T> reg[NV];

  __device__ __attribute__((always_inline)) void init(const T* b, int64_t ld_, int rows_total, int K_, int row0, ConvGeom g_, int tid) {
    base = b; ld = ld_; K = K_; g = g_;
    kv = (tid & 3) * VEC;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      lrow[i] = (tid >> 2) + i * (NT / 4);
      long m = (long)row0 + lrow[i];
      grow[i] = (m < rows_total) ? m : -1;
      if constexpr (IM2COL) {
        long rem = m % ((long)g.H * g.W);
        py[i] = (int)(rem / g.W);
        px[i] = (int)(rem % g.W);
      }
    }
  }
  __device__ __attribute__((always_inline)) void load(int k0) {
    const int k = k0 + kv;
    if constexpr (IM2COL) {
      const int tap = k0 / g.C;  // k-tile never straddles a tap (convC % BK == 0)
      const int c = k - tap * g.C;
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
      for (int i = 0; i < NV; i++) {
        const int sy = py[i] + dy, sx = px[i] + dx;
        const bool ok = grow[i] >= 0 && k < K && sy >= 0 && sy < g.H && sx >= 0 && sx < g.W;
        reg[i] = ok ? ldg16(base + (grow[i] + dy * g.W + dx) * ld + c) : zero16<T>();
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; i++) {
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
  __device__ __attribute__((always_inline)) void store(T* tile) const {
#pragma unroll
    for (int i = 0; i < NV; i++) stg16(tile + lrow[i] * ROW + kv, reg[i]);
  }
};

// ---- transposed-operand loader: row-major copy of [BK] reduction rows x COLS columns --------------
// MODE 0: dense  mem[k][col]            (row stride ld)
// MODE 1: conv3x3 dgrad weights: k = tap'*C + co -> W[co][8-tap'][col]   (row stride ld)
// MODE 2: wgrad im2col: k = pixel, col = tap*C + c -> X[pixel shifted by tap][c] (row stride ld)
template <typename T, int MODE, int COLS, int NT>
struct TrLoader {
  static constexpr int VEC = TileCfg<T>::VEC, BK = TileCfg<T>::BK, ROW = COLS + TileCfg<T>::TR_PAD;
  static constexpr int CPR = COLS / VEC;   // chunks per row
  static constexpr int RSTEP = NT / CPR;   // row step between the thread's chunks
  static constexpr int NV = BK / RSTEP;
  static_assert(NV >= 1 && NT % CPR == 0 && BK % RSTEP == 0, "tile/threads mismatch");
  const T* base;
  int64_t ld;
  int K, ncols;  // reduction length, number of valid columns
  int col;       // global column of this thread's chunk
  int lcol, lrow0;
  ConvGeom g;
  int tdy, tdx, tc;  // MODE 2: tap offset and channel of this thread's chunk
  Vec16<T> reg[NV];

  __device__ __attribute__((always_inline)) void init(const T* b, int64_t ld_, int ncols_, int K_, int col0, ConvGeom g_, int tid) {
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
  __device__ __attribute__((always_inline)) Vec16<T> guarded(const T* src) const {
    if (col + VEC <= ncols) return ldg16(src);
    Vec16<T> v = zero16<T>();
#pragma unroll
    for (int e = 0; e < VEC; e++)
      if (col + e < ncols) v.v[e] = src[e];
    return v;
  }
  __device__ __attribute__((always_inline)) void load(int k0) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
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
  __device__ __attribute__((always_inline)) void store(T* tile) const {
#pragma unroll
    for (int i = 0; i < NV; i++) stg16(tile + (lrow0 + i * RSTEP) * ROW + lcol, reg[i]);
  }
};

template <typename T, int AL, int ROWS, int NT> struct ALoaderSel;
template <typename T, int R, int NT> struct ALoaderSel<T, CROG_A_KC, R, NT> { using type = KcLoader<T, false, R, NT>; static constexpr bool TR = false; };
template <typename T, int R, int NT> struct ALoaderSel<T, CROG_A_IM2COL, R, NT> { using type = KcLoader<T, true, R, NT>; static constexpr bool TR = false; };
template <typename T, int R, int NT> struct ALoaderSel<T, CROG_A_MC, R, NT> { using type = TrLoader<T, 0, R, NT>; static constexpr bool TR = true; };
template <typename T, int BL, int ROWS, int NT> struct BLoaderSel;
template <typename T, int R, int NT> struct BLoaderSel<T, CROG_B_KC, R, NT> { using type = KcLoader<T, false, R, NT>; static constexpr bool TR = false; };
template <typename T, int R, int NT> struct BLoaderSel<T, CROG_B_NC, R, NT> { using type = TrLoader<T, 0, R, NT>; static constexpr bool TR = true; };
template <typename T, int R, int NT> struct BLoaderSel<T, CROG_B_NC_DGRAD, R, NT> { using type = TrLoader<T, 1, R, NT>; static constexpr bool TR = true; };
template <typename T, int R, int NT> struct BLoaderSel<T, CROG_B_NC_IM2COL, R, NT> { using type = TrLoader<T, 2, R, NT>; static constexpr bool TR = true; };

__device__ inline float frag_sum(const Frag<bf16>& f) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; j++) s += (float)f.v[j];
  return s;
}
__device__ inline float frag_sum(const Frag<float>& f) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; j++) s += f.v[j];
  return s;
}
// a_sum[m] += sum over this block's k range of A(m, k): lanes of the two half-waves hold disjoint k slots of the same row
template <int WM>
__device__ inline void flush_a_sum(const float (&asum)[WM], float* dst, int mbase, int M, int lane) {
#pragma unroll
  for (int i = 0; i < WM; i++) {
    const float v = xor32_sum(asum[i]);
    const int m = mbase + i * 32 + (lane & 31);
    if ((lane >> 5) == 0 && m < M) atomicAdd(dst + m, v);
  }
}

// Branch-free activations: a per-lane branch inside the register loops (ocml's tanhf has one at |x| = 0.625) makes the compiler
// carry the f32x16 accumulators through divergent control flow as whole vectors and spill them.
// (v_rcp_f32 is within 1 ulp; an IEEE division would add a dozen temporaries per element to loops that hold 64-128 accumulators)

// value of lane ^ 1 (quad_perm [1, 0, 3, 2]): a DPP move, no LDS
__device__ inline float dpp_swap1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}

// LEAN: the tile is only ever launched with the plain bf16 epilogue (alpha 1, no bias / activation / residual, dtype output, even N;
// column statistics optional) and the other epilogue paths are not compiled into it (lean_epilogue_ok() is the dispatch-side check)
// NSTAGE: depth of the LDS-DMA ring (0 = the default of the tile size, dma_nstage()).
// BWD: a LEAN tile whose statistics are the BatchNorm-BACKWARD ones (crog_gemm_desc.bwd_z); a kernel of its own so that the
// gather of z does not cost the other instantiations registers.
// ATOM: a LEAN tile whose ONLY epilogue is the fp32 atomic add of a split-K weight gradient (out_mode CROG_OUT_F32_ATOMIC, alpha 1, no
// a_sum): the 256 x 256 tile for the large 3x3 weight gradients.
template <int WM_, int WN_, int WVM_, int WVN_, bool LEAN_ = false, int NSTAGE_ = 0, bool BWD_ = false, bool ATOM_ = false>
struct Shape {
  static constexpr int WM = WM_, WN = WN_, WVM = WVM_, WVN = WVN_, NSTAGE = NSTAGE_;
  static constexpr bool LEAN = LEAN_, BWD = BWD_, ATOM = ATOM_;
  static_assert(!BWD_ || LEAN_, "the backward-statistics epilogue is a lean one");
  static_assert(!ATOM_ || (LEAN_ && !BWD_), "the atomic-only epilogue is a lean one");
  static constexpr int NT = 64 * WVM * WVN, BM = 32 * WM * WVM, BN = 32 * WN * WVN;
};
// (a 256 x 128 register-staged variant was measured slower than 128 x 128 -- occupancy-bound staging -- and spilled at 256 VGPRs: removed)
using ShapeMid = Shape<2, 2, 2, 2>;    // 128 x 128, 256 threads
using ShapeSmall = Shape<1, 1, 2, 2>;  //  64 x  64, 256 threads

template <typename T, typename S> constexpr int lds_bytes() {
  constexpr int ring = 2 * (op_bytes<T, S::BM>() + op_bytes<T, S::BN>());
  constexpr int stage = (S::NT / 64) * 32 * (32 * S::WN + 8) * 2;           // bf16 epilogue staging, wave-private
  constexpr int red = S::WVM * S::BN * 2 * 4;                              // column-statistics exchange
  constexpr int m1 = ring > stage ? ring : stage;
  return m1 > red ? m1 : red;
}

// ------------------------------------------ epilogue ------------------------------------------
// Shared by the register-staged and the LDS-DMA kernels.  `smem` may be reused: the caller has passed its last barrier.
template <typename T, typename S>
__device__ __attribute__((always_inline)) inline void gemm_epilogue(f32x16 (&acc)[S::WM][S::WN], const crog_gemm_desc& p, char* smem, int m0, int n0, int zs, int64_t coff) {
  constexpr int NT = S::NT, BM = S::BM, BN = S::BN, WM = S::WM, WN = S::WN, WVN = S::WVN, WVM = S::WVM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WVN, wc = wave % WVN;
  const int r = lane & 31, h = lane >> 5;
  const float alpha = p.alpha;
  const float* bias = (zs == 0) ? p.bias : nullptr;
  float bcol[WN];
  int ncol[WN];
#pragma unroll
  for (int j = 0; j < WN; j++) {
    ncol[j] = n0 + (wc * WN + j) * 32 + r;
    bcol[j] = (bias && ncol[j] < p.N) ? bias[ncol[j]] : 0.f;
  }
  float s1[WN], s2[WN];
#pragma unroll
  for (int j = 0; j < WN; j++) s1[j] = s2[j] = 0.f;
  // every flag below is block-uniform: test it once around the register loops, never per element.  The loops go one 32 x 32 block
  // at a time with a scheduling barrier between blocks: left to itself the scheduler interleaves all WM * WN * 16 elements of a pass
  // (division sequences of the activations, 64-bit store addresses) and spills the accumulators to make room.
  auto per_block = [&](auto&& body) {
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
      for (int j = 0; j < WN; j++) {
        body(i, j);
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  constexpr bool LEAN = S::LEAN;
  if constexpr (S::ATOM) {
    // split-K weight gradient: accumulators -> fp32 atomic adds, one 32 x 32 block at a time (a register covers 2 rows x 32
    // consecutive columns: a wave-instruction is two 128-byte runs).  Nothing else is compiled into this tile.
    float* Cf = reinterpret_cast<float*>(p.C) + coff;
    const bool interior = m0 + BM <= p.M && n0 + BN <= p.N;
    auto add_all = [&](auto guarded) {
      constexpr bool G = decltype(guarded)::value;
#pragma unroll
      for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++) {
          const int mb = m0 + (wr * WM + i) * 32 + 4 * h;
          float* cb = Cf + (int64_t)mb * p.ldc + ncol[j];
#pragma unroll
          for (int e = 0; e < 16; e++) {
            const int ro = (e & 3) + 8 * (e >> 2);
            if (!G || (mb + ro < p.M && ncol[j] < p.N)) atomicAdd(cb + (int64_t)ro * p.ldc, acc[i][j][e]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (interior) add_all(std::false_type{});
    else add_all(std::true_type{});
    return;
  }
  if (!LEAN && (alpha != 1.f || bias)) {
    per_block([&](int i, int j) {
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = alpha * acc[i][j][e] + bcol[j];
    });
  }
  if (S::BWD && p.col_stats && p.bwd_z) {
    // BatchNorm-backward statistics (crog_hip.h: bwd_z): the accumulators are a gradient dy; gate it with the ReLU mask recomputed
    // from z, keep the gated value (it is what gets stored) and accumulate (sum g, sum g * z).  z is fetched in the pair layout of
    // the bf16 store path — even lanes read row(2q), odd lanes row(2q + 1), two adjacent columns each — and one DPP swap hands
    // every lane the value of its own column in the other row.  Compiled only into the BWD tile (Shape<..., BWD>).
    if constexpr (sizeof(T) == 2 && S::BWD) {
      const T* Z = reinterpret_cast<const T*>(p.bwd_z);
      const bool odd = lane & 1;
      auto gated_stats = [&](auto guarded) {
        constexpr bool G = decltype(guarded)::value;
        per_block([&](int i, int j) {
          const bool okc = p.bwd_ss && ncol[j] < p.N;
          const float gsc = okc ? p.bwd_ss[2 * ncol[j]] : 0.f;
          const float gsh = okc ? p.bwd_ss[2 * ncol[j] + 1] : 1.f;      // no ReLU: gate = 0 * z + 1 > 0
          const int colp = n0 + (wc * WN + j) * 32 + (r & ~1);
          const int mrow = m0 + (wr * WM + i) * 32 + 4 * h + (odd ? 1 : 0);
          const T* zb = Z + (int64_t)mrow * p.ldz + colp;
          unsigned P[8];
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const int ro = (2 * q & 3) + 8 * (2 * q >> 2);
            if (!G || (mrow + ro < p.M && colp < p.N)) P[q] = *reinterpret_cast<const unsigned*>(zb + (int64_t)ro * p.ldz);
            else P[q] = 0u;
          }
          // residual layers (out = relu(bn(z) + identity)): the identity path's gradient R joins BEFORE the gate, and the gate is the
          // forward's bit mask (crog_bn_apply: one byte per 8 columns) - z alone cannot tell which elements passed that ReLU
          unsigned RP[8], MB[8];
          const bool has_r = p.R != nullptr, has_m = p.bwd_mask != nullptr;
          if (has_r) {
            const T* rb = reinterpret_cast<const T*>(p.R) + (int64_t)mrow * p.ldr + colp;
#pragma unroll
            for (int q = 0; q < 8; q++) {
              const int ro = (2 * q & 3) + 8 * (2 * q >> 2);
              if (!G || (mrow + ro < p.M && colp < p.N)) RP[q] = *reinterpret_cast<const unsigned*>(rb + (int64_t)ro * p.ldr);
              else RP[q] = 0u;
            }
          }
          if (has_m) {
            const unsigned char* mb = p.bwd_mask + (int64_t)mrow * (p.N >> 3) + (colp >> 3);
#pragma unroll
            for (int q = 0; q < 8; q++) {
              const int ro = (2 * q & 3) + 8 * (2 * q >> 2);
              if (!G || (mrow + ro < p.M && colp < p.N)) MB[q] = mb[(int64_t)ro * (p.N >> 3)];
              else MB[q] = 0u;
            }
          }
          const int mbit = (colp & 7) + (odd ? 1 : 0);      // this lane's column inside its mask byte
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const unsigned O = (unsigned)__builtin_amdgcn_update_dpp(0, (int)P[q], 0xB1, 0xF, 0xF, false);   // lane ^ 1
            // bf16 -> f32 is a 16-bit shift: low half = first column of the pair, high half = second
            const float z0 = __builtin_bit_cast(float, odd ? (O & 0xffff0000u) : (P[q] << 16));
            const float z1 = __builtin_bit_cast(float, odd ? (P[q] & 0xffff0000u) : (O << 16));
            float a0 = acc[i][j][2 * q], a1 = acc[i][j][2 * q + 1];
            if (has_r) {
              const unsigned RO = (unsigned)__builtin_amdgcn_update_dpp(0, (int)RP[q], 0xB1, 0xF, 0xF, false);
              a0 += __builtin_bit_cast(float, odd ? (RO & 0xffff0000u) : (RP[q] << 16));
              a1 += __builtin_bit_cast(float, odd ? (RP[q] & 0xffff0000u) : (RO << 16));
            }
            if (has_m) {
              const unsigned MO = (unsigned)__builtin_amdgcn_update_dpp(0, (int)MB[q], 0xB1, 0xF, 0xF, false);
              const unsigned b0 = odd ? MO : MB[q], b1 = odd ? MB[q] : MO;      // the mask bytes of row(2q) / row(2q + 1)
              a0 = ((b0 >> mbit) & 1u) ? a0 : 0.f;
              a1 = ((b1 >> mbit) & 1u) ? a1 : 0.f;
            }
            a0 = (z0 * gsc + gsh > 0.f) ? a0 : 0.f;
            a1 = (z1 * gsc + gsh > 0.f) ? a1 : 0.f;
            acc[i][j][2 * q] = a0;
            acc[i][j][2 * q + 1] = a1;
            s1[j] += a0 + a1;
            s2[j] += a0 * z0 + a1 * z1;
          }
        });
      };
      if (m0 + BM <= p.M && n0 + BN <= p.N) gated_stats(std::false_type{});
      else gated_stats(std::true_type{});
    }
  } else if (p.col_stats) {
    if (m0 + BM <= p.M) {   // interior tile: no row guard
      per_block([&](int i, int j) {
#pragma unroll
        for (int e = 0; e < 16; e++) { const float v = acc[i][j][e]; s1[j] += v; s2[j] += v * v; }
      });
    } else {
      per_block([&](int i, int j) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int m = m0 + (wr * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const float v = m < p.M ? acc[i][j][e] : 0.f;
          s1[j] += v;
          s2[j] += v * v;
        }
      });
    }
  }
  if (LEAN) {
  } else if (p.act == CROG_ACT_RELU) {
    per_block([&](int i, int j) {
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = fmaxf(acc[i][j][e], 0.f);
    });
  } else if (p.act == CROG_ACT_QUICKGELU) {
    per_block([&](int i, int j) {
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = act_quickgelu(acc[i][j][e]);
    });
  } else if (p.act == CROG_ACT_TANH) {
    per_block([&](int i, int j) {
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = act_tanh(acc[i][j][e]);
    });
  }

  if (p.col_stats) {  // block-uniform branch; slab rows are 128 matrix rows each
    constexpr int RG = 32 * WM;          // rows per wave-row
    constexpr int GPS = 128 / RG;        // wave-rows per slab row (MID: 2, BIG: 1)
    static_assert(RG <= 128 && 128 % RG == 0 && BM % 128 == 0 || BM < 128, "stats slab granularity");
    float* red = reinterpret_cast<float*>(smem);  // [WVM][BN][2]; the k-loop's last barrier has passed
#pragma unroll
    for (int j = 0; j < WN; j++) {
      s1[j] = xor32_sum(s1[j]);
      s2[j] = xor32_sum(s2[j]);
      if (h == 0) {
        const int c = (wc * WN + j) * 32 + r;
        red[(wr * BN + c) * 2 + 0] = s1[j];
        red[(wr * BN + c) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    constexpr int SLABS = (BM >= 128) ? BM / 128 : 1;
    // one thread per float of the tile's [BN][2] statistics row: consecutive lanes store / add to consecutive addresses (a wave's
    // atomics are 256 contiguous bytes, the shape the memory-side atomic units take at full rate)
    for (int idx = tid; idx < SLABS * BN * 2; idx += NT) {
      const int sl = idx / (BN * 2), k = idx % (BN * 2), c = k >> 1;
      if (n0 + c < p.N && m0 + sl * 128 < p.M) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < GPS; q++) v += red[(sl * GPS + q) * BN * 2 + k];
        if (p.stat_replicas > 0) {   // accumulate into one of `stat_replicas` pre-zeroed [N][2] rows: no slab, no reduction kernel
          atomicAdd(p.col_stats + ((int64_t)((m0 / 128 + sl) % p.stat_replicas) * p.N + n0) * 2 + k, v);
        } else {
          p.col_stats[((int64_t)(m0 / 128 + sl) * p.N + n0) * 2 + k] = v;
        }
      }
    }
    __syncthreads();
    if constexpr (S::BWD) {      // SyncBatchNorm backward: the last block exchanges the totals (crog_gemm_desc.stat_sync, comm_dev.h)
      if (p.stat_sync) crog_stat_sync_tail(reinterpret_cast<const CrogSyncBlock*>(p.stat_sync), p.col_stats, p.stat_replicas, 2 * p.N, gridDim.x * gridDim.y);
    }
  }

  const T* R = LEAN ? nullptr : reinterpret_cast<const T*>(p.R);
  if (!LEAN && (p.out_mode != CROG_OUT_T || sizeof(T) == 4)) {
    // direct stores from the accumulator layout: a register covers 2 rows x 32 consecutive columns.  One 32 x 32 block at a time
    // (scheduling barrier between blocks) with one base pointer per block: the compiler otherwise forms all WM * WN * 16 64-bit
    // addresses first and spills the accumulators to make room.
    const bool interior = m0 + BM <= p.M && n0 + BN <= p.N;     // block-uniform: no guards inside the matrix
    auto row_of = [](int e) { return (e & 3) + 8 * (e >> 2); };
    // acc element = body(acc element, base + row offset) for every element of the tile, one 32 x 32 block at a time
    auto for_each_elem = [&](auto guarded, auto* base0, int64_t ld, auto&& body) {
#pragma unroll
      for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++) {
          const int mb = m0 + (wr * WM + i) * 32 + 4 * h;
          auto* cb = base0 + (int64_t)mb * ld + ncol[j];
#pragma unroll
          for (int e = 0; e < 16; e++)
            if (!decltype(guarded)::value || (mb + row_of(e) < p.M && ncol[j] < p.N)) acc[i][j][e] = body(acc[i][j][e], cb + (int64_t)row_of(e) * ld);
          __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto all_elems = [&](auto* base0, int64_t ld, auto&& body) {
      if (interior) for_each_elem(std::false_type{}, base0, ld, body);
      else for_each_elem(std::true_type{}, base0, ld, body);
    };
    if (R) all_elems(R, p.ldr, [](float a, const T* r) { return a + Elem<T>::to_f(*r); });
    if (p.act == CROG_ACT_RELU_POST) {
      per_block([&](int i, int j) {
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][j][e] = fmaxf(acc[i][j][e], 0.f);
      });
    }
    if (p.out_mode == CROG_OUT_F32_ATOMIC) {
      if (p.debug & 32) return;          // timing-only ablation: what the atomic adds cost
      all_elems(reinterpret_cast<float*>(p.C) + coff, p.ldc, [](float a, float* c) { atomicAdd(c, a); return a; });
    } else if (p.out_mode == CROG_OUT_F32) {
      all_elems(reinterpret_cast<float*>(p.C) + coff, p.ldc, [](float a, float* c) { *c = a; return a; });
    } else {
      all_elems(reinterpret_cast<T*>(p.C) + coff, p.ldc, [](float a, T* c) { *c = Elem<T>::from_f(a); return a; });
    }
  } else {
    // 2-byte output: each wave stages 32 x (32*WN) of its tile in a private LDS region, then adds the residual
    // and stores 16 bytes per lane (full 64/128-byte row segments)
    if constexpr (sizeof(T) == 2) {
      constexpr int CW = 32 * WN, CROW = CW + 8, VPR = CW / 8, RPP = 64 / VPR;  // vectors per row, rows per pass
      T* Cs = reinterpret_cast<T*>(smem) + wave * 32 * CROW;
      T* C = reinterpret_cast<T*>(p.C) + coff;
      // (crog_gemm has checked C, R, ldc, ldr for 16-byte alignment: an even N is all the 4-byte pairs need)
      if (LEAN || ((p.N & 1) == 0 && !(p.debug & 4) && !(R && (p.debug & 8)))) {      // debug bits 2 / 3: staged path for all / for residual launches
        // Skip the LDS transpose.  Neighbouring lanes hold neighbouring columns of the same rows, so one DPP swap per register
        // gives every lane two adjacent columns: even lanes store the pair of row(e), odd lanes the pair of row(e + 1) — 32
        // four-byte stores per 32 x 32 block and lane instead of 64 ds_write_b16 + barriers + ds_read_b128 + 16-byte stores.
        // (The fp32-output path, which stores straight from the accumulators, moves TWICE the bytes of the staged bf16 path in
        // 1.2x its time on the large-M 1x1 layers: the staging, not HBM, bounded them.)  The residual, when there is one, is
        // fetched in the same pair layout, a wave-row's 8 * WN loads in flight before the first is used, and added in fp32
        // (one rounding; the staged path rounds the product first, as a separate bf16 add would).
        const bool odd = lane & 1;
        const int64_t ldc = p.ldc, ldr = p.ldr;
        auto store_pairs = [&](auto guarded, auto with_res, auto relu_post) {
          constexpr bool G = decltype(guarded)::value, WR = decltype(with_res)::value, POST = decltype(relu_post)::value;
#pragma unroll
          for (int i = 0; i < WM; i++) {
            const int mrow = m0 + (wr * WM + i) * 32 + 4 * h + (odd ? 1 : 0);     // row of register 0 (even lanes) / 1 (odd lanes)
            bf16x2 rv[WR ? WN : 1][8];
            if constexpr (WR) {
#pragma unroll
              for (int j = 0; j < WN; j++) {
                const int col = n0 + (wc * WN + j) * 32 + (r & ~1);
                const T* rbase = R + (int64_t)mrow * ldr + col;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                  const int ro = (2 * q & 3) + 8 * (2 * q >> 2);
                  if (!G || (mrow + ro < p.M && col < p.N)) rv[j][q] = *reinterpret_cast<const bf16x2*>(rbase + ro * ldr);
                  else rv[j][q] = bf16x2{(bf16)0.f, (bf16)0.f};
                }
              }
            }
#pragma unroll
            for (int j = 0; j < WN; j++) {
              const int col = n0 + (wc * WN + j) * 32 + (r & ~1);
              T* base = C + (int64_t)mrow * ldc + col;
#pragma unroll
              for (int q = 0; q < 8; q++) {
                const float a0 = acc[i][j][2 * q], a1 = acc[i][j][2 * q + 1];
                const float b0 = dpp_swap1(a0), b1 = dpp_swap1(a1);     // the neighbour lane's values (lane ^ 1)
                float lo = odd ? b1 : a0, hi = odd ? a1 : b0;
                if constexpr (WR) { lo += (float)rv[j][q][0]; hi += (float)rv[j][q][1]; }
                if constexpr (POST) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }      // CROG_ACT_RELU_POST: the ReLU after the residual
                bf16x2 v;
                v[0] = (bf16)lo;
                v[1] = (bf16)hi;
                const int ro = (2 * q & 3) + 8 * (2 * q >> 2);          // row offset of register 2q: 0, 2, 8, 10, 16, 18, 24, 26
                if (!G || (mrow + ro < p.M && col < p.N)) *reinterpret_cast<bf16x2*>(base + ro * ldc) = v;
              }
              __builtin_amdgcn_sched_barrier(0);      // keep one 32 x 32 block's temporaries live at a time
            }
          }
        };
        const bool interior = m0 + BM <= p.M && n0 + BN <= p.N;     // block-uniform: no guards inside the matrix
        if constexpr (!LEAN) {
          if (p.act == CROG_ACT_RELU_POST) {      // block-uniform; its own instantiations so that the common stores carry no extra max
            if (R) { if (interior) store_pairs(std::false_type{}, std::true_type{}, std::true_type{}); else store_pairs(std::true_type{}, std::true_type{}, std::true_type{}); }
            else   { if (interior) store_pairs(std::false_type{}, std::false_type{}, std::true_type{}); else store_pairs(std::true_type{}, std::false_type{}, std::true_type{}); }
            return;
          }
        }
        if (R) { if (interior) store_pairs(std::false_type{}, std::true_type{}, std::false_type{}); else store_pairs(std::true_type{}, std::true_type{}, std::false_type{}); }
        else   { if (interior) store_pairs(std::false_type{}, std::false_type{}, std::false_type{}); else store_pairs(std::true_type{}, std::false_type{}, std::false_type{}); }
        return;
      }
#pragma unroll
      for (int i = 0; i < WM; i++) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
          for (int e = 0; e < 16; e++) Cs[((e & 3) + 8 * (e >> 2) + 4 * h) * CROW + j * 32 + r] = (T)acc[i][j][e];
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 32 / RPP; ps++) {
          const int lr = ps * RPP + lane / VPR, cv = (lane % VPR) * 8;
          const int m = m0 + (wr * WM + i) * 32 + lr, n = n0 + wc * CW + cv;
          if (m < p.M && n < p.N) {
            Vec16<T> o = *reinterpret_cast<const Vec16<T>*>(Cs + lr * CROW + cv);
            if (n + 8 <= p.N) {
              if (R) {
                Vec16<T> rv = ldg16(R + (int64_t)m * p.ldr + n);
#pragma unroll
                for (int e = 0; e < 8; e++) o.v[e] = (T)((float)o.v[e] + (float)rv.v[e]);
              }
              if (p.act == CROG_ACT_RELU_POST) {
#pragma unroll
                for (int e = 0; e < 8; e++) o.v[e] = (T)fmaxf((float)o.v[e], 0.f);
              }
              stg16(C + (int64_t)m * p.ldc + n, o);
            } else {
#pragma unroll
              for (int e = 0; e < 8; e++)      // constant trip count: a data-dependent bound would put `o` in scratch memory
                if (n + e < p.N) {
                  float f = (float)o.v[e];
                  if (R) f += (float)R[(int64_t)m * p.ldr + n + e];
                  if (p.act == CROG_ACT_RELU_POST) f = fmaxf(f, 0.f);
                  C[(int64_t)m * p.ldc + n + e] = (T)f;
                }
            }
          }
        }
      }
    }
  }
}


// (tile order across the 8 XCDs: xcd_map, gemm_dma.h)
inline bool splitk_by_xcd(const crog_gemm_desc& d) { return d.batch == 1 && d.splitk > 1; }

template <typename T, int AL, int BL, bool HWTR, typename S>
__global__ void __launch_bounds__(S::NT, (S::WM >= 4 ? 2 : (S::WM == 2 ? (sizeof(T) == 4 ? 2 : 3) : (sizeof(T) == 4 ? 3 : 4)))) gemm_kernel(const crog_gemm_desc p) {
  using Cfg = TileCfg<T>;
  constexpr int BK = Cfg::BK, NT = S::NT, BM = S::BM, BN = S::BN, WM = S::WM, WN = S::WN, WVN = S::WVN, WVM = S::WVM;
  using ALd = typename ALoaderSel<T, AL, BM, NT>::type;
  using BLd = typename BLoaderSel<T, BL, BN, NT>::type;
  constexpr bool ATR = ALoaderSel<T, AL, BM, NT>::TR, BTR = BLoaderSel<T, BL, BN, NT>::TR;
  constexpr int OPA = op_bytes<T, BM>(), OPB = op_bytes<T, BN>();
  constexpr int AROW = BM + Cfg::TR_PAD, BROW = BN + Cfg::TR_PAD;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WVN, wc = wave % WVN;
  const int r = lane & 31, h = lane >> 5;

  // XCD-aware tile order: blocks that share an XCD (id % 8) get a contiguous run of tiles.
  const int tilesN = (p.N + BN - 1) / BN, tilesM = (p.M + BM - 1) / BM;
  const int nwg = tilesM * tilesN;
  int id = blockIdx.x, z = blockIdx.y;
  xcd_map(nwg, p.splitk, id, z);
  const int tm = id / tilesN, tn = id % tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  const int zb = z / p.splitk, zs = z % p.splitk;
  const int zo = zb / p.batch_inner, zi = zb % p.batch_inner;
  const T* A = reinterpret_cast<const T*>(p.A) + zo * p.sAo + zi * p.sAi;
  const T* B = reinterpret_cast<const T*>(p.B) + zo * p.sBo + zi * p.sBi;
  // (split-K with plain fp32 stores: slice zs writes slab zs of a [splitk][M][ldc] workspace, crog_splitk_reduce sums them)
  const int64_t coff = zo * p.sCo + zi * p.sCi + ((p.out_mode == CROG_OUT_F32 && p.splitk > 1) ? (int64_t)zs * p.M * p.ldc : 0);

  const int ktiles = (p.K + BK - 1) / BK;
  const int per = (ktiles + p.splitk - 1) / p.splitk;
  const int kt0 = zs * per;
  const int kt1 = min(kt0 + per, ktiles);
  // an empty trailing slice (the split does not divide the k-tiles) has nothing to add - except in the slab form, where its slab must
  // hold zeros for crog_splitk_reduce: it then runs the loop zero times and stores its all-zero accumulators
  if (kt0 >= kt1 && !(p.out_mode == CROG_OUT_F32 && p.splitk > 1)) return;

  const ConvGeom g{p.convH, p.convW, p.convC};
  ALd la;
  BLd lb;
  la.init(A, p.lda, p.M, p.K, m0, g, tid);
  lb.init(B, p.ldb, p.N, p.K, n0, g, tid);

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
  constexpr bool KBLOCK = sizeof(T) == 4;
  f32x16 tot[KBLOCK ? WM : 1][KBLOCK ? WN : 1];
  if constexpr (KBLOCK) {
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
      for (int j = 0; j < WN; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) tot[i][j][e] = 0.f;
  }

  // Reduction order.  For the 3x3 forward / data-gradient kernels the k-tiles are visited channel-chunk-major, tap-minor
  // (the 9 taps of one 32-channel chunk back to back): the 9 shifted reads of the same pixels then hit L1/L2 instead of
  // being 9 separate sweeps of the map 16 k-tiles apart.  A and B use the same memory offset, so the sum is unchanged.
  auto kmem = [&](int kt) -> int {
    if constexpr (AL == CROG_A_IM2COL) {
      int tap, chunk;
      conv_ktile(kt, p.convC / BK, tap, chunk);
      return tap * p.convC + chunk * BK;
    } else {
      return kt * BK;
    }
  };
  // LDS ring: A slots at [0, 2*OPA), B slots at [2*OPA, 2*OPA + 2*OPB)
  la.load(kmem(kt0));
  lb.load(kmem(kt0));
  la.store(reinterpret_cast<T*>(smem));
  lb.store(reinterpret_cast<T*>(smem + 2 * OPA));
  __syncthreads();

  const bool do_asum = p.a_sum != nullptr && tn == 0 && wc == 0;   // block- and wave-uniform
  float asum[WM];
#pragma unroll
  for (int i = 0; i < WM; i++) asum[i] = 0.f;
  int cur = 0;
  for (int kt = kt0; kt < kt1; kt++) {
    const bool more = kt + 1 < kt1;
    if (more) {
      const int kn = kmem(kt + 1);
      la.load(kn);
      lb.load(kn);
    }
    const T* at = reinterpret_cast<const T*>(smem + cur * OPA);
    const T* bt = reinterpret_cast<const T*>(smem + 2 * OPA + cur * OPB);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      Frag<T> fa[WM], fb[WN];
#pragma unroll
      for (int i = 0; i < WM; i++) {
        if constexpr (ATR) fa[i] = frag_tr<HWTR, AROW>(at, (wr * WM + i) * 32, ks, lane);
        else fa[i] = frag_kc(at, (wr * WM + i) * 32 + r, ks, h);
      }
#pragma unroll
      for (int j = 0; j < WN; j++) {
        if constexpr (BTR) fb[j] = frag_tr<HWTR, BROW>(bt, (wc * WN + j) * 32, ks, lane);
        else fb[j] = frag_kc(bt, (wc * WN + j) * 32 + r, ks, h);
      }
      if (do_asum) {
#pragma unroll
        for (int i = 0; i < WM; i++) asum[i] += frag_sum(fa[i]);
      }
#pragma unroll
      for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++) mma16(fa[i], fb[j], acc[i][j]);
    }
    if (more) {
      la.store(reinterpret_cast<T*>(smem + (cur ^ 1) * OPA));
      lb.store(reinterpret_cast<T*>(smem + 2 * OPA + (cur ^ 1) * OPB));
    }
    if constexpr (KBLOCK) {
      if (((kt - kt0 + 1) % KBLOCK_F32) == 0) kblock_fold<WM, WN>(acc, tot);
    }
    __syncthreads();
    cur ^= 1;
  }
  if constexpr (KBLOCK) kblock_finish<WM, WN>(acc, tot);
  if (do_asum) flush_a_sum<WM>(asum, p.a_sum, m0 + wr * WM * 32, p.M, lane);

  gemm_epilogue<T, S>(acc, p, smem, m0, n0, zs, coff);
}


// =================================================================================================
// LDS-DMA variant of the 128x128 tile (all operand layouts).
// Tiles go global -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR staging, no ds_write), three LDS stages,
// two k-tiles always in flight behind a counted vmcnt, one raw s_barrier per k-tile.  The LDS image of a tile is
// exactly what the DMA writes (wave-uniform base + lane*16, 1 KiB per wave-instruction), so bank conflicts are
// avoided by XOR-swizzling the 16-byte chunk index on the SOURCE address and again on the fragment read:
//   K-contiguous tile  [128 rows][64 B]:   chunk ^= (row >> 2) & 3            -> ds_read_b128 conflict-free
//   transposed tile    [BK rows][128 cols]: chunk ^= ((row&3)<<2)|((row>>2)&3) (bf16) -> ds_read_b64_tr_b16 conflict-free
// Out-of-range rows/columns, conv padding and ragged K are an out-of-bounds buffer offset: the hardware writes zeros.
// =================================================================================================
// LDS ring depth (k-tiles in flight = depth - 1); tile edge = 32 * (waves per block): 128 (4 waves) or 256 (8 waves); one
// operand tile = edge * 64 bytes.  128^2: 3 stages x 16 KiB, 3 blocks per CU (a 4-stage ring at 2 blocks per CU measured 10 %
// slower: occupancy beats depth).  256^2: one 8-wave block per CU owns all 256 VGPRs per lane (128 of them accumulators), so
// the ring is 4 x 32 KiB deep instead.
template <typename S> constexpr int dma_nstage() { return S::NSTAGE ? S::NSTAGE : (S::NT == 512 ? 4 : 3); }
// blocks per CU the kernel is compiled for (register budget): by tile size, or what a deep ring leaves room for in 160 KiB of LDS
template <typename S> constexpr int dma_blocks_per_cu() {
  if constexpr (S::NSTAGE != 0) { constexpr int fit = 160 * 1024 / (S::NSTAGE * (S::BM + S::BN) * 64); return fit < 1 ? 1 : (fit > 4 ? 4 : fit); }
  if constexpr (S::BWD) return 2;     // the gated-statistics epilogue needs a few registers more than 3 blocks per CU leave
  return S::NT == 512 ? 1 : (S::BM + S::BN > 256 ? 2 : (S::BM + S::BN <= 128 ? 4 : 3));
}

// XOR applied to the 16-byte chunk index of a K-contiguous tile row (64-byte rows: four rows per 256-byte bank row).
// VAR 0: fragments are read by v_mfma_f32_32x32x16 lanes (row = lane & 31, chunk = 2 ks + (lane >> 5)): chunk ^= (row >> 2) & 3.
// VAR 1 / 2: fragments are read by v_mfma_f32_16x16x32 lanes (row = lane & 15, chunk = lane >> 4).  ds_read_b128 is served in the
// lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32): a group holds the row quads Q = (row >> 2) & 3 of {0, 3} at one
// chunk and {1, 2} at the next, so the four quads need distinct f(Q0), f(Q3), f(Q1) ^ 1, f(Q2) ^ 1: f = {0, 2, 3, 1}.
template <int VAR>
__device__ inline int kc_swz(int row) {
  const int q = (row >> 2) & 3;
  return VAR == 0 ? q : ((0x78 >> (2 * q)) & 3);
}
// VAR 2: the tile's rows are stored de-interleaved by 4 inside every group of 64: LDS row 16 b + c holds tile row 4 c + b.  A wave's
// four 16-column MFMA blocks then hold columns 4 c + {0, 1, 2, 3} in lane c: four adjacent output columns per lane, 8-byte stores.
template <int VAR>
__device__ inline int kc_tile_row(int lds_row) {
  return VAR == 2 ? ((lds_row & ~63) | ((lds_row & 15) << 2) | ((lds_row >> 4) & 3)) : lds_row;
}

// K-contiguous operand (MODE 0 dense rows, MODE 1 im2col patches of an NHWC map)
template <typename T, int MODE, int NI, int VAR = 0>
struct DmaKc {
  static constexpr int VEC = TileCfg<T>::VEC, BK = TileCfg<T>::BK;
  static constexpr bool TR = false;
  unsigned base[NI];  // byte offset of (row, swizzled chunk) at k = 0; DMA_OOB when the row is out of range
  unsigned cb[NI];    // MODE 0: byte position of the lane's logical chunk inside the k-tile; MODE 1: 9-bit mask of the taps that land inside the image
  int ldb_;           // row stride in bytes
  unsigned kbytes;

  // The buffer resource is NOT a member: it is rebuilt from the (kernel-scope, wave-uniform) base pointer and extent at the point
  // of use, so that it is provably uniform and lives in SGPRs.  As a field of a loader that is modified inside the loop it came back
  // from private memory, i.e. formally divergent, and every LDS-DMA request was wrapped in a waterfall loop.
  __device__ __attribute__((always_inline)) static int extent(int64_t ld, int rows_total, int K, const ConvGeom&) {
    // num_records = the operand's true extent (rows_total rows of ld elements; the last row ends at K rounded up to a
    // chunk): anything past it reads as zero instead of touching a neighbouring allocation
    return (int)(((long)(rows_total - 1) * ld + ((K + VEC - 1) / VEC) * VEC) * (long)sizeof(T));
  }
  __device__ __attribute__((always_inline)) void init(const T* ptr, int64_t ld, int rows_total, int K, int row0, int wave, int lane, const ConvGeom& g) {
    ldb_ = (int)(ld * sizeof(T));
    kbytes = (unsigned)(K * sizeof(T));
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int row = (wave * NI + i) * 16 + (lane >> 2);       // row of the LDS image
      const unsigned chunk = (unsigned)(((lane & 3) ^ kc_swz<VAR>(row)) * 16);
      const long m = (long)row0 + kc_tile_row<VAR>(row);
      base[i] = (m < rows_total) ? (unsigned)(m * ld * sizeof(T)) + chunk : DMA_OOB;
      if constexpr (MODE == 1) {
        // the pixel's position decides once which of the 9 taps exist; per k-tile the test is one shift + select instead of
        // four compares under an exec mask (which also cut the loop body into basic blocks the scheduler cannot interleave)
        const long rem = m % ((long)g.H * g.W);
        const int py = (int)(rem / g.W), px = (int)(rem % g.W);
        unsigned mask = 0;
        if (m < rows_total) {
#pragma unroll
          for (int tap = 0; tap < 9; tap++) {
            const int sy = py + tap / 3 - 1, sx = px + tap % 3 - 1;
            if (sy >= 0 && sy < g.H && sx >= 0 && sx < g.W) mask |= 1u << tap;
          }
        }
        cb[i] = mask;
      } else {
        cb[i] = chunk;
      }
    }
  }
  __device__ __attribute__((always_inline)) void start(int, const ConvGeom&) {}
  __device__ __attribute__((always_inline)) void advance(const ConvGeom&) {}
  // kt: k-tile index; kmem: element offset of the k-tile inside the reduction (see kmem in the kernels); live: the tile exists
  // (block-uniform; a request past the end of the reduction range becomes an out-of-bounds offset = no traffic, zeros written)
  __device__ __attribute__((always_inline)) unsigned off(int i, int kt, int kmem, const ConvGeom& g, bool live) const {
    if constexpr (MODE == 1) {
      int tap, chunk;
      conv_ktile(kt, g.C / BK, tap, chunk);
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const unsigned shift = (unsigned)((dy * g.W + dx) * ldb_ + chunk * BK * (int)sizeof(T));   // block-uniform; wrap-around = negative shift
      // `live` enters as a mask, not as a condition: a block-uniform condition here is turned into a branch around the request,
      // which splits the loop body
      const unsigned lv = live ? ~0u : 0u;
      const bool ok = ((cb[i] & lv) >> tap) & 1u;
      return ok ? base[i] + shift : DMA_OOB;
    } else {
      const unsigned kb = (unsigned)(kmem * sizeof(T));
      const unsigned lim = live ? kbytes : 0u;
      const bool ok = (base[i] != DMA_OOB) & (kb + cb[i] < lim);     // '&': a short-circuit '&&' became an exec-masked branch that cut the loop body in two
      return ok ? base[i] + kb : DMA_OOB;
    }
  }
  __device__ static Frag<T> frag(const char* tile, int rbase, int ks, int lane);
};
template <typename T> struct DmaKcFrag;
template <> struct DmaKcFrag<bf16> {
  __device__ static Frag<bf16> get(const char* tile, int rbase, int ks, int lane) {
    const int row = rbase + (lane & 31), c = (ks * 2 + (lane >> 5)) ^ ((row >> 2) & 3);
    Frag<bf16> f;
    f.v = *reinterpret_cast<const bf16x8*>(tile + row * 64 + c * 16);
    return f;
  }
};
template <> struct DmaKcFrag<float> {
  __device__ static Frag<float> get(const char* tile, int rbase, int ks, int lane) {
    const int row = rbase + (lane & 31), h = lane >> 5, sw = (row >> 2) & 3;
    const f32x4 a = *reinterpret_cast<const f32x4*>(tile + row * 64 + ((2 * h) ^ sw) * 16);
    const f32x4 b = *reinterpret_cast<const f32x4*>(tile + row * 64 + ((2 * h + 1) ^ sw) * 16);
    Frag<float> f;
    f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
    f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
    return f;
  }
};
template <typename T, int MODE, int NI, int VAR>
__device__ inline Frag<T> DmaKc<T, MODE, NI, VAR>::frag(const char* tile, int rbase, int ks, int lane) { return DmaKcFrag<T>::get(tile, rbase, ks, lane); }
// fragment of v_mfma_f32_16x16x32_bf16: lane (c = lane & 15, g = lane >> 4) supplies row rbase + c, reduction indices 8 g .. 8 g + 7
__device__ inline bf16x8 frag16_kc(const char* tile, int rbase, int lane) {
  const int row = rbase + (lane & 15), c = (lane >> 4) ^ kc_swz<1>(row);
  return *reinterpret_cast<const bf16x8*>(tile + row * 64 + c * 16);
}

// Transposed operand: the tile is [BK reduction rows][128 columns] of memory (MODE as TrLoader: 0 dense, 1 dgrad weights, 2 wgrad im2col)
// XOR applied to the 16-byte chunk index of a transposed bf16 tile row (the four reduction rows a ds_read_b64_tr_b16 group
// touches must land in different quarters of the 256-byte bank row).  Rows of >= 256 bytes: quarter ^= row & 3.  128-byte rows
// (64-column tiles) alternate between the two halves of the bank row by themselves, so only rows r and r+2 collide: half ^= bit 1.
template <int COLS>
__device__ inline int tr_swz(int row) { return COLS >= 128 ? (((row & 3) << 2) | ((row >> 2) & 3)) : (((row >> 1) & 1) << 2); }

template <typename T, int MODE, int COLS, int NI>
struct DmaTr {
  static constexpr int VEC = TileCfg<T>::VEC, BK = TileCfg<T>::BK;
  static constexpr bool TR = true;
  static constexpr int CPR = COLS / VEC;         // chunks per tile row
  static constexpr int ROWB = COLS * (int)sizeof(T);
  static constexpr int RPI1024 = 1024 / ROWB;    // whole tile rows per 1 KiB wave-instruction (>= 1 for all shapes used)
  static_assert(RPI1024 >= 1, "a transposed tile row must fit one 1 KiB wave-instruction");
  int rowin[NI];       // reduction row of this lane inside the k-tile
  unsigned colb[NI];   // byte offset of the lane's logical column chunk (DMA_OOB when the columns are out of range)
  int tdy[NI], tdx[NI]; // MODE 2: tap of the lane's column chunk
  int px[NI], py[NI];   // MODE 2: image coordinates of the lane's reduction row (pixel), advanced k-tile by k-tile
  unsigned cur[NI];     // MODE 0 / 2: byte offset of the lane's chunk in the current k-tile, advanced by BK rows per k-tile (no multiply in the loop)
  int ldb_, K;
  int stepx, stepy;     // MODE 2: BK pixels = stepy rows (mod H) + stepx columns

  __device__ static int swz(int row) { return sizeof(T) == 2 ? tr_swz<COLS>(row) : 0; }

  __device__ __attribute__((always_inline)) static int extent(int64_t ld, int ncols, int K_, const ConvGeom& g) {
    long rows = K_;                                    // rows of memory the tile can touch
    if (MODE == 1) rows = 9L * g.C;                    // KRSC weight rows (co*9 + tap)
    const long width = (MODE == 2) ? g.C : ((ncols + VEC - 1) / VEC) * VEC;   // im2col rows are one pixel's channels
    return (int)(((rows - 1) * ld + width) * (long)sizeof(T));
  }
  __device__ __attribute__((always_inline)) void init(const T* ptr, int64_t ld, int ncols, int K_, int col0, int wave, int lane, const ConvGeom& g) {
    ldb_ = (int)(ld * sizeof(T));
    K = K_;
    stepx = stepy = 0;
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int row = (wave * NI + i) * RPI1024 + lane / CPR;
      const int c = (lane % CPR) ^ swz(row);
      rowin[i] = row;
      px[i] = py[i] = 0;
      cur[i] = 0;
      const int col = col0 + c * VEC;
      const bool oob = col >= ncols;      // selects, not an early `continue`: with divergent control flow here the compiler keeps the arrays in scratch memory
      if constexpr (MODE == 2) {
        const int tap = col / g.C;
        colb[i] = oob ? DMA_OOB : (unsigned)((col - tap * g.C) * sizeof(T));
        tdy[i] = oob ? 0 : tap / 3 - 1;
        tdx[i] = oob ? 0 : tap % 3 - 1;
      } else {
        colb[i] = oob ? DMA_OOB : (unsigned)(col * sizeof(T));
        tdy[i] = tdx[i] = 0;
      }
    }
  }
  // MODE 2: position of the lane's pixels at the first k-tile of this block's reduction range (the only divisions), and the
  // per-k-tile step; afterwards `advance` keeps (px, py) current with two adds and two selects per k-tile
  __device__ __attribute__((always_inline)) void start(int kmem0, const ConvGeom& g) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int i = 0; i < NI; i++) cur[i] = (unsigned)(kmem0 + rowin[i]) * (unsigned)ldb_ + colb[i];
    }
    if constexpr (MODE == 2) {
      stepx = BK % g.W;
      stepy = (BK / g.W) % g.H;
#pragma unroll
      for (int i = 0; i < NI; i++) {
        const int k = kmem0 + rowin[i];
        px[i] = k % g.W;
        py[i] = (k / g.W) % g.H;
        cur[i] = (unsigned)(k + tdy[i] * g.W + tdx[i]) * (unsigned)ldb_ + colb[i];   // wrap-around = negative shift
      }
    }
  }
  __device__ __attribute__((always_inline)) void advance(const ConvGeom& g) {
    if constexpr (MODE == 0 || MODE == 2) {
      const unsigned step = (unsigned)(BK * ldb_);
#pragma unroll
      for (int i = 0; i < NI; i++) cur[i] += step;
    }
    if constexpr (MODE == 2) {
#pragma unroll
      for (int i = 0; i < NI; i++) {
        int x = px[i] + stepx, y = py[i] + stepy;
        const bool wrap = x >= g.W;
        x = wrap ? x - g.W : x;
        y = wrap ? y + 1 : y;
        y = y >= g.H ? y - g.H : y;      // stepy < H and at most one carry: y < 2 H
        px[i] = x;
        py[i] = y;
      }
    }
  }
  __device__ __attribute__((always_inline)) unsigned off(int i, int kt, int kmem, const ConvGeom& g, bool live) const {
    const int k = kmem + rowin[i];
    const int klim = live ? K : 0;      // (a mask, not a condition: see DmaKc::off)
    if constexpr (MODE == 0) {
      const bool ok = (colb[i] != DMA_OOB) & (k < klim);
      return ok ? cur[i] : DMA_OOB;
    } else if constexpr (MODE == 1) {
      if (!live || colb[i] == DMA_OOB || k >= K) return DMA_OOB;
      const int tap = k / g.C, co = k - tap * g.C;
      return (unsigned)(co * 9 + (8 - tap)) * (unsigned)ldb_ + colb[i];
    } else {
      const int sy = py[i] + tdy[i], sx = px[i] + tdx[i];
      const bool ok = (colb[i] != DMA_OOB) & (k < klim) & ((unsigned)sy < (unsigned)g.H) & ((unsigned)sx < (unsigned)g.W);
      return ok ? cur[i] : DMA_OOB;
    }
  }
  __device__ static Frag<T> frag(const char* tile, int cbase, int ks, int lane);
};
template <int COLS> struct DmaTrFrag {
  __device__ static Frag<bf16> get(const char* tile, int cbase, int ks, int lane) {
    // ds_read_b64_tr_b16 on the swizzled image: lane 4q+p of each 16-lane group addresses row q, columns 4p..4p+3
    constexpr int ROWB = COLS * 2;
    const int h = lane >> 5, i = lane & 15, q = i >> 2, pp = i & 3;
    const int col = cbase + 16 * ((lane >> 4) & 1) + 4 * pp;
    const int chunk = col >> 3, inb = (col & 7) * 2;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int r0 = ks * 16 + 8 * h + q, r1 = r0 + 4;
    const int x0 = tr_swz<COLS>(r0), x1 = tr_swz<COLS>(r1);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + r0 * ROWB + ((chunk ^ x0) << 4) + inb));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + r1 * ROWB + ((chunk ^ x1) << 4) + inb));
    Frag<bf16> f;
    f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return f;
  }
  __device__ static Frag<float> getf(const char* tile, int cbase, int ks, int lane) {
    const int h = lane >> 5, col = cbase + (lane & 31);
    const float* t = reinterpret_cast<const float*>(tile);
    Frag<float> f;
#pragma unroll
    for (int j = 0; j < 8; j++) f.v[j] = t[(8 * h + j) * COLS + col];
    return f;
  }
};
template <typename T, int MODE, int COLS> struct DmaTrFragSel;
template <int MODE, int COLS> struct DmaTrFragSel<bf16, MODE, COLS> {
  __device__ static Frag<bf16> get(const char* t, int c, int ks, int l) { return DmaTrFrag<COLS>::get(t, c, ks, l); }
};
template <int MODE, int COLS> struct DmaTrFragSel<float, MODE, COLS> {
  __device__ static Frag<float> get(const char* t, int c, int ks, int l) { return DmaTrFrag<COLS>::getf(t, c, ks, l); }
};
template <typename T, int MODE, int COLS, int NI>
__device__ inline Frag<T> DmaTr<T, MODE, COLS, NI>::frag(const char* t, int c, int ks, int l) { return DmaTrFragSel<T, MODE, COLS>::get(t, c, ks, l); }

template <typename T, int AL, int EDGE, int NI> struct DmaASel;
template <typename T, int E, int NI> struct DmaASel<T, CROG_A_KC, E, NI> { using type = DmaKc<T, 0, NI>; };
template <typename T, int E, int NI> struct DmaASel<T, CROG_A_IM2COL, E, NI> { using type = DmaKc<T, 1, NI>; };
template <typename T, int E, int NI> struct DmaASel<T, CROG_A_MC, E, NI> { using type = DmaTr<T, 0, E, NI>; };
template <typename T, int BL, int EDGE, int NI> struct DmaBSel;
template <typename T, int E, int NI> struct DmaBSel<T, CROG_B_KC, E, NI> { using type = DmaKc<T, 0, NI>; };
template <typename T, int E, int NI> struct DmaBSel<T, CROG_B_NC, E, NI> { using type = DmaTr<T, 0, E, NI>; };
template <typename T, int E, int NI> struct DmaBSel<T, CROG_B_NC_DGRAD, E, NI> { using type = DmaTr<T, 1, E, NI>; };
template <typename T, int E, int NI> struct DmaBSel<T, CROG_B_NC_IM2COL, E, NI> { using type = DmaTr<T, 2, E, NI>; };

// Issue the LDS-DMA loads of one k-tile into ring stage `stage`: every wave moves NIA 1-KiB slices of the A tile and NIB of
// the B tile.  A macro on purpose: as a lambda it makes hipcc's HOST pass drop the kernel stub silently, and as a function template
// the host pass rejects its second instantiation context ("substitution failure") while the device pass accepts it.
// `live`: the tile exists (block-uniform); a request past the end of the reduction range is an out-of-bounds offset.
// The buffer resources are rebuilt from the kernel-scope base pointers here, so they are provably wave-uniform (SGPRs).
// (one 1-KiB request: dma_piece, gemm_dma.h)
#define CROG_DMA_ISSUE(KT, KMEM, STAGE, LIVE)                                                                          \
  do {                                                                                                                 \
    const int kt_ = (KT), km_ = (KMEM);                                                                                \
    const bool lv_ = (LIVE);                                                                                           \
    char* sa_ = smem + (STAGE) * (TILE_A_B + TILE_B_B) + wave * NIA * 1024;                                            \
    char* sb_ = smem + (STAGE) * (TILE_A_B + TILE_B_B) + TILE_A_B + wave * NIB * 1024;                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < NIA; i_++) dma_piece(A, exa, sa_ + i_ * 1024, da.off(i_, kt_, km_, g, lv_)); \
    _Pragma("unroll") for (int i_ = 0; i_ < NIB; i_++) dma_piece(B, exb, sb_ + i_ * 1024, db.off(i_, kt_, km_, g, lv_)); \
    da.advance(g); /* k-tiles are requested strictly in order, one call per tile */                                    \
    db.advance(g);                                                                                                     \
  } while (0)

// Wait until at most `behind` of this wave's most recently requested k-tiles (P DMA instructions each) are still in flight.
// MAXB = the deepest value `behind` can take (ring depth - 1): deeper cases are not instantiated.
template <int P, int MAXB>
__device__ __attribute__((always_inline)) inline void wait_tiles(int behind) {
  if constexpr (MAXB >= 3) { if (behind >= 3) { wait_vmcnt<3 * P>(); return; } }
  if constexpr (MAXB >= 2) { if (behind == 2) { wait_vmcnt<2 * P>(); return; } }
  if (behind == 1) wait_vmcnt<P>();
  else wait_vmcnt<0>();
}

// Tile shapes of the LDS-DMA kernel.  Measured and dropped (rounds 1-2): 256 x 128 with 8 waves of 64 x 64 at one block per CU
// (barrier stalls are not hidden by a second block: 5-30 % slower than 128 x 128 at three blocks per CU), 128 x 256 with 4 waves of
// 64 x 128 (20-45 % slower), 64 x 256 for Cout <= 64 weight gradients (two blocks per CU, 20 KiB per k-tile: 14 % slower).
using ShapeDma8 = Shape<4, 2, 2, 4, true>;   // 256 x 256, 8 waves, 128 accumulator registers per lane (lean epilogue: at the 256-VGPR limit the full one spills)
using ShapeDma8A = Shape<4, 2, 2, 4, true, 0, false, true>;   // the same tile with the atomic-only epilogue: large 3x3 weight gradients
#ifndef CROG_TALL_NSTAGE
#define CROG_TALL_NSTAGE 0             // (A/B builds: depth of the 256 x 64 tile's LDS-DMA ring; 0 = the default of its size)
#endif
using ShapeTall = Shape<2, 2, 4, 1, false, CROG_TALL_NSTAGE>;   // 256 x  64, 4 waves: layers with <= 64 output columns (N = 32 / 64)
#ifndef CROG_DMA64_NSTAGE
#define CROG_DMA64_NSTAGE 0            // (A/B builds: depth of the 64 x 64 tile's LDS-DMA ring; 0 = the default of its size, 3)
#endif
using ShapeDma64 = Shape<1, 1, 2, 2, false, CROG_DMA64_NSTAGE>;  //  64 x  64, 4 waves: small GEMMs (text tower, attention pooling), no BN statistics
using ShapeMidBwd = Shape<2, 2, 2, 2, true, 0, true>;   // 128 x 128 data gradient that does the consumer BatchNorm's first backward pass
// (8-deep rings for launches of <= 1-2 blocks per CU were tried: no gain standalone -- those launches are not bound by request
// latency -- and 3 % slower in the step, where a 128 KiB block keeps the other streams' blocks off the CU.)

// ASUM: the launch also accumulates a_sum[m] += sum_k A(m, k) (bias gradient inside a weight-gradient GEMM); a template flag so
// that the main loop of every other launch is one basic block
// The FORWARD statistics' exchange in the tail of the kernel that accumulated them (SyncBatchNorm: crog_gemm_desc.stat_sync with col_stats in
// replica rows, no bwd_z), called by every block after its last store - no accumulator is live any more.  Round 6: the LDS-DMA tiles carry it
// like the ping-pong tile, so crog_gemm adds no single-block finish launch behind them (27 of the 33 that were left per CROG-R50 step).
__device__ __attribute__((always_inline)) inline void fwd_stat_sync_tail(const crog_gemm_desc& p) {
  if (p.stat_sync && p.col_stats && p.stat_replicas > 0 && !p.bwd_z && p.splitk <= 1 && p.batch == 1)
    crog_stat_sync_tail(reinterpret_cast<const CrogSyncBlock*>(p.stat_sync), p.col_stats, p.stat_replicas, 2 * p.N, gridDim.x * gridDim.y * gridDim.z);
}

template <typename T, int AL, int BL, typename S, int ASUM>
__global__ void __launch_bounds__(S::NT, (sizeof(T) == 4 && dma_blocks_per_cu<S>() > 2 ? dma_blocks_per_cu<S>() - 1 : dma_blocks_per_cu<S>()))
gemm_dma_kernel(const crog_gemm_desc p) {   // (fp32 carries a second accumulator set, KBLOCK_F32: one block per CU less)
  constexpr int DMA_NSTAGE = dma_nstage<S>();
  constexpr int NW = S::NT / 64, NIA = S::BM / (16 * NW), NIB = S::BN / (16 * NW);   // 1-KiB DMA slices per wave and k-tile
  static_assert(NIA >= 1 && NIB >= 1 && NIA * 16 * NW == S::BM && NIB * 16 * NW == S::BN, "every wave must move whole 1 KiB slices of both tiles");
  using OA = typename DmaASel<T, AL, S::BM, NIA>::type;
  using OB = typename DmaBSel<T, BL, S::BN, NIB>::type;
  constexpr int BK = TileCfg<T>::BK, BM = S::BM, BN = S::BN, WM = S::WM, WN = S::WN;
  constexpr int TILE_A_B = BM * 64, TILE_B_B = BN * 64, DMA_STAGE_B = TILE_A_B + TILE_B_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / S::WVN, wc = wave % S::WVN;

  const int tilesN = (p.N + BN - 1) / BN, tilesM = (p.M + BM - 1) / BM;
  const int nwg = tilesM * tilesN;
  int id = blockIdx.x, z = blockIdx.y;
  xcd_map(nwg, p.splitk, id, z);
  const int tm = id / tilesN, tn = id % tilesN;     // (row tiles fastest instead: measured equal on the 3x3 shapes, scripts/bench_conv_n.py)
  const int m0 = tm * BM, n0 = tn * BN;
  const int zb = z / p.splitk, zs = z % p.splitk;
  const int zo = zb / p.batch_inner, zi = zb % p.batch_inner;
  const T* A = reinterpret_cast<const T*>(p.A) + zo * p.sAo + zi * p.sAi;
  const T* B = reinterpret_cast<const T*>(p.B) + zo * p.sBo + zi * p.sBi;
  // (split-K with plain fp32 stores: slice zs writes slab zs of a [splitk][M][ldc] workspace, crog_splitk_reduce sums them)
  const int64_t coff = zo * p.sCo + zi * p.sCi + ((p.out_mode == CROG_OUT_F32 && p.splitk > 1) ? (int64_t)zs * p.M * p.ldc : 0);

  const int ktiles = (p.K + BK - 1) / BK;
  const int per = (ktiles + p.splitk - 1) / p.splitk;
  const int kt0 = zs * per;
  const int kt1 = min(kt0 + per, ktiles);
  // an empty trailing slice (the split does not divide the k-tiles) has nothing to add - except in the slab form, where its slab must
  // hold zeros for crog_splitk_reduce: it then runs the loop zero times and stores its all-zero accumulators
  if (kt0 >= kt1 && !(p.out_mode == CROG_OUT_F32 && p.splitk > 1)) return;
  const int nt = kt1 - kt0;

  const ConvGeom g{p.convH, p.convW, p.convC};
  OA da;
  OB db;
  da.init(A, p.lda, p.M, p.K, m0, wave, lane, g);
  db.init(B, p.ldb, p.N, p.K, n0, wave, lane, g);
  const int exa = OA::extent(p.lda, p.M, p.K, g), exb = OB::extent(p.ldb, p.N, p.K, g);

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; i++)
#pragma unroll
    for (int j = 0; j < WN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
  constexpr bool KBLOCK = sizeof(T) == 4;     // fp32: k-blocked accumulation (KBLOCK_F32)
  f32x16 tot[KBLOCK ? WM : 1][KBLOCK ? WN : 1];
  if constexpr (KBLOCK) {
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
      for (int j = 0; j < WN; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) tot[i][j][e] = 0.f;
  }

  // channel-chunk-major, tap-minor reduction order for the 3x3 forward / data-gradient forms (see gemm_kernel)
  auto kmem_of = [&](int kt) -> int {   // no builtins inside: safe for the host pass (see dma_issue)
    if constexpr (AL == CROG_A_IM2COL) {
      int tap, chunk;
      conv_ktile(kt, p.convC / BK, tap, chunk);
      return tap * p.convC + chunk * BK;
    } else {
      return kt * BK;
    }
  };
#define CROG_KMEM(kt) kmem_of(kt)
  const bool do_asum = ASUM != 0 && p.a_sum != nullptr && tn == 0 && wc == 0;   // block- and wave-uniform
  float asum[WM];
#pragma unroll
  for (int i = 0; i < WM; i++) asum[i] = 0.f;
  constexpr int PER_TILE = NIA + NIB;   // DMA instructions per k-tile and wave
  da.start(kt0 * BK, g);
  db.start(kt0 * BK, g);
  if constexpr (BK / 16 == 2) {
    // Software-pipelined, branch-free main loop (two 16-deep MFMA steps per k-tile).  Exactly DEPTH k-tiles are always requested
    // ahead — a request past the end of this block's reduction range is an out-of-bounds offset (no traffic, zeros into a stage
    // nobody reads) — so the vmcnt waits are compile-time constants and the loop body is ONE basic block: the DMA requests of tile
    // t + DEPTH, the LDS reads of tile t + 1 and the second MFMA group of tile t can be interleaved instruction by instruction
    // (an LDS-DMA request holds the wave's issue slot for 60-180 cycles; issued in a bunch by all waves right after the barrier, as
    // before, the matrix pipe of every SIMD idled through that phase: the 256 x 256 tile ran at a third of its MFMA rate).
#pragma unroll
    for (int s = 0; s < DMA_NSTAGE; s++)
      CROG_DMA_ISSUE(kt0 + s, CROG_KMEM(kt0 + s), s, s < nt);
    wait_vmcnt<(DMA_NSTAGE - 1) * PER_TILE>();     // tile 0 has landed; depth - 1 tiles stay in flight
    __builtin_amdgcn_s_barrier();
    Frag<T> fa0[WM], fb0[WN], fa1[WM], fb1[WN];
#pragma unroll
    for (int i = 0; i < WM; i++) fa0[i] = OA::frag(smem, (wr * WM + i) * 32, 0, lane);
#pragma unroll
    for (int j = 0; j < WN; j++) fb0[j] = OB::frag(smem + TILE_A_B, (wc * WN + j) * 32, 0, lane);
    int stage = 0;
    for (int t = 0; t < nt; t++) {
      const char* at = smem + stage * DMA_STAGE_B;
      const char* bt = at + TILE_A_B;
#pragma unroll
      for (int i = 0; i < WM; i++) fa1[i] = OA::frag(at, (wr * WM + i) * 32, 1, lane);
#pragma unroll
      for (int j = 0; j < WN; j++) fb1[j] = OB::frag(bt, (wc * WN + j) * 32, 1, lane);
      if constexpr (ASUM) {
        if (do_asum) {
#pragma unroll
          for (int i = 0; i < WM; i++) asum[i] += frag_sum(fa0[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++) mma16(fa0[i], fb0[j], acc[i][j]);
      const int nstage = stage == DMA_NSTAGE - 1 ? 0 : stage + 1;
      // this wave's reads of tile t are complete (its stage is about to be handed back to the DMA engine by any wave), tile t + 1 has
      // landed for this wave, and after the barrier for every wave; tiles t + 2 .. t + DEPTH - 1 stay in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wait_vmcnt<(DMA_NSTAGE - 2) * PER_TILE>();
      __builtin_amdgcn_s_barrier();
      {
        const int tq = t + DMA_NSTAGE;
        CROG_DMA_ISSUE(kt0 + tq, CROG_KMEM(kt0 + tq), stage, tq < nt);
      }
      const char* an = smem + nstage * DMA_STAGE_B;
#pragma unroll
      for (int i = 0; i < WM; i++) fa0[i] = OA::frag(an, (wr * WM + i) * 32, 0, lane);
#pragma unroll
      for (int j = 0; j < WN; j++) fb0[j] = OB::frag(an + TILE_A_B, (wc * WN + j) * 32, 0, lane);
      if constexpr (ASUM) {
        if (do_asum) {
#pragma unroll
          for (int i = 0; i < WM; i++) asum[i] += frag_sum(fa1[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++) mma16(fa1[i], fb1[j], acc[i][j]);
      if constexpr (sizeof(T) == 2 && !ASUM) {
        // order of the second half of the k-tile: the next tile's fragment reads first (they complete under the MFMAs), then the
        // MFMAs with the DMA requests spread between them
        constexpr int NMF = WM * WN, MPD = NMF / PER_TILE > 0 ? NMF / PER_TILE : 1;
        constexpr int NDS = WM * (OA::TR ? 2 : 1) + WN * (OB::TR ? 2 : 1);     // a transposed fragment is two ds_read_b64_tr_b16
        __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
#pragma unroll
        for (int q = 0; q < PER_TILE; q++) {
          if (q * MPD < NMF) __builtin_amdgcn_sched_group_barrier(0x008, MPD, 0);
          __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
        if (PER_TILE * MPD < NMF) __builtin_amdgcn_sched_group_barrier(0x008, NMF - PER_TILE * MPD, 0);
      }
      stage = nstage;
    }
    // The trailing out-of-range requests write zeros into the ring, and the epilogue reuses the ring (statistics exchange, staged
    // stores): EVERY wave's requests must have landed before ANY wave writes there — its own wait is not enough (a slower wave's
    // zeros landed on a faster wave's column sums: BatchNorm statistics, and with them the whole forward, differed from run to run
    // by ~2 % of the layer-2 output until this barrier was added).
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
  } else {
#pragma unroll
  for (int s = 0; s < DMA_NSTAGE - 1; s++)
    if (s < nt) CROG_DMA_ISSUE(kt0 + s, CROG_KMEM(kt0 + s), s, true);
  int stage = 0;
  for (int t = 0; t < nt; t++) {
    // k-tiles still allowed in flight behind tile t: min(depth - 2, tiles left); 4 DMA instructions per tile and wave
    wait_tiles<NIA + NIB, DMA_NSTAGE - 2>(min(DMA_NSTAGE - 2, nt - 1 - t));
    __builtin_amdgcn_s_barrier();
    if (t + DMA_NSTAGE - 1 < nt && !(p.debug & 1)) {
      const int tn = t + DMA_NSTAGE - 1;
      CROG_DMA_ISSUE(kt0 + tn, CROG_KMEM(kt0 + tn), stage == 0 ? DMA_NSTAGE - 1 : stage - 1, true);
    }
    const char* at = smem + stage * DMA_STAGE_B;
    const char* bt = at + TILE_A_B;
    if (!(p.debug & 2))
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      Frag<T> fa[WM], fb[WN];
#pragma unroll
      for (int i = 0; i < WM; i++) fa[i] = OA::frag(at, (wr * WM + i) * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < WN; j++) fb[j] = OB::frag(bt, (wc * WN + j) * 32, ks, lane);
      if (do_asum) {
#pragma unroll
        for (int i = 0; i < WM; i++) asum[i] += frag_sum(fa[i]);
      }
#pragma unroll
      for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++) mma16(fa[i], fb[j], acc[i][j]);
    }
    if constexpr (KBLOCK) {
      if (((t + 1) % KBLOCK_F32) == 0) kblock_fold<WM, WN>(acc, tot);
    }
    stage = stage == DMA_NSTAGE - 1 ? 0 : stage + 1;
  }
  if constexpr (KBLOCK) kblock_finish<WM, WN>(acc, tot);
  }
#undef CROG_KMEM
  if (do_asum) flush_a_sum<WM>(asum, p.a_sum, m0 + wr * WM * 32, p.M, lane);
  __syncthreads();
  gemm_epilogue<T, S>(acc, p, smem, m0, n0, zs, coff);
  if constexpr (!S::BWD && !S::ATOM && sizeof(T) == 2) fwd_stat_sync_tail(p);
}


// =================================================================================================
// v_mfma_f32_16x16x32_bf16 variant of the LDS-DMA kernel: bf16, both operands K-contiguous (1x1 / linear / 3x3 implicit-GEMM forward
// and the data gradients on transposed weight copies), lean epilogue (BatchNorm statistics + bf16 stores).
// Why a second MFMA shape: the large launches are power-bound, not issue-bound (the GEMM loop holds ~1.97 GHz at the
// package limit), and at equal cycles per FLOP the chip holds a higher clock on the 16x16x32 shape than on 32x32x16
// (MI355X_MICROARCH.md, DVFS give-back item 7: x1.12-1.14 with operands re-read from LDS).
// One k-tile (BK = 32) is ONE reduction step of the 16x16x32 shape, so the two halves of the pipelined loop body are the two ROW
// halves of the wave tile: half 1 = rows 0 .. 16 HALF - 1 with the B fragments of tile t while the A fragments of the second row
// half are read; barrier + LDS-DMA requests; half 2 = second row half while tile t + 1's first-half A and its B fragments are read
// (B is double-buffered in registers: the loop is unrolled by two, and an odd reduction range runs one extra all-zero tile).
// Accumulators: lane (c = lane & 15, g = lane >> 4), block (bi, bj), register e = C[16 bi + 4 g + e][4 c + bj] - the B tile is
// stored de-interleaved (kc_tile_row<2>) so that a lane owns four ADJACENT output columns: one 8-byte store per row.
// =================================================================================================
template <int AL, typename S>
__global__ void __launch_bounds__(S::NT, S::NT == 512 ? 1 : 3) gemm_dma16_kernel(const crog_gemm_desc p) {
  using T = bf16;
  constexpr int DMA_NSTAGE = dma_nstage<S>();
  constexpr int NW = S::NT / 64, NIA = S::BM / (16 * NW), NIB = S::BN / (16 * NW);
  static_assert(NIA >= 1 && NIB >= 1 && NIA * 16 * NW == S::BM && NIB * 16 * NW == S::BN, "every wave must move whole 1 KiB slices of both tiles");
  using OA = DmaKc<T, AL == CROG_A_IM2COL ? 1 : 0, NIA, 1>;
  using OB = DmaKc<T, 0, NIB, 2>;
  constexpr int BK = 32, BM = S::BM, BN = S::BN, RB = 2 * S::WM, CB = 2 * S::WN, HALF = RB / 2;
  static_assert(CB == 4 && RB % 2 == 0, "a wave tile is 64 columns wide (four interleaved 16-column blocks)");
  constexpr int TILE_A_B = BM * 64, TILE_B_B = BN * 64, DMA_STAGE_B = TILE_A_B + TILE_B_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / S::WVN, wc = wave % S::WVN;
  const int tilesN = (p.N + BN - 1) / BN, tilesM = (p.M + BM - 1) / BM;
  const int nwg = tilesM * tilesN;
  int id = blockIdx.x, z = blockIdx.y;
  xcd_map(nwg, 1, id, z);
  const int tm = id / tilesN, tn = id % tilesN;
  const int m0 = tm * BM, n0 = tn * BN;
  const int zo = z / p.batch_inner, zi = z % p.batch_inner;
  const T* A = reinterpret_cast<const T*>(p.A) + zo * p.sAo + zi * p.sAi;
  const T* B = reinterpret_cast<const T*>(p.B) + zo * p.sBo + zi * p.sBi;
  const int64_t coff = zo * p.sCo + zi * p.sCi;
  const int kt0 = 0;
  const int nt = (p.K + BK - 1) / BK;
  const int nt2 = (nt + 1) & ~1;

  const ConvGeom g{p.convH, p.convW, p.convC};
  OA da;
  OB db;
  da.init(A, p.lda, p.M, p.K, m0, wave, lane, g);
  db.init(B, p.ldb, p.N, p.K, n0, wave, lane, g);
  const int exa = OA::extent(p.lda, p.M, p.K, g), exb = OB::extent(p.ldb, p.N, p.K, g);

  f32x4 acc[RB][CB];
#pragma unroll
  for (int i = 0; i < RB; i++)
#pragma unroll
    for (int j = 0; j < CB; j++)
#pragma unroll
      for (int e = 0; e < 4; e++) acc[i][j][e] = 0.f;

  auto kmem_of = [&](int kt) -> int {
    if constexpr (AL == CROG_A_IM2COL) {
      int tap, chunk;
      conv_ktile(kt, p.convC / BK, tap, chunk);
      return tap * p.convC + chunk * BK;
    } else {
      return kt * BK;
    }
  };
  constexpr int PER_TILE = NIA + NIB;
  da.start(0, g);
  db.start(0, g);
#pragma unroll
  for (int s = 0; s < DMA_NSTAGE; s++)
    CROG_DMA_ISSUE(kt0 + s, kmem_of(kt0 + s), s, s < nt);
  wait_vmcnt<(DMA_NSTAGE - 1) * PER_TILE>();
  __builtin_amdgcn_s_barrier();
  const int arow = wr * RB * 16, brow = wc * CB * 16;      // the wave's first row of the A / B LDS image
  bf16x8 alo[HALF], ahi[HALF], b0[CB], b1[CB];
#pragma unroll
  for (int i = 0; i < HALF; i++) alo[i] = frag16_kc(smem, arow + i * 16, lane);
#pragma unroll
  for (int j = 0; j < CB; j++) b0[j] = frag16_kc(smem + TILE_A_B, brow + j * 16, lane);
  int stage = 0;
  // one k-tile: BC = the B fragments of this tile (in registers), BN_ = where the next tile's go
#define CROG_MF16_TILE(T_, BC, BN_)                                                                                    \
  do {                                                                                                                 \
    const char* at_ = smem + stage * DMA_STAGE_B;                                                                      \
    _Pragma("unroll") for (int i = 0; i < HALF; i++) ahi[i] = frag16_kc(at_, arow + (HALF + i) * 16, lane);            \
    _Pragma("unroll") for (int i = 0; i < HALF; i++)                                                                   \
      _Pragma("unroll") for (int j = 0; j < CB; j++) mma32(alo[i], BC[j], acc[i][j]);                                  \
    const int nstage_ = stage == DMA_NSTAGE - 1 ? 0 : stage + 1;                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
    wait_vmcnt<(DMA_NSTAGE - 2) * PER_TILE>();                                                                         \
    __builtin_amdgcn_s_barrier();                                                                                      \
    {                                                                                                                  \
      const int tq_ = (T_) + DMA_NSTAGE;                                                                               \
      CROG_DMA_ISSUE(kt0 + tq_, kmem_of(kt0 + tq_), stage, tq_ < nt);                                                  \
    }                                                                                                                  \
    const char* an_ = smem + nstage_ * DMA_STAGE_B;                                                                    \
    _Pragma("unroll") for (int i = 0; i < HALF; i++) alo[i] = frag16_kc(an_, arow + i * 16, lane);                     \
    _Pragma("unroll") for (int j = 0; j < CB; j++) BN_[j] = frag16_kc(an_ + TILE_A_B, brow + j * 16, lane);            \
    _Pragma("unroll") for (int i = 0; i < HALF; i++)                                                                   \
      _Pragma("unroll") for (int j = 0; j < CB; j++) mma32(ahi[i], BC[j], acc[HALF + i][j]);                           \
    {                                                                                                                  \
      constexpr int NMF = HALF * CB, MPD = NMF / PER_TILE > 0 ? NMF / PER_TILE : 1;                                    \
      __builtin_amdgcn_sched_group_barrier(0x100, HALF + CB, 0);                                                       \
      _Pragma("unroll") for (int q = 0; q < PER_TILE; q++) {                                                           \
        if (q * MPD < NMF) __builtin_amdgcn_sched_group_barrier(0x008, MPD, 0);                                        \
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                                                             \
      }                                                                                                                \
      if (PER_TILE * MPD < NMF) __builtin_amdgcn_sched_group_barrier(0x008, NMF - PER_TILE * MPD, 0);                  \
    }                                                                                                                  \
    stage = nstage_;                                                                                                   \
  } while (0)
  for (int t = 0; t < nt2; t += 2) {
    CROG_MF16_TILE(t, b0, b1);
    CROG_MF16_TILE(t + 1, b1, b0);
  }
#undef CROG_MF16_TILE
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();     // every wave's trailing (zero-writing) requests have landed before the ring is reused

  // ---- lean epilogue: column statistics, bf16 stores --------------------------------------------
  const int c = lane & 15, gq = lane >> 4;
  const bool rows_in = m0 + BM <= p.M;
  if (p.col_stats) {
    float s1[CB], s2[CB];
#pragma unroll
    for (int j = 0; j < CB; j++) s1[j] = s2[j] = 0.f;
#pragma unroll
    for (int i = 0; i < RB; i++) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const bool ok = rows_in || m0 + arow + i * 16 + 4 * gq + e < p.M;
#pragma unroll
        for (int j = 0; j < CB; j++) { const float v = ok ? acc[i][j][e] : 0.f; s1[j] += v; s2[j] += v * v; }
      }
    }
    constexpr int WVM = S::WVM, NT = S::NT;
    constexpr int RG = 16 * RB, GPS = 128 / RG;
    static_assert(RG <= 128 && 128 % RG == 0 && BM % 128 == 0, "stats slab granularity");
    float* red = reinterpret_cast<float*>(smem);      // [WVM][BN][2]
#pragma unroll
    for (int j = 0; j < CB; j++) {
      s1[j] = xor32_sum(xor16_sum(s1[j]));
      s2[j] = xor32_sum(xor16_sum(s2[j]));
    }
    if (gq == 0) {
#pragma unroll
      for (int j = 0; j < CB; j++) {
        const int col = brow + 4 * c + j;
        red[(wr * BN + col) * 2 + 0] = s1[j];
        red[(wr * BN + col) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    constexpr int SLABS = BM / 128;
    for (int idx = tid; idx < SLABS * BN * 2; idx += NT) {
      const int sl = idx / (BN * 2), k = idx % (BN * 2), cc = k >> 1;
      if (n0 + cc < p.N && m0 + sl * 128 < p.M) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < GPS; q++) v += red[(sl * GPS + q) * BN * 2 + k];
        if (p.stat_replicas > 0) atomicAdd(p.col_stats + ((int64_t)((m0 / 128 + sl) % p.stat_replicas) * p.N + n0) * 2 + k, v);
        else p.col_stats[((int64_t)(m0 / 128 + sl) * p.N + n0) * 2 + k] = v;
      }
    }
  }
  T* C = reinterpret_cast<T*>(p.C) + coff;
  const int col = n0 + brow + 4 * c;
  const bool interior = rows_in && n0 + BN <= p.N;
#pragma unroll
  for (int i = 0; i < RB; i++) {
    const int row0 = m0 + arow + i * 16 + 4 * gq;
    T* cb = C + (int64_t)row0 * p.ldc + col;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      bf16x4 v;
      v[0] = (bf16)acc[i][0][e]; v[1] = (bf16)acc[i][1][e]; v[2] = (bf16)acc[i][2][e]; v[3] = (bf16)acc[i][3][e];
      if (interior || (row0 + e < p.M && col < p.N)) *reinterpret_cast<bf16x4*>(cb + (int64_t)e * p.ldc) = v;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  fwd_stat_sync_tail(p);
}

#ifdef CROG_GEMM_PROBE
// asm-inspection build (never linked): hipcc -DCROG_GEMM_PROBE -S instantiates only these kernels
template __global__ void gemm_dma_kernel<bf16, CROG_A_IM2COL, CROG_B_KC, ShapeMid, 0>(const crog_gemm_desc);
template __global__ void gemm_dma_kernel<bf16, CROG_A_MC, CROG_B_NC_IM2COL, ShapeMid, 0>(const crog_gemm_desc);
template __global__ void gemm_dma_kernel<bf16, CROG_A_IM2COL, CROG_B_KC, ShapeDma8, 0>(const crog_gemm_desc);
template __global__ void gemm_dma16_kernel<CROG_A_IM2COL, ShapeDma8>(const crog_gemm_desc);
template __global__ void gemm_dma16_kernel<CROG_A_KC, ShapeMid>(const crog_gemm_desc);
}  // namespace
#else
// forward statistics + stat_sync: did the launched kernel carry the exchange in its own tail (the ping-pong tile; round 6: the LDS-DMA tiles too),
// or does crog_gemm add the single-block finish launch behind it?
static thread_local bool g_fwd_tail = false;
inline bool wants_fwd_tail(const crog_gemm_desc& d) { return d.stat_sync && d.col_stats && d.stat_replicas > 0 && !d.bwd_z && d.splitk <= 1 && d.batch == 1; }

template <typename T, int AL, int BL, typename S>
int launch_dma(const crog_gemm_desc& d, hipStream_t s) {
  constexpr int ring = dma_nstage<S>() * (S::BM + S::BN) * 64, epi = lds_bytes<T, S>();   // the epilogue reuses the ring
  constexpr int LDS = ring > epi ? ring : epi;
  static bool attr_set = false;
  constexpr bool CAN_ASUM = AL == CROG_A_MC && !S::ATOM;     // a_sum is only requested by weight-gradient launches (not on the atomic-only tile)
  auto kern = gemm_dma_kernel<T, AL, BL, S, 0>;
  auto kern_asum = gemm_dma_kernel<T, AL, BL, S, CAN_ASUM ? 1 : 0>;
  if (!attr_set) {
    for (auto k : {kern, kern_asum}) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (e != hipSuccess) {
        crog_set_error("crog_gemm: hipFuncSetAttribute(%d bytes) failed: %s", LDS, hipGetErrorString(e));
        return CROG_ERR_LAUNCH;
      }
    }
    attr_set = true;
  }
  if (d.a_sum && !CAN_ASUM) {
    crog_set_error("crog_gemm: a_sum needs the A_MC layout");
    return CROG_ERR_ARG;
  }
  dim3 grid(cdiv(d.M, S::BM) * cdiv(d.N, S::BN), d.batch * d.splitk, 1);
  if (splitk_by_xcd(d)) grid = dim3(grid.x * d.splitk, 1, 1);
  hipLaunchKernelGGL(d.a_sum ? kern_asum : kern, grid, dim3(S::NT), LDS, s, d);
  CROG_LAUNCH_CHECK();
  if (!S::BWD && !S::ATOM && sizeof(T) == 2 && wants_fwd_tail(d)) g_fwd_tail = true;
  return CROG_OK;
}


// v_mfma_f32_16x16x32 variant (gemm_dma16_kernel): bf16, A_KC / A_IM2COL x B_KC, lean epilogue, no split-K
template <int AL, typename S>
int launch_dma16(const crog_gemm_desc& d, hipStream_t s) {
  constexpr int ring = dma_nstage<S>() * (S::BM + S::BN) * 64, epi = S::WVM * S::BN * 2 * 4;
  constexpr int LDS = ring > epi ? ring : epi;
  static bool attr_set = false;
  auto kern = gemm_dma16_kernel<AL, S>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      crog_set_error("crog_gemm: hipFuncSetAttribute(%d bytes) failed: %s", LDS, hipGetErrorString(e));
      return CROG_ERR_LAUNCH;
    }
    attr_set = true;
  }
  dim3 grid(cdiv(d.M, S::BM) * cdiv(d.N, S::BN), d.batch, 1);
  hipLaunchKernelGGL(kern, grid, dim3(S::NT), LDS, s, d);
  CROG_LAUNCH_CHECK();
  if (wants_fwd_tail(d)) g_fwd_tail = true;
  return CROG_OK;
}

template <typename T, typename S>
int dispatch_dma(const crog_gemm_desc& d, hipStream_t s) {
  const int a = d.a_layout, b = d.b_layout;
  if (a == CROG_A_KC && b == CROG_B_KC) return launch_dma<T, CROG_A_KC, CROG_B_KC, S>(d, s);
  if (a == CROG_A_IM2COL && b == CROG_B_KC) return launch_dma<T, CROG_A_IM2COL, CROG_B_KC, S>(d, s);
  if (a == CROG_A_KC && b == CROG_B_NC) return launch_dma<T, CROG_A_KC, CROG_B_NC, S>(d, s);
  if (a == CROG_A_IM2COL && b == CROG_B_NC_DGRAD) return launch_dma<T, CROG_A_IM2COL, CROG_B_NC_DGRAD, S>(d, s);
  if (a == CROG_A_MC && b == CROG_B_NC) return launch_dma<T, CROG_A_MC, CROG_B_NC, S>(d, s);
  if (a == CROG_A_MC && b == CROG_B_NC_IM2COL) return launch_dma<T, CROG_A_MC, CROG_B_NC_IM2COL, S>(d, s);
  if (a == CROG_A_MC && b == CROG_B_KC) return launch_dma<T, CROG_A_MC, CROG_B_KC, S>(d, s);
  crog_set_error("crog_gemm: unsupported layout combination a=%d b=%d", a, b);
  return CROG_ERR_ARG;
}

// The DMA kernel addresses each operand through a 32-bit byte offset from its (batch-adjusted) base pointer and moves whole
// 16-byte chunks.  A ragged last chunk (K or a transposed operand's column count not a multiple of the chunk) is read in full:
// that is memory-safe when the row stride covers the rounded-up length, and value-safe because the partner operand's rows
// beyond K are zero-filled (transposed partner, exact row guard) or the extra columns only feed outputs that are never stored.
bool dma_eligible(const crog_gemm_desc& d) {
  const long esz = d.dtype == CROG_BF16 ? 2 : 4, vec = 16 / esz;
  const bool a_tr = d.a_layout == CROG_A_MC, b_tr = d.b_layout != CROG_B_KC;
  auto up = [&](long v) { return (v + vec - 1) / vec * vec; };
  if (d.K % vec != 0) {
    if (!a_tr && !b_tr) return false;                  // both K-contiguous: nobody zero-fills the tail
    if (!a_tr && (d.a_layout != CROG_A_KC || up(d.K) > d.lda)) return false;
    if (!b_tr && up(d.K) > d.ldb) return false;
  }
  if (a_tr && up(d.M) > d.lda) return false;
  if (b_tr && d.b_layout == CROG_B_NC && up(d.N) > d.ldb) return false;
  if (b_tr && d.b_layout != CROG_B_NC && d.N % vec != 0) return false;
  const long rows_a = a_tr ? d.K : d.M;
  long rows_b = d.N;
  if (d.b_layout == CROG_B_NC || d.b_layout == CROG_B_NC_IM2COL) rows_b = d.K;
  else if (d.b_layout == CROG_B_NC_DGRAD) rows_b = 9L * d.convC;
  if (rows_a * d.lda * esz >= 0x7fffffffL || rows_b * d.ldb * esz >= 0x7fffffffL) return false;
  return true;
}

template <typename T, int AL, int BL, bool HWTR, typename S>
int launch(const crog_gemm_desc& d, hipStream_t s) {
  constexpr int LDS = lds_bytes<T, S>();
  static bool attr_set = false;
  auto kern = gemm_kernel<T, AL, BL, HWTR, S>;
  if (!attr_set) {
    if (LDS > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (e != hipSuccess) {
        crog_set_error("crog_gemm: hipFuncSetAttribute(%d bytes) failed: %s", LDS, hipGetErrorString(e));
        return CROG_ERR_LAUNCH;
      }
    }
    attr_set = true;
  }
  const int tilesM = cdiv(d.M, S::BM), tilesN = cdiv(d.N, S::BN);
  dim3 grid(tilesM * tilesN, d.batch * d.splitk, 1);
  if (splitk_by_xcd(d)) grid = dim3(grid.x * d.splitk, 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(S::NT), LDS, s, d);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

template <typename T, bool HWTR, typename S>
int dispatch_layout(const crog_gemm_desc& d, hipStream_t s) {
  const int a = d.a_layout, b = d.b_layout;
  if (a == CROG_A_KC && b == CROG_B_KC) return launch<T, CROG_A_KC, CROG_B_KC, true, S>(d, s);
  if (a == CROG_A_IM2COL && b == CROG_B_KC) return launch<T, CROG_A_IM2COL, CROG_B_KC, true, S>(d, s);
  if (a == CROG_A_KC && b == CROG_B_NC) return launch<T, CROG_A_KC, CROG_B_NC, HWTR, S>(d, s);
  if (a == CROG_A_IM2COL && b == CROG_B_NC_DGRAD) return launch<T, CROG_A_IM2COL, CROG_B_NC_DGRAD, HWTR, S>(d, s);
  if (a == CROG_A_MC && b == CROG_B_NC) return launch<T, CROG_A_MC, CROG_B_NC, HWTR, S>(d, s);
  if (a == CROG_A_MC && b == CROG_B_NC_IM2COL) return launch<T, CROG_A_MC, CROG_B_NC_IM2COL, HWTR, S>(d, s);
  if (a == CROG_A_MC && b == CROG_B_KC) return launch<T, CROG_A_MC, CROG_B_KC, HWTR, S>(d, s);
  crog_set_error("crog_gemm: unsupported layout combination a=%d b=%d", a, b);
  return CROG_ERR_ARG;
}

// 1 = 64 x 64 tiles (latency-bound small problems without BatchNorm statistics), 2 = 128 x 128
int pick_shape(const crog_gemm_desc& d) {
  const long mid = (long)cdiv(d.M, 128) * cdiv(d.N, 128) * d.batch * d.splitk;
  static const long small_max = [] { const char* e = getenv("CROG_SMALL_TILE_MAX"); return e ? atol(e) : 192L; }();      // (A/B: below this many 128 x 128 tiles, 64 x 64)
  return (mid < small_max && !d.col_stats) ? 1 : 2;
}

// Small-output weight gradients are bound by L2 -> LDS bytes and by the fp32-atomic epilogue, not by MFMA (scripts/ablate_wgrad.py,
// HBM-cold operands: 512 x 512 over K = 21632 with 64 x 64 tiles and 16 splits = 44 us, 33 us of it without the atomics, 354 MB
// through L2 for 44 MB of operands).  Launched ALONE, 64 x 64 tiles win (4x the blocks of 128 x 128 at the same atomic bytes:
// 8-36 % faster, scripts/bench_wgrad2.py) -- but inside the training step the weight gradients run on a side stream next to the
// main stream's kernels, the chip is full anyway, and what counts is the work per launch: 128 x 128 tiles at ~256 blocks (half
// the L2 bytes, a third of the splits) make the step 2 % faster (33.97 vs 34.65 ms, two interleaved A/B pairs).  Outputs above 1 M
// elements (transformer FFN / ViT weights: many tiles already) aim at 512 blocks: CROG-R50 does not care, the ViT-B/16 tower, whose
// main stream leaves more of the chip free, loses 12 % with 256 (17.3 vs 15.4 ms per forward + backward).  The 3x3 form keeps
// 64 x 64 for outputs up to 128 x 1152 (Cout <= 128: 128-wide tiles would be mostly padding) and targets 512 blocks otherwise.
inline bool small_wgrad(int a_layout, int b_layout, int out_mode, long M, long N) {
  if (out_mode != CROG_OUT_F32_ATOMIC || a_layout != CROG_A_MC) return false;
  return b_layout == CROG_B_NC_IM2COL && M * N <= 160L * 1024;      // 3x3 weight gradients with Cout <= 128: 64 x 64 tiles
}

// Large weight gradients (both output sides multiples of 256, a reduction of >= 8192) on the 256 x 256 tile with the atomic-only
// epilogue (round 3): half the L2 -> LDS bytes and half the LDS fragment reads per MFMA of the 128 x 128 tile, which is what bounds a
// weight gradient (both operands are read transposed out of LDS).  One 8-wave block owns a CU, so a launch is kept to ~144 blocks: the
// side stream works a little over half of the chip at the better per-CU rate and leaves the rest to the main stream.  Measured in the
// step (interleaved runs of 60 steps, finite losses checked): 3x3 forms from 512 K outputs 32.55-33.03 ms against 33.2-33.6 on
// 128 x 128 tiles, and the roofline kernel next to them 573-580 instead of 528-537 TFLOP/s; adding the 1x1 / linear forms from 1 M
// outputs (the decoder's 2048 x 512 FFN weights) 32.36-32.73, from 256 K outputs 32.65-33.15 (no gain: twelve 512 x 512 launches
// whose four tiles need 21 splits).  Block targets of 80 / 96 / 112 help the neighbour more (617-629 TFLOP/s) and the step less,
// 176-208 lose: 144.  (A first measurement of 31.8-32.2 ms for the 256 K variant was of a step whose text tower had gone NaN - an
// unordered workspace, fixed - and NaN operands let the chip hold a higher clock: bench.py now refuses non-finite statistics.)
// Round 6, after the LDS-DMA requests became inline assembly (gemm_dma.h: a block is 18 % faster): 112 blocks are 0.1-0.4 ms ahead of 144 in the
// step (26.45-26.56 against 26.52-26.71 ms, six interleaved pairs; profiles/r06_ab_wgrad_blocks.txt) but read the operands TWICE: 18 tiles x 6
// slices are 13.5 blocks per XCD, every reduction slice lands in two or three L2s (PMC: 1149 MB fetched per launch for 531 MB of operands; with
// 8 slices = one per XCD: 1.0 x).  Whole slices on 6 of the 8 XCDs (CROG_PPT_XCDS=1, gemm_ppt.hip) fetch 1.00 x and LOSE 0.5 ms - the main
// chain's blocks on the six busy XCDs become the stragglers.  144 stays: one slice per XCD, and 0.4 % is inside the spread between boxes.
static const long WGRAD256_BLOCKS = [] { const char* e = getenv("CROG_WGRAD_BLOCKS"); return e ? atol(e) : 144L; }();      // (the environment override is for scripts/ A-B runs)
inline bool big_wgrad(int dtype, int a_layout, int b_layout, int out_mode, long M, long N, long K) {
  if (dtype != CROG_BF16 || out_mode != CROG_OUT_F32_ATOMIC || a_layout != CROG_A_MC) return false;
  if (M % 256 != 0 || N % 256 != 0 || K < 8192) return false;
  if (b_layout == CROG_B_NC_IM2COL) return M * N >= (1L << 19);
  return b_layout == CROG_B_NC && M * N >= (1L << 20);
}

// Which weight gradients take the ping-pong kernel (gemm_ppt.hip): the ones the 256 x 256 atomic tile took (big_wgrad), in either
// output form (atomic adds, or split-K slabs for crog_splitk_reduce).  CROG_PPT = 0 in the environment keeps the previous kernel;
// debug bit 15 of a descriptor forces the ping-pong kernel for any launch it can express (ragged tiles included), bit 16 forbids it.
inline bool ppt_wanted(const crog_gemm_desc& d) {
  static const int env = [] { const char* e = getenv("CROG_PPT"); return e ? atoi(e) : 1; }();
  if ((d.debug & 65536) || d.a_layout != CROG_A_MC) return false;
  if (d.debug & 32768) return crog_gemm_ppt_eligible(d);
  if (env <= 0) return false;
  const int om = d.out_mode == CROG_OUT_F32 ? CROG_OUT_F32_ATOMIC : d.out_mode;      // the slab form is sized like the atomic one
  return big_wgrad(d.dtype, d.a_layout, d.b_layout, om, d.M, d.N, d.K) && crog_gemm_ppt_eligible(d);
}

// Which launches take the v_mfma_f32_16x16x32 kernel (gemm_dma16_kernel): 2 (default) = the 256 x 256 tile of the large 3x3 forward /
// data-gradient launches and every lean 128 x 128 launch with K-contiguous operands and whole tiles (the BatchNorm'd convolutions and
// their plain data gradients), 1 = the 256 x 256 tile only, 0 = none.  CROG_MFMA16 in the environment overrides the default; debug
// bits 6 / 7 / 8 of a descriptor force 1 / 2 / 0 for that launch (tests, scripts/ab_mfma16.py).  Same bits in, same bits out: both
// shapes accumulate a k-tile's 32 products in fp32 in the same order (the A/B outputs are bit-identical).  Measured: back to
// back (the chip at its power limit) 1068 -> 1160, 1054 -> 1127, 1083 -> 1154 TFLOP/s on the 346112 x 512 x 2304, 346112 x 256 x 4608
// and 86528 x 512 x 4608 forwards (+6.5-8.7 %), 128 x 128 tile -1 ... +9 %; inside the training step, where BatchNorm passes between
// the GEMMs keep the chip off its power limit, less: 32.05-32.12 ms (0) / 31.93-31.97 (1) / 31.81-31.85 (2), three interleaved runs
// each; the 128 x 128 kernel at two or three blocks per CU: 31.80 either way.
inline int mf16_mode(const crog_gemm_desc& d) {
  static const int env = [] { const char* e = getenv("CROG_MFMA16"); return e ? atoi(e) : 2; }();
  if (d.debug & 256) return 0;
  if (d.debug & 64) return (d.debug & 128) ? 2 : 1;
  return env;
}
inline bool lean_epilogue_ok(const crog_gemm_desc& d) {
  return d.alpha == 1.f && !d.bias && d.act == CROG_ACT_NONE && !d.R && d.out_mode == CROG_OUT_T && d.dtype == CROG_BF16 && (d.N & 1) == 0;
}
// ... and what the BatchNorm-backward statistics tile takes: the same, or with a residual that joins before the gate
inline bool bwd_epilogue_ok(const crog_gemm_desc& d) {
  return d.alpha == 1.f && !d.bias && d.act == CROG_ACT_NONE && d.out_mode == CROG_OUT_T && d.dtype == CROG_BF16 && (d.N & 1) == 0 &&
         (!d.R || (d.ldr % 2 == 0 && ((uintptr_t)d.R % 4) == 0));
}

// Which launches take the ping-pong kernel (gemm_pp.hip), and with which tile height: 0 = none, 256 or 192.  The tile height is the one
// with the lower cost in (rounds of 256 one-per-CU blocks) x (rows per tile): 21632 x 512 is 170 tiles of 256 rows (one round at
// 66 % of the CUs) but 226 tiles of 192 rows (one round, three quarters of the work per tile).  CROG_PP in the environment: 0 = never,
// 1 = 3x3 launches from PP_MIN_TILES tiles, 2 (default) = also the 1x1 / linear forwards and data gradients.  Debug bits of a
// descriptor (tests, scripts/ab_pp.py): 9 = force 256 rows, 10 = force 192 rows, 11 = never; bits 12-14 = DMA distance (3 .. 7, 0 = default).
constexpr long PP_MIN_TILES = 150;
inline int pp_rows(const crog_gemm_desc& d) {
  static const int env = [] { const char* e = getenv("CROG_PP"); return e ? atoi(e) : 2; }();
  if (d.debug & 2048) return 0;
  if (d.debug & 512) return crog_gemm_pp_eligible(d, 256) ? 256 : 0;
  if (d.debug & 1024) return crog_gemm_pp_eligible(d, 192) ? 192 : 0;
  if (d.debug & 524288) return crog_gemm_pp_eligible(d, 128) ? 128 : 0;      // bit 19: the 128-row tile
  if (env <= 0 || !crog_gemm_pp_eligible(d, 256)) return 0;
  if (d.a_layout != CROG_A_IM2COL && env < 2) return 0;
  const long t256 = (long)cdiv(d.M, 256) * (d.N / 256), t192 = (long)cdiv(d.M, 192) * (d.N / 256);
  if (t256 < PP_MIN_TILES) {
    // 21632 x 256 (85 tiles of 256 rows): 169 tiles of 128 rows fill two thirds of the chip in one round - 1x1 / linear launches with
    // K >= 512 gain 15-25 % over the 128 x 128 tile (K = 1024: 27.2 -> 20.4 us), the 3x3 form does not (43.0 vs 43.8)
    const long t128 = (long)cdiv(d.M, 128) * (d.N / 256);
    static const bool pp128 = [] { const char* e = getenv("CROG_PP128"); return !e || atoi(e) != 0; }();
    if (pp128 && d.a_layout == CROG_A_KC && d.K >= 512 && t128 >= PP_MIN_TILES && crog_gemm_pp_eligible(d, 128)) return 128;
    return 0;
  }
  const long c256 = ((t256 + 255) / 256) * 256, c192 = ((t192 + 255) / 256) * 192;
  return (c192 * 11 < c256 * 10 && crog_gemm_pp_eligible(d, 192)) ? 192 : 256;      // (a 192-row tile runs ~10 % below the 256-row tile's rate per row)
}

__global__ void __launch_bounds__(256) stat_sync_finish_kernel(const CrogSyncBlock* sb, float* sums, int R, int n2) {
  crog_stat_sync_tail(sb, sums, R, n2, 1u);
}

template <typename T, bool HWTR>
int dispatch_shape(const crog_gemm_desc& d, hipStream_t s) {
  if (d.bwd_z) {      // BatchNorm-backward statistics: one dedicated tile, the three data-gradient layouts
    if constexpr (sizeof(T) == 2) {
      // round 5: the ping-pong tile with the same epilogue where its shape rules hold (N a multiple of 256, K >= 512 as for its full epilogue,
      // >= 150 tiles of 256 / 192 / 128 rows).  CROG_PP_BWDZ=0 / debug bit 11: the 128 x 128 tile
      static const bool pp_bwdz = [] { const char* e = getenv("CROG_PP_BWDZ"); return !e || atoi(e) != 0; }();
      if (pp_bwdz && !(d.debug & 2048) && dma_eligible(d) && bwd_epilogue_ok(d) && crog_gemm_pp_bwdz_ok(d) && (d.K >= 512 || (d.debug & (512 | 1024 | 524288)))) {
        const int rows = pp_rows(d);
        if (rows) return crog_gemm_pp_launch(d, rows, 0, s, 2);
      }
      if (dma_eligible(d) && bwd_epilogue_ok(d)) {
        if (d.a_layout == CROG_A_KC && d.b_layout == CROG_B_NC) return launch_dma<T, CROG_A_KC, CROG_B_NC, ShapeMidBwd>(d, s);
        if (d.a_layout == CROG_A_KC && d.b_layout == CROG_B_KC) return launch_dma<T, CROG_A_KC, CROG_B_KC, ShapeMidBwd>(d, s);
        if (d.a_layout == CROG_A_IM2COL && d.b_layout == CROG_B_KC) return launch_dma<T, CROG_A_IM2COL, CROG_B_KC, ShapeMidBwd>(d, s);
        if (d.a_layout == CROG_A_IM2COL && d.b_layout == CROG_B_NC_DGRAD) return launch_dma<T, CROG_A_IM2COL, CROG_B_NC_DGRAD, ShapeMidBwd>(d, s);
      }
    }
    crog_set_error("crog_gemm: bwd_z is implemented for bf16 data gradients (A_KC x B_NC / B_KC, A_IM2COL x B_KC / B_NC_DGRAD) with a plain epilogue "
                   "and operands the LDS-DMA path can address (crog_gemm_supports_bwd_z)");
    return CROG_ERR_ARG;
  }
  if constexpr (sizeof(T) == 2) {
    // the stem's first convolution on its im2col rows (N = K = 32, 1.38 M rows): streamed, no LDS stage.  CROG_SKINNY=0: the tiled kernel
    static const bool skinny = [] { const char* e = getenv("CROG_SKINNY"); return !e || atoi(e) != 0; }();
    if (skinny && d.M >= 16384 && crog_gemm_skinny_eligible(d)) {
      if (wants_fwd_tail(d)) g_fwd_tail = true;      // (its last block runs the exchange)
      return crog_gemm_skinny_launch(d, s);
    }
  }
  if constexpr (sizeof(T) == 2) {
    // small-channel 3x3 convolutions (stem, layer1: N = 32 / 64, K = 288 / 576) with at least 64 K pixels: the sliding-window kernel
    // fetches every input row once instead of nine times through L2 -> LDS.  CROG_CONV_SW=0 / debug bit 20: the implicit GEMM
    static const bool conv_sw = [] { const char* e = getenv("CROG_CONV_SW"); return !e || atoi(e) != 0; }();
    if (conv_sw && !(d.debug & 1048576) && d.a_layout == CROG_A_IM2COL && (d.M >= 65536 || (d.debug & 2097152)) && crog_conv_sw_eligible(d)) {
      if (wants_fwd_tail(d)) g_fwd_tail = true;      // (its last block runs the exchange)
      return crog_conv_sw_launch(d, s);
    }
  }
  if constexpr (sizeof(T) == 2) {
    // ... and their weight gradients (the tail of the step): CROG_WGRAD_SW=0 / debug bit 22: the implicit GEMM; bit 23 lifts the size threshold
    static const bool wgrad_sw = [] { const char* e = getenv("CROG_WGRAD_SW"); return !e || atoi(e) != 0; }();
    if (wgrad_sw && !(d.debug & 4194304) && d.a_layout == CROG_A_MC && d.b_layout == CROG_B_NC_IM2COL && (d.K >= 65536 || (d.debug & 8388608)) &&
        crog_wgrad_sw_eligible(d) && (d.convC == 32 || d.out_mode == CROG_OUT_F32 || (d.debug & 8388608)))
      // (64 -> 64 with atomic adds: 89 us either way - its 9.4 M adds cost 49 us; 256 slabs + crog_splitk_reduce: 108 us.  The slab form
      // serves deterministic mode, whose weight gradients are all slabs)
      return crog_wgrad_sw_launch(d, s);
  }
  const int shape = pick_shape(d);
  if (dma_eligible(d)) {
    // Tile shape of the LDS-DMA kernel.
    // Large 3x3 forward / data-gradient launches (bf16, N a multiple of 256, >= 160 tiles of 256 x 256 - with 170 tiles the
    // 21632 x 512 launches of the neck gain 25 % standalone, 135 -> 108 us, and the step 0.5 %; at 85 tiles the tile loses): the 8-wave
    // 256 x 256 tile halves the L2 -> LDS bytes per FLOP, which is what bounds the 128 x 128 tile (ablation: DMA-only 831 us vs
    // MFMA-only 603 us of a 1072 us launch).  Standalone +22-35 % on K = 4608 forwards (856 vs 691, 1031 vs 845 TFLOP/s), in the
    // training step -1.5 % (37.5 vs 38.2 ms, two interleaved A/B pairs); smaller launches lose to tile quantisation at one block per
    // CU and stay on 128 x 128.  (The same tile for 1x1 / linear forwards and data gradients, K = 128 ... 2048: no gain standalone -
    // their reductions are too short to amortise the 128 KiB ring's fill - and +0.4 ms in the step.)
    if constexpr (sizeof(T) == 2) {
      // (launches with a bias / activation / residual take the same tile with its full epilogue: the attention-pool and decoder
      // linears, the 1x1 data gradients that add the identity path's gradient.  CROG_PP_FULL=0 / debug bit 17: back to the 128 x 128 tile)
      static const bool pp_full = [] { const char* e = getenv("CROG_PP_FULL"); return !e || atoi(e) != 0; }();
      const bool lean = lean_epilogue_ok(d);
      // (measured per shape, single stream: K >= 512 gains 7-29 % - 5408 x 2048 x 2048 72.6 -> 51.6 us, 21632 x 512 x 2048 63 -> 54 -; the
      // two- and four-k-tile launches that only these epilogues have lose 13-50 %: the ring fill is most of their run)
      if (lean || (pp_full && !(d.debug & 131072) && (d.K >= 512 || (d.debug & (512 | 1024))) && crog_gemm_pp_full_epilogue_ok(d))) {
        const int rows = pp_rows(d);
        if (rows) {
          g_fwd_tail = true;      // (gemm_pp_kernel<.., EPI 0 / 1> ends with crog_stat_sync_tail when stat_sync is set)
          return crog_gemm_pp_launch(d, rows, (d.debug >> 12) & 7, s, !lean);
        }
      }
    }
    if constexpr (sizeof(T) == 2) {
      if (lean_epilogue_ok(d) && d.a_layout == CROG_A_IM2COL && d.b_layout == CROG_B_KC && d.N % 256 == 0 && (long)cdiv(d.M, 256) * (d.N / 256) >= 160) {
        if (mf16_mode(d) >= 1) return launch_dma16<CROG_A_IM2COL, ShapeDma8>(d, s);
        return launch_dma<T, CROG_A_IM2COL, CROG_B_KC, ShapeDma8>(d, s);
      }
    }
    if constexpr (sizeof(T) == 2) {
      if (ppt_wanted(d)) return crog_gemm_ppt_launch(d, (d.debug >> 12) & 7, s);
    }
    if constexpr (sizeof(T) == 2) {
      if (big_wgrad(d.dtype, d.a_layout, d.b_layout, d.out_mode, d.M, d.N, d.K) && d.alpha == 1.f && !d.a_sum && !d.bias && !d.R && d.batch == 1) {
        if (d.b_layout == CROG_B_NC_IM2COL) return launch_dma<T, CROG_A_MC, CROG_B_NC_IM2COL, ShapeDma8A>(d, s);
        return launch_dma<T, CROG_A_MC, CROG_B_NC, ShapeDma8A>(d, s);
      }
    }
    // Otherwise by padding waste: 64-wide sides for <= 64 columns, 64 x 64 for small problems and small 3x3 weight gradients
    if (d.col_stats) {
      if (d.N <= 64) return dispatch_dma<T, ShapeTall>(d, s);
    } else {
      if (d.M <= 64 && d.N <= 64) return dispatch_dma<T, ShapeDma64>(d, s);
      if (small_wgrad(d.a_layout, d.b_layout, d.out_mode, d.M, d.N)) return dispatch_dma<T, ShapeDma64>(d, s);
      if (d.N <= 64) return dispatch_dma<T, ShapeTall>(d, s);
      if (shape == 1) return dispatch_dma<T, ShapeDma64>(d, s);
    }
    if constexpr (sizeof(T) == 2) {     // (opt-in: the 128 x 128 tile on the 16x16x32 shape, lean launches with whole tiles)
      if (mf16_mode(d) >= 2 && lean_epilogue_ok(d) && d.b_layout == CROG_B_KC && d.N % 128 == 0 && d.splitk == 1 && d.M % 128 == 0) {
        if (d.a_layout == CROG_A_IM2COL) return launch_dma16<CROG_A_IM2COL, ShapeMid>(d, s);
        if (d.a_layout == CROG_A_KC) return launch_dma16<CROG_A_KC, ShapeMid>(d, s);
      }
    }
    return dispatch_dma<T, ShapeMid>(d, s);
  }
  switch (shape) {
    case 1: return dispatch_layout<T, HWTR, ShapeSmall>(d, s);
    default: return dispatch_layout<T, HWTR, ShapeMid>(d, s);
  }
}

}  // namespace

extern "C" int crog_gemm_stat_tiles(int M) { return cdiv(M, 128); }

// Split count for a weight-gradient GEMM C[M,N] += A^T B over K (out_mode CROG_OUT_F32_ATOMIC), matched to the tile shape
// crog_gemm will pick: about one block per CU (the side stream shares the chip with the main stream), at least 16-24 k-tiles per
// block (every split pays an atomic epilogue).
extern "C" int crog_gemm_splitk_hint(int dtype, int a_layout, int b_layout, int M, int N, int K) {
  const int bk = dtype == CROG_BF16 ? 32 : 16;
  const long ktiles = cdiv(K, bk);
  long s;
  if (big_wgrad(dtype, a_layout, b_layout, CROG_OUT_F32_ATOMIC, M, N, K)) {      // 256 x 256 tiles, >= 32 k-tiles per block
    const long tiles = (long)(M / 256) * (N / 256);
    s = std::min(std::max(1L, WGRAD256_BLOCKS / tiles), std::max(1L, ktiles / 32));
  } else if (small_wgrad(a_layout, b_layout, CROG_OUT_F32_ATOMIC, M, N)) {      // 64 x 64 tiles, ~2048 blocks, >= 16 k-tiles per block
    const long tiles = (long)cdiv(M, 64) * cdiv(N, 64);
    // (the layer1 / stem gradients - reductions over >= 346112 pixels - are the LAST of the step and run after the main chain has
    // finished, alone on the chip: twice the blocks there, except for the 9-tile 64 x 576 form, whose 2043 blocks already pay
    // 33 MB of atomics: 128 x 1152 273 -> 227 us, 32 x 288 over 1.38 M 145 -> 127 us alone, scripts/bench_tail_wgrad.py)
    const long target = (K >= (1 << 18) && (tiles >= 16 || tiles <= 5)) ? 4096 : 2048;
    s = std::min(std::max(1L, target / tiles), std::max(1L, ktiles / 16));
  } else {      // 128 x 128 tiles: ~256 blocks (512 for the 3x3 form and for outputs above 1 M elements), >= 24 k-tiles per block
    const long tiles = (long)cdiv(M, 128) * cdiv(N, 128);
    // (>= 346112-pixel reductions: the tail of the step, see above: 256 x 64 83 -> 60 us, 64 x 256 78 -> 55 us, 128 x 256 87 -> 70 us alone)
    const long target = (b_layout == CROG_B_NC_IM2COL || K >= (1 << 18)) ? 512 : ((long)M * N <= (1L << 20) ? 256 : 512);
    s = std::min(std::max(1L, target / tiles), std::max(1L, ktiles / 24));
  }
  return (int)std::max(1L, std::min(s, 1024L));
}

// Edge of the tile crog_gemm takes for a weight-gradient GEMM (256: the atomic-only 256 x 256 tile, which cannot also form a_sum - the caller
// then sums the bias gradient separately, crog_colsum).
extern "C" int crog_wgrad_sw_slabs(int M, int convH, int convW, int convC, int K) {
  static const bool wgrad_sw = [] { const char* e = getenv("CROG_WGRAD_SW"); return !e || atoi(e) != 0; }();
  static const bool slabs = [] { const char* e = getenv("CROG_WGRAD_SW_SLABS"); return !e || atoi(e) != 0; }();
  if (!wgrad_sw || !slabs || K < 65536 || convC <= 0) return 0;
  crog_gemm_desc d{};
  d.dtype = CROG_BF16; d.a_layout = CROG_A_MC; d.b_layout = CROG_B_NC_IM2COL;
  d.M = M; d.N = 9 * convC; d.K = K; d.lda = M; d.ldb = convC; d.ldc = 9 * convC; d.batch = 1; d.batch_inner = 1; d.splitk = 256;
  d.convH = convH; d.convW = convW; d.convC = convC; d.alpha = 1.f; d.out_mode = CROG_OUT_F32;
  return crog_wgrad_sw_eligible(d) ? 256 : 0;
}

extern "C" int crog_gemm_wgrad_tile(int dtype, int a_layout, int b_layout, int M, int N, int K) {
  if (big_wgrad(dtype, a_layout, b_layout, CROG_OUT_F32_ATOMIC, M, N, K)) return 256;
  return small_wgrad(a_layout, b_layout, CROG_OUT_F32_ATOMIC, M, N) ? 64 : 128;
}

// Can crog_gemm do the BatchNorm-backward statistics (bwd_z) for this descriptor?  The caller decides BEFORE the producer layer commits
// to skipping its own first pass (crog_amd/functional.py BnLink).
extern "C" int crog_gemm_supports_bwd_z(const crog_gemm_desc* dp) {
  if (!dp) return 0;
  const crog_gemm_desc& d = *dp;
  if (d.dtype != CROG_BF16 || !dma_eligible(d) || !bwd_epilogue_ok(d)) return 0;
  return (d.a_layout == CROG_A_KC && (d.b_layout == CROG_B_NC || d.b_layout == CROG_B_KC)) ||
         (d.a_layout == CROG_A_IM2COL && (d.b_layout == CROG_B_KC || d.b_layout == CROG_B_NC_DGRAD));
}

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
  CROG_CHECK_ARG(d.splitk == 1 || ((d.out_mode == CROG_OUT_F32_ATOMIC || d.out_mode == CROG_OUT_F32) && d.act == CROG_ACT_NONE && !d.R),
                 "crog_gemm: splitk > 1 needs fp32 output (atomic adds, or plain stores into [splitk][M][ldc] slabs), no activation, no residual");
  CROG_CHECK_ARG(!(d.splitk > 1 && d.out_mode == CROG_OUT_F32) || d.batch == 1, "crog_gemm: split-K slabs need batch == 1");
  CROG_CHECK_ARG(!d.R || d.batch == 1, "crog_gemm: residual only for unbatched GEMM");
  CROG_CHECK_ARG(!d.col_stats || (d.batch == 1 && d.splitk == 1), "crog_gemm: col_stats needs batch == 1 and splitk == 1");
  if (d.bwd_z)
    CROG_CHECK_ARG(d.col_stats && d.stat_replicas > 0 && d.dtype == CROG_BF16 && d.out_mode == CROG_OUT_T && (d.N & 1) == 0 &&
                       d.ldz % vec == 0 && ((uintptr_t)d.bwd_z % 16) == 0 && d.act == CROG_ACT_NONE && (!d.bwd_mask || d.N % 8 == 0),
                   "crog_gemm: bwd_z needs bf16 dtype output, col_stats with stat_replicas > 0, no activation, even N (a multiple of 8 with bwd_mask), aligned z");
  CROG_CHECK_ARG(!d.bwd_mask || d.bwd_z, "crog_gemm: bwd_mask only with bwd_z");
  CROG_CHECK_ARG(!d.stat_sync || d.bwd_z || (d.col_stats && d.stat_replicas > 0 && d.dtype == CROG_BF16),
                 "crog_gemm: stat_sync (an exchange in the kernel's tail) needs bwd_z, or forward statistics in replica rows (bf16)");
  CROG_CHECK_ARG(!d.a_sum || d.batch == 1, "crog_gemm: a_sum needs batch == 1");
  CROG_CHECK_ARG((long)d.batch * d.splitk <= 65535, "crog_gemm: batch*splitk too large");
  hipStream_t s = (hipStream_t)stream;
  g_fwd_tail = false;
  const int rc = d.dtype == CROG_BF16 ? dispatch_shape<bf16, true>(d, s) : dispatch_shape<float, true>(d, s);
  if (rc == CROG_OK && d.stat_sync && !d.bwd_z && !g_fwd_tail) {
    // SyncBatchNorm forward statistics behind a kernel that does not carry the exchange in its own tail (everything but the ping-pong tile):
    // one single-block launch adds the replica rows up and runs it - the same totals in the same place for the consumer
    hipLaunchKernelGGL(stat_sync_finish_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<const CrogSyncBlock*>(d.stat_sync), d.col_stats, d.stat_replicas, 2 * d.N);
    CROG_LAUNCH_CHECK();
  }
  return rc;
}
#endif  // CROG_GEMM_PROBE
