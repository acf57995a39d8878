// Sliding-window 3x3 WEIGHT GRADIENT for small channel counts (gfx950): dW[co][tap][ci] += sum over pixels dy[p][co] * x[p + tap][ci]
// for the stem's 32 / 64-channel layers (1.38 M pixels) and layer1's 64 -> 64 (346112 pixels) - the LAST launches of the step's backward
// pass, alone on the chip after the main chain has finished (clip.py:44-57, 208-213; crog_engine.py:87).
//
// As an implicit GEMM (A = dy^T, B = the im2col view of x, split 227- to 819-fold over the pixels) every 64 x 64 output tile fetches
// its operands through nine shifted L2 -> LDS passes and adds 147 KB of partial sums atomically: 100 us for 88 MB of operands
// (346112 x 64 x 576), 170 / 140 us for the stem's.  Here, as in conv_sw.hip, a workgroup walks down a strip of image rows:
//   * the last four x rows (with a zero border pixel on either side and zero padding up to a multiple of 32 pixels) and two dy rows
//     live in LDS rings, each row fetched once, a row ahead of the products that read it;
//   * NINE waves, one per tap: wave t keeps the whole [Cout][Cin] accumulator of its tap (16 tiles of 16 x 16 at 64 -> 64: 64 VGPRs)
//     for the entire strip and, per 32 pixels, multiplies the dy^T fragments (A operand) by the x fragments read at its tap's shift
//     (B operand).  Both operands have the PIXEL as the reduction index and as the slow memory index: fragments come out of LDS
//     through ds_read_b64_tr_b16 (cdna_hip_programming.md T10: lane 4 q + p of a 16-lane group addresses row q, columns 4 p .. 4 p + 3);
//   * one round of fp32 atomic adds per workgroup at the very end (Cout x 9 Cin floats).
// Dispatch: crog_gemm -> crog_wgrad_sw_eligible (bf16, CROG_A_MC x CROG_B_NC_IM2COL, CROG_OUT_F32_ATOMIC, 32 / 64 channels).
#include "gemm_dma.h"
#include <algorithm>

namespace {

constexpr int WS_NT = 576, WS_WAVES = 9;

// 16 (column) x 32 (k) fragment of a [k][columns] LDS image: two transposed reads (k rows 8 g .. 8 g + 3 and + 4 .. + 7 of lane group g)
__device__ __attribute__((always_inline)) inline bf16x8 ws_tr_frag(const char* lo, const char* hi) {
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)lo);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)hi);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Bank swizzle of a [pixel][C] image for the transposed reads: a 32-lane half reads eight pixel rows (8 g + q, g in a pair of lane groups)
// x 32 bytes of the same 16-channel block; unswizzled, rows two apart share their banks at 128-byte pixels (4-way), rows four apart
// at 64-byte pixels (2-way).  The 32-byte segment index of a block is XORed with (row >> 1) & 3 resp. (row >> 2) & 1: eight rows,
// eight different 32-byte columns of the 256-byte bank row.
template <int C>
__device__ __attribute__((always_inline)) inline unsigned ws_seg(unsigned row, unsigned seg) {
  return C == 64 ? (seg ^ ((row >> 1) & 3u)) : (seg ^ ((row >> 2) & 1u));
}

// CI, CO in {32, 64}: x is [pixels][CI], dy is [pixels][CO], C is the fp32 gradient [CO][ldc] with columns tap * CI + ci.
template <int CI, int CO>
__global__ void __launch_bounds__(WS_NT, 1) wgrad_sw_kernel(const crog_gemm_desc p, int rows_per_wg) {
  constexpr int CB = CO / 16, IB = CI / 16;                // 16-wide blocks of the accumulator
  constexpr int XPX = CI * 2, DPX = CO * 2;                // bytes per pixel
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int H = p.convH, W = p.convW;
  const int WP = (W + 31) & ~31;                           // pixels per row, padded to whole k-steps (the padding stays zero)
  const int total_rows = p.K / W;                          // B * H
  const int tid = threadIdx.x, lane = tid & 63;
  const int tap = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ty = tap / 3, tx = tap - 3 * ty;
  const int l15 = lane & 15, g4 = lane >> 4, q4 = l15 >> 2, pp = l15 & 3;
  const unsigned XRS = (unsigned)(WP + 2) * XPX, DRS = (unsigned)WP * DPX;      // bytes of an x / dy ring row
  char* xring = smem;                                      // 4 rows: x row r in slot r & 3, pixel q at slot pixel q + 1
  char* xzero = smem + 4 * XRS;                            // the all-zero x row (rows above / below the image)
  char* dring = xzero + XRS;                               // 2 rows: dy row r in slot r & 1

  const int r0 = blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, total_rows);
  // CROG_OUT_F32 (slab form): workgroup b STORES its sums into slab b of a [strips][Cout][ldc] workspace; the caller adds the slabs up
  // in order (crog_splitk_reduce).  For 64 -> 64 channels the 9.4 M atomic adds of the other form cost more than the products.
  const bool slab = p.out_mode == CROG_OUT_F32;
  float* G = reinterpret_cast<float*>(p.C) + (slab ? (int64_t)blockIdx.x * p.M * p.ldc : 0);
  if (r0 >= r1) {
    if (slab)
      for (int i = tid; i < p.M * 9 * CI; i += WS_NT) G[(int64_t)(i / (9 * CI)) * p.ldc + i % (9 * CI)] = 0.f;
    return;
  }

  // everything starts as zeros: borders, padding pixels and the zero row are never written again
  for (unsigned i = tid * 16u; i < 5u * XRS + 2u * DRS; i += WS_NT * 16u) *reinterpret_cast<f32x4*>(smem + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  const bf16* X = reinterpret_cast<const bf16*>(p.B);      // B operand of the GEMM form: x
  const bf16* DY = reinterpret_cast<const bf16*>(p.A);     // A operand: dy
  const unsigned xch = (unsigned)W * (CI / 8), dch = (unsigned)W * (CO / 8);      // 16-byte chunks of an x / dy row
  constexpr int LX = 2, LD = 3;                            // chunks per thread and row (the launcher checks W * C / 8 <= L * 576)
  f32x4 sx[LX], sd[LD], tx_[LX], td_[LD];                 // two staging sets: two rows of operands in flight
  const int last_x = min(r1, total_rows - 1);

  // unconditional loads (an unwanted row reads row r0 and is dropped by the commit): see conv_sw.hip
  auto fetch_x = [&](int gr, f32x4 (&sx)[LX]) {
    const int gs = (gr >= 0 && gr <= last_x) ? gr : r0;
#pragma unroll
    for (int i = 0; i < LX; i++) {
      const unsigned q = min((unsigned)tid + (unsigned)i * WS_NT, xch - 1);
      sx[i] = *reinterpret_cast<const f32x4*>(X + ((int64_t)gs * xch + q) * 8);
    }
  };
  auto commit_x = [&](int gr, const f32x4 (&sx)[LX]) {
    if (gr < 0 || gr > last_x) return;
    char* dst = xring + (unsigned)(gr & 3) * XRS;
#pragma unroll
    for (int i = 0; i < LX; i++) {
      const unsigned q = (unsigned)tid + (unsigned)i * WS_NT;
      if (q < xch) {
        const unsigned px = q / (CI / 8) + 1, ch = q % (CI / 8);      // ring pixel (one border pixel), 16-byte chunk inside the pixel
        *reinterpret_cast<f32x4*>(dst + px * XPX + (ws_seg<CI>(px, ch >> 1) << 5) + ((ch & 1) << 4)) = sx[i];
      }
    }
  };
  auto fetch_d = [&](int gr, f32x4 (&sd)[LD]) {
    const int gs = gr < r1 ? gr : r0;
#pragma unroll
    for (int i = 0; i < LD; i++) {
      const unsigned q = min((unsigned)tid + (unsigned)i * WS_NT, dch - 1);
      sd[i] = *reinterpret_cast<const f32x4*>(DY + ((int64_t)gs * dch + q) * 8);
    }
  };
  auto commit_d = [&](int gr, const f32x4 (&sd)[LD]) {
    if (gr >= r1) return;
    char* dst = dring + (unsigned)(gr & 1) * DRS;
#pragma unroll
    for (int i = 0; i < LD; i++) {
      const unsigned q = (unsigned)tid + (unsigned)i * WS_NT;
      if (q < dch) {
        const unsigned px = q / (CO / 8), ch = q % (CO / 8);
        *reinterpret_cast<f32x4*>(dst + px * DPX + (ws_seg<CO>(px, ch >> 1) << 5) + ((ch & 1) << 4)) = sd[i];
      }
    }
  };

  // prologue: x rows r0 - 1, r0, r0 + 1 and dy row r0 - all requested before the first is written to LDS (one memory round trip) - and,
  // left in flight in the first staging set, x row r0 + 2 and dy row r0 + 1 (committed at the end of the first iteration)
  {
    f32x4 p0[LX], p1[LX];
    fetch_x(r0 - 1, p0);
    fetch_x(r0, p1);
    fetch_x(r0 + 1, tx_);
    fetch_d(r0, td_);
    fetch_x(r0 + 2, sx);
    fetch_d(r0 + 1, sd);
    commit_x(r0 - 1, p0);
    commit_x(r0, p1);
    commit_x(r0 + 1, tx_);
    commit_d(r0, td_);
  }
  __syncthreads();

  f32x4 acc[CB][IB];
#pragma unroll
  for (int c = 0; c < CB; c++)
#pragma unroll
    for (int i = 0; i < IB; i++) acc[c][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // lane's share of a transposed read: k row 8 g4 + q4 (+ 4), columns 4 pp .. + 3 of a 16-column block
  const unsigned krow = (unsigned)(8 * g4 + q4);
  // one row of products; `fs` receives the operands of two rows ahead (x row gr + 3, dy row gr + 2), `cs` holds what was requested an
  // iteration ago (x row gr + 2, dy row gr + 1) and is written to LDS after the products: its slots (x: (gr + 2) & 3 = (gr - 2) & 3,
  // dy: (gr + 1) & 1) were last read by row gr - 1's products, which every wave has left (the barrier that ended that iteration).
  // Two rows of products (~2.5 us) cover a memory round trip; one did not (the first version spent two thirds of its time waiting).
  auto row = [&](int gr, f32x4 (&fsx)[LX], f32x4 (&fsd)[LD], const f32x4 (&csx)[LX], const f32x4 (&csd)[LD]) {
    fetch_x(gr + 3, fsx);
    fetch_d(gr + 2, fsd);
    const int y = gr % H, yy = y + ty - 1;
    const char* xrow = (yy < 0 || yy >= H) ? xzero : xring + (unsigned)((gr + ty - 1) & 3) * XRS;
    const char* drow = dring + (unsigned)(gr & 1) * DRS;
    for (int ks = 0; ks < WP; ks += 32) {
      bf16x8 af[CB], bf[IB];
      const unsigned dr = (unsigned)ks + krow, xr = (unsigned)(ks + tx) + krow;      // tap shift: ring pixel = pixel + tx (one border pixel)
      const char* dlo = drow + dr * DPX + pp * 8;
      const char* xlo = xrow + xr * XPX + pp * 8;
#pragma unroll
      for (int c = 0; c < CB; c++) af[c] = ws_tr_frag(dlo + (ws_seg<CO>(dr, c) << 5), dlo + 4 * DPX + (ws_seg<CO>(dr + 4, c) << 5));
#pragma unroll
      for (int i = 0; i < IB; i++) bf[i] = ws_tr_frag(xlo + (ws_seg<CI>(xr, i) << 5), xlo + 4 * XPX + (ws_seg<CI>(xr + 4, i) << 5));
#pragma unroll
      for (int c = 0; c < CB; c++)
#pragma unroll
        for (int i = 0; i < IB; i++) mma32(af[c], bf[i], acc[c][i]);
    }
    commit_x(gr + 2, csx);
    commit_d(gr + 1, csd);
    __syncthreads();
  };
  for (int gr = r0; gr < r1; gr += 2) {
    row(gr, tx_, td_, sx, sd);
    if (gr + 1 < r1) row(gr + 1, sx, sd, tx_, td_);      // (wave-uniform)
  }

  // acc[c][i][e] = dW[co = 16 c + 4 g4 + e][tap][ci = 16 i + l15]
  if (p.debug & 32) return;          // timing-only ablation (as in crog_gemm's split-K kernels): what the atomic adds cost
#pragma unroll
  for (int c = 0; c < CB; c++)
#pragma unroll
    for (int i = 0; i < IB; i++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float* dst = G + (int64_t)(16 * c + 4 * g4 + e) * p.ldc + tap * CI + 16 * i + l15;
        if (slab) *dst = acc[c][i][e];
        else atomicAdd(dst, acc[c][i][e]);
      }
}

template <int CI, int CO>
int launch_wsw(const crog_gemm_desc& d, hipStream_t s) {
  const int W = d.convW, WP = (W + 31) & ~31;
  const int rows = d.K / W;
  // atomic form: ~one workgroup per CU; slab form: exactly d.splitk strips (slabs), empty ones store zeros
  const bool slab = d.out_mode == CROG_OUT_F32;
  const int per = slab ? cdiv(rows, d.splitk) : std::max(cdiv(rows, 256), std::min(4, rows));
  const int wgs = slab ? d.splitk : cdiv(rows, per);
  const int lds = 5 * (WP + 2) * CI * 2 + 2 * WP * CO * 2;
  static bool attr_set = false;
  auto kern = wgrad_sw_kernel<CI, CO>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    if (e != hipSuccess) {
      crog_set_error("crog_gemm: hipFuncSetAttribute failed for the sliding-window weight gradient: %s", hipGetErrorString(e));
      return CROG_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(WS_NT), lds, s, d, per);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

}  // namespace

// Can the sliding-window weight-gradient kernel take this launch?
bool crog_wgrad_sw_eligible(const crog_gemm_desc& d) {
  if (d.dtype != CROG_BF16 || d.a_layout != CROG_A_MC || d.b_layout != CROG_B_NC_IM2COL || d.batch != 1) return false;
  if (d.out_mode != CROG_OUT_F32_ATOMIC && !(d.out_mode == CROG_OUT_F32 && d.splitk >= 1 && d.splitk <= 4096)) return false;
  if (d.alpha != 1.f || d.bias || d.R || d.a_sum || d.col_stats || d.act != CROG_ACT_NONE) return false;
  if ((d.M != 32 && d.M != 64) || (d.convC != 32 && d.convC != 64) || d.N != 9 * d.convC) return false;
  if (d.lda != d.M || d.ldb != d.convC) return false;                              // dense [pixels][C] operands
  if (d.convW < 16 || d.convH < 1 || d.K % ((long)d.convH * d.convW) != 0) return false;
  const int WP = (d.convW + 31) & ~31;
  if ((long)d.convW * (d.convC / 8) > 2L * WS_NT || (long)d.convW * (d.M / 8) > 3L * WS_NT) return false;      // staging registers per row
  if (5L * (WP + 2) * d.convC * 2 + 2L * WP * d.M * 2 > 160 * 1024 - 512) return false;
  if (((uintptr_t)d.A % 16) != 0 || ((uintptr_t)d.B % 16) != 0 || ((uintptr_t)d.C % 4) != 0) return false;
  return true;
}

int crog_wgrad_sw_launch(const crog_gemm_desc& d, hipStream_t s) {
  if (d.convC == 32 && d.M == 32) return launch_wsw<32, 32>(d, s);
  if (d.convC == 32 && d.M == 64) return launch_wsw<32, 64>(d, s);
  if (d.convC == 64 && d.M == 32) return launch_wsw<64, 32>(d, s);
  if (d.convC == 64 && d.M == 64) return launch_wsw<64, 64>(d, s);
  crog_set_error("crog_gemm: no sliding-window weight-gradient instantiation for %d -> %d channels", d.convC, d.M);
  return CROG_ERR_ARG;
}
