// Ping-pong LDS-DMA kernel for WEIGHT GRADIENTS (gfx950): C[M][N] (+)= sum_k A[k][m] * B[k][n] with both operands stored with
// the reduction index (the pixel / token) as the SLOW memory index — A = dy [K][M], B = x [K][N] (1x1 / linear) or the 3x3 im2col
// view of an NHWC map (n = tap * C + c reads x[k shifted by the tap][c]).  256 x 256 output tile, 64-deep k-tiles, split-K over
// the grid.  The schedule is gemm_pp.hip's (two wave groups one barrier apart; four 16-KiB half-tiles per k-tile; LOAD / MFMA
// phases; the same RAW / WAR argument), what differs is the operand image:
//   * a half-tile is [64 reduction rows][128 columns] (256-byte rows, contiguous in memory): A-h0 / A-h1 = tile columns 0-127 /
//     128-255 of dy, B-h0 / B-h1 likewise of x.  Wave (wr, wc) owns output rows {64 wr .. +63} of EACH A half and output columns
//     {32 wc .. +31} of EACH B half, so the phase order (A-h0 x B-h0, A-h0 x B-h1, A-h1 x B-h1, A-h1 x B-h0) and the staggered
//     deadlines of the half-tiles are the forward kernel's.
//   * fragments are read TRANSPOSED out of LDS: ds_read_b64_tr_b16 delivers a 4 (k) x 16 (column) block column-major, two reads per
//     16x16x32 fragment (k = 8 g .. 8 g + 3 and + 4 .. + 7 of lane group g).  The 16-byte chunk index of a row is XOR-swizzled by
//     ((row & 3) << 2) | ((row >> 2) & 3) on the DMA source address and again on the read (cdna_hip_programming.md T10, image (b)):
//     conflict-free for two lane groups 8 rows apart in the same columns, which is what a 32-lane half of this read is.
//   * 3x3 form: a lane's chunk has a fixed (tap, channel); the pixel of its row advances by 64 per k-tile — image coordinates are
//     kept incrementally (two adds, two selects) and the tap's validity is four compares per request.
// Output: fp32 atomic adds (out_mode CROG_OUT_F32_ATOMIC), or — CROG_OUT_F32 with splitk > 1 — plain stores of the block's partial
// tile into slab z of a [splitk][M][ldc] workspace that crog_splitk_reduce then sums in slice order (bit-reproducible).
#include "gemm_dma.h"
#include <algorithm>
#include <type_traits>

namespace {

__device__ __attribute__((always_inline)) inline bf16x8 tr_frag(const char* lo, const char* hi) {
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)lo);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)hi);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// One problem of a GROUPED launch (crog_gemm_group): what the kernel body reads of a descriptor, and the first block of the problem
struct PptProb {
  const void* A;
  const void* B;
  void* C;
  float* a_sum;      // bias gradient riding along: a_sum[m] += sum_k A[k][m], by cdiv(M, 256) * splitk extra blocks behind the problem's GEMM blocks
  int M, N, K, lda, ldb, ldc, splitk, convH, convW, convC, conv3, start;
};
constexpr int PPT_GROUP_MAX = 32;
struct PptGroup {
  int n, blocks;
  PptProb p[PPT_GROUP_MAX];
};

// BL: CROG_B_NC (dense) or CROG_B_NC_IM2COL; SLAB: plain stores into the split's slab instead of atomic adds; D: DMA distance;
// P: crog_gemm_desc or PptProb; blk: the block's index inside its problem
template <int BL, bool SLAB, int D, typename P>
__device__ __attribute__((always_inline)) inline void ppt_body(const P& p, const int blk) {
  static_assert(BL == CROG_B_NC || BL == CROG_B_NC_IM2COL, "transposed B operands");
  static_assert(D >= 3 && D <= 7, "DMA distance in half-tiles");
  constexpr int BM = 256, BN = 256, BK = 64, RBQ = 4, RB = 8, CB = 4;
  constexpr int SLOT = 16384;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;      // (waves w and w + 4 share a SIMD: scripts/simd_map.hip)
  const int tilesN = (p.N + BN - 1) / BN, tilesM = (p.M + BM - 1) / BM;
  const int nwg = tilesM * tilesN;
  int id = blk, z = 0;
  bool whole = false;
  if constexpr (std::is_same<P, crog_gemm_desc>::value) {
    // WHOLE reduction slices per XCD (launch_ppt sets bit 30 and sizes the grid to 8 x blocks-per-XCD): with S <= 8 slices, 8 / S XCDs share one slice
    // and S * (8 / S) XCDs work - the others' blocks leave at once, i.e. those XCDs and their L2s stay with the main chain; with S > 8, ceil(S / 8)
    // slices per XCD.  An operand slice is then fetched into 8 / S L2s (one for S >= 5) instead of the two or three that a cut of the slice-major
    // work list into eight equal runs touches when S is not a multiple of 8 (xcd_map below: 6 slices x 18 tiles = 13.5 blocks per XCD).
    if (p.debug & (1 << 30)) {
      const int S = p.splitk, xcd = blk & 7, loc = blk >> 3;
      if (S <= 8) {
        const int c = 8 / S, per = (nwg + c - 1) / c;
        z = xcd / c;
        id = (xcd - z * c) * per + loc;
        if (xcd >= S * c || id >= nwg) return;
      } else {
        const int spx = (S + 7) >> 3;
        z = xcd * spx + loc / nwg;
        id = loc % nwg;
        if (z >= S) return;
      }
      whole = true;
    }
  }
  if (!whole) xcd_map(nwg, p.splitk, id, z);          // (reduction slice, tile) runs per XCD: an operand slice lands in one L2
  const int tm = id / tilesN, tn = id - tm * tilesN;
  const int m0 = tm * BM, n0 = tn * BN;
  const bf16* A = reinterpret_cast<const bf16*>(p.A);
  const bf16* B = reinterpret_cast<const bf16*>(p.B);
  const int H = p.convH, W = p.convW, Cc = p.convC;

  const int ktiles = (p.K + BK - 1) / BK;
  const int per = (ktiles + p.splitk - 1) / p.splitk;
  const int kt0 = z * per;
  const int nt = min(per, ktiles - kt0);          // may be <= 0 for a trailing slice: it then only writes zeros (slab) / nothing
  const int nt2 = (max(nt, 0) + 1) & ~1;
  const int kbase = kt0 * BK;

  // ---- DMA side: lane l of piece i of wave w fills slot row 8 w + 4 i + (l >> 4), chunk l & 15 from source chunk (l & 15) ^ swz(row)
  const unsigned ldab = (unsigned)(p.lda * 2), ldbb = (unsigned)(p.ldb * 2);
  unsigned abase[2], bbase[2];
  unsigned avalid = 0, bvalid = 0;      // bit 2 i + h: the lane's 8 columns of request (i, h) exist
  unsigned tapdd = 0;                   // 3x3: (dy + 1) | (dx + 1) << 2 of the tap of request (i, h), at bits 8 i + 4 h
  int btap[2][2];                       // 3x3: byte offset of the request's (tap shift, channel)
  int px[2], py[2];                     // 3x3: image coordinates of the lane's two rows in the k-tile the next B request belongs to
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int row = 8 * wave + 4 * i + (lane >> 4);
    const int sc = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
    abase[i] = (unsigned)(kbase + row) * ldab + (unsigned)((m0 + 8 * sc) * 2);
    px[i] = py[i] = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (m0 + 128 * h + 8 * sc < p.M) avalid |= 1u << (2 * i + h);
      const int ncol = n0 + 128 * h + 8 * sc;
      if (ncol < p.N) bvalid |= 1u << (2 * i + h);
      btap[i][h] = 0;
      if constexpr (BL == CROG_B_NC_IM2COL) {
        const int tap = min(ncol / Cc, 8), c = ncol - tap * Cc;
        const int dy = tap / 3 - 1, dx = tap - 3 * (tap / 3) - 1;
        btap[i][h] = (dy * W + dx) * (int)ldbb + c * 2;
        tapdd |= (unsigned)((dy + 1) | ((dx + 1) << 2)) << (8 * i + 4 * h);
      }
    }
    if constexpr (BL == CROG_B_NC_IM2COL) {
      bbase[i] = (unsigned)(kbase + row) * ldbb;
      px[i] = (kbase + row) % W;
      py[i] = ((kbase + row) / W) % H;
    } else {
      bbase[i] = (unsigned)(kbase + row) * ldbb + (unsigned)((n0 + 8 * sc) * 2);
    }
  }
  const int stepx = BL == CROG_B_NC_IM2COL ? BK % W : 0, stepy = BL == CROG_B_NC_IM2COL ? (BK / W) % H : 0;
  const int exa = (int)(((long)(p.K - 1) * p.lda + p.M) * 2);
  const int exb = (int)(((long)(p.K - 1) * p.ldb + (BL == CROG_B_NC_IM2COL ? Cc : p.N)) * 2);

  // ---- fragment side: lane (i = lane & 15: q = i >> 2, pp = i & 3; g = lane >> 4) addresses row 8 g + q (+ 4 for the second read),
  // columns 4 pp .. 4 pp + 3 of the block
  const int c16 = lane & 15, gq = lane >> 4, q4 = c16 >> 2, pp = c16 & 3;
  unsigned aad[2][RBQ], bad[2][2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const unsigned rowb = (unsigned)((8 * gq + q4 + 4 * s) * 256), sw = (unsigned)((q4 << 2) | ((2 * gq + s) & 3));
#pragma unroll
    for (int r = 0; r < RBQ; r++) aad[s][r] = rowb + ((((unsigned)(8 * wr + 2 * r + (pp >> 1))) ^ sw) << 4) + (unsigned)((pp & 1) * 8);
#pragma unroll
    for (int c = 0; c < 2; c++) bad[s][c] = rowb + ((((unsigned)(4 * wc + 2 * c + (pp >> 1))) ^ sw) << 4) + (unsigned)((pp & 1) * 8);
  }

  f32x4 acc[RB][CB];
#pragma unroll
  for (int i = 0; i < RB; i++)
#pragma unroll
    for (int j = 0; j < CB; j++)
#pragma unroll
      for (int e = 0; e < 4; e++) acc[i][j][e] = 0.f;

  // the two requests of half-tile type J (0: A-h0, 1: B-h0, 2: B-h1, 3: A-h1) of k-tile TI into buffer BUF.  k-tiles are requested
  // strictly in order, so the 3x3 form advances its image coordinates after the B-h1 request of every k-tile.
#define PT_ISSUE(J, BUF, TI) PT_ISSUE_RANGE(J, BUF, TI, 0, 2)
#define PT_ISSUE_RANGE(J, BUF, TI, I0, I1)                                                                             \
  do {                                                                                                                 \
    const int ti_ = (TI);                                                                                              \
    const bool lv_ = ti_ < nt;                                                                                         \
    char* dst_ = smem + ((BUF) * 4 + (J)) * SLOT + wave * 2048;                                                        \
    if constexpr ((J) == 0 || (J) == 3) {                                                                              \
      constexpr int h_ = (J) == 3 ? 1 : 0;                                                                             \
      const unsigned kk_ = (unsigned)(ti_ * BK) * ldab + (h_ ? 256u : 0u);                                             \
      _Pragma("unroll") for (int i_ = (I0); i_ < (I1); i_++) {                                                         \
        const bool ok_ = lv_ & (((avalid >> (2 * i_ + h_)) & 1u) != 0);                                                \
        dma_piece(A, exa, dst_ + i_ * 1024, ok_ ? abase[i_] + kk_ : DMA_OOB);                                          \
      }                                                                                                                \
    } else {                                                                                                           \
      constexpr int h_ = (J) == 2 ? 1 : 0;                                                                             \
      const unsigned kk_ = (unsigned)(ti_ * BK) * ldbb;                                                                \
      _Pragma("unroll") for (int i_ = (I0); i_ < (I1); i_++) {                                                         \
        bool ok_ = lv_ & (((bvalid >> (2 * i_ + h_)) & 1u) != 0);                                                      \
        unsigned off_;                                                                                                 \
        if constexpr (BL == CROG_B_NC_IM2COL) {                                                                        \
          const int dd_ = (int)((tapdd >> (8 * i_ + 4 * h_)) & 15u);                                                   \
          const int sy_ = py[i_] + (dd_ & 3) - 1, sx_ = px[i_] + (dd_ >> 2) - 1;                                        \
          const int kr_ = kbase + ti_ * BK + 8 * wave + 4 * i_ + (lane >> 4);                                          \
          ok_ = ok_ & ((unsigned)sy_ < (unsigned)H) & ((unsigned)sx_ < (unsigned)W) & (kr_ < p.K);                      \
          off_ = bbase[i_] + kk_ + (unsigned)btap[i_][h_];                                                             \
        } else {                                                                                                       \
          off_ = bbase[i_] + kk_ + (h_ ? 256u : 0u);                                                                   \
        }                                                                                                              \
        dma_piece(B, exb, dst_ + i_ * 1024, ok_ ? off_ : DMA_OOB);                                                     \
      }                                                                                                                \
      if constexpr (BL == CROG_B_NC_IM2COL && (J) == 2 && (I1) == 2) {                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; i_++) {                                                             \
          int x_ = px[i_] + stepx, y_ = py[i_] + stepy;                                                                \
          const bool wrap_ = x_ >= W;                                                                                  \
          x_ = wrap_ ? x_ - W : x_;                                                                                    \
          y_ = wrap_ ? y_ + 1 : y_;                                                                                    \
          y_ = y_ >= H ? y_ - H : y_;                                                                                  \
          px[i_] = x_;                                                                                                 \
          py[i_] = y_;                                                                                                 \
        }                                                                                                              \
      }                                                                                                                \
    }                                                                                                                  \
  } while (0)

#define PT_PROLOGUE_ONE(M_)                                                                                            \
  if constexpr ((M_) < D) PT_ISSUE((M_) & 3, ((M_) >> 2) & 1, (M_) >> 2)
  PT_PROLOGUE_ONE(0); PT_PROLOGUE_ONE(1); PT_PROLOGUE_ONE(2); PT_PROLOGUE_ONE(3);
  PT_PROLOGUE_ONE(4); PT_PROLOGUE_ONE(5); PT_PROLOGUE_ONE(6);
#undef PT_PROLOGUE_ONE
  wait_vmcnt<2 * (D - 2)>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();      // the second group runs one barrier behind the first

  bf16x8 fa[RBQ][2], fb0[2][2], fb1[2][2];
#define PT_READ_A(SLOTIDX)                                                                                             \
  _Pragma("unroll") for (int r_ = 0; r_ < RBQ; r_++)                                                                   \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++)                                                                \
      fa[r_][ks_] = tr_frag(smem + (SLOTIDX) * SLOT + ks_ * 8192 + aad[0][r_], smem + (SLOTIDX) * SLOT + ks_ * 8192 + aad[1][r_])
#define PT_READ_B(DST, SLOTIDX)                                                                                        \
  _Pragma("unroll") for (int c_ = 0; c_ < 2; c_++)                                                                     \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++)                                                                \
      DST[c_][ks_] = tr_frag(smem + (SLOTIDX) * SLOT + ks_ * 8192 + bad[0][c_], smem + (SLOTIDX) * SLOT + ks_ * 8192 + bad[1][c_])
#define PT_MFMA(FB, I0, J0)                                                                                            \
  _Pragma("unroll") for (int r_ = 0; r_ < RBQ; r_++)                                                                   \
    _Pragma("unroll") for (int c_ = 0; c_ < 2; c_++)                                                                   \
      _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++) mma32(fa[r_][ks_], FB[c_][ks_], acc[(I0) + r_][(J0) + c_])
// CROG_PPT_PROBE (scripts/build_variant.py pptN -DCROG_PPT_PROBE=N, TIMING ONLY - the results are wrong): what does a phase spend its time on?
//   1: no second barrier per phase   2: no LDS-DMA requests after the prologue   3: no fragment reads (the MFMAs run on stale registers)
//   4: no MFMAs (the compiler then drops the fragment reads too: LDS-DMA requests + barriers only; 5 is the same build)
//   6: barriers only   7: MFMAs only (no requests, no reads)
#ifndef CROG_PPT_PROBE
#define CROG_PPT_PROBE 0
#endif
// CROG_PPT_ILV = 1: the fragment reads of the NEXT phase ride between this phase's MFMAs of the same wave instead of opening the next phase
// (B-h1 during phase 0; A-h1 row block r after the last phase-1 MFMA that reads A-h0's block r; the next k-tile's A-h0 likewise in phase 3;
// only B-h0 is still read ahead of its phase).  Deadlines are unchanged: a read in the MFMA part of phase ph touches half-tile <= ph + 2, which
// that phase's vmcnt wait + barrier cover, and a slot is last read >= 2 phases before the request that refills it.
// CROG_PPT_LGKM_LATE = 1 (A/B): the wait for the phase's own fragment reads sits AFTER the phase's first barrier (the reads' latency then overlaps
// the barrier wait).  WAR is unchanged: the request that refills a slot is issued >= 2 phases after the slot's last read (D <= 6).
#ifndef CROG_PPT_LGKM_LATE
#define CROG_PPT_LGKM_LATE 0
#endif
#ifndef CROG_PPT_ILV
#define CROG_PPT_ILV 0
#endif
// CROG_PPT_DMA_MFMA: where a phase's two LDS-DMA requests are issued.  0: both in the load half, behind the fragment reads (rounds 4-5, and the
// forward kernel's order).  1: both by the wave in its MFMA half (between its 8th and 9th MFMA).  **2 (default, round 6)**: one in the load half, the
// other in the MFMA half.  A load half is in-order ISSUE on one wave - 12 transposed reads on average, then requests that queue behind its group's
// 8 KiB on the CU's fill path (32 B/clk = 256 cycles) - against 256 cycles of MFMAs on the partner wave; with one request per half the fill path has
// work from both groups in every barrier-to-barrier segment and a load half queues behind 4 KiB: 1601 / 1633 / 1744 -> 1492 / 1527 / 1632 ns per
// k-tile (754 -> 810 TFLOP/s on 144 CUs; 1: 1603 - no gain; the forward kernel, whose load half has a third of the reads, LOSES 2-3 % with the same
// split: gemm_pp.hip CROG_PP_SPLIT_DMA).  Hazards: a request issued in the MFMA half of phase ph follows that phase's first barrier, so the wait in
// the load half covers one request less per moved request - vmcnt(2 (D - 2) - 1) for 2, vmcnt(2 (D - 3)) for 1, for the same guarantee
// (half-tiles <= ph + 2 landed) - and the refill of a slot moves later, never earlier (legal up to D = 7).
#ifndef CROG_PPT_DMA_MFMA
#define CROG_PPT_DMA_MFMA 2
#endif
#define PT_MFMA_ROW(FB, I0, J0, R)                                                                                     \
  _Pragma("unroll") for (int c_ = 0; c_ < 2; c_++)                                                                     \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++) mma32(fa[R][ks_], FB[c_][ks_], acc[(I0) + (R)][(J0) + c_])
#define PT_READ_A_ROW(SLOTIDX, R)                                                                                      \
  _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++)                                                                  \
    fa[R][ks_] = tr_frag(smem + (SLOTIDX) * SLOT + ks_ * 8192 + aad[0][R], smem + (SLOTIDX) * SLOT + ks_ * 8192 + aad[1][R])
#define PT_LOAD_READS(P, BUF)                                                                                          \
  do {                                                                                                                 \
    if constexpr (CROG_PPT_PROBE != 3 && CROG_PPT_PROBE != 5 && CROG_PPT_PROBE != 7) {                                 \
      if constexpr (CROG_PPT_ILV) {                                                                                    \
        if constexpr ((P) == 0) { PT_READ_B(fb0, (BUF) * 4 + 1); }                                                     \
      } else {                                                                                                         \
        if constexpr ((P) == 0) { PT_READ_A((BUF) * 4 + 0); PT_READ_B(fb0, (BUF) * 4 + 1); }                           \
        if constexpr ((P) == 1) { PT_READ_B(fb1, (BUF) * 4 + 2); }                                                     \
        if constexpr ((P) == 2) { PT_READ_A((BUF) * 4 + 3); }                                                          \
      }                                                                                                                \
    }                                                                                                                  \
  } while (0)
#define PT_LOAD_FILLS(P, BUF, T)                                                                                       \
  do {                                                                                                                 \
    if constexpr (CROG_PPT_PROBE != 2 && CROG_PPT_PROBE != 6 && CROG_PPT_PROBE != 7 && CROG_PPT_DMA_MFMA != 1) {       \
      constexpr int m_ = (P) + D, j_ = m_ & 3, dt_ = m_ >> 2, buf_ = ((BUF) + dt_) & 1;                                 \
      PT_ISSUE_RANGE(j_, buf_, (T) + dt_, 0, CROG_PPT_DMA_MFMA == 2 ? 1 : 2);                                          \
    }                                                                                                                  \
  } while (0)
#define PT_PHASE(P, BUF, T)                                                                                            \
  do {                                                                                                                 \
    PT_LOAD_READS(P, BUF);                                                                                             \
    PT_LOAD_FILLS(P, BUF, T);                                                                                          \
    wait_vmcnt<2 * (D - 2 - (CROG_PPT_DMA_MFMA ? 1 : 0)) + (CROG_PPT_DMA_MFMA == 2 ? 1 : 0)>();                                                          \
    if constexpr (!(CROG_PPT_LGKM_LATE && D <= 6)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    __builtin_amdgcn_s_barrier();                                                                                      \
    if constexpr (CROG_PPT_LGKM_LATE && D <= 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                     \
    if constexpr (CROG_PPT_PROBE != 4 && CROG_PPT_PROBE != 5 && CROG_PPT_PROBE != 6) {                                 \
    if constexpr (CROG_PPT_ILV) {                                                                                      \
      if constexpr ((P) == 0) {                                                                                        \
        PT_MFMA(fb0, 0, 0);                                                                                            \
        PT_READ_B(fb1, (BUF) * 4 + 2);                                                                                 \
        _Pragma("unroll") for (int g_ = 0; g_ < 8; g_++) {                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                           \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                           \
        }                                                                                                              \
      }                                                                                                                \
      if constexpr ((P) == 1) {                                                                                        \
        PT_MFMA_ROW(fb1, 0, 2, 0); PT_READ_A_ROW((BUF) * 4 + 3, 0);                                                    \
        PT_MFMA_ROW(fb1, 0, 2, 1); PT_READ_A_ROW((BUF) * 4 + 3, 1);                                                    \
        PT_MFMA_ROW(fb1, 0, 2, 2); PT_READ_A_ROW((BUF) * 4 + 3, 2);                                                    \
        PT_MFMA_ROW(fb1, 0, 2, 3); PT_READ_A_ROW((BUF) * 4 + 3, 3);                                                    \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; g_++) {                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                           \
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                           \
        }                                                                                                              \
      }                                                                                                                \
      if constexpr ((P) == 2) { PT_MFMA(fb1, RBQ, 2); }                                                                \
      if constexpr ((P) == 3) {                                                                                        \
        PT_MFMA_ROW(fb0, RBQ, 0, 0); PT_READ_A_ROW(((BUF) ^ 1) * 4 + 0, 0);                                            \
        PT_MFMA_ROW(fb0, RBQ, 0, 1); PT_READ_A_ROW(((BUF) ^ 1) * 4 + 0, 1);                                            \
        PT_MFMA_ROW(fb0, RBQ, 0, 2); PT_READ_A_ROW(((BUF) ^ 1) * 4 + 0, 2);                                            \
        PT_MFMA_ROW(fb0, RBQ, 0, 3); PT_READ_A_ROW(((BUF) ^ 1) * 4 + 0, 3);                                            \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; g_++) {                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                           \
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                           \
        }                                                                                                              \
      }                                                                                                                \
    } else if constexpr (CROG_PPT_DMA_MFMA) {                                                                          \
      if constexpr ((P) == 0) { PT_MFMA_ROW(fb0, 0, 0, 0); PT_MFMA_ROW(fb0, 0, 0, 1); }                                \
      if constexpr ((P) == 1) { PT_MFMA_ROW(fb1, 0, 2, 0); PT_MFMA_ROW(fb1, 0, 2, 1); }                                \
      if constexpr ((P) == 2) { PT_MFMA_ROW(fb1, RBQ, 2, 0); PT_MFMA_ROW(fb1, RBQ, 2, 1); }                            \
      if constexpr ((P) == 3) { PT_MFMA_ROW(fb0, RBQ, 0, 0); PT_MFMA_ROW(fb0, RBQ, 0, 1); }                            \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      {                                                                                                                \
        constexpr int m_ = (P) + D, j_ = m_ & 3, dt_ = m_ >> 2, buf_ = ((BUF) + dt_) & 1;                               \
        PT_ISSUE_RANGE(j_, buf_, (T) + dt_, CROG_PPT_DMA_MFMA == 2 ? 1 : 0, 2);                                        \
      }                                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      if constexpr ((P) == 0) { PT_MFMA_ROW(fb0, 0, 0, 2); PT_MFMA_ROW(fb0, 0, 0, 3); }                                \
      if constexpr ((P) == 1) { PT_MFMA_ROW(fb1, 0, 2, 2); PT_MFMA_ROW(fb1, 0, 2, 3); }                                \
      if constexpr ((P) == 2) { PT_MFMA_ROW(fb1, RBQ, 2, 2); PT_MFMA_ROW(fb1, RBQ, 2, 3); }                            \
      if constexpr ((P) == 3) { PT_MFMA_ROW(fb0, RBQ, 0, 2); PT_MFMA_ROW(fb0, RBQ, 0, 3); }                            \
    } else {                                                                                                           \
    if constexpr ((P) == 0) { PT_MFMA(fb0, 0, 0); }                                                                    \
    if constexpr ((P) == 1) { PT_MFMA(fb1, 0, 2); }                                                                    \
    if constexpr ((P) == 2) { PT_MFMA(fb1, RBQ, 2); }                                                                  \
    if constexpr ((P) == 3) { PT_MFMA(fb0, RBQ, 0); }                                                                  \
    }                                                                                                                  \
    }                                                                                                                  \
    __builtin_amdgcn_s_setprio(0);                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    if constexpr (CROG_PPT_PROBE != 1) __builtin_amdgcn_s_barrier();                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  } while (0)

  if constexpr (CROG_PPT_ILV) { PT_READ_A(0); }      // (half-tile 0 landed with the prologue's wait + barrier)
  for (int t = 0; t < nt2; t += 2) {
    PT_PHASE(0, 0, t); PT_PHASE(1, 0, t); PT_PHASE(2, 0, t); PT_PHASE(3, 0, t);
    PT_PHASE(0, 1, t + 1); PT_PHASE(1, 1, t + 1); PT_PHASE(2, 1, t + 1); PT_PHASE(3, 1, t + 1);
  }
#undef PT_PHASE
#undef PT_LOAD_READS
#undef PT_LOAD_FILLS
#undef PT_MFMA
#undef PT_MFMA_ROW
#undef PT_READ_A_ROW
#undef PT_READ_A
#undef PT_READ_B
#undef PT_ISSUE
#undef PT_ISSUE_RANGE
  if (wr == 0) __builtin_amdgcn_s_barrier();      // equal barrier counts for both groups
  wait_vmcnt<0>();

  // ---- epilogue: acc[i][j][e] = C[m0 + 128 (i >> 2) + 64 wr + 16 (i & 3) + 4 g + e][n0 + 128 (j >> 1) + 32 wc + 16 (j & 1) + c]
  float* Cf = reinterpret_cast<float*>(p.C) + (SLAB ? (int64_t)z * p.M * p.ldc : 0);
  const bool interior = m0 + BM <= p.M && n0 + BN <= p.N;
  // (an unsplit reduction: this block is the only writer of its tile - plain read-modify-write instead of 64 K atomic adds)
  const bool solo = !SLAB && p.splitk == 1;
  auto emit = [&](auto guarded, auto alone) {
    constexpr bool G = decltype(guarded)::value, SOLO = decltype(alone)::value;
#pragma unroll
    for (int i = 0; i < RB; i++) {
      const int mrow = m0 + 128 * (i >> 2) + 64 * wr + 16 * (i & 3) + 4 * gq;
#pragma unroll
      for (int j = 0; j < CB; j++) {
        const int ncol = n0 + 128 * (j >> 1) + 32 * wc + 16 * (j & 1) + c16;
        float* cb = Cf + (int64_t)mrow * p.ldc + ncol;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          if (!G || (mrow + e < p.M && ncol < p.N)) {
            if constexpr (SLAB) cb[(int64_t)e * p.ldc] = acc[i][j][e];
            else if constexpr (SOLO) cb[(int64_t)e * p.ldc] += acc[i][j][e];
            else atomicAdd(cb + (int64_t)e * p.ldc, acc[i][j][e]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (solo) {
    if (interior) emit(std::false_type{}, std::true_type{});
    else emit(std::true_type{}, std::true_type{});
  } else {
    if (interior) emit(std::false_type{}, std::false_type{});
    else emit(std::true_type{}, std::false_type{});
  }
}

// The bias gradient of a grouped linear weight gradient: a_sum[m] += sum over this block's reduction slice of A[k][m] for 256 columns.
// 512 threads = 16 rows x 32 lanes of 8 columns per step; the 16 row groups meet in LDS.
__device__ __attribute__((always_inline)) inline void ppt_colsum(const PptProb& p, const int blk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tilesM = (p.M + 255) / 256;
  const int tm = blk % tilesM, z = blk / tilesM;
  const int ktiles = (p.K + 63) / 64, per = (ktiles + p.splitk - 1) / p.splitk;
  const int k0 = z * per * 64, k1 = min(p.K, k0 + per * 64);
  const int tid = threadIdx.x, cg = tid & 31, rg = tid >> 5;
  const int col = tm * 256 + cg * 8;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; e++) s[e] = 0.f;
  if (col < p.M) {      // (M % 8 == 0: a vector is inside the row or outside it)
    const bf16* A = reinterpret_cast<const bf16*>(p.A) + col;
    int k = k0 + rg;
    for (; k + 48 < k1; k += 64) {      // four rows in flight per thread
      const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(A + (int64_t)k * p.lda), v1 = *reinterpret_cast<const bf16x8*>(A + (int64_t)(k + 16) * p.lda);
      const bf16x8 v2 = *reinterpret_cast<const bf16x8*>(A + (int64_t)(k + 32) * p.lda), v3 = *reinterpret_cast<const bf16x8*>(A + (int64_t)(k + 48) * p.lda);
#pragma unroll
      for (int e = 0; e < 8; e++) s[e] += ((float)v0[e] + (float)v1[e]) + ((float)v2[e] + (float)v3[e]);
    }
    for (; k < k1; k += 16) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(A + (int64_t)k * p.lda);
#pragma unroll
      for (int e = 0; e < 8; e++) s[e] += (float)v[e];
    }
  }
  float* red = reinterpret_cast<float*>(smem);      // [16][256]
#pragma unroll
  for (int e = 0; e < 8; e++) red[rg * 256 + cg * 8 + e] = s[e];
  __syncthreads();
  if (tid < 256 && tm * 256 + tid < p.M) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) t += red[r * 256 + tid];
    if (p.splitk == 1) p.a_sum[tm * 256 + tid] += t;
    else atomicAdd(p.a_sum + tm * 256 + tid, t);
  }
}

template <int BL, bool SLAB, int D>
__global__ void __launch_bounds__(512, 2) gemm_ppt_kernel(const crog_gemm_desc p) {
  ppt_body<BL, SLAB, D>(p, (int)blockIdx.x);
}

// Several small weight gradients in ONE launch: block b belongs to the problem whose [start, next start) holds it.  A 512 x 512 x 21632
// gradient alone is 4 tiles: split 16-fold to fill the chip it runs 21 k-tiles per block and spends its time in fills and atomic adds
// (56 us); twelve of them side by side at split 4 take 206 us together (scripts/ab_group.py).
template <int D>
__global__ void __launch_bounds__(512, 2) gemm_ppt_group_kernel(const PptGroup g) {
  const int b = (int)blockIdx.x;
  int i = 0;
  while (i + 1 < g.n && b >= g.p[i + 1].start) i++;
  const PptProb p = g.p[i];
  const int gemm_blocks = ((p.M + 255) / 256) * ((p.N + 255) / 256) * p.splitk;
  if (b - p.start >= gemm_blocks) {      // (block-uniform: only problems with a_sum own such blocks)
    ppt_colsum(p, b - p.start - gemm_blocks);
    return;
  }
  if (p.conv3) ppt_body<CROG_B_NC_IM2COL, false, D>(p, b - p.start);
  else ppt_body<CROG_B_NC, false, D>(p, b - p.start);
}

template <int BL, bool SLAB, int D>
int launch_ppt(const crog_gemm_desc& d, hipStream_t s) {
  #ifdef CROG_PROBE_LDS160
  constexpr int LDS = 160 * 1024;      // probe build: the whole CU's LDS, nothing that uses LDS can share the CU
#else
  constexpr int LDS = 8 * 16384;
#endif
  static bool attr_set = false;
  auto kern = gemm_ppt_kernel<BL, SLAB, D>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      crog_set_error("crog_gemm: hipFuncSetAttribute(%d bytes) failed: %s", LDS, hipGetErrorString(e));
      return CROG_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int nwg = cdiv(d.M, 256) * cdiv(d.N, 256);
  dim3 grid(nwg * d.splitk, 1, 1);
  // CROG_PPT_XCDS=1: whole reduction slices per XCD (ppt_body); 0: the slice-major list cut into eight equal runs
  static const bool whole = [] { const char* e = getenv("CROG_PPT_XCDS"); return e && atoi(e) != 0; }();
  crog_gemm_desc dd = d;
  if (whole && d.splitk > 1) {
    dd.debug |= 1 << 30;
    const int S = d.splitk;
    grid.x = S <= 8 ? 8 * cdiv(nwg, 8 / S) : 8 * cdiv(S, 8) * nwg;
  }
  hipLaunchKernelGGL(kern, grid, dim3(512), LDS, s, dd);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

// out[i] (+)= sum_z ws[z][i] over the slabs of a split-K launch, slice order (the same sum on every run)
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int N, int64_t ldws, float* __restrict__ out,
                                                            int64_t ldo, int accumulate) {
  const int nv = N >> 2;
  const int64_t total = (int64_t)M * nv, slab = (int64_t)M * ldws;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int m = (int)(idx / nv), v = (int)(idx - (int64_t)m * nv);
    const float* src = ws + (int64_t)m * ldws + 4 * v;
    f32x4 s = *reinterpret_cast<const f32x4*>(src);
    for (int zz = 1; zz < splits; zz++) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(src + zz * slab);
      s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
    }
    float* dst = out + (int64_t)m * ldo + 4 * v;
    if (accumulate) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
      s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3];
    }
    *reinterpret_cast<f32x4*>(dst) = s;
  }
}

// The same sum for MANY slabs of a SMALL output (the sliding-window weight gradients: 256 slabs of 64 x 576): one float4 per lane, the
// W waves of a block each add a contiguous run of slabs, wave 0 adds the W partial sums in run order (a fixed order as well).  The
// kernel above would put 256 dependent-latency loads on each of 9216 threads (36 blocks).
template <int W>
__global__ void __launch_bounds__(64 * W) splitk_reduce_par_kernel(const float* __restrict__ ws, int splits, int M, int N, int64_t ldws,
                                                                   float* __restrict__ out, int64_t ldo, int accumulate) {
  __shared__ f32x4 part[W][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nv = N >> 2;
  const int64_t total = (int64_t)M * nv, slab = (int64_t)M * ldws;
  const int64_t idx = (int64_t)blockIdx.x * 64 + lane;
  const bool valid = idx < total;
  const int m = valid ? (int)(idx / nv) : 0, v = valid ? (int)(idx - (int64_t)m * nv) : 0;
  const float* src = ws + (int64_t)m * ldws + 4 * v;
  const int per = (splits + W - 1) / W, z0 = w * per, z1 = min(splits, z0 + per);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int z = z0;
  for (; z + 4 <= z1; z += 4) {
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(src + (int64_t)z * slab), t1 = *reinterpret_cast<const f32x4*>(src + (int64_t)(z + 1) * slab);
    const f32x4 t2 = *reinterpret_cast<const f32x4*>(src + (int64_t)(z + 2) * slab), t3 = *reinterpret_cast<const f32x4*>(src + (int64_t)(z + 3) * slab);
#pragma unroll
    for (int e = 0; e < 4; e++) s[e] = (((s[e] + t0[e]) + t1[e]) + t2[e]) + t3[e];
  }
  for (; z < z1; z++) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(src + (int64_t)z * slab);
#pragma unroll
    for (int e = 0; e < 4; e++) s[e] += t[e];
  }
  part[w][lane] = s;
  __syncthreads();
  if (w != 0 || !valid) return;
  for (int ww = 1; ww < W; ww++) {
    const f32x4 t = part[ww][lane];
#pragma unroll
    for (int e = 0; e < 4; e++) s[e] += t[e];
  }
  float* dst = out + (int64_t)m * ldo + 4 * v;
  if (accumulate) {
    const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
    for (int e = 0; e < 4; e++) s[e] += o[e];
  }
  *reinterpret_cast<f32x4*>(dst) = s;
}

}  // namespace

// Can the ping-pong weight-gradient kernel take this launch?  (the caller has checked dma_eligible)
bool crog_gemm_ppt_eligible(const crog_gemm_desc& d) {
  if (d.dtype != CROG_BF16 || d.a_layout != CROG_A_MC || d.batch != 1) return false;
  if (d.b_layout != CROG_B_NC && d.b_layout != CROG_B_NC_IM2COL) return false;
  if (d.out_mode != CROG_OUT_F32_ATOMIC && !(d.out_mode == CROG_OUT_F32 && d.splitk >= 1)) return false;
  if (d.alpha != 1.f || d.bias || d.R || d.a_sum || d.act != CROG_ACT_NONE || d.col_stats) return false;
  if (d.M % 8 != 0 || d.N % 8 != 0 || d.K < 128) return false;
  if (d.b_layout == CROG_B_NC_IM2COL && (d.convC % 8 != 0 || d.N != 9 * d.convC)) return false;
  if (d.out_mode == CROG_OUT_F32 && d.ldc % 4 != 0) return false;
  if (((long)d.K + 64) * d.lda * 2 >= 0x7fffffffL || ((long)d.K + 64) * d.ldb * 2 >= 0x7fffffffL) return false;
  return true;
}

// Default DMA distance.  Round 4 measured 5 as 4-7 % ahead of 3 / 4 / 6 - with the compiler's vmcnt(0) in front of every phase's transposed reads
// (gemm_dma.h), i.e. with a ring that never held more than one phase of requests.  Without it depth pays stand-alone as designed: 1820 / 1647 /
// 1599 / 1575 / 1543 ns per k-tile at distance 3 / 4 / 5 / 6 / 7 (profiles/r06_ppt_probe.txt); in the step, beside the main chain, 7 is no better
// than 5 (26.74 / 26.69 / 26.61 against 26.55 / 26.67 / 26.49 ms, three interleaved pairs), so 5 stays.  CROG_PPT_DIST=7 overrides (A/B).
static int ppt_default_dist() {
  static const int v = [] { const char* e = getenv("CROG_PPT_DIST"); const int x = e ? atoi(e) : 5; return (x == 5 || x == 7) ? x : 5; }();
  return v;
}

int crog_gemm_ppt_launch(const crog_gemm_desc& d, int dist, hipStream_t s) {
  if (dist == 0) dist = ppt_default_dist();
  const bool conv = d.b_layout == CROG_B_NC_IM2COL, slab = d.out_mode == CROG_OUT_F32;
#define PT_CASE(DD)                                                                                                    \
  if (dist == (DD)) {                                                                                                  \
    if (conv) return slab ? launch_ppt<CROG_B_NC_IM2COL, true, DD>(d, s) : launch_ppt<CROG_B_NC_IM2COL, false, DD>(d, s); \
    return slab ? launch_ppt<CROG_B_NC, true, DD>(d, s) : launch_ppt<CROG_B_NC, false, DD>(d, s);                       \
  }
  PT_CASE(3) PT_CASE(4) PT_CASE(5) PT_CASE(6) PT_CASE(7)
#undef PT_CASE
  crog_set_error("crog_gemm: no ping-pong weight-gradient instantiation for dist=%d", dist);
  return CROG_ERR_ARG;
}

extern "C" int crog_gemm_group(const crog_gemm_desc* descs, int n, crog_stream_t stream) {
  CROG_CHECK_ARG(descs && n >= 1 && n <= PPT_GROUP_MAX, "crog_gemm_group: 1 .. %d descriptors", PPT_GROUP_MAX);
  PptGroup g;
  g.n = n;
  int blocks = 0;
  for (int i = 0; i < n; i++) {
    crog_gemm_desc d = descs[i];
    if (d.batch < 1) d.batch = 1;
    if (d.batch_inner < 1) d.batch_inner = 1;
    if (d.splitk < 1) d.splitk = 1;
    CROG_CHECK_ARG(d.A && d.B && d.C && d.M > 0 && d.N > 0 && d.K > 0, "crog_gemm_group: descriptor %d: null operand or empty size", i);
    float* a_sum = d.a_sum;      // (the single-launch kernel has no a_sum path; the group sums the bias gradient with blocks of its own)
    d.a_sum = nullptr;
    CROG_CHECK_ARG(crog_gemm_ppt_eligible(d) && d.out_mode == CROG_OUT_F32_ATOMIC,
                   "crog_gemm_group: descriptor %d is not a bf16 weight gradient with atomic fp32 output the ping-pong kernel takes "
                   "(A_MC x B_NC / B_NC_IM2COL, M and N multiples of 8, K >= 128, no bias / statistics)", i);
    CROG_CHECK_ARG(!a_sum || d.b_layout == CROG_B_NC, "crog_gemm_group: descriptor %d: a_sum only with the dense (B_NC) form", i);
    CROG_CHECK_ARG(((uintptr_t)d.A % 16) == 0 && ((uintptr_t)d.B % 16) == 0 && d.lda % 8 == 0 && d.ldb % 8 == 0,
                   "crog_gemm_group: descriptor %d: A / B must be 16-byte aligned with lda, ldb multiples of 8", i);
    CROG_CHECK_ARG((long)d.M * d.ldc < 0x7fffffffL && d.lda < 0x7fffffffL && d.ldb < 0x7fffffffL, "crog_gemm_group: descriptor %d: leading dimensions out of range", i);
    PptProb& q = g.p[i];
    q.A = d.A; q.B = d.B; q.C = d.C;
    q.a_sum = a_sum;
    q.M = d.M; q.N = d.N; q.K = d.K;
    q.lda = (int)d.lda; q.ldb = (int)d.ldb; q.ldc = (int)d.ldc;
    q.splitk = d.splitk;
    q.convH = d.convH; q.convW = d.convW; q.convC = d.convC;
    q.conv3 = d.b_layout == CROG_B_NC_IM2COL ? 1 : 0;
    q.start = blocks;
    blocks += cdiv(d.M, 256) * cdiv(d.N, 256) * d.splitk;
    if (a_sum) blocks += cdiv(d.M, 256) * d.splitk;
  }
  g.blocks = blocks;
  #ifdef CROG_PROBE_LDS160
  constexpr int LDS = 160 * 1024;      // probe build: the whole CU's LDS, nothing that uses LDS can share the CU
#else
  constexpr int LDS = 8 * 16384;
#endif
  static bool attr_set5 = false, attr_set7 = false;
  const bool d7 = ppt_default_dist() == 7;
  auto kern = d7 ? gemm_ppt_group_kernel<7> : gemm_ppt_group_kernel<5>;
  bool& attr_set = d7 ? attr_set7 : attr_set5;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      crog_set_error("crog_gemm_group: hipFuncSetAttribute(%d bytes) failed: %s", LDS, hipGetErrorString(e));
      return CROG_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), LDS, (hipStream_t)stream, g);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

extern "C" int crog_splitk_reduce(const float* ws, int splits, int M, int N, int64_t ldws, float* out, int64_t ldo, int accumulate,
                                  crog_stream_t stream) {
  CROG_CHECK_ARG(ws && out && splits >= 1 && M > 0 && N > 0, "crog_splitk_reduce: bad arguments");
  CROG_CHECK_ARG(N % 4 == 0 && ldws % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)ws % 16) == 0 && ((uintptr_t)out % 16) == 0,
                 "crog_splitk_reduce: N, ldws, ldo must be multiples of 4 and the buffers 16-byte aligned");
  const int64_t total = (int64_t)M * (N / 4);
  if (splits >= 16 && total <= 65536) {      // few outputs, many slabs: spread the slabs over the waves of a block too
    if (splits >= 64)
      hipLaunchKernelGGL(splitk_reduce_par_kernel<16>, dim3((unsigned)((total + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, ws, splits, M, N, ldws,
                         out, ldo, accumulate);
    else
      hipLaunchKernelGGL(splitk_reduce_par_kernel<8>, dim3((unsigned)((total + 63) / 64)), dim3(512), 0, (hipStream_t)stream, ws, splits, M, N, ldws,
                         out, ldo, accumulate);
    CROG_LAUNCH_CHECK();
    return CROG_OK;
  }
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ws, splits, M, N, ldws, out, ldo, accumulate);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
