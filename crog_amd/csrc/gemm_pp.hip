// Ping-pong LDS-DMA GEMM for gfx950: the 256 x 256 (or 192 x 256) x 64 tile of the large K-contiguous launches — 3x3 implicit-GEMM
// forward / data gradient and wide 1x1 / linear layers (bf16, lean epilogue: BatchNorm column statistics + bf16 stores).
//
// Why a second main loop beside gemm_dma16_kernel (gemm.hip): there the two waves that share a SIMD run the SAME program in
// lockstep — both issue their LDS-DMA requests and fragment reads together and then compete for the SIMD's one matrix pipe — so
// the pipe idles through every load phase (1.07-1.16 PFLOP/s back to back on the 21632..346112-row forwards).  Here the eight
// waves are two GROUPS (wr = 0 / 1: the two row halves of the tile, one wave of each group per SIMD) that run the same phase
// sequence ONE BARRIER APART: while group 0 issues the 16 (12) MFMAs of a phase, group 1 reads its next fragments and posts its
// LDS-DMA requests, then the roles swap (MI355X_MICROARCH.md "Two waves per SIMD", items 1, 5, 9; cdna_hip_programming.md §5 "The
// 256^2 8-phase template").  A k-tile is 64 deep (128-byte LDS rows = whole cache lines of a K-contiguous operand, half the
// barriers per FLOP of the 32-deep tile) and is cut into four 16-KiB HALF-TILES, the unit of both the DMA pipeline and the phases:
//     j = 0: A-h0   rows [0, 16 RBQ) of BOTH row halves      (the first row quad of every wave)
//     j = 1: B-h0   column blocks 0, 1 of every wave's 64 columns
//     j = 2: B-h1   column blocks 2, 3
//     j = 3: A-h1   rows [16 RBQ, 32 RBQ) of both row halves
// Phase p of k-tile t (global index n = 4 t + p) multiplies one quadrant of the wave's 32 RBQ x 64 output by the whole 64-deep
// k-tile, in snake order so that only ONE operand changes per phase:
//     p = 0: A-h0 x B-h0 (reads 2 RBQ + 4 fragments)   p = 1: A-h0 x B-h1 (4)   p = 2: A-h1 x B-h1 (2 RBQ)   p = 3: A-h1 x B-h0 (0)
//     LOAD(n):  the fragment reads of phase n; the 2 DMA requests of half-tile n + D; s_waitcnt vmcnt(2 (D - 2)), lgkmcnt(0); s_barrier
//     MFMA(n):  s_setprio 1; 4 RBQ MFMAs (v_mfma_f32_16x16x32_bf16); s_setprio 0; s_barrier
// Half-tile m lives in LDS slot m mod 8 (8 x 16 KiB = two k-tiles).  Hazards, by construction (D <= 7):
//   RAW  half-tile m is first read in LOAD(m) (j = 0), LOAD(m - 1) (j = 1, 2, 3); every wave's requests for it are behind the
//        vmcnt at the end of that wave's LOAD(n) for n >= m - 2, and a barrier separates that wait from the first read in either
//        group (the groups are one barrier apart: the wait of LOAD(n) precedes the reads of LOAD(n + 1) in BOTH groups).
//   WAR  half-tile n + D overwrites the slot of half-tile n + D - 8, last read in LOAD(<= n + D - 8); those reads were retired
//        (lgkmcnt(0)) before that phase's barrier, and the lagging group's LOAD(k) ends at the barrier that opens the leading
//        group's LOAD(k + 1): k + 1 <= n needs D <= 7.
// Accumulators / epilogue: as gemm_dma16_kernel — lane (c = lane & 15, g = lane >> 4), acc[i][j][e] = C[16 i + 4 g + e][4 c + j]
// of the wave's tile; the B image is stored de-interleaved so that a lane owns four ADJACENT output columns (one 8-byte store per row).
#include "gemm_dma.h"
#include "comm_dev.h"

namespace {

struct PpGeom { int H, W, C; };

// RBQ: 16-row blocks per row quad (4: 256-row tile, 3: 192-row tile for launches the 256-row tile would quantise badly);
// D: how many half-tiles the DMA runs ahead of the phase that issues it;
// EPI: 0 = the lean epilogue (column statistics + bf16 stores), 1 = also bias, activation, residual, ReLU after the residual
// (crog_gemm's order: + bias, statistics, activation, + R, CROG_ACT_RELU_POST), 2 = the BatchNorm-backward statistics epilogue of a data
// gradient (crog_gemm_desc.bwd_z: the accumulators are a gradient; + R, gate with the forward's ReLU bit mask and / or the sign of
// bwd_ss.scale * z + shift, store the gated value, column sums (sum g, sum g * z) into the replica rows) - kernels of their own so that the
// lean launches carry none of it
template <int AL, int RBQ, int D, int EPI = 0>
__global__ void __launch_bounds__(512, 2) gemm_pp_kernel(const crog_gemm_desc p) {
  static_assert(AL == CROG_A_KC || AL == CROG_A_IM2COL, "K-contiguous A operands");
  static_assert(RBQ == 2 || RBQ == 3 || RBQ == 4, "row quad of 2, 3 or 4 blocks");
  static_assert(D >= 3 && D <= 7, "DMA distance in half-tiles (see the hazard notes)");
  constexpr int BM = 64 * RBQ, BN = 256, BK = 64, RB = 2 * RBQ, CB = 4;
  constexpr int SLOT = 16384;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tilesN = p.N / BN, tilesM = (p.M + BM - 1) / BM;
  int id = blockIdx.x, z = 0;
  xcd_map(tilesM * tilesN, 1, id, z);
  const int tm = id / tilesN, tn = id - tm * tilesN;
  const int m0 = tm * BM, n0 = tn * BN;
  const bf16* A = reinterpret_cast<const bf16*>(p.A);
  const bf16* B = reinterpret_cast<const bf16*>(p.B);
  const PpGeom g{p.convH, p.convW, p.convC};
  const int nt = p.K / BK;              // whole k-tiles (the dispatcher checks K % 64 == 0, and convC % 64 == 0 for the 3x3 form)
  const int nt2 = (nt + 1) & ~1;        // the loop is unrolled over the two k-tile buffers; an odd count runs one all-zero tile

  // ---- DMA side: lane l of piece i (i = 0, 1) of wave w fills slot row r' = 16 w + 8 i + (l >> 3), 16-byte chunk l & 7, from the
  // source chunk (l & 7) ^ f(r'), f(r) = (r >> 1) & 7 (the XOR the fragment reads undo: conflict-free ds_read_b128 on 128-byte rows)
  const unsigned ldab = (unsigned)(p.lda * 2), ldbb = (unsigned)(p.ldb * 2);
  unsigned abase[2], bbase[2], amask[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int rs = 16 * wave + 8 * i + (lane >> 3);
    const unsigned chunk = (unsigned)(((lane & 7) ^ ((rs >> 1) & 7)) * 16);
    // A: slot row -> (row half, row inside the quad); rows >= 16 RBQ of a quad do not exist in the 192-row tile
    const int within = rs & 63;
    const long m = (long)m0 + (rs >> 6) * (BM / 2) + within;        // first row quad (h = 0); the second is 16 RBQ rows further
    const bool exists = within < 16 * RBQ;
    abase[i] = exists ? (unsigned)(m * (long)ldab) + chunk : DMA_OOB;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      unsigned mask = 0;
      if constexpr (AL == CROG_A_IM2COL) {
        const long mh = m + h * 16 * RBQ;
        if (exists && mh < p.M) {
          const int rem = (int)(mh % ((long)g.H * g.W));
          const int py = rem / g.W, px = rem - py * g.W;
#pragma unroll
          for (int tap = 0; tap < 9; tap++) {
            const int sy = py + tap / 3 - 1, sx = px + tap % 3 - 1;
            if (sy >= 0 && sy < g.H && sx >= 0 && sx < g.W) mask |= 1u << tap;
          }
        }
      }
      amask[i][h] = mask;
    }
    // B: slot row -> output column n0 + 64 wc' + 4 c + (2 h + cbp): the de-interleaved image (a lane's four columns are adjacent)
    const int wcs = rs >> 5, cbp = (rs >> 4) & 1, cs = rs & 15;
    const long nrow = (long)n0 + wcs * 64 + 4 * cs + cbp;           // h = 0; h = 1 is two rows (columns of C) further
    bbase[i] = (unsigned)(nrow * (long)ldbb) + chunk;
  }
  // extents for the buffer resources: anything past them reads as zero
  const int exa = (int)(((long)(p.M - 1) * p.lda + (AL == CROG_A_IM2COL ? g.C : p.K)) * 2);
  const int exb = (int)(((long)(p.N - 1) * p.ldb + p.K) * 2);
  const unsigned a_h1 = (unsigned)(16 * RBQ) * ldab, b_h1 = 2u * ldbb;

  // ---- fragment side: lane (c = lane & 15, g = lane >> 4) reads slot row base + c, chunk (4 ks + g) ^ f, f = (c >> 1) & 7
  const int c16 = lane & 15, gq = lane >> 4, fsw = (c16 >> 1) & 7;
  unsigned ra[2], rb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    const unsigned ch = (unsigned)(((4 * ks + gq) ^ fsw) << 4);
    ra[ks] = (unsigned)((wr * 64 + c16) * 128) + ch;
    rb[ks] = (unsigned)((wc * 32 + c16) * 128) + ch;
  }

  f32x4 acc[RB][CB];
#pragma unroll
  for (int i = 0; i < RB; i++)
#pragma unroll
    for (int j = 0; j < CB; j++)
#pragma unroll
      for (int e = 0; e < 4; e++) acc[i][j][e] = 0.f;

  // byte offsets of k-tile `ti` inside a row of A / B (block-uniform; scalar arithmetic)
  auto k_off = [&](int ti, unsigned& ka, unsigned& kb, int& tap) {
    if constexpr (AL == CROG_A_IM2COL) {
      // 64-channel chunk major, tap minor: the nine shifted reads of the same pixels are nine consecutive k-tiles (L2 reuse)
      const int chunk = ti / 9;
      tap = ti - 9 * chunk;
      const int dy = tap / 3 - 1, dx = tap - 3 * (tap / 3) - 1;
      ka = (unsigned)((dy * g.W + dx) * (int)ldab + chunk * 128);      // wrap-around = negative shift
      kb = (unsigned)((tap * g.C + chunk * 64) * 2);
    } else {
      tap = 0;
      ka = kb = (unsigned)(ti * 128);
    }
  };
  // the two requests of half-tile type J of k-tile TI into buffer BUF (J, BUF compile-time).  `live` enters as a mask, not as a
  // branch (a block-uniform branch around a request splits the loop body)
#define PP_ISSUE(J, BUF, TI) PP_ISSUE_RANGE(J, BUF, TI, 0, 2)
#define PP_ISSUE_RANGE(J, BUF, TI, I0, I1)                                                                             \
  do {                                                                                                                 \
    const int ti_ = (TI);                                                                                              \
    const bool lv_ = ti_ < nt;                                                                                         \
    unsigned ka_, kb_;                                                                                                 \
    int tap_;                                                                                                          \
    k_off(ti_, ka_, kb_, tap_);                                                                                        \
    char* dst_ = smem + ((BUF) * 4 + (J)) * SLOT + wave * 2048;                                                        \
    if constexpr ((J) == 0 || (J) == 3) {                                                                              \
      constexpr int h_ = (J) == 3 ? 1 : 0;                                                                             \
      _Pragma("unroll") for (int i_ = (I0); i_ < (I1); i_++) {                                                         \
        bool ok_;                                                                                                      \
        if constexpr (AL == CROG_A_IM2COL) ok_ = ((amask[i_][h_] & (lv_ ? ~0u : 0u)) >> tap_) & 1u;                     \
        else ok_ = lv_;                                                                                                \
        dma_piece(A, exa, dst_ + i_ * 1024, ok_ ? abase[i_] + (h_ ? a_h1 : 0u) + ka_ : DMA_OOB);                       \
      }                                                                                                                \
    } else {                                                                                                           \
      constexpr int h_ = (J) == 2 ? 1 : 0;                                                                             \
      _Pragma("unroll") for (int i_ = (I0); i_ < (I1); i_++)                                                           \
        dma_piece(B, exb, dst_ + i_ * 1024, lv_ ? bbase[i_] + (h_ ? b_h1 : 0u) + kb_ : DMA_OOB);                        \
    }                                                                                                                  \
  } while (0)

  // ---- prologue: half-tiles 0 .. D - 1 (k-tiles 0 and 1), then half-tiles 0 and 1 have landed for every wave
#define PP_PROLOGUE_ONE(M_)                                                                                            \
  if constexpr ((M_) < D) PP_ISSUE((M_) & 3, ((M_) >> 2) & 1, (M_) >> 2)
  PP_PROLOGUE_ONE(0); PP_PROLOGUE_ONE(1); PP_PROLOGUE_ONE(2); PP_PROLOGUE_ONE(3);
  PP_PROLOGUE_ONE(4); PP_PROLOGUE_ONE(5); PP_PROLOGUE_ONE(6);
#undef PP_PROLOGUE_ONE
  wait_vmcnt<2 * (D - 2)>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();      // the second group runs one barrier behind the first (wave-uniform branch)

  bf16x8 fa[RBQ][2], fb0[2][2], fb1[2][2];
#define PP_READ_A(SLOTIDX)                                                                                             \
  _Pragma("unroll") for (int r_ = 0; r_ < RBQ; r_++)                                                                   \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++)                                                                \
      fa[r_][ks_] = *reinterpret_cast<const bf16x8*>(smem + (SLOTIDX) * SLOT + r_ * 2048 + ra[ks_])
#define PP_READ_B(DST, SLOTIDX)                                                                                        \
  _Pragma("unroll") for (int c_ = 0; c_ < 2; c_++)                                                                     \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++)                                                                \
      DST[c_][ks_] = *reinterpret_cast<const bf16x8*>(smem + (SLOTIDX) * SLOT + c_ * 2048 + rb[ks_])
#define PP_MFMA(FB, I0, J0)                                                                                            \
  _Pragma("unroll") for (int r_ = 0; r_ < RBQ; r_++)                                                                   \
    _Pragma("unroll") for (int c_ = 0; c_ < 2; c_++)                                                                   \
      _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++) mma32(fa[r_][ks_], FB[c_][ks_], acc[(I0) + r_][(J0) + c_])
  // CROG_PP_LGKM_LATE = 1 (A/B): the wait for the phase's own fragment reads sits AFTER the phase's first barrier, so the reads' latency
  // overlaps the barrier wait.  The WAR argument above then needs the refill >= 2 phases behind the last read: D <= 6.
#ifndef CROG_PP_LGKM_LATE
#define CROG_PP_LGKM_LATE 0
#endif
  constexpr bool LGKM_LATE = CROG_PP_LGKM_LATE != 0 && D <= 6;
  // CROG_PP_SPLIT_DMA = 1: the second of a phase's two requests is issued by the wave in its MFMA half (after half of its MFMAs) instead of in the
  // load half - the CU's fill path then has a request from each group in every barrier-to-barrier segment, and a load half queues behind 4 KiB of its
  // group's requests instead of 8 (gemm_ppt.hip: + 7 % there).  The wait in the load half then covers one request less: vmcnt(2 (D - 2) - 1).
#ifndef CROG_PP_SPLIT_DMA
#define CROG_PP_SPLIT_DMA 0
#endif
  constexpr bool SPLIT_DMA = CROG_PP_SPLIT_DMA != 0;
#define PP_MFMA_ROWS(FB, I0, J0, R0, R1)                                                                               \
  _Pragma("unroll") for (int r_ = (R0); r_ < (R1); r_++)                                                               \
    _Pragma("unroll") for (int c_ = 0; c_ < 2; c_++)                                                                   \
      _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++) mma32(fa[r_][ks_], FB[c_][ks_], acc[(I0) + r_][(J0) + c_])
  // one phase: P = 0 .. 3, BUF = buffer of k-tile T (compile-time), T = k-tile index
#define PP_PHASE(P, BUF, T)                                                                                            \
  do {                                                                                                                 \
    if constexpr ((P) == 0) { PP_READ_A((BUF) * 4 + 0); PP_READ_B(fb0, (BUF) * 4 + 1); }                               \
    if constexpr ((P) == 1) { PP_READ_B(fb1, (BUF) * 4 + 2); }                                                         \
    if constexpr ((P) == 2) { PP_READ_A((BUF) * 4 + 3); }                                                              \
    {                                                                                                                  \
      constexpr int m_ = (P) + D, j_ = m_ & 3, dt_ = m_ >> 2, buf_ = ((BUF) + dt_) & 1;                                 \
      PP_ISSUE_RANGE(j_, buf_, (T) + dt_, 0, SPLIT_DMA ? 1 : 2);                                                       \
    }                                                                                                                  \
    wait_vmcnt<2 * (D - 2) - (SPLIT_DMA ? 1 : 0)>();                                                                   \
    if constexpr (!LGKM_LATE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    __builtin_amdgcn_s_barrier();                                                                                      \
    if constexpr (LGKM_LATE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                     \
    if constexpr (SPLIT_DMA) {                                                                                         \
      if constexpr ((P) == 0) { PP_MFMA_ROWS(fb0, 0, 0, 0, RBQ / 2); }                                                 \
      if constexpr ((P) == 1) { PP_MFMA_ROWS(fb1, 0, 2, 0, RBQ / 2); }                                                 \
      if constexpr ((P) == 2) { PP_MFMA_ROWS(fb1, RBQ, 2, 0, RBQ / 2); }                                               \
      if constexpr ((P) == 3) { PP_MFMA_ROWS(fb0, RBQ, 0, 0, RBQ / 2); }                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      {                                                                                                                \
        constexpr int m_ = (P) + D, j_ = m_ & 3, dt_ = m_ >> 2, buf_ = ((BUF) + dt_) & 1;                               \
        PP_ISSUE_RANGE(j_, buf_, (T) + dt_, 1, 2);                                                                     \
      }                                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      if constexpr ((P) == 0) { PP_MFMA_ROWS(fb0, 0, 0, RBQ / 2, RBQ); }                                               \
      if constexpr ((P) == 1) { PP_MFMA_ROWS(fb1, 0, 2, RBQ / 2, RBQ); }                                               \
      if constexpr ((P) == 2) { PP_MFMA_ROWS(fb1, RBQ, 2, RBQ / 2, RBQ); }                                             \
      if constexpr ((P) == 3) { PP_MFMA_ROWS(fb0, RBQ, 0, RBQ / 2, RBQ); }                                             \
    } else {                                                                                                           \
    if constexpr ((P) == 0) { PP_MFMA(fb0, 0, 0); }                                                                    \
    if constexpr ((P) == 1) { PP_MFMA(fb1, 0, 2); }                                                                    \
    if constexpr ((P) == 2) { PP_MFMA(fb1, RBQ, 2); }                                                                  \
    if constexpr ((P) == 3) { PP_MFMA(fb0, RBQ, 0); }                                                                  \
    }                                                                                                                  \
    __builtin_amdgcn_s_setprio(0);                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    __builtin_amdgcn_s_barrier();                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  } while (0)

  for (int t = 0; t < nt2; t += 2) {
    PP_PHASE(0, 0, t); PP_PHASE(1, 0, t); PP_PHASE(2, 0, t); PP_PHASE(3, 0, t);
    PP_PHASE(0, 1, t + 1); PP_PHASE(1, 1, t + 1); PP_PHASE(2, 1, t + 1); PP_PHASE(3, 1, t + 1);
  }
#undef PP_PHASE
#undef PP_MFMA
#undef PP_READ_A
#undef PP_READ_B
#undef PP_ISSUE
#undef PP_ISSUE_RANGE
#undef PP_MFMA_ROWS
  if (wr == 0) __builtin_amdgcn_s_barrier();      // the leading group waits for the lagging one (equal barrier counts)
  // the trailing out-of-range requests write zeros into the ring and the epilogue reuses it: every wave's requests must have landed
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // ---- lean epilogue (as gemm_dma16_kernel): BatchNorm column statistics, bf16 stores -------------------------------
  const int arow = wr * (BM / 2), brow = wc * 64;
  const bool rows_in = m0 + BM <= p.M;
  if constexpr (EPI == 1) {
    if (p.bias) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n0 + brow + 4 * c16);      // (N % 256 == 0: always in range, 16-byte aligned)
#pragma unroll
      for (int i = 0; i < RB; i++)
#pragma unroll
        for (int j = 0; j < CB; j++)
#pragma unroll
          for (int e = 0; e < 4; e++) acc[i][j][e] += bv[j];
    }
  }
  if (p.col_stats) {
    float s1[CB], s2[CB];
#pragma unroll
    for (int j = 0; j < CB; j++) s1[j] = s2[j] = 0.f;
    if constexpr (EPI == 2) {
      // the lane's four adjacent columns of a row are one 8-byte load of z (and of R), one mask byte holds their four bits
      const int colz = n0 + wc * 64 + 4 * c16;
      const bf16* Z = reinterpret_cast<const bf16*>(p.bwd_z);
      const bf16* Rz = reinterpret_cast<const bf16*>(p.R);
      const unsigned char* MK = p.bwd_mask;
      float gsc[CB], gsh[CB];
#pragma unroll
      for (int j = 0; j < CB; j++) { gsc[j] = p.bwd_ss ? p.bwd_ss[2 * (colz + j)] : 0.f; gsh[j] = p.bwd_ss ? p.bwd_ss[2 * (colz + j) + 1] : 1.f; }      // (no ReLU: 0 * z + 1 > 0)
      const int mbit = colz & 7;
#pragma unroll
      for (int i = 0; i < RB; i++) {
        const int row0 = m0 + arow + i * 16 + 4 * gq;
        bf16x4 zv[4], rv[4];
        unsigned mb[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {      // the block's loads in flight before the first is used
          const bool ok = rows_in || row0 + e < p.M;
          const int64_t r = ok ? row0 + e : 0;
          zv[e] = *reinterpret_cast<const bf16x4*>(Z + r * p.ldz + colz);
          if (Rz) rv[e] = *reinterpret_cast<const bf16x4*>(Rz + r * p.ldr + colz);
          mb[e] = MK ? (unsigned)MK[r * (p.N >> 3) + (colz >> 3)] : 0xffu;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const bool ok = rows_in || row0 + e < p.M;
#pragma unroll
          for (int j = 0; j < CB; j++) {
            const float z = (float)zv[e][j];
            float a = acc[i][j][e];
            if (Rz) a += (float)rv[e][j];
            a = ((mb[e] >> (mbit + j)) & 1u) ? a : 0.f;
            a = (z * gsc[j] + gsh[j] > 0.f) ? a : 0.f;
            a = ok ? a : 0.f;
            acc[i][j][e] = a;
            s1[j] += a;
            s2[j] += a * z;
          }
        }
      }
    } else {
#pragma unroll
    for (int i = 0; i < RB; i++) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const bool ok = rows_in || m0 + arow + i * 16 + 4 * gq + e < p.M;
#pragma unroll
        for (int j = 0; j < CB; j++) { const float v = ok ? acc[i][j][e] : 0.f; s1[j] += v; s2[j] += v * v; }
      }
    }
    }
    float* red = reinterpret_cast<float*>(smem);      // [2 row halves][BN][2]
#pragma unroll
    for (int j = 0; j < CB; j++) {
      s1[j] = xor32_sum(xor16_sum(s1[j]));
      s2[j] = xor32_sum(xor16_sum(s2[j]));
    }
    if (gq == 0) {
#pragma unroll
      for (int j = 0; j < CB; j++) {
        const int col = brow + 4 * c16 + j;
        red[(wr * BN + col) * 2 + 0] = s1[j];
        red[(wr * BN + col) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    // one thread per float of the tile's [BN][2] statistics rows: a wave's adds / stores are 256 contiguous bytes.  The 256-row tile
    // owns two 128-row slab rows (one per row half); the 192-row tile only runs with replicas (any row of the pre-zeroed buffer will do).
    for (int idx = tid; idx < 2 * BN * 2; idx += 512) {
      const int sl = idx / (BN * 2), k = idx - sl * (BN * 2);
      if (m0 + sl * (BM / 2) < p.M) {
        const float v = red[sl * BN * 2 + k];
        if (p.stat_replicas > 0) atomicAdd(p.col_stats + ((int64_t)((2 * tm + sl) % p.stat_replicas) * p.N + n0) * 2 + k, v);
        else p.col_stats[((int64_t)(m0 / 128 + sl) * p.N + n0) * 2 + k] = v;
      }
    }
    if constexpr (EPI == 2) {      // SyncBatchNorm backward: the last block exchanges the totals (crog_gemm_desc.stat_sync, comm_dev.h)
      if (p.stat_sync) crog_stat_sync_tail(reinterpret_cast<const CrogSyncBlock*>(p.stat_sync), p.col_stats, p.stat_replicas, 2 * p.N, gridDim.x);
    }
  }
  bf16* C = reinterpret_cast<bf16*>(p.C);
  const int col = n0 + brow + 4 * c16;
  if constexpr (EPI == 1) {
    if (p.act == CROG_ACT_RELU || p.act == CROG_ACT_QUICKGELU || p.act == CROG_ACT_TANH) {
#pragma unroll
      for (int i = 0; i < RB; i++)
#pragma unroll
        for (int j = 0; j < CB; j++)
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float v = acc[i][j][e];
            acc[i][j][e] = p.act == CROG_ACT_RELU ? fmaxf(v, 0.f) : p.act == CROG_ACT_QUICKGELU ? act_quickgelu(v) : act_tanh(v);
          }
    }
  }
  const bf16* R = EPI == 1 ? reinterpret_cast<const bf16*>(p.R) : nullptr;      // (EPI == 2 has added R before its gate)
  const bool post = EPI == 1 && p.act == CROG_ACT_RELU_POST;
#pragma unroll
  for (int i = 0; i < RB; i++) {
    const int row0 = m0 + arow + i * 16 + 4 * gq;
    bf16* cb = C + (int64_t)row0 * p.ldc + col;
    bf16x4 rv[4];
    if constexpr (EPI == 1) {
      if (R) {      // the block's four residual rows in flight before the first is used; added in fp32 (one rounding)
        const bf16* rb = R + (int64_t)row0 * p.ldr + col;
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (rows_in || row0 + e < p.M) rv[e] = *reinterpret_cast<const bf16x4*>(rb + (int64_t)e * p.ldr);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float f[4] = {acc[i][0][e], acc[i][1][e], acc[i][2][e], acc[i][3][e]};
      if constexpr (EPI == 1) {
        if (R && (rows_in || row0 + e < p.M)) {
#pragma unroll
          for (int j = 0; j < 4; j++) f[j] += (float)rv[e][j];
        }
        if (post) {
#pragma unroll
          for (int j = 0; j < 4; j++) f[j] = fmaxf(f[j], 0.f);
        }
      }
      bf16x4 v;
      v[0] = (bf16)f[0]; v[1] = (bf16)f[1]; v[2] = (bf16)f[2]; v[3] = (bf16)f[3];
      if (rows_in || row0 + e < p.M) *reinterpret_cast<bf16x4*>(cb + (int64_t)e * p.ldc) = v;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (EPI != 2) {
    // SyncBatchNorm FORWARD: the block that takes the last ticket adds the replica rows of (sum x, sum x^2) up and exchanges them with the other
    // ranks (crog_gemm_desc.stat_sync, comm_dev.h) - at the very END of the kernel, where no accumulator is live any more
    if (p.stat_sync && p.col_stats) crog_stat_sync_tail(reinterpret_cast<const CrogSyncBlock*>(p.stat_sync), p.col_stats, p.stat_replicas, 2 * p.N, gridDim.x);
  }
}

template <int AL, int RBQ, int D, int EPI = 0>
int launch_pp(const crog_gemm_desc& d, hipStream_t s) {
  #ifdef CROG_PROBE_LDS160
  constexpr int LDS = 160 * 1024;      // probe build: the whole CU's LDS, nothing that uses LDS can share the CU
#else
  constexpr int LDS = 8 * 16384;
#endif
  static bool attr_set = false;
  auto kern = gemm_pp_kernel<AL, RBQ, D, EPI>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      crog_set_error("crog_gemm: hipFuncSetAttribute(%d bytes) failed: %s", LDS, hipGetErrorString(e));
      return CROG_ERR_LAUNCH;
    }
    attr_set = true;
  }
  dim3 grid(cdiv(d.M, 64 * RBQ) * (d.N / 256), 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(512), LDS, s, d);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

}  // namespace

// Can the ping-pong kernel take this launch?  (bf16, K-contiguous operands, plain bf16 epilogue; the caller — dispatch_shape in
// gemm.hip — has already checked lean_epilogue_ok and dma_eligible.)  rows: 256 or 192.
bool crog_gemm_pp_eligible(const crog_gemm_desc& d, int rows) {
  if (d.dtype != CROG_BF16 || d.b_layout != CROG_B_KC || d.batch != 1 || d.splitk != 1) return false;
  if (d.a_layout != CROG_A_KC && d.a_layout != CROG_A_IM2COL) return false;
  if (d.N % 256 != 0 || d.K % 64 != 0 || d.K < 128) return false;
  if (d.a_layout == CROG_A_IM2COL && d.convC % 64 != 0) return false;
  if (rows != 256 && d.col_stats && d.stat_replicas <= 0) return false;      // the 128-row slab rows only line up with the 256-row tile's halves
  // 32-bit byte offsets, with the rows of the last (ragged) tile included
  if (((long)d.M + 256) * d.lda * 2 >= 0x7fffffffL || (long)d.N * d.ldb * 2 >= 0x7fffffffL) return false;
  return rows == 256 || rows == 192 || rows == 128;
}

// The full epilogue the EPI instantiations carry: alpha 1, bf16 output, optional bias / activation / residual (16-byte aligned rows).
bool crog_gemm_pp_full_epilogue_ok(const crog_gemm_desc& d) {
  if (d.alpha != 1.f || d.out_mode != CROG_OUT_T || d.dtype != CROG_BF16 || d.bwd_z) return false;
  if (d.R && (d.ldr % 4 != 0 || ((uintptr_t)d.R % 8) != 0)) return false;
  if (d.bias && ((uintptr_t)d.bias % 16) != 0) return false;
  return true;
}

// The BatchNorm-backward statistics epilogue (EPI = 2): what crog_gemm checks for bwd_z, plus rows the 8-byte z / R loads can take
bool crog_gemm_pp_bwdz_ok(const crog_gemm_desc& d) {
  if (!d.bwd_z || !d.col_stats || d.stat_replicas <= 0 || d.alpha != 1.f || d.bias || d.act != CROG_ACT_NONE || d.out_mode != CROG_OUT_T) return false;
  if (d.ldz % 4 != 0 || ((uintptr_t)d.bwd_z % 8) != 0) return false;
  if (d.R && (d.ldr % 4 != 0 || ((uintptr_t)d.R % 8) != 0)) return false;
  return true;
}

// dist: DMA distance in half-tiles (3 .. 7; 0 = the default of the tile height); full: 1 = the launch needs the EPI epilogue, 2 = the
// bwd_z epilogue (default distances only)
int crog_gemm_pp_launch(const crog_gemm_desc& d, int rows, int dist, hipStream_t s, int full) {
  const bool conv = d.a_layout == CROG_A_IM2COL;
  if (full == 2) {
    if (rows == 256) return conv ? launch_pp<CROG_A_IM2COL, 4, 4, 2>(d, s) : launch_pp<CROG_A_KC, 4, 5, 2>(d, s);
    if (rows == 192) return conv ? launch_pp<CROG_A_IM2COL, 3, 4, 2>(d, s) : launch_pp<CROG_A_KC, 3, 5, 2>(d, s);
    if (rows == 128) return conv ? launch_pp<CROG_A_IM2COL, 2, 4, 2>(d, s) : launch_pp<CROG_A_KC, 2, 5, 2>(d, s);
    crog_set_error("crog_gemm: no ping-pong instantiation with the bwd_z epilogue for rows=%d", rows);
    return CROG_ERR_ARG;
  }
  if (full) {
    if (rows == 256) return conv ? launch_pp<CROG_A_IM2COL, 4, 4, 1>(d, s) : launch_pp<CROG_A_KC, 4, 5, 1>(d, s);
    if (rows == 192) return conv ? launch_pp<CROG_A_IM2COL, 3, 4, 1>(d, s) : launch_pp<CROG_A_KC, 3, 5, 1>(d, s);
    if (rows == 128) return conv ? launch_pp<CROG_A_IM2COL, 2, 4, 1>(d, s) : launch_pp<CROG_A_KC, 2, 5, 1>(d, s);
    crog_set_error("crog_gemm: no ping-pong instantiation with the full epilogue for rows=%d", rows);      // (never fall through to a lean kernel)
    return CROG_ERR_ARG;
  }
  // measured (scripts/ab_pp.py, distances 3 .. 7 on every 3x3 / 1x1 shape of the step): 4 for the 3x3 form (1333-1369 TFLOP/s on the
  // >= 676-tile forwards; 5: -1 %, 6: -6 %, 3: -11 %), 5 for the 1x1 / linear form (its k-loops are 4-32 tiles: the ring fill counts)
  if (dist == 0) dist = conv ? 4 : 5;
#define PP_CASE(R, DD)                                                                      \
  if (rows == 64 * (R) && dist == (DD))                                                     \
    return conv ? launch_pp<CROG_A_IM2COL, R, DD>(d, s) : launch_pp<CROG_A_KC, R, DD>(d, s)
  PP_CASE(4, 3); PP_CASE(4, 4); PP_CASE(4, 5); PP_CASE(4, 6); PP_CASE(4, 7);
  PP_CASE(3, 3); PP_CASE(3, 4); PP_CASE(3, 5); PP_CASE(3, 6); PP_CASE(3, 7);
  PP_CASE(2, 4); PP_CASE(2, 5);      // 128-row tile: default distances only
#undef PP_CASE
  crog_set_error("crog_gemm: no ping-pong instantiation for rows=%d dist=%d", rows, dist);
  return CROG_ERR_ARG;
}
