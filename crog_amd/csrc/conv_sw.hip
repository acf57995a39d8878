// Sliding-window 3x3 convolution for SMALL channel counts (gfx950): the stem's 32 / 64-channel layers at 208 x 208 and layer1's
// 64 -> 64 at 104 x 104 (clip.py:44-57, 208-213), forward and - on the [Cin][flipped tap][Cout] copy of the weights - data gradient.
//
// Why a kernel of its own: as an implicit GEMM these launches have K = 9 C = 288 ... 576 and N = 32 / 64, and every output tile
// fetches its A operand through nine shifted L2 -> LDS passes: 346112 x 64 x 576 moves 400 MB into LDS for 44 MB of activations and
// runs at 60 us against 20 us of HBM time; the 1.38 M-pixel stem layers at 123-132 us against 40-59.  Here
//   * a workgroup walks DOWN a strip of image rows and keeps the last four input rows in an LDS ring (each row, with its two zero
//     border pixels, is fetched ONCE per strip: 16-byte global loads a row ahead of the row being computed, ds_write_b128 into the
//     ring, one barrier per output row); a fifth, all-zero row stands in for the rows above and below the image;
//   * the WEIGHTS are the MFMA's A operand and live in registers for the whole launch: wave (cg, ps) holds the nine taps of 16
//     output channels (36 / 72 VGPRs for Cin = 32 / 64) and streams 16-pixel blocks of the row through v_mfma_f32_16x16x32_bf16 with
//     the pixel fragment as B operand, read from the ring at the tap's shift (ds_read_b128; 64- / 128-byte pixel rows with the
//     16-byte chunk index XOR-swizzled by (pixel >> 2) & 3 / (pixel >> 1) & 7);
//   * D = W . X^T puts four ADJACENT output channels of one pixel into a lane: 8-byte stores, and the BatchNorm (sum, sum of squares)
//     of the lean epilogue are per-lane running sums over the whole strip - one shuffle tree, one LDS exchange and Cout x 2 atomic
//     adds per workgroup at the very end.
// Lean launches only (alpha 1, no bias / activation / residual, statistics in replica mode or none); everything else stays with
// crog_gemm's implicit-GEMM kernels.  Dispatch: crog_gemm -> crog_conv_sw_eligible (below).
#include "gemm_dma.h"
#include "comm_dev.h"
#include <algorithm>

namespace {

constexpr int SW_NT = 512, SW_WAVES = 8;

template <int CI>
__device__ __attribute__((always_inline)) inline unsigned sw_swz(unsigned pixel) {
  return CI == 64 ? ((pixel >> 1) & 7u) : ((pixel >> 2) & 3u);
}

// CI, CO in {32, 64}.  grid.x workgroups, each a strip of `rows_per_wg` consecutive rows of the [B * H] row space, walked in groups of G
// output rows: one barrier and one batch of row fetches per group (a single row's products - half a microsecond - cannot hide a
// global load).  Ring of NS = 2 G + 2 row slots (input row r lives in slot r & (NS - 1)) + the zero row.
template <int CI, int CO, int G>
__global__ void __launch_bounds__(SW_NT, 1) conv_sw_kernel(const crog_gemm_desc p, int rows_per_wg) {
  constexpr int CPP = CI / 8;                  // 16-byte chunks per pixel
  constexpr int KS = CI / 32;                  // MFMA k-steps per tap
  constexpr int COG = CO / 16;                 // 16-channel output groups
  // ... of which a wave holds NCG in registers (all of them at Cin = 32, two at Cin = 64: 144 VGPRs of weights at most): every pixel
  // fragment read from LDS then feeds NCG MFMAs - with one group per wave the kernel was LDS-bandwidth-bound (1 KiB per MFMA and wave:
  // 256 B per clock and CU against the 128 the LDS delivers; 64 -> 64 ran at 60 us, no faster than the implicit GEMM)
  constexpr int NCG = CI == 32 ? COG : 2;
  constexpr int CW = COG / NCG;                // waves across the output channels
  constexpr int PS = SW_WAVES / CW;            // pixel streams (waves that share their output channels)
  constexpr int NS = G == 1 ? 4 : 8;
  static_assert(2 * G + 2 <= NS, "ring too small for the group");
  constexpr int LPT = G == 1 ? 4 : 5;          // 16-byte chunks a thread moves per batch of G rows (the launcher checks G * W * CPP <= LPT * 512)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int H = p.convH, W = p.convW;
  const int total_rows = p.M / W;              // B * H
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave % CW, ps = wave / CW;
  const int l15 = lane & 15, g4 = lane >> 4;
  const unsigned RS = (unsigned)(W + 2) * CI * 2;          // bytes of a ring row (pixel slots 0 and W + 1 stay zero)
  char* ring = smem;
  char* zrow = smem + NS * RS;                             // the all-zero row

  const int r0 = blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, total_rows);
  if (r0 >= r1) {      // (a workgroup without rows - the launcher creates none - still takes its ticket in the statistics' exchange)
    if (p.stat_sync && p.col_stats && p.stat_replicas > 0 && !p.bwd_z)
      crog_stat_sync_tail(reinterpret_cast<const CrogSyncBlock*>(p.stat_sync), p.col_stats, p.stat_replicas, 2 * p.N, gridDim.x * gridDim.y * gridDim.z);
    return;
  }

  // zero the border pixels of the ring rows and the whole zero row (once)
  for (unsigned i = tid * 16u; i < (unsigned)(NS + 1) * RS; i += SW_NT * 16u) {
    const unsigned row = i / RS, off = i - row * RS;
    if (row == NS || off < (unsigned)CI * 2 || off >= RS - (unsigned)CI * 2) *reinterpret_cast<f32x4*>(smem + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // weights: A fragments of the wave's 16 output channels, all nine taps.  B_KC layout: W[co][tap * CI + ci], row stride ldb
  const bf16* Wt = reinterpret_cast<const bf16*>(p.B);
  bf16x8 wreg[NCG][9][KS];
#pragma unroll
  for (int n = 0; n < NCG; n++) {
    const bf16* wrow = Wt + (int64_t)((cw * NCG + n) * 16 + l15) * p.ldb + g4 * 8;
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int ks = 0; ks < KS; ks++) wreg[n][t][ks] = *reinterpret_cast<const bf16x8*>(wrow + t * CI + ks * 32);
  }

  const bf16* X = reinterpret_cast<const bf16*>(p.A);
  const unsigned row_chunks = (unsigned)W * CPP;           // 16-byte chunks of an input row
  const int last_in = min(r1, total_rows - 1);             // last input row this strip reads
  f32x4 stage[LPT];

  // rows first .. first + count - 1 (those inside [0, last_in]) -> registers / registers -> their ring slots
  auto fetch = [&](int first, int count) {
#pragma unroll
    for (int i = 0; i < LPT; i++) {
      const unsigned q = (unsigned)tid + (unsigned)i * SW_NT;
      const unsigned rr = q / row_chunks, c = q - rr * row_chunks;
      const int gr = first + (int)rr;
      // UNCONDITIONAL loads (a row that is not wanted reads row r0 instead and is dropped by commit): under a per-load branch the
      // compiler waited for each load before issuing the next - five serialised round trips to memory per batch
      const int gs = ((int)rr < count && gr >= 0 && gr <= last_in) ? gr : r0;
      stage[i] = *reinterpret_cast<const f32x4*>(X + ((int64_t)gs * row_chunks + c) * 8);
    }
  };
  auto commit = [&](int first, int count) {
#pragma unroll
    for (int i = 0; i < LPT; i++) {
      const unsigned q = (unsigned)tid + (unsigned)i * SW_NT;
      const unsigned rr = q / row_chunks, c = q - rr * row_chunks;
      const int gr = first + (int)rr;
      if ((int)rr < count && gr >= 0 && gr <= last_in) {
        const unsigned px = c / CPP + 1, ch = c % CPP;
        *reinterpret_cast<f32x4*>(ring + (unsigned)(gr & (NS - 1)) * RS + px * (CI * 2) + ((ch ^ sw_swz<CI>(px)) << 4)) = stage[i];
      }
    }
  };

  // prologue: input rows r0 - 1 .. r0 + G; in the loop the NEXT group's new rows (a + G + 1 .. a + 2 G) are fetched before
  // this group's products and committed after them - their slots held rows a - G - 1 ... a - 2, last read by the previous group, which
  // every wave has left (the barrier that ended that iteration)
  for (int f = r0 - 1; f <= r0 + G; f += G) {              // (batches of at most G rows: what the staging registers hold)
    const int cnt = min(G, r0 + G - f + 1);
    fetch(f, cnt);
    commit(f, cnt);
  }
  __syncthreads();

  float s1[NCG][4], s2[NCG][4];
#pragma unroll
  for (int n = 0; n < NCG; n++)
#pragma unroll
    for (int e = 0; e < 4; e++) s1[n][e] = s2[n][e] = 0.f;
  bf16* Cout = reinterpret_cast<bf16*>(p.C);
  const int nblk = (W + 15) >> 4;
  const bool stats = p.col_stats != nullptr;

  for (int a = r0; a < r1; a += G) {
    fetch(a + G + 1, G);
    const int nrow = min(G, r1 - a);
    for (int u = ps; u < nrow * nblk; u += PS) {
      const int ri = u / nblk, pb = u - ri * nblk;
      const int gr = a + ri, y = gr % H;
      const char* rows[3];
      rows[0] = y > 0 ? ring + (unsigned)((gr - 1) & (NS - 1)) * RS : zrow;
      rows[1] = ring + (unsigned)(gr & (NS - 1)) * RS;
      rows[2] = y + 1 < H ? ring + (unsigned)((gr + 1) & (NS - 1)) * RS : zrow;
      f32x4 acc[NCG];
#pragma unroll
      for (int n = 0; n < NCG; n++) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned px0 = (unsigned)(pb * 16 + l15);      // output pixel of this lane's B column; tap dx reads ring pixel px0 + dx
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const int dy = t / 3, dx = t % 3;                  // tap (dy - 1, dx - 1)
        const unsigned rp = min(px0 + (unsigned)dx, (unsigned)(W + 1));      // (lanes past the row's end read the zero border)
        const char* src = rows[dy] + rp * (CI * 2);
        const unsigned sw = sw_swz<CI>(rp);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(src + ((((unsigned)(ks * 4 + g4)) ^ sw) << 4));
#pragma unroll
          for (int n = 0; n < NCG; n++) mma32(wreg[n][t][ks], xf, acc[n]);
        }
      }
      // acc[n][e] = out[pixel px0][channel (cw * NCG + n) * 16 + 4 g4 + e]
      if (px0 < (unsigned)W) {
        bf16* crow = Cout + ((int64_t)gr * W + px0) * p.ldc + cw * NCG * 16 + 4 * g4;
#pragma unroll
        for (int n = 0; n < NCG; n++) {
          if (stats) {
#pragma unroll
            for (int e = 0; e < 4; e++) { s1[n][e] += acc[n][e]; s2[n][e] += acc[n][e] * acc[n][e]; }
          }
          bf16x4 v;
          v[0] = (bf16)acc[n][0]; v[1] = (bf16)acc[n][1]; v[2] = (bf16)acc[n][2]; v[3] = (bf16)acc[n][3];
          *reinterpret_cast<bf16x4*>(crow + n * 16) = v;
        }
      }
    }
    commit(a + G + 1, G);
    __syncthreads();
  }

  if (stats) {
    // lanes with the same g4 hold the same channels: fold the 16 pixel lanes, then the PS waves that share them through LDS
#pragma unroll
    for (int n = 0; n < NCG; n++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        s1[n][e] = row16_sum(s1[n][e]);
        s2[n][e] = row16_sum(s2[n][e]);
      }
    float* red = reinterpret_cast<float*>(smem);           // [PS][CO][2]; the loop's last barrier has passed: the ring is dead
    if (l15 == 0) {
#pragma unroll
      for (int n = 0; n < NCG; n++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int co = (cw * NCG + n) * 16 + 4 * g4 + e;
          red[(ps * CO + co) * 2 + 0] = s1[n][e];
          red[(ps * CO + co) * 2 + 1] = s2[n][e];
        }
    }
    __syncthreads();
    if (tid < CO * 2) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < PS; q++) v += red[q * CO * 2 + tid];
      atomicAdd(p.col_stats + (int64_t)(blockIdx.x % p.stat_replicas) * p.N * 2 + tid, v);
    }
  }
  // SyncBatchNorm forward statistics: the last block exchanges the totals (crog_gemm_desc.stat_sync, comm_dev.h; round 6: no finish launch)
  if (p.stat_sync && stats && p.stat_replicas > 0 && !p.bwd_z)
    crog_stat_sync_tail(reinterpret_cast<const CrogSyncBlock*>(p.stat_sync), p.col_stats, p.stat_replicas, 2 * p.N, gridDim.x * gridDim.y * gridDim.z);
}

template <int CI, int CO, int G>
int launch_sw_g(const crog_gemm_desc& d, hipStream_t s) {
  const int rows = d.M / d.convW;
  // strips of whole rows, one workgroup per CU: the halo (two rows per strip) stays below ~15 % from 13 rows per strip on
  int per = std::max(cdiv(rows, 256), std::min(8, rows));
  per = cdiv(per, G) * G;
  const int wgs = cdiv(rows, per);
  const int lds = ((G == 1 ? 4 : 8) + 1) * (d.convW + 2) * CI * 2;
  static bool attr_set = false;
  auto kern = conv_sw_kernel<CI, CO, G>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    if (e != hipSuccess) {
      crog_set_error("crog_gemm: hipFuncSetAttribute failed for the sliding-window convolution: %s", hipGetErrorString(e));
      return CROG_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(SW_NT), lds, s, d, per);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

// groups of three rows when nine ring rows fit LDS and three rows fit the staging registers, single rows otherwise
inline bool sw_group3(const crog_gemm_desc& d) {
  return 9L * (d.convW + 2) * d.convC * 2 <= 160 * 1024 - 512 && 3L * d.convW * (d.convC / 8) <= 5 * SW_NT;
}
template <int CI, int CO>
int launch_sw(const crog_gemm_desc& d, hipStream_t s) {
  return sw_group3(d) ? launch_sw_g<CI, CO, 3>(d, s) : launch_sw_g<CI, CO, 1>(d, s);
}

}  // namespace

// Can the sliding-window kernel take this launch?  (the caller, crog_gemm's dispatcher, has checked the lean epilogue)
bool crog_conv_sw_eligible(const crog_gemm_desc& d) {
  if (d.dtype != CROG_BF16 || d.a_layout != CROG_A_IM2COL || d.b_layout != CROG_B_KC || d.batch != 1 || d.splitk != 1) return false;
  if ((d.convC != 32 && d.convC != 64) || (d.N != 32 && d.N != 64) || d.K != 9 * d.convC) return false;
  if (d.lda != d.convC || d.ldb < 9 * d.convC || d.ldb % 8 != 0 || d.ldc % 4 != 0) return false;
  if (d.convW < 16 || d.convH < 1 || d.M % ((long)d.convH * d.convW) != 0) return false;
  if ((long)d.convW * (d.convC / 8) > 4 * SW_NT) return false;                    // a row must fit the four staging registers per thread
  if (5L * (d.convW + 2) * d.convC * 2 > 160 * 1024 - 512) return false;
  if (d.col_stats && d.stat_replicas <= 0) return false;                           // (the per-128-row slab form is the implicit GEMM's)
  if (d.bwd_z || d.R || d.bias || d.alpha != 1.f || d.act != CROG_ACT_NONE || d.out_mode != CROG_OUT_T) return false;
  if (((uintptr_t)d.A % 16) != 0 || ((uintptr_t)d.B % 16) != 0 || ((uintptr_t)d.C % 8) != 0) return false;
  return true;
}

int crog_conv_sw_launch(const crog_gemm_desc& d, hipStream_t s) {
  if (d.convC == 32 && d.N == 32) return launch_sw<32, 32>(d, s);
  if (d.convC == 32 && d.N == 64) return launch_sw<32, 64>(d, s);
  if (d.convC == 64 && d.N == 32) return launch_sw<64, 32>(d, s);
  if (d.convC == 64 && d.N == 64) return launch_sw<64, 64>(d, s);
  crog_set_error("crog_gemm: no sliding-window instantiation for %d -> %d channels", d.convC, d.N);
  return CROG_ERR_ARG;
}
