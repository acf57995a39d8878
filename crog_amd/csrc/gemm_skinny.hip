// C[M][32] = A[M][32] . B[32][32]^T for a HUGE M (gfx950, bf16): the stem's first convolution as a GEMM on its im2col rows
// (clip.py:156,208: conv1 3 -> 32, stride 2; 27 taps padded to 32; M = B * 208 * 208 = 1.38 M at B = 32).
//
// Why a kernel of its own: the tiled LDS-DMA kernels run one k-tile per 128-row block here - ring fill, one MFMA step, epilogue, 10816 blocks
// that are all latency: 169 us for 176 MB of operand + output bytes (35 us at the HBM rate).  This one has no LDS stage at all:
//   * the WEIGHTS are the MFMA's A operand and stay in registers (two bf16x8 per lane);
//   * a wave streams 32-row tiles of A straight from memory into the B operand of v_mfma_f32_32x32x16_bf16 (lane (m, h) loads the 16 bytes
//     k = 8 h .. 8 h + 7 of row m for each of the two k-steps: a tile is 2 KiB of contiguous memory), four tiles in flight per wave;
//   * D = W . A^T leaves one row's channels {4 h + 8 g + j} in a lane: 8-byte stores, and the BatchNorm (sum, sum of squares) of the lean
//     epilogue are per-lane running sums over all the wave's tiles - one LDS exchange and 64 atomic adds per workgroup at the end.
// Lean launches only (alpha 1, no bias / activation / residual; statistics in replica mode or none).  Dispatch: crog_gemm ->
// crog_gemm_skinny_eligible.
#include "gemm_dma.h"
#include "comm_dev.h"
#include <algorithm>

namespace {

constexpr int SK_NT = 256, SK_TILES = 4;

__global__ void __launch_bounds__(SK_NT) gemm_skinny32_kernel(const crog_gemm_desc p, int tiles, int waves_total) {
  __shared__ float red[SK_NT / 64][64][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, h = lane >> 5;
  const bf16* A = reinterpret_cast<const bf16*>(p.A);
  const bf16* Bw = reinterpret_cast<const bf16*>(p.B);
  bf16* C = reinterpret_cast<bf16*>(p.C);
  // weights: lane (n, h) holds W[n][16 s + 8 h .. + 7]
  bf16x8 wf[2];
#pragma unroll
  for (int s = 0; s < 2; s++) wf[s] = *reinterpret_cast<const bf16x8*>(Bw + (long)m * 32 + 16 * s + 8 * h);
  float s1[16], s2[16];
#pragma unroll
  for (int e = 0; e < 16; e++) s1[e] = s2[e] = 0.f;
  const bool stats = p.col_stats != nullptr;
  const int gw = blockIdx.x * (SK_NT / 64) + wave;
  for (int t0 = gw * SK_TILES; t0 < tiles; t0 += waves_total * SK_TILES) {
    bf16x8 af[SK_TILES][2];
#pragma unroll
    for (int u = 0; u < SK_TILES; u++) {
      const long row = min((long)(t0 + u) * 32 + m, (long)p.M - 1);      // (rows past the end re-read the last row; masked below)
#pragma unroll
      for (int s = 0; s < 2; s++) af[u][s] = *reinterpret_cast<const bf16x8*>(A + row * 32 + 16 * s + 8 * h);
    }
#pragma unroll
    for (int u = 0; u < SK_TILES; u++) {
      if (t0 + u >= tiles) break;      // (wave-uniform)
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], af[u][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], af[u][1], acc, 0, 0, 0);
      const long row = (long)(t0 + u) * 32 + m;
      const bool live = row < p.M;
      if (stats) {
#pragma unroll
        for (int e = 0; e < 16; e++) { const float v = live ? acc[e] : 0.f; s1[e] += v; s2[e] += v * v; }
      }
      if (live) {
        // acc[4 g + j] = channel 8 g + 4 h + j of row `row`
#pragma unroll
        for (int g = 0; g < 4; g++) {
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; j++) o[j] = (bf16)acc[4 * g + j];
          *reinterpret_cast<bf16x4*>(C + row * 32 + 8 * g + 4 * h) = o;
        }
      }
    }
  }
  if (!stats) return;
  // per-lane sums -> per-channel sums of the workgroup -> one replica row (any row will do: the consumer adds the replicas up)
#pragma unroll
  for (int e = 0; e < 16; e++) { red[wave][lane][e] = s1[e]; red[wave][lane][16 + e] = s2[e]; }
  __syncthreads();
  if (tid < 64) {
    const int c = tid >> 1, k = tid & 1;                  // channel, (sum | sum of squares)
    const int hh = (c >> 2) & 1, e = (c & 3) + 4 * (c >> 3);
    float v = 0.f;
    for (int w = 0; w < SK_NT / 64; w++)
      for (int mm = 0; mm < 32; mm++) v += red[w][32 * hh + mm][16 * k + e];
    atomicAdd(p.col_stats + ((int64_t)(blockIdx.x % p.stat_replicas) * 32 + c) * 2 + k, v);
  }
  // SyncBatchNorm forward statistics: the last block exchanges the totals (crog_gemm_desc.stat_sync, comm_dev.h; round 6: no finish launch)
  if (p.stat_sync && stats && p.stat_replicas > 0 && !p.bwd_z)
    crog_stat_sync_tail(reinterpret_cast<const CrogSyncBlock*>(p.stat_sync), p.col_stats, p.stat_replicas, 2 * 32, gridDim.x * gridDim.y * gridDim.z);
}

}  // namespace

bool crog_gemm_skinny_eligible(const crog_gemm_desc& d) {
  if (d.dtype != CROG_BF16 || d.a_layout != CROG_A_KC || d.b_layout != CROG_B_KC || d.batch != 1 || d.splitk > 1) return false;
  if (d.N != 32 || d.K != 32 || d.lda != 32 || d.ldb != 32 || d.ldc != 32 || d.M < 32) return false;
  if (d.out_mode != CROG_OUT_T || d.alpha != 1.f || d.bias || d.R || d.a_sum || d.bwd_z || d.act != CROG_ACT_NONE) return false;
  if (d.col_stats && d.stat_replicas <= 0) return false;      // (slab-mode statistics - fp32 / deterministic - stay with the tiled kernel)
  if (((uintptr_t)d.A % 16) != 0 || ((uintptr_t)d.B % 16) != 0 || ((uintptr_t)d.C % 8) != 0) return false;
  return true;
}

int crog_gemm_skinny_launch(const crog_gemm_desc& d, hipStream_t s) {
  const int tiles = (d.M + 31) / 32;
  const int wpb = SK_NT / 64;
  // ~8 workgroups per CU at most; every wave gets at least one batch of SK_TILES tiles
  const int blocks = std::max(1, std::min(2048, (tiles + wpb * SK_TILES - 1) / (wpb * SK_TILES)));
  hipLaunchKernelGGL(gemm_skinny32_kernel, dim3(blocks), dim3(SK_NT), 0, s, d, tiles, blocks * wpb);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
