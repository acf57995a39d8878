// Which SIMD does each wave of a 512-thread block land on?  The ping-pong GEMMs (gemm_pp.hip / gemm_ppt.hip) put waves 0-3 in one group and
// 4-7 in the other and count on "one wave of each group per SIMD" (wave w on SIMD w & 3): this prints what the hardware did.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/simd_map.bin scripts/simd_map.hip ; GPU box: scripts/simd_map.bin [lds_bytes]
// HW_REG_HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>

__global__ __launch_bounds__(512) void who(unsigned* out, int spin) {
  extern __shared__ char smem[];
  const int wave = threadIdx.x >> 6;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if (spin) {                                       // keep the block resident a while, so that later blocks meet an occupied chip
    float x = (float)threadIdx.x;
    for (int i = 0; i < spin; i++) x = x * 1.0001f + 0.5f;
    if (x == 12345.678f) smem[0] = 1;
  }
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = id;
}

int main(int argc, char** argv) {
  const int lds = argc > 1 ? atoi(argv[1]) : 131072;
  const int blocks = 1024;
  unsigned *d, *h = (unsigned*)malloc(blocks * 8 * 4);
  if (hipMalloc(&d, blocks * 8 * 4) != hipSuccess) return 1;
  (void)hipFuncSetAttribute((const void*)who, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int spin : {0, 20000}) {
    hipLaunchKernelGGL(who, dim3(blocks), dim3(512), lds, 0, d, spin);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    (void)hipMemcpy(h, d, blocks * 8 * 4, hipMemcpyDeviceToHost);
    std::map<std::string, int> seen;
    for (int b = 0; b < blocks; b++) {
      char s[32];
      for (int w = 0; w < 8; w++) s[w] = '0' + ((h[b * 8 + w] >> 4) & 3);
      s[8] = 0;
      seen[s]++;
    }
    printf("lds %d B, spin %d: SIMD of waves 0..7 -> number of blocks (of %d)\n", lds, spin, blocks);
    for (auto& kv : seen) printf("  %s  %d\n", kv.first.c_str(), kv.second);
  }
  return 0;
}
