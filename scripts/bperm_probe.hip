// What makes ds_bpermute_b32 (hipcc's lowering of __shfl_xor) return a wrong lane?  (round 5, LAB_NOTES section 10; gfx950 box)
//
// scripts/det_probe.py found LayerNorm-backward rows whose two row sums had lost ONE lane's contribution when the text tower's kernels ran
// beside the image tower's; with the reductions rewritten on DPP modifiers / v_readlane (no LDS unit) the difference was gone (0 of 297
// passes against 26 of 238).  This probe isolates it: a VICTIM kernel sums exact small integers across the 64 lanes of every wave, once
// with the ds_bpermute butterfly and once with DPP + v_readlane, and counts disagreements, while an AGGRESSOR kernel of a chosen kind
// runs beside it on another stream.
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/bperm_probe.bin scripts/bperm_probe.hip
//   run:    scripts/bperm_probe.bin [rounds=20]
// kind 1-4 found nothing; scripts/bperm_hunt.py then pinned the neighbour to the 3x3 ping-pong kernel, whose LDS-DMA requests carry
// out-of-range lanes (taps outside the image): kinds 5 / 6 are that ingredient alone.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                                \
      exit(1);                                                                             \
    }                                                                                      \
  } while (0)

template <int CTRL>
__device__ inline float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ inline float readlane_f(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
__device__ inline float sum_dpp(float v) {
  v += dpp_get<0xB1>(v);
  v += dpp_get<0x4E>(v);
  v += dpp_get<0x141>(v);
  v += dpp_get<0x140>(v);
  return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ inline float sum_bperm(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// the same butterfly through LDS MEMORY (ds_write_b32 / ds_read_b32 of a wave-private 256-byte strip): are ordinary LDS accesses affected?
__device__ inline float sum_ldsmem(float v, float* strip, int lane) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    strip[lane] = v;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    v += strip[lane ^ o];
    __builtin_amdgcn_wave_barrier();
  }
  return v;
}

// errs[0]: bpermute butterfly != DPP sum; errs[1]: LDS-memory butterfly != DPP sum; errs[2]: DPP sum != the closed form (sanity)
__global__ void __launch_bounds__(256) victim(int iters, unsigned* errs, unsigned* detail) {
  __shared__ float strips[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int it = 0; it < iters; it++) {
    const int base = it * 11 + blockIdx.x * 3 + wv;
    const float v = (float)((lane * 37 + base) & 255);
    float want = 0.f;      // closed form in every lane (64 integer adds: cheap enough, keeps the waves busy between the reductions)
    for (int l = 0; l < 64; l++) want += (float)((l * 37 + base) & 255);
    const float d = sum_dpp(v);
    const float b = sum_bperm(v);
    const float m = sum_ldsmem(v, strips[wv], lane);
    if (b != d) {
      const unsigned n = atomicAdd(&errs[0], 1u);
      if (n < 16) { detail[4 * n] = (unsigned)lane; detail[4 * n + 1] = __builtin_bit_cast(unsigned, b); detail[4 * n + 2] = __builtin_bit_cast(unsigned, d); detail[4 * n + 3] = (unsigned)it; }
    }
    if (m != d) atomicAdd(&errs[1], 1u);
    if (d != want) atomicAdd(&errs[2], 1u);
  }
}

// ---- aggressors: 256 threads, 64 KiB of dynamic LDS (two blocks per CU leave room for the victim's) --------------------------------
__device__ __attribute__((always_inline)) inline void dma_piece(const void* base, int extent, char* dst, unsigned off) {
  typedef __attribute__((address_space(3))) void lds_void;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, extent, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, off, 0, 0, 0);
}
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// kind 1: LDS-DMA (buffer_load ... lds) of 1-KiB pieces into the block's LDS, counted waits, fragments read back with ds_read_b128
// kind 2: the same bytes through registers: global_load_dwordx4 + ds_write_b128 + ds_read_b128
// kind 3: MFMA only (no LDS, no memory);  kind 4: global loads only
__global__ void __launch_bounds__(256) aggressor(int kind, const char* src, int extent, int loops, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  bf16x8 fa, fb;
  for (int e = 0; e < 8; e++) { fa[e] = (__bf16)(float)(lane + e); fb[e] = (__bf16)1.f; }
  const unsigned stride = 256u * 1024u;
  unsigned off = ((blockIdx.x * 4u + wave) * 16384u + lane * 16u) % (unsigned)(extent - 65536);
  for (int it = 0; it < loops; it++) {
    if (kind == 1 || kind == 5 || kind == 6) {
      // kind 5: every fourth lane's offset is out of range (the hardware writes zeros for it: a 3x3 tap outside the image); kind 6: all lanes
      const bool oob = kind == 6 || (kind == 5 && (lane & 3) == ((it + wave) & 3));
#pragma unroll
      for (int i = 0; i < 16; i++) dma_piece(src, extent, smem + wave * 16384 + i * 1024, oob ? 0x80000000u : off + i * 1024u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(smem + ((wave * 16 + i) * 1024 + lane * 16));
        acc[0] += t[0];
      }
    } else if (kind == 2) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + off + i * 1024u);
        *reinterpret_cast<f32x4*>(smem + ((wave * 16 + i) * 1024 + lane * 16)) = t;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(smem + ((((wave + 1) & 3) * 16 + i) * 1024 + lane * 16));
        acc[0] += t[0];
      }
      __syncthreads();
    } else if (kind == 3) {
#pragma unroll
      for (int i = 0; i < 64; i++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + off + i * 1024u);
        acc[0] += t[0];
      }
    }
    off = (off + stride) % (unsigned)(extent - 65536);
  }
  if (acc[0] + acc[1] == 12345.678f) sink[0] = acc[0];
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20;
  hipStream_t sa, sv, svp;
  CK(hipStreamCreate(&sa));
  CK(hipStreamCreate(&sv));
  CK(hipStreamCreateWithPriority(&svp, hipStreamDefault, -1));
  const int extent = 256 << 20;
  char* src;
  float* sink;
  unsigned *errs, *detail;
  CK(hipMalloc(&src, extent));
  CK(hipMemset(src, 1, extent));
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&errs, 64));
  CK(hipMalloc(&detail, 16 * 16));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const char* names[] = {"none", "LDS-DMA (buffer_load .. lds) + ds_read_b128", "global_load + ds_write_b128 + ds_read_b128", "MFMA only", "global loads only",
                         "LDS-DMA, a quarter of the lanes OUT OF RANGE", "LDS-DMA, every lane out of range"};
  const int loops[] = {0, 3000, 3000, 40000, 6000, 3000, 12000};
  for (int prio = 0; prio < 2; prio++) {
    for (int kind = 0; kind < 7; kind++) {
      for (int small = 0; small < 2; small++) {      // victim grid: 2048 blocks (fills the chip) or 20 blocks (the text tower's LayerNorm: 160 rows)
        CK(hipMemset(errs, 0, 64));
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipStream_t s = prio ? svp : sv;
        CK(hipEventRecord(e0, s));
        long waves = 0;
        for (int r = 0; r < rounds; r++) {
          if (kind) hipLaunchKernelGGL(aggressor, dim3(512), dim3(256), 65536, sa, kind, src, extent, loops[kind], sink);
          const int vb = small ? 20 : 2048, it = small ? 4000 : 300;
          for (int q = 0; q < (small ? 8 : 4); q++) {
            hipLaunchKernelGGL(victim, dim3(vb), dim3(256), 0, s, it, errs, detail);
            waves += (long)vb * 4 * it;
          }
          CK(hipStreamSynchronize(s));
          CK(hipStreamSynchronize(sa));
        }
        CK(hipEventRecord(e1, s));
        CK(hipDeviceSynchronize());
        unsigned h[3], d[64];
        CK(hipMemcpy(h, errs, 12, hipMemcpyDeviceToHost));
        CK(hipMemcpy(d, detail, 256, hipMemcpyDeviceToHost));
        printf("victim on a %s stream, %4d blocks | aggressor: %-46s | %10ld wave sums: bpermute wrong %u, LDS-memory butterfly wrong %u, DPP wrong %u\n",
               prio ? "HIGH-priority" : "default      ", small ? 20 : 2048, names[kind], waves, h[0], h[1], h[2]);
        if (h[0]) {
          const unsigned n = h[0] < 4 ? h[0] : 4;
          for (unsigned i = 0; i < n; i++)
            printf("    lane %2u: bpermute sum %.1f, DPP sum %.1f (iteration %u)\n", d[4 * i], *(float*)&d[4 * i + 1], *(float*)&d[4 * i + 2], d[4 * i + 3]);
        }
        fflush(stdout);
      }
    }
  }
  return 0;
}
