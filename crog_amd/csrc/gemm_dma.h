// Helpers shared by the LDS-DMA GEMM kernels (gemm.hip, gemm_pp.hip): the buffer-addressed global -> LDS request, counted waits,
// the XCD-aware tile order.  gfx950 only.
#pragma once
#include "common.h"

namespace {

// An out-of-bounds buffer offset: the request moves no bytes and the hardware writes zeros into LDS.
constexpr unsigned DMA_OOB = 0x80000000u;

// One 1-KiB LDS-DMA request (buffer_load_dwordx4 ... lds: wave-uniform LDS base in M0 + lane * 16, per-lane byte offset into the buffer).
// The buffer resource is rebuilt from the (kernel-scope, wave-uniform) base pointer and extent at the point of use, so that it lives in SGPRs.
//
// Round 6: the request is INLINE ASSEMBLY (CROG_ASM_DMA, default 1; 0 = __builtin_amdgcn_raw_ptr_buffer_load_lds).  With the builtin the compiler knows
// that LDS is written behind the VM counter and guards every LDS read it cannot prove disjoint with an s_waitcnt vmcnt(0) of its own: it proves the plain
// ds_read_b128 of the K-contiguous kernels disjoint, but NOT ds_read_b64_tr_b16 - every kernel with a transposed operand (gemm_ppt, the A_MC / B_NC
// forms of gemm_dma_kernel) drained its whole ring once per phase, whatever the counted waits of its schedule said (found with in-kernel stamps:
// LAB_NOTES section 11; gemm_ppt 640 -> 750 TFLOP/s on 144 CUs).  As assembly the request is opaque: the "memory" clobber keeps it in program order
// against every LDS / global access, and the schedule's own vmcnt / barrier pairs are the only waits - as they were designed to be.
#ifndef CROG_ASM_DMA
#define CROG_ASM_DMA 1
#endif
typedef int dma_i32x4 __attribute__((ext_vector_type(4)));
__device__ __attribute__((always_inline)) inline void dma_piece(const void* base, int extent, char* dst, unsigned off) {
#ifdef CROG_PROBE_NO_OOB      // probe build (LAB_NOTES section 10): no lane is ever out of range - border taps read the operand's first bytes (WRONG values, on purpose)
  off = off == DMA_OOB ? 0u : off;
#endif
#if CROG_ASM_DMA
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned long long b = (unsigned long long)base;
  dma_i32x4 r;      // V# words: base[31:0]; base[47:32], stride 0; num_records = extent (bytes); DST_SEL / format word as make_buffer_rsrc(…, 0x00020000)
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane(extent);
  r.w = 0x00020000;
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_char*)dst);
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(off), "s"(r) : "memory", "m0");
#else
  typedef __attribute__((address_space(3))) void lds_void;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, extent, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, off, 0, 0, 0);
#endif
}

template <int N> __device__ inline void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Workgroups are dealt to the 8 XCDs round-robin in launch order (linear id % 8), and every XCD has its own L2.
//  * plain grid (tiles, batch*splitk): each XCD gets a contiguous run of output tiles, so neighbouring tiles share their
//    A rows / B columns in one L2;
//  * split-K grid (tiles*splitk, 1): the work items (reduction slice, tile), slice-major, are cut into 8 contiguous runs, so an
//    XCD runs all output tiles of a slice back to back and a slice of the two operands (the huge dimension of a weight-gradient
//    GEMM) is fetched into one L2 (two at a run boundary) instead of all eight.
__device__ inline void xcd_map(int nwg, int splitk, int& id, int& z) {
  const int xcd = id & 7, loc = id >> 3;
  const int total = (gridDim.y == 1 && splitk > 1) ? nwg * splitk : nwg;
  const int q = total >> 3, rr = total & 7;
  const int w = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  if (total != nwg) {
    z = w / nwg;
    id = w - z * nwg;
  } else {
    id = w;
  }
}

__device__ inline void mma32(const bf16x8& a, const bf16x8& b, f32x4& c) { c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// epilogue activations (crog_gemm_desc.act)
__device__ inline float act_quickgelu(float v) { return v * __builtin_amdgcn_rcpf(1.f + expf(-1.702f * v)); }
__device__ inline float act_tanh(float v) {      // (1 - t) / (1 + t), t = e^(-2|v|) in (0, 1]: absolute error ~1 ulp of 1
  const float t = expf(-2.f * fabsf(v));
  return copysignf((1.f - t) * __builtin_amdgcn_rcpf(1.f + t), v);
}

}  // namespace
