// Library-level entry points: version and error text.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void crog_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int crog_hip_version(void) { return 100; }
extern "C" const char* crog_last_error(void) { return g_err; }

// ---- peak probes (bench.py `measured_peaks`, SURVEY.md §8d: confirm the vendor peaks on the box before quoting fractions) ----
// bf16 MFMA issue rate: every wave runs `iters` x 8 independent v_mfma_f32_32x32x16_bf16 on register operands (non-zero,
// lane-dependent data: all-zero operands let the chip hold a higher clock than any real GEMM sees), no memory traffic in the
// loop; the accumulators are folded into `sink` so nothing is dead code.  FLOP = blocks * waves * iters * 8 * 32768.
namespace {
__global__ void __launch_bounds__(256) mfma_probe_kernel(float* __restrict__ sink, int iters, float seed) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    a[j] = (bf16)(seed * (float)((lane * 7 + j * 3) % 13 - 6) * 0.125f);
    b[j] = (bf16)(seed * (float)((lane * 5 + j * 11) % 17 - 8) * 0.0625f);
  }
  f32x16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; q++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[q][e] = 0.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < 8; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 8; q++)
#pragma unroll
    for (int e = 0; e < 16; e++) s += acc[q][e];
  if (s == 12345.678f) sink[blockIdx.x * 256 + threadIdx.x] = s;   // never true for these operands; keeps the chain alive
}
// streaming copy, 16 bytes per lane, grid-stride: bytes moved = 2 * n16 * 16
__global__ void __launch_bounds__(256) copy_probe_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n16) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {      // four independent 16-byte loads in flight per lane
    const f32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}
}  // namespace

extern "C" int crog_probe_mfma_bf16(float* sink, int blocks, int iters, crog_stream_t stream) {
  CROG_CHECK_ARG(sink && blocks > 0 && iters > 0, "probe_mfma: bad arguments");
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters, 1.0f);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_probe_copy(const void* src, void* dst, int64_t bytes, crog_stream_t stream) {
  CROG_CHECK_ARG(src && dst && bytes > 0 && bytes % 16 == 0, "probe_copy: bytes must be a positive multiple of 16");
  hipLaunchKernelGGL(copy_probe_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst, (long)(bytes / 16));
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

// ---- timing-only events (see include/crog_hip.h) ------------------------------------------------------------------------------
#define CROG_HIP_CALL(expr, what)                                                   \
  do {                                                                              \
    hipError_t e__ = (expr);                                                        \
    if (e__ != hipSuccess) {                                                        \
      crog_set_error("%s failed: %s", what, hipGetErrorString(e__));                \
      return CROG_ERR_LAUNCH;                                                       \
    }                                                                               \
  } while (0)

extern "C" int crog_timer_create(void** timer) {
  CROG_CHECK_ARG(timer != nullptr, "timer_create: null handle pointer");
  hipEvent_t ev;
  CROG_HIP_CALL(hipEventCreateWithFlags(&ev, hipEventDisableSystemFence), "hipEventCreateWithFlags");
  *timer = (void*)ev;
  return CROG_OK;
}
extern "C" int crog_timer_record(void* timer, crog_stream_t stream) {
  CROG_CHECK_ARG(timer != nullptr, "timer_record: null timer");
  CROG_HIP_CALL(hipEventRecord((hipEvent_t)timer, (hipStream_t)stream), "hipEventRecord");
  return CROG_OK;
}
extern "C" int crog_timer_elapsed_ms(void* start, void* stop, float* ms) {
  CROG_CHECK_ARG(start && stop && ms, "timer_elapsed_ms: null argument");
  CROG_HIP_CALL(hipEventSynchronize((hipEvent_t)stop), "hipEventSynchronize");
  CROG_HIP_CALL(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop), "hipEventElapsedTime");
  return CROG_OK;
}
extern "C" int crog_timer_destroy(void* timer) {
  if (timer) CROG_HIP_CALL(hipEventDestroy((hipEvent_t)timer), "hipEventDestroy");
  return CROG_OK;
}
