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
// How the device code of THIS library was compiled (crog_amd/_lib.py passes -DCROG_NO_PACKED_F32=1 together with the target-feature switch
// that removes v_pk_*_f32; a library built by other means says "packed-fp32-ops=on" and the loader warns / deterministic mode refuses).
#ifndef CROG_NO_PACKED_F32
#define CROG_NO_PACKED_F32 0
#endif
extern "C" const char* crog_build_flags(void) { return CROG_NO_PACKED_F32 ? "arch=gfx950 packed-fp32-ops=off" : "arch=gfx950 packed-fp32-ops=on"; }

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
// streaming copy, 16 bytes per lane.  V vectors per lane, a block owns 256 * V consecutive vectors (lane-contiguous within each of
// its V sub-rows), no grid-stride loop: the whole copy is in flight as independent wave-wide 1-KiB requests and the workgroup
// dispatcher, not a loop, walks the buffer (MI355X_MICROARCH.md: 6.29 TB/s for this shape; the 4096-block grid-stride form it
// replaces read 4.2-4.6 TB/s).  NT: non-temporal stores (the destination is not re-read: keep it out of L2 / Infinity Cache).
template <int V, bool NT>
__global__ void __launch_bounds__(256) copy_probe_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n16) {
  const long base = (long)blockIdx.x * (256 * V) + threadIdx.x;
  f32x4 r[V];
#pragma unroll
  for (int k = 0; k < V; k++) {
    const long i = base + k * 256;
    if (i < n16) r[k] = NT ? __builtin_nontemporal_load(src + i) : src[i];
  }
#pragma unroll
  for (int k = 0; k < V; k++) {
    const long i = base + k * 256;
    if (i < n16) {
      if (NT) __builtin_nontemporal_store(r[k], dst + i);
      else dst[i] = r[k];
    }
  }
}
__global__ void counter_add_kernel(uint64_t* __restrict__ c, uint64_t inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *c += inc;
}
}  // namespace

extern "C" int crog_probe_mfma_bf16(float* sink, int blocks, int iters, crog_stream_t stream) {
  CROG_CHECK_ARG(sink && blocks > 0 && iters > 0, "probe_mfma: bad arguments");
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters, 1.0f);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_probe_copy(const void* src, void* dst, int64_t bytes, int mode, crog_stream_t stream) {
  CROG_CHECK_ARG(src && dst && bytes > 0 && bytes % 16 == 0, "probe_copy: bytes must be a positive multiple of 16");
  CROG_CHECK_ARG(mode >= 0 && mode <= 5, "probe_copy: mode 0..5");
  const long n16 = bytes / 16;
#define CROG_COPY(V, NT) hipLaunchKernelGGL((copy_probe_kernel<V, NT>), dim3(cdiv(n16, 256 * V)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst, n16)
  switch (mode) {
    case 0: CROG_COPY(1, true); break;
    case 1: CROG_COPY(2, true); break;
    case 2: CROG_COPY(4, true); break;
    case 3: CROG_COPY(1, false); break;
    case 4: CROG_COPY(2, false); break;
    default: CROG_COPY(4, false); break;
  }
#undef CROG_COPY
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}

// ---- per-step launch state that lives in DEVICE memory, so that a captured hipGraph of the whole training step replays with fresh
// values (crog_amd/graphs.py): the dropout seed offset and (eltwise.hip) Adam's step count / learning rates -----------------------
static const uint64_t* g_seed_epoch = nullptr;
const uint64_t* crog_seed_epoch() { return g_seed_epoch; }

extern "C" int crog_set_seed_epoch(const uint64_t* epoch_dev) {
  g_seed_epoch = epoch_dev;
  return CROG_OK;
}
// ---- deterministic mode (include/crog_hip.h: crog_set_deterministic): a process-wide launch state like the seed epoch.  The scratch
// (per-block partial sums of the loss kernel) is allocated HERE, once, outside any capture: no launch ever allocates.
static int g_deterministic = 0;
static float* g_det_scratch = nullptr;
bool crog_deterministic() { return g_deterministic != 0; }
float* crog_det_scratch() { return g_det_scratch; }

extern "C" int crog_set_deterministic(int on) {
  if (on && !g_det_scratch) {
    hipError_t e = hipMalloc((void**)&g_det_scratch, CROG_DET_SCRATCH_FLOATS * sizeof(float));
    if (e != hipSuccess) {
      crog_set_error("crog_set_deterministic: hipMalloc failed: %s", hipGetErrorString(e));
      return CROG_ERR_LAUNCH;
    }
  }
  g_deterministic = on ? 1 : 0;
  return CROG_OK;
}

extern "C" int crog_counter_add(uint64_t* counter_dev, uint64_t inc, crog_stream_t stream) {
  CROG_CHECK_ARG(counter_dev != nullptr, "counter_add: null counter");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter_dev, inc);
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
