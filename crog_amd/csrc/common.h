// Shared device/host helpers for the crog_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/crog_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CROG_WAVE 64

// ---- error plumbing (host) ---------------------------------------------------------------
void crog_set_error(const char* fmt, ...);
#define CROG_CHECK_ARG(cond, ...)                 \
  do {                                            \
    if (!(cond)) {                                \
      crog_set_error(__VA_ARGS__);                \
      return CROG_ERR_ARG;                        \
    }                                             \
  } while (0)
#define CROG_LAUNCH_CHECK()                                             \
  do {                                                                  \
    hipError_t e__ = hipGetLastError();                                 \
    if (e__ != hipSuccess) {                                            \
      crog_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,     \
                     hipGetErrorString(e__));                           \
      return CROG_ERR_LAUNCH;                                           \
    }                                                                   \
  } while (0)

// per-step dropout seed offset in device memory (crog_set_seed_epoch, api.hip); null when none is installed
const uint64_t* crog_seed_epoch();

// deterministic mode (crog_set_deterministic, api.hip): launchers pick order-independent kernels; the scratch holds per-block partials
bool crog_deterministic();
float* crog_det_scratch();
constexpr int CROG_DET_LOSS_FLOATS = 8192, CROG_DET_TAP_FLOATS = 131072;      // head_loss's per-block partials | head_tap_sums's per-chunk partials
constexpr int CROG_DET_SCRATCH_FLOATS = CROG_DET_LOSS_FLOATS + CROG_DET_TAP_FLOATS;      // (disjoint regions: the two may run on different streams)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- element traits ----------------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VEC = 4;  // elements per 16-byte vector
  __device__ static inline float to_f(float v) { return v; }
  __device__ static inline float from_f(float v) { return v; }
};
template <> struct Elem<bf16> {
  static constexpr int VEC = 8;
  __device__ static inline float to_f(bf16 v) { return (float)v; }
  __device__ static inline bf16 from_f(float v) { return (bf16)v; }
};

// 16-byte vector of T
template <typename T> struct alignas(16) Vec16 {
  T v[16 / sizeof(T)];
};

template <typename T>
__device__ inline Vec16<T> ldg16(const T* p) {
  return *reinterpret_cast<const Vec16<T>*>(p);
}
// non-temporal 16-byte load: data that is read once and not again soon (the last reader of a tensor in a step)
template <typename T>
__device__ inline Vec16<T> ldg16_nt(const T* p) {
  const f32x4 r = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return __builtin_bit_cast(Vec16<T>, r);
}
template <typename T>
__device__ inline void stg16(T* p, const Vec16<T>& v) {
  *reinterpret_cast<Vec16<T>*>(p) = v;
}
template <typename T>
__device__ inline Vec16<T> zero16() {
  Vec16<T> z;
#pragma unroll
  for (int i = 0; i < (int)(16 / sizeof(T)); i++) z.v[i] = (T)0.f;
  return z;
}

// ---- wave / block reductions ---------------------------------------------------------------
// Cross-lane exchange through DPP modifiers and v_readlane only: no instruction of these reductions goes to the LDS unit (__shfl_xor lowers to
// ds_bpermute_b32, which does: one round trip more per step).  History: round 5 first blamed ds_bpermute_b32 for LayerNorm-backward rows that
// differed run to run beside another stream's GEMM; the cause was the v_pk_add_f32 pair the compiler made of the two row sums behind each
// shuffle - the library is built without packed-fp32 instructions now, and the shuffle form (-DCROG_BPERMUTE_SUMS) is clean as well
// (LAB_NOTES section 10).
template <int CTRL>
__device__ inline float dpp_get(float v) {      // the DPP-selected lane's v (every selected lane exists for the controls used here)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;   // quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror
__device__ inline float readlane_f(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
// sum / max over each row of 16 lanes, in every lane of the row (after the two quad steps a quad is uniform, so the mirrors pair each
// lane with the other quad / the other half: the xor-butterfly's tree)
__device__ inline float row16_sum(float v) {
  v += dpp_get<DPP_XOR1>(v);
  v += dpp_get<DPP_XOR2>(v);
  v += dpp_get<DPP_HALF_MIRROR>(v);
  v += dpp_get<DPP_MIRROR>(v);
  return v;
}
__device__ inline float row16_max(float v) {
  v = fmaxf(v, dpp_get<DPP_XOR1>(v));
  v = fmaxf(v, dpp_get<DPP_XOR2>(v));
  v = fmaxf(v, dpp_get<DPP_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_get<DPP_MIRROR>(v));
  return v;
}
// Exchange across rows of 16 lanes: gfx950's v_permlane16_swap_b32 / v_permlane32_swap_b32 (VALU).  With both operands = v the first
// comes back as [r0 r0 r2 r2] / [lo lo] and the second as [r1 r1 r3 r3] / [hi hi] (rows of 16 / halves of 32 lanes), so their sum (max)
// is v + v(lane ^ 16) resp. v + v(lane ^ 32) in EVERY lane.  Inline asm with its own wait states: the builtin's two results were seen
// folded into one register when added (hipcc 7.2: `v_permlane16_swap v1, v2; v_add v1, v1, v1`).
__device__ inline void permlane16_swap(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ inline void permlane32_swap(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
#if defined(CROG_BPERMUTE_SUMS) || defined(CROG_BPERM_XOR)
__device__ inline float xor16_sum(float v) { return v + __shfl_xor(v, 16, 64); }
__device__ inline float xor32_sum(float v) { return v + __shfl_xor(v, 32, 64); }
__device__ inline float xor32_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ inline float xor16_max(float v) { return fmaxf(v, __shfl_xor(v, 16, 64)); }
#else
__device__ inline float xor16_sum(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  permlane16_swap(a, b);
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ inline float xor32_sum(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  permlane32_swap(a, b);
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ inline float xor16_max(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  permlane16_swap(a, b);
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
__device__ inline float xor32_max(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  permlane32_swap(a, b);
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
#endif
// sum / max over aligned groups of LPR lanes (2 .. 64), in every lane of the group
template <int LPR>
__device__ inline float group_sum(float v) {
#if defined(CROG_BPERMUTE_SUMS) || defined(CROG_BPERM_GROUP)
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
#else
  if constexpr (LPR >= 2) v += dpp_get<DPP_XOR1>(v);
  if constexpr (LPR >= 4) v += dpp_get<DPP_XOR2>(v);
  if constexpr (LPR >= 8) v += dpp_get<DPP_HALF_MIRROR>(v);
  if constexpr (LPR >= 16) v += dpp_get<DPP_MIRROR>(v);
  if constexpr (LPR >= 32) v = xor16_sum(v);
  if constexpr (LPR >= 64) v = xor32_sum(v);
#endif
  return v;
}
template <int LPR>
__device__ inline float group_max(float v) {
#if defined(CROG_BPERMUTE_SUMS) || defined(CROG_BPERM_GROUP)
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
#else
  if constexpr (LPR >= 2) v = fmaxf(v, dpp_get<DPP_XOR1>(v));
  if constexpr (LPR >= 4) v = fmaxf(v, dpp_get<DPP_XOR2>(v));
  if constexpr (LPR >= 8) v = fmaxf(v, dpp_get<DPP_HALF_MIRROR>(v));
  if constexpr (LPR >= 16) v = fmaxf(v, dpp_get<DPP_MIRROR>(v));
  if constexpr (LPR >= 32) v = xor16_max(v);
  if constexpr (LPR >= 64) v = xor32_max(v);
#endif
  return v;
}
#if defined(CROG_BPERMUTE_SUMS) || defined(CROG_BPERM_WAVE)      // (A/B builds of scripts/build_variant.py: the ds_bpermute butterfly of rounds 1-4)
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#else
// all 64 lanes (the wave must be converged): the four row totals meet through scalar registers, the result is wave-uniform
__device__ inline float wave_sum(float v) {
  v = row16_sum(v);
  return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ inline float wave_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
#endif

// Counter-based RNG for dropout: the keep-mask is reproducible between forward and backward (and between the fused and unfused
// attention paths) because every kernel recomputes it from the same (seed, element index).

// Attention dropout (softmax probabilities: crog_softmax_fwd / bwd and the fused attention kernels): ONE hash decides the TWO
// neighbouring keys (2j, 2j + 1) of a score row, 16 bits each (threshold = p * 2^16: 0.1 -> 6554 / 65536), and the hash is
// murmur3's 32-bit finaliser - two multiplies instead of hash_u32's three.  v_mul_lo_u32 runs at a quarter of the VALU rate, and
// with one three-multiply hash per score the fused forward was VALU-bound on its dropout (141 us with p = 0.1 against 72 us with
// p = 0 at the decoder's shape): a multiply per score instead of three.  Index of a pair: row * ceil(ldp / 2) + (key >> 1).
// The seed enters through a hash of its own (seed_mix: loop-invariant, a handful of instructions per thread), XORed into the counter:
// consecutive seeds - Runtime.next_seed() hands them to consecutive dropout ops, and the per-step epoch advances by the number of ops -
// give unrelated masks.  (Until round 4 the raw seed was ADDED to the index: op k + 1's mask was op k's shifted by one pair.)
__device__ inline uint32_t seed_mix(uint64_t seed) {
  uint32_t s = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B1u);
  s ^= s >> 15;
  s *= 0x2C1B3C6Du;
  s ^= s >> 12;
  s *= 0x297A2D39u;
  s ^= s >> 15;
  return s;
}
__device__ inline uint32_t attn_hash(uint64_t seed, uint64_t pair_idx) {
  const uint32_t y = (uint32_t)(pair_idx >> 32);
  uint32_t x = ((uint32_t)pair_idx ^ seed_mix(seed)) ^ ((y << 16) | (y >> 16));
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
__device__ inline uint32_t attn_thr16(float p) { return (uint32_t)(p * 65536.0f); }
__device__ inline bool attn_keep_lo(uint32_t h, uint32_t thr16) { return (h & 0xffffu) >= thr16; }   // key 2j
__device__ inline bool attn_keep_hi(uint32_t h, uint32_t thr16) { return (h >> 16) >= thr16; }       // key 2j + 1
__device__ inline bool attn_keep(uint64_t seed, uint64_t row, int key, int ldp2, uint32_t thr16) {
  const uint32_t h = attn_hash(seed, row * (uint64_t)ldp2 + (uint64_t)(key >> 1));
  return (key & 1) ? attn_keep_hi(h, thr16) : attn_keep_lo(h, thr16);
}

// Element-wise dropout (LayerNorm input / output dropout, add_dropout): the same pair scheme - element index i of the [rows][C] tensor
// takes the low (i even) or high (i odd) 16 bits of attn_hash(seed, i >> 1).  One three-multiply hash per element made the decoder's
// LayerNorm kernels VALU-bound at a tenth of the HBM rate (67 us for 21632 x 512 bf16).  f[0 .. VEC) are VEC consecutive elements
// starting at the EVEN index idx0.
template <int VEC>
__device__ inline void dropout_apply(float (&f)[VEC], uint64_t seed, uint64_t idx0, uint32_t thr16, float sc) {
  static_assert(VEC % 2 == 0, "pairs");
#pragma unroll
  for (int e = 0; e < VEC; e += 2) {
    const uint32_t hh = attn_hash(seed, (idx0 + e) >> 1);
    f[e] = attn_keep_lo(hh, thr16) ? f[e] * sc : 0.f;
    f[e + 1] = attn_keep_hi(hh, thr16) ? f[e + 1] * sc : 0.f;
  }
}
