// Fused multi-head attention for the head_dim = 64, bf16 case: the decoder's 676-token self-attention (layers.py:291-296,324) and - with a
// key padding mask - its vision-to-text cross-attention over 20 word keys (layers.py:329-332), the ViT tower's 197-token blocks
// (clip.py:246-260) — the "MFMA attention path" of BASELINE config 4 — and, causal, the text tower's 20-token blocks (clip.py:446-452).  Scores are never written to HBM: the unfused path moves ~1 GB per decoder layer forward (S, P, dropout(P)) and
// ~2 GB backward at B = 32; this one reads Q, K, V (+ O, dO) once per 128-row block.
//
// Layout trick (all three kernels): the 32x32x16 MFMA returns C with lane = column, 16 rows in registers.  Computing the
// score tile TRANSPOSED to what the next product needs makes those registers directly an MFMA operand of the next product —
//   forward / dQ :  S^T = K Q^T  [keys x queries]   lane = query: the softmax row statistics are per-lane scalars, and the
//                   registers are the operand of  O^T += V^T P^T  resp.  dQ += dS K   (reduction over keys)
//   dK, dV       :  S   = Q K^T  [queries x keys]   lane = key:   registers are the operand of  dV += P^T dO,  dK += dS^T Q
// with the reduction index permuted consistently on both operands (slot (h, j) of k-step t <-> row 16t + 8(j>>2) + 4h + (j&3));
// the other operand comes out of LDS through ds_read_b64_tr_b16 with the same permutation.  No shuffles, no LDS round trip
// for P.  Attention dropout uses the same pair hash and index (attn_hash, common.h: one hash per two neighbouring keys) as crog_softmax_fwd, so the fused and the
// unfused paths drop identical elements for a given seed.  Round 5: the forward can leave its keep decisions as a bit map (AttnArgs::keep, 32 bits per
// score row and key tile = 15 MB for the decoder's 676 x 676 x 8 heads x B = 32) which both backward kernels read instead of hashing again: the
// hash is two quarter-rate multiplies per key pair and made the three kernels VALU-bound (common.h).
#include "common.h"

namespace {

constexpr int DH = 64;     // head dimension
constexpr int TT = 32;     // rows of a streamed LDS tile (keys in fwd / dQ, queries in dK/dV)
constexpr int NTHR = 256;  // 4 waves, 32 rows of the resident operand each
constexpr int TILE = TT * DH;

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

struct AttnArgs {
  const bf16 *Q, *K, *V, *O, *dO;
  bf16 *Out, *dQ, *dK, *dV;
  float *lse, *D;
  long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int Lq, Lk, heads, ldp;
  int causal;              // key k contributes to query q only if k <= q (CLIP text transformer, clip.py:446-452 build_attention_mask)
  const uint8_t* kpm;      // key padding mask [B][Lk], non-zero = the key is padding and reaches no query (layers.py:332, crog.py:55); null = none
  float scale, p_drop;
  uint64_t seed;
  const uint64_t* epoch;   // crog_set_seed_epoch: per-step seed offset in device memory (null = none)
  uint16_t* keep;          // dropout keep bits (null = every kernel hashes for itself): written by the forward, read by both backward kernels.
                           // Word [((bh * ceil(Lk / 32) + key tile) * Lq + q) * 2 + h], bit r = key 32 * tile + acc_row(r, h) of score row q is kept
};

// ---- LDS images of a [32][64] bf16 tile (128-byte rows) -------------------------------------------------------------
// "kc": read as rows (ds_read_b128, lane = row): 16-byte slot = chunk ^ ((row >> 1) & 7) -> the 16 rows of a b128 lane group
//       hit 16 different slots of the 256-byte bank row.
// "tr": read transposed (ds_read_b64_tr_b16, 4 rows x 64 bytes per 32-lane half): rows r and r+2 share a bank-row half, so
//       the 64-byte half is flipped by bit 1 of the row.
__device__ inline int kc_off(int row, int chunk) { return row * DH + ((chunk ^ ((row >> 1) & 7)) << 3); }
__device__ inline int tr_off(int row, int chunk) { return row * DH + ((chunk ^ (((row >> 1) & 1) << 2)) << 3); }

__device__ inline bf16x8 kc_frag(const bf16* tile, int row, int chunk) { return *reinterpret_cast<const bf16x8*>(tile + kc_off(row, chunk)); }

// transposed fragment: lane receives column cbase + (lane & 31), rows R0..R0+3 and R1..R1+3
__device__ inline bf16x8 tr_frag(const bf16* tile, int cbase, int R0, int R1, int lane) {
  const int i = lane & 15, q = i >> 2, pp = i & 3;
  const int col = cbase + 16 * ((lane >> 4) & 1) + 4 * pp;
  const int chunk = col >> 3, inb = col & 7;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + tr_off(R0 + q, chunk) + inb));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + tr_off(R1 + q, chunk) + inb));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ inline f32x16 mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// row (0..31) of accumulator register r for the lane's half h
__device__ inline int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// bit r: key kb + acc_row(r, h) of batch element b exists and is not padding (only called for ragged or padded key tiles)
__device__ inline unsigned key_bits16(const AttnArgs& a, int b, int kb, int h) {
  unsigned m = 0u;
  if (a.kpm) {      // (uniform branch; the 16 byte loads are unconditional - clamped - so that they are in flight together)
    const uint8_t* row = a.kpm + (long)b * a.Lk;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int kk = kb + acc_row(r, h);
      const unsigned pad = row[min(kk, a.Lk - 1)];
      m |= ((kk < a.Lk && pad == 0u) ? 1u : 0u) << r;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; r++) m |= ((kb + acc_row(r, h) < a.Lk) ? 1u : 0u) << r;
  }
  return m;
}

__device__ inline bf16x8 pack8(const f32x16& v, int t) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; j++) o[j] = (bf16)v[8 * t + j];
  return o;
}

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
__device__ inline float ex2(float x) { return __builtin_amdgcn_exp2f(x); }      // v_exp_f32 (arguments here are <= 0 or -inf)

// =====================================================================================================================
// forward: O = softmax(scale Q K^T) [dropout] V, lse = row log-sum-exp.  grid (ceil(Lq / 128), B * heads)
// =====================================================================================================================
__global__ void __launch_bounds__(NTHR, 3) flash_fwd_kernel(const AttnArgs a) {
  const uint64_t seed = a.seed + (a.epoch ? *a.epoch : 0ull);
  __shared__ __attribute__((aligned(16))) bf16 sK[2][TILE];
  __shared__ __attribute__((aligned(16))) bf16 sV[2][TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, ln = lane & 31;
  const int bh = blockIdx.y, b = bh / a.heads, hd = bh % a.heads;
  const int q = blockIdx.x * 128 + wave * 32 + ln;
  const int qc = min(q, a.Lq - 1);
  const bf16* qp = a.Q + ((long)b * a.Lq + qc) * a.ldq + hd * DH + h * 8;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  f32x16 o[2];
#pragma unroll
  for (int e = 0; e < 16; e++) o[0][e] = o[1][e] = 0.f;
  float m = -INFINITY, l = 0.f;
  const float c2 = a.scale * LOG2E;      // scores in log2 units: exp(x) = v_exp_f32(x log2 e) - the factor folded into the scale saves a multiply per score

  const int lrow = tid >> 3, lch = tid & 7;
  const bf16* kg = a.K + (long)b * a.Lk * a.ldk + hd * DH + lch * 8;
  const bf16* vg = a.V + (long)b * a.Lk * a.ldv + hd * DH + lch * 8;
  const int nkt = (a.Lk + TT - 1) / TT;
  const uint32_t thr = attn_thr16(a.p_drop);
  const float sc = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const uint64_t rowbase2 = ((uint64_t)bh * a.Lq + q) * (uint64_t)((a.ldp + 1) >> 1);      // first key pair of the lane's score row

  bf16x8 rk, rv;
  {
    const int key = min(lrow, a.Lk - 1);
    rk = *reinterpret_cast<const bf16x8*>(kg + (long)key * a.ldk);
    rv = *reinterpret_cast<const bf16x8*>(vg + (long)key * a.ldv);
    *reinterpret_cast<bf16x8*>(&sK[0][kc_off(lrow, lch)]) = rk;
    *reinterpret_cast<bf16x8*>(&sV[0][tr_off(lrow, lch)]) = rv;
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) {
      const int key = min((kt + 1) * TT + lrow, a.Lk - 1);
      rk = *reinterpret_cast<const bf16x8*>(kg + (long)key * a.ldk);
      rv = *reinterpret_cast<const bf16x8*>(vg + (long)key * a.ldv);
    }
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; e++) s[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) s = mfma(kc_frag(sK[buf], ln, 2 * ks + h), qf[ks], s);
    const int kb = kt * TT;
    float mt = -INFINITY;
    if (kb + TT <= a.Lk && !a.causal && !a.kpm) {
#pragma unroll
      for (int r = 0; r < 16; r++) { s[r] *= c2; mt = fmaxf(mt, s[r]); }
    } else {
      const unsigned kv = key_bits16(a, b, kb, h);
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int kk = kb + acc_row(r, h);
        const bool ok = (((kv >> r) & 1u) != 0u) & ((a.causal == 0) | (kk <= q));
        s[r] = ok ? s[r] * c2 : -INFINITY;
        mt = fmaxf(mt, s[r]);
      }
    }
    mt = xor32_max(mt);
    const float mn = fmaxf(m, mt);
    const float alpha = ex2(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) { s[r] = ex2(s[r] - mn); ps += s[r]; }
    l = l * alpha + ps;
    m = mn;
#pragma unroll
    for (int e = 0; e < 16; e++) { o[0][e] *= alpha; o[1][e] *= alpha; }
    if (a.p_drop > 0.f) {
      // registers 4 g .. 4 g + 3 hold keys kb + 8 g + 4 h + {0, 1, 2, 3}: two pairs, one hash each (attn_hash, common.h)
      unsigned bits = 0u;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const uint32_t hh = attn_hash(seed, rowbase2 + (uint64_t)((kb + acc_row(r, h)) >> 1));
        const bool k0 = attn_keep_lo(hh, thr), k1 = attn_keep_hi(hh, thr);
        s[r] = k0 ? s[r] * sc : 0.f;
        s[r + 1] = k1 ? s[r + 1] * sc : 0.f;
        bits |= ((k0 ? 1u : 0u) | (k1 ? 2u : 0u)) << r;
      }
      // the wave's 64 lanes are 32 consecutive score rows x 2 halves: one 256-byte store per key tile
      if (a.keep && q < a.Lq) a.keep[(((long)bh * nkt + kt) * a.Lq + q) * 2 + h] = (uint16_t)bits;
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const bf16x8 pb = pack8(s, t);
#pragma unroll
      for (int mt2 = 0; mt2 < 2; mt2++) o[mt2] = mfma(tr_frag(sV[buf], mt2 * 32, 16 * t + 4 * h, 16 * t + 8 + 4 * h, lane), pb, o[mt2]);
    }
    if (kt + 1 < nkt) {
      *reinterpret_cast<bf16x8*>(&sK[buf ^ 1][kc_off(lrow, lch)]) = rk;
      *reinterpret_cast<bf16x8*>(&sV[buf ^ 1][tr_off(lrow, lch)]) = rv;
    }
    __syncthreads();
  }
  l = xor32_sum(l);
  if (q < a.Lq) {
    const float inv = 1.f / l;
    bf16* op = a.Out + ((long)b * a.Lq + q) * a.ldo + hd * DH + 4 * h;
#pragma unroll
    for (int mt2 = 0; mt2 < 2; mt2++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        bf16x4 w;
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = (bf16)(o[mt2][4 * g + j] * inv);
        *reinterpret_cast<bf16x4*>(op + mt2 * 32 + 8 * g) = w;
      }
    if (h == 0) a.lse[(long)bh * a.Lq + q] = m * LN2 + __logf(l);      // (natural-log units for the backward kernels and the caller)
  }
}

// =====================================================================================================================
// backward 1: D = rowsum(dO * O);  dQ = scale * dS K  with dS = P * (dropout'(dO V^T) - D).  Same blocking as the forward.
// =====================================================================================================================
__global__ void __launch_bounds__(NTHR, 3) flash_bwd_dq_kernel(const AttnArgs a) {
  const uint64_t seed = a.seed + (a.epoch ? *a.epoch : 0ull);
  __shared__ __attribute__((aligned(16))) bf16 sKc[2][TILE];
  __shared__ __attribute__((aligned(16))) bf16 sKt[2][TILE];
  __shared__ __attribute__((aligned(16))) bf16 sVc[2][TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, ln = lane & 31;
  const int bh = blockIdx.y, b = bh / a.heads, hd = bh % a.heads;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int q = q0 + ln;
  const int qc = min(q, a.Lq - 1);
  const long qrow = (long)b * a.Lq + qc;
  bf16x8 qf[4], dof[4];
  float Dq = 0.f;
  // One key tile (the cross-attention's 20 words, the text tower's 20-token blocks): D = rowsum(dO * O) = sum_k dropout(P)_k (dO V^T)_k is
  // taken from the tile itself below and O (a quarter of this kernel's bytes there) is not read.
  const bool one_tile = a.Lk <= TT;
  {
    const bf16* qp = a.Q + qrow * a.ldq + hd * DH + h * 8;
    const bf16* dp = a.dO + qrow * a.lddo + hd * DH + h * 8;
    const bf16* op = a.O + qrow * a.ldo + hd * DH + h * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
      dof[ks] = *reinterpret_cast<const bf16x8*>(dp + ks * 16);
      if (!one_tile) {
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(op + ks * 16);
#pragma unroll
        for (int j = 0; j < 8; j++) Dq += (float)dof[ks][j] * (float)of[j];
      }
    }
    if (!one_tile) Dq = xor32_sum(Dq);
  }
  const float Lr = a.lse[(long)bh * a.Lq + qc] * LOG2E, c2 = a.scale * LOG2E;      // (log2 units, as in the forward)
  if (!one_tile && h == 0 && q < a.Lq) a.D[(long)bh * a.Lq + q] = Dq;
  f32x16 dq[2];
#pragma unroll
  for (int e = 0; e < 16; e++) dq[0][e] = dq[1][e] = 0.f;

  const int lrow = tid >> 3, lch = tid & 7;
  const bf16* kg = a.K + (long)b * a.Lk * a.ldk + hd * DH + lch * 8;
  const bf16* vg = a.V + (long)b * a.Lk * a.ldv + hd * DH + lch * 8;
  const int nkt = (a.Lk + TT - 1) / TT;
  const uint32_t thr = attn_thr16(a.p_drop);
  const float sc = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const uint64_t rowbase2 = ((uint64_t)bh * a.Lq + q) * (uint64_t)((a.ldp + 1) >> 1);      // first key pair of the lane's score row

  bf16x8 rk, rv;
  {
    const int key = min(lrow, a.Lk - 1);
    rk = *reinterpret_cast<const bf16x8*>(kg + (long)key * a.ldk);
    rv = *reinterpret_cast<const bf16x8*>(vg + (long)key * a.ldv);
    *reinterpret_cast<bf16x8*>(&sKc[0][kc_off(lrow, lch)]) = rk;
    *reinterpret_cast<bf16x8*>(&sKt[0][tr_off(lrow, lch)]) = rk;
    *reinterpret_cast<bf16x8*>(&sVc[0][kc_off(lrow, lch)]) = rv;
  }
  // the forward's dropout decisions (a.keep: non-null only with p_drop > 0), same lane <-> (row, half) mapping as there: one coalesced
  // 2-byte load per key tile, fetched a tile ahead
  const uint16_t* kw = a.keep ? a.keep + ((long)bh * nkt * a.Lq + qc) * 2 + h : nullptr;
  unsigned kbits = kw ? (unsigned)kw[0] : 0u, kbits_next = 0u;
  __syncthreads();
  for (int kt = 0; kt < nkt; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) {
      const int key = min((kt + 1) * TT + lrow, a.Lk - 1);
      rk = *reinterpret_cast<const bf16x8*>(kg + (long)key * a.ldk);
      rv = *reinterpret_cast<const bf16x8*>(vg + (long)key * a.ldv);
      if (kw) kbits_next = kw[(long)(kt + 1) * a.Lq * 2];
    }
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; e++) s[e] = dp[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      s = mfma(kc_frag(sKc[buf], ln, 2 * ks + h), qf[ks], s);
      dp = mfma(kc_frag(sVc[buf], ln, 2 * ks + h), dof[ks], dp);
    }
    const int kb = kt * TT;
    if (kb + TT <= a.Lk && !a.kpm && !a.causal) {      // a whole, unmasked tile (uniform): no per-score conditions
#pragma unroll
      for (int r = 0; r < 16; r++) s[r] = ex2(s[r] * c2 - Lr);
    } else {
      const unsigned kv = key_bits16(a, b, kb, h);
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int kk = kb + acc_row(r, h);
        // the exponential of a SELECTED ARGUMENT (v_exp_f32(-inf) = 0), not a selected exponential: hipcc compiles `cond ? ex2(x) : 0.f` into
        // an exec-mask branch per score (16 per tile, each with its own waits)
        const bool ok = (((kv >> r) & 1u) != 0u) & ((a.causal == 0) | (kk <= q));      // (bitwise: no short-circuit control flow)
        const float x = s[r] * c2 - Lr;
        s[r] = ex2(ok ? x : -INFINITY);
      }
    }
    if (a.p_drop > 0.f) {
      if (kw) {
#pragma unroll
        for (int r = 0; r < 16; r++) dp[r] = ((kbits >> r) & 1u) ? dp[r] * sc : 0.f;
        kbits = kbits_next;
      } else {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const uint32_t hh = attn_hash(seed, rowbase2 + (uint64_t)((kb + acc_row(r, h)) >> 1));
          dp[r] = attn_keep_lo(hh, thr) ? dp[r] * sc : 0.f;
          dp[r + 1] = attn_keep_hi(hh, thr) ? dp[r + 1] * sc : 0.f;
        }
      }
    }
    if (one_tile) {
      float d = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) d += s[r] * dp[r];
      Dq = xor32_sum(d);
      if (h == 0 && q < a.Lq) a.D[(long)bh * a.Lq + q] = Dq;
    }
#pragma unroll
    for (int r = 0; r < 16; r++) s[r] = s[r] * (dp[r] - Dq) * a.scale;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const bf16x8 af = pack8(s, t);
#pragma unroll
      for (int nt = 0; nt < 2; nt++) dq[nt] = mfma(af, tr_frag(sKt[buf], nt * 32, 16 * t + 4 * h, 16 * t + 8 + 4 * h, lane), dq[nt]);
    }
    if (kt + 1 < nkt) {
      *reinterpret_cast<bf16x8*>(&sKc[buf ^ 1][kc_off(lrow, lch)]) = rk;
      *reinterpret_cast<bf16x8*>(&sKt[buf ^ 1][tr_off(lrow, lch)]) = rk;
      *reinterpret_cast<bf16x8*>(&sVc[buf ^ 1][kc_off(lrow, lch)]) = rv;
    }
    __syncthreads();
  }
  // dq[nt]: lane = column (dh = nt*32 + ln), register r = query q0 + acc_row(r, h)
#pragma unroll
  for (int nt = 0; nt < 2; nt++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int qq = q0 + acc_row(r, h);
      if (qq < a.Lq) a.dQ[((long)b * a.Lq + qq) * a.lddq + hd * DH + nt * 32 + ln] = (bf16)dq[nt][r];
    }
}

// =====================================================================================================================
// backward 2: dV = dropout(P)^T dO,  dK = scale * dS^T Q.  Each wave owns 32 keys; queries stream through LDS.
// grid (ceil(Lk / 128), B * heads).  Needs lse (forward) and D (flash_bwd_dq_kernel).
// =====================================================================================================================
__global__ void __launch_bounds__(NTHR, 2) flash_bwd_dkdv_kernel(const AttnArgs a) {
  const uint64_t seed = a.seed + (a.epoch ? *a.epoch : 0ull);
  __shared__ __attribute__((aligned(16))) bf16 sQc[2][TILE];
  __shared__ __attribute__((aligned(16))) bf16 sQt[2][TILE];
  __shared__ __attribute__((aligned(16))) bf16 sOc[2][TILE];
  __shared__ __attribute__((aligned(16))) bf16 sOt[2][TILE];
  __shared__ __attribute__((aligned(16))) float sL[2][TT];
  __shared__ __attribute__((aligned(16))) float sD[2][TT];
  __shared__ __attribute__((aligned(16))) uint32_t sM[2][4][TT];      // a.keep: the 32 + 32 keep bits of (query, this wave's key tile)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, ln = lane & 31;
  const int bh = blockIdx.y, b = bh / a.heads, hd = bh % a.heads;
  const int k0 = blockIdx.x * 128 + wave * 32;
  const int key = k0 + ln;
  const int kc = min(key, a.Lk - 1);
  const bool kvalid = key < a.Lk && !(a.kpm && a.kpm[(long)b * a.Lk + key]);      // a padded key takes no probability: dK = dV = 0
  bf16x8 kf[4], vf[4];
  {
    const bf16* kp = a.K + ((long)b * a.Lk + kc) * a.ldk + hd * DH + h * 8;
    const bf16* vp = a.V + ((long)b * a.Lk + kc) * a.ldv + hd * DH + h * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(kp + ks * 16);
      vf[ks] = *reinterpret_cast<const bf16x8*>(vp + ks * 16);
    }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int e = 0; e < 16; e++) dk[0][e] = dk[1][e] = dv[0][e] = dv[1][e] = 0.f;

  const int lrow = tid >> 3, lch = tid & 7;
  const bf16* qg = a.Q + (long)b * a.Lq * a.ldq + hd * DH + lch * 8;
  const bf16* og = a.dO + (long)b * a.Lq * a.lddo + hd * DH + lch * 8;
  const float* lg = a.lse + (long)bh * a.Lq;
  const float* dg = a.D + (long)bh * a.Lq;
  const int nqt = (a.Lq + TT - 1) / TT;
  const uint32_t thr = attn_thr16(a.p_drop);
  const float sc = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const float c2 = a.scale * LOG2E;

  // keep bits: thread t < 128 stages the word of (key tile of wave t / 32, query t % 32); this lane's key is bit 16 hf + rf of it, with
  // (hf, rf) the half and register that hold key k0 + ln in the forward's tile: acc_row(rf, hf) = ln
  const int nktk = (a.Lk + TT - 1) / TT;
  const uint32_t* mg = a.keep ? reinterpret_cast<const uint32_t*>(a.keep) + ((long)bh * nktk + min((int)blockIdx.x * 4 + (tid >> 5), nktk - 1)) * a.Lq
                              : nullptr;
  const bool mstage = mg && tid < 128;
  const int mbit = 16 * ((ln >> 2) & 1) + (ln & 3) + 4 * (ln >> 3);
  uint32_t rm = 0u;

  bf16x8 rq, ro;
  float rs = 0.f;
  {
    if (mstage) sM[0][tid >> 5][tid & 31] = mg[min(tid & 31, a.Lq - 1)];
    const int qq = min(lrow, a.Lq - 1);
    rq = *reinterpret_cast<const bf16x8*>(qg + (long)qq * a.ldq);
    ro = *reinterpret_cast<const bf16x8*>(og + (long)qq * a.lddo);
    *reinterpret_cast<bf16x8*>(&sQc[0][kc_off(lrow, lch)]) = rq;
    *reinterpret_cast<bf16x8*>(&sQt[0][tr_off(lrow, lch)]) = rq;
    *reinterpret_cast<bf16x8*>(&sOc[0][kc_off(lrow, lch)]) = ro;
    *reinterpret_cast<bf16x8*>(&sOt[0][tr_off(lrow, lch)]) = ro;
    if (tid < TT) sL[0][tid] = lg[min(tid, a.Lq - 1)] * LOG2E;      // (log2 units, as in the forward)
    else if (tid < 2 * TT) sD[0][tid - TT] = dg[min(tid - TT, a.Lq - 1)];
  }
  __syncthreads();
  for (int qt = 0; qt < nqt; qt++) {
    const int buf = qt & 1;
    if (qt + 1 < nqt) {
      const int qq = min((qt + 1) * TT + lrow, a.Lq - 1);
      rq = *reinterpret_cast<const bf16x8*>(qg + (long)qq * a.ldq);
      ro = *reinterpret_cast<const bf16x8*>(og + (long)qq * a.lddo);
      if (tid < TT) rs = lg[min((qt + 1) * TT + tid, a.Lq - 1)] * LOG2E;
      else if (tid < 2 * TT) rs = dg[min((qt + 1) * TT + tid - TT, a.Lq - 1)];
      if (mstage) rm = mg[min((qt + 1) * TT + (tid & 31), a.Lq - 1)];
    }
    if (k0 < a.Lk) {      // (a wave whose 32 keys all lie beyond Lk - three of four at the cross-attention's 20 keys - only helps with the loads)
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; e++) s[e] = dp[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      s = mfma(kc_frag(sQc[buf], ln, 2 * ks + h), kf[ks], s);      // S[query][key]
      dp = mfma(kc_frag(sOc[buf], ln, 2 * ks + h), vf[ks], dp);    // (dO V^T)[query][key]
    }
    const int qb = qt * TT;
    f32x16 pd;
    // dropout decisions: a hash covers the key pair (2 j, 2 j + 1) of one score row = lanes (2 j, 2 j + 1) of one register, so of
    // every two rows the even lane hashes the first and the odd lane the second, and a DPP swap hands each its partner's value
    unsigned keepbits = 0xffffu;
    if (mg) {
      keepbits = 0u;
      const uint4* mrow = reinterpret_cast<const uint4*>(&sM[buf][wave][0]);
#pragma unroll
      for (int g = 0; g < 4; g++) {      // registers 4 g .. 4 g + 3 are queries qb + 8 g + 4 h + {0, 1, 2, 3}
        const uint4 w = mrow[2 * g + h];
        keepbits |= (((w.x >> mbit) & 1u) | (((w.y >> mbit) & 1u) << 1) | (((w.z >> mbit) & 1u) << 2) | (((w.w >> mbit) & 1u) << 3)) << (4 * g);
      }
    } else if (a.p_drop > 0.f) {
      const bool odd = lane & 1;
      keepbits = 0u;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int qmine = qb + acc_row(r, h) + (odd ? 1 : 0);
        const uint32_t hm = attn_hash(seed, ((uint64_t)bh * a.Lq + qmine) * (uint64_t)((a.ldp + 1) >> 1) + (uint64_t)(key >> 1));
        const uint32_t ho = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hm, 0xB1, 0xF, 0xF, false);      // lane ^ 1
        const uint32_t h0 = odd ? ho : hm, h1 = odd ? hm : ho;      // rows r, r + 1
        const bool k0 = odd ? attn_keep_hi(h0, thr) : attn_keep_lo(h0, thr);
        const bool k1 = odd ? attn_keep_hi(h1, thr) : attn_keep_lo(h1, thr);
        keepbits |= (k0 ? 1u : 0u) << r;
        keepbits |= (k1 ? 1u : 0u) << (r + 1);
      }
    }
    // lse / D of the lane's 16 query rows (8 g + 4 h + {0 .. 3}) as four 16-byte reads each, and the exponential of a SELECTED ARGUMENT
    // (v_exp_f32(-inf) = 0): `cond ? ex2(s - sL[qr]) : 0.f` became an exec-mask branch per score with its own ds_read_b32 + wait inside
    float lq[16], dq_[16];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const float4 l4 = *reinterpret_cast<const float4*>(&sL[buf][8 * g + 4 * h]), d4 = *reinterpret_cast<const float4*>(&sD[buf][8 * g + 4 * h]);
      lq[4 * g] = l4.x; lq[4 * g + 1] = l4.y; lq[4 * g + 2] = l4.z; lq[4 * g + 3] = l4.w;
      dq_[4 * g] = d4.x; dq_[4 * g + 1] = d4.y; dq_[4 * g + 2] = d4.z; dq_[4 * g + 3] = d4.w;
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int qr = acc_row(r, h);
      // (bitwise: no short-circuit control flow.  A uniform `whole tile` shortcut of these conditions measured SLOWER here: 177 -> 201 us)
      const bool ok = kvalid & (qb + qr < a.Lq) & ((a.causal == 0) | (key <= qb + qr));
      const float x = s[r] * c2 - lq[r];
      const float p = ex2(ok ? x : -INFINITY);
      const bool keep = (keepbits >> r) & 1u;
      const float g = keep ? dp[r] * sc : 0.f;
      pd[r] = keep ? p * sc : 0.f;
      s[r] = p * (g - dq_[r]) * a.scale;
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const bf16x8 ap = pack8(pd, t), as = pack8(s, t);
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        dv[nt] = mfma(ap, tr_frag(sOt[buf], nt * 32, 16 * t + 4 * h, 16 * t + 8 + 4 * h, lane), dv[nt]);
        dk[nt] = mfma(as, tr_frag(sQt[buf], nt * 32, 16 * t + 4 * h, 16 * t + 8 + 4 * h, lane), dk[nt]);
      }
    }
    }
    if (qt + 1 < nqt) {
      *reinterpret_cast<bf16x8*>(&sQc[buf ^ 1][kc_off(lrow, lch)]) = rq;
      *reinterpret_cast<bf16x8*>(&sQt[buf ^ 1][tr_off(lrow, lch)]) = rq;
      *reinterpret_cast<bf16x8*>(&sOc[buf ^ 1][kc_off(lrow, lch)]) = ro;
      *reinterpret_cast<bf16x8*>(&sOt[buf ^ 1][tr_off(lrow, lch)]) = ro;
      if (tid < TT) sL[buf ^ 1][tid] = rs;
      else if (tid < 2 * TT) sD[buf ^ 1][tid - TT] = rs;
      if (mstage) sM[buf ^ 1][tid >> 5][tid & 31] = rm;
    }
    __syncthreads();
  }
  // dk/dv[nt]: lane = column (dh = nt*32 + ln), register r = key k0 + acc_row(r, h)
#pragma unroll
  for (int nt = 0; nt < 2; nt++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int kk = k0 + acc_row(r, h);
      if (kk < a.Lk) {
        a.dK[((long)b * a.Lk + kk) * a.lddk + hd * DH + nt * 32 + ln] = (bf16)dk[nt][r];
        a.dV[((long)b * a.Lk + kk) * a.lddv + hd * DH + nt * 32 + ln] = (bf16)dv[nt][r];
      }
    }
}

// =====================================================================================================================
// backward 2 for ONE key tile (Lk <= 32: the decoder's vision-to-text cross-attention, 20 word keys).  In flash_bwd_dkdv_kernel a wave
// owns 32 keys, so three of the four waves of its single block per (batch, head) would idle while the fourth walks all 22 query tiles
// (54 us in situ at B = 32).  Here the four waves own the SAME 32 keys and take every fourth query tile each, with wave-private tiles and no
// block barrier in the loop: the Q / dO rows go from global memory straight into the MFMA A operand (lane = query row, 16 bytes per k-step:
// what kc_frag reads from the row image elsewhere) and the same registers are stored as the transposed-read image for dV += P^T dO,
// dK += dS^T Q; the next tile's loads are in flight during the products.  dK / dV are summed across the waves through LDS at the end.
// grid (1, B * heads).
// =====================================================================================================================
__global__ void __launch_bounds__(NTHR, 1) flash_bwd_dkdv_short_kernel(const AttnArgs a) {
  const uint64_t seed = a.seed + (a.epoch ? *a.epoch : 0ull);
  __shared__ __attribute__((aligned(16))) bf16 sQt[4][TILE];
  __shared__ __attribute__((aligned(16))) bf16 sOt[4][TILE];
  __shared__ __attribute__((aligned(16))) float sL[4][TT];
  __shared__ __attribute__((aligned(16))) float sD[4][TT];
  __shared__ __attribute__((aligned(16))) uint32_t sM[4][TT];
  __shared__ __attribute__((aligned(16))) float red[3][32][64];      // dK, then dV, of waves 1 .. 3 on their way to wave 0
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, ln = lane & 31;
  const int bh = blockIdx.y, b = bh / a.heads, hd = bh % a.heads;
  const int key = ln;
  const int kc = min(key, a.Lk - 1);
  const bool kvalid = key < a.Lk && !(a.kpm && a.kpm[(long)b * a.Lk + key]);
  bf16x8 kf[4], vf[4];
  {
    const bf16* kp = a.K + ((long)b * a.Lk + kc) * a.ldk + hd * DH + h * 8;
    const bf16* vp = a.V + ((long)b * a.Lk + kc) * a.ldv + hd * DH + h * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(kp + ks * 16);
      vf[ks] = *reinterpret_cast<const bf16x8*>(vp + ks * 16);
    }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int e = 0; e < 16; e++) dk[0][e] = dk[1][e] = dv[0][e] = dv[1][e] = 0.f;

  const bf16* qg = a.Q + (long)b * a.Lq * a.ldq + hd * DH + h * 8;
  const bf16* og = a.dO + (long)b * a.Lq * a.lddo + hd * DH + h * 8;
  const float* lg = a.lse + (long)bh * a.Lq;
  const float* dg = a.D + (long)bh * a.Lq;
  const uint32_t* mg = a.keep ? reinterpret_cast<const uint32_t*>(a.keep) + (long)bh * a.Lq : nullptr;      // one key tile: word (bh, q)
  const int mbit = 16 * ((ln >> 2) & 1) + (ln & 3) + 4 * (ln >> 3);      // (see flash_bwd_dkdv_kernel)
  const int nqt = (a.Lq + TT - 1) / TT;
  const uint32_t thr = attn_thr16(a.p_drop);
  const float sc = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const float c2 = a.scale * LOG2E;

  bf16x8 nq[4], no[4];      // the NEXT tile: row qb + ln, k-steps 16 ks + 8 h
  float rs = 0.f;
  uint32_t rm = 0u;
  auto fetch = [&](int qt) {
    const int qq = min(qt * TT + ln, a.Lq - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      nq[ks] = *reinterpret_cast<const bf16x8*>(qg + (long)qq * a.ldq + ks * 16);
      no[ks] = *reinterpret_cast<const bf16x8*>(og + (long)qq * a.lddo + ks * 16);
    }
    rs = h == 0 ? lg[qq] * LOG2E : dg[qq];      // (log2 units, as in the forward)
    if (mg && h == 0) rm = mg[qq];
  };
  if (wave < nqt) fetch(wave);
  for (int qt = wave; qt < nqt; qt += 4) {
    bf16x8 cq[4], co[4];
    // this wave's earlier reads of its tiles are done (one wave: LDS instructions complete in order); the fences keep the compiler from
    // moving LDS accesses across
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      cq[ks] = nq[ks];
      co[ks] = no[ks];
      *reinterpret_cast<bf16x8*>(&sQt[wave][tr_off(ln, 2 * ks + h)]) = cq[ks];
      *reinterpret_cast<bf16x8*>(&sOt[wave][tr_off(ln, 2 * ks + h)]) = co[ks];
    }
    if (h == 0) { sL[wave][ln] = rs; if (mg) sM[wave][ln] = rm; }
    else sD[wave][ln] = rs;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (qt + 4 < nqt) fetch(qt + 4);
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; e++) s[e] = dp[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      s = mfma(cq[ks], kf[ks], s);      // S[query][key]
      dp = mfma(co[ks], vf[ks], dp);    // (dO V^T)[query][key]
    }
    const int qb = qt * TT;
    unsigned keepbits = 0xffffu;
    if (mg) {
      keepbits = 0u;
      const uint4* mrow = reinterpret_cast<const uint4*>(&sM[wave][0]);
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const uint4 w = mrow[2 * g + h];
        keepbits |= (((w.x >> mbit) & 1u) | (((w.y >> mbit) & 1u) << 1) | (((w.z >> mbit) & 1u) << 2) | (((w.w >> mbit) & 1u) << 3)) << (4 * g);
      }
    } else if (a.p_drop > 0.f) {      // (the pair hash of flash_bwd_dkdv_kernel)
      const bool odd = lane & 1;
      keepbits = 0u;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int qmine = qb + acc_row(r, h) + (odd ? 1 : 0);
        const uint32_t hm = attn_hash(seed, ((uint64_t)bh * a.Lq + qmine) * (uint64_t)((a.ldp + 1) >> 1) + (uint64_t)(key >> 1));
        const uint32_t ho = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hm, 0xB1, 0xF, 0xF, false);      // lane ^ 1
        const uint32_t h0 = odd ? ho : hm, h1 = odd ? hm : ho;
        const bool k0 = odd ? attn_keep_hi(h0, thr) : attn_keep_lo(h0, thr);
        const bool k1 = odd ? attn_keep_hi(h1, thr) : attn_keep_lo(h1, thr);
        keepbits |= (k0 ? 1u : 0u) << r;
        keepbits |= (k1 ? 1u : 0u) << (r + 1);
      }
    }
    f32x16 pd;
    float lq[16], dq_[16];      // (as in flash_bwd_dkdv_kernel)
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const float4 l4 = *reinterpret_cast<const float4*>(&sL[wave][8 * g + 4 * h]), d4 = *reinterpret_cast<const float4*>(&sD[wave][8 * g + 4 * h]);
      lq[4 * g] = l4.x; lq[4 * g + 1] = l4.y; lq[4 * g + 2] = l4.z; lq[4 * g + 3] = l4.w;
      dq_[4 * g] = d4.x; dq_[4 * g + 1] = d4.y; dq_[4 * g + 2] = d4.z; dq_[4 * g + 3] = d4.w;
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int qr = acc_row(r, h);
      const bool ok = kvalid & (qb + qr < a.Lq);
      const float x = s[r] * c2 - lq[r];
      const float p = ex2(ok ? x : -INFINITY);
      const bool keep = (keepbits >> r) & 1u;
      const float g = keep ? dp[r] * sc : 0.f;
      pd[r] = keep ? p * sc : 0.f;
      s[r] = p * (g - dq_[r]) * a.scale;
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const bf16x8 ap = pack8(pd, t), as = pack8(s, t);
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        dv[nt] = mfma(ap, tr_frag(sOt[wave], nt * 32, 16 * t + 4 * h, 16 * t + 8 + 4 * h, lane), dv[nt]);
        dk[nt] = mfma(as, tr_frag(sQt[wave], nt * 32, 16 * t + 4 * h, 16 * t + 8 + 4 * h, lane), dk[nt]);
      }
    }
  }
  // sums over the four waves: dK, then dV, through `red` (row = nt * 16 + r, lane-major)
#pragma unroll
  for (int which = 0; which < 2; which++) {
    f32x16* acc = which == 0 ? dk : dv;
    if (wave > 0) {
#pragma unroll
      for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) red[wave - 1][nt * 16 + r][lane] = acc[nt][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[nt][r] += red[0][nt * 16 + r][lane] + red[1][nt * 16 + r][lane] + red[2][nt * 16 + r][lane];
    }
    __syncthreads();
  }
  if (wave == 0) {
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int kk = acc_row(r, h);
        if (kk < a.Lk) {
          a.dK[((long)b * a.Lk + kk) * a.lddk + hd * DH + nt * 32 + ln] = (bf16)dk[nt][r];
          a.dV[((long)b * a.Lk + kk) * a.lddv + hd * DH + nt * 32 + ln] = (bf16)dv[nt][r];
        }
      }
  }
}

bool aligned8(long ld, const void* p) { return ld % 8 == 0 && ((uintptr_t)p % 16) == 0; }

}  // namespace

extern "C" int crog_flash_attn_fwd_bits(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, void* O, int64_t ldo,
                                        float* lse, int B, int heads, int Lq, int Lk, int head_dim, float scale, float p_drop, uint64_t seed,
                                        int ldp, int causal, const void* key_padding_mask, void* keep_bits, crog_stream_t stream) {
  CROG_CHECK_ARG(head_dim == DH, "flash_attn: head_dim must be %d (got %d)", DH, head_dim);
  CROG_CHECK_ARG(B > 0 && heads > 0 && Lq > 0 && Lk > 0 && ldp >= Lk && p_drop >= 0.f && p_drop < 1.f, "flash_attn_fwd: bad sizes");
  CROG_CHECK_ARG(aligned8(ldq, Q) && aligned8(ldk, K) && aligned8(ldv, V) && ldo % 4 == 0 && ((uintptr_t)O % 8) == 0 && lse,
                 "flash_attn_fwd: Q/K/V rows must be 16-byte aligned (ld %% 8 == 0), O 8-byte aligned");
  CROG_CHECK_ARG((long)B * heads <= 65535, "flash_attn_fwd: B * heads too large");
  AttnArgs a{};
  a.Q = (const bf16*)Q; a.K = (const bf16*)K; a.V = (const bf16*)V; a.Out = (bf16*)O; a.lse = lse;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.Lq = Lq; a.Lk = Lk; a.heads = heads; a.ldp = ldp; a.scale = scale; a.p_drop = p_drop; a.seed = seed; a.epoch = crog_seed_epoch();
  a.causal = causal != 0;
  CROG_CHECK_ARG(!causal || Lq == Lk, "flash_attn: the causal mask is defined for self-attention (Lq == Lk)");
  CROG_CHECK_ARG(((uintptr_t)keep_bits % 4) == 0, "flash_attn_fwd: keep_bits must be 4-byte aligned");
  a.keep = p_drop > 0.f ? (uint16_t*)keep_bits : nullptr;
  a.kpm = (const uint8_t*)key_padding_mask;
  CROG_CHECK_ARG(!(causal && key_padding_mask), "flash_attn: causal and key_padding_mask together are not built");
  hipLaunchKernelGGL(flash_fwd_kernel, dim3(cdiv(Lq, 128), B * heads), dim3(NTHR), 0, (hipStream_t)stream, a);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_flash_attn_fwd_masked(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, void* O, int64_t ldo,
                                          float* lse, int B, int heads, int Lq, int Lk, int head_dim, float scale, float p_drop, uint64_t seed,
                                          int ldp, int causal, crog_stream_t stream) {
  return crog_flash_attn_fwd_bits(Q, ldq, K, ldk, V, ldv, O, ldo, lse, B, heads, Lq, Lk, head_dim, scale, p_drop, seed, ldp, causal, nullptr, nullptr, stream);
}
extern "C" int crog_flash_attn_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, void* O, int64_t ldo,
                                   float* lse, int B, int heads, int Lq, int Lk, int head_dim, float scale, float p_drop, uint64_t seed, int ldp,
                                   crog_stream_t stream) {
  return crog_flash_attn_fwd_masked(Q, ldq, K, ldk, V, ldv, O, ldo, lse, B, heads, Lq, Lk, head_dim, scale, p_drop, seed, ldp, 0, stream);
}

extern "C" int crog_flash_attn_bwd_bits(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, const void* O, int64_t ldo,
                                        const void* dO, int64_t lddo, const float* lse, float* D, void* dQ, int64_t lddq, void* dK, int64_t lddk,
                                        void* dV, int64_t lddv, int B, int heads, int Lq, int Lk, int head_dim, float scale, float p_drop,
                                        uint64_t seed, int ldp, int causal, const void* key_padding_mask, const void* keep_bits, crog_stream_t stream) {
  CROG_CHECK_ARG(head_dim == DH, "flash_attn: head_dim must be %d (got %d)", DH, head_dim);
  CROG_CHECK_ARG(B > 0 && heads > 0 && Lq > 0 && Lk > 0 && ldp >= Lk && p_drop >= 0.f && p_drop < 1.f, "flash_attn_bwd: bad sizes");
  CROG_CHECK_ARG(aligned8(ldq, Q) && aligned8(ldk, K) && aligned8(ldv, V) && aligned8(ldo, O) && aligned8(lddo, dO) && lse && D && dQ && dK && dV,
                 "flash_attn_bwd: Q/K/V/O/dO rows must be 16-byte aligned (ld %% 8 == 0)");
  CROG_CHECK_ARG((long)B * heads <= 65535, "flash_attn_bwd: B * heads too large");
  AttnArgs a{};
  a.Q = (const bf16*)Q; a.K = (const bf16*)K; a.V = (const bf16*)V; a.O = (const bf16*)O; a.dO = (const bf16*)dO;
  a.dQ = (bf16*)dQ; a.dK = (bf16*)dK; a.dV = (bf16*)dV; a.lse = const_cast<float*>(lse); a.D = D;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.Lq = Lq; a.Lk = Lk; a.heads = heads; a.ldp = ldp; a.scale = scale; a.p_drop = p_drop; a.seed = seed; a.epoch = crog_seed_epoch();
  a.causal = causal != 0;
  CROG_CHECK_ARG(!causal || Lq == Lk, "flash_attn: the causal mask is defined for self-attention (Lq == Lk)");
  CROG_CHECK_ARG(((uintptr_t)keep_bits % 4) == 0, "flash_attn_bwd: keep_bits must be 4-byte aligned");
  a.keep = p_drop > 0.f ? (uint16_t*)const_cast<void*>(keep_bits) : nullptr;
  a.kpm = (const uint8_t*)key_padding_mask;
  CROG_CHECK_ARG(!(causal && key_padding_mask), "flash_attn: causal and key_padding_mask together are not built");
  hipLaunchKernelGGL(flash_bwd_dq_kernel, dim3(cdiv(Lq, 128), B * heads), dim3(NTHR), 0, (hipStream_t)stream, a);
  CROG_LAUNCH_CHECK();
  static const bool short_keys = [] { const char* e = getenv("CROG_FLASH_SHORT"); return !e || atoi(e) != 0; }();
  if (Lk <= TT && !causal && Lq >= 4 * TT && short_keys)      // one key tile, many query tiles: the four waves share the keys and split the queries
    hipLaunchKernelGGL(flash_bwd_dkdv_short_kernel, dim3(1, B * heads), dim3(NTHR), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(flash_bwd_dkdv_kernel, dim3(cdiv(Lk, 128), B * heads), dim3(NTHR), 0, (hipStream_t)stream, a);
  CROG_LAUNCH_CHECK();
  return CROG_OK;
}
extern "C" int crog_flash_attn_bwd_masked(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, const void* O, int64_t ldo,
                                          const void* dO, int64_t lddo, const float* lse, float* D, void* dQ, int64_t lddq, void* dK, int64_t lddk,
                                          void* dV, int64_t lddv, int B, int heads, int Lq, int Lk, int head_dim, float scale, float p_drop,
                                          uint64_t seed, int ldp, int causal, crog_stream_t stream) {
  return crog_flash_attn_bwd_bits(Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, lse, D, dQ, lddq, dK, lddk, dV, lddv, B, heads, Lq, Lk, head_dim, scale,
                                  p_drop, seed, ldp, causal, nullptr, nullptr, stream);
}
extern "C" int crog_flash_attn_bwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, const void* O, int64_t ldo,
                                   const void* dO, int64_t lddo, const float* lse, float* D, void* dQ, int64_t lddq, void* dK, int64_t lddk,
                                   void* dV, int64_t lddv, int B, int heads, int Lq, int Lk, int head_dim, float scale, float p_drop,
                                   uint64_t seed, int ldp, crog_stream_t stream) {
  return crog_flash_attn_bwd_masked(Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, lse, D, dQ, lddq, dK, lddk, dV, lddv, B, heads, Lq, Lk, head_dim, scale,
                                    p_drop, seed, ldp, 0, stream);
}
