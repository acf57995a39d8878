// Device side of the peer-write all-reduce (csrc/comm.hip) shared with the kernels that carry an exchange in their own tail
// (round 5: the producers of the SyncBatchNorm BACKWARD statistics - bn_bwd_partial_kernel, the bwd_z epilogues of the data-gradient GEMMs -
// finish with it, so that the exchange is not a launch of its own; train_crog.py:113-114).  gfx950 only.
#pragma once
#include "common.h"

constexpr int CROG_MAX_WORLD = 16;
struct CrogPeerPtrs { float* box[CROG_MAX_WORLD]; };
// What a kernel needs to run one exchange: lives in device memory, one per communicator (crog_comm_sync_block)
struct CrogSyncBlock {
  CrogPeerPtrs peers;
  int rank, world, S;                 // S: floats per mailbox slot
  unsigned long long wait_ticks;      // bound of a wait for the peers, in ticks of the 100 MHz wall clock
};

// One exchange by ONE workgroup (all its threads call): x[0 .. n) (global or LDS, fp32) becomes the sum over the ranks, added in rank
// order (bit-identical on every rank).  Mailbox layout and protocol: csrc/comm.hip.  A rank whose earlier exchange timed out, or whose
// peers do not show up within wait_ticks, gets NaN (and the mailbox's error word set).  n <= S.
__device__ inline void crog_peer_exchange(float* x, int n, const CrogPeerPtrs& peers, int rank, int world, int S, unsigned long long wait_ticks) {
  float* mine = peers.box[rank];
  unsigned* tail = reinterpret_cast<unsigned*>(mine + (size_t)2 * world * S + 2 * world);   // [seq, err]
  __shared__ unsigned s_seq, s_bad, s_dead;
  if (threadIdx.x == 0) {
    s_seq = tail[0] + 1u;
    tail[0] = s_seq;
    s_bad = 0u;
    s_dead = tail[1];
  }
  __syncthreads();
  const unsigned seq = s_seq;
  if (s_dead) {
    // an earlier exchange of this communicator timed out: its sequence numbers and slot parity are no longer aligned with the peers',
    // so nothing it could deliver is trustworthy.  Poison the statistics (NaN reaches the loss within one layer; the engine polls
    // crog_comm_status and raises) instead of exchanging
    for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = __builtin_nanf("");
    __syncthreads();
    return;
  }
  const int par = (int)(seq & 1u);
  // 1. my contribution into slot [par][rank] of EVERY mailbox (my own included: one code path, one summation order)
  for (int r = 0; r < world; r++) {
    float* dst = peers.box[r] + ((size_t)par * world + rank) * S;
    for (int i = threadIdx.x; i < n; i += blockDim.x) __builtin_nontemporal_store(x[i], dst + i);
  }
  // Round 6: ONE system-scope write-back per exchange instead of three (and by one wave instead of four).  Before: every thread ran
  // __threadfence_system() (buffer_wbl2 + buffer_inv per wave), the flag store was a release store of its own (another write-back) and a third
  // fence followed the wait - each write-back flushes whatever the XCD's L2 holds dirty, and an exchange in a GEMM's tail runs right behind
  // that GEMM's output stores (forced DDP at world size 1: 29.15 -> see LAB_NOTES section 11).  Now: the block meets (a workgroup barrier
  // completes every wave's stores: s_waitcnt vmcnt(0) + s_barrier), THEN the publishing lanes - one wave - execute a system-scope RELEASE
  // fence, which covers every store that happens-before it, the other waves' included (the barrier orders them), and store the flags
  // RELAXED behind it (fence-atomic synchronisation with the peer's acquire fence below).
  __syncthreads();
  // 2. publish: flag [par][rank] of every mailbox = seq
  if ((int)threadIdx.x < world) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    unsigned* f = reinterpret_cast<unsigned*>(peers.box[threadIdx.x] + (size_t)2 * world * S) + par * world + rank;
    __hip_atomic_store(f, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 3. wait for every rank's flag in MY mailbox (bounded)
  if ((int)threadIdx.x < world) {
    const unsigned* f = reinterpret_cast<const unsigned*>(mine + (size_t)2 * world * S) + par * world + threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    // RELAXED polls + ONE acquire fence once the flag has arrived (round 6): an acquire load per iteration is a system-scope L2 invalidate
    // per poll (buffer_inv sc0 sc1 in the loop), i.e. a rank waiting for a late peer kept throwing out the cache lines of whatever ran
    // beside it.  The fence after the loop synchronises with the peer's release store exactly as the acquire load did (fence-atomic rule);
    // the barrier below hands that to the other threads of the block.
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      if (wall_clock64() - t0 > wait_ticks) {
        atomicOr(&s_bad, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  __syncthreads();
  if (s_bad) {
    // A peer did not show up in time.  This rank has already published its data and flag, so the late peer may still complete with
    // the correct sums while this rank cannot: returning the local sums would let the ranks diverge silently.  Make it loud: NaN
    // statistics (the loss and bench.py's finite check trip on the same step) and the error word for crog_comm_status.
    if (threadIdx.x == 0) tail[1] = seq;
    for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = __builtin_nanf("");
    __syncthreads();
    return;
  }
  // (the polling lanes' acquire fence above + the barrier order the loads below behind the peers' stores: no further fence)
  // 4. sum in rank order (identical on every rank)
  const float* slots = mine + (size_t)par * world * S;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float acc = 0.f;
    for (int r = 0; r < world; r++) acc += __builtin_nontemporal_load(slots + (size_t)r * S + i);
    x[i] = acc;
  }
  __syncthreads();
}

// Tail of a kernel whose blocks have just ADDED their per-channel sums atomically into `sums` = R replica rows of n2 floats
// ([R][C][2], n2 = 2 C), followed in memory by n2 floats for the result and one counter word (zero before the launch):
//     [R * n2 replica rows][n2 totals][counter]
// Every block calls it with all its threads after its last atomic add.  The block that arrives LAST adds the R rows up, exchanges the
// n2 sums with the other ranks (sb != NULL; world 1 runs the protocol against its own mailbox) and stores the totals: the next kernel
// on the stream reads `sums + R * n2` as ONE row of global totals.  No block waits for another one (nothing spins while work is
// queued behind it: several processes can share a GPU), and the exchange is not a launch of its own.
__device__ inline void crog_stat_sync_tail(const CrogSyncBlock* sb, float* sums, int R, int n2, unsigned nblocks) {
  __shared__ unsigned s_last;
  // Every wave waits for its own atomic adds (they are counted in vmcnt and performed at the memory side: an agent-scope atomic does not
  // live in an XCD's L2), the block meets, ONE thread takes the ticket - itself an agent-scope atomic behind those.  No per-block
  // __threadfence(): an L2 write-back per wave of every block cost 120 us per launch (forced DDP at world size 1: 39.6 ms against 30.9 ms
  // with the exchanges as launches of their own); nothing this hand-off reads was written by a plain store.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned* counter = reinterpret_cast<unsigned*>(sums + (size_t)(R + 1) * n2);
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  // the last block (once per launch): drop what this CU's L1 may hold of the rows, then read them past L1 (sc1: agent-scope loads; the
  // atomics left no copy of the lines in any L2)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* tot = sums + (size_t)R * n2;
  for (int i = threadIdx.x; i < n2; i += blockDim.x) {
    float a = 0.f;
    for (int r = 0; r < R; r++) a += __hip_atomic_load(sums + (size_t)r * n2 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tot[i] = a;
  }
  __syncthreads();
  if (sb) crog_peer_exchange(tot, n2, sb->peers, sb->rank, sb->world, sb->S, sb->wait_ticks);
}
