// Collectives of the data-parallel step behind the C ABI (include/crog_hip.h: crog_comm_*, crog_allreduce_bucket, crog_syncbn_stats;
// SURVEY.md §8b).  Replaces what the reference gets from torch.distributed / NCCL: the DDP gradient all-reduce
// (train_crog.py:154-156), the SyncBatchNorm statistics exchange (train_crog.py:113-114) and the packed metric all-reduce
// (crog_engine.py:88-90).
//
// Two transports behind one handle:
//  * RCCL (ncclAllReduce over xGMI) for the large, asynchronous gradient buckets.  The library is the copy the process already has
//    resident (PyTorch-ROCm's librccl.so) and is bound at run time with dlopen / dlsym: libcrog_hip.so itself has no link-time
//    dependency on it, so it also loads on a box without RCCL (the CPU-side ABI tests).
//  * a one-shot peer-write all-reduce for the BatchNorm statistics: 142 exchanges of 2·C floats per CROG step, every one of them on
//    the critical path (a BatchNorm layer cannot normalise before the global sums exist), where a ring collective costs 20-40 us of
//    latency per call.  Every rank owns a mailbox in uncached device memory, opened by its peers through hipIpc.  An exchange is ONE
//    single-block kernel per rank: write my n floats into slot [parity][my rank] of every rank's mailbox (peer stores over xGMI),
//    workgroup barrier, ONE system-scope release fence by the publishing lanes, the sequence number stored (relaxed) in every mailbox's
//    flag word, relaxed polls until all `world` flags of my own mailbox carry this sequence number, one acquire fence, then the slots
//    added up in rank order - the same order on every rank, so all ranks hold bit-identical sums (comm_dev.h: crog_peer_exchange).
//    One hop, no ring; the latency is a peer write + a flag poll.
//    Two slot sets (sequence parity): a rank can only run ahead of a peer by one exchange - it needs that peer's contribution to finish
//    the next one - so exchange e + 2 can never overwrite data a slower peer is still reading from exchange e.
//    The sequence number lives in device memory and is advanced by the kernel, so the launch carries no per-step scalar and can be
//    captured / replayed (crog_amd/graphs.py).  Every wait is bounded (120 s of wall clock by default, CROG_COMM_TIMEOUT_S): a missing peer raises the mailbox's error word
//    instead of hanging the GPU.
#include "comm_dev.h"

#include <dlfcn.h>
#include <string.h>

#include <string>
#include <vector>

namespace {

#define CM_HIP(expr, what)                                                          \
  do {                                                                              \
    hipError_t e__ = (expr);                                                        \
    if (e__ != hipSuccess) {                                                        \
      crog_set_error("comm: %s failed: %s", what, hipGetErrorString(e__));          \
      return CROG_ERR_LAUNCH;                                                       \
    }                                                                               \
  } while (0)

// ---- RCCL, bound at run time ---------------------------------------------------------------------------------------------------
struct NcclUid { char internal[128]; };
enum { NCCL_INT8 = 0, NCCL_FLOAT32 = 7, NCCL_BFLOAT16 = 9 };   // ncclDataType_t (nccl.h; RCCL keeps NCCL's values)
enum { NCCL_SUM = 0, NCCL_MAX = 2, NCCL_AVG = 4 };             // ncclRedOp_t
struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(NcclUid*) = nullptr;
  int (*CommInitRank)(void**, int, NcclUid, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;      // optional (crog_comm_set_bucket_algo)
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::string g_rccl_err;

bool rccl_bind() {
  if (g_rccl.so) return true;
  const char* names[] = {getenv("CROG_RCCL_LIB"), "librccl.so", "librccl.so.1"};
  void* so = nullptr;
  for (const char* n : names) {                      // the copy that is already resident (PyTorch-ROCm's) wins: two RCCLs in one process
    if (!n) continue;                                // would each build their own topology and rings
    so = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (so) break;
  }
  for (const char* n : names) {
    if (so) break;
    if (n) so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!so) {
    g_rccl_err = std::string("librccl.so is not loadable (") + (dlerror() ? dlerror() : "?") + ")";
    return false;
  }
  Rccl r;
  r.so = so;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(so, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(so, "ncclCommInitRank");
  r.AllReduce = (decltype(r.AllReduce))dlsym(so, "ncclAllReduce");
  r.Broadcast = (decltype(r.Broadcast))dlsym(so, "ncclBroadcast");
  r.ReduceScatter = (decltype(r.ReduceScatter))dlsym(so, "ncclReduceScatter");
  r.AllGather = (decltype(r.AllGather))dlsym(so, "ncclAllGather");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(so, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(so, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.GetErrorString) {
    g_rccl_err = "librccl.so lacks an expected symbol";
    return false;
  }
  g_rccl = r;
  return true;
}
#define CM_NCCL(expr, what)                                                                   \
  do {                                                                                        \
    int rc__ = (expr);                                                                        \
    if (rc__ != 0) {                                                                          \
      crog_set_error("comm: %s failed: %s (%d)", what, g_rccl.GetErrorString(rc__), rc__);    \
      return CROG_ERR_LAUNCH;                                                                 \
    }                                                                                         \
  } while (0)

// ---- the mailbox of the peer-write all-reduce ----------------------------------------------------------------------------------
// layout (floats / 32-bit words):  [0 .. 2*W*S)            data  [parity][rank][S]
//                                  [2*W*S .. 2*W*S + 2*W)  flags [parity][rank]   (sequence number of the exchange that filled the slot)
//                                  then: seq (this rank's exchange counter), err (non-zero after a timed-out wait)
constexpr int MAX_WORLD = CROG_MAX_WORLD;
typedef CrogPeerPtrs PeerPtrs;

struct Comm {
  int rank = 0, world = 1;
  void* nccl = nullptr;
  // peer mailbox
  float* box = nullptr;          // mine
  size_t box_bytes = 0;
  int slot = 0;                  // floats per slot (S)
  bool connected = false;
  PeerPtrs peers{};
  std::vector<void*> opened;     // hipIpcOpenMemHandle mappings to close
  CrogSyncBlock* sync_dev = nullptr;      // device copy of what a kernel-tail exchange needs (crog_comm_sync_block)
  int bucket_algo = 0;           // crog_allreduce_bucket: 0 = ncclAllReduce, 1 = ncclReduceScatter + ncclAllGather (crog_comm_set_bucket_algo)
};

// How long a rank waits for its peers inside one exchange before it gives up and raises the mailbox's error word.  Ranks of one job
// are NOT in lock step at the first exchange (code-object loading, a slow data loader; two test processes time-slicing one GPU were
// seen > 2 s apart), so the default is generous - 120 s, RCCL's own watchdog is minutes - and CROG_COMM_TIMEOUT_S overrides it.
inline unsigned long long peer_wait_ticks() {
  static const unsigned long long t = [] {
    const char* e = getenv("CROG_COMM_TIMEOUT_S");
    double sec = e ? atof(e) : 120.0;
    if (!(sec > 0.0)) sec = 120.0;
    return (unsigned long long)(sec * 1e8);
  }();
  return t;
}

__global__ void __launch_bounds__(256) peer_allreduce_kernel(float* __restrict__ x, int n, PeerPtrs peers, int rank, int world, int S, unsigned long long wait_ticks) {
  crog_peer_exchange(x, n, peers, rank, world, S, wait_ticks);
}

}  // namespace

extern "C" int crog_comm_unique_id(void* id128) {
  CROG_CHECK_ARG(id128 != nullptr, "comm_unique_id: null buffer");
  if (!rccl_bind()) {
    crog_set_error("comm_unique_id: %s", g_rccl_err.c_str());
    return CROG_ERR_LAUNCH;
  }
  NcclUid uid;
  CM_NCCL(g_rccl.GetUniqueId(&uid), "ncclGetUniqueId");
  memcpy(id128, &uid, sizeof(uid));
  return CROG_OK;
}

extern "C" int crog_comm_init(int rank, int world, const void* id128, void** comm_out) {
  CROG_CHECK_ARG(comm_out && world >= 1 && world <= MAX_WORLD && rank >= 0 && rank < world, "comm_init: rank %d of world %d (max %d)", rank, world, MAX_WORLD);
  auto* c = new Comm();
  c->rank = rank;
  c->world = world;
  if (id128) {       // NULL: a peer-mailbox-only communicator (two test ranks sharing one GPU, which RCCL refuses)
    if (!rccl_bind()) {
      delete c;
      crog_set_error("comm_init: %s", g_rccl_err.c_str());
      return CROG_ERR_LAUNCH;
    }
    NcclUid uid;
    memcpy(&uid, id128, sizeof(uid));
    int rc = g_rccl.CommInitRank(&c->nccl, world, uid, rank);
    if (rc != 0) {
      delete c;
      crog_set_error("comm_init: ncclCommInitRank failed: %s (%d)", g_rccl.GetErrorString(rc), rc);
      return CROG_ERR_LAUNCH;
    }
  }
  *comm_out = c;
  return CROG_OK;
}

extern "C" int crog_comm_peer_handle(void* comm, int slot_floats, void* handle64) {
  CROG_CHECK_ARG(comm && handle64 && slot_floats >= 64 && slot_floats % 4 == 0, "comm_peer_handle: slot_floats must be a multiple of 4, >= 64");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  auto* c = (Comm*)comm;
  CROG_CHECK_ARG(c->box == nullptr, "comm_peer_handle: the mailbox already exists");
  c->slot = slot_floats;
  c->box_bytes = ((size_t)2 * c->world * slot_floats + 2 * c->world + 64) * 4;
  // uncached: peers poll and write this memory while kernels of this device run; a cached line would not see them
  CM_HIP(hipExtMallocWithFlags((void**)&c->box, c->box_bytes, hipDeviceMallocUncached), "hipExtMallocWithFlags(uncached)");
  CM_HIP(hipMemset(c->box, 0, c->box_bytes), "hipMemset");
  CM_HIP(hipDeviceSynchronize(), "hipDeviceSynchronize");
  hipIpcMemHandle_t h;
  CM_HIP(hipIpcGetMemHandle(&h, c->box), "hipIpcGetMemHandle");
  memcpy(handle64, &h, 64);
  return CROG_OK;
}

extern "C" int crog_comm_peer_connect(void* comm, const void* handles) {
  CROG_CHECK_ARG(comm && handles, "comm_peer_connect: null argument");
  auto* c = (Comm*)comm;
  CROG_CHECK_ARG(c->box != nullptr && !c->connected, "comm_peer_connect: call crog_comm_peer_handle first (once)");
  for (int r = 0; r < c->world; r++) {
    if (r == c->rank) {
      c->peers.box[r] = c->box;
      continue;
    }
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + 64 * (size_t)r, 64);
    void* p = nullptr;
    CM_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
    c->opened.push_back(p);
    c->peers.box[r] = (float*)p;
  }
  c->connected = true;
  return CROG_OK;
}

extern "C" int crog_comm_sync_block(void* comm, void** dev_block) {
  CROG_CHECK_ARG(comm && dev_block, "comm_sync_block: null argument");
  auto* c = (Comm*)comm;
  CROG_CHECK_ARG(c->connected, "comm_sync_block: the communicator has no peer mailboxes (crog_comm_peer_connect)");
  if (!c->sync_dev) {
    CrogSyncBlock h{};
    h.peers = c->peers;
    h.rank = c->rank;
    h.world = c->world;
    h.S = c->slot;
    h.wait_ticks = peer_wait_ticks();
    CM_HIP(hipMalloc((void**)&c->sync_dev, sizeof(CrogSyncBlock)), "hipMalloc(sync block)");
    CM_HIP(hipMemcpy(c->sync_dev, &h, sizeof(h), hipMemcpyHostToDevice), "hipMemcpy(sync block)");
  }
  *dev_block = c->sync_dev;
  return CROG_OK;
}

extern "C" int crog_comm_status(void* comm, int* timed_out_seq) {
  CROG_CHECK_ARG(comm && timed_out_seq, "comm_status: null argument");
  auto* c = (Comm*)comm;
  *timed_out_seq = 0;
  if (c->box) {
    unsigned tail[2] = {0, 0};
    CM_HIP(hipMemcpy(tail, c->box + (size_t)2 * c->world * c->slot + 2 * c->world, 8, hipMemcpyDeviceToHost), "hipMemcpy");
    *timed_out_seq = (int)tail[1];
  }
  return CROG_OK;
}

extern "C" int crog_syncbn_stats(void* comm, float* ptr, int64_t count, crog_stream_t stream) {
  CROG_CHECK_ARG(comm && ptr && count > 0, "syncbn_stats: null argument");
  auto* c = (Comm*)comm;
  // (a one-rank communicator WITH a mailbox - bench.py's CROG_FORCE_DDP=1 - runs the protocol against its own mailbox, so that the
  // world-size-1 number carries the launch and fence cost of every exchange; without a mailbox a one-rank sum is the identity)
  if (c->world == 1 && !c->connected) return CROG_OK;
  if (c->connected && count <= c->slot) {
    hipLaunchKernelGGL(peer_allreduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ptr, (int)count, c->peers, c->rank, c->world, c->slot, peer_wait_ticks());
    CROG_LAUNCH_CHECK();
    return CROG_OK;
  }
  CROG_CHECK_ARG(c->nccl != nullptr, "syncbn_stats: %ld floats do not fit the peer mailbox (%d) and there is no RCCL communicator", (long)count, c->slot);
  CM_NCCL(g_rccl.AllReduce(ptr, ptr, (size_t)count, NCCL_FLOAT32, NCCL_SUM, c->nccl, (hipStream_t)stream), "ncclAllReduce");
  return CROG_OK;
}

// How crog_allreduce_bucket moves a gradient bucket (588 MB of gradients per step; a single ring is bound by ONE xGMI
// link).  0: one ncclAllReduce - the schedule is RCCL's choice.  1: ncclReduceScatter + ncclAllGather on the same stream, in place (rank r reduces
// chunk r of the bucket, then every rank gathers the chunks): the two-phase form whose traffic is spread over all peers by construction; the
// count's remainder modulo the world size goes through a small ncclAllReduce.  Which one is faster is a property of the node: the Python side
// times both on a full-size bucket at start-up and sets the faster one on every rank (rccl.DirectComm.tune_bucket_algo: collective verdict).
extern "C" int crog_comm_set_bucket_algo(void* comm, int algo) {
  CROG_CHECK_ARG(comm && (algo == 0 || algo == 1), "comm_set_bucket_algo: algo is 0 (all-reduce) or 1 (reduce-scatter + all-gather)");
  auto* c = (Comm*)comm;
  CROG_CHECK_ARG(algo == 0 || (c->nccl && g_rccl.ReduceScatter && g_rccl.AllGather), "comm_set_bucket_algo: no RCCL communicator, or librccl without ncclReduceScatter / ncclAllGather");
  c->bucket_algo = algo;
  return CROG_OK;
}

extern "C" int crog_allreduce_bucket(void* comm, void* ptr, int64_t count, int dtype, int average, crog_stream_t stream) {
  CROG_CHECK_ARG(comm && ptr && count > 0 && (dtype == CROG_F32 || dtype == CROG_BF16), "allreduce_bucket: bad argument");
  auto* c = (Comm*)comm;
  CROG_CHECK_ARG(c->nccl != nullptr, "allreduce_bucket: the communicator was created without RCCL (peer-only)");
  const int nt = dtype == CROG_BF16 ? NCCL_BFLOAT16 : NCCL_FLOAT32, op = average ? NCCL_AVG : NCCL_SUM;
  const size_t esz = dtype == CROG_BF16 ? 2 : 4;
  const int64_t chunk = (count / c->world) & ~(int64_t)63;      // per-rank chunk, whole 256-byte runs
  if (c->bucket_algo == 1 && chunk > 0) {
    char* base = (char*)ptr;
    CM_NCCL(g_rccl.ReduceScatter(base, base + (size_t)c->rank * chunk * esz, (size_t)chunk, nt, op, c->nccl, (hipStream_t)stream), "ncclReduceScatter");
    CM_NCCL(g_rccl.AllGather(base + (size_t)c->rank * chunk * esz, base, (size_t)chunk, nt, c->nccl, (hipStream_t)stream), "ncclAllGather");
    const int64_t done = chunk * c->world;
    if (done < count)
      CM_NCCL(g_rccl.AllReduce(base + (size_t)done * esz, base + (size_t)done * esz, (size_t)(count - done), nt, op, c->nccl, (hipStream_t)stream), "ncclAllReduce");
    return CROG_OK;
  }
  CM_NCCL(g_rccl.AllReduce(ptr, ptr, (size_t)count, nt, op, c->nccl, (hipStream_t)stream), "ncclAllReduce");
  return CROG_OK;
}

extern "C" int crog_comm_destroy(void* comm) {
  if (!comm) return CROG_OK;
  auto* c = (Comm*)comm;
  for (void* p : c->opened) (void)hipIpcCloseMemHandle(p);
  if (c->box) (void)hipFree(c->box);
  if (c->sync_dev) (void)hipFree(c->sync_dev);
  if (c->nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->nccl);
  delete c;
  return CROG_OK;
}
