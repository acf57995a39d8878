// Stream-faithful replay of a captured training step (include/crog_hip.h: crog_replay_*).
//
// Why not hipGraphLaunch: the step's concurrency is three long stream-ordered chains (main: forward + data gradients + BatchNorm; weight
// gradients; text tower) with ~180 fork / join edges between them.  ROCm's graph executor re-partitions such a graph over its own pool of
// queues in segments - measured on MI355X, ROCm 7.0: the critical chain hops between queues and the chains overlap far less than the
// streams they were captured from (38.6 ms per step replayed by hipGraphLaunch, 40.0 ms on one stream, 33.4 ms issued eagerly from
// Python).  What the step needs from a graph is only that the HOST stops paying ~25 us of Python per launch.
//
// So: the step is captured once with ordinary stream capture (torch drives it; ATen and RCCL launches are captured too), and this file
// walks the captured hipGraph - kernel / memset / memcpy nodes with their dependency edges - recovers the stream-ordered chains (a
// node continues the chain of its FIRST dependency when that dependency is the chain's tail: stream capture lists the in-stream
// predecessor first), and re-issues the nodes in capture order on the caller's own streams, one chain per stream, with an event
// record / stream wait for every edge that crosses chains (redundant ones pruned).  Same kernels, same arguments (the node-owned
// copies), same streams and the same edges as the eager step: ~3 us of host time per launch from one C loop.
#include "common.h"
#include <hip/hip_ext.h>
#include <stdlib.h>

#include <stdlib.h>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

#define RP_HIP(expr, what)                                                          \
  do {                                                                              \
    hipError_t e__ = (expr);                                                        \
    if (e__ != hipSuccess) {                                                        \
      crog_set_error("replay: %s failed: %s", what, hipGetErrorString(e__));        \
      return CROG_ERR_LAUNCH;                                                       \
    }                                                                               \
  } while (0)

struct RNode {
  hipGraphNodeType type;
  int chain = -1;
  bool module_launch = false;   // func is a hipFunction_t (hipModuleLaunchKernel capture), not a host stub
  hipKernelNodeParams kp{};
  hipMemsetParams ms{};
  hipMemcpy3DParms cp{};
  std::vector<int> waits;       // events (indices into Replay::events) this node's stream waits for before the launch
  int record = -1;              // event recorded after the launch (the node has a dependent on another chain)
  int prof = -1;                // index of its timer pair while profiling, -1 = none
};

struct Replay {
  hipGraph_t graph = nullptr;
  std::vector<RNode> nodes;                 // in issue order (capture order, verified topological)
  std::vector<hipGraphNode_t> handles;      // same order
  std::vector<hipEvent_t> events;
  std::vector<hipEvent_t> tail;             // per chain: end-of-launch join into streams[0]
  hipEvent_t start = nullptr;
  int nchains = 0, nkernels = 0, ncross = 0, nwaits = 0;
  std::vector<int> chain_size;
  std::vector<hipEvent_t> prof_ev;          // 2 per profiled node
  bool profiling = false;
};

int issue(const RNode& n, hipStream_t s) {
  switch (n.type) {
    case hipGraphNodeTypeKernel:
      if (n.module_launch) {
        RP_HIP(hipModuleLaunchKernel((hipFunction_t)n.kp.func, n.kp.gridDim.x, n.kp.gridDim.y, n.kp.gridDim.z, n.kp.blockDim.x, n.kp.blockDim.y,
                                     n.kp.blockDim.z, n.kp.sharedMemBytes, s, n.kp.kernelParams, n.kp.extra), "hipModuleLaunchKernel");
      } else {
        RP_HIP(hipLaunchKernel(n.kp.func, n.kp.gridDim, n.kp.blockDim, n.kp.kernelParams, n.kp.sharedMemBytes, s), "hipLaunchKernel");
      }
      return CROG_OK;
    case hipGraphNodeTypeMemset: {
      const size_t count = n.ms.width * (n.ms.height ? n.ms.height : 1);
      if (n.ms.elementSize == 4) RP_HIP(hipMemsetD32Async((hipDeviceptr_t)n.ms.dst, (int)n.ms.value, count, s), "hipMemsetD32Async");
      else if (n.ms.elementSize == 2) RP_HIP(hipMemsetD16Async((hipDeviceptr_t)n.ms.dst, (unsigned short)n.ms.value, count, s), "hipMemsetD16Async");
      else RP_HIP(hipMemsetAsync(n.ms.dst, (int)n.ms.value, count, s), "hipMemsetAsync");
      return CROG_OK;
    }
    case hipGraphNodeTypeMemcpy:
      RP_HIP(hipMemcpyAsync(n.cp.dstPtr.ptr, n.cp.srcPtr.ptr, n.cp.extent.width, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
      return CROG_OK;
    default:
      return CROG_OK;   // empty nodes: ordering only
  }
}

}  // namespace

extern "C" int crog_capture_last_node(crog_stream_t stream, void** node_out) {
  CROG_CHECK_ARG(node_out != nullptr, "capture_last_node: null output");
  *node_out = nullptr;
  hipStreamCaptureStatus st;
  unsigned long long id = 0;
  hipGraph_t g = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t nd = 0;
  RP_HIP(hipStreamGetCaptureInfo_v2((hipStream_t)stream, &st, &id, &g, &deps, &nd), "hipStreamGetCaptureInfo_v2");
  if (st == hipStreamCaptureStatusActive && nd >= 1) *node_out = (void*)deps[nd - 1];
  return CROG_OK;
}

namespace {
// tags: node -> chain index known from the capture (the stream the launch went to), or null: topology only
int build_replay(void* hip_graph, int max_chains, const std::unordered_map<hipGraphNode_t, int>* tags, int tagged_chains, void** replay_out) {
  CROG_CHECK_ARG(hip_graph && replay_out && max_chains >= 1, "replay_build: graph, output and max_chains >= 1 are required");
  hipGraph_t graph = (hipGraph_t)hip_graph;
  size_t n = 0;
  RP_HIP(hipGraphGetNodes(graph, nullptr, &n), "hipGraphGetNodes");
  CROG_CHECK_ARG(n > 0, "replay_build: the graph has no nodes");
  std::vector<hipGraphNode_t> hs(n);
  RP_HIP(hipGraphGetNodes(graph, hs.data(), &n), "hipGraphGetNodes");
  std::unordered_map<hipGraphNode_t, int> index;
  for (size_t i = 0; i < n; i++) index[hs[i]] = (int)i;
  std::vector<std::vector<int>> deps(n);
  for (size_t i = 0; i < n; i++) {
    size_t nd = 0;
    RP_HIP(hipGraphNodeGetDependencies(hs[i], nullptr, &nd), "hipGraphNodeGetDependencies");
    if (!nd) continue;
    std::vector<hipGraphNode_t> d(nd);
    RP_HIP(hipGraphNodeGetDependencies(hs[i], d.data(), &nd), "hipGraphNodeGetDependencies");
    for (size_t j = 0; j < nd; j++) {
      auto it = index.find(d[j]);
      CROG_CHECK_ARG(it != index.end(), "replay_build: dependency outside the graph");
      deps[i].push_back(it->second);
    }
  }
  // issue order: Kahn's algorithm, always taking the lowest capture index that is ready - capture order when that is topological
  std::vector<int> indeg(n), order;
  std::vector<std::vector<int>> out(n);
  for (size_t i = 0; i < n; i++) {
    indeg[i] = (int)deps[i].size();
    for (int d : deps[i]) out[d].push_back((int)i);
  }
  {
    std::vector<int> ready;
    for (size_t i = 0; i < n; i++) if (!indeg[i]) ready.push_back((int)i);
    std::make_heap(ready.begin(), ready.end(), std::greater<int>());
    while (!ready.empty()) {
      std::pop_heap(ready.begin(), ready.end(), std::greater<int>());
      const int v = ready.back();
      ready.pop_back();
      order.push_back(v);
      for (int w : out[v])
        if (--indeg[w] == 0) {
          ready.push_back(w);
          std::push_heap(ready.begin(), ready.end(), std::greater<int>());
        }
    }
    CROG_CHECK_ARG(order.size() == n, "replay_build: the graph has a cycle");
  }
  std::vector<int> pos(n);   // capture index -> issue position
  for (size_t p = 0; p < n; p++) pos[order[p]] = (int)p;

  auto* R = new Replay();
  R->graph = graph;
  R->nodes.resize(n);
  R->handles.resize(n);
  std::vector<int> chain_tail(tagged_chains, -1);   // per chain: issue position of its last node
  for (size_t p = 0; p < n; p++) {
    const int v = order[p];
    RNode& nd = R->nodes[p];
    R->handles[p] = hs[v];
    RP_HIP(hipGraphNodeGetType(hs[v], &nd.type), "hipGraphNodeGetType");
    switch (nd.type) {
      case hipGraphNodeTypeKernel: {
        RP_HIP(hipGraphKernelNodeGetParams(hs[v], &nd.kp), "hipGraphKernelNodeGetParams");
        hipFuncAttributes attr;
        nd.module_launch = hipFuncGetAttributes(&attr, nd.kp.func) != hipSuccess;   // not a host stub of this process: a hipFunction_t
        if (nd.module_launch) (void)hipGetLastError();
        if (!nd.module_launch && nd.kp.kernelParams == nullptr) {
          delete R;
          crog_set_error("replay_build: kernel node %zu carries its arguments in `extra` (unsupported)", p);
          return CROG_ERR_ARG;
        }
        R->nkernels++;
        break;
      }
      case hipGraphNodeTypeMemset:
        RP_HIP(hipGraphMemsetNodeGetParams(hs[v], &nd.ms), "hipGraphMemsetNodeGetParams");
        if (nd.ms.height > 1 && nd.ms.pitch != nd.ms.width * nd.ms.elementSize) {
          delete R;
          crog_set_error("replay_build: pitched 2-D memset node (unsupported)");
          return CROG_ERR_ARG;
        }
        break;
      case hipGraphNodeTypeMemcpy:
        RP_HIP(hipGraphMemcpyNodeGetParams(hs[v], &nd.cp), "hipGraphMemcpyNodeGetParams");
        if (nd.cp.extent.height > 1 || nd.cp.extent.depth > 1 || nd.cp.extent.width == 0 || !nd.cp.dstPtr.ptr || !nd.cp.srcPtr.ptr ||
            nd.cp.kind == hipMemcpyHostToDevice || nd.cp.kind == hipMemcpyDeviceToHost) {
          delete R;
          crog_set_error("replay_build: only 1-D device-to-device memcpy nodes are supported");
          return CROG_ERR_ARG;
        }
        break;
      case hipGraphNodeTypeEmpty:
        break;
      default:
        delete R;
        crog_set_error("replay_build: node type %d (host / child-graph / event / mem-alloc node) is not supported", (int)nd.type);
        return CROG_ERR_ARG;
    }
    // chain = the chain of the first dependency whose node is still a chain tail (stream capture lists the in-stream predecessor first)
    int chain = -1;
    if (tags) {
      // the capture knows the stream of every launch of this library; the few others (ATen, memsets, a process group's collectives)
      // take the topological rule below
      auto it = tags->find(hs[v]);
      if (it != tags->end()) chain = it->second;
    }
    for (int d : deps[v]) {
      if (chain >= 0) break;
      const int c = R->nodes[pos[d]].chain;
      if (chain_tail[c] == pos[d]) {
        chain = c;
        break;
      }
    }
    if (chain < 0 && tags) {
      // an untagged node without an in-stream predecessor (a root: the first ATen fill of the step; a node after a fork): stay with the
      // first dependency's chain - a tagged capture knows its streams, a stray chain would only add a queue
      chain = deps[v].empty() ? 0 : R->nodes[pos[deps[v][0]]].chain;
    }
    if (chain < 0) {
      if ((int)chain_tail.size() < max_chains) {
        chain = (int)chain_tail.size();
        chain_tail.push_back(-1);
      } else {                      // out of streams: append to the chain of the first dependency (or chain 0)
        chain = deps[v].empty() ? 0 : R->nodes[pos[deps[v][0]]].chain;
      }
    }
    nd.chain = chain;
    chain_tail[chain] = (int)p;
  }
  R->nchains = (int)chain_tail.size();
  R->chain_size.assign(R->nchains, 0);
  for (auto& nd : R->nodes) R->chain_size[nd.chain]++;
  // cross-chain edges -> events; prune waits that an earlier wait of the same chain already covers
  std::vector<std::vector<int>> covered(R->nchains, std::vector<int>(R->nchains, -1));   // covered[X][Y] = latest position of Y that X waited for
  for (size_t p = 0; p < n; p++) {
    RNode& nd = R->nodes[p];
    std::vector<int> need(R->nchains, -1);
    for (int d : deps[order[p]]) {
      const int q = pos[d], c = R->nodes[q].chain;
      if (c != nd.chain) need[c] = std::max(need[c], q);
    }
    for (int c = 0; c < R->nchains; c++) {
      if (need[c] <= covered[nd.chain][c]) continue;
      covered[nd.chain][c] = need[c];
      RNode& src = R->nodes[need[c]];
      if (src.record < 0) {
        src.record = (int)R->events.size();
        R->events.push_back(nullptr);
        R->ncross++;
      }
      nd.waits.push_back(src.record);
      R->nwaits++;
    }
  }
  // Plain no-timing events.  Measured (round 3, 2 x 40 steps each): + hipEventReleaseToDevice 33.4-33.8 ms per step, default 33.25,
  // + hipEventDisableSystemFence 32.9-33.0 - a 1 % gain that is not worth an event whose release semantics are documented for timing only.
  const unsigned evflags = hipEventDisableTiming;
  for (auto& e : R->events) RP_HIP(hipEventCreateWithFlags(&e, evflags), "hipEventCreateWithFlags");
  R->tail.resize(R->nchains, nullptr);
  for (auto& e : R->tail) RP_HIP(hipEventCreateWithFlags(&e, evflags), "hipEventCreateWithFlags");
  RP_HIP(hipEventCreateWithFlags(&R->start, evflags), "hipEventCreateWithFlags");
  *replay_out = R;
  return CROG_OK;
}
}  // namespace

extern "C" int crog_replay_build(void* hip_graph, int max_chains, void** replay_out) {
  return build_replay(hip_graph, max_chains, nullptr, 0, replay_out);
}

extern "C" int crog_replay_build_tagged(void* hip_graph, int max_chains, void* const* nodes, const int* chains, int n_tags, void** replay_out) {
  CROG_CHECK_ARG(n_tags >= 0 && (n_tags == 0 || (nodes && chains)), "replay_build_tagged: tag arrays missing");
  std::unordered_map<hipGraphNode_t, int> tags;
  int top = 0;
  for (int i = 0; i < n_tags; i++) {
    CROG_CHECK_ARG(chains[i] >= 0 && chains[i] < max_chains, "replay_build_tagged: chain index out of range");
    tags[(hipGraphNode_t)nodes[i]] = chains[i];
    top = std::max(top, chains[i] + 1);
  }
  return build_replay(hip_graph, max_chains, &tags, top, replay_out);
}

extern "C" int crog_replay_info(void* replay, int* n_nodes, int* n_kernels, int* n_chains, int* n_events, int* n_waits, int* chain_sizes,
                                int chain_sizes_cap) {
  CROG_CHECK_ARG(replay != nullptr, "replay_info: null handle");
  auto* R = (Replay*)replay;
  if (n_nodes) *n_nodes = (int)R->nodes.size();
  if (n_kernels) *n_kernels = R->nkernels;
  if (n_chains) *n_chains = R->nchains;
  if (n_events) *n_events = R->ncross;
  if (n_waits) *n_waits = R->nwaits;
  for (int c = 0; chain_sizes && c < R->nchains && c < chain_sizes_cap; c++) chain_sizes[c] = R->chain_size[c];
  return CROG_OK;
}

extern "C" int crog_replay_launch(void* replay, const crog_stream_t* streams, int n_streams) {
  CROG_CHECK_ARG(replay && streams, "replay_launch: null argument");
  auto* R = (Replay*)replay;
  CROG_CHECK_ARG(n_streams >= R->nchains, "replay_launch: %d streams for %d chains", n_streams, R->nchains);
  hipStream_t s0 = (hipStream_t)streams[0];
  RP_HIP(hipEventRecord(R->start, s0), "hipEventRecord");
  for (int c = 1; c < R->nchains; c++) RP_HIP(hipStreamWaitEvent((hipStream_t)streams[c], R->start, 0), "hipStreamWaitEvent");
  for (const RNode& nd : R->nodes) {
    hipStream_t s = (hipStream_t)streams[nd.chain];
    for (int w : nd.waits) RP_HIP(hipStreamWaitEvent(s, R->events[w], 0), "hipStreamWaitEvent");
    const bool prof = R->profiling && nd.prof >= 0;
    if (prof) RP_HIP(hipEventRecord(R->prof_ev[2 * nd.prof], s), "hipEventRecord");
    // A kernel that other chains wait for carries its event as the launch's own stop event (hipExtLaunchKernel: the dispatch packet's
    // completion signal) instead of a marker packet behind it: a marker costs the producer's queue 6-12 us before its next kernel starts
    // (scripts/chain_gaps.py: ~90 such gaps on the main chain of a step)
    static const bool ext = [] { const char* e = getenv("CROG_REPLAY_EXT"); return !e || e[0] != '0'; }();
    if (ext && nd.record >= 0 && !prof && nd.type == hipGraphNodeTypeKernel && !nd.module_launch) {
      RP_HIP(hipExtLaunchKernel(nd.kp.func, nd.kp.gridDim, nd.kp.blockDim, nd.kp.kernelParams, nd.kp.sharedMemBytes, s, nullptr, R->events[nd.record], 0),
             "hipExtLaunchKernel");
      continue;
    }
    const int rc = issue(nd, s);
    if (rc != CROG_OK) return rc;
    if (prof) RP_HIP(hipEventRecord(R->prof_ev[2 * nd.prof + 1], s), "hipEventRecord");
    if (nd.record >= 0) RP_HIP(hipEventRecord(R->events[nd.record], s), "hipEventRecord");
  }
  for (int c = 1; c < R->nchains; c++) {
    RP_HIP(hipEventRecord(R->tail[c], (hipStream_t)streams[c]), "hipEventRecord");
    RP_HIP(hipStreamWaitEvent(s0, R->tail[c], 0), "hipStreamWaitEvent");
  }
  return CROG_OK;
}

extern "C" int crog_replay_profile_nodes(void* replay, void* const* nodes, int n) {
  CROG_CHECK_ARG(replay != nullptr && (n == 0 || nodes != nullptr), "replay_profile_nodes: null argument");
  auto* R = (Replay*)replay;
  for (auto& nd : R->nodes) nd.prof = -1;
  for (auto e : R->prof_ev) (void)hipEventDestroy(e);
  R->prof_ev.clear();
  std::unordered_map<hipGraphNode_t, int> where;
  for (size_t p = 0; p < R->handles.size(); p++) where[R->handles[p]] = (int)p;
  for (int i = 0; i < n; i++) {
    auto it = where.find((hipGraphNode_t)nodes[i]);
    CROG_CHECK_ARG(it != where.end(), "replay_profile_nodes: node %d is not part of the captured graph", i);
    R->nodes[it->second].prof = i;
  }
  R->prof_ev.resize(2 * (size_t)n, nullptr);
  for (auto& e : R->prof_ev) RP_HIP(hipEventCreateWithFlags(&e, hipEventDisableSystemFence), "hipEventCreateWithFlags");   // timing-only, no fence
  return CROG_OK;
}

extern "C" int crog_replay_profile_enable(void* replay, int on) {
  CROG_CHECK_ARG(replay != nullptr, "replay_profile_enable: null handle");
  ((Replay*)replay)->profiling = on != 0;
  return CROG_OK;
}

extern "C" int crog_replay_profile_read(void* replay, float* ms_out, int cap) {
  CROG_CHECK_ARG(replay && ms_out, "replay_profile_read: null argument");
  auto* R = (Replay*)replay;
  const int n = (int)(R->prof_ev.size() / 2);
  CROG_CHECK_ARG(cap >= n, "replay_profile_read: room for %d values needed", n);
  for (int i = 0; i < n; i++) {
    RP_HIP(hipEventSynchronize(R->prof_ev[2 * i + 1]), "hipEventSynchronize");
    RP_HIP(hipEventElapsedTime(&ms_out[i], R->prof_ev[2 * i], R->prof_ev[2 * i + 1]), "hipEventElapsedTime");
  }
  return CROG_OK;
}

extern "C" int crog_replay_destroy(void* replay) {
  if (!replay) return CROG_OK;
  auto* R = (Replay*)replay;
  for (auto e : R->events) if (e) (void)hipEventDestroy(e);
  for (auto e : R->tail) if (e) (void)hipEventDestroy(e);
  for (auto e : R->prof_ev) if (e) (void)hipEventDestroy(e);
  if (R->start) (void)hipEventDestroy(R->start);
  delete R;
  return CROG_OK;
}
