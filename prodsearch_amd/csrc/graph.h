// graph.h — replay of an entry point's launch sequence as a HIP graph (gfx950).
//
// An entry point captures its own launch sequence once per distinct argument set (stream capture of the very code that
// otherwise runs eagerly — same kernels, same order, same results) and replays it afterwards.  What varies from call to
// call (Philox step, the caller's batch / loss tensors) never enters a captured kernel's arguments except through
// designated "patch" nodes, whose parameters are refreshed before every replay.
//
// What it buys, measured on MI355X / ROCm 7.2: the HOST cost of a training step drops from 347 to 285 us (two graph
// launches + two node patches instead of ~27 kernel launches), which is what limits a step fed by the native loader
// thread (0.592 -> 0.564 ms/step).  The DEVICE side gains nothing: dependent kernels of this library already issue
// back to back from C, and each graph launch plus the staging prologue cost ~8 us (pre-built batches: 0.505 ->
// 0.522 ms/step).  tools/graph_probe.py's 4.7 vs 1.9 us per kernel is a host-bound torch loop, not a device gap.
// Hence OPT-IN: PS_GRAPHS=1.  An argument set is captured the SECOND time it is seen (the first, eager, call also
// performs every lazy initialisation: function attributes, side streams).  Anything that fails to capture marks its
// entry bad and keeps running eagerly.
#pragma once
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PS_GRAPH_SLOTS 128
#define PS_GRAPH_PATCH 2

struct PsGraphEntry {
  uint64_t key;
  int state;                                // 0 empty, 1 seen once (eager), 2 captured, -1 bad
  hipGraph_t graph;
  hipGraphExec_t exec;
  hipGraphNode_t patch_node[PS_GRAPH_PATCH];
  hipKernelNodeParams patch_params[PS_GRAPH_PATCH];
  int n_patch;
};

inline bool ps_graphs_enabled() {
  static const bool on = ps_env_int("PS_GRAPHS", 0) != 0;
  return on;
}

inline uint64_t ps_fnv(uint64_t h, const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
#define PS_FNV0 1469598103934665603ull

inline PsGraphEntry* ps_graph_lookup(uint64_t key) {
  static PsGraphEntry table[PS_GRAPH_SLOTS];
  static bool init = false;
  if (!init) { memset(table, 0, sizeof(table)); init = true; }
  if (key == 0) key = 1;
  for (int probe = 0; probe < PS_GRAPH_SLOTS; ++probe) {
    PsGraphEntry& e = table[(key + probe) % PS_GRAPH_SLOTS];
    if (e.state == 0) { e.key = key; return &e; }
    if (e.key == key) return &e;
  }
  return nullptr;                           // table full: caller runs eagerly
}

// Capture happens on a private stream (the caller's is often the legacy default stream, which cannot capture); the
// instantiated graph is then launched on the caller's stream.  Returns the stream to issue the body on, or null.
inline hipStream_t ps_graph_begin() {
  static hipStream_t cap = nullptr;
  static bool tried = false;
  if (!tried) {
    tried = true;
    if (hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) != hipSuccess) { cap = nullptr; (void)hipGetLastError(); }
  }
  if (!cap) return nullptr;
  if (hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return cap;
}

// Ends the capture on `st`; on success the entry holds an instantiated graph.  `patch_funcs`: kernels (device function
// addresses) whose nodes the caller will re-parameterise before each replay, located here once.
inline bool ps_graph_debug() {
  static const bool on = ps_diag_int("PS_GRAPH_DEBUG", 0) != 0;
  return on;
}
inline int ps_graph_end(hipStream_t st, PsGraphEntry* e, const void* const* patch_funcs, int n_patch) {
  hipGraph_t g = nullptr;
  const hipError_t ec = hipStreamEndCapture(st, &g);
  if (ec != hipSuccess || !g) {
    if (ps_graph_debug()) fprintf(stderr, "[ps graph] end capture failed: %s\n", hipGetErrorString(ec));
    e->state = -1; (void)hipGetLastError(); return 1;
  }
  e->n_patch = 0;
  if (n_patch > 0) {
    size_t n = 0;
    if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) { e->state = -1; return 1; }
    hipGraphNode_t* nodes = (hipGraphNode_t*)malloc(sizeof(hipGraphNode_t) * (n ? n : 1));
    (void)hipGraphGetNodes(g, nodes, &n);
    for (int p = 0; p < n_patch; ++p) {
      bool found = false;
      for (size_t i = 0; i < n && !found; ++i) {
        hipGraphNodeType ty;
        if (hipGraphNodeGetType(nodes[i], &ty) != hipSuccess || ty != hipGraphNodeTypeKernel) continue;
        hipKernelNodeParams kp;
        if (hipGraphKernelNodeGetParams(nodes[i], &kp) != hipSuccess) continue;
        if (kp.func == patch_funcs[p]) {
          e->patch_node[p] = nodes[i];
          e->patch_params[p] = kp;
          found = true;
        }
      }
      if (!found) {
        if (ps_graph_debug()) fprintf(stderr, "[ps graph] patch kernel %d not found among %zu nodes\n", p, n);
        free(nodes); (void)hipGraphDestroy(g); e->state = -1; return 1;
      }
    }
    free(nodes);
    e->n_patch = n_patch;
  }
  hipGraphExec_t x = nullptr;
  if (hipGraphInstantiate(&x, g, nullptr, nullptr, 0) != hipSuccess || !x) {
    (void)hipGraphDestroy(g); (void)hipGetLastError(); e->state = -1; return 1;
  }
  e->graph = g; e->exec = x; e->state = 2;
  if (ps_graph_debug()) { size_t n = 0; (void)hipGraphGetNodes(g, nullptr, &n); fprintf(stderr, "[ps graph] captured %zu nodes, %d patch\n", n, n_patch); }
  return PS_OK;
}

// Re-parameterise patch node p with one by-value argument blob per kernel parameter (kernel_params[i] -> argument i).
inline int ps_graph_patch(PsGraphEntry* e, int p, void** kernel_params) {
  hipKernelNodeParams kp = e->patch_params[p];
  kp.kernelParams = kernel_params;
  kp.extra = nullptr;
  return hipGraphExecKernelNodeSetParams(e->exec, e->patch_node[p], &kp) == hipSuccess ? PS_OK : 1;
}

inline int ps_graph_launch(PsGraphEntry* e, hipStream_t st) {
  return hipGraphLaunch(e->exec, st) == hipSuccess ? PS_OK : 1;
}
