// internal.h -- host-side objects shared by build.hip, trace.hip and rtcore_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/embree_amd_hip.h"

namespace mi355 {

int set_error(hipError_t e, const char* what);          // records text for mi355_last_error(), returns (int)e
#define HIP_TRY(expr)                                                             \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) return ::mi355::set_error(_e, #expr);                   \
  } while (0)

struct TraceScratch {
  std::mutex* enqueue = nullptr; // a launch = cursor reset + kernel: the two must reach the stream back to back (several host threads may query one scene)
  uint32_t* counter = nullptr;   // ray cursor of the persistent kernel
  void* spill = nullptr;         // stack spill area
  uint64_t* stats = nullptr;     // 8 counters for the counting build
  void* pkt = nullptr; size_t pktCap = 0;    // AoS staging of the packet entry points (grows on demand)
  uint32_t* defer = nullptr; size_t deferCap = 0;   // RTC_RAY_QUERY_FLAG_COHERENT: [0] = number of packets the packet kernel gave up on, [64 ...] = their indices (grows on demand)
  volatile uint32_t* statusHost = nullptr;   // pinned host memory the kernels raise their "work was dropped" words in (iteration cap, stack overflow) ...
  uint32_t divergeStreak = 0;                // packet samples in a row (large coherent queries on this (tree, stream)) that said "the packets do not stay together"
  uint32_t coherentCalls = 0;                // large RTC_RAY_QUERY_FLAG_COHERENT queries that skipped the packet sample since the last one that took it
  volatile uint32_t* statusDev = nullptr;    // ... and its device address
  bool cursorsDirty = false;                 // a launch on this (tree, stream) left through the iteration cap: its ray cursors are zeroed before the next launch (trace.hip)
};

struct Bvh {
  int device = 0;
  int numCUs = 256;
  void* d_nodes = nullptr;       // CNode[numNodes]
  void* d_tris = nullptr;        // TriRec[numTris]
  size_t nodesCap = 0, trisCap = 0;   // capacities of the two arrays when they came from the build arena's spare list (0: plain hipMalloc)
  uint32_t root = 0xFFFFFFFFu;
  void* d_insts = nullptr;       // scenes with instances: InstRec[] (64 B: world2local | root node, instID, mask, flags); d_nodes / d_tris = top tree + the objects' trees
  void* d_rules = nullptr;       // device-side filter rules: 48 B per geometry id (instanced scenes: the own geometries', then every object's behind its base) + bit arrays
  std::vector<uint32_t> h_rules; uint32_t numRuleGeoms = 0;   // ... and the host copy of a FLAT tree's table (what an instanced build concatenates)
  bool robust = false;           // TriRec holds v0,v1,v2 (instead of v0,e1,e2); traversal = conservative node test + Pluecker
  mi355_bvh_info info{};
  // refit data (params.refit): leaf order (geometry table index, internal triangle) per TriRec, first node of every level, the mesh list it was built from
  void* d_ids = nullptr;
  std::vector<uint32_t> lvlStart;
  struct MeshSig { uint32_t geomID, numPrims, numVerts, quads; };
  std::vector<MeshSig> sig;
  // instanced trees: what mi355_bvh_refit_instanced needs when only transforms / masks changed -- the top tree (built with refit data) and its "mesh" of one box per
  // instance record (9 floats each), the records as they were built (root node, rule base: unchanged by a move), and which object each record names
  Bvh* top = nullptr; float* d_topVerts = nullptr; uint32_t* d_topIdx = nullptr; bool topHasOwn = false;
  std::vector<uint8_t> h_insts;                     // InstRec[] (64 B each)
  std::vector<const void*> instObjects;             // per record: the object tree it was built from (nullptr: the scene's own geometry)
  std::mutex mtx;
  std::map<hipStream_t, TraceScratch> scratch;
  TraceScratch* scratch_for(hipStream_t s);
  ~Bvh();
};

#define MI355_MAX_BLOCKS_PER_CU 8
size_t trace_spill_bytes(int numCUs, uint32_t depth);   // stack spill area of one persistent launch (trace.hip)

}  // namespace mi355
