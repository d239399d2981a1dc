// build.hip -- GPU construction of the 8-wide quantised BVH (scene commit) for gfx950.
//
// Replaces BVHNBuilderSAH<8,Triangle4>::build (kernels/bvh/bvh_builder_sah.cpp:112-193) and what it
// calls: createPrimRefArray (kernels/builders/primrefgen.cpp:35-57, TriangleMesh::buildBounds
// kernels/common/scene_triangle_mesh.h:195-215), the binned-SAH heuristic (kernels/builders/
// heuristic_binning.h:16-111 BinMapping, :210-257 bin, :339-386 best; heuristic_binning_array_aligned.h
// :141-176 split, :50-65 fallback), BuilderT::recurse (kernels/builders/bvh_builder_sah.h:214-308)
// and CreateLeaf / TriangleM::fill (bvh_builder_sah.cpp:32-55, kernels/geometry/triangle.h:98-120).
//
// The reference recurses depth-first on host threads.  Here the same decisions are taken by five
// data-parallel stages, all on the GPU:
//   K1 primref_gen    1 thread / triangle: validity test, AABB, compaction, scene + centroid bounds
//   K2 top phase      level-synchronous binary binned-SAH splits of every segment > small_threshold:
//                     setup (bin mapping, chunk table) -> bin (LDS-staged histograms of 2048-triangle
//                     chunks, merged with ordered-uint atomics) -> split (one wavefront per segment
//                     evaluates all 3x31 candidates) -> partition (block-aggregated scatter into the
//                     ping-pong buffer + child centroid bounds) -> emit (children -> next level / small list)
//   K3 small phase    one wavefront finishes each sub-tree of <= small_threshold triangles on its own:
//                     bins in LDS, explicit stack (larger child pushed), splits down to min_leaf
//   K4 wide collapse  top-down, one thread per 8-wide node: the reference's greedy "split the child
//                     with the largest half-area until 8 children" + leaf-vs-split SAH test evaluated
//                     on the binary tree; children placed in the slot matching their octant, inner
//                     children / leaf triangles numbered consecutively, bounds quantised to 8 bits
//   K5 tri_records    TriRec array in node order (v0, e1, e2, ids, mask)
// Bin bounds/counts are combined with integer min/max/add, so the tree TOPOLOGY does not depend on
// thread timing; leaves are sorted by (primID, geomID) like heuristic.deterministic_order
// (heuristic_binning_array_aligned.h:178-182).  The tree need not equal the reference's tree:
// t/u/v/Ng/IDs of a closest hit do not depend on tree shape.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <vector>
#include "bvh_common.h"
#include "internal.h"

namespace {
// The kernels live in build_*.inl, all part of this one translation unit (they share the anonymous namespace and are listed in pipeline order):
#include "build_common.inl"      // constants, build-time structs, ordered-uint float codes, wave64 DPP reductions
#include "build_primref.inl"     // K1: PrimRef generation and the stable compaction of invalid triangles
#include "build_presplit.inl"    // RTC_BUILD_QUALITY_HIGH: pre-splitting of large triangles
#include "build_binning.inl"     // bin mapping, binning helpers (rows / runs), SAH sweep of one wavefront
#include "build_top.inl"         // K2: level-synchronous top phase
#include "build_spatial.inl"     // RTC_BUILD_QUALITY_HIGH: spatial splits inside the top phase
#include "build_small.inl"       // K3: sub-trees finished by one wavefront in LDS
#include "build_morton.inl"      // RTC_BUILD_QUALITY_LOW: Morton-code build
#include "build_sort.inl"        // ... its radix sort (onesweep, 7 passes of 9 bits) and the Morton codes
#include "build_wide.inl"        // K4: collapse of the binary tree into 8-wide quantised nodes
#include "build_leaves.inl"      // K5: leaf records; refit; node rebasing for instanced scenes

// Build scratch comes from a per-device arena that survives the commit: rtcCommitScene is timed on the wall clock
// (tutorials/buildbench/buildbench_device.cpp:385-387) and ~20 hipMalloc/hipFree pairs cost 3 ms of a 10.7 ms commit of 4.76 M
// triangles.  Blocks are kept until mi355_release_build_scratch(); commits on one device are serialised by the arena's mutex
// (the reference's rtcCommitScene is blocking as well and runs on its own worker pool).
struct Arena {
  struct Block { char* p; size_t cap, used; };
  std::mutex mtx;
  std::vector<Block> blocks;
  // The launch sequence of a default-quality commit, captured once as a HIP graph and replayed while its kernel arguments stay what they were (same
  // triangle count, same parameters, same arena addresses): a dynamic scene that is re-committed every frame pays ONE host call for its ~200 launches.
  hipStream_t stream = nullptr;                              // commits enqueue here when the caller gives no stream (a capture needs a real stream)
  std::vector<uint64_t> graphKey; hipGraphExec_t graphExec = nullptr; bool graphBroken = false;   // graphBroken: capture / instantiate failed once on this device: plain launches from then on
  uint32_t marginFailedN = 0;                                // a commit of this many triangles outgrew the level margins of the one-round-trip path: the next one goes stepwise at once
  // What the last one-round-trip commit of a scene of this many triangles needed: its top-phase levels and the depth of its wide tree.  The next commit of the
  // same size (a scene re-committed every frame) enqueues those + 1 instead of the blind margins (levels N implies + 8, 16 wide levels): every level that does
  // not exist still costs its launches (~4.7 us each, ~35 of the 169 of a crown commit).  A commit that outgrows the learned counts runs again with the blind margins.
  // (round 5) The counts are kept PER KIND of commit -- triangle count, geometry count, build parameters (quality, spatial, leaf sizes, outlier cut ...) -- not per
  // triangle count alone: bench.py commits its scene MEDIUM, then HIGH, then MEDIUM again, and every other commit found the counts of the other quality, failed and ran
  // twice.  Within a kind the counts only GROW (maximum of what the commits of that kind needed; first top_local level: minimum): two scenes of one kind that need different
  // depths, committed in turn, cost the shallower one a few empty launches instead of costing the deeper one a second commit every time.
  struct Learned { uint64_t kind = 0; uint32_t top = 0, wide = 0, chunked = 0, localFirst = 255u, nodes = 0; uint64_t used = 0; };   // nodes: wide nodes of the last commit of the kind (sizes the tree's node buffer up front)
  Learned learned[8]; uint64_t learnedClock = 0;               // (eight kinds per device, the least recently used one is replaced)
  Learned* find_learned(uint64_t kind) { for (auto& l : learned) if (l.kind == kind && l.top != 0u) { l.used = ++learnedClock; return &l; } return nullptr; }
  void learn(uint64_t kind, uint32_t top, uint32_t wide, uint32_t chunked, uint32_t localFirst, uint32_t nodes) {
    Learned* l = find_learned(kind);
    if (!l) { l = &learned[0]; for (auto& c : learned) if (c.used < l->used) l = &c; *l = Learned(); l->kind = kind; }
    l->top = top > l->top ? top : l->top; l->wide = wide > l->wide ? wide : l->wide; l->chunked = chunked > l->chunked ? chunked : l->chunked;
    l->localFirst = localFirst < l->localFirst ? localFirst : l->localFirst; l->nodes = nodes; l->used = ++learnedClock;
  }
  void drop_graph() { if (graphExec) { hipGraphExecDestroy(graphExec); graphExec = nullptr; } graphKey.clear(); }
  void reset() { for (auto& b : blocks) b.used = 0; }
  hipError_t take(size_t bytes, void** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    for (auto& b : blocks) if (b.cap - b.used >= bytes) { *out = b.p + b.used; b.used += bytes; return hipSuccess; }
    Block nb; nb.cap = bytes > ((size_t)64 << 20) ? bytes : ((size_t)64 << 20); nb.used = bytes; nb.p = nullptr;
    const hipError_t e = hipMalloc((void**)&nb.p, nb.cap);
    if (e != hipSuccess) return e;
    blocks.push_back(nb); *out = nb.p; return hipSuccess;
  }
  // Tree buffers (CNode / TriRec arrays) of destroyed trees are kept for the next commit of a similar size: a scene that is re-committed every frame would
  // otherwise pay a hipFree + hipMalloc of ~300 MB per commit, and every second or third of those takes the driver 8 ms (commit 7.6 -> 16 ms, measured).
  struct Spare { void* p; size_t cap; };
  std::vector<Spare> spares;                                 // at most 4
  void* take_output(size_t bytes, size_t* cap) {
    for (size_t i = 0; i < spares.size(); i++)
      if (spares[i].cap >= bytes && spares[i].cap <= bytes + bytes / 2 + 4096) { void* p = spares[i].p; *cap = spares[i].cap; spares.erase(spares.begin() + i); return p; }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); for (auto& sp : spares) hipFree(sp.p); spares.clear(); if (hipMalloc(&p, bytes) != hipSuccess) return nullptr; }
    *cap = bytes; return p;
  }
  void give_output(void* p, size_t cap) {
    // at most four arrays and 1 GiB in all (the API has told the memory monitor they are freed: what sits here is memory the application believes it has)
    spares.push_back({p, cap});
    size_t total = 0; for (auto& sp : spares) total += sp.cap;
    while (!spares.empty() && (spares.size() > 4 || total > ((size_t)1 << 30))) { total -= spares.front().cap; hipFree(spares.front().p); spares.erase(spares.begin()); }
  }
  void release() { drop_graph(); if (stream) { hipStreamDestroy(stream); stream = nullptr; } for (auto& b : blocks) hipFree(b.p); blocks.clear(); }
  void release_spares() { for (auto& sp : spares) hipFree(sp.p); spares.clear(); }
};
static std::mutex g_arenaMtx;
static std::map<int, Arena*> g_arenas;
static Arena* arena_of(int device) {
  std::lock_guard<std::mutex> lk(g_arenaMtx);
  Arena*& a = g_arenas[device];
  if (!a) a = new Arena;
  return a;
}
static thread_local Arena* t_arena = nullptr;
// tree buffers go through the arena's spare list (its own lock: trees are destroyed from any thread, also while a commit holds the arena)
static std::mutex g_spareMtx;
static void* output_alloc(int device, size_t bytes, size_t* cap) {
  std::lock_guard<std::mutex> lk(g_spareMtx);
  void* p = arena_of(device)->take_output(bytes, cap);
  if (getenv("MI355_BUILD_DEBUG")) fprintf(stderr, "[mi355 build] output_alloc %zu -> %p cap %zu\n", bytes, p, *cap);
  return p;
}
static void output_free(int device, void* p, size_t cap) {
  if (!p) return;
  hipDeviceSynchronize();                                    // (what hipFree would have waited for: nothing may still read the tree)
  std::lock_guard<std::mutex> lk(g_spareMtx);
  if (getenv("MI355_BUILD_DEBUG")) fprintf(stderr, "[mi355 build] output_free %p cap %zu\n", p, cap);
  arena_of(device)->give_output(p, cap);
}

template <typename T> struct DevBuf {
  T* p = nullptr;
  hipError_t alloc(size_t n) { return t_arena->take((n ? n : 1) * sizeof(T), (void**)&p); }
};

}  // namespace

namespace mi355 {

static thread_local std::string g_err;
int set_error(hipError_t e, const char* what) {
  g_err = std::string(what ? what : "") + ": " + hipGetErrorString(e);
  return e == hipSuccess ? 1 : (int)e;
}

TraceScratch* Bvh::scratch_for(hipStream_t s) {
  std::lock_guard<std::mutex> lk(mtx);
  auto it = scratch.find(s);
  if (it != scratch.end()) return &it->second;
  TraceScratch sc;
  if (hipMalloc((void**)&sc.counter, 4096) != hipSuccess) return nullptr;
  // the ray cursors start at zero and every launch leaves them at zero (trace.hip, wave_exit).  Zeroed ON THE STREAM the scratch belongs to: a hipMemset goes to
  // the null stream, which a non-blocking stream does not wait for -- the first launch could start on recycled memory, or be zeroed half way (rays skipped in the
  // launch after it: found as an intermittent idempotence failure at full size)
  if (hipMemsetAsync(sc.counter, 0, 4096, s) != hipSuccess) return nullptr;
  if (hipMalloc(&sc.spill, trace_spill_bytes(numCUs, info.depth)) != hipSuccess) return nullptr;
  if (hipMalloc((void**)&sc.stats, 256) != hipSuccess) return nullptr;
  { void* h = nullptr; void* d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) return nullptr;
    memset(h, 0, 64);
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) return nullptr;
    sc.statusHost = (volatile uint32_t*)h; sc.statusDev = (volatile uint32_t*)d; }
  sc.enqueue = new std::mutex;
  return &(scratch[s] = sc);
}
Bvh::~Bvh() {
  hipSetDevice(device);
  for (auto& kv : scratch) { hipFree(kv.second.counter); hipFree(kv.second.spill); hipFree(kv.second.stats); if (kv.second.pkt) hipFree(kv.second.pkt); if (kv.second.defer) hipFree(kv.second.defer); hipHostFree((void*)kv.second.statusHost); delete kv.second.enqueue; }
  if (d_nodes) { if (nodesCap) output_free(device, d_nodes, nodesCap); else hipFree(d_nodes); }
  if (d_tris) { if (trisCap) output_free(device, d_tris, trisCap); else hipFree(d_tris); }
  if (d_ids) hipFree(d_ids);
  if (d_insts) hipFree(d_insts);
  if (d_rules) hipFree(d_rules);
  if (d_topVerts) hipFree(d_topVerts);
  if (d_topIdx) hipFree(d_topIdx);
  delete top;
}

// A commit is enqueued as ONE sequence of launches with no host round trip inside (MEDIUM quality, the default): the number of valid triangles, the
// scene bounds, the lengths of the work lists all stay on the device (build_begin, root_setup, guarded compaction, grids that are upper bounds), the level
// loops are enqueued with a margin beyond the levels N implies, and the host waits ONCE, at the end, where it also learns whether the margins were
// enough (top phase finished, wide levels finished, no overflow); if not -- pathological input -- the commit is repeated on the stepwise path, which
// looks at the counters between groups of levels like the first generations of this builder did.  Why: a host round trip costs 20-40 us on an idle
// box but was measured at ~0.8 ms each on the round-end driver's box (commit 13.9 ms there, 7.0 ms here, same code), and there were nine of them.
// LOW (Morton: the sort needs n on the host) and HIGH (presplit: the budget loop) keep one round trip after primref_gen.
static int build_impl(int device, const mi355_mesh* meshes, uint32_t numMeshes, const mi355_build_params* bp, hipStream_t st, Bvh** out, bool allowFast = true, bool allowTopSplits = true, bool allowLearned = true) {
  HIP_TRY(hipSetDevice(device));
  Arena* arena = arena_of(device);
  std::lock_guard<std::mutex> arenaLock(arena->mtx);
  arena->reset(); t_arena = arena;
  static std::mutex propMtx; static std::map<int, int> cuCount;
  int numCUs;
  { std::lock_guard<std::mutex> lk(propMtx); auto it = cuCount.find(device);
    if (it == cuCount.end()) { hipDeviceProp_t prop; HIP_TRY(hipGetDeviceProperties(&prop, device)); it = cuCount.emplace(device, prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256).first; }
    numCUs = it->second; }
  Bvh* bvh = new Bvh; bvh->device = device; bvh->numCUs = numCUs;
  struct Guard { Bvh*& b; bool ok = false; ~Guard() { if (!ok) { delete b; b = nullptr; } } } guard{bvh};

  Params prm; prm.shift = bp->sah_block_shift; prm.minLeaf = bp->min_leaf ? bp->min_leaf : 1u;
  prm.maxLeaf = bp->max_leaf > MI355_MAX_LEAF ? MI355_MAX_LEAF : bp->max_leaf; if (prm.maxLeaf < prm.minLeaf) prm.maxLeaf = prm.minLeaf;
  if (prm.minLeaf > MI355_MAX_LEAF) prm.minLeaf = prm.maxLeaf = MI355_MAX_LEAF;
  prm.small = bp->small_threshold < 64u ? 64u : (bp->small_threshold > 65536u ? 65536u : bp->small_threshold); prm.travCost = bp->trav_cost; prm.intCost = bp->int_cost; prm.quality = bp->quality;
  // RTC_BUILD_QUALITY_HIGH: spatial splits inside the recursion (the reference's default, bvh_builder_sah_spatial.cpp:93-160); "presplits=1" selects the
  // reference's other form, splitting big triangles up front (state.cpp:88 useSpatialPreSplits).  The spatial splits live in the top phase: lower its end.
  prm.spatial = (bp->quality == 2u && !bp->presplits && numMeshes < (1u << 27)) ? 1u : 0u;
  if (prm.spatial && prm.small > 256u) prm.small = 256u;
  // RTC_BUILD_QUALITY_MEDIUM (the default): the few references that dwarf all others are cut into grid pieces before the build (build_presplit.inl,
  // outlier_*; "top_splits=0" switches it off, and so does refit data: a refit walks one leaf record per triangle).  The reference's MEDIUM builder never
  // cuts a triangle (BVHBuilderBinnedSAH, kernels/builders/bvh_builder_sah.h:446); answers do not depend on it.
  const bool topSplits = allowTopSplits && bp->quality == 0u && bp->top_splits != 0u && !bp->refit && numMeshes < (1u << 27);
  const float topSplitRel = bp->top_split_rel > 0.0f ? bp->top_split_rel : 32.0f;   // an outlier: box area >= this many times the mean box area ...
  const float topSplitCell = bp->top_split_cell > 0.0f ? bp->top_split_cell : 1.0f / 8.0f;   // ... and longer than this fraction of the scene's largest extent (= the grid its pieces are cut on)

  std::vector<GeomDesc> gd; uint64_t total = 0;
  for (uint32_t i = 0; i < numMeshes; i++) {
    const mi355_mesh& m = meshes[i];
    if (m.num_triangles == 0) continue;
    // (no "stride >= element size": the reference accepts views whose elements overlap -- BufferStrideTest, tutorials/verify/verify.cpp:995-1008, binds UINT4 quad
    // indices with a 12-byte stride -- and every kernel here reads element i word by word at offset + i * stride: prim_indices, primref_gen, tri_records, refit_level)
    if ((m.vertex_stride & 3) || (m.index_stride & 3)) return set_error(hipErrorInvalidValue, "buffer stride");
    GeomDesc g{}; g.verts = (const char*)m.d_vertices; g.idx = (const char*)m.d_indices; g.vstride = (uint32_t)m.vertex_stride; g.istride = (uint32_t)m.index_stride;
    g.nv = m.num_vertices; g.quad = m.quads ? 1u : 0u; g.nt = m.num_triangles * (g.quad ? 2u : 1u); g.geomID = m.geom_id; g.mask = m.mask; g.primOffset = (uint32_t)total;
    total += g.nt; gd.push_back(g);
  }
  if (total >= (1ull << 31)) return set_error(hipErrorInvalidValue, "more than 2^31 triangles are not supported by the 32-bit triangle index");
  mi355_bvh_info& info = bvh->info; memset(&info, 0, sizeof(info));
  info.max_leaf = prm.maxLeaf; info.root_ref = MI355_EMPTY_REF;
  for (int d = 0; d < 3; d++) { info.bounds_lower[d] = INFINITY; info.bounds_upper[d] = -INFINITY; }
  if (total == 0) { guard.ok = true; *out = bvh; return 0; }
  const uint32_t N = (uint32_t)total;
  static const bool envSync = getenv("MI355_BUILD_STEPWISE") != nullptr;         // A/B: force the stepwise path
  // (HIGH with spatial splits takes the one-round-trip path as well -- round 4: its nine host round trips were ~0.4 ms of idle GPU; HIGH with presplits keeps its budget loop on the host)
  const bool fast = allowFast && !envSync && (prm.quality == 0u || prm.spatial != 0u) && !(arena->marginFailedN != 0u && arena->marginFailedN == N);
  static const bool envGraph = !(getenv("MI355_BUILD_GRAPH") && atoi(getenv("MI355_BUILD_GRAPH")) == 0);
  if (fast && envGraph && !st && !arena->graphBroken) {        // the graph needs a stream of its own
    // a BLOCKING stream (default flags): it keeps the implicit ordering with the legacy null stream that a commit on the null stream had -- work the application
    // queued there to fill device-resident shared buffers (rtcSetSharedGeometryBufferHostDevice) is finished before the build reads them
    if (!arena->stream && hipStreamCreateWithFlags(&arena->stream, hipStreamDefault) != hipSuccess) { (void)hipGetLastError(); arena->stream = nullptr; }
    st = arena->stream;
  }
  const bool useGraph = fast && envGraph && st != nullptr && !arena->graphBroken;
  static const bool envLearn = !(getenv("MI355_BUILD_LEARN") && atoi(getenv("MI355_BUILD_LEARN")) == 0);
  // the kind of this commit (see Arena::Learned): FNV-1a over everything that shapes the launch sequence
  uint64_t kind = 1469598103934665603ull;
  { auto mix = [&](uint64_t v) { for (int i = 0; i < 8; i++) { kind ^= (v >> (8 * i)) & 0xFFu; kind *= 1099511628211ull; } };
    uint32_t pw[sizeof(Params) / 4]; memcpy(pw, &prm, sizeof(prm)); for (uint32_t w : pw) mix(w);
    mix(N); mix(numMeshes); mix(bp->robust); mix(bp->presplits); mix(bp->refit); mix(topSplits ? 1u : 0u); }
  const Arena::Learned* const lc = (fast && allowLearned && envLearn) ? arena->find_learned(kind) : nullptr;
  const bool learned = lc != nullptr;
  const uint32_t learnedTop = lc ? lc->top : 0u, learnedWide = lc ? lc->wide : 0u, learnedChunked = lc ? lc->chunked : 0u, learnedLocalFirst = lc ? lc->localFirst : 0u;
  uint32_t launches = 0, syncs = 0;
  uint32_t preCap = 0;                                          // nodes the tree's own node buffer was sized for before the commit knew its node count (tri_records copies them), 0 = not
  bool replay = false, capturing = false;                       // fast path: the launches below are replayed from the cached graph / are being captured into one
  // MI355_BUILD_DEBUG=1: every launch is named on stderr and waited for (finds the kernel behind a device fault; use with MI355_BUILD_GRAPH=0)
  static const bool envDebug = getenv("MI355_BUILD_DEBUG") != nullptr;
#define LAUNCH(...) do { if (!replay) { hipLaunchKernelGGL(__VA_ARGS__); if (envDebug && !capturing) { fprintf(stderr, "[mi355 build] %s\n", #__VA_ARGS__); const hipError_t de_ = hipStreamSynchronize(st); if (de_ != hipSuccess) fprintf(stderr, "[mi355 build]   -> %s\n", hipGetErrorString(de_)); } } launches++; } while (0)
#define SYNC_READ(h) do { HIP_TRY(hipGetLastError()); HIP_TRY(hipMemcpyAsync(&(h), ctr.p, sizeof(h), hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st)); syncs++; } while (0)

  DevBuf<GeomDesc> dGeoms; HIP_TRY(dGeoms.alloc(gd.size()));
  HIP_TRY(hipMemcpyAsync(dGeoms.p, gd.data(), gd.size() * sizeof(GeomDesc), hipMemcpyHostToDevice, st));
  const bool spatial = prm.spatial != 0u;
  // sets of fewer references than this split by object only (the reference tries a spatial split wherever a set has an extended range): measured, the SAH
  // of the long-triangle scene is 12.53 instead of 12.52 and the crown stand-in's tree does not change, for 1.2 ms less (spatial_bin clips a small set's
  // references through most of its 16 bins per axis); 2048 would cost 2 % of the SAH gain, 65536 all of it on small scenes (profiles/r02_sah_vs_reference.md)
  static const uint32_t spatialMinHigh = getenv("MI355_SPATIAL_MIN") ? (uint32_t)atol(getenv("MI355_SPATIAL_MIN")) : 512u;
  const uint32_t spatialMin = spatialMinHigh;
  const bool presplit = prm.quality == 2u && !spatial;         // up to 20 % more references than triangles, either way
  const uint32_t splitBudget = prm.quality == 2u ? (uint32_t)((double)N * (bp->split_factor > 1.0f ? (double)bp->split_factor - 1.0 : 0.2))
                             : (topSplits ? N / 16u + 65536u : 0u);   // MEDIUM: the reserve for the pieces of outlier references
  const uint64_t cap64 = (uint64_t)N + splitBudget;
  if (cap64 >= (1ull << 31)) return set_error(hipErrorInvalidValue, "more than 2^31 references are not supported by the 32-bit triangle index");
  const uint32_t NC = (uint32_t)cap64;                        // capacity of every per-reference array
  const uint32_t maxSegs = NC / prm.small + 1024u, maxSmall = 8u * (NC / prm.small) + 1024u, maxChunks = NC / CHUNK + maxSegs + 16u;
  const uint32_t maxWide = NC + 64u;
  DevBuf<PrimRef> bufA, bufB; DevBuf<uint2> finalIds; DevBuf<BNode> bnodes; DevBuf<Seg> segs0, segs1; DevBuf<uint32_t> bins;
  DevBuf<Chunk> chunks; DevBuf<SmallEntry> small; DevBuf<Counters> ctr; DevBuf<WideItem> w0, w1; DevBuf<CNode> wnodes; DevBuf<uint2> outIds;
  HIP_TRY(bufA.alloc(NC)); HIP_TRY(bufB.alloc(NC)); HIP_TRY(finalIds.alloc(NC)); HIP_TRY(bnodes.alloc(2ull * NC + 2));
  HIP_TRY(segs0.alloc(maxSegs)); HIP_TRY(segs1.alloc(maxSegs)); HIP_TRY(bins.alloc((size_t)maxSegs * BINS_WORDS));
  HIP_TRY(chunks.alloc(maxChunks)); HIP_TRY(small.alloc((size_t)maxSmall + maxSmall / 10u + 8u)); HIP_TRY(ctr.alloc(1));   // (+ 4 bytes per entry behind the list: their sizes, side by side)
  DevBuf<uint32_t> smallOrder; HIP_TRY(smallOrder.alloc(maxSmall));   // the list's indices, largest sub-tree first (small_order)
  HIP_TRY(w0.alloc(maxWide)); HIP_TRY(w1.alloc(maxWide)); HIP_TRY(wnodes.alloc(maxWide)); HIP_TRY(outIds.alloc(NC));
  const uint32_t maxLevelItems = NC / 2u + 64u;                 // the nodes of one level are disjoint sub-trees of >= 2 triangles each
  DevBuf<WidePlan> plans; DevBuf<uint2> itemCnt, groupSum; DevBuf<uint32_t> tileCount;
  HIP_TRY(plans.alloc(maxLevelItems)); HIP_TRY(itemCnt.alloc(maxLevelItems)); HIP_TRY(groupSum.alloc(maxLevelItems / 8u + 16u));
  const uint32_t tiles = (N + 255u) / 256u;
  HIP_TRY(tileCount.alloc((NC + 255u) / 256u));
  const uint32_t genBlocks = (N + 1023u) / 1024u < 4096u ? (N + 1023u) / 1024u : 4096u;   // primref_gen: 1024 triangles per workgroup and step
  DevBuf<AreaPart> areaPart; HIP_TRY(areaPart.alloc(genBlocks));                             // ... and what every workgroup leaves for outlier_stats
  DevBuf<uint32_t> outlierCnt, outlierTile, outlierTotal;      // MEDIUM: grid cells every reference asks for (0 unless it is an outlier), their tile sums / offsets, the total
  DevBuf<OutlierWork> outlierWork;                            // ... the outliers themselves (every one asks for >= 2 cells: at most half the reserve)
  if (topSplits) { HIP_TRY(outlierCnt.alloc(N)); HIP_TRY(outlierTile.alloc(tiles)); HIP_TRY(outlierTotal.alloc(1)); HIP_TRY(outlierWork.alloc((NC - N) / 2u + 16u)); }
  DevBuf<uint32_t> accTop, binsTop; HIP_TRY(accTop.alloc((size_t)ACC_SETS * ACC_REPL * ACC_STRIDE)); HIP_TRY(binsTop.alloc((size_t)ACC_SETS * ACC_REPL * BINS_WORDS));   // copies of the upper levels' sets' bounds records (build_top.inl, acc_copy)
  DevBuf<uint32_t> chunkCnt; DevBuf<uint2> chunkBase;          // per chunk of a level: its bin counts, then its places in the two children (top_bin -> top_split -> top_partition)
  HIP_TRY(chunkCnt.alloc((size_t)maxChunks * 3u * NBINS)); HIP_TRY(chunkBase.alloc(maxChunks));
  DevBuf<SegX> segx0, segx1; DevBuf<uint32_t> sbins, sbinsTop;            // spatial-split builds: extended ranges of the top phase's sets, their spatial bins
  DevBuf<uint32_t> chunkFlag;                                  // ... per chunk: what it sends to either side of a spatial split (spatial_partition)
  if (spatial) { HIP_TRY(segx0.alloc(maxSegs)); HIP_TRY(segx1.alloc(maxSegs)); HIP_TRY(sbins.alloc((size_t)maxSegs * SBINS_WORDS)); HIP_TRY(sbinsTop.alloc((size_t)ACC_SETS * ACC_REPL * SBINS_WORDS)); HIP_TRY(chunkFlag.alloc(maxChunks)); }

  hipEvent_t ev0, ev1; HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1));
  struct EvGuard { hipEvent_t a, b; ~EvGuard() { hipEventDestroy(a); hipEventDestroy(b); } } evg{ev0, ev1};
  if (useGraph) {
    HIP_TRY(hipStreamSynchronize(st));                         // (the geometry table above is in place before anything is captured)
    const void* ptrs[] = {dGeoms.p, bufA.p, bufB.p, finalIds.p, bnodes.p, segs0.p, segs1.p, bins.p, chunks.p, small.p, ctr.p, w0.p, w1.p, wnodes.p, outIds.p, plans.p, itemCnt.p, groupSum.p, tileCount.p, chunkCnt.p, chunkBase.p, accTop.p, binsTop.p, segx0.p, segx1.p, sbins.p, sbinsTop.p, outlierCnt.p, outlierTile.p, outlierTotal.p, outlierWork.p, areaPart.p, smallOrder.p, (const void*)st};
    std::vector<uint64_t> key; for (const void* q : ptrs) key.push_back((uint64_t)(uintptr_t)q);
    uint32_t pw[sizeof(Params) / 4]; memcpy(pw, &prm, sizeof(prm)); for (uint32_t w : pw) key.push_back(w);
    key.push_back(N); key.push_back(gd.size()); key.push_back(bp->robust); key.push_back(spatialMin); key.push_back(NC); { uint32_t w; memcpy(&w, &topSplitRel, 4); key.push_back(w); memcpy(&w, &topSplitCell, 4); key.push_back(w); } key.push_back(topSplits ? 1u : 0u); key.push_back(learned ? (learnedTop << 8) | learnedWide : 0u); key.push_back(learned ? (learnedChunked << 8) | (learnedLocalFirst & 0xFFu) : 0u);
    if (arena->graphExec && arena->graphKey == key) replay = true;
    else {
      arena->drop_graph(); arena->graphKey = key;
      if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) capturing = true; else (void)hipGetLastError();
    }
  }
  struct CapGuard { hipStream_t s; bool& on; ~CapGuard() { if (on) { hipGraph_t g = nullptr; hipStreamEndCapture(s, &g); if (g) hipGraphDestroy(g); (void)hipGetLastError(); } } } capGuard{st, capturing};
  if (!capturing && !replay) HIP_TRY(hipEventRecord(ev0, st));

  Counters h{};
  LAUNCH(build_begin, dim3(1), dim3(256), 0, st, ctr.p);
  LAUNCH(primref_gen, dim3(genBlocks), dim3(256), 0, st, dGeoms.p, (uint32_t)gd.size(), N, bufA.p, ctr.p, areaPart.p);
  uint32_t n = N;                                              // fast path: an upper bound of the valid triangles (the device knows the number)
  auto decf = [](uint32_t u) { uint32_t v = u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu); float f; memcpy(&f, &v, 4); return f; };
  float glo[3], ghi[3], clo[3], chi[3];
  uint32_t numSegs = 0, numSmall = 0;
  if (fast) {
    const uint32_t ctiles = (NC + 255u) / 256u;                  // tiles of the compaction: the references and, behind them, the reserve for outlier pieces
    if (topSplits) {                                             // references that dwarf all others are cut into grid pieces (build_presplit.inl): holes and originals become "invalid"
      LAUNCH(outlier_stats, dim3(1), dim3(1024), 0, st, (const AreaPart*)areaPart.p, genBlocks, ctr.p);
      LAUNCH(outlier_mark, dim3(tiles), dim3(256), 0, st, bufA.p, N, ctr.p, topSplitRel, topSplitCell, outlierCnt.p, outlierTile.p, tileCount.p);
      LAUNCH(presplit_scan, dim3(1), dim3(1024), 0, st, outlierTile.p, tiles, outlierTotal.p);
      LAUNCH(outlier_emit, dim3(tiles), dim3(256), 0, st, bufA.p, N, NC - N, outlierCnt.p, outlierTile.p, outlierTotal.p, ctr.p, outlierWork.p, tileCount.p);
      LAUNCH(outlier_clip, dim3(1024), dim3(256), 0, st, bufA.p, NC - N, dGeoms.p, outlierTotal.p, ctr.p, outlierWork.p, topSplitRel, topSplitCell);
      LAUNCH(outlier_retire, dim3(16), dim3(256), 0, st, bufA.p, NC - N, outlierTotal.p, (const Counters*)ctr.p, outlierWork.p);
    }
    // invalid triangles (and the holes of the outlier grid) are squeezed out on the device, if there are any; then the root
    // (with the outlier cut the tiles in front of the one N falls into are counted already: outlier_mark / outlier_emit)
    const uint32_t ctile0 = topSplits ? N / 256u : 0u;
    LAUNCH(compact_count, dim3(ctiles - ctile0), dim3(256), 0, st, bufA.p, NC, tileCount.p, ctr.p, N, ctile0);
    LAUNCH(compact_scan, dim3(1), dim3(1024), 0, st, tileCount.p, ctiles, ctr.p, 1u);
    LAUNCH(compact_scatter, dim3(ctiles), dim3(256), 0, st, bufA.p, NC, tileCount.p, bufB.p, ctr.p, N);
    LAUNCH(compact_copyback, dim3(ctiles < 2048u ? ctiles : 2048u), dim3(256), 0, st, bufB.p, bufA.p, ctr.p);
    LAUNCH(root_setup, dim3(1), dim3(64), 0, st, ctr.p, bnodes.p, segs0.p, small.p, N, prm.small);
    numSegs = N > prm.small ? 1u : 0u;
    if (prm.spatial && numSegs) {                                 // split budgets of the references; the root set owns everything behind them
      LAUNCH(spatial_area_sum, dim3(tiles < 2048u ? tiles : 2048u), dim3(256), 0, st, bufA.p, N, ctr.p, 1u);
      LAUNCH(spatial_budgets, dim3(tiles), dim3(256), 0, st, bufA.p, N, (const Counters*)ctr.p, 1u);
      LAUNCH(segx_root, dim3(1), dim3(1), 0, st, segx0.p, NC);
    }
  } else {
    LAUNCH(bounds_fold, dim3(1), dim3(64), 0, st, ctr.p);
    SYNC_READ(h);
    h.numPrims = N - h.numInvalid;
    if (h.numInvalid) {                                          // rare: squeeze the invalid triangles out (stable)
      LAUNCH(compact_count, dim3(tiles), dim3(256), 0, st, bufA.p, N, tileCount.p, (const Counters*)nullptr, N, 0u);
      LAUNCH(compact_scan, dim3(1), dim3(1024), 0, st, tileCount.p, tiles, ctr.p, 0u);
      LAUNCH(compact_scatter, dim3(tiles), dim3(256), 0, st, bufA.p, N, tileCount.p, bufB.p, (const Counters*)nullptr, N);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(bufA.p, bufB.p, (size_t)h.numPrims * sizeof(PrimRef), hipMemcpyDeviceToDevice, st));
      HIP_TRY(hipStreamSynchronize(st)); syncs++;
    }
    n = h.numPrims;
    if (n == 0) { guard.ok = true; *out = bvh; return 0; }
    for (int d = 0; d < 3; d++) { glo[d] = decf(h.bounds[d]); ghi[d] = decf(h.bounds[3 + d]); clo[d] = decf(h.bounds[6 + d]); chi[d] = decf(h.bounds[9 + d]); }
    for (int d = 0; d < 3; d++) { info.bounds_lower[d] = glo[d]; info.bounds_upper[d] = ghi[d]; }
    if (presplit && splitBudget > 0u && n > 0u) {
      SplitGrid grid; float ext = 0.0f;
      for (int d = 0; d < 3; d++) { grid.base[d] = glo[d]; ext = fmaxf(ext, ghi[d] - glo[d]); }
      grid.extend = ext; grid.scale = ext == 0.0f ? 0.0f : 1024.0f / ext;
      const uint32_t ptiles = (n + 255u) / 256u;
      DevBuf<float> prio, partial, psum; DevBuf<uint32_t> cnt, tileSum, totalExtra;
      HIP_TRY(prio.alloc(n)); HIP_TRY(partial.alloc(ptiles)); HIP_TRY(psum.alloc(1)); HIP_TRY(cnt.alloc(n)); HIP_TRY(tileSum.alloc(ptiles)); HIP_TRY(totalExtra.alloc(1));
      LAUNCH(presplit_priority, dim3(ptiles), dim3(256), 0, st, bufA.p, n, dGeoms.p, grid, prio.p, partial.p);
      LAUNCH(presplit_sum, dim3(1), dim3(1024), 0, st, partial.p, ptiles, psum.p);
      float budget = (float)splitBudget; uint32_t extra = 0; bool fits = false;
      for (int attempt = 0; attempt < 6 && !fits; attempt++, budget *= 0.5f) {
        LAUNCH(presplit_count, dim3(ptiles), dim3(256), 0, st, bufA.p, n, dGeoms.p, grid, prio.p, psum.p, budget, cnt.p, tileSum.p);
        LAUNCH(presplit_scan, dim3(1), dim3(1024), 0, st, tileSum.p, ptiles, totalExtra.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&extra, totalExtra.p, 4, hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st)); syncs++;
        fits = extra <= splitBudget;
      }
      if (fits && extra > 0u) {
        LAUNCH(presplit_emit, dim3(ptiles), dim3(256), 0, st, bufA.p, n, dGeoms.p, grid, cnt.p, tileSum.p);
        n += extra;
        Counters hb = h; for (int k = 6; k < 12; k++) hb.bounds[k] = (k < 9) ? ENC_POS_INF : ENC_NEG_INF;
        HIP_TRY(hipMemcpyAsync(ctr.p, &hb, sizeof(hb), hipMemcpyHostToDevice, st));
        const uint32_t cb = (n + 255u) / 256u < 1024u ? (n + 255u) / 256u : 1024u;
        LAUNCH(centroid_bounds, dim3(cb), dim3(256), 0, st, bufA.p, n, ctr.p);
        Counters hb2; SYNC_READ(hb2);
        for (int d = 0; d < 3; d++) { clo[d] = decf(hb2.bounds[6 + d]); chi[d] = decf(hb2.bounds[9 + d]); }
        h.numPrims = n;
      }
      info.num_presplit = extra <= splitBudget ? extra : 0u;
    }
    // root binary node + first work item
    BNode rootB{}; for (int d = 0; d < 3; d++) { rootB.lo[d] = glo[d]; rootB.hi[d] = ghi[d]; } rootB.begin = 0; rootB.end = n; rootB.left = rootB.right = NIL; rootB.splitSah = INFINITY;
    HIP_TRY(hipMemcpyAsync(bnodes.p, &rootB, sizeof(rootB), hipMemcpyHostToDevice, st));
    h.numSegsNext = 0; h.numChunks = 0; h.numSmall = 0; h.numSegs = (n > prm.small && prm.quality != 1u) ? 1u : 0u; h.topLevels = 0;   // (LOW has no top phase: a work list that nobody empties would read as "unfinished" to wide_root)
    h.rootArea = fmaf(ghi[0] - glo[0], (ghi[1] - glo[1]) + (ghi[2] - glo[2]), (ghi[1] - glo[1]) * (ghi[2] - glo[2]));
    if (n > prm.small) {
      Seg s0{}; s0.begin = 0; s0.end = n; s0.bnode = 0; for (int d = 0; d < 3; d++) { s0.cmin[d] = clo[d]; s0.cmax[d] = chi[d]; }
      HIP_TRY(hipMemcpyAsync(segs0.p, &s0, sizeof(s0), hipMemcpyHostToDevice, st)); numSegs = 1;
    } else {
      SmallEntry se{}; se.begin = 0; se.end = n; se.bnode = 0; se.buf = 0; for (int d = 0; d < 3; d++) { se.cmin[d] = clo[d]; se.cmax[d] = chi[d]; }
      HIP_TRY(hipMemcpyAsync(small.p, &se, sizeof(se), hipMemcpyHostToDevice, st)); h.numSmall = 1;
    }
    HIP_TRY(hipMemcpyAsync(ctr.p, &h, sizeof(h), hipMemcpyHostToDevice, st));
    if (spatial && n > prm.small) {                              // split budgets of the references; the root set owns everything behind them
      const uint32_t ab = (n + 255u) / 256u < 2048u ? (n + 255u) / 256u : 2048u;
      LAUNCH(spatial_area_sum, dim3(ab), dim3(256), 0, st, bufA.p, n, ctr.p, 0u);
      LAUNCH(spatial_budgets, dim3((n + 255u) / 256u), dim3(256), 0, st, bufA.p, n, (const Counters*)ctr.p, 0u);
      SegX x0{}; x0.extEnd = NC;
      HIP_TRY(hipMemcpyAsync(segx0.p, &x0, sizeof(x0), hipMemcpyHostToDevice, st));
    }
  }

  if (prm.quality == 1u) {                                     // RTC_BUILD_QUALITY_LOW: Morton codes -> sort -> hierarchy -> boxes
    DevBuf<unsigned long long> keys, keysSorted, sortStatus; DevBuf<uint32_t> vals, valsSorted, parent, flags, sortHist;
    const uint32_t tiles = (n + RS_TILE - 1u) / RS_TILE;          // of the sort (build_sort.inl)
    HIP_TRY(keys.alloc(n)); HIP_TRY(keysSorted.alloc(n)); HIP_TRY(vals.alloc(n)); HIP_TRY(valsSorted.alloc(n)); HIP_TRY(parent.alloc(2ull * n)); HIP_TRY(flags.alloc(n));
    HIP_TRY(sortHist.alloc(RS_HIST_WORDS)); HIP_TRY(sortStatus.alloc((size_t)tiles * RS_RADIX));
    float3 cmin = make_float3(clo[0], clo[1], clo[2]), cscale;
    { const float e[3] = {chi[0] - clo[0], chi[1] - clo[1], chi[2] - clo[2]}; float s3[3]; for (int d = 0; d < 3; d++) s3[d] = e[d] > 0.0f ? 2097152.0f / e[d] : 0.0f; cscale = make_float3(s3[0], s3[1], s3[2]); }
    const uint32_t g = (n + 255u) / 256u;
    HIP_TRY(hipMemsetAsync(sortHist.p, 0, (size_t)RS_HIST_WORDS * 4, st));                      // histograms and tile tickets
    HIP_TRY(hipMemsetAsync(sortStatus.p, 0, (size_t)tiles * RS_RADIX * 8, st));                 // what the tiles publish (tag 0 = nothing; one array for the seven passes)
    LAUNCH(morton_keys, dim3(tiles), dim3(256), 0, st, bufA.p, n, cmin, cscale, keys.p, sortHist.p);
    for (uint32_t pass = 0; pass < RS_PASSES; pass++) {          // (seven passes: the last one lands in keysSorted / valsSorted)
      const bool even = (pass & 1u) == 0u;
      LAUNCH(radix_pass, dim3(tiles), dim3(RS_THREADS), 0, st, (const unsigned long long*)(even ? keys.p : keysSorted.p), (const uint32_t*)(pass == 0u ? nullptr : (even ? vals.p : valsSorted.p)),
             even ? keysSorted.p : keys.p, even ? valsSorted.p : vals.p, n, pass, sortHist.p, sortStatus.p, sortHist.p + RS_PASSES * RS_RADIX);
    }
    LAUNCH(morton_gather, dim3(g), dim3(256), 0, st, bufA.p, valsSorted.p, n, bufB.p, finalIds.p);
    HIP_TRY(hipMemsetAsync(flags.p, 0, (size_t)n * 4, st));
    if (n > 1u) LAUNCH(lbvh_hierarchy, dim3((n + 254u) / 256u), dim3(256), 0, st, keysSorted.p, n, bnodes.p, parent.p);
    LAUNCH(lbvh_bounds, dim3(g), dim3(256), 0, st, bufB.p, n, bnodes.p, parent.p, flags.p, ctr.p);
    HIP_TRY(hipGetLastError());
    numSegs = 0; numSmall = 0;
  }
  const bool sahBuild = prm.quality != 1u;
  // ---- top phase: one pass over the data per binary level.  Work-list sizes stay on the device: every kernel is launched with
  //      an upper bound of its grid (<= 2^level segments, <= N/CHUNK + #segments chunks) and surplus blocks exit at once, so
  //      the levels are enqueued back to back.
  uint32_t level = 0;
  Seg* cur = segs0.p; Seg* nxt = segs1.p; SegX* xcur = segx0.p; SegX* xnxt = segx1.p;     // (the SegX arrays: nullptr unless spatial)
  auto enqueue_top_level = [&]() {
    PrimRef* src = (level & 1u) ? bufB.p : bufA.p; PrimRef* dst = (level & 1u) ? bufA.p : bufB.p;
    const uint32_t segBound = level < 31u && (1u << level) < maxSegs ? (1u << level) : maxSegs;
    const uint32_t chunkBound = (spatial ? NC : n) / CHUNK + segBound + 1u;
    // sets of at most CHUNK references are one workgroup's (top_local); the others go through the chunked path.  A commit that knows the last commit of
    // this size enqueues only what that one needed (+ a level of margin either way): the chunked path up to its last level with a large set, top_local
    // from its first level with a small one.  A large set where no chunked path was enqueued raises overflow 3: the commit runs again, blind.
    const bool local = !spatial && (!learned || level + 1u >= learnedLocalFirst);
    const bool chunked = !local || !learned || level <= learnedChunked;
    const uint32_t localMax = local ? CHUNK : 0u, dstBuf = (level & 1u) ? 0u : 1u, forceFallback = level >= 96u ? 1u : 0u;
    if (!chunked) {
      LAUNCH(top_local, dim3(segBound), dim3(256), 0, st, (const Seg*)cur, (const PrimRef*)src, dst, bnodes.p, nxt, small.p, ctr.p, prm, dstBuf, maxSegs, maxSmall, forceFallback, level, 1u);
      LAUNCH(top_emit, dim3((segBound + 255u) / 256u), dim3(256), 0, st, cur, bnodes.p, nxt, small.p, ctr.p, prm, dstBuf, maxSegs, maxSmall, (const SegX*)nullptr, (SegX*)nullptr, 0xFFFFFFFFu, (const uint32_t*)nullptr);   // (moves the work lists on)
      Seg* t = cur; cur = nxt; nxt = t; level++;
      return;
    }
    const uint32_t listBlocks = (segBound + SETUP_SEGS - 1u) / SETUP_SEGS;
    LAUNCH(top_setup, dim3(listBlocks + segBound), dim3(256), 0, st, cur, bins.p, chunks.p, ctr.p, localMax, level, chunkFlag.p, binsTop.p, listBlocks);
    LAUNCH(top_bin, dim3(chunkBound), dim3(256), 0, st, cur, chunks.p, src, bins.p, ctr.p, chunkCnt.p, maxChunks, binsTop.p);
    LAUNCH(top_split, dim3(segBound), dim3(64), 0, st, cur, bins.p, bnodes.p, ctr.p, prm, forceFallback, xcur, (const uint32_t*)chunkCnt.p, chunkBase.p, localMax, maxChunks, accTop.p, (const uint32_t*)binsTop.p);
    if (spatial) {                                          // sets whose object split leaves overlapping children try a spatial split
      LAUNCH(spatial_decide, dim3(segBound), dim3(128), 0, st, cur, xcur, bnodes.p, sbins.p, ctr.p, spatialMin, sbinsTop.p);
      LAUNCH(spatial_bin, dim3(chunkBound), dim3(256), 0, st, cur, xcur, chunks.p, src, dGeoms.p, sbins.p, ctr.p, sbinsTop.p);
      LAUNCH(spatial_best, dim3(segBound), dim3(64), 0, st, cur, xcur, sbins.p, bnodes.p, ctr.p, prm, (const uint32_t*)sbinsTop.p);
    }
    LAUNCH(top_partition, dim3(chunkBound), dim3(256), 0, st, cur, chunks.p, src, dst, ctr.p, (const uint2*)chunkBase.p, accTop.p);
    if (spatial) LAUNCH(spatial_partition, dim3(chunkBound), dim3(256), 0, st, cur, xcur, chunks.p, src, dst, dGeoms.p, ctr.p, chunkFlag.p, accTop.p);
    if (local) LAUNCH(top_local, dim3(segBound), dim3(256), 0, st, (const Seg*)cur, (const PrimRef*)src, dst, bnodes.p, nxt, small.p, ctr.p, prm, dstBuf, maxSegs, maxSmall, forceFallback, level, 0u);
    LAUNCH(top_emit, dim3((segBound + 255u) / 256u), dim3(256), 0, st, cur, bnodes.p, nxt, small.p, ctr.p, prm,
           dstBuf, maxSegs, maxSmall, (const SegX*)xcur, xnxt, localMax, (const uint32_t*)accTop.p);
    Seg* t = cur; cur = nxt; nxt = t; SegX* tx = xcur; xcur = xnxt; xnxt = tx; level++;
  };
  if (numSegs && sahBuild) {
    uint32_t sure = 1; while (sure < 40u && ((uint64_t)prm.small << sure) < n) sure++;   // the largest segment halves at best: that many levels exist
    if (fast) {                                                                             // + margin: SAH splits are uneven (crown: 17 levels where 13 are implied)
      const uint32_t margin = 10u;                                           // (spatial splits add references on the way down: the powerplant stand-in's HIGH tree has 25 levels where 16 are implied)
      const uint32_t levels = learned ? min(sure + margin, learnedTop + 1u) : sure + margin;   // (what the last commit of this size needed, + 1)
      // (round 6: small_threshold 1024 -> 512 -- one more level is top_local's, a whole workgroup per set with the references in registers, instead of the large mode of
      // small_build, one wavefront per set through L2: crown 5.15 -> 5.06 ms, same tree; the powerplant stand-in then has 24 top levels where 15 are implied, hence a
      // margin of 10 for every quality (8 sent it to the stepwise path: 12.97 ms instead of 12.5))
      for (uint32_t i = 0; i < levels; i++) enqueue_top_level();
    }
    else {
      for (uint32_t i = 0; i < sure; i++) enqueue_top_level();
      for (;;) {
        SYNC_READ(h);
        if (h.overflow == 4u) return set_error(hipErrorLaunchFailure, "spatial_partition gave up waiting for a predecessor chunk (workgroups not started in index order?)");
        if (h.overflow) return set_error(hipErrorOutOfMemory, h.overflow == 2u ? "spatial split ran out of its extended range" : "top-phase work list overflow (pathological input)");
        if (h.numSegs == 0) break;
        enqueue_top_level(); enqueue_top_level();
      }
    }
  } else if (!fast) { SYNC_READ(h); }

  // ---- small phase
  const uint32_t microW = prm.minLeaf >= 2u ? 32u : 48u;       // LDS words per triangle of the micro mode (see micro_subtree)
  static const size_t smallLdsPad = getenv("MI355_SMALL_LDS_PAD") ? (size_t)atol(getenv("MI355_SMALL_LDS_PAD")) : 0u;   // A/B: fewer workgroups per CU
  const size_t smallLds = sizeof(uint32_t) * (64u * microW > (uint32_t)BINS_WORDS ? 64u * microW : (uint32_t)BINS_WORDS) + smallLdsPad;
  if (fast) {
    // every small entry covers > small / 2^k ... triangles: at most one entry per top-phase leaf; the list cannot be longer than maxSmall (top_emit raises overflow)
    const uint32_t bound = N > prm.small ? maxSmall : 1u;
    static const bool envOrder = !(getenv("MI355_SMALL_ORDER") && atoi(getenv("MI355_SMALL_ORDER")) == 0);   // A/B: 0 = the list as it is
    if (envOrder && N > prm.small) LAUNCH(small_order, dim3(1), dim3(1024), 0, st, (const SmallEntry*)small.p, (const Counters*)ctr.p, smallOrder.p, maxSmall, prm.small);
    LAUNCH(small_build, dim3(bound), dim3(64), smallLds, st, small.p, bufA.p, bufB.p, bnodes.p, finalIds.p, ctr.p, prm, microW, envOrder && N > prm.small ? (const uint32_t*)smallOrder.p : (const uint32_t*)nullptr);
  } else {
    info.top_levels = h.topLevels;
    numSmall = sahBuild ? h.numSmall : 0u;
    if (numSmall > maxSmall) return set_error(hipErrorOutOfMemory, "small list overflow");
    if (numSmall) LAUNCH(small_build, dim3(numSmall), dim3(64), smallLds, st, small.p, bufA.p, bufB.p, bnodes.p, finalIds.p, ctr.p, prm, microW, (const uint32_t*)nullptr);
  }
  HIP_TRY(hipGetLastError());

  // ---- wide collapse, level by level
  LAUNCH(wide_root, dim3(1), dim3(1), 0, st, w0.p, ctr.p);
  WideItem* wc = w0.p; WideItem* wn = w1.p;
  uint32_t wlevel = 0;
  auto enqueue_wide_level = [&]() {
    uint64_t bound = 1; for (uint32_t i = 0; i < wlevel && bound < maxLevelItems; i++) bound *= 8u;     // <= 8^level items
    if (bound > maxLevelItems) bound = maxLevelItems;
    const uint32_t blocks = (uint32_t)((bound + 7u) / 8u) < 8192u ? (uint32_t)((bound + 7u) / 8u) : 8192u;   // = the waves resident at once
    const uint32_t parity = wlevel & 1u;
    LAUNCH(wide_plan, dim3(blocks), dim3(64), 0, st, wc, bnodes.p, plans.p, itemCnt.p, groupSum.p, ctr.p, prm, parity);
    LAUNCH(wide_scan, dim3(1), dim3(1024), 0, st, groupSum.p, ctr.p, parity, maxWide, blocks);
    LAUNCH(wide_emit, dim3(blocks), dim3(64), 0, st, wc, bnodes.p, plans.p, itemCnt.p, groupSum.p, wnodes.p, finalIds.p, outIds.p, wn, ctr.p, parity);
    WideItem* t = wc; wc = wn; wn = t; wlevel++;
  };
  if (fast) {
    // the depth of the wide tree is unknown here: 16 levels cover every scene measured so far (crown 12, powerplant 13); a deeper tree is finished below
    // (every level enqueued beyond the last one costs three empty launches, ~14 us)
    // (a commit that knows the depth the commits of its kind needed enqueues exactly that many levels: a deeper tree is finished level by level below -- one more round trip and
    // the leaf records written again, once, after which the kind's depth has grown -- where a level of margin cost EVERY commit three empty launches, ~17 us)
    { const uint32_t levels = learned && learnedWide ? min(16u, learnedWide) : 16u; for (uint32_t i = 0; i < levels; i++) enqueue_wide_level(); }
    if (capturing) {                                             // end of the captured sequence: instantiate, keep, run
      hipGraph_t graph = nullptr;
      const hipError_t e = hipStreamEndCapture(st, &graph);
      capturing = false;
      if (e != hipSuccess || !graph) { (void)hipGetLastError(); arena->drop_graph(); arena->graphBroken = true; return -1000; }   // the caller repeats the commit with plain launches
      const hipError_t e2 = hipGraphInstantiate(&arena->graphExec, graph, nullptr, nullptr, 0);
      hipGraphDestroy(graph);
      if (e2 != hipSuccess) { (void)hipGetLastError(); arena->graphExec = nullptr; arena->drop_graph(); arena->graphBroken = true; return -1000; }
      replay = true;
    }
    if (replay) { HIP_TRY(hipEventRecord(ev0, st)); HIP_TRY(hipGraphLaunch(arena->graphExec, st)); replay = false; }
    // the leaf records can be written as soon as the leaf order is known; their array is sized by the upper bound N
    bvh->d_tris = output_alloc(device, (size_t)NC * sizeof(TriRec) + 128, &bvh->trisCap);
    if (!bvh->d_tris) return set_error(hipErrorOutOfMemory, "leaf record array");
    // the node array leaves the arena beside the record gathers if the last commit of this kind said how large it gets (tri_records; + 1/64 of slack)
    if (lc && lc->nodes) {
      preCap = lc->nodes + lc->nodes / 64u + 256u;
      bvh->d_nodes = output_alloc(device, (size_t)preCap * sizeof(CNode), &bvh->nodesCap);
      if (!bvh->d_nodes) { preCap = 0u; bvh->nodesCap = 0; }
    }
    const uint32_t triBlocks = (NC + 1023u) / 1024u;
    const uint32_t copyBlocks = preCap ? 1024u : 0u;
    LAUNCH(tri_records, dim3(copyBlocks + triBlocks), dim3(256), 0, st, outIds.p, NC, dGeoms.p, (TriRec*)bvh->d_tris, bp->robust ? 1u : 0u, (const Counters*)ctr.p,
           (const uint4*)wnodes.p, (uint4*)bvh->d_nodes, preCap, copyBlocks);
    SYNC_READ(h);                                                // the ONE round trip of the commit
    // (the overflow word is written with atomicMax -- a later kernel's 1 / 2 / 3 no longer hides a 4 -- and a wait that timed out (2^20 naps: heavy sharing of the GPU,
    // preemption) is a reason to run the commit AGAIN on the stepwise path, not an error; only if that one times out as well does the host report it)
    if (h.overflow == 4u && spatial) { if (allowFast) return -1000; return set_error(hipErrorLaunchFailure, "spatial_partition gave up waiting for a predecessor chunk (workgroups not started in index order?)"); }
    if (h.overflow == 2u && spatial) return set_error(hipErrorOutOfMemory, "spatial split ran out of its extended range");
    // Any overflow of a commit that ran on learned counts is first of all a doubt about those counts (a large set below the last level the chunked path was enqueued for raises 3;
    // a work list that a later kernel found too short may overwrite that 3 with 1): the commit runs again with the blind margins, which report what this scene needs -- the
    // counts of its kind grow to that (Arena::learn takes the maximum) -- and only an overflow of THAT run is an error.  (it was a hard out-of-memory error.)
    if (h.overflow && learned) return -1001;
    if (h.overflow) return set_error(hipErrorOutOfMemory, "work list overflow (pathological input)");
    if (h.numSegs != 0u) {                                       // the top phase needed more levels than were enqueued: what came after it worked on an unfinished tree
      if (learned) return -1001;                                 // (counts learned from another scene of this kind: again, with the blind margins; what that run needs is added to them)
      arena->marginFailedN = N;                                  // more than N implies + 8
      return -1000;                                              // (the guard frees the half-built tree) the caller repeats the commit on the stepwise path
    }
    n = h.numPrims;
    info.num_presplit = h.outlierPieces > h.numOutliers ? h.outlierPieces - h.numOutliers : 0u;   // leaf records beyond one per triangle: the pieces of the cut outliers
    if (n == 0) { output_free(device, bvh->d_tris, bvh->trisCap); bvh->d_tris = nullptr; bvh->trisCap = 0; if (bvh->d_nodes) { output_free(device, bvh->d_nodes, bvh->nodesCap); bvh->d_nodes = nullptr; bvh->nodesCap = 0; } info.num_launches = launches; info.num_host_syncs = syncs; guard.ok = true; *out = bvh; return 0; }
    for (int d = 0; d < 3; d++) { info.bounds_lower[d] = decf(h.bounds[d]); info.bounds_upper[d] = decf(h.bounds[3 + d]); }
    info.top_levels = h.topLevels;
    bool redoLeaves = false;
    while (h.wideCount[wlevel & 1u] != 0u) {                     // deeper than the levels enqueued: go on level by level, then write the leaf records again
      for (uint32_t i = 0; i < 4u; i++) enqueue_wide_level();
      SYNC_READ(h);
      if (h.overflow) return set_error(hipErrorOutOfMemory, "wide node pool overflow");
      redoLeaves = true; preCap = 0u;                            // (the nodes copied beside the first tri_records were not all of them)
    }
    if (spatial) { info.num_presplit = h.numTrisOut > n ? h.numTrisOut - n : 0u; n = h.numTrisOut; }   // the references the spatial splits created are leaf entries like any other
    arena->learn(kind, h.topLevels, h.wideDepth, h.chunkedLevels, h.localFirst < 255u ? h.localFirst : 255u, h.numWide);   // (a tree deeper than the wide levels enqueued is finished below either way)
    if (redoLeaves) LAUNCH(tri_records, dim3((NC + 1023u) / 1024u), dim3(256), 0, st, outIds.p, NC, dGeoms.p, (TriRec*)bvh->d_tris, bp->robust ? 1u : 0u, (const Counters*)ctr.p, (const uint4*)nullptr, (uint4*)nullptr, 0u, 0u);
  } else {
    for (uint32_t i = 0; i < 8u; i++) enqueue_wide_level();
    for (;;) {
      SYNC_READ(h);
      if (h.overflow) return set_error(hipErrorOutOfMemory, "wide node pool overflow");
      if (h.wideCount[wlevel & 1u] == 0) break;
      for (uint32_t i = 0; i < 4u; i++) enqueue_wide_level();
    }
    if (spatial) { info.num_presplit = h.numTrisOut > n ? h.numTrisOut - n : 0u; n = h.numTrisOut; }   // the references the splits created are leaf entries like any other
    bvh->d_tris = output_alloc(device, (size_t)n * sizeof(TriRec) + 128, &bvh->trisCap);
    if (!bvh->d_tris) return set_error(hipErrorOutOfMemory, "leaf record array");
    LAUNCH(tri_records, dim3((n + 1023u) / 1024u), dim3(256), 0, st, outIds.p, n, dGeoms.p, (TriRec*)bvh->d_tris, bp->robust ? 1u : 0u, (const Counters*)nullptr, (const uint4*)nullptr, (uint4*)nullptr, 0u, 0u);
  }
  info.num_triangles = n;
#ifdef SM_TIME
  fprintf(stderr, "[mi355 build] small_build wave Mcycles: bin %.1f, price %.1f, partition %.1f, micro %.1f\n", h.smTime[0] * 1e-6, h.smTime[1] * 1e-6, h.smTime[2] * 1e-6, h.smTime[3] * 1e-6);
#endif
#ifdef SM_STATS
  fprintf(stderr, "[mi355 build] micro level passes %u, lanes in use %.1f of 64\n", h.padC[0], h.padC[0] ? (double)h.padC[1] / h.padC[0] : 0.0);
#endif
  const uint32_t depth = h.wideDepth;

  // ---- final node array (exact size)
  const uint32_t numNodes = h.numWide;
  if (bvh->d_nodes && !(preCap && numNodes && numNodes <= preCap)) { output_free(device, bvh->d_nodes, bvh->nodesCap); bvh->d_nodes = nullptr; bvh->nodesCap = 0; }   // (sized up front, too small or not filled)
  if (numNodes && !bvh->d_nodes) {
    bvh->d_nodes = output_alloc(device, (size_t)numNodes * sizeof(CNode), &bvh->nodesCap);
    if (!bvh->d_nodes) return set_error(hipErrorOutOfMemory, "node array");
    HIP_TRY(hipMemcpyAsync(bvh->d_nodes, wnodes.p, (size_t)numNodes * sizeof(CNode), hipMemcpyDeviceToDevice, st));
  }
  bvh->robust = bp->robust != 0;
  if (bp->refit && h.numInvalid == 0u && depth < 64u && prm.quality != 2u && !spatial) {        // keep the leaf order and the level table for mi355_bvh_refit
    { const int rc = mi355_malloc_retry(device, (size_t)n * sizeof(uint2), &bvh->d_ids); if (rc) return rc; }
    HIP_TRY(hipMemcpyAsync(bvh->d_ids, outIds.p, (size_t)n * sizeof(uint2), hipMemcpyDeviceToDevice, st));
    bvh->lvlStart.assign(h.lvlStart, h.lvlStart + depth); bvh->lvlStart.push_back(numNodes);
    for (const GeomDesc& g : gd) bvh->sig.push_back({g.geomID, g.nt, g.nv, g.quad});
    info.bytes_refit = (uint64_t)n * sizeof(uint2);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ev1, st)); HIP_TRY(hipEventSynchronize(ev1)); syncs++;
  float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
  bvh->root = h.rootRef;
  unsigned long long sahFixed = 0ull; uint32_t numLeaves = 0u, numBLeaves = 0u;            // the statistics every workgroup reported to its stripe (Counters::Stripe)
  for (uint32_t r = 0; r < Counters::STRIPES; r++) { sahFixed += h.stripe[r].sahFixed; numLeaves += h.stripe[r].numLeaves; numBLeaves += h.stripe[r].numBLeaves; }
  info.root_ref = h.rootRef; info.num_nodes = numNodes; info.num_leaves = numLeaves; info.num_binary_nodes = 2ull * numBLeaves - 1ull;
  info.bytes_nodes = (uint64_t)numNodes * sizeof(CNode); info.bytes_triangles = (uint64_t)n * sizeof(TriRec);
  info.sah = (float)((double)sahFixed / 16777216.0) + (numNodes ? prm.travCost : 0.0f); info.build_ms = ms; info.depth = depth;
  info.num_launches = launches; info.num_host_syncs = syncs;
  guard.ok = true; *out = bvh;
  return 0;
#undef LAUNCH
#undef SYNC_READ
}

// the one-round-trip path can ask for the commit to be repeated on the stepwise path (-1000: level margins exceeded); that path does not cut outliers
static int build_retry(int device, const mi355_mesh* meshes, uint32_t numMeshes, const mi355_build_params* bp, hipStream_t st, Bvh** out) {
  uint32_t attempts = 1;
  int rc = build_impl(device, meshes, numMeshes, bp, st, out);
  if (rc == -1001) { attempts++; rc = build_impl(device, meshes, numMeshes, bp, st, out, true, true, false); }   // the level counts learned from the last commit of this size were too few
  if (rc == -1000) { attempts++; rc = build_impl(device, meshes, numMeshes, bp, st, out, false, false); }
  if (rc == 0 && *out) (*out)->info.build_attempts = attempts;
  return rc;
}

static int refit_impl(Bvh* bvh, const mi355_mesh* meshes, uint32_t numMeshes, hipStream_t st) {
  HIP_TRY(hipSetDevice(bvh->device));
  const uint32_t n = (uint32_t)bvh->info.num_triangles, numNodes = (uint32_t)bvh->info.num_nodes;
  if (!bvh->d_ids || n == 0u || numNodes == 0u) return MI355_REFIT_IMPOSSIBLE;
  std::vector<GeomDesc> gd; uint64_t total = 0;
  for (uint32_t i = 0; i < numMeshes; i++) {
    const mi355_mesh& m = meshes[i];
    if (m.num_triangles == 0) continue;
    if ((m.vertex_stride & 3) || (m.index_stride & 3)) return set_error(hipErrorInvalidValue, "buffer stride");
    GeomDesc g{}; g.verts = (const char*)m.d_vertices; g.idx = (const char*)m.d_indices; g.vstride = (uint32_t)m.vertex_stride; g.istride = (uint32_t)m.index_stride;
    g.nv = m.num_vertices; g.quad = m.quads ? 1u : 0u; g.nt = m.num_triangles * (g.quad ? 2u : 1u); g.geomID = m.geom_id; g.mask = m.mask; g.primOffset = (uint32_t)total;
    total += g.nt; gd.push_back(g);
  }
  if (gd.size() != bvh->sig.size()) return MI355_REFIT_IMPOSSIBLE;
  for (size_t i = 0; i < gd.size(); i++) {
    const Bvh::MeshSig& s = bvh->sig[i];
    if (s.geomID != gd[i].geomID || s.numPrims != gd[i].nt || s.numVerts != gd[i].nv || s.quads != gd[i].quad) return MI355_REFIT_IMPOSSIBLE;
  }
  Arena* arena = arena_of(bvh->device);
  std::lock_guard<std::mutex> arenaLock(arena->mtx);
  arena->reset(); t_arena = arena;
  DevBuf<GeomDesc> dGeoms; DevBuf<float4> boxes; DevBuf<uint32_t> flag;
  HIP_TRY(dGeoms.alloc(gd.size())); HIP_TRY(boxes.alloc(2ull * numNodes)); HIP_TRY(flag.alloc(1));
  hipEvent_t ev0, ev1; HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1));
  struct EvGuard { hipEvent_t a, b; ~EvGuard() { hipEventDestroy(a); hipEventDestroy(b); } } evg{ev0, ev1};
  HIP_TRY(hipEventRecord(ev0, st));
  HIP_TRY(hipMemcpyAsync(dGeoms.p, gd.data(), gd.size() * sizeof(GeomDesc), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(flag.p, 0, 4, st));
  hipLaunchKernelGGL(tri_records, dim3((n + 1023u) / 1024u), dim3(256), 0, st, (const uint2*)bvh->d_ids, n, dGeoms.p, (TriRec*)bvh->d_tris, bvh->robust ? 1u : 0u, (const Counters*)nullptr, (const uint4*)nullptr, (uint4*)nullptr, 0u, 0u);
  for (size_t l = bvh->lvlStart.size() - 1; l-- > 0;) {        // deepest level first
    const uint32_t first = bvh->lvlStart[l], count = bvh->lvlStart[l + 1] - first;
    if (!count) continue;
    const uint32_t blocks = (count + 7u) / 8u < 8192u ? (count + 7u) / 8u : 8192u;
    hipLaunchKernelGGL(refit_level, dim3(blocks), dim3(64), 0, st, (CNode*)bvh->d_nodes, boxes.p, (const uint2*)bvh->d_ids, dGeoms.p, first, count, flag.p);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ev1, st));
  uint32_t hflag = 0; float4 rb[2];
  HIP_TRY(hipMemcpyAsync(&hflag, flag.p, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(rb, boxes.p, sizeof(rb), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
  if (hflag) return MI355_REFIT_BROKEN;
  mi355_bvh_info& info = bvh->info;
  info.bounds_lower[0] = rb[0].x; info.bounds_lower[1] = rb[0].y; info.bounds_lower[2] = rb[0].z;
  info.bounds_upper[0] = rb[1].x; info.bounds_upper[1] = rb[1].y; info.bounds_upper[2] = rb[1].z;
  info.build_ms = ms; info.num_refits++;
  return 0;
}

// ---- scenes with RTC_GEOMETRY_TYPE_INSTANCE (one level).  Host arithmetic of Instance::commit / Instance::bounds:
// world2local = rcp(local2world) = (il = adjoint / det, -(il * p)) (kernels/common/scene_instance.cpp:150-153, common/math/affinespace.h:83,
// linearspace3.h:44-51; cross = fma(a.y, b.z, -(a.z * b.y)) ..., vec3fa.h:334-341; det = DPPS), world box = xfmBounds (affinespace.h:106-118).
static void affine_inverse(const float* m, float* o) {
  const float *vx = m, *vy = m + 3, *vz = m + 6, *p = m + 9;
  auto cross = [](const float* a, const float* b, float* c) { c[0] = fmaf(a[1], b[2], -(a[2] * b[1])); c[1] = fmaf(a[2], b[0], -(a[0] * b[2])); c[2] = fmaf(a[0], b[1], -(a[1] * b[0])); };
  float c0[3], c1[3], c2[3]; cross(vy, vz, c0); cross(vz, vx, c1); cross(vx, vy, c2);
  const float det = (vx[0] * c0[0] + vx[1] * c0[1]) + (vx[2] * c0[2] + 0.0f);
  for (int r = 0; r < 3; r++) { const float a = r == 0 ? c0[0] : r == 1 ? c1[0] : c2[0], b = r == 0 ? c0[1] : r == 1 ? c1[1] : c2[1], c = r == 0 ? c0[2] : r == 1 ? c1[2] : c2[2]; o[r] = a / det; o[3 + r] = b / det; o[6 + r] = c / det; }
  for (int r = 0; r < 3; r++) o[9 + r] = -fmaf(p[0], o[r], fmaf(p[1], o[3 + r], p[2] * o[6 + r]));
}
static void affine_point(const float* m, const float* p, float* o) { for (int r = 0; r < 3; r++) o[r] = fmaf(p[0], m[r], fmaf(p[1], m[3 + r], fmaf(p[2], m[6 + r], m[9 + r]))); }

struct InstRec { float w2l[12]; uint32_t root, instID, mask, flags; };
static_assert(sizeof(InstRec) == 64, "InstRec must be 64 bytes");

static int build_instanced_impl(int device, Bvh* own, const mi355_instance* insts, uint32_t numInsts, const mi355_build_params* bp, hipStream_t st, Bvh** out) {
  HIP_TRY(hipSetDevice(device));
  // ---- the trees that go into the combined arrays: the scene's own geometry first, then every distinct instanced tree
  std::vector<Bvh*> objs; std::map<Bvh*, uint32_t> objIndex;
  auto use = [&](Bvh* b) -> int {
    if (objIndex.count(b)) return 0;
    if (b->device != device) return set_error(hipErrorInvalidValue, "instanced scene lives on another device");
    if (b->d_insts) return set_error(hipErrorInvalidValue, "the tree of an instanced scene must be flat");
    if (b->robust != (bp->robust != 0)) return set_error(hipErrorInvalidValue, "RTC_SCENE_FLAG_ROBUST must be the same for a scene and the scenes it instances");
    objIndex[b] = (uint32_t)objs.size(); objs.push_back(b); return 0;
  };
  const bool hasOwn = own && own->info.num_triangles > 0;
  if (hasOwn) { const int rc = use(own); if (rc) return rc; }
  std::vector<InstRec> recs; std::vector<float> boxes;           // boxes: lo.xyz hi.xyz per record
  float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
  auto add_rec = [&](const InstRec& r, const float* lo, const float* hi) {
    recs.push_back(r); for (int d = 0; d < 3; d++) boxes.push_back(lo[d]); for (int d = 0; d < 3; d++) boxes.push_back(hi[d]);
    for (int d = 0; d < 3; d++) { blo[d] = fminf(blo[d], lo[d]); bhi[d] = fmaxf(bhi[d], hi[d]); }
  };
  if (hasOwn) {
    InstRec r{}; r.w2l[0] = r.w2l[4] = r.w2l[8] = 1.0f; r.root = 0; r.instID = MI355_EMPTY_REF; r.mask = 0xFFFFFFFFu; r.flags = 1u;
    add_rec(r, own->info.bounds_lower, own->info.bounds_upper);
  }
  std::vector<Bvh*> recObj; if (hasOwn) recObj.push_back(own);
  for (uint32_t i = 0; i < numInsts; i++) {
    Bvh* ob = (Bvh*)insts[i].object;
    if (!ob || ob->info.num_triangles == 0) continue;              // an empty object scene has empty bounds: Instance::buildBounds says invalid
    const float* l2w = insts[i].local2world;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int c = 0; c < 8; c++) {
      const float p[3] = {(c & 4) ? ob->info.bounds_upper[0] : ob->info.bounds_lower[0], (c & 2) ? ob->info.bounds_upper[1] : ob->info.bounds_lower[1], (c & 1) ? ob->info.bounds_upper[2] : ob->info.bounds_lower[2]};
      float q[3]; affine_point(l2w, p, q);
      for (int d = 0; d < 3; d++) { lo[d] = fminf(lo[d], q[d]); hi[d] = fmaxf(hi[d], q[d]); }
    }
    bool ok = true; for (int d = 0; d < 3; d++) ok = ok && lo[d] > -1.844E18f && hi[d] < 1.844E18f && lo[d] <= hi[d];
    if (!ok) continue;
    const int rc = use(ob); if (rc) return rc;
    InstRec r{}; affine_inverse(l2w, r.w2l); r.root = 0; r.instID = insts[i].inst_id; r.mask = insts[i].mask; r.flags = 0u;
    add_rec(r, lo, hi); recObj.push_back(ob);
  }
  Bvh* bvh = new Bvh; bvh->device = device;
  struct Guard { Bvh*& b; bool ok = false; ~Guard() { if (!ok) { delete b; b = nullptr; } } } guard{bvh};
  bvh->robust = bp->robust != 0;
  mi355_bvh_info& info = bvh->info; memset(&info, 0, sizeof(info));
  info.root_ref = MI355_EMPTY_REF; info.max_leaf = MI355_MAX_LEAF;
  for (int d = 0; d < 3; d++) { info.bounds_lower[d] = blo[d]; info.bounds_upper[d] = bhi[d]; }
  if (recs.empty()) { bvh->numCUs = own ? own->numCUs : 256; guard.ok = true; *out = bvh; return 0; }

  // ---- top tree: the SAH builder over one box per record.  The box enters as a triangle (lo, hi, lo): its PrimRef is exactly the box.
  const uint32_t R = (uint32_t)recs.size();
  std::vector<float> fv((size_t)R * 9 + 4, 0.0f); std::vector<uint32_t> fi((size_t)R * 3);
  for (uint32_t k = 0; k < R; k++) {
    const float* b = &boxes[(size_t)k * 6];
    for (int d = 0; d < 3; d++) { fv[(size_t)k * 9 + d] = b[d]; fv[(size_t)k * 9 + 3 + d] = b[3 + d]; fv[(size_t)k * 9 + 6 + d] = b[d]; }
    fi[(size_t)k * 3] = 3 * k; fi[(size_t)k * 3 + 1] = 3 * k + 1; fi[(size_t)k * 3 + 2] = 3 * k + 2;
  }
  float* dfv = nullptr; uint32_t* dfi = nullptr; InstRec* dRecs = nullptr;
  struct TmpGuard { float*& a; uint32_t*& b; ~TmpGuard() { if (a) hipFree(a); if (b) hipFree(b); } } tmpGuard{dfv, dfi};   // (handed to the tree at the end: then both are nullptr)
  HIP_TRY(hipMalloc((void**)&dfv, fv.size() * 4)); HIP_TRY(hipMalloc((void**)&dfi, fi.size() * 4));
  HIP_TRY(hipMemcpyAsync(dfv, fv.data(), fv.size() * 4, hipMemcpyHostToDevice, st)); HIP_TRY(hipMemcpyAsync(dfi, fi.data(), fi.size() * 4, hipMemcpyHostToDevice, st));
  mi355_mesh fake{}; fake.d_vertices = dfv; fake.vertex_stride = 12; fake.num_vertices = 3 * R; fake.d_indices = dfi; fake.index_stride = 12; fake.num_triangles = R; fake.geom_id = 0; fake.mask = 0xFFFFFFFFu;
  mi355_build_params tp = *bp; tp.refit = 1; tp.quality = 0;     // (refit data: instances that only MOVE refit this tree, mi355_bvh_refit_instanced)
  Bvh* top = nullptr;
  tp.top_splits = 0;                                            // (the "triangles" of this build are the instances' boxes: one leaf record each)
  { const int rc = build_retry(device, &fake, 1, &tp, st, &top); if (rc) return rc; }
  struct TopGuard { Bvh*& t; ~TopGuard() { delete t; } } topGuard{top};   // (handed to the tree at the end: then nullptr)
  if (top->info.num_triangles != R) return set_error(hipErrorInvalidValue, "top-level build dropped an instance");
  bvh->numCUs = top->numCUs;

  // ---- combined arrays: [top | object 0 | object 1 ...], the objects' indices moved behind
  std::vector<uint32_t> nodeOfs(objs.size()), triOfs(objs.size());
  uint64_t nNodes = top->info.num_nodes, nTris = top->info.num_triangles; uint32_t maxDepth = 0; uint64_t objTris = 0;
  for (size_t k = 0; k < objs.size(); k++) {
    nodeOfs[k] = (uint32_t)nNodes; triOfs[k] = (uint32_t)nTris;
    nNodes += objs[k]->info.num_nodes; nTris += objs[k]->info.num_triangles; objTris += objs[k]->info.num_triangles;
    if (objs[k]->info.depth > maxDepth) maxDepth = objs[k]->info.depth;
  }
  if (nNodes >= (1ull << 32) || nTris >= (1ull << 32)) return set_error(hipErrorInvalidValue, "instanced scene exceeds the 32-bit node / triangle index");
  { int rc = mi355_malloc_retry(device, (size_t)nNodes * sizeof(CNode), &bvh->d_nodes); if (rc) return rc; rc = mi355_malloc_retry(device, (size_t)nTris * sizeof(TriRec) + 128, &bvh->d_tris); if (rc) return rc; }
  HIP_TRY(hipMemcpyAsync(bvh->d_nodes, top->d_nodes, (size_t)top->info.num_nodes * sizeof(CNode), hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(bvh->d_tris, top->d_tris, (size_t)top->info.num_triangles * sizeof(TriRec), hipMemcpyDeviceToDevice, st));
  for (size_t k = 0; k < objs.size(); k++) {
    const uint32_t n = (uint32_t)objs[k]->info.num_nodes;
    if (n) hipLaunchKernelGGL(rebase_nodes, dim3((n + 255u) / 256u), dim3(256), 0, st, (const uint4*)objs[k]->d_nodes, (uint4*)bvh->d_nodes + 5ull * nodeOfs[k], n, nodeOfs[k], triOfs[k]);
    HIP_TRY(hipMemcpyAsync((char*)bvh->d_tris + (size_t)triOfs[k] * sizeof(TriRec), objs[k]->d_tris, (size_t)objs[k]->info.num_triangles * sizeof(TriRec), hipMemcpyDeviceToDevice, st));
  }
  // device-side filter rules: one table for the combined tree -- every distinct object's entries behind a base of its own (InstRec.flags >> 8), bit arrays at the end
  std::vector<uint32_t> ruleBase(objs.size(), 0u);
  { bool any = false; uint32_t entries = 0;
    for (size_t k = 0; k < objs.size(); k++) { ruleBase[k] = entries; entries += objs[k]->numRuleGeoms; any = any || !objs[k]->h_rules.empty(); }
    if (any && entries < (1u << 24)) {
      std::vector<uint32_t> tab((size_t)entries * 12u, 0u);
      for (size_t k = 0; k < objs.size(); k++) {
        const std::vector<uint32_t>& h = objs[k]->h_rules; const uint32_t ng = objs[k]->numRuleGeoms;
        if (h.size() < (size_t)ng * 12u) continue;                  // (an object without rules: its entries stay zero = no rule)
        const uint32_t bitsAt = (uint32_t)tab.size(), oldBitsAt = ng * 12u;
        for (uint32_t g = 0; g < ng; g++) {
          uint32_t* e = &tab[((size_t)ruleBase[k] + g) * 12u];
          memcpy(e, &h[(size_t)g * 12u], 48);
          if (e[0] & 2u) e[8] = bitsAt + (e[8] - oldBitsAt);        // RULE_BITS: where this object's bit arrays land
        }
        tab.insert(tab.end(), h.begin() + oldBitsAt, h.end());
      }
      HIP_TRY(hipMalloc(&bvh->d_rules, tab.size() * 4u));
      HIP_TRY(hipMemcpyAsync(bvh->d_rules, tab.data(), tab.size() * 4u, hipMemcpyHostToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));                            // (tab is a local)
    }
  }
  for (uint32_t k = 0; k < R; k++) { const uint32_t oi = objIndex[recObj[k]]; recs[k].root = nodeOfs[oi]; if (bvh->d_rules) recs[k].flags |= ruleBase[oi] << 8; }
  HIP_TRY(hipMalloc((void**)&dRecs, (size_t)R * sizeof(InstRec))); bvh->d_insts = dRecs;
  HIP_TRY(hipMemcpyAsync(dRecs, recs.data(), (size_t)R * sizeof(InstRec), hipMemcpyHostToDevice, st));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  bvh->root = 0;
  info.root_ref = 0; info.num_triangles = objTris; info.num_nodes = nNodes; info.num_leaves = top->info.num_leaves;
  info.bytes_nodes = nNodes * sizeof(CNode); info.bytes_triangles = nTris * sizeof(TriRec) + (uint64_t)R * sizeof(InstRec);
  info.sah = top->info.sah; info.build_ms = top->info.build_ms; info.top_levels = top->info.top_levels;
  info.depth = 2u * top->info.depth + maxDepth + 2u;             // a top level leaves up to two entries on a lane's stack (inner children, other instances)
  // kept for mi355_bvh_refit_instanced: the top tree, its box "mesh", the records, the objects behind them
  bvh->h_insts.assign((const uint8_t*)recs.data(), (const uint8_t*)recs.data() + recs.size() * sizeof(InstRec));
  bvh->instObjects.assign(recObj.begin(), recObj.end()); bvh->topHasOwn = hasOwn; if (hasOwn) bvh->instObjects[0] = nullptr;
  if (hasOwn) {                                                  // the own geometry's box stays on the host for the refits (kept in the top tree's otherwise unused record bytes:
    float ob[6]; for (int d = 0; d < 3; d++) { ob[d] = own->info.bounds_lower[d]; ob[3 + d] = own->info.bounds_upper[d]; }   // every refit read it back from the device)
    top->h_insts.assign((const uint8_t*)ob, (const uint8_t*)ob + sizeof(ob));
  }
  bvh->top = top; top = nullptr; bvh->d_topVerts = dfv; dfv = nullptr; bvh->d_topIdx = dfi; dfi = nullptr;
  guard.ok = true; *out = bvh;
  return 0;
}

// Instances that only moved (or changed their mask): the top tree keeps its topology and is refitted over the new world boxes (the reference: BVHNRefitT,
// kernels/bvh/bvh_refit.cpp, for the top level of its two-level scenes); the object trees behind it are not touched, nothing is concatenated again.
// Same records in the same order naming the same objects, else MI355_REFIT_IMPOSSIBLE (the caller builds anew).
static int refit_instanced_impl(Bvh* bvh, const mi355_instance* insts, uint32_t numInsts, hipStream_t st) {
  if (!bvh || !bvh->top || !bvh->d_insts || !bvh->d_topVerts) return MI355_REFIT_IMPOSSIBLE;
  HIP_TRY(hipSetDevice(bvh->device));
  const size_t R = bvh->h_insts.size() / sizeof(InstRec);
  std::vector<InstRec> recs(R); memcpy(recs.data(), bvh->h_insts.data(), R * sizeof(InstRec));
  std::vector<float> fv(R * 9 + 4, 0.0f);
  float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
  size_t k = 0;
  auto put_box = [&](size_t at, const float* lo, const float* hi) {
    for (int d = 0; d < 3; d++) { fv[at * 9 + d] = lo[d]; fv[at * 9 + 3 + d] = hi[d]; fv[at * 9 + 6 + d] = lo[d]; blo[d] = fminf(blo[d], lo[d]); bhi[d] = fmaxf(bhi[d], hi[d]); }
  };
  if (bvh->topHasOwn) {                                          // the scene's own geometry: record 0, where it was (box: what the build copied into the top tree's mesh)
    if (bvh->top->h_insts.size() != 6 * sizeof(float)) return MI355_REFIT_IMPOSSIBLE;
    float own[6]; memcpy(own, bvh->top->h_insts.data(), sizeof(own));
    put_box(0, own, own + 3); k = 1;
  }
  for (uint32_t i = 0; i < numInsts; i++) {
    Bvh* ob = (Bvh*)insts[i].object;
    if (!ob || ob->info.num_triangles == 0) continue;
    const float* l2w = insts[i].local2world;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int c = 0; c < 8; c++) {
      const float p[3] = {(c & 4) ? ob->info.bounds_upper[0] : ob->info.bounds_lower[0], (c & 2) ? ob->info.bounds_upper[1] : ob->info.bounds_lower[1], (c & 1) ? ob->info.bounds_upper[2] : ob->info.bounds_lower[2]};
      float q[3]; affine_point(l2w, p, q);
      for (int d = 0; d < 3; d++) { lo[d] = fminf(lo[d], q[d]); hi[d] = fmaxf(hi[d], q[d]); }
    }
    bool ok = true; for (int d = 0; d < 3; d++) ok = ok && lo[d] > -1.844E18f && hi[d] < 1.844E18f && lo[d] <= hi[d];
    if (!ok) continue;
    if (k >= R || bvh->instObjects[k] != (const void*)ob) return MI355_REFIT_IMPOSSIBLE;   // another object, another count: not a move
    affine_inverse(l2w, recs[k].w2l); recs[k].instID = insts[i].inst_id; recs[k].mask = insts[i].mask;
    put_box(k, lo, hi); k++;
  }
  if (k != R) return MI355_REFIT_IMPOSSIBLE;
  Bvh* top = bvh->top;
  if (!top->d_ids || top->info.num_triangles != R || top->info.num_nodes == 0) return MI355_REFIT_IMPOSSIBLE;   // (what refit_impl would refuse: nothing has been written yet)
  HIP_TRY(hipMemcpyAsync(bvh->d_topVerts, fv.data(), fv.size() * 4, hipMemcpyHostToDevice, st));
  mi355_mesh fake{}; fake.d_vertices = bvh->d_topVerts; fake.vertex_stride = 12; fake.num_vertices = (uint32_t)(3 * R); fake.d_indices = bvh->d_topIdx; fake.index_stride = 12;
  fake.num_triangles = (uint32_t)R; fake.geom_id = 0; fake.mask = 0xFFFFFFFFu;
  const int rc = refit_impl(top, &fake, 1, st);                  // (waits for its kernels; fv may go)
  if (rc != 0) return rc;
  HIP_TRY(hipMemcpyAsync(bvh->d_nodes, top->d_nodes, (size_t)top->info.num_nodes * sizeof(CNode), hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(bvh->d_tris, top->d_tris, (size_t)top->info.num_triangles * sizeof(TriRec), hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(bvh->d_insts, recs.data(), R * sizeof(InstRec), hipMemcpyHostToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  memcpy(bvh->h_insts.data(), recs.data(), R * sizeof(InstRec));
  for (int d = 0; d < 3; d++) { bvh->info.bounds_lower[d] = blo[d]; bvh->info.bounds_upper[d] = bhi[d]; }
  bvh->info.build_ms = top->info.build_ms; bvh->info.num_refits++;
  return 0;
}

}  // namespace mi355

extern "C" {

void mi355_default_build_params(mi355_build_params* p) {
  memset(p, 0, sizeof(*p));
  p->sah_block_shift = 0; p->min_leaf = 2; p->max_leaf = 3; p->small_threshold = 512; p->trav_cost = 1.0f; p->int_cost = 1.0f; p->split_factor = 1.2f; p->presplits = 0; p->top_splits = 1; p->top_split_min = 0; p->top_split_rel = 32.0f; p->top_split_cell = 1.0f / 8.0f;
}
const char* mi355_last_error(void) { return mi355::g_err.c_str(); }
// The sort of the Morton build on its own (build_sort.inl): n keys of 63 bits (device memory; bit 63 clear) -> the keys in order and, per place, the index the key came from;
// equal keys keep their index order.  tests/test_gpu_round6.py checks it against a stable argsort; *ms = time of the seven passes (HIP events), if asked for.
int mi355_sort_keys63(int device, const void* d_keys, void* d_keys_sorted, void* d_index_sorted, uint32_t n, float* ms) {
  using namespace mi355;
  HIP_TRY(hipSetDevice(device));
  if (ms) *ms = 0.0f;
  if (n == 0u) return 0;
  if (n >= (1u << 30)) return set_error(hipErrorInvalidValue, "mi355_sort_keys63: at most 2^30 - 1 keys");
  const uint32_t tiles = (n + RS_TILE - 1u) / RS_TILE;
  unsigned long long *tmpK = nullptr, *status = nullptr; uint32_t *tmpV = nullptr, *hist = nullptr;
  struct Free { void** p[4]; ~Free() { for (void** q : p) if (*q) hipFree(*q); } } fr{{(void**)&tmpK, (void**)&status, (void**)&tmpV, (void**)&hist}};
  HIP_TRY(hipMalloc((void**)&tmpK, (size_t)n * 8)); HIP_TRY(hipMalloc((void**)&tmpV, (size_t)n * 4));
  HIP_TRY(hipMalloc((void**)&status, (size_t)tiles * RS_RADIX * 8)); HIP_TRY(hipMalloc((void**)&hist, (size_t)RS_HIST_WORDS * 4));
  hipStream_t st = nullptr;
  hipEvent_t e0, e1; HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  struct EvGuard { hipEvent_t a, b; ~EvGuard() { hipEventDestroy(a); hipEventDestroy(b); } } evg{e0, e1};
  HIP_TRY(hipMemsetAsync(hist, 0, (size_t)RS_HIST_WORDS * 4, st));
  HIP_TRY(hipMemsetAsync(status, 0, (size_t)tiles * RS_RADIX * 8, st));
  hipLaunchKernelGGL(sort_hist0, dim3(tiles), dim3(256), 0, st, (const unsigned long long*)d_keys, n, hist);
  HIP_TRY(hipEventRecord(e0, st));
  // (seven passes: source -> sorted, sorted -> tmp, tmp -> sorted, ...: the caller's keys are never written)
  for (uint32_t pass = 0; pass < RS_PASSES; pass++) {
    const bool even = (pass & 1u) == 0u;
    const unsigned long long* kin = pass == 0u ? (const unsigned long long*)d_keys : (even ? tmpK : (const unsigned long long*)d_keys_sorted);
    const uint32_t* vin = pass == 0u ? nullptr : (even ? tmpV : (const uint32_t*)d_index_sorted);
    hipLaunchKernelGGL(radix_pass, dim3(tiles), dim3(RS_THREADS), 0, st, kin, vin, even ? (unsigned long long*)d_keys_sorted : tmpK, even ? (uint32_t*)d_index_sorted : tmpV,
                       n, pass, hist, status, hist + RS_PASSES * RS_RADIX);
  }
  HIP_TRY(hipEventRecord(e1, st));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventSynchronize(e1));
  if (ms) HIP_TRY(hipEventElapsedTime(ms, e0, e1));
  return 0;
}
int mi355_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
int mi355_device_name(int device, char* out, size_t n) {
  hipDeviceProp_t prop; HIP_TRY(hipGetDeviceProperties(&prop, device));
  snprintf(out, n, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount); return 0;
}
int mi355_bvh_build(int device, const mi355_mesh* meshes, uint32_t num_meshes, const mi355_build_params* params, void* stream, mi355_bvh_t* out) {
  mi355_build_params def; if (!params) { mi355_default_build_params(&def); params = &def; }
  mi355::Bvh* b = nullptr;
  const int rc = mi355::build_retry(device, meshes, num_meshes, params, (hipStream_t)stream, &b);
  *out = (mi355_bvh_t)b; return rc;
}
void mi355_bvh_destroy(mi355_bvh_t bvh) { delete (mi355::Bvh*)bvh; }
int mi355_bvh_build_instanced(int device, mi355_bvh_t own, const mi355_instance* instances, uint32_t num_instances, const mi355_build_params* params, void* stream, mi355_bvh_t* out) {
  mi355_build_params def; if (!params) { mi355_default_build_params(&def); params = &def; }
  mi355::Bvh* b = nullptr;
  const int rc = mi355::build_instanced_impl(device, (mi355::Bvh*)own, instances, num_instances, params, (hipStream_t)stream, &b);
  *out = (mi355_bvh_t)b; return rc;
}
int mi355_bvh_refit_instanced(mi355_bvh_t bvh, const mi355_instance* instances, uint32_t num_instances, void* stream) {
  return mi355::refit_instanced_impl((mi355::Bvh*)bvh, instances, num_instances, (hipStream_t)stream);
}
int mi355_bvh_refit(mi355_bvh_t bvh, const mi355_mesh* meshes, uint32_t num_meshes, void* stream) {
  if (!bvh) return MI355_REFIT_IMPOSSIBLE;
  return mi355::refit_impl((mi355::Bvh*)bvh, meshes, num_meshes, (hipStream_t)stream);
}
void mi355_release_build_scratch(int device) {
  Arena* a = arena_of(device);
  std::lock_guard<std::mutex> lk(a->mtx);
  hipSetDevice(device); a->release();
  std::lock_guard<std::mutex> lk2(g_spareMtx); a->release_spares();
}
int mi355_bvh_set_filter_rules(mi355_bvh_t bvh, const uint32_t* words, size_t num_words, uint32_t num_geoms) {
  mi355::Bvh* b = (mi355::Bvh*)bvh; if (!b) return mi355::set_error(hipErrorInvalidValue, "mi355_bvh_set_filter_rules: no tree");
  HIP_TRY(hipSetDevice(b->device));
  if (b->d_rules) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(b->d_rules)); b->d_rules = nullptr; }   // (queries in flight may still read the old table)
  b->h_rules.clear(); b->numRuleGeoms = num_geoms;
  if (!words || num_words == 0) return 0;
  if (num_words < (size_t)num_geoms * 12u) return mi355::set_error(hipErrorInvalidValue, "mi355_bvh_set_filter_rules: table shorter than 12 words per geometry");
  b->h_rules.assign(words, words + num_words);
  if (b->d_insts) return 0;                                      // (an instanced tree carries the combined table it was built with)
  HIP_TRY(hipMalloc(&b->d_rules, num_words * 4u));
  HIP_TRY(hipMemcpy(b->d_rules, words, num_words * 4u, hipMemcpyHostToDevice));
  return 0;
}
int mi355_bvh_get_info(mi355_bvh_t bvh, mi355_bvh_info* info) { *info = ((mi355::Bvh*)bvh)->info; return 0; }
int mi355_bvh_download(mi355_bvh_t bvh, void* nodes, size_t nb, void* tris, size_t tb) {
  mi355::Bvh* b = (mi355::Bvh*)bvh; HIP_TRY(hipSetDevice(b->device));
  if (nodes && nb) { if (nb > b->info.bytes_nodes) nb = b->info.bytes_nodes; if (nb) { const int rc = mi355_memcpy_d2h(nodes, b->d_nodes, nb); if (rc) return rc; } }
  if (tris && tb) { if (tb > b->info.bytes_triangles) tb = b->info.bytes_triangles; if (tb) { const int rc = mi355_memcpy_d2h(tris, b->d_tris, tb); if (rc) return rc; } }
  return 0;
}
int mi355_malloc_retry(int device, size_t bytes, void** d) {
  HIP_TRY(hipSetDevice(device));
  hipError_t e = hipMalloc(d, bytes ? bytes : 1);
  if (e == hipErrorOutOfMemory) {                               // the arena's spare tree arrays first, then once more
    (void)hipGetLastError();
    { std::lock_guard<std::mutex> lk(g_spareMtx); arena_of(device)->release_spares(); }
    e = hipMalloc(d, bytes ? bytes : 1);
  }
  if (e != hipSuccess) { *d = nullptr; return mi355::set_error(e, "hipMalloc"); }
  return 0;
}
int mi355_malloc(int device, size_t bytes, void** d) { HIP_TRY(hipSetDevice(device)); HIP_TRY(hipMalloc(d, bytes ? bytes : 1)); return 0; }
int mi355_free(void* d) { HIP_TRY(hipFree(d)); return 0; }
// (round 6) Blocking copies between pageable host memory and the device go through pinned staging of this library: the GPU never maps the caller's pages (a "userptr"
// mapping of ordinary heap memory faulted about once in ten runs of the GPU suite, in the middle of a copy or a query: rtcore_api.cpp, "host memory never meets the GPU")
namespace {
struct PinStage {
  static constexpr size_t PIECE = (size_t)8 << 20;
  std::mutex m; char* buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; hipStream_t st = nullptr;
};
int staged_copy(void* dst, const void* src, size_t n, bool toDevice) {
  if (n == 0) return 0;
  static std::mutex mapMtx; static std::map<int, PinStage*> stages;
  hipPointerAttribute_t at; int dev = 0;
  if (hipPointerGetAttributes(&at, toDevice ? dst : src) == hipSuccess) dev = at.device; else { (void)hipGetLastError(); HIP_TRY(hipGetDevice(&dev)); }
  int prev = 0; HIP_TRY(hipGetDevice(&prev));
  struct Back { int d; ~Back() { hipSetDevice(d); } } back{prev};
  HIP_TRY(hipSetDevice(dev));
  PinStage* ps;
  { std::lock_guard<std::mutex> lk(mapMtx); PinStage*& e = stages[dev]; if (!e) e = new PinStage; ps = e; }
  std::lock_guard<std::mutex> lk(ps->m);
  if (!ps->st) {
    HIP_TRY(hipStreamCreateWithFlags(&ps->st, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) { void* h = nullptr; HIP_TRY(hipHostMalloc(&h, PinStage::PIECE, hipHostMallocPortable)); ps->buf[k] = (char*)h; HIP_TRY(hipEventCreateWithFlags(&ps->ev[k], hipEventDisableTiming)); }
  }
  HIP_TRY(hipDeviceSynchronize());                              // (a blocking hipMemcpy is ordered behind the work queued before it)
  size_t c = 0;
  for (size_t ofs = 0; ofs < n; ofs += PinStage::PIECE, c++) {
    const size_t nb = n - ofs < PinStage::PIECE ? n - ofs : PinStage::PIECE; const int k = (int)(c & 1u);
    if (toDevice) {
      if (c >= 2) HIP_TRY(hipEventSynchronize(ps->ev[k]));
      memcpy(ps->buf[k], (const char*)src + ofs, nb);
      HIP_TRY(hipMemcpyAsync((char*)dst + ofs, ps->buf[k], nb, hipMemcpyHostToDevice, ps->st));
      HIP_TRY(hipEventRecord(ps->ev[k], ps->st));
    } else {
      HIP_TRY(hipMemcpyAsync(ps->buf[k], (const char*)src + ofs, nb, hipMemcpyDeviceToHost, ps->st));
      HIP_TRY(hipEventRecord(ps->ev[k], ps->st));
      if (c >= 1) { const size_t pofs = ofs - PinStage::PIECE; HIP_TRY(hipEventSynchronize(ps->ev[k ^ 1])); memcpy((char*)dst + pofs, ps->buf[k ^ 1], PinStage::PIECE); }
    }
  }
  if (toDevice) HIP_TRY(hipStreamSynchronize(ps->st));
  else { const size_t last = (c - 1) * PinStage::PIECE; HIP_TRY(hipEventSynchronize(ps->ev[(c - 1) & 1u])); memcpy((char*)dst + last, ps->buf[(c - 1) & 1u], n - last); }
  return 0;
}
}  // namespace
int mi355_memcpy_h2d(void* d, const void* h, size_t n) { return staged_copy(d, h, n, true); }
int mi355_memcpy_d2h(void* h, const void* d, size_t n) { return staged_copy(h, d, n, false); }
int mi355_synchronize(void* stream) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return 0; }
int mi355_device_synchronize(int device) { HIP_TRY(hipSetDevice(device)); HIP_TRY(hipDeviceSynchronize()); return 0; }
int mi355_memcpy_d2d_async(void* d, const void* s, size_t n, void* stream) { HIP_TRY(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, (hipStream_t)stream)); return 0; }
int mi355_stream_create(int device, void** stream) { HIP_TRY(hipSetDevice(device)); hipStream_t s; HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); *stream = (void*)s; return 0; }
int mi355_stream_destroy(void* stream) { HIP_TRY(hipStreamDestroy((hipStream_t)stream)); return 0; }
int mi355_event_create(void** e) {
  // timing events without the system-scope fence: a default event makes the kernel behind it start on flushed caches
  static const unsigned flags = getenv("MI355_EVENT_FLAGS") ? (unsigned)strtoul(getenv("MI355_EVENT_FLAGS"), nullptr, 0) : (unsigned)hipEventDisableSystemFence;
  hipEvent_t ev; HIP_TRY(hipEventCreateWithFlags(&ev, flags)); *e = (void*)ev; return 0;
}
int mi355_event_record(void* e, void* stream) { HIP_TRY(hipEventRecord((hipEvent_t)e, (hipStream_t)stream)); return 0; }
int mi355_event_elapsed_ms(void* a, void* b, float* ms) { HIP_TRY(hipEventSynchronize((hipEvent_t)b)); HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b)); return 0; }
int mi355_event_destroy(void* e) { HIP_TRY(hipEventDestroy((hipEvent_t)e)); return 0; }
}
