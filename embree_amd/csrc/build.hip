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
#include <hipcub/hipcub.hpp>     // DeviceRadixSort for the Morton build (a plain library sort; everything else here is hand-written)
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

constexpr uint32_t NIL = 0xFFFFFFFFu;
constexpr int NBINS = 32;                       // NUM_OBJECT_BINS, kernels/builders/bvh_builder_sah.h:10
constexpr int BINW = 7;                         // lo.xyz, hi.xyz (ordered uint), count
constexpr int BINS_WORDS = 3 * NBINS * BINW;    // 672 words = 2688 B per segment
#ifndef MI355_CHUNK
#define MI355_CHUNK 2048
#endif
constexpr uint32_t CHUNK = MI355_CHUNK;         // triangles per top-phase workgroup
constexpr int CHUNK_ROUNDS = CHUNK / 256;       // triangles per thread of top_partition
constexpr uint32_t ENC_POS_INF = 0xFF800000u;   // enc(+inf)
constexpr uint32_t ENC_NEG_INF = 0x007FFFFFu;   // enc(-inf)

struct PrimRef { float lo[3]; uint32_t geom; float hi[3]; uint32_t prim; };   // kernels/builders/primref.h:11-107 (geom = table index)
struct GeomDesc { const char* verts; const char* idx; uint32_t vstride, istride, nv, nt, geomID, mask, primOffset, quad; };   // nt = internal triangles (2 per quad)
// internal triangle j of a geometry -> its three vertex indices; quads: j>>1 = quad, odd j = second half (v2,v1,v3), even = (v0,v1,v3)
__device__ __forceinline__ void prim_indices(const GeomDesc& g, uint32_t j, uint32_t& i0, uint32_t& i1, uint32_t& i2, uint32_t& id) {
  if (g.quad) {
    const uint32_t* q = (const uint32_t*)(g.idx + (size_t)(j >> 1) * g.istride);
    i0 = (j & 1u) ? q[2] : q[0]; i1 = q[1]; i2 = q[3]; id = (j >> 1) | ((j & 1u) << 31);
    if (q[0] >= g.nv || q[2] >= g.nv) i0 = 0xFFFFFFFFu;           // a quad with ANY invalid index is skipped as a whole (QuadMesh::buildBounds)
  } else {
    const uint32_t* t = (const uint32_t*)(g.idx + (size_t)j * g.istride);
    i0 = t[0]; i1 = t[1]; i2 = t[2]; id = j;
  }
}
struct BNode { float lo[3]; uint32_t begin; float hi[3]; uint32_t end; uint32_t left, right; float splitSah; uint32_t pad; };
struct Seg {
  uint32_t begin, end, bnode, flags;            // flags bit0: fallback (median) split
  float cmin[3]; uint32_t dim;
  float cmax[3]; uint32_t pos;
  float ofs[3]; uint32_t nb;
  float scale[3]; uint32_t nL;
  uint32_t childL, childR, curL, curR;
  uint32_t acc[2][12];                          // per side: centroid lo/hi (6) + geometry lo/hi (6), ordered uint
};
struct SmallEntry { uint32_t begin, end, bnode, buf; float cmin[3], cmax[3]; };
struct Chunk { uint32_t seg, begin, end; };
struct WideItem { uint32_t bnode, node; };
struct Counters {
  uint32_t numPrims, numBLeaves, numSegsNext, numChunks, numSmall, numWide, numWideNext, numLeaves;
  uint32_t bounds[12];                          // scene geom lo/hi + centroid lo/hi (ordered uint)
  uint32_t overflow, rootRef, numTrisOut, numInvalid;
  uint32_t numSegs, topLevels, wideDepth, lvlNodeBase, lvlTriBase, wideCount[2];      // level loops are driven from the device: no host readback per level
  unsigned long long sahFixed;                    // SAH statistics, 2^-24 fixed point (order-independent sum)
  uint32_t lvlStart[64];                          // first node of every level of the wide tree (numbering is breadth first): what a refit walks bottom-up
};
struct Params { uint32_t shift, minLeaf, maxLeaf, small; float travCost, intCost; uint32_t quality; };

// order-preserving float <-> uint so that integer atomicMin/Max reduce floats exactly
__device__ __forceinline__ uint32_t enc(float f) { uint32_t u = __float_as_uint(f); return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__device__ __forceinline__ float dec(uint32_t u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }
__device__ __forceinline__ float half_area3(float dx, float dy, float dz) { return fmaf(dx, dy + dz, dy * dz); }  // common/math/vec3fa.h:349
__device__ __forceinline__ float sel3(uint32_t d, float a, float b, float c) { return d == 0u ? a : (d == 1u ? b : c); }   // no dynamically indexed register arrays (scratch)
__device__ __forceinline__ bool valid_f(float x) { return x > -1.844E18f && x < 1.844E18f; }  // isvalid, FLT_LARGE constants.h:21

__device__ __forceinline__ PrimRef load_prim(const PrimRef* p) {
  const float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
  PrimRef r; r.lo[0] = a.x; r.lo[1] = a.y; r.lo[2] = a.z; r.geom = __float_as_uint(a.w);
  r.hi[0] = b.x; r.hi[1] = b.y; r.hi[2] = b.z; r.prim = __float_as_uint(b.w); return r;
}
__device__ __forceinline__ void store_prim(PrimRef* p, const PrimRef& r) {
  ((float4*)p)[0] = make_float4(r.lo[0], r.lo[1], r.lo[2], __uint_as_float(r.geom));
  ((float4*)p)[1] = make_float4(r.hi[0], r.hi[1], r.hi[2], __uint_as_float(r.prim));
}

// ---- wave64 reductions on ordered-uint codes (DPP: quad_perm, row_shr:4/8, row_bcast:15/31); the result is valid in lane 63.
// LDS/L2 atomics of a wave that all hit the same word are executed one lane after the other (measured: ~1 lane-atomic per
// clock per CU on mesh-ordered input, where neighbouring triangles fall into the same bin), so the lanes are combined
// in registers first and one lane issues the atomic.
template <int CTRL, int ROWMASK> __device__ __forceinline__ uint32_t dpp_u(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROWMASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_umin63(uint32_t v) {
  v = min(v, dpp_u<0xB1, 0xF>(v, v)); v = min(v, dpp_u<0x4E, 0xF>(v, v)); v = min(v, dpp_u<0x114, 0xF>(v, v));
  v = min(v, dpp_u<0x118, 0xF>(v, v)); v = min(v, dpp_u<0x142, 0xA>(v, v)); v = min(v, dpp_u<0x143, 0xC>(v, v));
  return v;
}
__device__ __forceinline__ uint32_t wave_umax63(uint32_t v) {
  v = max(v, dpp_u<0xB1, 0xF>(v, v)); v = max(v, dpp_u<0x4E, 0xF>(v, v)); v = max(v, dpp_u<0x114, 0xF>(v, v));
  v = max(v, dpp_u<0x118, 0xF>(v, v)); v = max(v, dpp_u<0x142, 0xA>(v, v)); v = max(v, dpp_u<0x143, 0xC>(v, v));
  return v;
}

// ---------------------------------------------------------------------------------- K1 primref_gen
// Triangle p of the concatenated geometries lands at out[p]: no compaction counter (a returning global atomic per
// block costs ~11 ns each, 0.2 ms for a 4.8 M triangle scene, and makes the order depend on block timing).  Invalid
// triangles (index out of range, non-finite or huge coordinate) are marked geom = NIL and counted; only if there are
// any does primref_compact squeeze them out afterwards (stable, so the order is still the input order).
__global__ __launch_bounds__(256) void primref_gen(const GeomDesc* geoms, uint32_t numGeoms, uint32_t totalPrims,
                                                   PrimRef* out, Counters* ctr) {
  __shared__ uint32_t s_acc[12];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  if (tid < 12) s_acc[tid] = (tid % 6 < 3) ? ENC_POS_INF : ENC_NEG_INF;
  __syncthreads();
  uint32_t acc[12]; for (int k = 0; k < 12; k++) acc[k] = (k % 6 < 3) ? 0xFFFFFFFFu : 0u;
  uint32_t gi = 0, nInvalid = 0; GeomDesc g = geoms[0];
  for (uint32_t p = blockIdx.x * 256u + tid; p < totalPrims; p += gridDim.x * 256u) {
    if (p - g.primOffset >= g.nt) {                             // not in the cached geometry: last geometry with primOffset <= p
      uint32_t lo = 0, hi = numGeoms - 1;
      while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (geoms[mid].primOffset <= p) lo = mid; else hi = mid - 1; }
      gi = lo; g = geoms[lo];
    }
    const uint32_t j = p - g.primOffset;
    uint32_t i0, i1, i2, pid;
    prim_indices(g, j, i0, i1, i2, pid);
    bool ok = false; PrimRef r{};
    if (i0 < g.nv && i1 < g.nv && i2 < g.nv) {
      const float* a = (const float*)(g.verts + (size_t)i0 * g.vstride);
      const float* b = (const float*)(g.verts + (size_t)i1 * g.vstride);
      const float* c = (const float*)(g.verts + (size_t)i2 * g.vstride);
      ok = true;
      for (int d = 0; d < 3; d++) {
        const float x = a[d], y = b[d], z = c[d];
        ok = ok && valid_f(x) && valid_f(y) && valid_f(z);
        r.lo[d] = fminf(fminf(x, y), z); r.hi[d] = fmaxf(fmaxf(x, y), z);
      }
    }
    if (ok && g.quad) {                                       // non-finite fourth vertex: the reference drops the whole quad
      const uint32_t* q = (const uint32_t*)(g.idx + (size_t)(j >> 1) * g.istride);
      const float* o4 = (const float*)(g.verts + (size_t)((j & 1u) ? q[0] : q[2]) * g.vstride);
      ok = valid_f(o4[0]) && valid_f(o4[1]) && valid_f(o4[2]);
    }
    r.geom = ok ? gi : NIL; r.prim = j;
    store_prim(out + p, r);
    if (ok) {
      for (int d = 0; d < 3; d++) {
        const uint32_t l = enc(r.lo[d]), h = enc(r.hi[d]), c2 = enc(r.lo[d] + r.hi[d]);   // centroid proxy = lower+upper, never halved (priminfo.h:46-52)
        acc[d] = min(acc[d], l); acc[3 + d] = max(acc[3 + d], h); acc[6 + d] = min(acc[6 + d], c2); acc[9 + d] = max(acc[9 + d], c2);
      }
    } else nInvalid++;
  }
  for (int k = 0; k < 12; k++) {
    const uint32_t x = (k % 6 < 3) ? wave_umin63(acc[k]) : wave_umax63(acc[k]);
    if (lane == 63u) { if (k % 6 < 3) atomicMin(&s_acc[k], x); else atomicMax(&s_acc[k], x); }
  }
  const unsigned long long bad = __ballot(nInvalid != 0u);
  if (bad != 0ull && nInvalid) atomicAdd(&ctr->numInvalid, nInvalid);
  __syncthreads();
  if (tid < 12) { if (tid % 6 < 3) atomicMin(&ctr->bounds[tid], s_acc[tid]); else atomicMax(&ctr->bounds[tid], s_acc[tid]); }
}

// rare path: stable compaction of the valid PrimRefs (tile = 256 consecutive entries)
__global__ __launch_bounds__(256) void compact_count(const PrimRef* in, uint32_t n, uint32_t* tileCount) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  const bool ok = p < n && in[p].geom != NIL;
  const int c = __syncthreads_count(ok);
  if (threadIdx.x == 0) tileCount[blockIdx.x] = (uint32_t)c;
}
__global__ __launch_bounds__(1024) void compact_scan(uint32_t* tileCount, uint32_t numTiles, Counters* ctr) {
  __shared__ uint32_t s_part[1024];
  const uint32_t tid = threadIdx.x, per = (numTiles + 1023u) / 1024u, b = tid * per, e = min(b + per, numTiles);
  uint32_t sum = 0; for (uint32_t i = b; i < e; i++) sum += tileCount[i];
  s_part[tid] = sum; __syncthreads();
  if (tid == 0) { uint32_t run = 0; for (int i = 0; i < 1024; i++) { const uint32_t t = s_part[i]; s_part[i] = run; run += t; } ctr->numPrims = run; }
  __syncthreads();
  uint32_t run = s_part[tid]; for (uint32_t i = b; i < e; i++) { const uint32_t t = tileCount[i]; tileCount[i] = run; run += t; }
}
__global__ __launch_bounds__(256) void compact_scatter(const PrimRef* in, uint32_t n, const uint32_t* tileOfs, PrimRef* out) {
  __shared__ uint32_t s_w[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, p = blockIdx.x * 256u + tid;
  PrimRef r{}; bool ok = false;
  if (p < n) { r = load_prim(in + p); ok = r.geom != NIL; }
  const unsigned long long m = __ballot(ok);
  if (lane == 0) s_w[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t off = tileOfs[blockIdx.x]; for (uint32_t w = 0; w < wave; w++) off += s_w[w];
  if (ok) store_prim(out + off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), r);
}

// --------------------------------------------------------------------------------- presplit (RTC_BUILD_QUALITY_HIGH)
// The reference's high-quality builder in its "presplits" form (BVHNBuilderFastSpatialSAH with usePreSplits, kernels/bvh/bvh_builder_sah_spatial.cpp:93-125):
// before the ordinary binned-SAH build, triangles whose box is much larger than the triangle are cut along the planes of a 1024^3 grid over the scene
// (kernels/builders/primrefgen_presplit.h): every piece is a PrimRef of the SAME triangle with the box of the clipped piece.  The tree gets tighter boxes
// where long or diagonal triangles used to blow them up; the leaves may name a triangle several times (each reference becomes a leaf record).
//   priority(ref) = sqrt(sqrt((area(box) - projected area(triangle)) * 1.5^(highest differing Morton bit)))           :125-143
//   pieces(ref)   = 2^clamp(ceil(log2(budget * priority / sum of priorities)), 1, 5), 1 if that ratio is < 1                :296-313
//   a piece is cut at the grid plane of the highest Morton bit in which its corners differ (SplittingGrid::split_pos :39-78), the triangle is
//   clipped edge by edge (splitPolygon, kernels/builders/splitter.h:16-49) and the piece's box is the clipped box intersected with the parent's.
// The budget is max_spatial_split_replications - 1 = 20 % extra references (kernels/common/state.cpp:87).  Where the reference sorts the candidates
// and drops the lowest ones when the pieces exceed the budget, this build halves the budget and counts again (at most 6 times): no sort.
struct SplitGrid { float base[3]; float scale, extend; };
__device__ __forceinline__ uint32_t part1by2(uint32_t x) { x &= 0x3FFu; x = (x | (x << 16)) & 0x030000FFu; x = (x | (x << 8)) & 0x0300F00Fu; x = (x | (x << 4)) & 0x030C30C3u; x = (x | (x << 2)) & 0x09249249u; return x; }
__device__ __forceinline__ void grid_codes(const SplitGrid& g, const float* lo, const float* hi, int (&iu)[3], uint32_t& lc, uint32_t& uc) {
  int il[3];
  for (int d = 0; d < 3; d++) {
    const float gl = (lo[d] - g.base[d]) * g.scale + 0.2f, gu = (hi[d] - g.base[d]) * g.scale - 0.2f;
    il[d] = (int)floorf(gl); iu[d] = (int)floorf(gu);
    if ((int)rintf(gl) >= (int)rintf(gu)) iu[d] = il[d];          // "this ignores dimensions that are empty"
  }
  lc = part1by2((uint32_t)il[0]) | (part1by2((uint32_t)il[1]) << 1) | (part1by2((uint32_t)il[2]) << 2);
  uc = part1by2((uint32_t)iu[0]) | (part1by2((uint32_t)iu[1]) << 1) | (part1by2((uint32_t)iu[2]) << 2);
}
__device__ __forceinline__ void load_tri(const GeomDesc* geoms, const PrimRef& r, float (&v)[3][3]) {
  const GeomDesc g = geoms[r.geom];
  uint32_t i0, i1, i2, pid; prim_indices(g, r.prim, i0, i1, i2, pid);
  const float* a = (const float*)(g.verts + (size_t)i0 * g.vstride); const float* b = (const float*)(g.verts + (size_t)i1 * g.vstride); const float* c = (const float*)(g.verts + (size_t)i2 * g.vstride);
  for (int d = 0; d < 3; d++) { v[0][d] = a[d]; v[1][d] = b[d]; v[2][d] = c[d]; }
}
struct Piece { float lo[3], hi[3]; uint32_t want; };
// splitPrimitive (primrefgen_presplit.h:144-181) without recursion: pieces come out in the reference's order (left before right)
template <bool EMIT>
__device__ uint32_t presplit_walk(const PrimRef& ref, uint32_t want, const float (&v)[3][3], const SplitGrid& g, PrimRef* first, PrimRef* rest) {
  Piece stack[7]; int sp = 0;
  Piece p0; for (int d = 0; d < 3; d++) { p0.lo[d] = ref.lo[d]; p0.hi[d] = ref.hi[d]; } p0.want = want;
  stack[sp++] = p0;
  uint32_t num = 0;
  while (sp > 0) {
    const Piece cur = stack[--sp];
    bool leaf = cur.want <= 1u; uint32_t dim = 0; float pos = 0.0f;
    if (!leaf) {
      int iu[3]; uint32_t lc, uc; grid_codes(g, cur.lo, cur.hi, iu, lc, uc);
      if (lc == uc) leaf = true;
      else {
        const uint32_t diff = 31u - (uint32_t)__clz((int)(lc ^ uc)), level = diff / 3u; dim = diff % 3u;
        const int isplit = (dim == 0u ? iu[0] : dim == 1u ? iu[1] : iu[2]) & ~((1 << level) - 1);
        pos = sel3(dim, g.base[0], g.base[1], g.base[2]) + (float)isplit * (1.0f / 1024.0f) * g.extend;
      }
    }
    if (leaf || sp + 2 > 7) {
      if (EMIT) { PrimRef o = ref; for (int d = 0; d < 3; d++) { o.lo[d] = cur.lo[d]; o.hi[d] = cur.hi[d]; } store_prim(num == 0u ? first : rest + (num - 1u), o); }
      num++; continue;
    }
    Piece L, R;
    for (int d = 0; d < 3; d++) { L.lo[d] = __builtin_inff(); L.hi[d] = -__builtin_inff(); R.lo[d] = __builtin_inff(); R.hi[d] = -__builtin_inff(); }
    for (int e = 0; e < 3; e++) {                                  // splitPolygon<3>: every edge (v[e], v[e+1])
      const int e1 = e == 2 ? 0 : e + 1;
      const float a0 = sel3(dim, v[e][0], v[e][1], v[e][2]), a1 = sel3(dim, v[e1][0], v[e1][1], v[e1][2]);
      if (a0 <= pos) for (int d = 0; d < 3; d++) { L.lo[d] = fminf(L.lo[d], v[e][d]); L.hi[d] = fmaxf(L.hi[d], v[e][d]); }
      if (a0 >= pos) for (int d = 0; d < 3; d++) { R.lo[d] = fminf(R.lo[d], v[e][d]); R.hi[d] = fmaxf(R.hi[d], v[e][d]); }
      if ((a0 < pos && pos < a1) || (a1 < pos && pos < a0)) {
        const float t = (pos - a0) * (1.0f / (a1 - a0));
        for (int d = 0; d < 3; d++) { const float c = fmaf(t, v[e1][d] - v[e][d], v[e][d]); L.lo[d] = fminf(L.lo[d], c); L.hi[d] = fmaxf(L.hi[d], c); R.lo[d] = fminf(R.lo[d], c); R.hi[d] = fmaxf(R.hi[d], c); }
      }
    }
    bool okL = true, okR = true;
    for (int d = 0; d < 3; d++) {                                  // intersect with the piece that is being split
      L.lo[d] = fmaxf(L.lo[d], cur.lo[d]); L.hi[d] = fminf(L.hi[d], cur.hi[d]); R.lo[d] = fmaxf(R.lo[d], cur.lo[d]); R.hi[d] = fminf(R.hi[d], cur.hi[d]);
      okL = okL && L.lo[d] <= L.hi[d]; okR = okR && R.lo[d] <= R.hi[d];
    }
    if (!okL || !okR) {                                            // (the reference asserts this away) keep the piece whole
      if (EMIT) { PrimRef o = ref; for (int d = 0; d < 3; d++) { o.lo[d] = cur.lo[d]; o.hi[d] = cur.hi[d]; } store_prim(num == 0u ? first : rest + (num - 1u), o); }
      num++; continue;
    }
    L.want = cur.want / 2u; R.want = cur.want - L.want;
    stack[sp++] = R; stack[sp++] = L;
  }
  return num;
}
__global__ __launch_bounds__(256) void presplit_priority(const PrimRef* prims, uint32_t n, const GeomDesc* geoms, SplitGrid g, float* prio, float* partial) {
  __shared__ float s_w[4];
  const uint32_t tid = threadIdx.x, i = blockIdx.x * 256u + tid;
  float p = 0.0f;
  if (i < n) {
    const PrimRef r = load_prim(prims + i);
    int iu[3]; uint32_t lc, uc; grid_codes(g, r.lo, r.hi, iu, lc, uc);
    if (lc != uc) {
      float v[3][3]; load_tri(geoms, r, v);
      const float e0[3] = {v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]}, e1[3] = {v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2]};
      const float cx = fmaf(e0[1], e1[2], -(e0[2] * e1[1])), cy = fmaf(e0[2], e1[0], -(e0[0] * e1[2])), cz = fmaf(e0[0], e1[1], -(e0[1] * e1[0]));
      const float areaPrim = fabsf(cx) + fabsf(cy) + fabsf(cz);        // areaProjectedTriangle, kernels/builders/priminfo.h:11-17
      const float areaBox = 2.0f * half_area3(r.hi[0] - r.lo[0], r.hi[1] - r.lo[1], r.hi[2] - r.lo[2]);
      if (areaPrim != 0.0f) {
        const int diff = 31 - __clz((int)(lc ^ uc));
        p = sqrtf(sqrtf(fmaxf(0.0f, areaBox - areaPrim) * powf(1.5f, (float)diff)));
        if (!(p >= 0.0f && p < 1.844E18f)) p = 0.0f;
      }
    }
    prio[i] = p;
  }
  // block sum in a fixed order (the reference's sum is "undeterministic", :289; this one is not)
  for (int o = 32; o > 0; o >>= 1) p += __shfl_down(p, o, 64);
  if ((tid & 63u) == 0u) s_w[tid >> 6] = p;
  __syncthreads();
  if (tid == 0u) partial[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}
__global__ __launch_bounds__(1024) void presplit_sum(const float* partial, uint32_t nb, float* out) {
  __shared__ float s_p[1024];
  const uint32_t tid = threadIdx.x, per = (nb + 1023u) / 1024u, b = min(tid * per, nb), e = min(b + per, nb);
  float sum = 0.0f; for (uint32_t i = b; i < e; i++) sum += partial[i];
  s_p[tid] = sum; __syncthreads();
  for (uint32_t o = 512u; o > 0u; o >>= 1) { if (tid < o) s_p[tid] += s_p[tid + o]; __syncthreads(); }
  if (tid == 0u) out[0] = s_p[0];
}
// pieces per reference (cnt = pieces - 1 = extra references), tile sums for the scan
__global__ __launch_bounds__(256) void presplit_count(const PrimRef* prims, uint32_t n, const GeomDesc* geoms, SplitGrid g, const float* prio, const float* psum, float budget,
                                                      uint32_t* cnt, uint32_t* tileSum) {
  __shared__ uint32_t s_w[4];
  const uint32_t tid = threadIdx.x, i = blockIdx.x * 256u + tid;
  uint32_t extra = 0u;
  if (i < n) {
    const float p = prio[i], inv = psum[0] > 0.0f ? 1.0f / psum[0] : 1.0f;
    uint32_t want = 1u;
    if (p > 0.0f) {
      const float rel = budget * p * inv;
      if (rel >= 1.0f) { const float l = fmaxf(fminf(ceilf(logf(rel) / logf(2.0f)), 5.0f), 1.0f); want = 1u << (uint32_t)l; }
    }
    if (want > 1u) {
      const PrimRef r = load_prim(prims + i);
      float v[3][3]; load_tri(geoms, r, v);
      extra = presplit_walk<false>(r, want, v, g, nullptr, nullptr) - 1u;
    }
    cnt[i] = extra | (want << 16);
  }
  uint32_t x = extra; for (int o = 32; o > 0; o >>= 1) x += (uint32_t)__shfl_down((int)x, o, 64);
  if ((tid & 63u) == 0u) s_w[tid >> 6] = x;
  __syncthreads();
  if (tid == 0u) tileSum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(1024) void presplit_scan(uint32_t* tileSum, uint32_t numTiles, uint32_t* total) {
  __shared__ uint32_t s_part[1024];
  const uint32_t tid = threadIdx.x, per = (numTiles + 1023u) / 1024u, b = min(tid * per, numTiles), e = min(b + per, numTiles);
  uint32_t sum = 0; for (uint32_t i = b; i < e; i++) sum += tileSum[i];
  s_part[tid] = sum; __syncthreads();
  if (tid == 0) { uint32_t run = 0; for (int i = 0; i < 1024; i++) { const uint32_t t = s_part[i]; s_part[i] = run; run += t; } total[0] = run; }
  __syncthreads();
  uint32_t run = s_part[tid]; for (uint32_t i = b; i < e; i++) { const uint32_t t = tileSum[i]; tileSum[i] = run; run += t; }
}
// piece 0 replaces the reference, the others go behind the n original references at the scanned offset
__global__ __launch_bounds__(256) void presplit_emit(PrimRef* prims, uint32_t n, const GeomDesc* geoms, SplitGrid g, const uint32_t* cnt, const uint32_t* tileOfs) {
  __shared__ uint32_t s_scan[256];
  const uint32_t tid = threadIdx.x, i = blockIdx.x * 256u + tid;
  const uint32_t c = i < n ? cnt[i] : 0u, extra = c & 0xFFFFu, want = c >> 16;
  s_scan[tid] = extra; __syncthreads();
  for (uint32_t o = 1; o < 256u; o <<= 1) { uint32_t x = 0; if (tid >= o) x = s_scan[tid - o]; __syncthreads(); s_scan[tid] += x; __syncthreads(); }
  if (extra == 0u) return;
  const uint32_t off = n + tileOfs[blockIdx.x] + s_scan[tid] - extra;
  const PrimRef r = load_prim(prims + i);
  float v[3][3]; load_tri(geoms, r, v);
  presplit_walk<true>(r, want, v, g, prims + i, prims + off);
}
__global__ __launch_bounds__(256) void centroid_bounds(const PrimRef* prims, uint32_t n, Counters* ctr) {
  __shared__ uint32_t s_acc[6];                                  // one global atomic per block and word: same-address atomics from every wave cost 0.4 ms here
  if (threadIdx.x < 6u) s_acc[threadIdx.x] = threadIdx.x < 3u ? 0xFFFFFFFFu : 0u;
  __syncthreads();
  uint32_t acc[6]; for (int k = 0; k < 6; k++) acc[k] = k < 3 ? 0xFFFFFFFFu : 0u;
  for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < n; p += gridDim.x * 256u) {
    const PrimRef r = load_prim(prims + p);
    for (int d = 0; d < 3; d++) { const uint32_t c2 = enc(r.lo[d] + r.hi[d]); acc[d] = min(acc[d], c2); acc[3 + d] = max(acc[3 + d], c2); }
  }
  for (int k = 0; k < 6; k++) {
    const uint32_t x = k < 3 ? wave_umin63(acc[k]) : wave_umax63(acc[k]);
    if ((threadIdx.x & 63u) == 63u) { if (k < 3) atomicMin(&s_acc[k], x); else atomicMax(&s_acc[k], x); }
  }
  __syncthreads();
  if (threadIdx.x < 6u) { if (threadIdx.x < 3u) atomicMin(&ctr->bounds[6 + threadIdx.x], s_acc[threadIdx.x]); else atomicMax(&ctr->bounds[6 + threadIdx.x], s_acc[threadIdx.x]); }
}

// -------------------------------------------------------------------------------- binning helpers
struct Mapping { float ofs[3], scale[3]; uint32_t nb; };
// BinMapping(pinfo): num = min(32, 4 + 0.05 n), scale = 0.99 num / diag (0 if diag <= 1e-34)  heuristic_binning.h:46-55
__device__ __forceinline__ Mapping make_mapping(uint32_t n, const float* cmin, const float* cmax) {
  Mapping m; const uint32_t num = (uint32_t)(4.0f + 0.05f * (float)n); m.nb = num < (uint32_t)NBINS ? num : (uint32_t)NBINS;
  for (int d = 0; d < 3; d++) {
    const float diag = fmaxf(1E-34f, cmax[d] - cmin[d]);
    m.scale[d] = diag > 1E-34f ? (0.99f * (float)m.nb) / diag : 0.0f;
    m.ofs[d] = cmin[d];
  }
  return m;
}
__device__ __forceinline__ int bin_unsafe(float c2, float ofs, float scale) { return (int)floorf((c2 - ofs) * scale); }
__device__ __forceinline__ int bin_clamped(float c2, float ofs, float scale, uint32_t nb) {
  int i = bin_unsafe(c2, ofs, scale); i = i < 0 ? 0 : i; return i > (int)nb - 1 ? (int)nb - 1 : i;
}
__device__ __forceinline__ void bins_clear(uint32_t* bins, uint32_t tid, uint32_t nthreads) {
  for (uint32_t w = tid; w < (uint32_t)BINS_WORDS; w += nthreads) { const uint32_t k = w % BINW; bins[w] = k < 3 ? ENC_POS_INF : (k < 6 ? ENC_NEG_INF : 0u); }
}

__device__ __forceinline__ uint32_t wave_uadd63(uint32_t v) {
  v += dpp_u<0xB1, 0xF>(0u, v); v += dpp_u<0x4E, 0xF>(0u, v); v += dpp_u<0x114, 0xF>(0u, v);
  v += dpp_u<0x118, 0xF>(0u, v); v += dpp_u<0x142, 0xA>(0u, v); v += dpp_u<0x143, 0xC>(0u, v);
  return v;
}

// BinInfoT::bin (heuristic_binning.h:210-257) with run merging in front of the bins.  Same-word LDS atomics are what the binning
// kernels wait for (PMC: SQ_WAIT_INST_LDS = 73 % of top_bin's wave cycles with one atomic per triangle, profiles/r01_build_history.md).
// A batch of 64 consecutive triangles of a mesh almost always falls into ONE bin per axis, so the wave keeps a current bin per axis
// (uniform) and every lane a private partial (bounds + count) of it in registers; only when the wave's bin changes, or at the end
// of the span, the partials are folded across the wave with DPP and lane 63 issues the run's 7 atomics.  Lanes of a batch that
// straddles bins and are not in the wave's bin go to the LDS bins directly.
struct BinRuns { int b[3]; uint32_t lo[3][3], hi[3][3], n[3]; };   // b: the WAVE's current bin per axis (uniform); the rest: this lane's partial of that bin
__device__ __forceinline__ void runs_init(BinRuns& r) { for (int d = 0; d < 3; d++) { r.b[d] = -1; r.n[d] = 0u; for (int k = 0; k < 3; k++) { r.lo[d][k] = 0xFFFFFFFFu; r.hi[d][k] = 0u; } } }
// fold the lanes' partials of axis d across the wave (DPP), lane 63 issues the 7 atomics of the run
__device__ __forceinline__ void runs_flush_axis(BinRuns& r, int d, uint32_t* bins, uint32_t lane) {
  if (r.b[d] >= 0) {                                            // wave-uniform
    uint32_t v[6];
    for (int k = 0; k < 3; k++) { v[k] = wave_umin63(r.lo[d][k]); v[3 + k] = wave_umax63(r.hi[d][k]); }
    const uint32_t cnt = wave_uadd63(r.n[d]);
    if (lane == 63u && cnt) {
      uint32_t* e = bins + (d * NBINS + r.b[d]) * BINW;
      atomicMin(&e[0], v[0]); atomicMin(&e[1], v[1]); atomicMin(&e[2], v[2]);
      atomicMax(&e[3], v[3]); atomicMax(&e[4], v[4]); atomicMax(&e[5], v[5]);
      atomicAdd(&e[6], cnt);
    }
  }
  r.n[d] = 0u; for (int k = 0; k < 3; k++) { r.lo[d][k] = 0xFFFFFFFFu; r.hi[d][k] = 0u; }
}
// every lane of the wave calls it with its triangle of the batch (valid = holds one)
__device__ __forceinline__ void runs_add(BinRuns& r, uint32_t* bins, const Mapping& m, const PrimRef& p, bool valid, uint32_t lane) {
  const unsigned long long vm = __ballot(valid);
  if (vm == 0ull) return;
  const int first = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(vm));
  uint32_t c[6];
  for (int k = 0; k < 3; k++) { c[k] = enc(p.lo[k]); c[3 + k] = enc(p.hi[k]); }
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const int b = valid ? bin_clamped(p.lo[d] + p.hi[d], m.ofs[d], m.scale[d], m.nb) : -1;
    const int b0 = __builtin_amdgcn_readlane(b, first);
    const bool uniform = __ballot(valid && b == b0) == vm;      // the usual case: 64 consecutive triangles, one bin
    if (uniform && b0 != r.b[d]) { runs_flush_axis(r, d, bins, lane); r.b[d] = b0; }
    if (valid) {
      if (b == r.b[d]) {                                        // into my partial of the wave's bin: registers only
        r.n[d]++;
        for (int k = 0; k < 3; k++) { r.lo[d][k] = min(r.lo[d][k], c[k]); r.hi[d][k] = max(r.hi[d][k], c[3 + k]); }
      } else {                                                  // the batch straddles bins: this lane goes to the LDS bins directly
        uint32_t* e = bins + (d * NBINS + b) * BINW;
        atomicMin(&e[0], c[0]); atomicMin(&e[1], c[1]); atomicMin(&e[2], c[2]);
        atomicMax(&e[3], c[3]); atomicMax(&e[4], c[4]); atomicMax(&e[5], c[5]);
        atomicAdd(&e[6], 1u);
      }
    }
  }
}
__device__ __forceinline__ void runs_flush_wave(BinRuns& r, uint32_t* bins, uint32_t lane) {
  for (int d = 0; d < 3; d++) runs_flush_axis(r, d, bins, lane);
}

// Lane-private variant for top_bin: a lane sees triangles i, i + 64, ... of its wave's span and keeps its OWN current bin per axis; a run
// ends with 7 LDS atomics of that lane, what is pending at the end of the span is folded across the wave.  Measured on the full-size
// passes of top_bin: 110 us against 129 us for the wave-uniform runs above (batches straddling a bin boundary send most lanes to the LDS
// bins there), while the wave-uniform runs are the faster ones inside small_build (2.65 against 2.85 ms).
struct LaneRuns { int b[3]; uint32_t lo[3][3], hi[3][3], n[3]; };
__device__ __forceinline__ void lane_runs_init(LaneRuns& r) { for (int d = 0; d < 3; d++) { r.b[d] = -1; r.n[d] = 0u; for (int k = 0; k < 3; k++) { r.lo[d][k] = 0xFFFFFFFFu; r.hi[d][k] = 0u; } } }
__device__ __forceinline__ void lane_runs_add(LaneRuns& r, uint32_t* bins, const Mapping& m, const PrimRef& p, bool valid) {
  if (!valid) return;
  uint32_t c[6];
  for (int k = 0; k < 3; k++) { c[k] = enc(p.lo[k]); c[3 + k] = enc(p.hi[k]); }
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const int b = bin_clamped(p.lo[d] + p.hi[d], m.ofs[d], m.scale[d], m.nb);
    if (b != r.b[d]) {
      if (r.n[d]) {                                             // the run ends: its 7 atomics
        uint32_t* e = bins + (d * NBINS + r.b[d]) * BINW;
        atomicMin(&e[0], r.lo[d][0]); atomicMin(&e[1], r.lo[d][1]); atomicMin(&e[2], r.lo[d][2]);
        atomicMax(&e[3], r.hi[d][0]); atomicMax(&e[4], r.hi[d][1]); atomicMax(&e[5], r.hi[d][2]);
        atomicAdd(&e[6], r.n[d]);
      }
      r.b[d] = b; r.n[d] = 1u;
      for (int k = 0; k < 3; k++) { r.lo[d][k] = c[k]; r.hi[d][k] = c[3 + k]; }
    } else {
      r.n[d]++;
      for (int k = 0; k < 3; k++) { r.lo[d][k] = min(r.lo[d][k], c[k]); r.hi[d][k] = max(r.hi[d][k], c[3 + k]); }
    }
  }
}
// every lane of the wave calls it: the pending runs of the lanes that share a bin are reduced in registers, lane 63 issues the atomics
__device__ __forceinline__ void lane_runs_flush_wave(LaneRuns& r, uint32_t* bins, uint32_t lane) {
#pragma unroll
  for (int d = 0; d < 3; d++) {
    int b = r.n[d] ? r.b[d] : -1;
    unsigned long long rem = __ballot(b >= 0);
    for (int round = 0; round < 3; round++) {
      if (__popcll(rem) < 8) break;
      const int b0 = __builtin_amdgcn_readlane(b, __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(rem)));
      const bool mt = b == b0;
      const unsigned long long mm = __ballot(mt);
      if (__popcll(mm) < 4) break;
      uint32_t v[6];
      for (int k = 0; k < 3; k++) { v[k] = wave_umin63(mt ? r.lo[d][k] : 0xFFFFFFFFu); v[3 + k] = wave_umax63(mt ? r.hi[d][k] : 0u); }
      const uint32_t cnt = wave_uadd63(mt ? r.n[d] : 0u);
      if (lane == 63u) {
        uint32_t* e = bins + (d * NBINS + b0) * BINW;
        atomicMin(&e[0], v[0]); atomicMin(&e[1], v[1]); atomicMin(&e[2], v[2]);
        atomicMax(&e[3], v[3]); atomicMax(&e[4], v[4]); atomicMax(&e[5], v[5]);
        atomicAdd(&e[6], cnt);
      }
      if (mt) b = -1;
      rem &= ~mm;
    }
    if (b >= 0) {
      uint32_t* e = bins + (d * NBINS + b) * BINW;
      atomicMin(&e[0], r.lo[d][0]); atomicMin(&e[1], r.lo[d][1]); atomicMin(&e[2], r.lo[d][2]);
      atomicMax(&e[3], r.hi[d][0]); atomicMax(&e[4], r.hi[d][1]); atomicMax(&e[5], r.hi[d][2]);
      atomicAdd(&e[6], r.n[d]);
    }
  }
}

// Row aggregation for top_bin: the 16 lanes of a DPP row hold 16 consecutive triangles, which sit in one bin per axis or straddle ONE bin
// boundary (measured with cycle counters on the crown stand-in: consecutive triangles march along a ring of a sphere, 16 of them cover
// about one bin width, so "everything in one bin" is the exception there).  A row therefore forms two groups, the lanes in its lowest and
// in its highest bin, reduces each with four row_shr steps (result in lane 15 of the row) and that lane issues 7 atomics per group;
// a lane strictly between the two, and rows holding the end of the chunk, go lane by lane.  At most 4 x 14 instead of 64 x 7 same-word
// LDS atomics per axis and batch -- the atomics are what top_bin waits for (PMC: SQ_WAIT_INST_LDS 73 % of the wave cycles).
__device__ __forceinline__ uint32_t row_umin15(uint32_t v) {
  v = min(v, dpp_u<0x111, 0xF>(v, v)); v = min(v, dpp_u<0x112, 0xF>(v, v)); v = min(v, dpp_u<0x114, 0xF>(v, v)); v = min(v, dpp_u<0x118, 0xF>(v, v));
  return v;
}
__device__ __forceinline__ uint32_t row_umax15(uint32_t v) {
  v = max(v, dpp_u<0x111, 0xF>(v, v)); v = max(v, dpp_u<0x112, 0xF>(v, v)); v = max(v, dpp_u<0x114, 0xF>(v, v)); v = max(v, dpp_u<0x118, 0xF>(v, v));
  return v;
}
__device__ __forceinline__ void bins_add_rows(uint32_t* bins, const Mapping& m, const PrimRef& p, bool valid, uint32_t lane) {
  uint32_t c[6];
  for (int k = 0; k < 3; k++) { c[k] = enc(p.lo[k]); c[3 + k] = enc(p.hi[k]); }
  const unsigned long long vm = __ballot(valid);
  const uint32_t rowBase = lane & 48u;
  const bool rowFull = ((vm >> rowBase) & 0xFFFFull) == 0xFFFFull;          // all 16 lanes of my row hold a triangle
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const uint32_t b = valid ? (uint32_t)bin_clamped(p.lo[d] + p.hi[d], m.ofs[d], m.scale[d], m.nb) : 0u;
    // the row's lowest and highest bin (lane 15 holds the reduction; everybody reads it from there)
    const uint32_t bmin = (uint32_t)__shfl((int)row_umin15(b), (int)(lane | 15u), 64), bmax = (uint32_t)__shfl((int)row_umax15(b), (int)(lane | 15u), 64);
    const bool inLo = rowFull && b == bmin, inHi = rowFull && b == bmax && bmax != bmin;
    // 16 consecutive triangles sit in one bin or straddle one boundary: two groups cover the row; a lane strictly between goes alone
    uint32_t lo[6], hi[6];
    for (int k = 0; k < 3; k++) {
      lo[k] = row_umin15(inLo ? c[k] : 0xFFFFFFFFu); lo[3 + k] = row_umax15(inLo ? c[3 + k] : 0u);
      hi[k] = row_umin15(inHi ? c[k] : 0xFFFFFFFFu); hi[3 + k] = row_umax15(inHi ? c[3 + k] : 0u);
    }
    const uint32_t nLo = (uint32_t)__popcll((__ballot(inLo) >> rowBase) & 0xFFFFull), nHi = (uint32_t)__popcll((__ballot(inHi) >> rowBase) & 0xFFFFull);
    if ((lane & 15u) == 15u && rowFull) {
      uint32_t* e = bins + (d * NBINS + bmin) * BINW;
      atomicMin(&e[0], lo[0]); atomicMin(&e[1], lo[1]); atomicMin(&e[2], lo[2]);
      atomicMax(&e[3], lo[3]); atomicMax(&e[4], lo[4]); atomicMax(&e[5], lo[5]);
      atomicAdd(&e[6], nLo);
      if (nHi) {
        uint32_t* f = bins + (d * NBINS + bmax) * BINW;
        atomicMin(&f[0], hi[0]); atomicMin(&f[1], hi[1]); atomicMin(&f[2], hi[2]);
        atomicMax(&f[3], hi[3]); atomicMax(&f[4], hi[4]); atomicMax(&f[5], hi[5]);
        atomicAdd(&f[6], nHi);
      }
    }
    if (valid && !inLo && !inHi) {
      uint32_t* e = bins + (d * NBINS + b) * BINW;
      atomicMin(&e[0], c[0]); atomicMin(&e[1], c[1]); atomicMin(&e[2], c[2]);
      atomicMax(&e[3], c[3]); atomicMax(&e[4], c[4]); atomicMax(&e[5], c[5]);
      atomicAdd(&e[6], 1u);
    }
  }
}

struct SplitResult { float sah; int dim, pos; uint32_t nL; float llo[3], lhi[3], rlo[3], rhi[3]; };

// BinInfoT::best (heuristic_binning.h:339-386) by ONE wavefront as two scans: lanes 0-31 hold the 32 bins of one axis, lanes
// 32-63 those of the next (second pass: the third axis).  An inclusive prefix scan gives "everything left of the plane", a
// suffix scan "everything right of it"; lane pos then prices the candidate (axis, pos).  The reference's "first strict minimum
// per axis, then first better axis" is the lexicographic minimum of (sah, axis, pos).  Result lands in `res` (LDS).
// (The first version let every candidate loop over all bins: 3 x 31 x 32 bin visits, ~1900 instructions per lane.)
__device__ void sah_best_wave(const uint32_t* bins, const Mapping& m, uint32_t shift, SplitResult* res, uint32_t lane) {
  const uint32_t b = lane & 31u, half = lane >> 5;
  const uint32_t add = (1u << shift) - 1u;
  unsigned long long bestKey = ~0ull; uint32_t bestNL = 0;
  float bl[3] = {0, 0, 0}, bh[3] = {0, 0, 0}, rl[3] = {0, 0, 0}, rh[3] = {0, 0, 0};
#pragma unroll
  for (uint32_t pass = 0; pass < 2u; pass++) {
    const uint32_t axis = pass * 2u + half;
    const bool live = axis < 3u && b < m.nb;
    float plo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, phi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    uint32_t pn = 0;
    if (live) {
      const uint32_t* e = bins + (axis * NBINS + b) * BINW;
      pn = e[6];
      if (pn) for (int d = 0; d < 3; d++) { plo[d] = dec(e[d]); phi[d] = dec(e[3 + d]); }
    }
    float slo[3] = {plo[0], plo[1], plo[2]}, shi[3] = {phi[0], phi[1], phi[2]}; uint32_t sn = pn;
    for (int o = 1; o < 32; o <<= 1) {
      const uint32_t un = (uint32_t)__shfl_up((int)pn, o, 32), dn = (uint32_t)__shfl_down((int)sn, o, 32);
      float ul[3], uh[3], dl[3], dh[3];
      for (int d = 0; d < 3; d++) { ul[d] = __shfl_up(plo[d], o, 32); uh[d] = __shfl_up(phi[d], o, 32); dl[d] = __shfl_down(slo[d], o, 32); dh[d] = __shfl_down(shi[d], o, 32); }
      if (b >= (uint32_t)o) { pn += un; for (int d = 0; d < 3; d++) { plo[d] = fminf(plo[d], ul[d]); phi[d] = fmaxf(phi[d], uh[d]); } }
      if (b + (uint32_t)o < 32u) { sn += dn; for (int d = 0; d < 3; d++) { slo[d] = fminf(slo[d], dl[d]); shi[d] = fmaxf(shi[d], dh[d]); } }
    }
    // candidate pos = b: left = prefix of lane b-1, right = my suffix
    const uint32_t lN = (uint32_t)__shfl_up((int)pn, 1, 32);
    float llo[3], lhi[3];
    for (int d = 0; d < 3; d++) { llo[d] = __shfl_up(plo[d], 1, 32); lhi[d] = __shfl_up(phi[d], 1, 32); }
    const bool cand = live && b != 0u && sel3(axis, m.scale[0], m.scale[1], m.scale[2]) != 0.0f && lN != 0u && sn != 0u;   // mapping.invalid(dim) :375, pos != 0 :379; an empty side is never selected
    if (cand) {
      const float lA = half_area3(lhi[0] - llo[0], lhi[1] - llo[1], lhi[2] - llo[2]);
      const float rA = half_area3(shi[0] - slo[0], shi[1] - slo[1], shi[2] - slo[2]);
      const float sah = fmaf(lA, (float)((lN + add) >> shift), rA * (float)((sn + add) >> shift));   // :367
      const unsigned long long key = ((unsigned long long)__float_as_uint(sah) << 32) | ((axis << 5) | b);   // sah >= 0: its bit pattern is order preserving
      if (key < bestKey) {
        bestKey = key; bestNL = lN;
        for (int d = 0; d < 3; d++) { bl[d] = llo[d]; bh[d] = lhi[d]; rl[d] = slo[d]; rh[d] = shi[d]; }
      }
    }
  }
  unsigned long long k = bestKey;
  for (int o = 32; o >= 1; o >>= 1) { const unsigned long long other = __shfl_xor(k, o, 64); k = other < k ? other : k; }
  if (lane == 0) { res->sah = __builtin_inff(); res->dim = -1; res->pos = 0; res->nL = 0; }
  if (k != ~0ull && bestKey == k) {                              // exactly one lane owns the minimum (the candidate index is unique)
    res->sah = __uint_as_float((uint32_t)(k >> 32)); res->dim = (int)((k >> 5) & 3u); res->pos = (int)(k & 31u); res->nL = bestNL;
    for (int d = 0; d < 3; d++) { res->llo[d] = bl[d]; res->lhi[d] = bh[d]; res->rlo[d] = rl[d]; res->rhi[d] = rh[d]; }
  }
}

// ------------------------------------------------------------------------------------ K2 top phase
__global__ __launch_bounds__(256) void top_setup(Seg* segs, uint32_t* bins, Chunk* chunks, Counters* ctr) {
  __shared__ uint32_t s_base;
  const uint32_t s = blockIdx.x, tid = threadIdx.x;
  if (s >= ctr->numSegs) return;                                // the grid is an upper bound (2^level segments at most)
  Seg* sg = segs + s;
  const uint32_t begin = sg->begin, end = sg->end, n = end - begin;
  bins_clear(bins + (size_t)s * BINS_WORDS, tid, 256u);
  const uint32_t nch = (n + CHUNK - 1u) / CHUNK;
  if (tid == 0) {
    const Mapping m = make_mapping(n, sg->cmin, sg->cmax);
    for (int d = 0; d < 3; d++) { sg->ofs[d] = m.ofs[d]; sg->scale[d] = m.scale[d]; }
    sg->nb = m.nb;
    s_base = atomicAdd(&ctr->numChunks, nch);
  }
  __syncthreads();
  for (uint32_t c = tid; c < nch; c += 256u) {
    Chunk ck; ck.seg = s; ck.begin = begin + c * CHUNK; ck.end = min(ck.begin + CHUNK, end);
    chunks[s_base + c] = ck;
  }
}

__global__ __launch_bounds__(256) void top_bin(const Seg* segs, const Chunk* chunks, const PrimRef* src, uint32_t* bins, const Counters* ctr) {
  __shared__ uint32_t s_bins[BINS_WORDS];
  const uint32_t tid = threadIdx.x;
  if (blockIdx.x >= ctr->numChunks) return;
  const Chunk ck = chunks[blockIdx.x];
  const Seg* sg = segs + ck.seg;
  Mapping m; for (int d = 0; d < 3; d++) { m.ofs[d] = sg->ofs[d]; m.scale[d] = sg->scale[d]; } m.nb = sg->nb;
  bins_clear(s_bins, tid, 256u);
  __syncthreads();
  {                                                             // each wave owns a contiguous quarter of the chunk (see BinRuns)
    const uint32_t lane = tid & 63u, span = ck.begin + (tid >> 6) * (CHUNK / 4u), spanEnd = min(span + CHUNK / 4u, ck.end);
    for (uint32_t i0 = span; i0 < spanEnd; i0 += 64u) {         // wave-uniform trip count
      const uint32_t i = i0 + lane; const bool v = i < spanEnd;
      PrimRef r{}; if (v) r = load_prim(src + i);
      bins_add_rows(s_bins, m, r, v, lane);
    }
  }
  __syncthreads();
  uint32_t* g = bins + (size_t)ck.seg * BINS_WORDS;
  for (uint32_t w = tid; w < (uint32_t)BINS_WORDS; w += 256u) {      // BinInfoT::merge :312-321
    const uint32_t k = w % BINW, cnt = s_bins[w - k + 6];
    if (cnt == 0u) continue;
    if (k < 3) atomicMin(&g[w], s_bins[w]); else if (k < 6) atomicMax(&g[w], s_bins[w]); else atomicAdd(&g[w], s_bins[w]);
  }
}

__global__ __launch_bounds__(64) void top_split(Seg* segs, const uint32_t* bins, BNode* bnodes, Counters* ctr, Params prm, uint32_t forceFallback) {
  __shared__ SplitResult s_res;
  const uint32_t s = blockIdx.x, lane = threadIdx.x;
  if (s >= ctr->numSegs) return;
  Seg* sg = segs + s;
  Mapping m; for (int d = 0; d < 3; d++) { m.ofs[d] = sg->ofs[d]; m.scale[d] = sg->scale[d]; } m.nb = sg->nb;
  sah_best_wave(bins + (size_t)s * BINS_WORDS, m, prm.shift, &s_res, lane);
  __syncthreads();
  if (lane == 0) {
    const uint32_t begin = sg->begin, end = sg->end, n = end - begin;
    SplitResult r = s_res;
    const bool fallback = (r.dim < 0) || forceFallback;        // split invalid -> median split (split_template :144-147)
    const uint32_t nL = fallback ? ((begin + end) / 2u - begin) : r.nL;
    const uint32_t idL = sg->bnode + 1u, idR = sg->bnode + 2u * nL;   // implicit pre-order numbering (see K3)
    BNode* par = bnodes + sg->bnode;
    par->left = idL; par->right = idR; par->splitSah = r.sah;
    BNode L{}, R{};
    L.begin = begin; L.end = begin + nL; R.begin = begin + nL; R.end = end;
    L.left = L.right = R.left = R.right = NIL; L.splitSah = R.splitSah = __builtin_inff();
    for (int d = 0; d < 3; d++) { L.lo[d] = r.llo[d]; L.hi[d] = r.lhi[d]; R.lo[d] = r.rlo[d]; R.hi[d] = r.rhi[d]; }
    bnodes[idL] = L; bnodes[idR] = R;
    sg->flags = fallback ? 1u : 0u; sg->dim = fallback ? 0u : (uint32_t)r.dim; sg->pos = (uint32_t)r.pos; sg->nL = nL;
    sg->childL = idL; sg->childR = idR; sg->curL = begin; sg->curR = begin + nL;
    for (int side = 0; side < 2; side++) for (int k = 0; k < 12; k++) sg->acc[side][k] = (k % 6 < 3) ? ENC_POS_INF : ENC_NEG_INF;
    (void)n;
  }
}

__global__ __launch_bounds__(256) void top_partition(Seg* segs, const Chunk* chunks, const PrimRef* src, PrimRef* dst, const Counters* ctr) {
  __shared__ uint32_t s_cnt[CHUNK_ROUNDS][4][2], s_off[CHUNK_ROUNDS][4][2], s_acc[2][12], s_baseL, s_baseR;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (blockIdx.x >= ctr->numChunks) return;
  const Chunk ck = chunks[blockIdx.x];
  Seg* sg = segs + ck.seg;
  const bool fallback = (sg->flags & 1u) != 0u;
  const uint32_t dim = sg->dim, pos = sg->pos, mid = sg->begin + sg->nL;
  const float ofs = sg->ofs[dim], scale = sg->scale[dim];
  if (tid < 24) s_acc[tid / 12][tid % 12] = (tid % 6 < 3) ? ENC_POS_INF : ENC_NEG_INF;
  __syncthreads();
  PrimRef pr[CHUNK_ROUNDS]; uint32_t sideBits = 0, validBits = 0; unsigned long long lm[CHUNK_ROUNDS], rm[CHUNK_ROUNDS];
  uint32_t aL[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u}, aR[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
#pragma unroll
  for (int r = 0; r < CHUNK_ROUNDS; r++) {
    const uint32_t i = ck.begin + (uint32_t)r * 256u + tid;
    const bool v = i < ck.end;
    if (v) pr[r] = load_prim(src + i);
    bool left = false;
    if (v) {
      const float c2 = sel3(dim, pr[r].lo[0] + pr[r].hi[0], pr[r].lo[1] + pr[r].hi[1], pr[r].lo[2] + pr[r].hi[2]);
      left = fallback ? (i < mid) : (bin_unsafe(c2, ofs, scale) < (int)pos);     // isLeft: bin_unsafe(center2) < pos (:161)
      const int side = left ? 0 : 1;
      for (int d = 0; d < 3; d++) {                                              // extend_center2 of the child (:168), thread-private first
        const uint32_t cc = enc(pr[r].lo[d] + pr[r].hi[d]);
        if (left) { aL[d] = min(aL[d], cc); aL[3 + d] = max(aL[3 + d], cc); } else { aR[d] = min(aR[d], cc); aR[3 + d] = max(aR[3 + d], cc); }
      }
      if (fallback) for (int d = 0; d < 3; d++) { atomicMin(&s_acc[side][6 + d], enc(pr[r].lo[d])); atomicMax(&s_acc[side][9 + d], enc(pr[r].hi[d])); }
    }
    lm[r] = __ballot(v && left); rm[r] = __ballot(v && !left);
    if (lane == 0) { s_cnt[r][wave][0] = (uint32_t)__popcll(lm[r]); s_cnt[r][wave][1] = (uint32_t)__popcll(rm[r]); }
    if (v) validBits |= 1u << r;
    if (left) sideBits |= 1u << r;
  }
  for (int k = 0; k < 6; k++) {                                                  // wave-reduce the private bounds, one lane publishes
    const uint32_t x = k < 3 ? wave_umin63(aL[k]) : wave_umax63(aL[k]), y = k < 3 ? wave_umin63(aR[k]) : wave_umax63(aR[k]);
    if (lane == 63u) { if (k < 3) { atomicMin(&s_acc[0][k], x); atomicMin(&s_acc[1][k], y); } else { atomicMax(&s_acc[0][k], x); atomicMax(&s_acc[1][k], y); } }
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t l = 0, rr = 0;
    for (int r = 0; r < CHUNK_ROUNDS; r++) for (int w = 0; w < 4; w++) { s_off[r][w][0] = l; s_off[r][w][1] = rr; l += s_cnt[r][w][0]; rr += s_cnt[r][w][1]; }
    s_baseL = l ? atomicAdd(&sg->curL, l) : 0u; s_baseR = rr ? atomicAdd(&sg->curR, rr) : 0u;
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < CHUNK_ROUNDS; r++) {
    if (!(validBits & (1u << r))) continue;
    const bool left = (sideBits >> r) & 1u;
    const uint32_t o = left ? s_baseL + s_off[r][wave][0] + (uint32_t)__popcll(lm[r] & lt)
                            : s_baseR + s_off[r][wave][1] + (uint32_t)__popcll(rm[r] & lt);
    store_prim(dst + o, pr[r]);
  }
  if (tid < 24) {
    const uint32_t side = tid / 12, k = tid % 12, v = s_acc[side][k];
    if (k % 6 < 3) { if (v != ENC_POS_INF) atomicMin(&sg->acc[side][k], v); } else { if (v != ENC_NEG_INF) atomicMax(&sg->acc[side][k], v); }
  }
}

__global__ void top_emit(const Seg* segs, BNode* bnodes, Seg* next, SmallEntry* small, Counters* ctr,
                         Params prm, uint32_t dstBuf, uint32_t maxNext, uint32_t maxSmall) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ctr->numSegs) return;
  const Seg* sg = segs + s;
  for (int side = 0; side < 2; side++) {
    const uint32_t b = side ? sg->begin + sg->nL : sg->begin, e = side ? sg->end : sg->begin + sg->nL;
    const uint32_t child = side ? sg->childR : sg->childL;
    float cmin[3], cmax[3];
    for (int d = 0; d < 3; d++) { cmin[d] = dec(sg->acc[side][d]); cmax[d] = dec(sg->acc[side][3 + d]); }
    if (sg->flags & 1u) for (int d = 0; d < 3; d++) { bnodes[child].lo[d] = dec(sg->acc[side][6 + d]); bnodes[child].hi[d] = dec(sg->acc[side][9 + d]); }
    if (e - b <= prm.small) {
      const uint32_t k = atomicAdd(&ctr->numSmall, 1u);
      if (k >= maxSmall) { ctr->overflow = 1u; continue; }
      SmallEntry se; se.begin = b; se.end = e; se.bnode = child; se.buf = dstBuf;
      for (int d = 0; d < 3; d++) { se.cmin[d] = cmin[d]; se.cmax[d] = cmax[d]; }
      small[k] = se;
    } else {
      const uint32_t k = atomicAdd(&ctr->numSegsNext, 1u);
      if (k >= maxNext) { ctr->overflow = 1u; continue; }
      Seg ns{}; ns.begin = b; ns.end = e; ns.bnode = child;
      for (int d = 0; d < 3; d++) { ns.cmin[d] = cmin[d]; ns.cmax[d] = cmax[d]; }
      next[k] = ns;
    }
  }
}

__global__ void top_advance(Counters* ctr, uint32_t maxNext) {      // end of a top level: next level's work list becomes current
  if (ctr->numSegs) ctr->topLevels++;
  if (ctr->numSegsNext > maxNext) ctr->overflow = 1u;
  ctr->numSegs = ctr->numSegsNext < maxNext ? ctr->numSegsNext : maxNext; ctr->numSegsNext = 0; ctr->numChunks = 0;
}

// ---------------------------------------------------------------------------------- K3 small phase
// One wavefront finishes a sub-tree of <= small_threshold triangles.  Two modes:
//   * segments of more than MICRO triangles: the wave splits ONE segment at a time (bins in LDS, ping-pong partition
//     through HBM/L2, explicit stack), exactly like the top phase but without leaving the CU;
//   * segments of <= MICRO (= 64) triangles -- 94 % of all binary nodes of a scene -- are finished by micro_subtree():
//     one triangle per lane, and ALL segments of a level are split in the same pass (per-segment bins, candidate
//     evaluation, argmin and partition all live in LDS; segments are contiguous lane ranges).  A wave instruction thus
//     serves up to 32 splits instead of one: the first version of this kernel spent 35 of the 45 ms of a 4.8 M triangle
//     commit walking those tiny segments one by one (profiles/r01_bench_kernel_stats_v2.md).
// Binary node numbering is implicit -- the children of node k over nL + nR triangles are k + 1 and k + 2 nL (pre-order,
// a sub-tree of n triangles owns ids [k, k + 2n - 1)) -- so no global counter is touched and the numbering is the same
// on every run.
struct StackEntry { uint32_t begin, end, bnode, buf; float cmin[3], cmax[3]; };
constexpr uint32_t MICRO = 64;

// zero-identity encodings for LDS atomicMax accumulators that are cleared with plain zero stores
__device__ __forceinline__ uint32_t zlo(float f) { return ~enc(f); }          // max of zlo = min of f
__device__ __forceinline__ float unzlo(uint32_t u) { return dec(~u); }
__device__ __forceinline__ uint32_t zhi(float f) { return enc(f); }           // enc(x) > 0 for every float
__device__ __forceinline__ float unzhi(uint32_t u) { return dec(u); }

// R: per-wave LDS scratch of 64 * W words (W = 32 words per triangle when min_leaf >= 2, 48 for min_leaf = 1):
//   bins of the segment starting at lane b live at R + b * W as [axis][bin][8] (3 * nb * 8 <= n * W words for every
//   splittable n); once the candidates are evaluated the same memory holds the split records (16 words per segment at
//   R + b * 16) and the exchange buffer the partition moves the triangles through (10 x 64 words at R + 1024).
__device__ void micro_subtree(uint32_t* R, uint32_t W, uint32_t (*s_cb)[64][6], unsigned long long* s_key, const PrimRef* src,
                              uint32_t gbegin, uint32_t n0, uint32_t rootNode, const float* cmin0, const float* cmax0,
                              BNode* bnodes, uint2* finalIds, Counters* ctr, const Params& prm, uint32_t lane) {
  PrimRef p{};
  if (lane < n0) p = load_prim(src + gbegin + lane);
  uint32_t segB = 0, segE = n0, node = rootNode;
  bool act = lane < n0 && n0 > prm.minLeaf;
  if (lane == 0u) for (int d = 0; d < 3; d++) { s_cb[0][0][d] = zlo(cmin0[d]); s_cb[0][0][3 + d] = zhi(cmax0[d]); }
  __syncthreads();
  const uint32_t addBlk = (1u << prm.shift) - 1u;
  uint32_t pp = 0;
  for (uint32_t level = 0; level < 64u; level++) {
    if (__ballot(act) == 0ull) break;
    // ---- L0: bin mapping of my segment (BinMapping, heuristic_binning.h:46-55); clear bins, keys, next level's centroid bounds
    const uint32_t n = segE - segB;
    float ofs[3] = {0, 0, 0}, scale[3] = {0, 0, 0}; uint32_t nb = 4;
    if (act) {
      float cmin[3], cmax[3];
      for (int d = 0; d < 3; d++) { cmin[d] = unzlo(s_cb[pp][segB][d]); cmax[d] = unzhi(s_cb[pp][segB][3 + d]); }
      const Mapping m = make_mapping(n, cmin, cmax);
      for (int d = 0; d < 3; d++) { ofs[d] = m.ofs[d]; scale[d] = m.scale[d]; }
      nb = m.nb;
    }
    __syncthreads();                                             // everybody has read s_cb[pp] and is done with the exchange buffer
    for (uint32_t i = 0; i < W / 4u; i++) ((uint4*)R)[i * 64u + lane] = make_uint4(0u, 0u, 0u, 0u);
    s_key[lane] = ~0ull;
    for (int k = 0; k < 6; k++) s_cb[pp ^ 1u][lane][k] = 0u;
    __syncthreads();
    // ---- L1: bin (BinInfoT::bin, heuristic_binning.h:210-257)
    uint32_t* const sb = R + segB * W;
    if (act) {
      for (int d = 0; d < 3; d++) {
        const int b = bin_clamped(p.lo[d] + p.hi[d], ofs[d], scale[d], nb);
        uint32_t* e = sb + ((uint32_t)d * nb + (uint32_t)b) * 8u;     // 8-word entries: lo.xyz hi.xyz count pad (two 16-byte reads)
        atomicMax(&e[0], zlo(p.lo[0])); atomicMax(&e[1], zlo(p.lo[1])); atomicMax(&e[2], zlo(p.lo[2]));
        atomicMax(&e[3], zhi(p.hi[0])); atomicMax(&e[4], zhi(p.hi[1])); atomicMax(&e[5], zhi(p.hi[2]));
        atomicAdd(&e[6], 1u);
      }
    }
    __syncthreads();
    // ---- L2: candidates (BinInfoT::best :339-386): lane j of a segment evaluates candidates j, j + n, ...;
    //      candidate c = axis * (nb - 1) + (pos - 1), so the minimum of (sah, c) is the reference's choice
    float bestSah = __builtin_inff(); uint32_t bestC = NIL, bestNL = 0;
    float bl[3] = {0, 0, 0}, bh[3] = {0, 0, 0}, rl[3] = {0, 0, 0}, rh[3] = {0, 0, 0};
    if (act && nb == 4u) {
      // the common case (n < 20): lane j of the segment sweeps axis j once -- suffix bounds S1..S3, then a running prefix; 4 bins are read
      // once (8 x 16 bytes) instead of once per candidate
      for (uint32_t axis = lane - segB; axis < 3u; axis += n) {
        if (sel3(axis, scale[0], scale[1], scale[2]) == 0.0f) continue;          // mapping.invalid(dim) :375
        const uint4* e = (const uint4*)(sb + axis * 32u);
        float lo[4][3], hi[4][3]; uint32_t cn[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const uint4 x = e[2 * b], y = e[2 * b + 1];
          cn[b] = y.z;
          const bool any = cn[b] != 0u;
          lo[b][0] = any ? unzlo(x.x) : __builtin_inff(); lo[b][1] = any ? unzlo(x.y) : __builtin_inff(); lo[b][2] = any ? unzlo(x.z) : __builtin_inff();
          hi[b][0] = any ? unzhi(x.w) : -__builtin_inff(); hi[b][1] = any ? unzhi(y.x) : -__builtin_inff(); hi[b][2] = any ? unzhi(y.y) : -__builtin_inff();
        }
        float slo[4][3], shi[4][3]; uint32_t sn[4];                            // suffix: bins pos..3
        for (int d = 0; d < 3; d++) { slo[3][d] = lo[3][d]; shi[3][d] = hi[3][d]; } sn[3] = cn[3];
#pragma unroll
        for (int b = 2; b >= 1; b--) { for (int d = 0; d < 3; d++) { slo[b][d] = fminf(lo[b][d], slo[b + 1][d]); shi[b][d] = fmaxf(hi[b][d], shi[b + 1][d]); } sn[b] = cn[b] + sn[b + 1]; }
        float llo[3] = {lo[0][0], lo[0][1], lo[0][2]}, lhi[3] = {hi[0][0], hi[0][1], hi[0][2]}; uint32_t lN = cn[0];
#pragma unroll
        for (int pos = 1; pos < 4; pos++) {
          if (lN != 0u && sn[pos] != 0u) {
            const float lA = half_area3(lhi[0] - llo[0], lhi[1] - llo[1], lhi[2] - llo[2]);
            const float rA = half_area3(shi[pos][0] - slo[pos][0], shi[pos][1] - slo[pos][1], shi[pos][2] - slo[pos][2]);
            const float sah = fmaf(lA, (float)((lN + addBlk) >> prm.shift), rA * (float)((sn[pos] + addBlk) >> prm.shift));
            if (sah < bestSah) {
              bestSah = sah; bestC = axis * 3u + (uint32_t)(pos - 1); bestNL = lN;
              for (int d = 0; d < 3; d++) { bl[d] = llo[d]; bh[d] = lhi[d]; rl[d] = slo[pos][d]; rh[d] = shi[pos][d]; }
            }
          }
          for (int d = 0; d < 3; d++) { llo[d] = fminf(llo[d], lo[pos][d]); lhi[d] = fmaxf(lhi[d], hi[pos][d]); }
          lN += cn[pos];
        }
      }
    } else if (act) {
      const uint32_t nb1 = nb - 1u, ncand = 3u * nb1;
      for (uint32_t c = lane - segB; c < ncand; c += n) {
        const uint32_t axis = (c >= nb1 ? 1u : 0u) + (c >= 2u * nb1 ? 1u : 0u), pos = c - axis * nb1 + 1u;
        if (sel3(axis, scale[0], scale[1], scale[2]) == 0.0f) continue;          // mapping.invalid(dim) :375
        float llo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, lhi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        float rlo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, rhi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        uint32_t lN = 0, rN = 0;
        for (uint32_t b = 0; b < nb; b++) {
          const uint32_t* e = sb + (axis * nb + b) * 8u;
          const uint32_t cnt = e[6];
          if (cnt == 0u) continue;
          if (b < pos) { lN += cnt; for (int d = 0; d < 3; d++) { llo[d] = fminf(llo[d], unzlo(e[d])); lhi[d] = fmaxf(lhi[d], unzhi(e[3 + d])); } }
          else         { rN += cnt; for (int d = 0; d < 3; d++) { rlo[d] = fminf(rlo[d], unzlo(e[d])); rhi[d] = fmaxf(rhi[d], unzhi(e[3 + d])); } }
        }
        if (lN == 0u || rN == 0u) continue;
        const float lA = half_area3(lhi[0] - llo[0], lhi[1] - llo[1], lhi[2] - llo[2]);
        const float rA = half_area3(rhi[0] - rlo[0], rhi[1] - rlo[1], rhi[2] - rlo[2]);
        const float sah = fmaf(lA, (float)((lN + addBlk) >> prm.shift), rA * (float)((rN + addBlk) >> prm.shift));
        if (sah < bestSah) {
          bestSah = sah; bestC = c; bestNL = lN;
          for (int d = 0; d < 3; d++) { bl[d] = llo[d]; bh[d] = lhi[d]; rl[d] = rlo[d]; rh[d] = rhi[d]; }
        }
      }
    }
    const unsigned long long key = bestC == NIL ? ~0ull : (((unsigned long long)__float_as_uint(bestSah) << 32) | bestC);
    if (act && key != ~0ull) atomicMin(&s_key[segB], key);
    __syncthreads();                                             // bins are dead from here on: R now holds split records + exchange buffer
    const unsigned long long win = act ? s_key[segB] : 0ull;
    const bool fb = act && win == ~0ull;                          // no valid candidate -> median split (split_template :144-147)
    uint32_t* const rec = R + segB * 16u;
    if (act && !fb && key == win) {
      const uint32_t nb1 = nb - 1u, axis = (bestC >= nb1 ? 1u : 0u) + (bestC >= 2u * nb1 ? 1u : 0u), pos = bestC - axis * nb1 + 1u;
      rec[0] = axis | (pos << 8); rec[1] = bestNL; rec[2] = __float_as_uint(bestSah);
      for (int d = 0; d < 3; d++) { rec[4 + d] = __float_as_uint(bl[d]); rec[7 + d] = __float_as_uint(bh[d]); rec[10 + d] = __float_as_uint(rl[d]); rec[13 + d] = __float_as_uint(rh[d]); }
    }
    if (fb && lane == segB) {
      rec[0] = 1u << 16; rec[1] = (gbegin + segB + gbegin + segE) / 2u - (gbegin + segB); rec[2] = __float_as_uint(__builtin_inff());
      for (int k = 4; k < 16; k++) rec[k] = 0u;
    }
    __syncthreads();
    if (__ballot(fb) != 0ull) {                                   // child geometry bounds of a median split: reduce over the triangles
      if (fb) {
        const uint32_t o = lane < segB + rec[1] ? 4u : 10u;
        for (int d = 0; d < 3; d++) { atomicMax(&rec[o + d], zlo(p.lo[d])); atomicMax(&rec[o + 3 + d], zhi(p.hi[d])); }
      }
      __syncthreads();
    }
    // ---- L4: partition (heuristic_binning_array_aligned.h:150-176): new lane of my triangle, child centroid bounds, node records
    bool left = false; uint32_t nL = 0;
    if (act) {
      const uint32_t w0 = rec[0], dim = w0 & 3u, pos = (w0 >> 8) & 0xFFu; nL = rec[1];
      const float c2 = sel3(dim, p.lo[0] + p.hi[0], p.lo[1] + p.hi[1], p.lo[2] + p.hi[2]);
      left = (w0 >> 16) ? (lane < segB + nL) : (bin_unsafe(c2, sel3(dim, ofs[0], ofs[1], ofs[2]), sel3(dim, scale[0], scale[1], scale[2])) < (int)pos);
    }
    const unsigned long long segMask = act ? ((n >= 64u ? ~0ull : ((1ull << n) - 1ull)) << segB) : 0ull;
    const unsigned long long lm = __ballot(act && left) & segMask, rm = __ballot(act && !left) & segMask, lt = (1ull << lane) - 1ull;
    if (act) {
      const uint32_t nSegB = left ? segB : segB + nL, nSegE = left ? segB + nL : segE, nNode = left ? node + 1u : node + 2u * nL;
      const uint32_t npos = left ? segB + (uint32_t)__popcll(lm & lt) : segB + nL + (uint32_t)__popcll(rm & lt);
      for (int d = 0; d < 3; d++) { const float cc = p.lo[d] + p.hi[d]; atomicMax(&s_cb[pp ^ 1u][nSegB][d], zlo(cc)); atomicMax(&s_cb[pp ^ 1u][nSegB][3 + d], zhi(cc)); }
      uint32_t* X = R + 1024u + npos;
      X[0] = __float_as_uint(p.lo[0]); X[64] = __float_as_uint(p.lo[1]); X[128] = __float_as_uint(p.lo[2]); X[192] = p.geom;
      X[256] = __float_as_uint(p.hi[0]); X[320] = __float_as_uint(p.hi[1]); X[384] = __float_as_uint(p.hi[2]); X[448] = p.prim;
      X[512] = nSegB | (nSegE << 8); X[576] = nNode;
      if (lane == segB) {                                        // one lane per segment: my links, my children's boxes and ranges
        const bool isfb = (rec[0] >> 16) != 0u;
        float cb[12];
        for (int k = 0; k < 12; k++) cb[k] = isfb ? ((k % 6) < 3 ? unzlo(rec[4 + k]) : unzhi(rec[4 + k])) : __uint_as_float(rec[4 + k]);
        const uint32_t L = node + 1u, Rr = node + 2u * nL;
        ((uint4*)(bnodes + node))[2] = make_uint4(L, Rr, rec[2], 0u);
        ((float4*)(bnodes + L))[0] = make_float4(cb[0], cb[1], cb[2], __uint_as_float(gbegin + segB));
        ((float4*)(bnodes + L))[1] = make_float4(cb[3], cb[4], cb[5], __uint_as_float(gbegin + segB + nL));
        ((float4*)(bnodes + Rr))[0] = make_float4(cb[6], cb[7], cb[8], __uint_as_float(gbegin + segB + nL));
        ((float4*)(bnodes + Rr))[1] = make_float4(cb[9], cb[10], cb[11], __uint_as_float(gbegin + segE));
      }
    }
    __syncthreads();
    // ---- L5: pick up the triangle that moved to my lane
    if (act) {
      const uint32_t* X = R + 1024u + lane;
      p.lo[0] = __uint_as_float(X[0]); p.lo[1] = __uint_as_float(X[64]); p.lo[2] = __uint_as_float(X[128]); p.geom = X[192];
      p.hi[0] = __uint_as_float(X[256]); p.hi[1] = __uint_as_float(X[320]); p.hi[2] = __uint_as_float(X[384]); p.prim = X[448];
      segB = X[512] & 0xFFu; segE = X[512] >> 8; node = X[576];
      act = segE - segB > prm.minLeaf;
    }
    pp ^= 1u;
  }
  // every remaining segment is a binary leaf (the reference never splits sets of <= minLeafSize, bvh_builder_sah.h:253)
  if (lane < n0) {
    finalIds[gbegin + lane] = make_uint2(p.geom, p.prim);
    if (lane == segB) ((uint4*)(bnodes + node))[2] = make_uint4(NIL, NIL, __float_as_uint(__builtin_inff()), 0u);
  }
  const unsigned long long leaves = __ballot(lane < n0 && lane == segB);
  if (lane == 0u) atomicAdd(&ctr->numBLeaves, (uint32_t)__popcll(leaves));
  __syncthreads();
}

__global__ __launch_bounds__(64) void small_build(const SmallEntry* entries, PrimRef* bufA, PrimRef* bufB, BNode* bnodes,
                                                  uint2* finalIds, Counters* ctr, Params prm, uint32_t W) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_R[];   // max(BINS_WORDS, 64 * W) words: bins / micro scratch
  __shared__ SplitResult s_res;
  __shared__ uint32_t s_acc[2][12];
  __shared__ StackEntry s_stack[24];
  __shared__ uint32_t s_cb[2][64][6];
  __shared__ unsigned long long s_key[64];
  uint32_t* const s_bins = s_R;
  const uint32_t lane = threadIdx.x;
  const SmallEntry e0 = entries[blockIdx.x];
  StackEntry cur; cur.begin = e0.begin; cur.end = e0.end; cur.bnode = e0.bnode; cur.buf = e0.buf;
  for (int d = 0; d < 3; d++) { cur.cmin[d] = e0.cmin[d]; cur.cmax[d] = e0.cmax[d]; }
  uint32_t sp = 0;
  for (uint32_t iter = 0; iter < (1u << 20); iter++) {         // the cap is a safety net only: <= 2*small_threshold iterations are possible
    const uint32_t n = cur.end - cur.begin;
    PrimRef* src = cur.buf ? bufB : bufA;
    PrimRef* dst = cur.buf ? bufA : bufB;
    if (n <= MICRO) {
      micro_subtree(s_R, W, s_cb, s_key, src, cur.begin, n, cur.bnode, cur.cmin, cur.cmax, bnodes, finalIds, ctr, prm, lane);
      if (sp == 0) break;
      cur = s_stack[--sp];
      __syncthreads();
      continue;
    }
    const Mapping m = make_mapping(n, cur.cmin, cur.cmax);
    bins_clear(s_bins, lane, 64u);
    if (lane < 24) s_acc[lane / 12][lane % 12] = (lane % 6 < 3) ? ENC_POS_INF : ENC_NEG_INF;
    __syncthreads();
    {
      BinRuns runs; runs_init(runs);
      for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
        const bool v = i0 + lane < n;
        PrimRef r{}; if (v) r = load_prim(src + cur.begin + i0 + lane);
        runs_add(runs, s_bins, m, r, v, lane);
      }
      runs_flush_wave(runs, s_bins, lane);
    }
    __syncthreads();
    sah_best_wave(s_bins, m, prm.shift, &s_res, lane);
    __syncthreads();
    const SplitResult r = s_res;
    const bool fallback = r.dim < 0;
    const uint32_t mid = fallback ? (cur.begin + cur.end) / 2u : cur.begin + r.nL;
    const uint32_t dim = fallback ? 0u : (uint32_t)r.dim;
    // partition into the other buffer (wave-synchronous compaction)
    uint32_t curL = cur.begin, curR = mid;
    uint32_t aL[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u}, aR[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
      const uint32_t i = cur.begin + i0 + lane;
      const bool v = i < cur.end;
      PrimRef p{}; bool left = false;
      if (v) {
        p = load_prim(src + i);
        left = fallback ? (i < mid) : (bin_unsafe(sel3(dim, p.lo[0] + p.hi[0], p.lo[1] + p.hi[1], p.lo[2] + p.hi[2]), sel3(dim, m.ofs[0], m.ofs[1], m.ofs[2]), sel3(dim, m.scale[0], m.scale[1], m.scale[2])) < r.pos);
        const int side = left ? 0 : 1;
        for (int d = 0; d < 3; d++) {
          const uint32_t cc = enc(p.lo[d] + p.hi[d]);
          if (left) { aL[d] = min(aL[d], cc); aL[3 + d] = max(aL[3 + d], cc); } else { aR[d] = min(aR[d], cc); aR[3 + d] = max(aR[3 + d], cc); }
          if (fallback) { atomicMin(&s_acc[side][6 + d], enc(p.lo[d])); atomicMax(&s_acc[side][9 + d], enc(p.hi[d])); }
        }
      }
      const unsigned long long lm = __ballot(v && left), rm = __ballot(v && !left), lt = (1ull << lane) - 1ull;
      if (v) store_prim(dst + (left ? curL + (uint32_t)__popcll(lm & lt) : curR + (uint32_t)__popcll(rm & lt)), p);
      curL += (uint32_t)__popcll(lm); curR += (uint32_t)__popcll(rm);
    }
    for (int k = 0; k < 6; k++) {
      const uint32_t x = k < 3 ? wave_umin63(aL[k]) : wave_umax63(aL[k]), y = k < 3 ? wave_umin63(aR[k]) : wave_umax63(aR[k]);
      if (lane == 63u) { s_acc[0][k] = x; s_acc[1][k] = y; }
    }
    __syncthreads();
    const uint32_t idL = cur.bnode + 1u, idR = cur.bnode + 2u * (mid - cur.begin);
    StackEntry L, R;
    L.begin = cur.begin; L.end = mid; L.bnode = idL; L.buf = cur.buf ^ 1u;
    R.begin = mid; R.end = cur.end; R.bnode = idR; R.buf = cur.buf ^ 1u;
    for (int d = 0; d < 3; d++) {
      L.cmin[d] = dec(s_acc[0][d]); L.cmax[d] = dec(s_acc[0][3 + d]);
      R.cmin[d] = dec(s_acc[1][d]); R.cmax[d] = dec(s_acc[1][3 + d]);
    }
    if (lane == 0) {
      BNode* par = bnodes + cur.bnode;
      par->left = idL; par->right = idR; par->splitSah = r.sah;
      BNode bl{}, br{};
      bl.begin = L.begin; bl.end = L.end; br.begin = R.begin; br.end = R.end;
      bl.left = bl.right = br.left = br.right = NIL; bl.splitSah = br.splitSah = __builtin_inff();
      for (int d = 0; d < 3; d++) {
        bl.lo[d] = fallback ? dec(s_acc[0][6 + d]) : r.llo[d]; bl.hi[d] = fallback ? dec(s_acc[0][9 + d]) : r.lhi[d];
        br.lo[d] = fallback ? dec(s_acc[1][6 + d]) : r.rlo[d]; br.hi[d] = fallback ? dec(s_acc[1][9 + d]) : r.rhi[d];
      }
      bnodes[idL] = bl; bnodes[idR] = br;
    }
    // continue with the smaller child, push the larger: the stack stays <= log2(small_threshold) deep
    const bool leftSmaller = (L.end - L.begin) <= (R.end - R.begin);
    StackEntry keep, push;                                       // field-wise selects: a struct-valued ?: goes through scratch memory
    keep.begin = leftSmaller ? L.begin : R.begin; keep.end = leftSmaller ? L.end : R.end; keep.bnode = leftSmaller ? L.bnode : R.bnode; keep.buf = L.buf;
    push.begin = leftSmaller ? R.begin : L.begin; push.end = leftSmaller ? R.end : L.end; push.bnode = leftSmaller ? R.bnode : L.bnode; push.buf = L.buf;
    for (int d = 0; d < 3; d++) {
      keep.cmin[d] = leftSmaller ? L.cmin[d] : R.cmin[d]; keep.cmax[d] = leftSmaller ? L.cmax[d] : R.cmax[d];
      push.cmin[d] = leftSmaller ? R.cmin[d] : L.cmin[d]; push.cmax[d] = leftSmaller ? R.cmax[d] : L.cmax[d];
    }
    __syncthreads();
    if (lane == 0) s_stack[sp] = push;
    sp++;
    cur = keep;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------- fast build (RTC_BUILD_QUALITY_LOW)
// The reference answers RTC_BUILD_QUALITY_LOW with its Morton builder (kernels/builders/bvh_builder_morton.h:  63-bit codes of the
// centroids, radix sort, recursive splits at the highest differing bit; selected per mesh by the two-level builder, kernels/bvh/
// bvh_builder_twolevel.cpp, kernels/common/scene.cpp:195-206).  The GPU formulation of the same tree: sort the 63-bit codes, then
// every internal node finds its own range and split from the codes alone (Karras 2012: the split of a range is where the common
// prefix of the codes is shortest; ties between equal codes are broken by the index), and the boxes are propagated from the leaves
// with one atomic flag per node.  The result is a binary tree in the BNode format, so the wide collapse, quantisation and leaf
// layout are the ones of the SAH build; only the decisions differ.  Leaf j is BNode (n-1)+j, internal node i is BNode i, root = 0.
__device__ __forceinline__ unsigned long long spread21(uint32_t v) {   // 21 bits -> every third bit
  unsigned long long x = v & 0x1FFFFFull;
  x = (x | x << 32) & 0x1F00000000FFFFull; x = (x | x << 16) & 0x1F0000FF0000FFull; x = (x | x << 8) & 0x100F00F00F00F00Full;
  x = (x | x << 4) & 0x10C30C30C30C30C3ull; x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
__global__ __launch_bounds__(256) void morton_keys(const PrimRef* prims, uint32_t n, float3 cmin, float3 cscale, unsigned long long* keys, uint32_t* vals) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const PrimRef p = load_prim(prims + i);
  const float fx = ((p.lo[0] + p.hi[0]) - cmin.x) * cscale.x, fy = ((p.lo[1] + p.hi[1]) - cmin.y) * cscale.y, fz = ((p.lo[2] + p.hi[2]) - cmin.z) * cscale.z;
  const uint32_t ix = (uint32_t)fminf(fmaxf(fx, 0.0f), 2097151.0f), iy = (uint32_t)fminf(fmaxf(fy, 0.0f), 2097151.0f), iz = (uint32_t)fminf(fmaxf(fz, 0.0f), 2097151.0f);
  keys[i] = spread21(ix) | (spread21(iy) << 1) | (spread21(iz) << 2);
  vals[i] = i;
}
__global__ __launch_bounds__(256) void morton_gather(const PrimRef* src, const uint32_t* order, uint32_t n, PrimRef* dst, uint2* finalIds) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const PrimRef p = load_prim(src + order[i]);
  store_prim(dst + i, p);
  finalIds[i] = make_uint2(p.geom, p.prim);
}
// length of the common prefix of the (code, index) pairs i and j; -1 outside the array
__device__ __forceinline__ int lbvh_delta(const unsigned long long* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const unsigned long long a = keys[i], b = keys[j];
  return a != b ? __clzll((long long)(a ^ b)) : 64 + __clz(i ^ j);
}
__global__ __launch_bounds__(256) void lbvh_hierarchy(const unsigned long long* keys, uint32_t n, BNode* bnodes, uint32_t* parent) {
  const int i = (int)(blockIdx.x * 256u + threadIdx.x), N = (int)n;
  if (i >= N - 1) return;
  const int d = lbvh_delta(keys, N, i, i + 1) - lbvh_delta(keys, N, i, i - 1) >= 0 ? 1 : -1;
  const int dmin = lbvh_delta(keys, N, i, i - d);
  int lmax = 2;
  while (lbvh_delta(keys, N, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1) if (lbvh_delta(keys, N, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = lbvh_delta(keys, N, i, j);
  int sft = 0;
  for (int t = (l + 1) >> 1; ; t = (t + 1) >> 1) { if (lbvh_delta(keys, N, i, i + (sft + t) * d) > dnode) sft += t; if (t == 1) break; }
  const int gamma = i + sft * d + min(d, 0);
  const int first = min(i, j), last = max(i, j);
  const uint32_t left = gamma == first ? (uint32_t)(N - 1 + gamma) : (uint32_t)gamma;
  const uint32_t right = gamma + 1 == last ? (uint32_t)(N - 1 + gamma + 1) : (uint32_t)(gamma + 1);
  ((uint4*)(bnodes + i))[2] = make_uint4(left, right, __float_as_uint(__builtin_inff()), 0u);   // splitSah = inf: <= max_leaf triangles always form a leaf slot
  ((uint32_t*)(bnodes + i))[3] = (uint32_t)first; ((uint32_t*)(bnodes + i))[7] = (uint32_t)last + 1u;
  parent[left] = (uint32_t)i; parent[right] = (uint32_t)i;
}
// Boxes from the leaves up: the second child to arrive at a node (atomic flag) merges the two child boxes and goes on.  The
// two children are usually processed by different CUs, often on different XCDs, whose L2s are not coherent with each other: the
// boxes are therefore written and read with system-scope (sc0 sc1) 16-byte accesses, which go through to memory, and a store is made to
// complete (s_waitcnt vmcnt(0)) before the flag is touched -- the "sc0 sc1 on both sides" hand-off of MI355X_MICROARCH.md; a
// __threadfence() per step would write back the whole L2 each time (microseconds) and a plain load may return a stale line.
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_sys(void* p, v4f v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory"); }
// the two 16-byte halves (lo|begin, hi|end) of two BNodes, system scope, one wait for the four loads
__device__ __forceinline__ void ld_boxes_sys(const BNode* x, const BNode* y, v4f& xl, v4f& xh, v4f& yl, v4f& yh) {
  asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc0 sc1\n\t"
               "global_load_dwordx4 %2, %5, off sc0 sc1\n\tglobal_load_dwordx4 %3, %5, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(xl), "=&v"(xh), "=&v"(yl), "=&v"(yh) : "v"(x), "v"(y) : "memory");
}
__global__ __launch_bounds__(256) void lbvh_bounds(const PrimRef* prims, uint32_t n, BNode* bnodes, const uint32_t* parent, uint32_t* flags, Counters* ctr) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= n) return;
  const PrimRef p = load_prim(prims + j);
  uint32_t id = n - 1u + j;
  {
    v4f l = {p.lo[0], p.lo[1], p.lo[2], __uint_as_float(j)}, h = {p.hi[0], p.hi[1], p.hi[2], __uint_as_float(j + 1u)};
    st16_sys(bnodes + id, l); st16_sys((char*)(bnodes + id) + 16, h);
    ((uint4*)(bnodes + id))[2] = make_uint4(NIL, NIL, __float_as_uint(__builtin_inff()), 0u);   // links: only read by later kernels
  }
  if (j == 0u) ctr->numBLeaves = n;
  while (id != 0u) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // my box is in memory before my arrival is announced
    const uint32_t par = parent[id];
    if (atomicAdd(&flags[par], 1u) == 0u) return;               // first to arrive: the sibling's thread takes over
    const uint32_t* pw = (const uint32_t*)(bnodes + par);       // links and range: written by lbvh_hierarchy, never changed here
    const uint32_t l = pw[8], r = pw[9], first = pw[3], end = pw[7];
    v4f al, ah, bl, bh;
    ld_boxes_sys(bnodes + l, bnodes + r, al, ah, bl, bh);
    v4f lo = {fminf(al.x, bl.x), fminf(al.y, bl.y), fminf(al.z, bl.z), __uint_as_float(first)};
    v4f hi = {fmaxf(ah.x, bh.x), fmaxf(ah.y, bh.y), fmaxf(ah.z, bh.z), __uint_as_float(end)};
    st16_sys(bnodes + par, lo); st16_sys((char*)(bnodes + par) + 16, hi);
    id = par;
  }
}

// -------------------------------------------------------------------------------- K4 wide collapse
__device__ __forceinline__ float bnode_area(const BNode& b) { return half_area3(b.hi[0] - b.lo[0], b.hi[1] - b.lo[1], b.hi[2] - b.lo[2]); }

// leaf-vs-split decision of BuilderT::recurse (bvh_builder_sah.h:229-236)
__device__ __forceinline__ bool make_leaf(const BNode& b, const Params& prm) {
  const uint32_t n = b.end - b.begin;
  if (n <= prm.minLeaf || b.left == NIL) return true;
  if (n > prm.maxLeaf) return false;
  const float A = bnode_area(b);
  const float leafSAH = prm.intCost * (A * (float)((n + (1u << prm.shift) - 1u) >> prm.shift));
  const float splitSAH = prm.travCost * A + prm.intCost * b.splitSah;
  return leafSAH <= splitSAH;
}
// heuristic.deterministic_order: sort the leaf's triangles by (primID << 32 | geomID)
__device__ void sort_leaf(uint2* ids, uint32_t b, uint32_t e) {
  for (uint32_t i = b + 1; i < e; i++) {
    const uint2 x = ids[i]; const unsigned long long kx = ((unsigned long long)x.y << 32) | x.x;
    uint32_t j = i;
    while (j > b) { const uint2 y = ids[j - 1]; if ((((unsigned long long)y.y << 32) | y.x) <= kx) break; ids[j] = y; j--; }
    ids[j] = x;
  }
}

// ---- The collapse runs level by level (children of a node get consecutive indices, so numbering is breadth first), three
// kernels per level, EIGHT LANES PER NODE (lane = child, later = slot), eight nodes per wavefront:
//   wide_plan   children of every node of the level: the reference's greedy "split the child with the largest half-area until
//               8 children" (bvh_builder_sah.h:247-272) on the binary tree + leaf-vs-split SAH test; each child becomes a leaf
//               slot (<= 3 triangles) or an inner slot and is PLACED in the slot whose octant fits its position (greedy
//               assignment on dot(child centre - node centre, octant signs)); the plan (child per slot, inner/leaf masks) and
//               the node's counts (#inner children, #leaf triangles) are stored
//   wide_scan   exclusive scan of the counts in item order -> first child index / first triangle index of every node.  No
//               atomic counter decides an index: the layout of the tree is identical on every run and on every GPU.
//   wide_emit   quantises the child boxes (8 bits, verified conservative in fp32), writes the 80-byte node, the next level's
//               work items, and the leaf triangles' ids in (primID, geomID) order
// The first version used one thread per node (206 VGPRs, 2 waves/SIMD, atomics for the numbering: 1.6 ms of a 9.4 ms commit).
struct WidePlan { uint32_t ch[8]; uint32_t imask, leafMask, nch, pad; };   // by slot; NIL = empty slot

__global__ void wide_root(WideItem* items, Counters* ctr) {
  items[0].bnode = 0; items[0].node = 0;                       // the root is always CNode 0
  ctr->rootRef = 0; ctr->numWide = 1; ctr->wideCount[0] = 1; ctr->wideCount[1] = 0; ctr->wideDepth = 0; ctr->lvlStart[0] = 0; ctr->numLeaves = 0; ctr->numTrisOut = 0; ctr->sahFixed = 0ull;
}

template <typename T> __device__ __forceinline__ T grp_get(T v, uint32_t lane, uint32_t idx) { return __shfl(v, (int)((lane & ~7u) | idx), 64); }
__device__ __forceinline__ float grp_min(float v) { v = fminf(v, __shfl_xor(v, 1, 64)); v = fminf(v, __shfl_xor(v, 2, 64)); return fminf(v, __shfl_xor(v, 4, 64)); }
__device__ __forceinline__ float grp_max(float v) { v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64)); return fmaxf(v, __shfl_xor(v, 4, 64)); }
__device__ __forceinline__ float grp_sum(float v) { v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); return v + __shfl_xor(v, 4, 64); }
// argmax over the 8 lanes of a group; ties go to the lower index (the serial formulation keeps the first maximum)
__device__ __forceinline__ void grp_argmax(float& v, uint32_t& idx) {
  for (int o = 1; o < 8; o <<= 1) {
    const float ov = __shfl_xor(v, o, 64); const uint32_t oi = (uint32_t)__shfl_xor((int)idx, o, 64);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
}
__device__ __forceinline__ BNode load_bnode(const BNode* p) {
  const float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1]; const uint4 c = ((const uint4*)p)[2];
  BNode r; r.lo[0] = a.x; r.lo[1] = a.y; r.lo[2] = a.z; r.begin = __float_as_uint(a.w); r.hi[0] = b.x; r.hi[1] = b.y; r.hi[2] = b.z; r.end = __float_as_uint(b.w);
  r.left = c.x; r.right = c.y; r.splitSah = __uint_as_float(c.z); r.pad = 0; return r;
}

__global__ __launch_bounds__(64) void wide_plan(const WideItem* items, const BNode* bnodes, WidePlan* plans, uint2* itemCnt, uint2* groupSum,
                                                Counters* ctr, Params prm, uint32_t parity, float rootArea) {
  const uint32_t numItems = ctr->wideCount[parity];
  const uint32_t lane = threadIdx.x, c = lane & 7u, g = lane >> 3;
  unsigned long long sahAcc = 0ull; uint32_t leafAcc = 0u;      // per-wave partial sums: one atomic per wave at the end (a same-address atomic costs ~2 ns)
  for (uint32_t base = blockIdx.x * 8u; base < numItems; base += gridDim.x * 8u) {
    const uint32_t t = base + g; const bool valid = t < numItems;
    WideItem it; it.bnode = 0; it.node = 0; if (valid) it = items[t];
    const BNode root = load_bnode(bnodes + it.bnode);
    // ---- children: lane c holds child c
    uint32_t nch, my = NIL; BNode mb = root;
    if (root.left == NIL || make_leaf(root, prm)) { nch = 1; if (c == 0u) my = it.bnode; }        // only the tree root can be a leaf itself
    else { nch = 2; if (c < 2u) { my = c == 0u ? root.left : root.right; mb = load_bnode(bnodes + my); } }
    bool done = !valid || nch == 1u;
    while (__ballot(!done) != 0ull) {
      float ar = -__builtin_inff(); uint32_t bi = c;
      if (!done && c < nch && !(mb.end - mb.begin <= prm.minLeaf || mb.left == NIL)) ar = bnode_area(mb);
      grp_argmax(ar, bi);
      if (ar == -__builtin_inff()) done = true;
      const uint32_t l = grp_get(mb.left, lane, bi), r = grp_get(mb.right, lane, bi);
      if (!done) {
        if (c == bi) { my = l; mb = load_bnode(bnodes + l); }
        else if (c == nch) { my = r; mb = load_bnode(bnodes + r); }
        nch++;
        if (nch == 8u) done = true;
      }
    }
    const bool has = valid && c < nch;
    const bool leaf = has && make_leaf(mb, prm);
    const uint32_t cnt = has ? mb.end - mb.begin : 0u;
    float lo[3], hi[3], olo[3], ohi[3];
    for (int d = 0; d < 3; d++) { lo[d] = has ? mb.lo[d] : __builtin_inff(); hi[d] = has ? mb.hi[d] : -__builtin_inff(); olo[d] = grp_min(lo[d]); ohi[d] = grp_max(hi[d]); }
    // SAH of the finished tree (statistics only), accumulated in fixed point so that the sum does not depend on the order
    {
      const float A = has ? bnode_area(mb) : 0.0f;
      const float sa = grp_sum(has ? (leaf ? prm.intCost * A * (float)((cnt + (1u << prm.shift) - 1u) >> prm.shift) : prm.travCost * A) : 0.0f);
      if (valid && c == 0u && rootArea > 0.0f) sahAcc += (unsigned long long)((double)(sa / rootArea) * 16777216.0);
    }
    // ---- slot assignment: repeatedly take the (child, slot) pair with the largest dot(centre offset, octant signs)
    uint32_t slot = NIL;
    {
      float v[8];
      const float cx = has ? (lo[0] + hi[0]) - (olo[0] + ohi[0]) : 0.0f, cy = has ? (lo[1] + hi[1]) - (olo[1] + ohi[1]) : 0.0f, cz = has ? (lo[2] + hi[2]) - (olo[2] + ohi[2]) : 0.0f;   // 2 x centre offset
      for (uint32_t q = 0; q < 8u; q++) v[q] = ((q & 1u) ? cx : -cx) + ((q & 2u) ? cy : -cy) + ((q & 4u) ? cz : -cz);
      uint32_t freeSlots = 0xFFu;
      for (uint32_t k = 0; k < 8u; k++) {
        const bool pending = has && slot == NIL;
        float best = -__builtin_inff(); uint32_t bs = 8u;
        if (pending) for (uint32_t q = 0; q < 8u; q++) if (((freeSlots >> q) & 1u) && (v[q] > best || bs == 8u)) { best = v[q]; bs = q; }
        // a pending child always has a finite value; -inf means "nothing pending in this lane"
        float bv = pending ? fmaxf(best, -3.0e38f) : -__builtin_inff(); uint32_t bi = c;
        grp_argmax(bv, bi);
        const uint32_t ws = grp_get(bs, lane, bi);
        if (bv != -__builtin_inff()) { if (c == bi) slot = ws; freeSlots &= ~(1u << ws); }
      }
    }
    // ---- transpose: lane s now speaks for slot s
    uint32_t childAt = NIL;
    for (uint32_t i = 0; i < 8u; i++) { const uint32_t so = grp_get(slot, lane, i); if (so == c) childAt = i; }
    const uint32_t src = childAt == NIL ? c : childAt;
    const uint32_t sCh = grp_get(my, lane, src), sCnt = grp_get(cnt, lane, src); const bool sLeaf = grp_get((int)leaf, lane, src) != 0;
    const bool sHas = childAt != NIL;
    const uint32_t gshift = lane & ~7u;
    const uint32_t imask = (uint32_t)((__ballot(sHas && !sLeaf) >> gshift) & 0xFFull), leafMask = (uint32_t)((__ballot(sHas && sLeaf) >> gshift) & 0xFFull);
    uint32_t nTri = (sHas && sLeaf) ? sCnt : 0u;
    nTri += (uint32_t)__shfl_xor((int)nTri, 1, 64); nTri += (uint32_t)__shfl_xor((int)nTri, 2, 64); nTri += (uint32_t)__shfl_xor((int)nTri, 4, 64);
    const uint32_t nInner = (uint32_t)__popc(imask);
    if (valid) {
      plans[t].ch[c] = sHas ? sCh : NIL;
      if (c == 0u) { plans[t].imask = imask; plans[t].leafMask = leafMask; plans[t].nch = nch; plans[t].pad = 0u; }
    }
    // ---- counts: exclusive prefix over the 8 items of this wave, wave total for the scan
    const uint32_t ci = valid ? nInner : 0u, ct = valid ? nTri : 0u;          // every lane of a group holds the same pair
    uint32_t xi = ci, xt = ct;
    for (int o = 8; o < 64; o <<= 1) { const uint32_t ui = (uint32_t)__shfl_up((int)xi, o, 64), ut = (uint32_t)__shfl_up((int)xt, o, 64); if (lane >= (uint32_t)o) { xi += ui; xt += ut; } }
    if (valid && c == 0u) itemCnt[t] = make_uint2(xi - ci, xt - ct);
    const uint32_t ti = (uint32_t)__shfl((int)xi, 63, 64), tt = (uint32_t)__shfl((int)xt, 63, 64);
    if (lane == 0u) groupSum[base >> 3] = make_uint2(ti, tt);
    leafAcc += (uint32_t)__popcll(__ballot(valid && sHas && sLeaf));
  }
  for (int o = 8; o < 64; o <<= 1) sahAcc += (unsigned long long)__shfl_xor((long long)sahAcc, o, 64);   // lanes with c == 0 hold the partial sums
  if (lane == 0u) { if (sahAcc) atomicAdd(&ctr->sahFixed, sahAcc); if (leafAcc) atomicAdd(&ctr->numLeaves, leafAcc); }
}

// one block: exclusive scan of the per-wave totals in item order; publishes the level's bases and the next level's item count
__global__ __launch_bounds__(1024) void wide_scan(uint2* groupSum, Counters* ctr, uint32_t parity, uint32_t maxNodes) {
  __shared__ uint2 s_part[1024];
  const uint32_t numItems = ctr->wideCount[parity], numGroups = (numItems + 7u) / 8u;
  const uint32_t tid = threadIdx.x, per = (numGroups + 1023u) / 1024u, b = min(tid * per, numGroups), e = min(b + per, numGroups);
  uint2 sum = make_uint2(0, 0);
  for (uint32_t i = b; i < e; i++) { const uint2 x = groupSum[i]; sum.x += x.x; sum.y += x.y; }
  s_part[tid] = sum; __syncthreads();
  for (uint32_t o = 1; o < 1024u; o <<= 1) {                    // Hillis-Steele inclusive scan
    uint2 x = make_uint2(0, 0); if (tid >= o) x = s_part[tid - o];
    __syncthreads(); if (tid >= o) { s_part[tid].x += x.x; s_part[tid].y += x.y; } __syncthreads();
  }
  const uint2 total = s_part[1023];
  uint2 run = tid ? s_part[tid - 1] : make_uint2(0, 0);
  for (uint32_t i = b; i < e; i++) { const uint2 x = groupSum[i]; groupSum[i] = run; run.x += x.x; run.y += x.y; }
  __syncthreads();
  if (tid == 0) {
    ctr->lvlNodeBase = ctr->numWide; ctr->lvlTriBase = ctr->numTrisOut;
    if (numItems) { ctr->wideDepth++; if (ctr->wideDepth < 64u) ctr->lvlStart[ctr->wideDepth] = ctr->numWide; }
    if ((uint64_t)ctr->numWide + total.x > maxNodes) { ctr->overflow = 2u; ctr->wideCount[parity ^ 1u] = 0u; }
    else { ctr->numWide += total.x; ctr->numTrisOut += total.y; ctr->wideCount[parity ^ 1u] = total.x; }
  }
}

// Quantises the child boxes of 8 nodes at once (lane = 8 * node + slot): plane = org + q * 2^(e-127), lower planes rounded down, upper planes
// rounded up, verified in fp32.  Shared by wide_emit and refit_level so that a refitted node is what a build would have written for the same boxes.
__device__ __forceinline__ void quantise_slots(bool has, uint32_t lane, const float (&lo)[3], const float (&hi)[3], const float (&olo)[3], const float (&ohi)[3],
                                               uint32_t (&ex)[3], uint32_t (&qa)[3], uint32_t (&qb)[3]) {
  // ---- quantise: plane = org + q * 2^(e-127), lower rounded down, upper rounded up, verified in fp32
  for (int d = 0; d < 3; d++) {
    const float ext = ohi[d] - olo[d];
    int e = 1;                                                // biased exponent, scale = 2^(e-127)
    if (ext > 0.0f) { int fe; frexpf(ext / 255.0f, &fe); e = fe + 127; if (e < 1) e = 1; if (e > 254) e = 254; }
    for (;;) {                                                // grow the scale until every upper plane of the node fits in 8 bits
      const float sc = __uint_as_float((uint32_t)e << 23);
      bool fits = true;
      if (has) { float q = ceilf((hi[d] - olo[d]) / sc); while (fmaf(q, sc, olo[d]) < hi[d]) q += 1.0f; fits = q <= 255.0f; }
      const bool grpFits = ((__ballot(!fits) >> (lane & ~7u)) & 0xFFull) == 0ull;
      const bool stop = grpFits || e >= 254;
      if (!stop) e++;
      if (__ballot(!stop) == 0ull) break;
    }
    ex[d] = (uint32_t)e;
    qa[d] = 255u; qb[d] = 0u;                                 // empty slot: inverted box, never hit
    if (has) {
      const float sc = __uint_as_float(ex[d] << 23);
      float a = floorf((lo[d] - olo[d]) / sc); if (a < 0.0f) a = 0.0f; if (a > 255.0f) a = 255.0f;
      while (a > 0.0f && fmaf(a, sc, olo[d]) > lo[d]) a -= 1.0f;
      float b = ceilf((hi[d] - olo[d]) / sc); if (b < 0.0f) b = 0.0f;
      while (b < 255.0f && fmaf(b, sc, olo[d]) < hi[d]) b += 1.0f;
      if (b > 255.0f) b = 255.0f;
      qa[d] = (uint32_t)a; qb[d] = (uint32_t)b;
    }
  }
}

__global__ __launch_bounds__(64) void wide_emit(const WideItem* items, const BNode* bnodes, const WidePlan* plans, const uint2* itemCnt, const uint2* groupSum,
                                                CNode* nodes, uint2* finalIds, uint2* outIds, WideItem* next, const Counters* ctr, uint32_t parity) {
  __shared__ uint32_t s_node[8][20];
  const uint32_t numItems = ctr->wideCount[parity];
  if (ctr->overflow) return;
  const uint32_t nodeBase = ctr->lvlNodeBase, triLvl = ctr->lvlTriBase;
  const uint32_t lane = threadIdx.x, s = lane & 7u, g = lane >> 3;
  for (uint32_t base = blockIdx.x * 8u; base < numItems; base += gridDim.x * 8u) {
    const uint32_t t = base + g; const bool valid = t < numItems;
    uint32_t ch = NIL, imask = 0, leafMask = 0, node = 0; uint2 ofs = make_uint2(0, 0);
    if (valid) {
      ch = plans[t].ch[s]; imask = plans[t].imask; leafMask = plans[t].leafMask; node = items[t].node;
      const uint2 a = groupSum[base >> 3], b = itemCnt[t]; ofs = make_uint2(a.x + b.x, a.y + b.y);
    }
    const bool has = ch != NIL, inner = ((imask >> s) & 1u) != 0u, leaf = ((leafMask >> s) & 1u) != 0u;
    BNode cb{}; if (has) cb = load_bnode(bnodes + ch);
    float lo[3], hi[3], olo[3], ohi[3];
    for (int d = 0; d < 3; d++) { lo[d] = has ? cb.lo[d] : __builtin_inff(); hi[d] = has ? cb.hi[d] : -__builtin_inff(); olo[d] = grp_min(lo[d]); ohi[d] = grp_max(hi[d]); }
    const uint32_t cnt = leaf ? cb.end - cb.begin : 0u;
    // numbering: inner children consecutive in slot order, leaf triangles consecutive in slot order
    const uint32_t below = (1u << s) - 1u;
    const uint32_t childBase = nodeBase + ofs.x, nextBase = ofs.x, triBase = triLvl + ofs.y;
    uint32_t triOfs;                                            // exclusive prefix of the leaf counts over the slots
    { uint32_t x = cnt; for (int o = 1; o < 8; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)x, o, 64); if (s >= (uint32_t)o) x += u; } triOfs = x - cnt; }
    if (inner) { const uint32_t j = (uint32_t)__popc(imask & below); next[nextBase + j].bnode = ch; next[nextBase + j].node = childBase + j; }
    if (leaf) {
      sort_leaf(finalIds, cb.begin, cb.end);
      for (uint32_t j = cb.begin; j < cb.end; j++) outIds[triBase + triOfs + (j - cb.begin)] = finalIds[j];
    }
    uint32_t ex[3], qa[3], qb[3];
    quantise_slots(has, lane, lo, hi, olo, ohi, ex, qa, qb);
    const uint32_t meta = !has ? 0u : (leaf ? ((((1u << cnt) - 1u) << 5) | triOfs) : ((1u << 5) | (24u + s)));
    // ---- assemble the 80-byte node in LDS (the bytes of a word come from 4 lanes), 5 lanes store it
    __syncthreads();
    uint8_t* nb = (uint8_t*)&s_node[g][0];
    nb[24 + s] = (uint8_t)meta;
    for (int d = 0; d < 3; d++) { nb[32 + d * 8 + s] = (uint8_t)qa[d]; nb[56 + d * 8 + s] = (uint8_t)qb[d]; }
    if (s == 0u) {
      s_node[g][0] = __float_as_uint(olo[0]); s_node[g][1] = __float_as_uint(olo[1]); s_node[g][2] = __float_as_uint(olo[2]);
      s_node[g][3] = ex[0] | (ex[1] << 8) | (ex[2] << 16) | (imask << 24);
      s_node[g][4] = imask ? childBase : 0u; s_node[g][5] = leafMask ? triBase : 0u;
    }
    __syncthreads();
    if (valid && s < 5u) ((uint4*)(nodes + node))[s] = ((const uint4*)&s_node[g][0])[s];
  }
}

// --------------------------------------------------------------------------------- K5 tri_records
__global__ __launch_bounds__(256) void tri_records(const uint2* finalIds, uint32_t n, const GeomDesc* geoms, TriRec* out, uint32_t robust) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  uint2 id = finalIds[i];
  const GeomDesc g = geoms[id.x];
  uint32_t i0, i1, i2, pid;
  prim_indices(g, id.y, i0, i1, i2, pid);
  id.y = pid;                                                   // quads: quad index, bit 31 = second half (cleared again when a hit is written)
  const float* a = (const float*)(g.verts + (size_t)i0 * g.vstride);
  const float* b = (const float*)(g.verts + (size_t)i1 * g.vstride);
  const float* c = (const float*)(g.verts + (size_t)i2 * g.vstride);
  float4* o = (float4*)(out + i);
  if (robust) {   // TriangleMv: the three vertices (kernels/geometry/trianglev.h), same 48-byte record
    o[0] = make_float4(a[0], a[1], a[2], b[0]);
    o[1] = make_float4(b[1], b[2], c[0], c[1]);
    o[2] = make_float4(c[2], __uint_as_float(id.y), __uint_as_float(g.geomID), __uint_as_float(g.mask));
    return;
  }
  // TriangleM ctor: e1 = v0 - v1, e2 = v2 - v0 (kernels/geometry/triangle.h:40-41)
  o[0] = make_float4(a[0], a[1], a[2], a[0] - b[0]);
  o[1] = make_float4(a[1] - b[1], a[2] - b[2], c[0] - a[0], c[1] - a[1]);
  o[2] = make_float4(c[2] - a[2], __uint_as_float(id.y), __uint_as_float(g.geomID), __uint_as_float(g.mask));
}

// --------------------------------------------------------------------------------- refit (RTC_BUILD_QUALITY_REFIT, kernels/bvh/bvh_refit.cpp)
// The topology of the tree stays; the triangle records are rewritten from the moved vertices (tri_records) and the boxes are
// recomputed bottom-up, one launch per level of the wide tree (nodes are numbered breadth first, so a level is a contiguous range).
// Eight lanes per node as in wide_emit: lane = child slot; a leaf slot bounds its <= 3 triangles from the vertex buffers (the same
// min/max as primref_gen), an inner slot takes the exact box its child wrote one launch earlier; the node is re-quantised with the
// builder's own routine.  A triangle that has become invalid (non-finite / huge coordinate) raises *flag: the caller rebuilds.
__global__ __launch_bounds__(64) void refit_level(CNode* nodes, float4* boxes, const uint2* ids, const GeomDesc* geoms, uint32_t first, uint32_t count, uint32_t* flag) {
  __shared__ uint32_t s_node[8][20];
  const uint32_t lane = threadIdx.x, s = lane & 7u, g = lane >> 3;
  for (uint32_t base = blockIdx.x * 8u; base < count; base += gridDim.x * 8u) {
    const uint32_t t = base + g; const bool valid = t < count;
    const uint32_t node = first + (valid ? t : 0u);
    __syncthreads();
    if (valid && s < 5u) ((uint4*)&s_node[g][0])[s] = ((const uint4*)(nodes + node))[s];
    __syncthreads();
    const uint8_t* nbr = (const uint8_t*)&s_node[g][0];
    const uint32_t meta = valid ? nbr[24 + s] : 0u, imask = s_node[g][3] >> 24, childBase = s_node[g][4], triBase = s_node[g][5];
    const bool has = meta != 0u, inner = has && ((imask >> s) & 1u) != 0u;
    float lo[3], hi[3], olo[3], ohi[3];
    for (int d = 0; d < 3; d++) { lo[d] = __builtin_inff(); hi[d] = -__builtin_inff(); }
    if (inner) {
      const uint32_t c = childBase + (uint32_t)__popc(imask & ((1u << s) - 1u));
      const float4 a = boxes[2u * c], b = boxes[2u * c + 1u];
      lo[0] = a.x; lo[1] = a.y; lo[2] = a.z; hi[0] = b.x; hi[1] = b.y; hi[2] = b.z;
    } else if (has) {
      const uint32_t cnt = (uint32_t)__popc(meta >> 5), t0 = triBase + (meta & 31u);
      bool ok = true;
      for (uint32_t k = 0; k < cnt; k++) {
        const uint2 id = ids[t0 + k];
        const GeomDesc gd = geoms[id.x];
        uint32_t i0, i1, i2, pid;
        prim_indices(gd, id.y, i0, i1, i2, pid);
        const float* a = (const float*)(gd.verts + (size_t)i0 * gd.vstride);
        const float* b = (const float*)(gd.verts + (size_t)i1 * gd.vstride);
        const float* c = (const float*)(gd.verts + (size_t)i2 * gd.vstride);
        for (int d = 0; d < 3; d++) {
          const float x = a[d], y = b[d], z = c[d];
          ok = ok && valid_f(x) && valid_f(y) && valid_f(z);
          lo[d] = fminf(lo[d], fminf(fminf(x, y), z)); hi[d] = fmaxf(hi[d], fmaxf(fmaxf(x, y), z));
        }
        if (gd.quad) {
          const uint32_t* q = (const uint32_t*)(gd.idx + (size_t)(id.y >> 1) * gd.istride);
          const float* o4 = (const float*)(gd.verts + (size_t)((id.y & 1u) ? q[0] : q[2]) * gd.vstride);
          ok = ok && valid_f(o4[0]) && valid_f(o4[1]) && valid_f(o4[2]);
        }
      }
      if (!ok) { atomicOr(flag, 1u); for (int d = 0; d < 3; d++) { lo[d] = 0.0f; hi[d] = 0.0f; } }
    }
    for (int d = 0; d < 3; d++) { olo[d] = grp_min(lo[d]); ohi[d] = grp_max(hi[d]); }
    uint32_t ex[3], qa[3], qb[3];
    quantise_slots(has, lane, lo, hi, olo, ohi, ex, qa, qb);
    __syncthreads();
    uint8_t* nb = (uint8_t*)&s_node[g][0];
    for (int d = 0; d < 3; d++) { nb[32 + d * 8 + s] = (uint8_t)qa[d]; nb[56 + d * 8 + s] = (uint8_t)qb[d]; }
    if (s == 0u) {
      s_node[g][0] = __float_as_uint(olo[0]); s_node[g][1] = __float_as_uint(olo[1]); s_node[g][2] = __float_as_uint(olo[2]);
      s_node[g][3] = ex[0] | (ex[1] << 8) | (ex[2] << 16) | (imask << 24);
      if (valid) { boxes[2u * node] = make_float4(olo[0], olo[1], olo[2], 0.0f); boxes[2u * node + 1u] = make_float4(ohi[0], ohi[1], ohi[2], 0.0f); }
    }
    __syncthreads();
    if (valid && s < 5u) ((uint4*)(nodes + node))[s] = ((const uint4*)&s_node[g][0])[s];
  }
}

// --------------------------------------------------------------------------------- instanced scenes: object trees copied behind the top tree
// One thread per node: the 80 bytes are copied, child / triangle base indices moved by the object's offset in the combined arrays.
__global__ __launch_bounds__(256) void rebase_nodes(const uint4* src, uint4* dst, uint32_t n, uint32_t nodeOfs, uint32_t triOfs) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  uint4 w0 = src[5u * i], w1 = src[5u * i + 1u];
  w1.x += nodeOfs; w1.y += triOfs;
  dst[5u * i] = w0; dst[5u * i + 1u] = w1; dst[5u * i + 2u] = src[5u * i + 2u]; dst[5u * i + 3u] = src[5u * i + 3u]; dst[5u * i + 4u] = src[5u * i + 4u];
}

// Build scratch comes from a per-device arena that survives the commit: rtcCommitScene is timed on the wall clock
// (tutorials/buildbench/buildbench_device.cpp:385-387) and ~20 hipMalloc/hipFree pairs cost 3 ms of a 10.7 ms commit of 4.76 M
// triangles.  Blocks are kept until mi355_release_build_scratch(); commits on one device are serialised by the arena's mutex
// (the reference's rtcCommitScene is blocking as well and runs on its own worker pool).
struct Arena {
  struct Block { char* p; size_t cap, used; };
  std::mutex mtx;
  std::vector<Block> blocks;
  void reset() { for (auto& b : blocks) b.used = 0; }
  hipError_t take(size_t bytes, void** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    for (auto& b : blocks) if (b.cap - b.used >= bytes) { *out = b.p + b.used; b.used += bytes; return hipSuccess; }
    Block nb; nb.cap = bytes > ((size_t)64 << 20) ? bytes : ((size_t)64 << 20); nb.used = bytes; nb.p = nullptr;
    const hipError_t e = hipMalloc((void**)&nb.p, nb.cap);
    if (e != hipSuccess) return e;
    blocks.push_back(nb); *out = nb.p; return hipSuccess;
  }
  void release() { for (auto& b : blocks) hipFree(b.p); blocks.clear(); }
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
  if (hipMalloc(&sc.spill, trace_spill_bytes(numCUs, info.depth)) != hipSuccess) return nullptr;
  if (hipMalloc((void**)&sc.stats, 128) != hipSuccess) return nullptr;
  sc.enqueue = new std::mutex;
  return &(scratch[s] = sc);
}
Bvh::~Bvh() {
  hipSetDevice(device);
  for (auto& kv : scratch) { hipFree(kv.second.counter); hipFree(kv.second.spill); hipFree(kv.second.stats); delete kv.second.enqueue; }
  if (d_nodes) hipFree(d_nodes);
  if (d_tris) hipFree(d_tris);
  if (d_ids) hipFree(d_ids);
  if (d_insts) hipFree(d_insts);
}

static int build_impl(int device, const mi355_mesh* meshes, uint32_t numMeshes, const mi355_build_params* bp, hipStream_t st, Bvh** out) {
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

  std::vector<GeomDesc> gd; uint64_t total = 0;
  for (uint32_t i = 0; i < numMeshes; i++) {
    const mi355_mesh& m = meshes[i];
    if (m.num_triangles == 0) continue;
    if (m.vertex_stride < 12 || (m.vertex_stride & 3) || m.index_stride < (m.quads ? 16u : 12u) || (m.index_stride & 3)) return set_error(hipErrorInvalidValue, "buffer stride");
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

  DevBuf<GeomDesc> dGeoms; HIP_TRY(dGeoms.alloc(gd.size()));
  HIP_TRY(hipMemcpyAsync(dGeoms.p, gd.data(), gd.size() * sizeof(GeomDesc), hipMemcpyHostToDevice, st));
  const bool presplit = prm.quality == 2u;                     // RTC_BUILD_QUALITY_HIGH: up to 20 % more references than triangles
  const uint32_t splitBudget = presplit ? (uint32_t)((double)N * (bp->split_factor > 1.0f ? (double)bp->split_factor - 1.0 : 0.2)) : 0u;
  const uint64_t cap64 = (uint64_t)N + splitBudget;
  if (cap64 >= (1ull << 31)) return set_error(hipErrorInvalidValue, "more than 2^31 references are not supported by the 32-bit triangle index");
  const uint32_t NC = (uint32_t)cap64;                        // capacity of every per-reference array
  const uint32_t maxSegs = NC / prm.small + 1024u, maxSmall = 8u * (NC / prm.small) + 1024u, maxChunks = NC / CHUNK + maxSegs + 16u;
  const uint32_t maxWide = NC + 64u;
  DevBuf<PrimRef> bufA, bufB; DevBuf<uint2> finalIds; DevBuf<BNode> bnodes; DevBuf<Seg> segs0, segs1; DevBuf<uint32_t> bins;
  DevBuf<Chunk> chunks; DevBuf<SmallEntry> small; DevBuf<Counters> ctr; DevBuf<WideItem> w0, w1; DevBuf<CNode> wnodes; DevBuf<uint2> outIds;
  HIP_TRY(bufA.alloc(NC)); HIP_TRY(bufB.alloc(NC)); HIP_TRY(finalIds.alloc(NC)); HIP_TRY(bnodes.alloc(2ull * NC + 2));
  HIP_TRY(segs0.alloc(maxSegs)); HIP_TRY(segs1.alloc(maxSegs)); HIP_TRY(bins.alloc((size_t)maxSegs * BINS_WORDS));
  HIP_TRY(chunks.alloc(maxChunks)); HIP_TRY(small.alloc(maxSmall)); HIP_TRY(ctr.alloc(1));
  HIP_TRY(w0.alloc(maxWide)); HIP_TRY(w1.alloc(maxWide)); HIP_TRY(wnodes.alloc(maxWide)); HIP_TRY(outIds.alloc(NC));
  const uint32_t maxLevelItems = NC / 2u + 64u;                 // the nodes of one level are disjoint sub-trees of >= 2 triangles each
  DevBuf<WidePlan> plans; DevBuf<uint2> itemCnt, groupSum;
  HIP_TRY(plans.alloc(maxLevelItems)); HIP_TRY(itemCnt.alloc(maxLevelItems)); HIP_TRY(groupSum.alloc(maxLevelItems / 8u + 16u));

  hipEvent_t ev0, ev1; HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1));
  struct EvGuard { hipEvent_t a, b; ~EvGuard() { hipEventDestroy(a); hipEventDestroy(b); } } evg{ev0, ev1};
  HIP_TRY(hipEventRecord(ev0, st));

  Counters h{}; for (int k = 0; k < 12; k++) h.bounds[k] = (k % 6 < 3) ? ENC_POS_INF : ENC_NEG_INF; h.rootRef = MI355_EMPTY_REF;
  HIP_TRY(hipMemcpyAsync(ctr.p, &h, sizeof(h), hipMemcpyHostToDevice, st));
  const uint32_t genBlocks = (N + 255u) / 256u < 4096u ? (N + 255u) / 256u : 4096u;
  hipLaunchKernelGGL(primref_gen, dim3(genBlocks), dim3(256), 0, st, dGeoms.p, (uint32_t)gd.size(), N, bufA.p, ctr.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(&h, ctr.p, sizeof(h), hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st));
  h.numPrims = N - h.numInvalid;
  if (h.numInvalid) {                                          // rare: squeeze the invalid triangles out (stable)
    const uint32_t tiles = (N + 255u) / 256u;
    DevBuf<uint32_t> tileCount; HIP_TRY(tileCount.alloc(tiles));
    hipLaunchKernelGGL(compact_count, dim3(tiles), dim3(256), 0, st, bufA.p, N, tileCount.p);
    hipLaunchKernelGGL(compact_scan, dim3(1), dim3(1024), 0, st, tileCount.p, tiles, ctr.p);
    hipLaunchKernelGGL(compact_scatter, dim3(tiles), dim3(256), 0, st, bufA.p, N, tileCount.p, bufB.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(bufA.p, bufB.p, (size_t)h.numPrims * sizeof(PrimRef), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  uint32_t n = h.numPrims;
  auto decf = [](uint32_t u) { uint32_t v = u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu); float f; memcpy(&f, &v, 4); return f; };
  if (n == 0) { guard.ok = true; *out = bvh; return 0; }
  float glo[3], ghi[3], clo[3], chi[3];
  for (int d = 0; d < 3; d++) { glo[d] = decf(h.bounds[d]); ghi[d] = decf(h.bounds[3 + d]); clo[d] = decf(h.bounds[6 + d]); chi[d] = decf(h.bounds[9 + d]); }
  for (int d = 0; d < 3; d++) { info.bounds_lower[d] = glo[d]; info.bounds_upper[d] = ghi[d]; }
  if (presplit && splitBudget > 0u && n > 0u) {
    SplitGrid grid; float ext = 0.0f;
    for (int d = 0; d < 3; d++) { grid.base[d] = glo[d]; ext = fmaxf(ext, ghi[d] - glo[d]); }
    grid.extend = ext; grid.scale = ext == 0.0f ? 0.0f : 1024.0f / ext;
    const uint32_t tiles = (n + 255u) / 256u;
    DevBuf<float> prio, partial, psum; DevBuf<uint32_t> cnt, tileSum, total;
    HIP_TRY(prio.alloc(n)); HIP_TRY(partial.alloc(tiles)); HIP_TRY(psum.alloc(1)); HIP_TRY(cnt.alloc(n)); HIP_TRY(tileSum.alloc(tiles)); HIP_TRY(total.alloc(1));
    hipLaunchKernelGGL(presplit_priority, dim3(tiles), dim3(256), 0, st, bufA.p, n, dGeoms.p, grid, prio.p, partial.p);
    hipLaunchKernelGGL(presplit_sum, dim3(1), dim3(1024), 0, st, partial.p, tiles, psum.p);
    float budget = (float)splitBudget; uint32_t extra = 0; bool fits = false;
    for (int attempt = 0; attempt < 6 && !fits; attempt++, budget *= 0.5f) {
      hipLaunchKernelGGL(presplit_count, dim3(tiles), dim3(256), 0, st, bufA.p, n, dGeoms.p, grid, prio.p, psum.p, budget, cnt.p, tileSum.p);
      hipLaunchKernelGGL(presplit_scan, dim3(1), dim3(1024), 0, st, tileSum.p, tiles, total.p);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(&extra, total.p, 4, hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st));
      fits = extra <= splitBudget;
    }
    if (fits && extra > 0u) {
      hipLaunchKernelGGL(presplit_emit, dim3(tiles), dim3(256), 0, st, bufA.p, n, dGeoms.p, grid, cnt.p, tileSum.p);
      n += extra;
      Counters hb = h; for (int k = 6; k < 12; k++) hb.bounds[k] = (k < 9) ? ENC_POS_INF : ENC_NEG_INF;
      HIP_TRY(hipMemcpyAsync(ctr.p, &hb, sizeof(hb), hipMemcpyHostToDevice, st));
      const uint32_t cb = (n + 255u) / 256u < 1024u ? (n + 255u) / 256u : 1024u;
      hipLaunchKernelGGL(centroid_bounds, dim3(cb), dim3(256), 0, st, bufA.p, n, ctr.p);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(&hb, ctr.p, sizeof(hb), hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st));
      for (int d = 0; d < 3; d++) { clo[d] = decf(hb.bounds[6 + d]); chi[d] = decf(hb.bounds[9 + d]); }
      h.numPrims = n;
    }
    info.num_presplit = extra <= splitBudget ? extra : 0u;
  }
  info.num_triangles = n;

  // root binary node + first work item
  BNode rootB{}; for (int d = 0; d < 3; d++) { rootB.lo[d] = glo[d]; rootB.hi[d] = ghi[d]; } rootB.begin = 0; rootB.end = n; rootB.left = rootB.right = NIL; rootB.splitSah = INFINITY;
  HIP_TRY(hipMemcpyAsync(bnodes.p, &rootB, sizeof(rootB), hipMemcpyHostToDevice, st));
  h.numBLeaves = 0; h.numSegsNext = 0; h.numChunks = 0; h.numSmall = 0; h.numSegs = n > prm.small ? 1u : 0u; h.topLevels = 0;
  uint32_t numSegs = 0, numSmall = 0;
  if (n > prm.small) {
    Seg s0{}; s0.begin = 0; s0.end = n; s0.bnode = 0; for (int d = 0; d < 3; d++) { s0.cmin[d] = clo[d]; s0.cmax[d] = chi[d]; }
    HIP_TRY(hipMemcpyAsync(segs0.p, &s0, sizeof(s0), hipMemcpyHostToDevice, st)); numSegs = 1;
  } else {
    SmallEntry se{}; se.begin = 0; se.end = n; se.bnode = 0; se.buf = 0; for (int d = 0; d < 3; d++) { se.cmin[d] = clo[d]; se.cmax[d] = chi[d]; }
    HIP_TRY(hipMemcpyAsync(small.p, &se, sizeof(se), hipMemcpyHostToDevice, st)); h.numSmall = 1;
  }
  HIP_TRY(hipMemcpyAsync(ctr.p, &h, sizeof(h), hipMemcpyHostToDevice, st));

  if (prm.quality == 1u) {                                     // RTC_BUILD_QUALITY_LOW: Morton codes -> sort -> hierarchy -> boxes
    DevBuf<unsigned long long> keys, keysSorted; DevBuf<uint32_t> vals, valsSorted, parent, flags; DevBuf<char> tmp;
    HIP_TRY(keys.alloc(n)); HIP_TRY(keysSorted.alloc(n)); HIP_TRY(vals.alloc(n)); HIP_TRY(valsSorted.alloc(n)); HIP_TRY(parent.alloc(2ull * n)); HIP_TRY(flags.alloc(n));
    size_t tmpBytes = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmpBytes, keys.p, keysSorted.p, vals.p, valsSorted.p, (int)n, 0, 63, st));
    HIP_TRY(tmp.alloc(tmpBytes));
    float3 cmin = make_float3(clo[0], clo[1], clo[2]), cscale;
    { const float e[3] = {chi[0] - clo[0], chi[1] - clo[1], chi[2] - clo[2]}; float s3[3]; for (int d = 0; d < 3; d++) s3[d] = e[d] > 0.0f ? 2097152.0f / e[d] : 0.0f; cscale = make_float3(s3[0], s3[1], s3[2]); }
    const uint32_t g = (n + 255u) / 256u;
    hipLaunchKernelGGL(morton_keys, dim3(g), dim3(256), 0, st, bufA.p, n, cmin, cscale, keys.p, vals.p);
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmpBytes, keys.p, keysSorted.p, vals.p, valsSorted.p, (int)n, 0, 63, st));
    hipLaunchKernelGGL(morton_gather, dim3(g), dim3(256), 0, st, bufA.p, valsSorted.p, n, bufB.p, finalIds.p);
    HIP_TRY(hipMemsetAsync(flags.p, 0, (size_t)n * 4, st));
    if (n > 1u) hipLaunchKernelGGL(lbvh_hierarchy, dim3((n + 254u) / 256u), dim3(256), 0, st, keysSorted.p, n, bnodes.p, parent.p);
    hipLaunchKernelGGL(lbvh_bounds, dim3(g), dim3(256), 0, st, bufB.p, n, bnodes.p, parent.p, flags.p, ctr.p);
    HIP_TRY(hipGetLastError());
    numSegs = 0; numSmall = 0;
  }
  const bool sahBuild = prm.quality != 1u;
  // ---- top phase: one pass over the data per binary level.  Work-list sizes stay on the device: every kernel is launched with
  //      an upper bound of its grid (<= 2^level segments, <= N/CHUNK + #segments chunks) and surplus blocks exit at once, so
  //      the levels are enqueued back to back; the host looks at the counters only where the level count is not implied by N.
  uint32_t level = 0;
  Seg* cur = segs0.p; Seg* nxt = segs1.p;
  auto enqueue_top_level = [&]() {
    PrimRef* src = (level & 1u) ? bufB.p : bufA.p; PrimRef* dst = (level & 1u) ? bufA.p : bufB.p;
    const uint32_t segBound = level < 31u && (1u << level) < maxSegs ? (1u << level) : maxSegs;
    const uint32_t chunkBound = n / CHUNK + segBound + 1u;
    hipLaunchKernelGGL(top_setup, dim3(segBound), dim3(256), 0, st, cur, bins.p, chunks.p, ctr.p);
    hipLaunchKernelGGL(top_bin, dim3(chunkBound), dim3(256), 0, st, cur, chunks.p, src, bins.p, ctr.p);
    hipLaunchKernelGGL(top_split, dim3(segBound), dim3(64), 0, st, cur, bins.p, bnodes.p, ctr.p, prm, level >= 96u ? 1u : 0u);
    hipLaunchKernelGGL(top_partition, dim3(chunkBound), dim3(256), 0, st, cur, chunks.p, src, dst, ctr.p);
    hipLaunchKernelGGL(top_emit, dim3((segBound + 255u) / 256u), dim3(256), 0, st, cur, bnodes.p, nxt, small.p, ctr.p, prm,
                       (level & 1u) ? 0u : 1u, maxSegs, maxSmall);
    hipLaunchKernelGGL(top_advance, dim3(1), dim3(1), 0, st, ctr.p, maxSegs);
    Seg* t = cur; cur = nxt; nxt = t; level++;
  };
  if (numSegs && sahBuild) {
    uint32_t sure = 1; while (sure < 40u && ((uint64_t)prm.small << sure) < n) sure++;   // the largest segment halves at best: that many levels exist
    for (uint32_t i = 0; i < sure; i++) enqueue_top_level();
    for (;;) {
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(&h, ctr.p, sizeof(h), hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st));
      if (h.overflow) return set_error(hipErrorOutOfMemory, "top-phase work list overflow (pathological input)");
      if (h.numSegs == 0) break;
      enqueue_top_level(); enqueue_top_level();
    }
  } else { HIP_TRY(hipMemcpyAsync(&h, ctr.p, sizeof(h), hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st)); }
  info.top_levels = h.topLevels;
  numSmall = sahBuild ? h.numSmall : 0u;
  if (numSmall > maxSmall) return set_error(hipErrorOutOfMemory, "small list overflow");

  // ---- small phase
  const uint32_t microW = prm.minLeaf >= 2u ? 32u : 48u;       // LDS words per triangle of the micro mode (see micro_subtree)
  const size_t smallLds = sizeof(uint32_t) * (64u * microW > (uint32_t)BINS_WORDS ? 64u * microW : (uint32_t)BINS_WORDS);
  if (numSmall) hipLaunchKernelGGL(small_build, dim3(numSmall), dim3(64), smallLds, st, small.p, bufA.p, bufB.p, bnodes.p, finalIds.p, ctr.p, prm, microW);
  HIP_TRY(hipGetLastError());

  // ---- wide collapse, level by level
  const float rootArea = fmaf(ghi[0] - glo[0], (ghi[1] - glo[1]) + (ghi[2] - glo[2]), (ghi[1] - glo[1]) * (ghi[2] - glo[2]));
  hipLaunchKernelGGL(wide_root, dim3(1), dim3(1), 0, st, w0.p, ctr.p);
  WideItem* wc = w0.p; WideItem* wn = w1.p;
  uint32_t wlevel = 0;
  auto enqueue_wide_level = [&]() {
    uint64_t bound = 1; for (uint32_t i = 0; i < wlevel && bound < maxLevelItems; i++) bound *= 8u;     // <= 8^level items
    if (bound > maxLevelItems) bound = maxLevelItems;
    const uint32_t blocks = (uint32_t)((bound + 7u) / 8u) < 8192u ? (uint32_t)((bound + 7u) / 8u) : 8192u;   // = the waves resident at once
    const uint32_t parity = wlevel & 1u;
    hipLaunchKernelGGL(wide_plan, dim3(blocks), dim3(64), 0, st, wc, bnodes.p, plans.p, itemCnt.p, groupSum.p, ctr.p, prm, parity, rootArea);
    hipLaunchKernelGGL(wide_scan, dim3(1), dim3(1024), 0, st, groupSum.p, ctr.p, parity, maxWide);
    hipLaunchKernelGGL(wide_emit, dim3(blocks), dim3(64), 0, st, wc, bnodes.p, plans.p, itemCnt.p, groupSum.p, wnodes.p, finalIds.p, outIds.p, wn, ctr.p, parity);
    WideItem* t = wc; wc = wn; wn = t; wlevel++;
  };
  for (uint32_t i = 0; i < 8u; i++) enqueue_wide_level();
  for (;;) {
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&h, ctr.p, sizeof(h), hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st));
    if (h.overflow) return set_error(hipErrorOutOfMemory, "wide node pool overflow");
    if (h.wideCount[wlevel & 1u] == 0) break;
    for (uint32_t i = 0; i < 4u; i++) enqueue_wide_level();
  }
  const uint32_t depth = h.wideDepth;

  // ---- final arrays (exact size) + triangle records
  const uint32_t numNodes = h.numWide;
  HIP_TRY(hipMalloc(&bvh->d_tris, (size_t)n * sizeof(TriRec) + 128));
  if (numNodes) {
    HIP_TRY(hipMalloc(&bvh->d_nodes, (size_t)numNodes * sizeof(CNode)));
    HIP_TRY(hipMemcpyAsync(bvh->d_nodes, wnodes.p, (size_t)numNodes * sizeof(CNode), hipMemcpyDeviceToDevice, st));
  }
  hipLaunchKernelGGL(tri_records, dim3((n + 255u) / 256u), dim3(256), 0, st, outIds.p, n, dGeoms.p, (TriRec*)bvh->d_tris, bp->robust ? 1u : 0u);
  bvh->robust = bp->robust != 0;
  if (bp->refit && h.numInvalid == 0u && depth < 64u && !presplit) {        // keep the leaf order and the level table for mi355_bvh_refit
    HIP_TRY(hipMalloc(&bvh->d_ids, (size_t)n * sizeof(uint2)));
    HIP_TRY(hipMemcpyAsync(bvh->d_ids, outIds.p, (size_t)n * sizeof(uint2), hipMemcpyDeviceToDevice, st));
    bvh->lvlStart.assign(h.lvlStart, h.lvlStart + depth); bvh->lvlStart.push_back(numNodes);
    for (const GeomDesc& g : gd) bvh->sig.push_back({g.geomID, g.nt, g.nv, g.quad});
    info.bytes_refit = (uint64_t)n * sizeof(uint2);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ev1, st)); HIP_TRY(hipEventSynchronize(ev1));
  float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
  bvh->root = h.rootRef;
  info.root_ref = h.rootRef; info.num_nodes = numNodes; info.num_leaves = h.numLeaves; info.num_binary_nodes = 2ull * h.numBLeaves - 1ull;
  info.bytes_nodes = (uint64_t)numNodes * sizeof(CNode); info.bytes_triangles = (uint64_t)n * sizeof(TriRec);
  info.sah = (float)((double)h.sahFixed / 16777216.0) + (numNodes ? prm.travCost : 0.0f); info.build_ms = ms; info.depth = depth;
  guard.ok = true; *out = bvh;
  return 0;
}

static int refit_impl(Bvh* bvh, const mi355_mesh* meshes, uint32_t numMeshes, hipStream_t st) {
  HIP_TRY(hipSetDevice(bvh->device));
  const uint32_t n = (uint32_t)bvh->info.num_triangles, numNodes = (uint32_t)bvh->info.num_nodes;
  if (!bvh->d_ids || n == 0u || numNodes == 0u) return MI355_REFIT_IMPOSSIBLE;
  std::vector<GeomDesc> gd; uint64_t total = 0;
  for (uint32_t i = 0; i < numMeshes; i++) {
    const mi355_mesh& m = meshes[i];
    if (m.num_triangles == 0) continue;
    if (m.vertex_stride < 12 || (m.vertex_stride & 3) || m.index_stride < (m.quads ? 16u : 12u) || (m.index_stride & 3)) return set_error(hipErrorInvalidValue, "buffer stride");
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
  hipLaunchKernelGGL(tri_records, dim3((n + 255u) / 256u), dim3(256), 0, st, (const uint2*)bvh->d_ids, n, dGeoms.p, (TriRec*)bvh->d_tris, bvh->robust ? 1u : 0u);
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
  struct TmpGuard { float*& a; uint32_t*& b; ~TmpGuard() { if (a) hipFree(a); if (b) hipFree(b); } } tmpGuard{dfv, dfi};
  HIP_TRY(hipMalloc((void**)&dfv, fv.size() * 4)); HIP_TRY(hipMalloc((void**)&dfi, fi.size() * 4));
  HIP_TRY(hipMemcpyAsync(dfv, fv.data(), fv.size() * 4, hipMemcpyHostToDevice, st)); HIP_TRY(hipMemcpyAsync(dfi, fi.data(), fi.size() * 4, hipMemcpyHostToDevice, st));
  mi355_mesh fake{}; fake.d_vertices = dfv; fake.vertex_stride = 12; fake.num_vertices = 3 * R; fake.d_indices = dfi; fake.index_stride = 12; fake.num_triangles = R; fake.geom_id = 0; fake.mask = 0xFFFFFFFFu;
  mi355_build_params tp = *bp; tp.refit = 0; tp.quality = 0;
  Bvh* top = nullptr;
  { const int rc = build_impl(device, &fake, 1, &tp, st, &top); if (rc) return rc; }
  struct TopGuard { Bvh* t; ~TopGuard() { delete t; } } topGuard{top};
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
  HIP_TRY(hipMalloc(&bvh->d_nodes, (size_t)nNodes * sizeof(CNode)));
  HIP_TRY(hipMalloc(&bvh->d_tris, (size_t)nTris * sizeof(TriRec) + 128));
  HIP_TRY(hipMemcpyAsync(bvh->d_nodes, top->d_nodes, (size_t)top->info.num_nodes * sizeof(CNode), hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(bvh->d_tris, top->d_tris, (size_t)top->info.num_triangles * sizeof(TriRec), hipMemcpyDeviceToDevice, st));
  for (size_t k = 0; k < objs.size(); k++) {
    const uint32_t n = (uint32_t)objs[k]->info.num_nodes;
    if (n) hipLaunchKernelGGL(rebase_nodes, dim3((n + 255u) / 256u), dim3(256), 0, st, (const uint4*)objs[k]->d_nodes, (uint4*)bvh->d_nodes + 5ull * nodeOfs[k], n, nodeOfs[k], triOfs[k]);
    HIP_TRY(hipMemcpyAsync((char*)bvh->d_tris + (size_t)triOfs[k] * sizeof(TriRec), objs[k]->d_tris, (size_t)objs[k]->info.num_triangles * sizeof(TriRec), hipMemcpyDeviceToDevice, st));
  }
  for (uint32_t k = 0; k < R; k++) recs[k].root = nodeOfs[objIndex[recObj[k]]];
  HIP_TRY(hipMalloc((void**)&dRecs, (size_t)R * sizeof(InstRec))); bvh->d_insts = dRecs;
  HIP_TRY(hipMemcpyAsync(dRecs, recs.data(), (size_t)R * sizeof(InstRec), hipMemcpyHostToDevice, st));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  bvh->root = 0;
  info.root_ref = 0; info.num_triangles = objTris; info.num_nodes = nNodes; info.num_leaves = top->info.num_leaves;
  info.bytes_nodes = nNodes * sizeof(CNode); info.bytes_triangles = nTris * sizeof(TriRec) + (uint64_t)R * sizeof(InstRec);
  info.sah = top->info.sah; info.build_ms = top->info.build_ms; info.top_levels = top->info.top_levels;
  info.depth = 2u * top->info.depth + maxDepth + 2u;             // a top level leaves up to two entries on a lane's stack (inner children, other instances)
  guard.ok = true; *out = bvh;
  return 0;
}

}  // namespace mi355

extern "C" {

void mi355_default_build_params(mi355_build_params* p) {
  memset(p, 0, sizeof(*p));
  p->sah_block_shift = 0; p->min_leaf = 2; p->max_leaf = 3; p->small_threshold = 1024; p->trav_cost = 1.0f; p->int_cost = 1.0f; p->split_factor = 1.2f;
}
const char* mi355_last_error(void) { return mi355::g_err.c_str(); }
int mi355_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
int mi355_device_name(int device, char* out, size_t n) {
  hipDeviceProp_t prop; HIP_TRY(hipGetDeviceProperties(&prop, device));
  snprintf(out, n, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount); return 0;
}
int mi355_bvh_build(int device, const mi355_mesh* meshes, uint32_t num_meshes, const mi355_build_params* params, void* stream, mi355_bvh_t* out) {
  mi355_build_params def; if (!params) { mi355_default_build_params(&def); params = &def; }
  mi355::Bvh* b = nullptr;
  const int rc = mi355::build_impl(device, meshes, num_meshes, params, (hipStream_t)stream, &b);
  *out = (mi355_bvh_t)b; return rc;
}
void mi355_bvh_destroy(mi355_bvh_t bvh) { delete (mi355::Bvh*)bvh; }
int mi355_bvh_build_instanced(int device, mi355_bvh_t own, const mi355_instance* instances, uint32_t num_instances, const mi355_build_params* params, void* stream, mi355_bvh_t* out) {
  mi355_build_params def; if (!params) { mi355_default_build_params(&def); params = &def; }
  mi355::Bvh* b = nullptr;
  const int rc = mi355::build_instanced_impl(device, (mi355::Bvh*)own, instances, num_instances, params, (hipStream_t)stream, &b);
  *out = (mi355_bvh_t)b; return rc;
}
int mi355_bvh_refit(mi355_bvh_t bvh, const mi355_mesh* meshes, uint32_t num_meshes, void* stream) {
  if (!bvh) return MI355_REFIT_IMPOSSIBLE;
  return mi355::refit_impl((mi355::Bvh*)bvh, meshes, num_meshes, (hipStream_t)stream);
}
void mi355_release_build_scratch(int device) {
  Arena* a = arena_of(device);
  std::lock_guard<std::mutex> lk(a->mtx);
  hipSetDevice(device); a->release();
}
int mi355_bvh_get_info(mi355_bvh_t bvh, mi355_bvh_info* info) { *info = ((mi355::Bvh*)bvh)->info; return 0; }
int mi355_bvh_download(mi355_bvh_t bvh, void* nodes, size_t nb, void* tris, size_t tb) {
  mi355::Bvh* b = (mi355::Bvh*)bvh; HIP_TRY(hipSetDevice(b->device));
  if (nodes && nb) { if (nb > b->info.bytes_nodes) nb = b->info.bytes_nodes; if (nb) HIP_TRY(hipMemcpy(nodes, b->d_nodes, nb, hipMemcpyDeviceToHost)); }
  if (tris && tb) { if (tb > b->info.bytes_triangles) tb = b->info.bytes_triangles; if (tb) HIP_TRY(hipMemcpy(tris, b->d_tris, tb, hipMemcpyDeviceToHost)); }
  return 0;
}
int mi355_malloc(int device, size_t bytes, void** d) { HIP_TRY(hipSetDevice(device)); HIP_TRY(hipMalloc(d, bytes ? bytes : 1)); return 0; }
int mi355_free(void* d) { HIP_TRY(hipFree(d)); return 0; }
int mi355_memcpy_h2d(void* d, const void* h, size_t n) { HIP_TRY(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); return 0; }
int mi355_memcpy_d2h(void* h, const void* d, size_t n) { HIP_TRY(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); return 0; }
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
