// build_common.inl -- constants, build-time structs, ordered-uint float codes, wave64 DPP reductions.
// Part of build.hip (included inside its anonymous namespace); see the header of build.hip for the pipeline.


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
  uint32_t chunk0, pad0, pad1, pad2;            // first of the set's chunks in this level's chunk list (they are consecutive)
};
struct SmallEntry { uint32_t begin, end, bnode, buf; float cmin[3], cmax[3]; };
struct Chunk { uint32_t seg, begin, end; };
struct WideItem { uint32_t bnode, node; };
struct Counters {
  uint32_t numPrims, pad_a, pad_d, pad_f, pad_e, numWide, numWideNext, pad_b;
  uint32_t bounds[12];                          // scene geom lo/hi + centroid lo/hi (ordered uint)
  uint32_t overflow, rootRef, numTrisOut, numInvalid;
  uint32_t numSegs, topLevels, wideDepth, lvlNodeBase, lvlTriBase, wideCount[2];      // level loops are driven from the device: no host readback per level
  unsigned long long pad_c;
  float rootArea;                                 // half area of the scene bounds (SAH statistics are relative to it); written by root_setup
  uint32_t numOutliers;                           // MEDIUM builds: references cut up front because their box dwarfs the average one (build_presplit.inl, outlier_*)
  unsigned long long pad_g;
  double areaSum;                                 // MEDIUM builds: sum of the valid references' box areas (primref_gen's per-workgroup parts added in index order by outlier_stats)
  uint32_t compactFrom;                           // stable compaction: first position that moves (everything before the first hole stays where it is)
  uint32_t outlierCells, outlierPieces, outlierValid, outlierSkip;   // ... the places reserved for their pieces behind the references, the pieces that exist, the valid references counted, 1 = too many
  uint32_t emitBlocks;                            // top_emit: workgroups that are done with the level (the last one moves the work lists on)
  unsigned long long outlierWork;                 // outlier_emit -> outlier_clip: outliers listed << 32 | 256-cell chunks handed out so far (ONE atomic: list order = chunk order)
  uint32_t chunkedLevels, localFirst;             // top phase: levels that had a set of more than CHUNK references; the first level with a smaller one (top_local's) -- what the next commit enqueues
  uint32_t padC[2];
  unsigned long long smTime[4];                  // -DSM_TIME: wave cycles small_build spent binning / pricing / partitioning / in the micro mode
  uint32_t lvlStart[64];                          // first node of every level of the wide tree (numbering is breadth first): what a refit walks bottom-up
  alignas(128) uint32_t numSmall; uint32_t numSegsNext; uint32_t numChunks;   // lengths of the work lists the top phase appends to, on a line of their own: top_local appends twice per workgroup, and every
                                                  // one of its thousands of workgroups reads numSegs first (top_local 383 -> 358 us per commit with the appends off that line;
                                                  // the two lengths on a line EACH: top_emit 129 -> 191 us, top_local unchanged -- they stay together)
  // Words that EVERY workgroup of a large grid sends an atomic to.  Atomic instructions on ONE cache line are served one after the other, ~10 ns each when they come from all
  // CUs (MI355X_MICROARCH.md "fanin"): the 8192 workgroups of a large level of wide_plan ended on 2 x 8192 of them -- ~80 us of the 175 us of the widest level, 250 us per
  // commit (found when both words were moved onto ONE line: 577 -> 850 us) -- and outlier_mark with six words per workgroup on one line took 140 us instead of 25.  A
  // workgroup therefore reports to stripe[blockIdx mod STRIPES], a line of its own per stripe; sums and min / max do not care, whoever needs the value folds the stripes.
  struct alignas(128) Stripe {
    unsigned long long sahFixed;                  // SAH statistics of the wide tree, 2^-24 fixed point (order-independent sum): wide_plan
    uint32_t numLeaves;                           // leaf slots of the wide tree: wide_plan
    uint32_t numBLeaves;                          // leaves of the binary tree: small_build (LOW: stripe 0 holds n)
    uint32_t cb2[6];                              // MEDIUM builds that cut outliers: centroid lo/hi over the references that stay (outlier_mark) and the pieces (outlier_clip): the root's centroid box if anything was cut
    unsigned long long areaFixed;                 // spatial-split builds: sum of the references' box areas / scene area, 2^-32 fixed point (spatial_area_sum; spatial_budgets folds the stripes)
    uint32_t bounds[12];                          // what primref_gen's workgroups found (scene geom lo/hi + centroid lo/hi): folded into Counters::bounds by fold_bounds (outlier_stats / root_setup / bounds_fold)
  };
  static constexpr uint32_t STRIPES = 64u;
  Stripe stripe[STRIPES];
};
// word k of the scene bounds = the fold of what primref_gen's workgroups left in their stripes (idempotent: may run more than once)
__device__ __forceinline__ void fold_bounds(Counters* ctr, uint32_t k) {
  uint32_t x[Counters::STRIPES];
#pragma unroll
  for (uint32_t r = 0; r < Counters::STRIPES; r++) x[r] = ctr->stripe[r].bounds[k];   // (all loads before the first use: one round trip)
  uint32_t v = ctr->bounds[k];
#pragma unroll
  for (uint32_t r = 0; r < Counters::STRIPES; r++) v = (k % 6u) < 3u ? (x[r] < v ? x[r] : v) : (x[r] > v ? x[r] : v);
  ctr->bounds[k] = v;
}
struct Params { uint32_t shift, minLeaf, maxLeaf, small; float travCost, intCost; uint32_t quality, spatial; };

// order-preserving float <-> uint so that integer atomicMin/Max reduce floats exactly
__device__ __forceinline__ uint32_t enc(float f) { uint32_t u = __float_as_uint(f); return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__device__ __forceinline__ float dec(uint32_t u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }
// v_min_f32 / v_max_f32 as they are: fminf / fmaxf of values that come out of a bit cast are preceded by a canonicalising v_max x, x each (294 of the
// kernel's 3100 VALU instructions); a quiet NaN operand loses against a number either way, which lets empty bins decode to NaN and drop out
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

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
// (v_<op>_dpp dst, dst, dst: the DPP operand folded into the operation -- the compiler emits v_mov + v_mov_dpp + v_<op> per step; a lane without a source,
// or in a row the row mask leaves out, keeps its value; two wait states between a VALU write and the DPP read of the same register)
#define MI355_WAVE_REDUCE63(OP) \
  "s_nop 1\n " OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n " OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n" \
  "s_nop 1\n " OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n " OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n" \
  "s_nop 1\n " OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n " OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
__device__ __forceinline__ uint32_t wave_umin63(uint32_t v) { asm volatile(MI355_WAVE_REDUCE63("v_min_u32_dpp") : "+v"(v)); return v; }
__device__ __forceinline__ uint32_t wave_umax63(uint32_t v) { asm volatile(MI355_WAVE_REDUCE63("v_max_u32_dpp") : "+v"(v)); return v; }

// inclusive prefix sum over the 64 lanes of a wave: four row_shr steps, row_bcast:15, row_bcast:31 (the DPP operand folded into the add; a lane without a source keeps its value)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
               "s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"
               "s_nop 1\n v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 1\n" : "+v"(v));
  return v;
}
// Exclusive prefix sums of one pair per thread over a workgroup of 1024 threads: every wave scans its 64 pairs with DPP, wave 0 scans the 16 wave totals, three barriers.
// (round 6: was a Hillis-Steele scan through LDS, ten rounds of two barriers each = ~6 us of every single-block scan kernel -- compact_scan, presplit_scan and the 13
// wide_scan launches of a commit.)  s_w: 17 pairs of LDS.  Returns the exclusive prefix of this thread; total = the sum over the workgroup.
__device__ __forceinline__ uint2 block_exclusive_scan_1024_u2(uint2 v, uint2* s_w, uint32_t tid, uint2& total) {
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  uint2 incl; incl.x = wave_incl_scan_u32(v.x); incl.y = wave_incl_scan_u32(v.y);
  if (lane == 63u) s_w[wave] = incl;
  __syncthreads();
  if (tid < 64u) {
    uint2 w = tid < 16u ? s_w[tid] : make_uint2(0u, 0u);
    uint2 wi; wi.x = wave_incl_scan_u32(w.x); wi.y = wave_incl_scan_u32(w.y);
    if (tid < 16u) s_w[tid] = make_uint2(wi.x - w.x, wi.y - w.y);
    if (tid == 15u) s_w[16] = wi;
  }
  __syncthreads();
  const uint2 base = s_w[wave]; total = s_w[16];
  __syncthreads();                                               // (s_w may be used again at once)
  return make_uint2(base.x + incl.x - v.x, base.y + incl.y - v.y);
}
// Eight consecutive words of a thread's run as TWO 16-byte loads / stores (a run starts at a multiple of eight words of a 256-byte aligned array whose allocation is padded
// to 256 bytes: reading the words behind `e` is harmless, they are replaced by `fill`).  The single-workgroup scans are bound by the address path of their ONE CU -- a
// scattered one-word load costs it a clock per lane: presplit_scan over the crown's 18,605 tiles was 2 x 384 wave-loads of 64 addresses = 27.7 us, compact_scan 30.1 us.
__device__ __forceinline__ void load8_fill(const uint32_t* p, uint32_t i, uint32_t e, uint32_t fill, uint32_t (&x)[8]) {
  uint4 a = make_uint4(fill, fill, fill, fill), b = a;
  if (i < e) a = ((const uint4*)(p + i))[0];
  if (i + 4u < e) b = ((const uint4*)(p + i))[1];
  x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
#pragma unroll
  for (uint32_t k = 0; k < 8u; k++) if (i + k >= e) x[k] = fill;
}
__device__ __forceinline__ void store8_upto(uint32_t* p, uint32_t i, uint32_t e, const uint32_t (&y)[8]) {
  if (i + 8u <= e) { ((uint4*)(p + i))[0] = make_uint4(y[0], y[1], y[2], y[3]); ((uint4*)(p + i))[1] = make_uint4(y[4], y[5], y[6], y[7]); }
  else {
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++) if (i + k < e) p[i + k] = y[k];
  }
}
// s[0..1024) becomes its exclusive scan, the total is returned to every thread
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t* s, uint32_t tid) {
  __shared__ uint2 s_w[17];
  uint2 total;
  const uint2 ex = block_exclusive_scan_1024_u2(make_uint2(s[tid], 0u), s_w, tid, total);
  s[tid] = ex.x;
  __syncthreads();
  return total.x;
}
