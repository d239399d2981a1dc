// build_presplit.inl -- RTC_BUILD_QUALITY_HIGH: pre-splitting of large triangles.
// Part of build.hip (included inside its anonymous namespace); see the header of build.hip for the pipeline.
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
    for (int d = 0; d < 3; d++) {                                  // interpolated cut points: 4 ulp of slack before the clamp (see split_triangle, build_spatial.inl)
      const float eL = 4.76837158e-7f * fmaxf(fabsf(L.lo[d]), fabsf(L.hi[d])), eR = 4.76837158e-7f * fmaxf(fabsf(R.lo[d]), fabsf(R.hi[d]));
      if (L.lo[d] <= L.hi[d]) { L.lo[d] -= eL; L.hi[d] += eL; }
      if (R.lo[d] <= R.hi[d]) { R.lo[d] -= eR; R.hi[d] += eR; }
    }
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
  const uint32_t tid = threadIdx.x, per = ((numTiles + 1023u) / 1024u + 7u) & ~7u, b = min(tid * per, numTiles), e = min(b + per, numTiles);
  uint32_t sum = 0;
  for (uint32_t i = b; i < e; i += 8u) {                         // eight words per step as two 16-byte loads (load8_fill, build_common.inl)
    uint32_t x[8]; load8_fill(tileSum, i, e, 0u, x);
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++) sum += x[k];
  }
  s_part[tid] = sum; __syncthreads();
  const uint32_t totalAll = block_exclusive_scan_1024(s_part, tid);
  if (tid == 0) total[0] = totalAll;
  uint32_t run = s_part[tid];
  for (uint32_t i = b; i < e; i += 8u) {
    uint32_t x[8], y[8]; load8_fill(tileSum, i, e, 0u, x);
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++) { y[k] = run; run += x[k]; }
    store8_upto(tileSum, i, e, y);
  }
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

// --------------------------------------------------------------------------------- RTC_BUILD_QUALITY_MEDIUM: the few references that dwarf all others are cut up front
// A handful of triangles that span the scene (the walls of a room, a ground plane) sit in every node of the first levels and make every box above them as
// large as they are.  The reference's answer is RTC_BUILD_QUALITY_HIGH (spatial splits, above and in build_spatial.inl), at 2.5 x the build time.  Measured on
// the crown stand-in (profiles/r03_collapse.md): ALL of what the HIGH tree gains on the bench's rays (36.6 -> 32.7 node visits, 51.0 -> 46.9 triangle tests per
// ray, +11 % rays per second) comes from the twelve room-sized triangles -- and a tree built by object splits alone over the same room cut into a grid of
// pieces gains the same (tests/gpu_perf.py --tess-room: 32.5 / 47.2 with 16 pieces per wall triangle, 32.0 / 44.9 with 1024).  So a MEDIUM commit cuts such
// references before the build, on the device, inside the one-round-trip commit:
//   outlier_stats  sum of the valid references' box areas and their number (the per-workgroup parts primref_gen left, added in index order: order independent)
//   outlier_mark   a reference is an OUTLIER if its box area is >= top_split_rel (32) x the mean box area and longer than one cell of a uniform grid of
//                  top_split_cell (1/8) of the scene's largest extent; it asks for one place per grid cell its box covers (<= 32 cells per axis)
//   presplit_scan  places behind the references, in reference order (no atomic decides an index: rebuilds are bit-identical)
//   outlier_emit   the triangle is clipped against every cell (Sutherland-Hodgman, 6 planes); a cell it really crosses gets a reference of the SAME triangle
//                  with the clipped polygon's box (widened by 4 ulp, clamped to cell and box), an empty cell leaves a hole; the original is dropped
//   the stable compaction that squeezes out invalid triangles (compact_*) squeezes out the holes; the root takes the centroid box outlier_mark (the references
//   that stay) and outlier_clip (the pieces) measured on the way (ctr->cb2).
// Everything after that is the ordinary binned-SAH build over a few more references (crown stand-in: 12 triangles -> 3,3 k pieces, +0.07 %).  If the
// outliers' cells do not fit the reserve (N / 16 + 65536 places) nothing is cut: thousands of long pipe triangles are not what this is for.
constexpr int OUTLIER_MAX_AXIS = 32;
__device__ __forceinline__ float ctr_root_area2(const Counters* ctr, float (&ext)[3]) {
  for (int d = 0; d < 3; d++) ext[d] = dec(ctr->bounds[3 + d]) - dec(ctr->bounds[d]);
  return 2.0f * half_area3(ext[0], ext[1], ext[2]);
}
// the workgroups' parts of primref_gen (build_primref.inl) in index order: every thread its run, then a fixed tree -- the sum does not depend on timing
__global__ __launch_bounds__(1024) void outlier_stats(const AreaPart* part, uint32_t numParts, Counters* ctr) {
  __shared__ double s_a[1024]; __shared__ unsigned long long s_c[1024];
  const uint32_t tid = threadIdx.x, per = (numParts + 1023u) / 1024u, b = min(tid * per, numParts), e = min(b + per, numParts);
  if (tid < 12u) fold_bounds(ctr, tid);                          // (the scene bounds outlier_mark reads: primref_gen's stripes)
  double a = 0.0; unsigned long long c = 0ull;
  for (uint32_t i = b; i < e; i++) { a += part[i].area; c += part[i].count; }
  s_a[tid] = a; s_c[tid] = c; __syncthreads();
  for (uint32_t o = 512u; o > 0u; o >>= 1) { if (tid < o) { s_a[tid] += s_a[tid + o]; s_c[tid] += s_c[tid + o]; } __syncthreads(); }
  if (tid == 0u) { ctr->areaSum = s_a[0]; ctr->outlierValid = (uint32_t)s_c[0]; }
}
// grid cells of one reference (1 = not an outlier)
__device__ __forceinline__ uint32_t outlier_cells(const PrimRef& r, const Counters* ctr, float minRel, float cellFrac, uint32_t (&nc)[3], float (&cell)[3]) {
  nc[0] = nc[1] = nc[2] = 1u; cell[0] = cell[1] = cell[2] = 0.0f;
  if (r.geom == NIL) return 1u;
  float ext[3]; const float rootArea2 = ctr_root_area2(ctr, ext);
  const double sum = ctr->areaSum;
  const uint32_t valid = ctr->outlierValid;
  if (!(rootArea2 > 0.0f) || !(sum > 0.0) || valid < 1024u) return 1u;              // (small scenes have no "average" worth the name)
  const float a = 2.0f * half_area3(r.hi[0] - r.lo[0], r.hi[1] - r.lo[1], r.hi[2] - r.lo[2]);
  const float rel = (float)((double)valid * ((double)a / sum));                      // this box's area / the mean box area
  if (!(rel >= minRel)) return 1u;
  const float L = fmaxf(ext[0], fmaxf(ext[1], ext[2])) * cellFrac;
  if (!(L > 0.0f)) return 1u;
  uint32_t total = 1u;
  for (int d = 0; d < 3; d++) {
    const float e = r.hi[d] - r.lo[d];
    float k = ceilf(e / L); k = k < 1.0f ? 1.0f : (k > (float)OUTLIER_MAX_AXIS ? (float)OUTLIER_MAX_AXIS : k);
    nc[d] = (uint32_t)k; cell[d] = e / k; total *= nc[d];
  }
  return total;
}
// (round 6) While it has the references in its registers this kernel also takes (a) the number of valid references of every tile -- what compact_count would read them a
// third time for (tileCount; outlier_emit takes the cut ones off) -- and (b) the centroid box of the references that are NOT cut (ctr->cb2; outlier_clip adds the pieces'):
// what centroid_bounds_guarded read the whole array a fourth time for.  min / max over the same set: the root's centroid box does not change by a bit.
__global__ __launch_bounds__(256) void outlier_mark(const PrimRef* prims, uint32_t n, Counters* ctr, float minRel, float cellFrac, uint32_t* cnt, uint32_t* tileSum, uint32_t* tileCount) {
  __shared__ uint32_t s_w[4], s_v[4], s_acc[6];
  const uint32_t tid = threadIdx.x, i = blockIdx.x * 256u + tid;
  if (tid < 6u) s_acc[tid] = tid < 3u ? 0xFFFFFFFFu : 0u;
  uint32_t c = 0u; bool valid = false;
  uint32_t acc[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  if (i < n) {
    const PrimRef r = load_prim(prims + i);
    uint32_t nc[3]; float cell[3];
    const uint32_t cells = outlier_cells(r, ctr, minRel, cellFrac, nc, cell);
    c = cells > 1u ? cells : 0u;
    cnt[i] = c;
    valid = r.geom != NIL;
    if (valid && c == 0u) for (int d = 0; d < 3; d++) { const uint32_t c2 = enc(r.lo[d] + r.hi[d]); acc[d] = c2; acc[3 + d] = c2; }
  }
  uint32_t x = c; for (int o = 32; o > 0; o >>= 1) x += (uint32_t)__shfl_down((int)x, o, 64);
  const uint32_t nv = (uint32_t)__popcll(__ballot(valid));
  if ((tid & 63u) == 0u) { s_w[tid >> 6] = x; s_v[tid >> 6] = nv; }
  __syncthreads();                                               // (s_acc is initialised as well)
  for (int k = 0; k < 6; k++) {
    const uint32_t y = k < 3 ? wave_umin63(acc[k]) : wave_umax63(acc[k]);
    if ((tid & 63u) == 63u) { if (k < 3) atomicMin(&s_acc[k], y); else atomicMax(&s_acc[k], y); }
  }
  if (tid == 0u) { tileSum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3]; tileCount[blockIdx.x] = s_v[0] + s_v[1] + s_v[2] + s_v[3]; }
  __syncthreads();
  if (tid < 6u) { uint32_t* a = &ctr->stripe[blockIdx.x % Counters::STRIPES].cb2[tid]; const uint32_t v = s_acc[tid]; if (tid < 3u) { if (v != 0xFFFFFFFFu) atomicMin(a, v); } else { if (v != 0u) atomicMax(a, v); } }
}
// Sutherland-Hodgman: the polygon in `p` (n vertices) against the half space x[axis] <= pos (keepLess) or >= pos; returns the new vertex count (<= n + 1)
__device__ int clip_halfspace(float (*p)[3], int n, int axis, float pos, bool keepLess) {
  float q[10][3]; int m = 0;
  for (int i = 0; i < n; i++) {
    const float* a = p[i]; const float* b = p[i + 1 == n ? 0 : i + 1];
    const float da = keepLess ? pos - a[axis] : a[axis] - pos, db = keepLess ? pos - b[axis] : b[axis] - pos;   // >= 0: inside
    if (da >= 0.0f) { for (int d = 0; d < 3; d++) q[m][d] = a[d]; m++; }
    if ((da > 0.0f && db < 0.0f) || (da < 0.0f && db > 0.0f)) {
      const float t = da / (da - db);
      for (int d = 0; d < 3; d++) q[m][d] = fmaf(t, b[d] - a[d], a[d]);
      q[m][axis] = pos; m++;
    }
  }
  for (int i = 0; i < m; i++) for (int d = 0; d < 3; d++) p[i][d] = q[i][d];
  return m;
}
// outlier_emit lists the outliers of its tile -- {reference, first place of its pieces, cells} -- and retires the originals; outlier_clip does the clipping, one 256-cell chunk of
// one outlier per workgroup pass.  (One kernel did both: the twelve room triangles of the crown stand-in sit in ONE tile, whose single workgroup then clipped 12 x 1024 cells
// alone: 204 us of a 6.2 ms commit.)  Places come from the scans, so which workgroup clips what does not matter: rebuilds stay bit-identical.
struct OutlierWork { uint32_t src, base, cells, chunk0; };
__global__ __launch_bounds__(256) void outlier_emit(PrimRef* prims, uint32_t n, uint32_t cap, const uint32_t* cnt, const uint32_t* tileOfs, const uint32_t* total,
                                                    Counters* ctr, OutlierWork* work, uint32_t* tileCount) {
  __shared__ uint32_t s_scan[256];
  const uint32_t tid = threadIdx.x, i = blockIdx.x * 256u + tid;
  const uint32_t tot = total[0];
  if (tot == 0u) return;
  if (tot > cap) { if (blockIdx.x == 0u && tid == 0u) ctr->outlierSkip = 1u; return; }          // too many / too large outliers for the reserve: nothing is cut
  if (blockIdx.x == 0u && tid == 0u) ctr->outlierCells = tot;
  const uint32_t c = i < n ? cnt[i] : 0u;
  const int cutHere = __syncthreads_count(c != 0u);
  if (cutHere == 0) return;                                                                      // no outlier in this tile (all but a dozen of the 18,605 tiles of the crown stand-in)
  if (tid == 0u) tileCount[blockIdx.x] -= (uint32_t)cutHere;                                     // (outlier_mark counted them as valid: outlier_retire takes them out, the compaction squeezes them out)
  s_scan[tid] = c;
  __syncthreads();
  for (uint32_t o = 1; o < 256u; o <<= 1) { uint32_t x = 0; if (tid >= o) x = s_scan[tid - o]; __syncthreads(); s_scan[tid] += x; __syncthreads(); }
  if (c != 0u) {
    const uint32_t chunks = (c + 255u) / 256u;
    const unsigned long long w = atomicAdd(&ctr->outlierWork, (1ull << 32) | chunks);
    OutlierWork ow; ow.src = i; ow.base = n + tileOfs[blockIdx.x] + s_scan[tid] - c; ow.cells = c; ow.chunk0 = (uint32_t)w;
    work[(uint32_t)(w >> 32)] = ow;
  }
}
__global__ __launch_bounds__(256) void outlier_clip(PrimRef* prims, uint32_t cap, const GeomDesc* geoms, const uint32_t* total, Counters* ctr, const OutlierWork* work,
                                                    float minRel, float cellFrac) {
  __shared__ uint32_t s_pieces, s_holes, s_acc[6];
  const uint32_t tid = threadIdx.x;
  const uint32_t tot = total[0];
  if (tot == 0u || tot > cap) return;
  if (tid < 6u) s_acc[tid] = tid < 3u ? 0xFFFFFFFFu : 0u;
  uint32_t acc[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};                            // centroid box of this thread's pieces (ctr->cb2, see outlier_mark)
  const unsigned long long w = ctr->outlierWork;                                                 // (final: outlier_emit is over)
  const uint32_t numWork = (uint32_t)(w >> 32), numChunks = (uint32_t)w;
  if (tid == 0u) { s_pieces = 0u; s_holes = 0u; }
  __syncthreads();
  uint32_t pieces = 0u, holes = 0u, cut = 0u;
  for (uint32_t ch = blockIdx.x; ch < numChunks; ch += gridDim.x) {
    uint32_t lo_ = 0u, hi_ = numWork;                                                            // the outlier this chunk belongs to: the last one with chunk0 <= ch (the list is in chunk order)
    while (hi_ - lo_ > 1u) { const uint32_t mid = (lo_ + hi_) >> 1; if (work[mid].chunk0 <= ch) lo_ = mid; else hi_ = mid; }
    const OutlierWork ow = work[lo_];
    const PrimRef ref = load_prim(prims + ow.src);                                               // (its geometry word is not touched before the end of this kernel)
    uint32_t nc[3]; float cell[3];
    outlier_cells(ref, ctr, minRel, cellFrac, nc, cell);
    float v[3][3]; load_tri(geoms, ref, v);
    const uint32_t k = (ch - ow.chunk0) * 256u + tid;
    if (k < ow.cells) {
      const uint32_t cx = k % nc[0], cy = (k / nc[0]) % nc[1], cz = k / (nc[0] * nc[1]);
      float lo[3], hi[3];
      const uint32_t ci[3] = {cx, cy, cz};
      for (int d = 0; d < 3; d++) {                                                              // the cell (the outermost ones end exactly at the box)
        lo[d] = ci[d] == 0u ? ref.lo[d] : fmaf((float)ci[d], cell[d], ref.lo[d]);
        hi[d] = ci[d] + 1u == nc[d] ? ref.hi[d] : fmaf((float)(ci[d] + 1u), cell[d], ref.lo[d]);
      }
      float p[10][3]; int m = 3;
      for (int a = 0; a < 3; a++) for (int d = 0; d < 3; d++) p[a][d] = v[a][d];
      for (int d = 0; d < 3 && m > 0; d++) { m = clip_halfspace(p, m, d, lo[d], false); if (m > 0) m = clip_halfspace(p, m, d, hi[d], true); }
      PrimRef o = ref;
      if (m > 0) {
        float blo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, bhi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        for (int a = 0; a < m; a++) for (int d = 0; d < 3; d++) { blo[d] = fminf(blo[d], p[a][d]); bhi[d] = fmaxf(bhi[d], p[a][d]); }
        for (int d = 0; d < 3; d++) {                                                            // interpolated points: 4 ulp of their magnitude, then the cell and the triangle's own box
          const float e = 4.76837158e-7f * fmaxf(fabsf(blo[d]), fabsf(bhi[d]));
          o.lo[d] = fmaxf(fmaxf(blo[d] - e, lo[d] - e), ref.lo[d]); o.hi[d] = fminf(fminf(bhi[d] + e, hi[d] + e), ref.hi[d]);
          const uint32_t c2 = enc(o.lo[d] + o.hi[d]); acc[d] = min(acc[d], c2); acc[3 + d] = max(acc[3 + d], c2);
        }
        pieces++;
      } else { o.geom = NIL; holes++; }                                                          // the triangle does not cross this cell: a hole, squeezed out by the compaction
      store_prim(prims + ow.base + k, o);
    }
    if (tid == 0u && ch == ow.chunk0) cut++;                                                     // (counted once per outlier; the original is retired by outlier_retire)
  }
  if (pieces) atomicAdd(&s_pieces, pieces);
  if (holes) atomicAdd(&s_holes, holes);
  if (pieces) for (int k = 0; k < 6; k++) { if (k < 3) atomicMin(&s_acc[k], acc[k]); else atomicMax(&s_acc[k], acc[k]); }   // (a few dozen workgroups of a commit get here at all)
  __syncthreads();
  if (tid == 0u && (s_pieces | s_holes | cut)) { atomicAdd(&ctr->outlierPieces, s_pieces); atomicAdd(&ctr->numOutliers, cut); atomicAdd(&ctr->numInvalid, s_holes + cut); }
  if (tid < 6u && s_pieces) { uint32_t* a = &ctr->stripe[blockIdx.x % Counters::STRIPES].cb2[tid]; if (tid < 3u) atomicMin(a, s_acc[tid]); else atomicMax(a, s_acc[tid]); }
}
// the originals go once every chunk has read them: their pieces stand for them
__global__ __launch_bounds__(256) void outlier_retire(PrimRef* prims, uint32_t cap, const uint32_t* total, const Counters* ctr, const OutlierWork* work) {
  const uint32_t tot = total[0];
  if (tot == 0u || tot > cap) return;
  const uint32_t numWork = (uint32_t)(ctr->outlierWork >> 32);
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < numWork; i += gridDim.x * 256u) {
    PrimRef dead = load_prim(prims + work[i].src); dead.geom = NIL; store_prim(prims + work[i].src, dead);
  }
}
