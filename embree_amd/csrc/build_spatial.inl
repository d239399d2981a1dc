// build_spatial.inl -- RTC_BUILD_QUALITY_HIGH, the reference's default form: spatial splits INSIDE the recursion.
// Part of build.hip (included inside its anonymous namespace); see the header of build.hip for the pipeline.
//
// Reference: BVHBuilderBinnedFastSpatialSAH (kernels/builders/bvh_builder_sah.h:519-608) over HeuristicArraySpatialSAH
// (kernels/builders/heuristic_spatial_array.h) with SpatialBinInfo / SpatialBinMapping / SpatialBinSplit (heuristic_spatial.h) and the
// triangle splitter (splitter.h:16-49); chosen by kernels/bvh/bvh_builder_sah_spatial.cpp:93-160 when useSpatialPreSplits is off (the default,
// kernels/common/state.cpp:88).  What it does, per set of references:
//   * every reference carries a split budget in the top 5 bits of its geometry word: 4 + min(27, max(1, ceil(10 N area(box) / sum of areas)))   (:574-601)
//   * every set owns an EXTENDED range behind its references (the root: max_spatial_split_replications - 1 = 20 % of N); a split may create
//     at most that many new references, the rest of the range is handed down to the children in proportion to their size                 (:115-170)
//   * find(): the object split (32 centroid bins) as always; if the set has an extended range and the two children's boxes overlap by
//     >= 10 % of the set's area (and >= 5e-6 of the scene's), a spatial split is tried: 16 bins per axis over the set's GEOMETRY bounds, a
//     reference with budget is clipped bin by bin (its pieces extend the bins it crosses, numBegin / numEnd count where it starts and ends),
//     one without budget goes whole into the bin of its centre; the best plane wins if its SAH < 0.99 x the object split's and the new
//     references fit the extended range                                                                                                   (:171-200)
//   * split(): references that straddle the plane and have budget are cut in two (budget - 1 each; never into an empty piece), then the set
//     is partitioned by the bin of every reference's centre                                                                                (:266-330, :395-420)
// Here this happens in the level-synchronous TOP phase (sets of more than small_threshold references; HIGH builds lower that threshold to 256 so
// that more of the tree is covered); the sub-trees finished by one wavefront in LDS split by object only.  Differences to the reference, all in
// the direction of determinism: which straddling references are cut does not depend on an atomic counter (a reference without budget is binned
// whole, so the predicted counts are upper bounds and every budgeted straddler is cut), the extended range is shared by reference COUNT (the
// reference: by the sum of the budgets), sums are fixed point.  Binary nodes are numbered by CAPACITY (children of node k over a left capacity cL
// are k + 1 and k + 2 cL), so that references created further down cannot run into a sibling's numbers.
constexpr uint32_t GEOM_MASK = 0x07FFFFFFu;        // PrimRef.geom: geometry table index; top 5 bits: split budget (RESERVED_NUM_SPATIAL_SPLITS_GEOMID_BITS = 5, heuristic_spatial.h:10)
constexpr uint32_t SPLIT_SHIFT = 27u;
constexpr int SBINS = 16;                          // NUM_SPATIAL_BINS, kernels/builders/bvh_builder_sah.h:11
constexpr int SBINW = 8;                           // lo.xyz, hi.xyz (ordered uint), numBegin, numEnd
constexpr int SBINS_WORDS = 3 * SBINS * SBINW;     // 384 words per set
constexpr int SSLOTS = 3 * SBINS;                  // in LDS (spatial_bin) the bins are eight PLANES of 48 slots, word k of (axis, bin) at k * 48 + axis * 16 + bin: the lanes of
                                                   // an atomic fall on neighbouring banks (8-word records put them on 4 of the 32); the sets' bins in memory stay records

struct SegX {                                      // what a set of the top phase carries in addition to Seg when spatial splits are on
  uint32_t extEnd;                                 // end of the set's capacity: [begin, end) references, [end, extEnd) extended range
  uint32_t trySpatial;                             // the object split's children overlap enough: spatial bins are being filled
  uint32_t capL;                                   // capacity given to the left child (its references + its share of the extended range)
  uint32_t pad;
  float sofs[3], objSah;                           // SpatialBinMapping: ofs, scale (0 = axis invalid), inv_scale
  float sscale[3]; uint32_t pad1;
  float sinv[3]; uint32_t pad2;
};

__device__ __forceinline__ uint32_t segx_ext_end(const SegX* sx, uint32_t s) { return sx[s].extEnd; }
__device__ __forceinline__ void segx_object_split(SegX* sx, uint32_t s, uint32_t capL, float sah) { sx[s].capL = capL; sx[s].objSah = sah; sx[s].trySpatial = 0u; }
__device__ __forceinline__ uint32_t segx_cap_left(const SegX* sx, uint32_t s) { return sx[s].capL; }
__device__ __forceinline__ void segx_child(SegX* nx, uint32_t k, uint32_t extEnd) { nx[k].extEnd = extEnd; nx[k].trySpatial = 0u; nx[k].capL = 0u; }

__device__ __forceinline__ float safe_area2(const float* lo, const float* hi) {   // safeArea: 0 for an empty box, else the full surface area (bbox.h)
  if (lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]) return 0.0f;
  return 2.0f * half_area3(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]);
}
__device__ __forceinline__ int sbin(float p, float ofs, float scale) {             // SpatialBinMapping::bin: floori((p - ofs) * scale), clamped
  int i = (int)floorf((p - ofs) * scale); i = i < 0 ? 0 : i; return i > SBINS - 1 ? SBINS - 1 : i;
}
__device__ __forceinline__ float sbin_pos(int bin, float ofs, float inv) { return fmaf((float)bin, inv, ofs); }   // SpatialBinMapping::pos
// splitPolygon<3> (splitter.h:16-49): both sides' boxes of a triangle cut at `pos` on axis `dim`, intersected with the piece's current box
__device__ __forceinline__ void split_triangle(const float (&v)[3][3], uint32_t dim, float pos, const float* curLo, const float* curHi,
                                               float* Llo, float* Lhi, float* Rlo, float* Rhi) {
  for (int d = 0; d < 3; d++) { Llo[d] = __builtin_inff(); Lhi[d] = -__builtin_inff(); Rlo[d] = __builtin_inff(); Rhi[d] = -__builtin_inff(); }
  for (int e = 0; e < 3; e++) {
    const int e1 = e == 2 ? 0 : e + 1;
    const float a0 = sel3(dim, v[e][0], v[e][1], v[e][2]), a1 = sel3(dim, v[e1][0], v[e1][1], v[e1][2]);
    if (a0 <= pos) for (int d = 0; d < 3; d++) { Llo[d] = vmin(Llo[d], v[e][d]); Lhi[d] = vmax(Lhi[d], v[e][d]); }
    if (a0 >= pos) for (int d = 0; d < 3; d++) { Rlo[d] = vmin(Rlo[d], v[e][d]); Rhi[d] = vmax(Rhi[d], v[e][d]); }
    if ((a0 < pos && pos < a1) || (a1 < pos && pos < a0)) {
      const float t = (pos - a0) * (1.0f / (a1 - a0));
      for (int d = 0; d < 3; d++) { const float c = fmaf(t, v[e1][d] - v[e][d], v[e][d]); Llo[d] = vmin(Llo[d], c); Lhi[d] = vmax(Lhi[d], c); Rlo[d] = vmin(Rlo[d], c); Rhi[d] = vmax(Rhi[d], c); }
    }
  }
  // The cut points are interpolated (t = (pos - a0) / (a1 - a0), c = v + t (v' - v)): each coordinate carries a rounding error of a few ulp OF ITS MAGNITUDE, so
  // far from the origin the two pieces' boxes could leave a sliver of the triangle along the cut uncovered (crown stand-in at 1e5: 7 of 16384 rays slipped
  // through, tests/test_gpu_round2.py::test_fast_mode_far_from_the_origin).  Both boxes are therefore widened by 4 ulp of their largest coordinate before they
  // are clamped to the piece that is being cut (whose box is conservative by induction: the first one is the triangle's exact box).
  for (int d = 0; d < 3; d++) {
    const float eL = 4.76837158e-7f * vmax(fabsf(Llo[d]), fabsf(Lhi[d])), eR = 4.76837158e-7f * vmax(fabsf(Rlo[d]), fabsf(Rhi[d]));
    if (Llo[d] <= Lhi[d]) { Llo[d] -= eL; Lhi[d] += eL; }
    if (Rlo[d] <= Rhi[d]) { Rlo[d] -= eR; Rhi[d] += eR; }
  }
  for (int d = 0; d < 3; d++) { Llo[d] = vmax(Llo[d], curLo[d]); Lhi[d] = vmin(Lhi[d], curHi[d]); Rlo[d] = vmax(Rlo[d], curLo[d]); Rhi[d] = vmin(Rhi[d], curHi[d]); }
}
__device__ __forceinline__ bool box_empty(const float* lo, const float* hi) { return lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]; }
__device__ __forceinline__ void load_tri_masked(const GeomDesc* geoms, const PrimRef& r, float (&v)[3][3]) {
  PrimRef q = r; q.geom &= GEOM_MASK; load_tri(geoms, q, v);
}

// ---- split budgets (bvh_builder_sah.h:574-601): two passes over the references, the sum in fixed point relative to the scene's area
// fromCtr (one-round-trip commits): n is an upper bound, the number of valid references is on the device; nothing to do when the whole scene is one small sub-tree
__global__ __launch_bounds__(256) void spatial_area_sum(const PrimRef* prims, uint32_t n, Counters* ctr, uint32_t fromCtr) {
  __shared__ unsigned long long s_w[4];
  if (fromCtr) { if (ctr->numSegs == 0u) return; n = ctr->numPrims; }
  const float rootArea2 = 2.0f * ctr->rootArea;
  unsigned long long acc = 0ull;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const PrimRef r = load_prim(prims + i);
    const float a = 2.0f * half_area3(r.hi[0] - r.lo[0], r.hi[1] - r.lo[1], r.hi[2] - r.lo[2]);
    if (rootArea2 > 0.0f) acc += (unsigned long long)((double)(a / rootArea2) * 4294967296.0);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63u) == 0u) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0u) atomicAdd(&ctr->stripe[blockIdx.x % Counters::STRIPES].areaFixed, s_w[0] + s_w[1] + s_w[2] + s_w[3]);   // (<= 2048 workgroups on one word were half of this kernel: see Counters::stripe)
}
__global__ __launch_bounds__(256) void spatial_budgets(PrimRef* prims, uint32_t n, const Counters* ctr, uint32_t fromCtr) {
  __shared__ unsigned long long s_sum;
  if (fromCtr) { if (ctr->numSegs == 0u) return; n = ctr->numPrims; }
  if (threadIdx.x < 64u) {                                                         // the sum spatial_area_sum left in the stripes (an integer sum: any order)
    unsigned long long v = threadIdx.x < Counters::STRIPES ? ctr->stripe[threadIdx.x].areaFixed : 0ull;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (threadIdx.x == 0u) s_sum = v;
  }
  __syncthreads();
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const float rootArea2 = 2.0f * ctr->rootArea;
  const double sumRel = (double)s_sum / 4294967296.0;                              // sum of the boxes' areas / scene area
  PrimRef r = load_prim(prims + i);
  const float a = 2.0f * half_area3(r.hi[0] - r.lo[0], r.hi[1] - r.lo[1], r.hi[2] - r.lo[2]);
  int k = 1;
  if (rootArea2 > 0.0f && sumRel > 0.0) { const float nf = ceilf((float)(10.0 * (double)n * ((double)(a / rootArea2) / sumRel))); k = nf > 27.0f ? 27 : (nf < 1.0f ? 1 : (int)nf); }
  const uint32_t budget = 4u + (uint32_t)(k > 27 ? 27 : k);                         // 4 + min(maxSplits - 4, max(1, nf)), maxSplits = 31
  r.geom = (r.geom & GEOM_MASK) | (budget << SPLIT_SHIFT);
  store_prim(prims + i, r);
}

__global__ void segx_root(SegX* sx, uint32_t extEnd) { SegX x0{}; x0.extEnd = extEnd; sx[0] = x0; }   // the root set owns everything behind the references

// ---- per level, after top_split: does the object split leave overlapping children?  (HeuristicArraySpatialSAH::find, heuristic_spatial_array.h:171-186)
__global__ void spatial_decide(const Seg* segs, SegX* sx, const BNode* bnodes, uint32_t* sbins, const Counters* ctr, uint32_t minSize, uint32_t* sbinsTop) {
  const uint32_t s = blockIdx.x, tid = threadIdx.x;
  if (s >= ctr->numSegs) return;
  const Seg* sg = segs + s; SegX* x = sx + s;
  __shared__ uint32_t s_try;
  if (tid == 0u) {
    uint32_t t = 0u;
    const uint32_t ext = x->extEnd - sg->end;
    if (ext > 0u && !(sg->flags & 1u) && sg->end - sg->begin >= minSize) {
      const BNode& P = bnodes[sg->bnode]; const BNode& L = bnodes[sg->childL]; const BNode& R = bnodes[sg->childR];
      float olo[3], ohi[3];
      for (int d = 0; d < 3; d++) { olo[d] = fmaxf(L.lo[d], R.lo[d]); ohi[d] = fminf(L.hi[d], R.hi[d]); }
      const float ao = safe_area2(olo, ohi), rootA = 2.0f * ctr->rootArea, setA = safe_area2(P.lo, P.hi);
      if (ao >= 0.000005f * rootA && ao >= 0.1f * setA) {                           // SPATIAL_ASPLIT_AREA_THRESHOLD, SPATIAL_ASPLIT_OVERLAP_THRESHOLD
        t = 1u;
        for (int d = 0; d < 3; d++) {                                               // SpatialBinMapping(pinfo), heuristic_spatial.h:24-32
          const float eps = 128.0f * 1.1920929e-07f * fmaxf(fabsf(P.lo[d]), fabsf(P.hi[d]));
          const float diag = fmaxf(eps, P.hi[d] - P.lo[d]);
          x->sscale[d] = (P.hi[d] - P.lo[d] <= eps) ? 0.0f : (float)SBINS / diag;
          x->sofs[d] = P.lo[d];
          x->sinv[d] = 1.0f / x->sscale[d];
        }
      }
    }
    x->trySpatial = t; s_try = t;
  }
  __syncthreads();
  if (s_try) {
    // (round 6) levels of at most ACC_SETS sets: the chunks of a set merge into ACC_REPL copies of its spatial bins (as top_bin's, build_top.inl), spatial_best folds them
    const bool copies = sbinsTop && ctr->numSegs <= ACC_SETS && sg->end - sg->begin > CHUNK;
    uint32_t* b = copies ? sbinsTop + (size_t)s * ACC_REPL * SBINS_WORDS : sbins + (size_t)s * SBINS_WORDS;
    for (uint32_t w = tid; w < (uint32_t)SBINS_WORDS * (copies ? ACC_REPL : 1u); w += blockDim.x) { const uint32_t k = (w % (uint32_t)SBINS_WORDS) % SBINW; b[w] = k < 3 ? ENC_POS_INF : (k < 6 ? ENC_NEG_INF : 0u); }
  }
}

// ---- SpatialBinInfo::bin2 (heuristic_spatial.h:160-222) over the chunks of the sets that try a spatial split
__device__ __forceinline__ void sbin_extend(uint32_t* bins, int dim, int bin, const float* lo, const float* hi) {
  if (box_empty(lo, hi)) return;
  uint32_t* e = bins + (dim * SBINS + bin);
  for (int d = 0; d < 3; d++) { atomicMin(&e[d * SSLOTS], enc(lo[d])); atomicMax(&e[(3 + d) * SSLOTS], enc(hi[d])); }
}
// The chain of cuts bin2 makes for ONE reference with budget on ONE axis (:186-218): the reference is clipped bin by bin from its first to its last bin;
// l = the bin it is counted to begin in (numBegin), rr = the bin it is counted to end in (numEnd) -- a plane p then sees it on the left iff l < p and on the
// right iff rr >= p.  EXTEND: the pieces also extend the bins' boxes (binning); without it only (l, rr) come back, which is how spatial_partition decides
// the sides: by construction neither side can receive more references than spatial_best predicted from the counts.
// The triangle's vertices are only fetched (once, `have`) when the reference spans more than one bin on some axis: most do not.
template <bool EXTEND>
__device__ __forceinline__ void spatial_chain(float (&v)[3][3], bool& have, const GeomDesc* geoms, const PrimRef& r, int d, float ofs, float scale, float inv, uint32_t* bins, int& l, int& rr) {
  const float rlo = sel3((uint32_t)d, r.lo[0], r.lo[1], r.lo[2]), rhi = sel3((uint32_t)d, r.hi[0], r.hi[1], r.hi[2]);
  l = sbin(rlo, ofs, scale); rr = sbin(rhi, ofs, scale);
  if (l == rr) { if (EXTEND) sbin_extend(bins, d, l, r.lo, r.hi); return; }
  if (!have) { load_tri_masked(geoms, r, v); have = true; }
  int bs = l, be = rr;
  float restLo[3] = {r.lo[0], r.lo[1], r.lo[2]}, restHi[3] = {r.hi[0], r.hi[1], r.hi[2]};
  while (bs < be && sbin_pos(bs + 1, ofs, inv) <= rlo) bs++;                       // "assure that split position always overlaps the primitive bounds"
  while (bs < be && sbin_pos(be, ofs, inv) >= rhi) be--;
  int bin = bs;
  for (; bin < be; bin++) {
    float Llo[3], Lhi[3], Rlo[3], Rhi[3];
    split_triangle(v, (uint32_t)d, sbin_pos(bin + 1, ofs, inv), restLo, restHi, Llo, Lhi, Rlo, Rhi);
    if (box_empty(Llo, Lhi)) l++;
    if (EXTEND) sbin_extend(bins, d, bin, Llo, Lhi);
    for (int k = 0; k < 3; k++) { restLo[k] = Rlo[k]; restHi[k] = Rhi[k]; }
  }
  if (box_empty(restLo, restHi)) rr--;
  if (EXTEND) sbin_extend(bins, d, bin, restLo, restHi);
  rr = rr < 0 ? 0 : rr; l = l > rr ? rr : l;                                       // degenerate pieces: still counted on one side of every plane
}
// The clipping chains are not run where they are found: a reference that spans several bins of an axis is the exception in a batch of 64, its chain is
// up to 15 dependent cuts long, and a wave that runs it on the spot waits for its longest chain -- three times per batch, once per axis (spatial_bin was 8.2
// of the 17.5 ms of a HIGH commit; parked chains: 7.0 of 16.4.  At the lower levels, where a bin is about as wide as a triangle, nearly every pair is a
// chain and the LDS atomics of the pieces -- six per piece -- are what is left).  The references that need clipping go to a list in LDS instead (round 4: one entry per reference with its axes, not one per (reference, axis) pair -- 8 KB instead of 24, and the vertices are
// fetched once), and once the chunk's simple references are binned the workgroup's 256 lanes take one reference each.  What a chain adds to the bins is min / max / add atomics, so the bins do not depend on the order the chains are run in.
#ifndef MI355_SBIN_COPIES
#define MI355_SBIN_COPIES 8
#endif
constexpr uint32_t SBIN_COPIES = MI355_SBIN_COPIES, SCOPY_STRIDE = SBINS_WORDS + 1u;
constexpr uint32_t CHAIN_CAP = CHUNK;                                // every reference of a chunk fits (one task per reference: its axes in the top three bits)
__global__ __launch_bounds__(256) void spatial_bin(const Seg* segs, const SegX* sx, const Chunk* chunks, const PrimRef* src, const GeomDesc* geoms, uint32_t* sbins, const Counters* ctr, uint32_t* sbinsTop) {
  __shared__ uint32_t s_b[SBIN_COPIES * SCOPY_STRIDE];              // private copies of the bins, lane l works on copy l mod SBIN_COPIES (see bins_add_copies); folded below
  __shared__ uint32_t s_chain[CHAIN_CAP];                           // reference index | axes to clip on << 29
  __shared__ uint32_t s_numChains;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t numChunks = ctr->numChunks, c0 = blockIdx.x;
  if (c0 >= numChunks) return;
  Chunk ck = chunks[c0];
  const SegX* x = sx + ck.seg;
  if (!x->trySpatial) return;
  for (uint32_t w = tid; w < (uint32_t)SBINS_WORDS; w += 256u) {
    const uint32_t k = w / SSLOTS, v = k < 3 ? ENC_POS_INF : (k < 6 ? ENC_NEG_INF : 0u);
#pragma unroll
    for (uint32_t c = 0; c < SBIN_COPIES; c++) s_b[c * SCOPY_STRIDE + w] = v;
  }
  uint32_t* const mine = s_b + (lane & (SBIN_COPIES - 1u)) * SCOPY_STRIDE;
  if (tid == 0u) s_numChains = 0u;
  __syncthreads();
  float ofs[3], scale[3], inv[3];
  for (int d = 0; d < 3; d++) { ofs[d] = x->sofs[d]; scale[d] = x->sscale[d]; inv[d] = x->sinv[d]; }
  const uint32_t first = ck.begin, last = ck.end;                 // (one chunk per workgroup: doing several in a row left the top levels with too few workgroups)
  {
    // each wave owns a contiguous quarter of the chunk (as in top_bin): a row of 16 lanes sees 16 consecutive references
    const uint32_t span = ck.begin + (tid >> 6) * (CHUNK / 4u), spanEnd = min(span + CHUNK / 4u, ck.end);
    for (uint32_t i0 = span; i0 < spanEnd; i0 += 64u) {           // wave-uniform trip count
      const uint32_t i = i0 + lane; const bool v = i < spanEnd;
      PrimRef r{}; if (v) r = load_prim(src + i);
      const uint32_t budget = r.geom >> SPLIT_SHIFT;
      uint32_t c6[6], chainAxes = 0u;
      for (int k = 0; k < 3; k++) { c6[k] = enc(r.lo[k]); c6[3 + k] = enc(r.hi[k]); }
#pragma unroll
      for (int d = 0; d < 3; d++) {
        // a reference without budget goes whole into the bin of its centre on every axis (:170-178); one with budget into the bin its box lies in, if that
        // is ONE bin; otherwise it is clipped bin by bin (spatial_chain, below) -- on axes the mapping is valid for (mapping.invalid(dim))
        bool simple = false; uint32_t b = 0u;
        if (v) {
          if (budget <= 1u) { simple = true; b = (uint32_t)sbin(0.5f * (r.lo[d] + r.hi[d]), ofs[d], scale[d]); }
          else if (scale[d] != 0.0f) {
            const int l = sbin(r.lo[d], ofs[d], scale[d]), rr = sbin(r.hi[d], ofs[d], scale[d]);
            if (l == rr) { simple = true; b = (uint32_t)l; }
            else chainAxes |= 1u << d;
          }
        }
        if (simple) {
          uint32_t* e = mine + (d * SBINS + b);
          atomicMin(&e[0], c6[0]); atomicMin(&e[SSLOTS], c6[1]); atomicMin(&e[2 * SSLOTS], c6[2]);
          atomicMax(&e[3 * SSLOTS], c6[3]); atomicMax(&e[4 * SSLOTS], c6[4]); atomicMax(&e[5 * SSLOTS], c6[5]);
          atomicAdd(&e[6 * SSLOTS], 1u); atomicAdd(&e[7 * SSLOTS], 1u);
        }
      }
      if (chainAxes) s_chain[atomicAdd(&s_numChains, 1u)] = (i - first) | (chainAxes << 29);   // (at most one per reference of the chunk: the list cannot overflow)
    }
  }
  __syncthreads();
  {
    const uint32_t numChains = min(s_numChains, CHAIN_CAP);
    // few references to clip (the upper levels: big triangles, chains of up to 15 cuts): a lane per (reference, axis), the chains of a reference side by side;
    // many (the lower levels): a lane per reference, which fetches the vertices once (measured both ways on the crown and the powerplant stand-ins)
    const bool perAxis = numChains * 3u <= 256u;
    for (uint32_t t = tid; t < (perAxis ? numChains * 3u : numChains); t += 256u) {
      const uint32_t entry = s_chain[perAxis ? t / 3u : t];
      const uint32_t task = perAxis ? ((entry & 0x1FFFFFFFu) | (entry & (0x20000000u << (t % 3u)))) : entry;
      const PrimRef r = load_prim(src + first + (task & 0x1FFFFFFFu));
      float tv[3][3]; bool have = false;                         // (the vertices are fetched once for all axes of the reference: one task per (reference, axis) fetched them up to three times)
#pragma unroll
      for (uint32_t d = 0; d < 3u; d++) {
        if (!((task >> (29u + d)) & 1u)) continue;
        int l2, r2;
        spatial_chain<true>(tv, have, geoms, r, (int)d, ofs[d], scale[d], inv[d], mine, l2, r2);
        atomicAdd(&mine[6 * SSLOTS + d * SBINS + (uint32_t)l2], 1u); atomicAdd(&mine[7 * SSLOTS + d * SBINS + (uint32_t)r2], 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t w = tid; w < (uint32_t)SBINS_WORDS; w += 256u) {    // fold the copies into copy 0
    const uint32_t k = w / SSLOTS; uint32_t v = s_b[w];
#pragma unroll
    for (uint32_t c = 1; c < SBIN_COPIES; c++) { const uint32_t y = s_b[c * SCOPY_STRIDE + w]; v = k < 3 ? min(v, y) : (k < 6 ? max(v, y) : v + y); }
    s_b[w] = v;
  }
  __syncthreads();
  const Seg* sg = segs + ck.seg;
  uint32_t* g = sbins + (size_t)ck.seg * SBINS_WORDS;
  if (first == sg->begin && last == sg->end) { for (uint32_t w = tid; w < (uint32_t)SBINS_WORDS; w += 256u) g[w] = s_b[(w % SBINW) * SSLOTS + w / SBINW]; return; }   // the whole set: these ARE its bins
  if (sbinsTop && ctr->numSegs <= ACC_SETS) g = sbinsTop + ((size_t)ck.seg * ACC_REPL + (blockIdx.x & (ACC_REPL - 1u))) * SBINS_WORDS;   // (a set of several chunks at an upper level: one of its copies)
  for (uint32_t w = tid; w < (uint32_t)SBINS_WORDS; w += 256u) {
    const uint32_t k = w % SBINW, v = s_b[k * SSLOTS + w / SBINW];
    if (k < 3) { if (v != ENC_POS_INF) atomicMin(&g[w], v); } else if (k < 6) { if (v != ENC_NEG_INF) atomicMax(&g[w], v); } else if (v) atomicAdd(&g[w], v);
  }
}

// ---- SpatialBinInfo::best (heuristic_spatial.h:285-358) + the decision of find() (:187-198); one wavefront per set, lane d = axis d
__global__ __launch_bounds__(64) void spatial_best(Seg* segs, SegX* sx, const uint32_t* sbins, BNode* bnodes, const Counters* ctr, Params prm, const uint32_t* sbinsTop) {
  __shared__ float s_sah[3]; __shared__ uint32_t s_pos[3], s_l[3], s_r[3];
  const uint32_t s = blockIdx.x, lane = threadIdx.x;
  if (s >= ctr->numSegs) return;
  Seg* sg = segs + s; SegX* x = sx + s;
  if (!x->trySpatial) return;
  __shared__ uint32_t s_B[SBINS_WORDS];                            // the set's bins, fetched by the whole wave at once (three lanes walking them in global memory: 13 us per level)
  if (sbinsTop && ctr->numSegs <= ACC_SETS && sg->end - sg->begin > CHUNK) {          // the fold of the set's copies (all of them asked for before the first is used)
    constexpr uint32_t PER = (uint32_t)SBINS_WORDS / 64u;
    uint32_t xx[PER][ACC_REPL];
#pragma unroll
    for (uint32_t i = 0; i < PER; i++)
#pragma unroll
      for (uint32_t r = 0; r < ACC_REPL; r++) xx[i][r] = sbinsTop[((size_t)s * ACC_REPL + r) * SBINS_WORDS + i * 64u + lane];
#pragma unroll
    for (uint32_t i = 0; i < PER; i++) {
      const uint32_t w = i * 64u + lane, k = w % SBINW; uint32_t v = xx[i][0];
#pragma unroll
      for (uint32_t r = 1; r < ACC_REPL; r++) v = k < 3u ? min(v, xx[i][r]) : (k < 6u ? max(v, xx[i][r]) : v + xx[i][r]);
      s_B[w] = v;
    }
  } else
  for (uint32_t w = lane; w < (uint32_t)SBINS_WORDS; w += 64u) s_B[w] = sbins[(size_t)s * SBINS_WORDS + w];
  __syncthreads();
  const uint32_t* B = s_B;
  const uint32_t add = (1u << prm.shift) - 1u;
  if (lane < 3u) {
    const uint32_t d = lane;
    float rA[SBINS]; uint32_t rC[SBINS];
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    uint32_t cnt = 0;
    for (int i = SBINS - 1; i > 0; i--) {
      const uint32_t* e = B + (d * SBINS + i) * SBINW;
      cnt += e[7]; rC[i] = cnt;
      for (int k = 0; k < 3; k++) { lo[k] = vmin(lo[k], dec(e[k])); hi[k] = vmax(hi[k], dec(e[3 + k])); }
      rA[i] = (lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]) ? 0.0f : half_area3(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]);
    }
    for (int k = 0; k < 3; k++) { lo[k] = __builtin_inff(); hi[k] = -__builtin_inff(); }
    cnt = 0; float best = __builtin_inff(); uint32_t bpos = 0, bl = 0, br = 0;
    for (int i = 1; i < SBINS; i++) {
      const uint32_t* e = B + (d * SBINS + (i - 1)) * SBINW;
      cnt += e[6];
      for (int k = 0; k < 3; k++) { lo[k] = vmin(lo[k], dec(e[k])); hi[k] = vmax(hi[k], dec(e[3 + k])); }
      const float lA = (lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]) ? 0.0f : half_area3(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]);
      const float sah = fmaf(lA, (float)((cnt + add) >> prm.shift), rA[i] * (float)((rC[i] + add) >> prm.shift));
      if (sah < best) { best = sah; bpos = (uint32_t)i; bl = cnt; br = rC[i]; }
    }
    s_sah[d] = best; s_pos[d] = bpos; s_l[d] = bl; s_r[d] = br;
  }
  __syncthreads();
  if (lane == 0u) {
    float bestSah = __builtin_inff(); int bestDim = -1;
    for (int d = 0; d < 3; d++) { if (x->sscale[d] == 0.0f) continue; if (s_sah[d] < bestSah && s_pos[d] != 0u) { bestDim = d; bestSah = s_sah[d]; } }
    const uint32_t begin = sg->begin, end = sg->end, n = end - begin, ext = x->extEnd - end;
    bool spatial = false;
    if (bestDim >= 0) {
      const uint32_t l = s_l[bestDim], r = s_r[bestDim];
      spatial = bestSah < 0.99f * x->objSah && l + r >= n && l + r - n <= ext && l > 0u && r > 0u;   // SPATIAL_ASPLIT_SAH_THRESHOLD; the new references fit the extended range
      if (spatial) {
        const uint32_t extLeft = x->extEnd - begin - (l + r);                       // what is left of the extended range after the split's new references
        const uint32_t capL = l + (uint32_t)floorf((float)l / (float)(l + r) * (float)extLeft);
        const uint32_t idL = sg->bnode + 1u, idR = sg->bnode + 2u * capL;
        BNode* par = bnodes + sg->bnode; par->left = idL; par->right = idR; par->splitSah = bestSah;
        BNode L{}, R{};
        L.begin = begin; L.end = begin + l; R.begin = begin + capL; R.end = begin + capL + r;
        L.left = L.right = R.left = R.right = NIL; L.splitSah = R.splitSah = __builtin_inff();
        bnodes[idL] = L; bnodes[idR] = R;                                            // bounds and the final counts follow in top_emit (the partition knows them)
        sg->flags = 2u; sg->dim = (uint32_t)bestDim; sg->pos = s_pos[bestDim]; sg->nL = l;
        sg->childL = idL; sg->childR = idR; sg->curL = begin; sg->curR = begin + capL;
        x->capL = capL;
        for (int side = 0; side < 2; side++) for (int k = 0; k < 12; k++) sg->acc[side][k] = (k % 6 < 3) ? ENC_POS_INF : ENC_NEG_INF;
      }
    }
    if (!spatial) x->trySpatial = 0u;
  }
}

// ---- create_spatial_splits + the partition by centre (heuristic_spatial_array.h:266-330, :395-420) for the chunks of the sets that split spatially.
// Two passes over the chunk: the first decides the side(s) of every reference and counts, ONE pair of atomics then reserves the chunk's places behind the
// set's two cursors (a reservation per round of 256 made the big sets of the top levels wait for 37,000 same-address atomics: 1.5 ms a level), the second
// reads the references again (L2), cuts the straddling ones and writes.
__device__ __forceinline__ void spatial_sides(const PrimRef& r, uint32_t dim, int pos, float ofs, float scale, float inv, const GeomDesc* geoms, float (&tv)[3][3], bool& toL, bool& toR) {
  const uint32_t budget = r.geom >> SPLIT_SHIFT;
  const float rlo = sel3(dim, r.lo[0], r.lo[1], r.lo[2]), rhi = sel3(dim, r.hi[0], r.hi[1], r.hi[2]);
  if (budget > 1u) {                                             // the sides binning counted it on (spatial_chain)
    bool have = false; int l, rr;
    spatial_chain<false>(tv, have, geoms, r, (int)dim, ofs, scale, inv, nullptr, l, rr);
    toL = l < pos; toR = rr >= pos;
    if (toL && toR) {                                            // straddles the plane (so l != rr: the vertices are there): cut it, unless a piece would be empty
      float Llo[3], Lhi[3], Rlo[3], Rhi[3];
      split_triangle(tv, dim, sbin_pos(pos, ofs, inv), r.lo, r.hi, Llo, Lhi, Rlo, Rhi);
      if (box_empty(Llo, Lhi)) toL = false; else if (box_empty(Rlo, Rhi)) toR = false;
    }
  } else { toL = sbin(0.5f * (rlo + rhi), ofs, scale) < pos; toR = !toL; }   // whole, to the side its centre lies on
}
__global__ __launch_bounds__(256) void spatial_partition(Seg* segs, const SegX* sx, const Chunk* chunks, const PrimRef* src, PrimRef* dst, const GeomDesc* geoms, Counters* ctr, uint32_t* chunkFlag, uint32_t* accTop) {
  __shared__ uint32_t s_cnt[CHUNK_ROUNDS][4][2], s_off[CHUNK_ROUNDS][4][2], s_baseL, s_baseR, s_mineL, s_mineR;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (blockIdx.x >= ctr->numChunks) return;
  const Chunk ck = chunks[blockIdx.x];
  Seg* sg = segs + ck.seg; const SegX* x = sx + ck.seg;
  if (!(sg->flags & 2u)) return;
  const uint32_t dim = sg->dim; const int pos = (int)sg->pos;
  const float ofs = sel3(dim, x->sofs[0], x->sofs[1], x->sofs[2]), scale = sel3(dim, x->sscale[0], x->sscale[1], x->sscale[2]), inv = sel3(dim, x->sinv[0], x->sinv[1], x->sinv[2]);
  const float fpos = sbin_pos(pos, ofs, inv);
  uint32_t bitsL = 0u, bitsR = 0u;                               // this thread's decisions, one bit per round
  unsigned long long lm[CHUNK_ROUNDS], rm[CHUNK_ROUNDS];
#pragma unroll
  for (int round = 0; round < CHUNK_ROUNDS; round++) {
    const uint32_t i = ck.begin + (uint32_t)round * 256u + tid;
    bool toL = false, toR = false;
    if (i < ck.end) { const PrimRef r = load_prim(src + i); float tv[3][3]; spatial_sides(r, dim, pos, ofs, scale, inv, geoms, tv, toL, toR); }
    lm[round] = __ballot(toL); rm[round] = __ballot(toR);
    if (lane == 0u) { s_cnt[round][wave][0] = (uint32_t)__popcll(lm[round]); s_cnt[round][wave][1] = (uint32_t)__popcll(rm[round]); }
    if (toL) bitsL |= 1u << round;
    if (toR) bitsR |= 1u << round;
  }
  __syncthreads();
  if (tid == 0u) {
    uint32_t l = 0, rr = 0;
    for (int r = 0; r < CHUNK_ROUNDS; r++) for (int w = 0; w < 4; w++) { s_off[r][w][0] = l; s_off[r][w][1] = rr; l += s_cnt[r][w][0]; rr += s_cnt[r][w][1]; }
    // The chunk's places: behind everything the chunks BEFORE it in the set send to the same side -- every chunk says how many that is (one word: flag, left,
    // right) and sums what its predecessors said, waiting for those that have not yet (workgroups start in index order: a predecessor is running or done).
    // Reserving the places with the set's cursors handed them out in the order the chunks ARRIVED: the children held the same references in another order
    // from run to run, and what a median split further down cuts off depends on the order (two leaves of a HIGH tree swapped a triangle between commits).
    __hip_atomic_store(chunkFlag + blockIdx.x, 0x80000000u | (l << 12) | rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_mineL = l; s_mineR = rr;
    s_baseL = sg->begin; s_baseR = sg->begin + x->capL;
  }
  __syncthreads();
  {
    uint32_t pl = 0u, pr = 0u;
    for (uint32_t j = sg->chunk0 + tid; j < blockIdx.x; j += 256u) {
      uint32_t v, spins = 0u;
      // (the wait is BOUNDED.  It relies on workgroups being started in index order -- a predecessor is then running or done -- which the hardware does and HIP
      // does not promise: should a predecessor ever be kept off the chip by its waiting successors, they give up after ~2^20 naps (a tenth of a second), the commit reports
      // overflow = 4 and the host returns an error instead of the GPU hanging for ever)
      do { v = __hip_atomic_load(chunkFlag + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (!(v >> 31)) { __builtin_amdgcn_s_sleep(2); if (++spins > (1u << 20)) { atomicMax(&ctr->overflow, 4u); v = 0x80000000u; } } } while (!(v >> 31));
      pl += (v >> 12) & 0xFFFu; pr += v & 0xFFFu;
    }
    for (int o = 32; o > 0; o >>= 1) { pl += (uint32_t)__shfl_down((int)pl, o, 64); pr += (uint32_t)__shfl_down((int)pr, o, 64); }
    if (lane == 0u && (pl | pr)) { atomicAdd(&s_baseL, pl); atomicAdd(&s_baseR, pr); }
  }
  __syncthreads();
  // where the children end (top_emit reads it): the set's LAST chunk knows -- its own places end there.  (Every chunk used to add its counts to the set's cursors: two
  // atomics per chunk on the line all chunks of the set read their plane from, 2 x 2325 of them at the root's level.)
  if (tid == 0u && ck.end == sg->end) { sg->curL = s_baseL + s_mineL; sg->curR = s_baseR + s_mineR; }
  uint32_t acc[2][12];                                           // this thread's share of the children's centroid / geometry bounds (ordered uint), folded across the wave at the end
  for (int side = 0; side < 2; side++) for (int k = 0; k < 12; k++) acc[side][k] = (k % 6 < 3) ? ENC_POS_INF : ENC_NEG_INF;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const uint32_t limL = sg->begin + x->capL, limR = x->extEnd;
#pragma unroll
  for (int round = 0; round < CHUNK_ROUNDS; round++) {
    const bool toL = (bitsL >> round) & 1u, toR = (bitsR >> round) & 1u;
    if (!toL && !toR) continue;
    const uint32_t i = ck.begin + (uint32_t)round * 256u + tid;
    const PrimRef r = load_prim(src + i);
    PrimRef L = r, R = r;
    if (toL && toR) {                                            // the cut of pass one, now for its boxes
      float tv[3][3]; load_tri_masked(geoms, r, tv);
      float Llo[3], Lhi[3], Rlo[3], Rhi[3];
      split_triangle(tv, dim, fpos, r.lo, r.hi, Llo, Lhi, Rlo, Rhi);
      for (int d = 0; d < 3; d++) { L.lo[d] = Llo[d]; L.hi[d] = Lhi[d]; R.lo[d] = Rlo[d]; R.hi[d] = Rhi[d]; }
      const uint32_t budget = r.geom >> SPLIT_SHIFT;
      L.geom = (r.geom & GEOM_MASK) | ((budget - 1u) << SPLIT_SHIFT); R.geom = L.geom;
    }
#pragma unroll
    for (int side = 0; side < 2; side++) {
      if (!(side ? toR : toL)) continue;
      const PrimRef& q = side ? R : L;
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const uint32_t cc = enc(q.lo[d] + q.hi[d]), el = enc(q.lo[d]), eh = enc(q.hi[d]);
        acc[side][d] = min(acc[side][d], cc); acc[side][3 + d] = max(acc[side][3 + d], cc);
        acc[side][6 + d] = min(acc[side][6 + d], el); acc[side][9 + d] = max(acc[side][9 + d], eh);
      }
    }
    // (the limits cannot be reached -- the counts of spatial_best are upper bounds -- but a store past a set's capacity would corrupt a sibling: guarded)
    const uint32_t oL = s_baseL + s_off[round][wave][0] + (uint32_t)__popcll(lm[round] & lt), oR = s_baseR + s_off[round][wave][1] + (uint32_t)__popcll(rm[round] & lt);
    if (toL) { if (oL < limL) store_prim(dst + oL, L); else atomicMax(&ctr->overflow, 2u); }
    if (toR) { if (oR < limR) store_prim(dst + oR, R); else atomicMax(&ctr->overflow, 2u); }
  }
  // fold the bounds: lanes of a wave (DPP), the four waves of the workgroup (LDS), then 24 atomics per chunk -- atomics of many chunks on the one cache line
  // of a big set's record are what this kernel waited for (one per wave: 0.95 of the 1.14 ms of the root's level)
  __shared__ uint32_t s_acc[2][12];
  if (tid < 24u) s_acc[tid / 12u][tid % 12u] = (tid % 6u < 3u) ? ENC_POS_INF : ENC_NEG_INF;
  __syncthreads();
#pragma unroll
  for (int side = 0; side < 2; side++)
#pragma unroll
    for (int k = 0; k < 12; k++) {
      const bool lo = (k % 6) < 3;
      const uint32_t val = lo ? wave_umin63(acc[side][k]) : wave_umax63(acc[side][k]);
      if (lane == 63u) { if (lo) atomicMin(&s_acc[side][k], val); else atomicMax(&s_acc[side][k], val); }
    }
  __syncthreads();
  if (tid < 24u) {
    const uint32_t side = tid / 12u, k = tid % 12u, val = s_acc[side][k];
    uint32_t* const a = acc_copy(accTop, ctr->numSegs, sg, ck.seg, blockIdx.x) + side * 12u + k;
    if (k % 6u < 3u) { if (val != ENC_POS_INF) atomicMin(a, val); } else { if (val != ENC_NEG_INF) atomicMax(a, val); }
  }
}
