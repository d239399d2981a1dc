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
    if (a0 <= pos) for (int d = 0; d < 3; d++) { Llo[d] = fminf(Llo[d], v[e][d]); Lhi[d] = fmaxf(Lhi[d], v[e][d]); }
    if (a0 >= pos) for (int d = 0; d < 3; d++) { Rlo[d] = fminf(Rlo[d], v[e][d]); Rhi[d] = fmaxf(Rhi[d], v[e][d]); }
    if ((a0 < pos && pos < a1) || (a1 < pos && pos < a0)) {
      const float t = (pos - a0) * (1.0f / (a1 - a0));
      for (int d = 0; d < 3; d++) { const float c = fmaf(t, v[e1][d] - v[e][d], v[e][d]); Llo[d] = fminf(Llo[d], c); Lhi[d] = fmaxf(Lhi[d], c); Rlo[d] = fminf(Rlo[d], c); Rhi[d] = fmaxf(Rhi[d], c); }
    }
  }
  for (int d = 0; d < 3; d++) { Llo[d] = fmaxf(Llo[d], curLo[d]); Lhi[d] = fminf(Lhi[d], curHi[d]); Rlo[d] = fmaxf(Rlo[d], curLo[d]); Rhi[d] = fminf(Rhi[d], curHi[d]); }
}
__device__ __forceinline__ bool box_empty(const float* lo, const float* hi) { return lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]; }
__device__ __forceinline__ void load_tri_masked(const GeomDesc* geoms, const PrimRef& r, float (&v)[3][3]) {
  PrimRef q = r; q.geom &= GEOM_MASK; load_tri(geoms, q, v);
}

// ---- split budgets (bvh_builder_sah.h:574-601): two passes over the references, the sum in fixed point relative to the scene's area
__global__ __launch_bounds__(256) void spatial_area_sum(const PrimRef* prims, uint32_t n, Counters* ctr) {
  __shared__ unsigned long long s_w[4];
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
  if (threadIdx.x == 0u) atomicAdd(&ctr->areaFixed, s_w[0] + s_w[1] + s_w[2] + s_w[3]);
}
__global__ __launch_bounds__(256) void spatial_budgets(PrimRef* prims, uint32_t n, const Counters* ctr) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const float rootArea2 = 2.0f * ctr->rootArea;
  const double sumRel = (double)ctr->areaFixed / 4294967296.0;                     // sum of the boxes' areas / scene area
  PrimRef r = load_prim(prims + i);
  const float a = 2.0f * half_area3(r.hi[0] - r.lo[0], r.hi[1] - r.lo[1], r.hi[2] - r.lo[2]);
  int k = 1;
  if (rootArea2 > 0.0f && sumRel > 0.0) { const float nf = ceilf((float)(10.0 * (double)n * ((double)(a / rootArea2) / sumRel))); k = nf > 27.0f ? 27 : (nf < 1.0f ? 1 : (int)nf); }
  const uint32_t budget = 4u + (uint32_t)(k > 27 ? 27 : k);                         // 4 + min(maxSplits - 4, max(1, nf)), maxSplits = 31
  r.geom = (r.geom & GEOM_MASK) | (budget << SPLIT_SHIFT);
  store_prim(prims + i, r);
}

// ---- per level, after top_split: does the object split leave overlapping children?  (HeuristicArraySpatialSAH::find, heuristic_spatial_array.h:171-186)
__global__ void spatial_decide(const Seg* segs, SegX* sx, const BNode* bnodes, uint32_t* sbins, const Counters* ctr) {
  const uint32_t s = blockIdx.x, tid = threadIdx.x;
  if (s >= ctr->numSegs) return;
  const Seg* sg = segs + s; SegX* x = sx + s;
  __shared__ uint32_t s_try;
  if (tid == 0u) {
    uint32_t t = 0u;
    const uint32_t ext = x->extEnd - sg->end;
    if (ext > 0u && !(sg->flags & 1u)) {
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
  if (s_try) { uint32_t* b = sbins + (size_t)s * SBINS_WORDS; for (uint32_t w = tid; w < (uint32_t)SBINS_WORDS; w += blockDim.x) { const uint32_t k = w % SBINW; b[w] = k < 3 ? ENC_POS_INF : (k < 6 ? ENC_NEG_INF : 0u); } }
}

// ---- SpatialBinInfo::bin2 (heuristic_spatial.h:160-222) over the chunks of the sets that try a spatial split
__device__ __forceinline__ void sbin_extend(uint32_t* bins, int dim, int bin, const float* lo, const float* hi) {
  if (box_empty(lo, hi)) return;
  uint32_t* e = bins + (dim * SBINS + bin) * SBINW;
  for (int d = 0; d < 3; d++) { atomicMin(&e[d], enc(lo[d])); atomicMax(&e[3 + d], enc(hi[d])); }
}
// The chain of cuts bin2 makes for ONE reference with budget on ONE axis (:186-218): the reference is clipped bin by bin from its first to its last bin;
// l = the bin it is counted to begin in (numBegin), rr = the bin it is counted to end in (numEnd) -- a plane p then sees it on the left iff l < p and on the
// right iff rr >= p.  EXTEND: the pieces also extend the bins' boxes (binning); without it only (l, rr) come back, which is how spatial_partition decides
// the sides: by construction neither side can receive more references than spatial_best predicted from the counts.
template <bool EXTEND>
__device__ __forceinline__ void spatial_chain(const float (&v)[3][3], const PrimRef& r, int d, float ofs, float scale, float inv, uint32_t* bins, int& l, int& rr) {
  const float rlo = sel3((uint32_t)d, r.lo[0], r.lo[1], r.lo[2]), rhi = sel3((uint32_t)d, r.hi[0], r.hi[1], r.hi[2]);
  l = sbin(rlo, ofs, scale); rr = sbin(rhi, ofs, scale);
  if (l == rr) { if (EXTEND) sbin_extend(bins, d, l, r.lo, r.hi); return; }
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
__global__ __launch_bounds__(256) void spatial_bin(const Seg* segs, const SegX* sx, const Chunk* chunks, const PrimRef* src, const GeomDesc* geoms, uint32_t* sbins, const Counters* ctr) {
  __shared__ uint32_t s_b[SBINS_WORDS];
  const uint32_t tid = threadIdx.x;
  if (blockIdx.x >= ctr->numChunks) return;
  const Chunk ck = chunks[blockIdx.x];
  const SegX* x = sx + ck.seg;
  if (!x->trySpatial) return;
  for (uint32_t w = tid; w < (uint32_t)SBINS_WORDS; w += 256u) { const uint32_t k = w % SBINW; s_b[w] = k < 3 ? ENC_POS_INF : (k < 6 ? ENC_NEG_INF : 0u); }
  __syncthreads();
  float ofs[3], scale[3], inv[3];
  for (int d = 0; d < 3; d++) { ofs[d] = x->sofs[d]; scale[d] = x->sscale[d]; inv[d] = x->sinv[d]; }
  for (uint32_t i = ck.begin + tid; i < ck.end; i += 256u) {
    const PrimRef r = load_prim(src + i);
    const uint32_t budget = r.geom >> SPLIT_SHIFT;
    if (budget <= 1u) {                                                             // cannot be split: whole into the bin of its centre (:170-178)
      for (int d = 0; d < 3; d++) {
        const int b = sbin(0.5f * (r.lo[d] + r.hi[d]), ofs[d], scale[d]);
        sbin_extend(s_b, d, b, r.lo, r.hi);
        atomicAdd(&s_b[(d * SBINS + b) * SBINW + 6], 1u); atomicAdd(&s_b[(d * SBINS + b) * SBINW + 7], 1u);
      }
      continue;
    }
    float v[3][3]; load_tri_masked(geoms, r, v);
    for (int d = 0; d < 3; d++) {
      if (scale[d] == 0.0f) continue;                                              // mapping.invalid(dim)
      int l, rr;
      spatial_chain<true>(v, r, d, ofs[d], scale[d], inv[d], s_b, l, rr);
      atomicAdd(&s_b[(d * SBINS + l) * SBINW + 6], 1u); atomicAdd(&s_b[(d * SBINS + rr) * SBINW + 7], 1u);
    }
  }
  __syncthreads();
  uint32_t* g = sbins + (size_t)ck.seg * SBINS_WORDS;
  for (uint32_t w = tid; w < (uint32_t)SBINS_WORDS; w += 256u) {
    const uint32_t k = w % SBINW, v = s_b[w];
    if (k < 3) { if (v != ENC_POS_INF) atomicMin(&g[w], v); } else if (k < 6) { if (v != ENC_NEG_INF) atomicMax(&g[w], v); } else if (v) atomicAdd(&g[w], v);
  }
}

// ---- SpatialBinInfo::best (heuristic_spatial.h:285-358) + the decision of find() (:187-198); one wavefront per set, lane d = axis d
__global__ __launch_bounds__(64) void spatial_best(Seg* segs, SegX* sx, const uint32_t* sbins, BNode* bnodes, const Counters* ctr, Params prm) {
  __shared__ float s_sah[3]; __shared__ uint32_t s_pos[3], s_l[3], s_r[3];
  const uint32_t s = blockIdx.x, lane = threadIdx.x;
  if (s >= ctr->numSegs) return;
  Seg* sg = segs + s; SegX* x = sx + s;
  if (!x->trySpatial) return;
  const uint32_t* B = sbins + (size_t)s * SBINS_WORDS;
  const uint32_t add = (1u << prm.shift) - 1u;
  if (lane < 3u) {
    const uint32_t d = lane;
    float rA[SBINS]; uint32_t rC[SBINS];
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    uint32_t cnt = 0;
    for (int i = SBINS - 1; i > 0; i--) {
      const uint32_t* e = B + (d * SBINS + i) * SBINW;
      cnt += e[7]; rC[i] = cnt;
      for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], dec(e[k])); hi[k] = fmaxf(hi[k], dec(e[3 + k])); }
      rA[i] = (lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]) ? 0.0f : half_area3(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]);
    }
    for (int k = 0; k < 3; k++) { lo[k] = __builtin_inff(); hi[k] = -__builtin_inff(); }
    cnt = 0; float best = __builtin_inff(); uint32_t bpos = 0, bl = 0, br = 0;
    for (int i = 1; i < SBINS; i++) {
      const uint32_t* e = B + (d * SBINS + (i - 1)) * SBINW;
      cnt += e[6];
      for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], dec(e[k])); hi[k] = fmaxf(hi[k], dec(e[3 + k])); }
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

// ---- create_spatial_splits + the partition by centre (heuristic_spatial_array.h:266-330, :395-420) for the chunks of the sets that split spatially
__global__ __launch_bounds__(256) void spatial_partition(Seg* segs, const SegX* sx, const Chunk* chunks, const PrimRef* src, PrimRef* dst, const GeomDesc* geoms, Counters* ctr) {
  __shared__ uint32_t s_cnt[4][2], s_acc[2][12], s_baseL, s_baseR;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (blockIdx.x >= ctr->numChunks) return;
  const Chunk ck = chunks[blockIdx.x];
  Seg* sg = segs + ck.seg; const SegX* x = sx + ck.seg;
  if (!(sg->flags & 2u)) return;
  const uint32_t dim = sg->dim; const int pos = (int)sg->pos;
  const float ofs = sel3(dim, x->sofs[0], x->sofs[1], x->sofs[2]), scale = sel3(dim, x->sscale[0], x->sscale[1], x->sscale[2]), inv = sel3(dim, x->sinv[0], x->sinv[1], x->sinv[2]);
  const float fpos = sbin_pos(pos, ofs, inv);
  for (uint32_t round = 0; round < (uint32_t)CHUNK_ROUNDS; round++) {
    if (tid < 24u) s_acc[tid / 12u][tid % 12u] = (tid % 6u < 3u) ? ENC_POS_INF : ENC_NEG_INF;
    __syncthreads();
    const uint32_t i = ck.begin + round * 256u + tid;
    const bool v = i < ck.end;
    PrimRef L{}, R{}; bool toL = false, toR = false;
    if (v) {
      const PrimRef r = load_prim(src + i);
      const uint32_t budget = r.geom >> SPLIT_SHIFT;
      const float rlo = sel3(dim, r.lo[0], r.lo[1], r.lo[2]), rhi = sel3(dim, r.hi[0], r.hi[1], r.hi[2]);
      if (budget > 1u) {                                                              // the sides binning counted it on (spatial_chain)
        float tv[3][3]; load_tri_masked(geoms, r, tv);
        int l, rr; spatial_chain<false>(tv, r, (int)dim, ofs, scale, inv, nullptr, l, rr);
        toL = l < pos; toR = rr >= pos;
        if (toL && toR) {                                                             // straddles the plane: cut it, unless a piece would be empty
          float Llo[3], Lhi[3], Rlo[3], Rhi[3];
          split_triangle(tv, dim, fpos, r.lo, r.hi, Llo, Lhi, Rlo, Rhi);
          const bool eL = box_empty(Llo, Lhi), eR = box_empty(Rlo, Rhi);
          if (!eL && !eR) {
            L = r; R = r;
            for (int d = 0; d < 3; d++) { L.lo[d] = Llo[d]; L.hi[d] = Lhi[d]; R.lo[d] = Rlo[d]; R.hi[d] = Rhi[d]; }
            L.geom = (r.geom & GEOM_MASK) | ((budget - 1u) << SPLIT_SHIFT); R.geom = L.geom;
          } else if (eL) { toL = false; R = r; } else { toR = false; L = r; }
        } else if (toL) L = r; else R = r;
      } else {                                                                        // whole, to the side its centre lies on
        toL = sbin(0.5f * (rlo + rhi), ofs, scale) < pos; toR = !toL;
        if (toL) L = r; else R = r;
      }
      for (int side = 0; side < 2; side++) {
        if (!(side ? toR : toL)) continue;
        const PrimRef& q = side ? R : L;
        for (int d = 0; d < 3; d++) {
          const uint32_t cc = enc(q.lo[d] + q.hi[d]);
          atomicMin(&s_acc[side][d], cc); atomicMax(&s_acc[side][3 + d], cc);
          atomicMin(&s_acc[side][6 + d], enc(q.lo[d])); atomicMax(&s_acc[side][9 + d], enc(q.hi[d]));
        }
      }
    }
    const unsigned long long lm = __ballot(toL), rm = __ballot(toR);
    if (lane == 0u) { s_cnt[wave][0] = (uint32_t)__popcll(lm); s_cnt[wave][1] = (uint32_t)__popcll(rm); }
    __syncthreads();
    if (tid == 0u) {
      const uint32_t l = s_cnt[0][0] + s_cnt[1][0] + s_cnt[2][0] + s_cnt[3][0], rr = s_cnt[0][1] + s_cnt[1][1] + s_cnt[2][1] + s_cnt[3][1];
      s_baseL = l ? atomicAdd(&sg->curL, l) : 0u; s_baseR = rr ? atomicAdd(&sg->curR, rr) : 0u;
    }
    __syncthreads();
    uint32_t offL = s_baseL, offR = s_baseR;
    for (uint32_t w = 0; w < wave; w++) { offL += s_cnt[w][0]; offR += s_cnt[w][1]; }
    const unsigned long long lt = (1ull << lane) - 1ull;
    // (the limits cannot be reached -- the counts of spatial_best are upper bounds -- but a store past a set's capacity would corrupt a sibling: guarded)
    const uint32_t oL = offL + (uint32_t)__popcll(lm & lt), oR = offR + (uint32_t)__popcll(rm & lt);
    if (toL) { if (oL < sg->begin + x->capL) store_prim(dst + oL, L); else ctr->overflow = 2u; }
    if (toR) { if (oR < x->extEnd) store_prim(dst + oR, R); else ctr->overflow = 2u; }
    if (tid < 24u) {
      const uint32_t side = tid / 12u, k = tid % 12u, val = s_acc[side][k];
      if (k % 6u < 3u) { if (val != ENC_POS_INF) atomicMin(&sg->acc[side][k], val); } else { if (val != ENC_NEG_INF) atomicMax(&sg->acc[side][k], val); }
    }
    __syncthreads();
  }
}
