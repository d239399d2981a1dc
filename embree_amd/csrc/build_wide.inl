// build_wide.inl -- K4: collapse of the binary tree into 8-wide quantised nodes.
// Part of build.hip (included inside its anonymous namespace); see the header of build.hip for the pipeline.
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
  // A top phase that was enqueued with too few levels (one-round-trip commits: the margins, or the counts learned from another scene of this size) leaves sets
  // unsplit: binary "leaves" of thousands of references whose ids nobody wrote.  The commit is repeated anyway (the host sees numSegs / overflow); until then
  // nothing may walk that tree -- leaf ids from whatever the arena held sent tri_records to wild addresses (a rare abort in the test suite, found in round 4).
  const bool unfinished = ctr->numSegs != 0u || ctr->overflow != 0u;
  const bool any = ctr->numPrims != 0u && !unfinished;         // (a scene whose triangles are all invalid has no tree)
  ctr->rootRef = any ? 0u : MI355_EMPTY_REF; ctr->numWide = any ? 1u : 0u; ctr->wideCount[0] = any ? 1u : 0u; ctr->wideCount[1] = 0; ctr->wideDepth = 0; ctr->lvlStart[0] = 0; ctr->numTrisOut = 0;   // (leaf count and SAH sum: the stripes, zero since build_begin)
}

template <typename T> __device__ __forceinline__ T grp_get(T v, uint32_t lane, uint32_t idx) { return __shfl(v, (int)((lane & ~7u) | idx), 64); }
// Reductions over the 8 lanes of a group as three DPP butterflies -- quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror (lane i <-> i ^ 7: the other
// quad of the group, whose four lanes agree after the first two steps) -- instead of three ds_bpermute round trips through the LDS pipe (~100 cycles each,
// and the open loop of wide_plan is one dependent chain of them)
__device__ __forceinline__ float grp_xf(float v, int step) {
  const uint32_t u = __float_as_uint(v);
  return __uint_as_float(step == 0 ? dpp_u<0xB1, 0xF>(u, u) : (step == 1 ? dpp_u<0x4E, 0xF>(u, u) : dpp_u<0x141, 0xF>(u, u)));
}
__device__ __forceinline__ uint32_t grp_xu(uint32_t u, int step) { return step == 0 ? dpp_u<0xB1, 0xF>(u, u) : (step == 1 ? dpp_u<0x4E, 0xF>(u, u) : dpp_u<0x141, 0xF>(u, u)); }
__device__ __forceinline__ float grp_min(float v) { v = fminf(v, grp_xf(v, 0)); v = fminf(v, grp_xf(v, 1)); return fminf(v, grp_xf(v, 2)); }
__device__ __forceinline__ float grp_max(float v) { v = fmaxf(v, grp_xf(v, 0)); v = fmaxf(v, grp_xf(v, 1)); return fmaxf(v, grp_xf(v, 2)); }
__device__ __forceinline__ float grp_sum(float v) { v += grp_xf(v, 0); v += grp_xf(v, 1); return v + grp_xf(v, 2); }
// argmax over the 8 lanes of a group; ties go to the lower index (the serial formulation keeps the first maximum)
__device__ __forceinline__ void grp_argmax(float& v, uint32_t& idx) {
#pragma unroll
  for (int st = 0; st < 3; st++) {
    const float ov = grp_xf(v, st); const uint32_t oi = grp_xu(idx, st);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
}
__device__ __forceinline__ BNode load_bnode(const BNode* p) {
  const float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1]; const uint4 c = ((const uint4*)p)[2];
  BNode r; r.lo[0] = a.x; r.lo[1] = a.y; r.lo[2] = a.z; r.begin = __float_as_uint(a.w); r.hi[0] = b.x; r.hi[1] = b.y; r.hi[2] = b.z; r.end = __float_as_uint(b.w);
  r.left = c.x; r.right = c.y; r.splitSah = __uint_as_float(c.z); r.pad = 0; return r;
}

// items per workgroup of wide_plan / wide_emit: a multiple of eight (a wavefront takes eight items at a time), the same for both kernels of a level
__device__ __forceinline__ uint32_t wide_span(uint32_t numItems, uint32_t blocks) { return ((numItems + blocks * 8u - 1u) / (blocks * 8u)) * 8u; }

__global__ __launch_bounds__(64) void wide_plan(const WideItem* items, const BNode* bnodes, WidePlan* plans, uint2* itemCnt, uint2* groupSum,
                                                Counters* ctr, Params prm, uint32_t parity) {
  const uint32_t numItems = ctr->wideCount[parity];
  const float rootArea = ctr->rootArea;
  const uint32_t lane = threadIdx.x, c = lane & 7u, g = lane >> 3;
  unsigned long long sahAcc = 0ull; uint32_t leafAcc = 0u;      // per-wave partial sums: one atomic per wave at the end (a same-address atomic costs ~2 ns)
  // (round 6) A workgroup owns a CONTIGUOUS run of `per` items (wide_span): it carries the running counts of its run itself and hands wide_scan ONE pair -- the scan is over
  // <= 8192 workgroups instead of one pair per eight items (37 k for the crown's widest level: 56 us in one workgroup).  Same sums in the same item order: same numbering.
  const uint32_t per = wide_span(numItems, gridDim.x), first = blockIdx.x * per, end = min(first + per, numItems);
  uint32_t runI = 0u, runT = 0u;                                // (wave-uniform) inner children / leaf triangles of this workgroup's items so far
  for (uint32_t base = first; base < end; base += 8u) {
    const uint32_t t = base + g; const bool valid = t < end;
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
    nTri += grp_xu(nTri, 0); nTri += grp_xu(nTri, 1); nTri += grp_xu(nTri, 2);
    const uint32_t nInner = (uint32_t)__popc(imask);
    if (valid) {
      plans[t].ch[c] = sHas ? sCh : NIL;
      if (c == 0u) { plans[t].imask = imask; plans[t].leafMask = leafMask; plans[t].nch = nch; plans[t].pad = 0u; }
    }
    // ---- counts: exclusive prefix over the 8 items of this wave, wave total for the scan
    const uint32_t ci = valid ? nInner : 0u, ct = valid ? nTri : 0u;          // every lane of a group holds the same pair
    uint32_t xi = ci, xt = ct;
    for (int o = 8; o < 64; o <<= 1) { const uint32_t ui = (uint32_t)__shfl_up((int)xi, o, 64), ut = (uint32_t)__shfl_up((int)xt, o, 64); if (lane >= (uint32_t)o) { xi += ui; xt += ut; } }
    if (valid && c == 0u) itemCnt[t] = make_uint2(runI + xi - ci, runT + xt - ct);
    runI += (uint32_t)__builtin_amdgcn_readlane((int)xi, 63); runT += (uint32_t)__builtin_amdgcn_readlane((int)xt, 63);
    leafAcc += (uint32_t)__popcll(__ballot(valid && sHas && sLeaf));
  }
  if (lane == 0u) groupSum[blockIdx.x] = make_uint2(runI, runT);   // (every workgroup of the grid: wide_scan reads all of them)
  for (int o = 8; o < 64; o <<= 1) sahAcc += (unsigned long long)__shfl_xor((long long)sahAcc, o, 64);   // lanes with c == 0 hold the partial sums
  if (lane == 0u) { Counters::Stripe* sp = &ctr->stripe[blockIdx.x % Counters::STRIPES]; if (sahAcc) atomicAdd(&sp->sahFixed, sahAcc); if (leafAcc) atomicAdd(&sp->numLeaves, leafAcc); }   // (a stripe per 128 workgroups: see Counters)
}

// one block: exclusive scan of the workgroups' totals in workgroup (= item) order; publishes the level's bases and the next level's item count
__global__ __launch_bounds__(1024) void wide_scan(uint2* groupSum, Counters* ctr, uint32_t parity, uint32_t maxNodes, uint32_t numGroups) {
  __shared__ uint2 s_part[17];
  const uint32_t numItems = ctr->wideCount[parity];
  // (the grid of wide_plan / wide_emit is an upper bound: only the workgroups whose run holds an item count -- and only their sums are read back by wide_emit; a level of a
  // few hundred items scanned all 8192 entries: 8.3 us where 4.5 do)
  { const uint32_t span = wide_span(numItems, numGroups), used = span ? (numItems + span - 1u) / span : 0u; numGroups = used < numGroups ? used : numGroups; }
  // every thread owns a run of `per` consecutive workgroups (a multiple of 8: four 16-byte loads in flight per step)
  const uint32_t tid = threadIdx.x, per = ((numGroups + 1023u) / 1024u + 7u) & ~7u, b = min(tid * per, numGroups), e = min(b + per, numGroups);
  uint2 sum = make_uint2(0, 0);
  for (uint32_t i = b; i < e; i += 8u) {
    uint4 x[4];
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) x[k] = i + 2u * k < e ? ((const uint4*)(groupSum + i))[k] : make_uint4(0, 0, 0, 0);   // (an odd last group: the array has 16 spare entries; what lies there is not counted)
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) { sum.x += x[k].x; sum.y += x[k].y; if (i + 2u * k + 1u < e) { sum.x += x[k].z; sum.y += x[k].w; } }
  }
  uint2 total;
  uint2 run = block_exclusive_scan_1024_u2(sum, s_part, tid, total);   // (DPP wave scans + 16 wave totals: three barriers, build_common.inl)
  for (uint32_t i = b; i < e; i += 8u) {
    uint4 x[4];
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) x[k] = i + 2u * k < e ? ((const uint4*)(groupSum + i))[k] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
      if (i + 2u * k < e) { groupSum[i + 2u * k] = run; run.x += x[k].x; run.y += x[k].y; }
      if (i + 2u * k + 1u < e) { groupSum[i + 2u * k + 1u] = run; run.x += x[k].z; run.y += x[k].w; }
    }
  }
  __syncthreads();
  if (tid == 0) {
    ctr->lvlNodeBase = ctr->numWide; ctr->lvlTriBase = ctr->numTrisOut;
    if (numItems) { ctr->wideDepth++; if (ctr->wideDepth < 64u) ctr->lvlStart[ctr->wideDepth] = ctr->numWide; }
    if ((uint64_t)ctr->numWide + total.x > maxNodes) { atomicMax(&ctr->overflow, 2u); ctr->wideCount[parity ^ 1u] = 0u; }
    else { ctr->numWide += total.x; ctr->numTrisOut += total.y; ctr->wideCount[parity ^ 1u] = total.x; }
  }
}

// Quantises the child boxes of 8 nodes at once (lane = 8 * node + slot): plane = org + q * 2^(e-127), lower planes rounded down, upper planes
// rounded up, verified EXACTLY: org, q and the power-of-two scale span fewer than 53 bits, so org + q * scale is exact in fp64.  (An fp32 check
// accepts fl(org + q * scale) == lo while the exact plane lies up to half an ulp of lo INSIDE the child box -- 0.004 at coordinates of 1e5 -- and the
// fast traversal path evaluates the exact plane, q * (scale * rdir) + (org - ray.org) * rdir; the robust path's fmaf(q, scale, org) rounds a plane
// that is <= lo to a value <= lo, so it stays conservative too.)  Shared by wide_emit and refit_level so that a refitted node is what a build would
// have written for the same boxes.
__device__ __forceinline__ double plane_exact(float q, float sc, float org) { return (double)org + (double)q * (double)sc; }
__device__ __forceinline__ void quantise_slots(bool has, uint32_t lane, const float (&lo)[3], const float (&hi)[3], const float (&olo)[3], const float (&ohi)[3],
                                               uint32_t (&ex)[3], uint32_t (&qa)[3], uint32_t (&qb)[3]) {
  // ---- quantise: plane = org + q * 2^(e-127), lower rounded down, upper rounded up
  for (int d = 0; d < 3; d++) {
    const float ext = ohi[d] - olo[d];
    int e = 1;                                                // biased exponent, scale = 2^(e-127)
    if (ext > 0.0f) { int fe; frexpf(ext / 255.0f, &fe); e = fe + 127; if (e < 1) e = 1; if (e > 254) e = 254; }
    for (;;) {                                                // grow the scale until every upper plane of the node fits in 8 bits
      const float sc = __uint_as_float((uint32_t)e << 23);
      bool fits = true;
      if (has) { float q = ceilf((hi[d] - olo[d]) / sc); while (plane_exact(q, sc, olo[d]) < (double)hi[d]) q += 1.0f; fits = q <= 255.0f; }
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
      while (a > 0.0f && plane_exact(a, sc, olo[d]) > (double)lo[d]) a -= 1.0f;
      float b = ceilf((hi[d] - olo[d]) / sc); if (b < 0.0f) b = 0.0f;
      while (b < 255.0f && plane_exact(b, sc, olo[d]) < (double)hi[d]) b += 1.0f;
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
  const uint32_t per = wide_span(numItems, gridDim.x), first = blockIdx.x * per, end = min(first + per, numItems);   // the run wide_plan gave this workgroup
  const uint2 mine = groupSum[blockIdx.x];                     // what the workgroups before this one counted (wide_scan)
  for (uint32_t base = first; base < end; base += 8u) {
    const uint32_t t = base + g; const bool valid = t < end;
    uint32_t ch = NIL, imask = 0, leafMask = 0, node = 0; uint2 ofs = make_uint2(0, 0);
    if (valid) {
      ch = plans[t].ch[s]; imask = plans[t].imask; leafMask = plans[t].leafMask; node = items[t].node;
      const uint2 b = itemCnt[t]; ofs = make_uint2(mine.x + b.x, mine.y + b.y);
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
