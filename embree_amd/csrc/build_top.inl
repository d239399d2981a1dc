// build_top.inl -- K2: level-synchronous top phase.
// Part of build.hip (included inside its anonymous namespace); see the header of build.hip for the pipeline.
// ------------------------------------------------------------------------------------ K2 top phase
constexpr uint32_t ACC_SETS = 16u, ACC_REPL = 8u, ACC_STRIDE = 32u;   // levels of at most ACC_SETS sets: ACC_REPL copies of a set's bins (top_bin) and of its children's bounds record (acc_copy; ACC_STRIDE words per copy, 24 used: 128 bytes)
// localMax: sets of at most this many references are top_local's at this level (0: none are), the four kernels of the chunked path leave them alone.
// A workgroup sets up SETUP_SEGS sets: a lane per set takes its bin mapping and the number of its chunks, ONE atomic per workgroup reserves the chunks of all of them
// (was: a workgroup and a returning atomic per set -- at the levels of a HIGH commit where every one of ~10,000 sets takes the chunked path those atomics, ~10 ns each on
// one word, were the kernel: 100 us a level, 300 us of a 10.5 ms commit), then the 256 threads write the chunk list.
constexpr uint32_t SETUP_SEGS = 64u;
__global__ __launch_bounds__(256) void top_setup(Seg* segs, uint32_t* bins, Chunk* chunks, Counters* ctr, uint32_t localMax, uint32_t level, uint32_t* chunkFlag, uint32_t* binsTop, uint32_t listBlocks) {
  __shared__ uint32_t s_begin[SETUP_SEGS], s_end[SETUP_SEGS], s_c0[SETUP_SEGS], s_pre[SETUP_SEGS + 1u];
  const uint32_t tid = threadIdx.x, numSegs = ctr->numSegs;
  // the grid: listBlocks workgroups that reserve and write the chunk lists of SETUP_SEGS sets each, then one workgroup PER SET that clears the bins of a set of several
  // chunks (the copies the chunks merge into, at the upper levels: 8 x 2.7 KB per set -- all of a level's clears in 36 workgroups took 40 us where 2325 take 5)
  if (blockIdx.x >= listBlocks) {
    const uint32_t s = blockIdx.x - listBlocks;
    if (s >= numSegs) return;
    const uint32_t n = segs[s].end - segs[s].begin;
    if (n <= localMax || n <= CHUNK) return;                     // (a set of one chunk: top_bin writes its bins as they are -- clearing them was 54 MB per level of a HIGH commit)
    if (binsTop && numSegs <= ACC_SETS) { for (uint32_t r = 0; r < ACC_REPL; r++) bins_clear(binsTop + ((size_t)s * ACC_REPL + r) * BINS_WORDS, tid, 256u); }   // the copies the chunks merge into (top_bin); top_split folds them
    else bins_clear(bins + (size_t)s * BINS_WORDS, tid, 256u);
    return;
  }
  const uint32_t s0 = blockIdx.x * SETUP_SEGS;
  if (s0 >= numSegs) return;                                    // the grid is an upper bound (2^level segments at most)
  if (tid < SETUP_SEGS) {                                       // (wave 0, all of its lanes: the scan below needs them)
    const uint32_t s = s0 + tid;
    uint32_t begin = 0u, end = 0u, nch = 0u;
    if (s < numSegs) {
      Seg* sg = segs + s;
      begin = sg->begin; end = sg->end;
      const uint32_t n = end - begin;
      if (n > localMax) {
        nch = (n + CHUNK - 1u) / CHUNK;
        const Mapping m = make_mapping(n, sg->cmin, sg->cmax);
        for (int d = 0; d < 3; d++) { sg->ofs[d] = m.ofs[d]; sg->scale[d] = m.scale[d]; }
        sg->nb = m.nb;
      }
    }
    const uint32_t incl = wave_incl_scan_u32(nch), total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t base = 0u;
    if (tid == 0u && total != 0u) { base = atomicAdd(&ctr->numChunks, total); ctr->chunkedLevels = level + 1u; }   // (every writer of a level writes the same value; levels are launches, in order)
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    const uint32_t c0 = base + incl - nch;
    if (nch != 0u) segs[s].chunk0 = c0;
    s_begin[tid] = begin; s_end[tid] = end; s_c0[tid] = c0; s_pre[tid] = incl - nch;
    if (tid == SETUP_SEGS - 1u) s_pre[SETUP_SEGS] = incl;
  }
  __syncthreads();
  const uint32_t total = s_pre[SETUP_SEGS];
  for (uint32_t j = tid; j < total; j += 256u) {                  // chunk j of this workgroup's sets: the last set whose first chunk is <= j
    uint32_t l = 0u;
#pragma unroll
    for (uint32_t step = SETUP_SEGS / 2u; step != 0u; step >>= 1) if (s_pre[l + step] <= j) l += step;
    const uint32_t c = j - s_pre[l];
    Chunk ck; ck.seg = s0 + l; ck.begin = s_begin[l] + c * CHUNK; ck.end = min(ck.begin + CHUNK, s_end[l]);
    chunks[s_c0[l] + c] = ck;
    if (chunkFlag) chunkFlag[s_c0[l] + c] = 0u;                   // spatial-split builds: "this chunk has not said yet how many references it sends to either side" (spatial_partition)
  }
}

// chunkCnt: per chunk, how many of its references fell into every bin and the bins before it (3 x NBINS words): top_split turns them into the chunk's places in the two children,
// so that where a reference lands does not depend on which chunk reached a cursor first (top_partition)
__global__ __launch_bounds__(256) void top_bin(const Seg* segs, const Chunk* chunks, const PrimRef* src, uint32_t* bins, const Counters* ctr, uint32_t* chunkCnt, uint32_t chunkStride, uint32_t* binsTop) {
  __shared__ uint32_t s_bins[BIN_COPIES * COPY_STRIDE];
  const uint32_t tid = threadIdx.x;
  if (blockIdx.x >= ctr->numChunks) return;
  const Chunk ck = chunks[blockIdx.x];
  const Seg* sg = segs + ck.seg;
  Mapping m; for (int d = 0; d < 3; d++) { m.ofs[d] = sg->ofs[d]; m.scale[d] = sg->scale[d]; } m.nb = sg->nb;
  bins_clear_copies(s_bins, tid, 256u);
  __syncthreads();
  {
    const uint32_t lane = tid & 63u;
    for (uint32_t i0 = ck.begin; i0 < ck.end; i0 += 256u) {        // (workgroup-uniform trip count: every lane calls bins_add_copies, which exchanges registers inside pairs of lanes)
      const uint32_t i = i0 + tid; const bool v = i < ck.end;
      PrimRef r{}; if (v) r = load_prim(src + i);
      bins_add_copies(s_bins, m, r, v, lane);
    }
  }
  __syncthreads();
  bins_fold_copies(s_bins, tid, 256u);
  __syncthreads();
  if (tid < 3u * (uint32_t)NBINS) {                              // (running sums along every axis: top_split reads ONE word per chunk, "left of the plane")
    const uint32_t b = tid % (uint32_t)NBINS; uint32_t sum = 0u;
    for (uint32_t k = 0; k <= b; k++) sum += s_bins[(tid - b + k) * BINW + 6];
    chunkCnt[(size_t)tid * chunkStride + blockIdx.x] = sum;     // [axis][bin][chunk]: what top_split reads of a set's chunks -- one (axis, bin) -- lies together
  }
  uint32_t* g = bins + (size_t)ck.seg * BINS_WORDS;
  if (ck.begin == sg->begin && ck.end == sg->end) {             // the set's only chunk (every set of the lower levels): its bins ARE the set's bins, no atomics
    for (uint32_t w = tid; w < (uint32_t)BINS_WORDS; w += 256u) g[w] = s_bins[w];
    return;
  }
  // (round 6) at the upper levels -- while a level has at most ACC_SETS sets -- a chunk merges into one of ACC_REPL copies of its set's bins: 2325 chunks x 672 atomics on
  // the 42 cache lines of ONE set were what the root's level of top_bin ended on (65 us where the levels of many sets take 40)
  if (binsTop && ctr->numSegs <= ACC_SETS) g = binsTop + ((size_t)ck.seg * ACC_REPL + (blockIdx.x & (ACC_REPL - 1u))) * BINS_WORDS;
  for (uint32_t w = tid; w < (uint32_t)BINS_WORDS; w += 256u) {      // BinInfoT::merge :312-321
    const uint32_t k = w % BINW, cnt = s_bins[w - k + 6];
    if (cnt == 0u) continue;
    if (k < 3) atomicMin(&g[w], s_bins[w]); else if (k < 6) atomicMax(&g[w], s_bins[w]); else atomicAdd(&g[w], s_bins[w]);
  }
}

// (round 6) The children's bounds of a set are min / max atomics of every chunk on ONE record: at the upper levels -- the root's level has 2325 chunks -- 12 to 24 atomics
// per chunk on the same cache line are what the partition kernels end on (top_partition 90 us at the root's level, 57 where the sets are many).  While a level has at most
// ACC_SETS sets, a chunk works on one of ACC_REPL copies of its set's record, each on a line of its own (accTop); top_split clears them, top_emit folds them.  min / max: the
// bounds do not change by a bit.
__device__ __forceinline__ uint32_t* acc_copy(uint32_t* accTop, uint32_t numSegs, Seg* sg, uint32_t seg, uint32_t chunk) {
  return (accTop && numSegs <= ACC_SETS) ? accTop + ((size_t)seg * ACC_REPL + (chunk & (ACC_REPL - 1u))) * ACC_STRIDE : &sg->acc[0][0];
}
// the twelve words of one side of a set's record, folded over the copies (all copies are asked for before the first is used: one round trip)
__device__ __forceinline__ void acc_folded(const uint32_t* accTop, uint32_t numSegs, const Seg* sg, uint32_t seg, uint32_t side, uint32_t (&v)[12]) {
  for (uint32_t k = 0; k < 12u; k++) v[k] = sg->acc[side][k];
  if (accTop && numSegs <= ACC_SETS) {
    uint32_t x[ACC_REPL][12];
#pragma unroll
    for (uint32_t r = 0; r < ACC_REPL; r++)
#pragma unroll
      for (uint32_t k = 0; k < 12u; k++) x[r][k] = accTop[((size_t)seg * ACC_REPL + r) * ACC_STRIDE + side * 12u + k];
#pragma unroll
    for (uint32_t r = 0; r < ACC_REPL; r++)
#pragma unroll
      for (uint32_t k = 0; k < 12u; k++) v[k] = (k % 6u) < 3u ? min(v[k], x[r][k]) : max(v[k], x[r][k]);
  }
}
struct SegX;                                                    // build_spatial.inl: extended ranges of spatial-split builds (nullptr otherwise)
__device__ __forceinline__ uint32_t segx_ext_end(const SegX* sx, uint32_t s);
__device__ __forceinline__ void segx_object_split(SegX* sx, uint32_t s, uint32_t capL, float sah);
__device__ __forceinline__ uint32_t segx_cap_left(const SegX* sx, uint32_t s);
__device__ __forceinline__ void segx_child(SegX* nx, uint32_t k, uint32_t extEnd);
__global__ __launch_bounds__(64) void top_split(Seg* segs, const uint32_t* bins, BNode* bnodes, Counters* ctr, Params prm, uint32_t forceFallback, SegX* sx,
                                                const uint32_t* chunkCnt, uint2* chunkBase, uint32_t localMax, uint32_t chunkStride, uint32_t* accTop, const uint32_t* binsTop) {
  __shared__ SplitResult s_res;
  __shared__ uint32_t s_plan[4];                                // fallback, dim, pos, capacity of the left child
  const uint32_t s = blockIdx.x, lane = threadIdx.x;
  if (s >= ctr->numSegs) return;
  Seg* sg = segs + s;
  if (sg->end - sg->begin <= localMax) return;
  Mapping m; for (int d = 0; d < 3; d++) { m.ofs[d] = sg->ofs[d]; m.scale[d] = sg->scale[d]; } m.nb = sg->nb;
  __shared__ uint32_t s_fold[BINS_WORDS];
  const bool folded = binsTop && ctr->numSegs <= ACC_SETS && sg->end - sg->begin > CHUNK;   // (block-uniform) the set's bins = the fold of their copies (top_bin)
  if (folded) {
    constexpr uint32_t PER = ((uint32_t)BINS_WORDS + 63u) / 64u;    // words per lane: all their copies are asked for before the first one is used (one round trip, not eleven)
    uint32_t x[PER][ACC_REPL];
#pragma unroll
    for (uint32_t i = 0; i < PER; i++) {
      const uint32_t w = i * 64u + lane;
#pragma unroll
      for (uint32_t r = 0; r < ACC_REPL; r++) x[i][r] = w < (uint32_t)BINS_WORDS ? binsTop[((size_t)s * ACC_REPL + r) * BINS_WORDS + w] : 0u;
    }
#pragma unroll
    for (uint32_t i = 0; i < PER; i++) {
      const uint32_t w = i * 64u + lane, k = w % BINW;
      uint32_t v = x[i][0];
#pragma unroll
      for (uint32_t r = 1; r < ACC_REPL; r++) v = k < 3u ? min(v, x[i][r]) : (k < 6u ? max(v, x[i][r]) : v + x[i][r]);
      if (w < (uint32_t)BINS_WORDS) s_fold[w] = v;
    }
    __syncthreads();
  }
  sah_best_wave(folded ? (const uint32_t*)s_fold : bins + (size_t)s * BINS_WORDS, m, prm.shift, &s_res, lane);
  __syncthreads();
  if (lane == 0) {
    const uint32_t begin = sg->begin, end = sg->end, n = end - begin;
    SplitResult r = s_res;
    const bool fallback = (r.dim < 0) || forceFallback;        // split invalid -> median split (split_template :144-147)
    const uint32_t nL = fallback ? ((begin + end) / 2u - begin) : r.nL;
    // spatial-split builds: the set's extended range [end, extEnd) is shared between the children by their size, the right child starts behind the left
    // child's CAPACITY and binary nodes are numbered by capacity as well (setExtentedRanges / moveExtentedRange, heuristic_spatial_array.h:115-170)
    uint32_t capL = nL;
    if (sx) { const uint32_t ext = segx_ext_end(sx, s) - end; capL = nL + (uint32_t)floorf((float)nL / (float)n * (float)ext); segx_object_split(sx, s, capL, r.sah); }
    const uint32_t idL = sg->bnode + 1u, idR = sg->bnode + 2u * capL;   // implicit pre-order numbering (see K3)
    BNode* par = bnodes + sg->bnode;
    par->left = idL; par->right = idR; par->splitSah = r.sah;
    BNode L{}, R{};
    L.begin = begin; L.end = begin + nL; R.begin = begin + capL; R.end = begin + capL + (n - nL);
    L.left = L.right = R.left = R.right = NIL; L.splitSah = R.splitSah = __builtin_inff();
    for (int d = 0; d < 3; d++) { L.lo[d] = r.llo[d]; L.hi[d] = r.lhi[d]; R.lo[d] = r.rlo[d]; R.hi[d] = r.rhi[d]; }
    bnodes[idL] = L; bnodes[idR] = R;
    sg->flags = fallback ? 1u : 0u; sg->dim = fallback ? 0u : (uint32_t)r.dim; sg->pos = fallback ? capL : (uint32_t)r.pos; sg->nL = nL;   // (median split: pos = where the right child starts, see top_partition)
    // where the children END (top_emit of spatial-split builds reads it): an object split creates no reference, so the ends are known here.  (They were counted: two atomics
    // per chunk on the set's line -- the line every chunk of the set reads its plane from; a set that splits spatially after all is counted by spatial_partition's last chunk.)
    sg->childL = idL; sg->childR = idR; sg->curL = begin + nL; sg->curR = begin + capL + (n - nL);
    for (int side = 0; side < 2; side++) for (int k = 0; k < 12; k++) sg->acc[side][k] = (k % 6 < 3) ? ENC_POS_INF : ENC_NEG_INF;
    s_plan[0] = fallback ? 1u : 0u; s_plan[1] = fallback ? 0u : (uint32_t)r.dim; s_plan[2] = (uint32_t)r.pos; s_plan[3] = capL;
  }
  if (accTop && ctr->numSegs <= ACC_SETS)                        // the copies of this set's bounds record (acc_copy)
    for (uint32_t w = lane; w < ACC_REPL * ACC_STRIDE; w += 64u) { const uint32_t k = (w % ACC_STRIDE) % 12u; accTop[(size_t)s * ACC_REPL * ACC_STRIDE + w] = (k % 6u < 3u) ? ENC_POS_INF : ENC_NEG_INF; }
  __syncthreads();
  // Where every chunk of the set writes its left and its right references: an exclusive scan, in chunk order, of the per-chunk bin counts of top_bin.
  // (A cursor advanced with atomics hands the places out in the order the chunks ARRIVE: the sets come out the same, their order does not -- and a later
  // median split of coincident centroids cuts by order.)
  if (s_plan[0] == 0u) {
    const uint32_t begin = sg->begin, n = sg->end - begin, nch = (n + CHUNK - 1u) / CHUNK, c0 = sg->chunk0, dim = s_plan[1], pos = s_plan[2], capL = s_plan[3];
    uint32_t carry = 0u;
    for (uint32_t cb = 0; cb < nch; cb += 512u) {                // wave-uniform trip count; eight consecutive chunks per lane: their loads are in flight together
      uint32_t v[8], tot = 0u;
#pragma unroll
      for (uint32_t k = 0; k < 8u; k++) { const uint32_t c = cb + lane * 8u + k; v[k] = (c < nch && pos) ? chunkCnt[(size_t)(dim * NBINS + pos - 1u) * chunkStride + c0 + c] : 0u; }
#pragma unroll
      for (uint32_t k = 0; k < 8u; k++) tot += v[k];
      uint32_t incl = tot;
      for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= (uint32_t)o) incl += u; }
      uint32_t exclL = carry + incl - tot;
#pragma unroll
      for (uint32_t k = 0; k < 8u; k++) {
        const uint32_t c = cb + lane * 8u + k;
        if (c < nch) chunkBase[c0 + c] = make_uint2(begin + exclL, begin + capL + (c * CHUNK - exclL));
        exclL += v[k];
      }
      carry += (uint32_t)__shfl((int)incl, 63, 64);
    }
  }
}

__global__ __launch_bounds__(256) void top_partition(Seg* segs, const Chunk* chunks, const PrimRef* src, PrimRef* dst, const Counters* ctr, const uint2* chunkBase, uint32_t* accTop) {
  __shared__ uint32_t s_cnt[CHUNK_ROUNDS][4][2], s_off[CHUNK_ROUNDS][4][2], s_acc[2][12], s_baseL, s_baseR;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (blockIdx.x >= ctr->numChunks) return;
  const Chunk ck = chunks[blockIdx.x];
  Seg* sg = segs + ck.seg;
  if (sg->flags & 2u) return;                                    // this set splits spatially: spatial_partition (build_spatial.inl) moves it
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
    const uint2 base = chunkBase[blockIdx.x];                    // this chunk's places, in chunk order (top_split)
    s_baseL = base.x; s_baseR = base.y;
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < CHUNK_ROUNDS; r++) {
    if (!(validBits & (1u << r))) continue;
    const bool left = (sideBits >> r) & 1u;
    uint32_t o = left ? s_baseL + s_off[r][wave][0] + (uint32_t)__popcll(lm[r] & lt)
                      : s_baseR + s_off[r][wave][1] + (uint32_t)__popcll(rm[r] & lt);
    // a median split keeps the order (left = the first half as it lies, right = the rest): its positions do not depend on which chunk got to the
    // cursors first -- what the NEXT median split of the same triangles cuts off must be the same on every run (scenes of coincident triangles)
    if (fallback) { const uint32_t i = ck.begin + (uint32_t)r * 256u + tid; o = left ? i : sg->begin + pos + (i - mid); }
    store_prim(dst + o, pr[r]);
  }
  if (tid < 24) {
    const uint32_t side = tid / 12, k = tid % 12, v = s_acc[side][k];
    uint32_t* const a = acc_copy(accTop, ctr->numSegs, sg, ck.seg, blockIdx.x) + side * 12u + k;
    if (k % 6 < 3) { if (v != ENC_POS_INF) atomicMin(a, v); } else { if (v != ENC_NEG_INF) atomicMax(a, v); }
  }
}

// the end of a level: the last workgroup to get here makes the next level's work list the current one (was a launch of its own: 21 x 4.8 us per commit).
// The counters the others advanced are read with an atomic: a plain load may be served from this XCD's L2, which does not see the other XCDs' atomics.
__device__ __forceinline__ void top_level_end(Counters* ctr, uint32_t maxNext) {
  __syncthreads();
  if (threadIdx.x == 0u) {
    __threadfence();
    if (atomicAdd(&ctr->emitBlocks, 1u) == gridDim.x - 1u) {
      const uint32_t nextSegs = atomicAdd(&ctr->numSegsNext, 0u);
      if (ctr->numSegs) ctr->topLevels++;
      if (nextSegs > maxNext) atomicMax(&ctr->overflow, 1u);
      ctr->numSegs = nextSegs < maxNext ? nextSegs : maxNext; ctr->numSegsNext = 0; ctr->numChunks = 0; ctr->emitBlocks = 0;
    }
  }
}
__device__ __forceinline__ void top_emit_child(uint32_t b, uint32_t e, uint32_t child, const float* cmin, const float* cmax, Seg* next, SmallEntry* small, Counters* ctr,
                                               const Params& prm, uint32_t dstBuf, uint32_t maxNext, uint32_t maxSmall, SegX* nx, uint32_t childExtEnd) {
  if (e - b <= prm.small) {
    const uint32_t k = atomicAdd(&ctr->numSmall, 1u);
    if (k >= maxSmall) { atomicMax(&ctr->overflow, 1u); return; }
    SmallEntry se; se.begin = b; se.end = e; se.bnode = child; se.buf = dstBuf;
    for (int d = 0; d < 3; d++) { se.cmin[d] = cmin[d]; se.cmax[d] = cmax[d]; }
    small[k] = se;
    ((uint32_t*)(small + maxSmall))[k] = e - b;                   // the sizes again, side by side behind the list: what small_order sorts by (one coalesced read instead of 40-byte strides)
  } else {
    const uint32_t k = atomicAdd(&ctr->numSegsNext, 1u);
    if (k >= maxNext) { atomicMax(&ctr->overflow, 1u); return; }
    Seg ns{}; ns.begin = b; ns.end = e; ns.bnode = child;
    for (int d = 0; d < 3; d++) { ns.cmin[d] = cmin[d]; ns.cmax[d] = cmax[d]; }
    next[k] = ns;
    if (nx) segx_child(nx, k, childExtEnd);
  }
}

// A set of at most CHUNK references is ONE workgroup's: it holds them in registers, bins them into LDS, one of its waves prices the planes, and it writes
// them out partitioned -- one read and one write per level and one launch, where the chunked path below it needs two reads, the bins and the chunk counts
// through memory, and five launches (the crown stand-in's last eight levels: 740 us -> see profiles/r04_commit_timeline_medium.txt).  The partition is
// the same stable one (left and right keep their order), the bins are min / max / count: the tree does not depend on which path a set takes.
// lastOfLevel: the chunked path was not enqueued at this level (the last commit of this size had no large set here): a large set now means the commit is repeated.
// (The work lists are moved on by top_emit either way: every workgroup of this kernel arriving at a counter behind a fence cost 100 us per level.)
#ifdef MI355_LOCAL_WAVES
__attribute__((amdgpu_waves_per_eu(MI355_LOCAL_WAVES, MI355_LOCAL_WAVES)))
#endif
__global__ __launch_bounds__(256) void top_local(const Seg* segs, const PrimRef* src, PrimRef* dst, BNode* bnodes, Seg* next, SmallEntry* small, Counters* ctr,
                                                 Params prm, uint32_t dstBuf, uint32_t maxNext, uint32_t maxSmall, uint32_t forceFallback, uint32_t level, uint32_t lastOfLevel) {
  __shared__ uint32_t s_bins[BIN_COPIES * COPY_STRIDE];
  __shared__ SplitResult s_res;
  __shared__ uint32_t s_cnt[CHUNK_ROUNDS * 4][2], s_off[CHUNK_ROUNDS * 4][2], s_acc[2][12];
  const uint32_t s = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const Seg* sg = segs + s;
  const uint32_t numSegs = ctr->numSegs;
  const uint32_t begin = s < numSegs ? sg->begin : 0u, end = s < numSegs ? sg->end : 0u, n = end - begin;
  if (s < numSegs && n <= CHUNK) {                               // (workgroup-uniform)
    if (tid == 0u && level < __hip_atomic_load(&ctr->localFirst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&ctr->localFirst, level);   // (thousands of workgroups per level: only the first few send it)
    float cmn[3], cmx[3]; for (int d = 0; d < 3; d++) { cmn[d] = sg->cmin[d]; cmx[d] = sg->cmax[d]; }
    const Mapping m = make_mapping(n, cmn, cmx);
    bins_clear_copies(s_bins, tid, 256u);
    if (tid < 24u) s_acc[tid / 12u][tid % 12u] = (tid % 6u < 3u) ? ENC_POS_INF : ENC_NEG_INF;
    __syncthreads();
    PrimRef pr[CHUNK_ROUNDS];
#pragma unroll
    for (int r = 0; r < CHUNK_ROUNDS; r++) {
      const uint32_t i = begin + (uint32_t)r * 256u + tid;
      if (i - lane < end) {                                      // wave-uniform: this wave's 64 places of the round hold something
        const bool v = i < end;
        pr[r] = PrimRef{}; if (v) pr[r] = load_prim(src + i);
        bins_add_copies(s_bins, m, pr[r], v, lane);
      }
    }
    __syncthreads();
    bins_fold_copies(s_bins, tid, 256u);
    __syncthreads();
    if (wave == 0u) sah_best_wave(s_bins, m, prm.shift, &s_res, lane);
    __syncthreads();
    const int rdim = s_res.dim;
    const bool fallback = (rdim < 0) || forceFallback;          // split invalid -> median split (split_template :144-147)
    const uint32_t nL = fallback ? ((begin + end) / 2u - begin) : s_res.nL, mid = begin + nL;
    const uint32_t dim = fallback ? 0u : (uint32_t)rdim; const int pos = s_res.pos;
    const float ofs = sel3(dim, m.ofs[0], m.ofs[1], m.ofs[2]), scale = sel3(dim, m.scale[0], m.scale[1], m.scale[2]);
    uint32_t sideBits = 0u; unsigned long long lm[CHUNK_ROUNDS], rm[CHUNK_ROUNDS];
    uint32_t aL[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u}, aR[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < CHUNK_ROUNDS; r++) {
      const uint32_t i = begin + (uint32_t)r * 256u + tid;
      lm[r] = rm[r] = 0ull;
      if (i - lane < end) {
        const bool v = i < end;
        bool left = false;
        if (v) {
          const float c2 = sel3(dim, pr[r].lo[0] + pr[r].hi[0], pr[r].lo[1] + pr[r].hi[1], pr[r].lo[2] + pr[r].hi[2]);
          left = fallback ? (i < mid) : (bin_unsafe(c2, ofs, scale) < pos);           // isLeft: bin_unsafe(center2) < pos (:161)
          for (int d = 0; d < 3; d++) {                                                // extend_center2 of the child (:168), thread-private first
            const uint32_t cc = enc(pr[r].lo[d] + pr[r].hi[d]);
            if (left) { aL[d] = min(aL[d], cc); aL[3 + d] = max(aL[3 + d], cc); } else { aR[d] = min(aR[d], cc); aR[3 + d] = max(aR[3 + d], cc); }
          }
          if (fallback) for (int d = 0; d < 3; d++) { atomicMin(&s_acc[left ? 0 : 1][6 + d], enc(pr[r].lo[d])); atomicMax(&s_acc[left ? 0 : 1][9 + d], enc(pr[r].hi[d])); }
        }
        lm[r] = __ballot(v && left); rm[r] = __ballot(v && !left);
        if (left) sideBits |= 1u << r;
      }
      if (lane == 0u) { s_cnt[r * 4 + (int)wave][0] = (uint32_t)__popcll(lm[r]); s_cnt[r * 4 + (int)wave][1] = (uint32_t)__popcll(rm[r]); }
    }
    for (int k = 0; k < 6; k++) {                                                      // wave-reduce the private bounds, one lane publishes
      const uint32_t x = k < 3 ? wave_umin63(aL[k]) : wave_umax63(aL[k]), y = k < 3 ? wave_umin63(aR[k]) : wave_umax63(aR[k]);
      if (lane == 63u) { if (k < 3) { atomicMin(&s_acc[0][k], x); atomicMin(&s_acc[1][k], y); } else { atomicMax(&s_acc[0][k], x); atomicMax(&s_acc[1][k], y); } }
    }
    __syncthreads();
    if (tid < (uint32_t)(CHUNK_ROUNDS * 4)) {                                          // exclusive scan of the (round, wave) counts: where every wave's lefts and rights start
      uint32_t l = s_cnt[tid][0], rr = s_cnt[tid][1]; const uint32_t l0 = l, r0 = rr;
      for (int o = 1; o < CHUNK_ROUNDS * 4; o <<= 1) { const uint32_t ul = (uint32_t)__shfl_up((int)l, o, 64), ur = (uint32_t)__shfl_up((int)rr, o, 64); if (tid >= (uint32_t)o) { l += ul; rr += ur; } }
      s_off[tid][0] = l - l0; s_off[tid][1] = rr - r0;
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < CHUNK_ROUNDS; r++) {
      const uint32_t i = begin + (uint32_t)r * 256u + tid;
      if (i < end) {
        const bool left = (sideBits >> r) & 1u;
        const uint32_t o = left ? begin + s_off[r * 4 + (int)wave][0] + (uint32_t)__popcll(lm[r] & lt)
                                : mid + s_off[r * 4 + (int)wave][1] + (uint32_t)__popcll(rm[r] & lt);
        store_prim(dst + o, pr[r]);
      }
    }
    if (tid < 2u) {                                                                    // the two children: their binary nodes, their places in the work lists
      const uint32_t side = tid, parent = sg->bnode, idL = parent + 1u, idR = parent + 2u * nL, child = side ? idR : idL;
      if (side == 0u) { BNode* par = bnodes + parent; par->left = idL; par->right = idR; par->splitSah = s_res.sah; }
      BNode c{};
      c.begin = side ? mid : begin; c.end = side ? end : mid; c.left = c.right = NIL; c.splitSah = __builtin_inff();
      for (int d = 0; d < 3; d++) {
        c.lo[d] = fallback ? dec(s_acc[side][6 + d]) : (side ? s_res.rlo[d] : s_res.llo[d]);
        c.hi[d] = fallback ? dec(s_acc[side][9 + d]) : (side ? s_res.rhi[d] : s_res.lhi[d]);
      }
      bnodes[child] = c;
      float cmin[3], cmax[3];
      for (int d = 0; d < 3; d++) { cmin[d] = dec(s_acc[side][d]); cmax[d] = dec(s_acc[side][3 + d]); }
      top_emit_child(c.begin, c.end, child, cmin, cmax, next, small, ctr, prm, dstBuf, maxNext, maxSmall, nullptr, 0u);
    }
  } else if (s < numSegs && lastOfLevel) { if (tid == 0u) atomicMax(&ctr->overflow, 3u); }         // a set for the chunked path at a level that was enqueued without it: the commit is repeated
}

__global__ void top_emit(const Seg* segs, BNode* bnodes, Seg* next, SmallEntry* small, Counters* ctr,
                         Params prm, uint32_t dstBuf, uint32_t maxNext, uint32_t maxSmall, const SegX* sx, SegX* nx, uint32_t localMax, const uint32_t* accTop) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t numSegsNow = ctr->numSegs;
  if (s < ctr->numSegs && segs[s].end - segs[s].begin > localMax) {
  const Seg* sg = segs + s;
  const uint32_t capL = sx ? segx_cap_left(sx, s) : sg->nL;
  for (int side = 0; side < 2; side++) {
    // spatial-split builds: the partition's cursors say how many references each side received (a spatial split creates some); a child's capacity ends
    // where its share of the extended range ends
    const uint32_t b = side ? sg->begin + capL : sg->begin;
    const uint32_t e = sx ? (side ? sg->curR : sg->curL) : (side ? sg->end : sg->begin + sg->nL);
    const uint32_t childExtEnd = sx ? (side ? segx_ext_end(sx, s) : sg->begin + capL) : e;
    const uint32_t child = side ? sg->childR : sg->childL;
    float cmin[3], cmax[3];
    uint32_t av[12]; acc_folded(accTop, numSegsNow, sg, s, (uint32_t)side, av);
    for (int d = 0; d < 3; d++) { cmin[d] = dec(av[d]); cmax[d] = dec(av[3 + d]); }
    if (sg->flags & 3u) for (int d = 0; d < 3; d++) { bnodes[child].lo[d] = dec(av[6 + d]); bnodes[child].hi[d] = dec(av[9 + d]); }
    if (sg->flags & 2u) { bnodes[child].begin = b; bnodes[child].end = e; }
    top_emit_child(b, e, child, cmin, cmax, next, small, ctr, prm, dstBuf, maxNext, maxSmall, nx, childExtEnd);
  }
  }
  top_level_end(ctr, maxNext);
}
