// build_primref.inl -- K1: PrimRef generation and the stable compaction of invalid triangles.
// Part of build.hip (included inside its anonymous namespace); see the header of build.hip for the pipeline.
// ---------------------------------------------------------------------------------- K1 primref_gen
// Triangle p of the concatenated geometries lands at out[p]: no compaction counter (a returning global atomic per
// block costs ~11 ns each, 0.2 ms for a 4.8 M triangle scene, and makes the order depend on block timing).  Invalid
// triangles (index out of range, non-finite or huge coordinate) are marked geom = NIL and counted; only if there are
// any does primref_compact squeeze them out afterwards (stable, so the order is still the input order).
// (round 6) Four triangles per thread and step -- tile t of 1024 consecutive triangles is one workgroup's, thread i takes t * 1024 + k * 256 + i, k = 0 .. 3 -- so that the
// index loads of all four, then the vertex loads of all four are in flight together: one triangle per step was a chain of three dependent round trips (geometry table,
// indices, vertices) with nothing behind it, 133 us for 4.76 M triangles where the bytes need 40.  And the statistics of the outlier cut (build_presplit.inl) are taken here,
// where the boxes are in registers: every workgroup leaves the sum of its valid boxes' areas and their number in areaPart[blockIdx] -- a fixed assignment of triangles to
// threads and a fixed reduction order, so the sums do not depend on timing -- and outlier_stats adds the workgroups' parts in index order (was: outlier_area, a pass of its
// own over the 152 MB of references, 63 us).
struct AreaPart { double area; unsigned long long count; };
__global__ __launch_bounds__(256) void primref_gen(const GeomDesc* geoms, uint32_t numGeoms, uint32_t totalPrims,
                                                   PrimRef* out, Counters* ctr, AreaPart* areaPart) {
  __shared__ uint32_t s_acc[12];
  __shared__ double s_area[4]; __shared__ uint32_t s_cnt[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  if (tid < 12) s_acc[tid] = (tid % 6 < 3) ? ENC_POS_INF : ENC_NEG_INF;
  __syncthreads();
  uint32_t acc[12]; for (int k = 0; k < 12; k++) acc[k] = (k % 6 < 3) ? 0xFFFFFFFFu : 0u;
  uint32_t gi = 0, nInvalid = 0, nValid = 0; GeomDesc g = geoms[0];
  double areaSum = 0.0;
  for (uint32_t t0 = blockIdx.x * 1024u; t0 < totalPrims; t0 += gridDim.x * 1024u) {
    uint32_t gk[4], jk[4], i0[4], i1[4], i2[4], nv[4], vs[4], quad[4]; const char* vb[4]; const char* ib[4]; uint32_t is[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {                                 // geometry of every triangle, its three indices
      const uint32_t p = t0 + (uint32_t)k * 256u + tid;
      gk[k] = NIL; jk[k] = 0; i0[k] = i1[k] = i2[k] = 0xFFFFFFFFu; nv[k] = 0; vs[k] = 0; quad[k] = 0; vb[k] = nullptr; ib[k] = nullptr; is[k] = 0;
      if (p < totalPrims) {
        if (p - g.primOffset >= g.nt) {                           // not in the cached geometry: last geometry with primOffset <= p
          uint32_t lo = 0, hi = numGeoms - 1;
          while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (geoms[mid].primOffset <= p) lo = mid; else hi = mid - 1; }
          gi = lo; g = geoms[lo];
        }
        gk[k] = gi; jk[k] = p - g.primOffset; nv[k] = g.nv; vs[k] = g.vstride; quad[k] = g.quad; vb[k] = g.verts; ib[k] = g.idx; is[k] = g.istride;
        uint32_t pid;
        prim_indices(g, jk[k], i0[k], i1[k], i2[k], pid);
      }
    }
    float vx[4][9]; bool inr[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {                                 // the vertices
      inr[k] = gk[k] != NIL && i0[k] < nv[k] && i1[k] < nv[k] && i2[k] < nv[k];
      for (int q = 0; q < 9; q++) vx[k][q] = 0.0f;
      if (inr[k]) {
        const float* a = (const float*)(vb[k] + (size_t)i0[k] * vs[k]);
        const float* b = (const float*)(vb[k] + (size_t)i1[k] * vs[k]);
        const float* c = (const float*)(vb[k] + (size_t)i2[k] * vs[k]);
        for (int d = 0; d < 3; d++) { vx[k][d] = a[d]; vx[k][3 + d] = b[d]; vx[k][6 + d] = c[d]; }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (gk[k] == NIL) continue;                                 // (behind the last triangle)
      const uint32_t p = t0 + (uint32_t)k * 256u + tid;
      bool ok = inr[k]; PrimRef r{};
      if (ok) {
        for (int d = 0; d < 3; d++) {
          const float x = vx[k][d], y = vx[k][3 + d], z = vx[k][6 + d];
          ok = ok && valid_f(x) && valid_f(y) && valid_f(z);
          r.lo[d] = fminf(fminf(x, y), z); r.hi[d] = fmaxf(fmaxf(x, y), z);
        }
      }
      if (ok && quad[k]) {                                        // non-finite fourth vertex: the reference drops the whole quad
        const uint32_t* q = (const uint32_t*)(ib[k] + (size_t)(jk[k] >> 1) * is[k]);
        const float* o4 = (const float*)(vb[k] + (size_t)((jk[k] & 1u) ? q[0] : q[2]) * vs[k]);
        ok = valid_f(o4[0]) && valid_f(o4[1]) && valid_f(o4[2]);
      }
      r.geom = ok ? gk[k] : NIL; r.prim = jk[k];
      store_prim(out + p, r);
      if (ok) {
        for (int d = 0; d < 3; d++) {
          const uint32_t l = enc(r.lo[d]), h = enc(r.hi[d]), c2 = enc(r.lo[d] + r.hi[d]);   // centroid proxy = lower+upper, never halved (priminfo.h:46-52)
          acc[d] = min(acc[d], l); acc[3 + d] = max(acc[3 + d], h); acc[6 + d] = min(acc[6 + d], c2); acc[9 + d] = max(acc[9 + d], c2);
        }
        areaSum += (double)(2.0f * half_area3(r.hi[0] - r.lo[0], r.hi[1] - r.lo[1], r.hi[2] - r.lo[2]));
        nValid++;
      } else nInvalid++;
    }
  }
  for (int k = 0; k < 12; k++) {
    const uint32_t x = (k % 6 < 3) ? wave_umin63(acc[k]) : wave_umax63(acc[k]);
    if (lane == 63u) { if (k % 6 < 3) atomicMin(&s_acc[k], x); else atomicMax(&s_acc[k], x); }
  }
  for (int o = 32; o > 0; o >>= 1) { areaSum += __shfl_down(areaSum, o, 64); nValid += (uint32_t)__shfl_down((int)nValid, o, 64); }
  if (lane == 0u) { s_area[tid >> 6] = areaSum; s_cnt[tid >> 6] = nValid; }
  const unsigned long long bad = __ballot(nInvalid != 0u);
  if (bad != 0ull && nInvalid) atomicAdd(&ctr->numInvalid, nInvalid);
  __syncthreads();
  if (tid < 12) {                                                // (4096 workgroups: each to its stripe, see Counters::stripe; fold_bounds makes Counters::bounds of them)
    uint32_t* a = &ctr->stripe[blockIdx.x % Counters::STRIPES].bounds[tid]; const uint32_t v = s_acc[tid];
    if (tid % 6 < 3) { if (v != ENC_POS_INF) atomicMin(a, v); } else { if (v != ENC_NEG_INF) atomicMax(a, v); }
  }
  if (tid == 0u && areaPart) { AreaPart ap; ap.area = ((s_area[0] + s_area[1]) + s_area[2]) + s_area[3]; ap.count = (unsigned long long)s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]; areaPart[blockIdx.x] = ap; }
}

// first kernel of a commit: the counters' initial state (what the host used to upload)
__global__ void build_begin(Counters* ctr) {
  const uint32_t t = threadIdx.x;
  uint32_t* w = (uint32_t*)ctr;
  for (uint32_t i = t; i < sizeof(Counters) / 4u; i += blockDim.x) w[i] = 0u;
  __syncthreads();
  if (t < 12u) ctr->bounds[t] = (t % 6u < 3u) ? ENC_POS_INF : ENC_NEG_INF;
  for (uint32_t i = t; i < Counters::STRIPES * 6u; i += blockDim.x) ctr->stripe[i / 6u].cb2[i % 6u] = (i % 6u) < 3u ? ENC_POS_INF : ENC_NEG_INF;   // (the stripes' sums start at zero: cleared above)
  for (uint32_t i = t; i < Counters::STRIPES * 12u; i += blockDim.x) ctr->stripe[i / 12u].bounds[i % 12u] = (i % 6u) < 3u ? ENC_POS_INF : ENC_NEG_INF;
  if (t == 0u) { ctr->rootRef = MI355_EMPTY_REF; ctr->localFirst = 0xFFFFFFFFu; }
}

// rare path: stable compaction of the valid PrimRefs (tile = 256 consecutive entries).  `ctr` != nullptr: the commit runs without a host round trip
// after primref_gen, so the kernels are always enqueued and return at once when there is nothing to squeeze out.
// (one-round-trip commits with outlier pieces behind the references: `n` is the capacity, what is really there ends at base + ctr->outlierCells)
// firstTile: the tiles in front of it have been counted already (MEDIUM commits with the outlier cut: outlier_mark counts the valid references of every tile while it looks
// at them, outlier_emit takes the cut ones off -- this kernel then only covers the tile N falls into and the reserve behind it, not 152 MB of references a third time)
__global__ __launch_bounds__(256) void compact_count(const PrimRef* in, uint32_t n, uint32_t* tileCount, const Counters* ctr, uint32_t base, uint32_t firstTile) {
  if (ctr && ctr->numInvalid == 0u) return;
  if (ctr) n = min(n, base + ctr->outlierCells);
  const uint32_t tile = firstTile + blockIdx.x, p = tile * 256u + threadIdx.x;
  const bool ok = p < n && in[p].geom != NIL;
  const int c = __syncthreads_count(ok);
  if (threadIdx.x == 0) tileCount[tile] = (uint32_t)c;
}
__global__ __launch_bounds__(1024) void compact_scan(uint32_t* tileCount, uint32_t numTiles, Counters* ctr, uint32_t guarded) {
  __shared__ uint32_t s_part[1024], s_first[1024];
  if (guarded && ctr->numInvalid == 0u) return;
  const uint32_t tid = threadIdx.x, per = ((numTiles + 1023u) / 1024u + 7u) & ~7u, b = min(tid * per, numTiles), e = min(b + per, numTiles);
  uint32_t sum = 0, first = 0xFFFFFFFFu;                       // first tile that is not completely valid: nothing in front of it moves
  for (uint32_t i = b; i < e; i += 8u) {                         // eight words per step as two 16-byte loads (load8_fill)
    uint32_t x[8]; load8_fill(tileCount, i, e, 256u, x);
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++) { if (i + k < e) sum += x[k]; if (x[k] != 256u && first == 0xFFFFFFFFu) first = i + k; }
  }
  s_part[tid] = sum; s_first[tid] = first; __syncthreads();
  for (uint32_t o = 512u; o > 0u; o >>= 1) { if (tid < o) s_first[tid] = min(s_first[tid], s_first[tid + o]); __syncthreads(); }
  const uint32_t total = block_exclusive_scan_1024(s_part, tid);
  if (tid == 0) { const uint32_t f = s_first[0]; ctr->numPrims = total; ctr->compactFrom = f == 0xFFFFFFFFu ? total : f * 256u; }
  uint32_t run = s_part[tid];
  for (uint32_t i = b; i < e; i += 8u) {
    uint32_t x[8], y[8]; load8_fill(tileCount, i, e, 0u, x);
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++) { y[k] = run; run += x[k]; }
    store8_upto(tileCount, i, e, y);
  }
}
__global__ __launch_bounds__(256) void compact_scatter(const PrimRef* in, uint32_t n, const uint32_t* tileOfs, PrimRef* out, const Counters* ctr, uint32_t base) {
  __shared__ uint32_t s_w[4];
  if (ctr && ctr->numInvalid == 0u) return;
  if (ctr && blockIdx.x * 256u < ctr->compactFrom) return;     // (tiles in front of the first hole: their references stay where they are)
  if (ctr) n = min(n, base + ctr->outlierCells);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, p = blockIdx.x * 256u + tid;
  PrimRef r{}; bool ok = false;
  if (p < n) { r = load_prim(in + p); ok = r.geom != NIL; }
  const unsigned long long m = __ballot(ok);
  if (lane == 0) s_w[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t off = tileOfs[blockIdx.x]; for (uint32_t w = 0; w < wave; w++) off += s_w[w];
  if (ok) store_prim(out + off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), r);
}

__global__ __launch_bounds__(256) void compact_copyback(const PrimRef* in, PrimRef* out, const Counters* ctr) {   // the squeezed array goes back to where the build expects it
  if (ctr->numInvalid == 0u) return;
  const uint32_t n = ctr->numPrims;
  for (uint32_t i = ctr->compactFrom + blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) store_prim(out + i, load_prim(in + i));
}

// Root of the binary tree and the first work item, made on the device from what primref_gen (and the compaction) left in the counters: the commit
// needs no host round trip to learn the scene bounds or the number of valid triangles (the reference: PrimInfo pinfo = createPrimRefArray(...),
// bvh_builder_sah.cpp:136, then BVHBuilderBinnedSAH::build(pinfo) -- one address space, no round trip there either).
__global__ void bounds_fold(Counters* ctr) { if (blockIdx.x == 0u && threadIdx.x < 12u) fold_bounds(ctr, threadIdx.x); }   // (stepwise path: the host reads the bounds next)
__global__ void root_setup(Counters* ctr, BNode* bnodes, Seg* segs0, SmallEntry* small, uint32_t totalPrims, uint32_t smallThreshold) {   // one workgroup of 64 threads
  if (blockIdx.x != 0u) return;
  if (threadIdx.x < 12u) fold_bounds(ctr, threadIdx.x);          // (commits without the outlier cut: nobody has folded primref_gen's stripes yet)
  __syncthreads();
  if (threadIdx.x != 0u) return;
  const uint32_t n = ctr->numInvalid ? ctr->numPrims : totalPrims;
  ctr->numPrims = n;
  float glo[3], ghi[3], clo[3], chi[3];
  for (int d = 0; d < 3; d++) { glo[d] = dec(ctr->bounds[d]); ghi[d] = dec(ctr->bounds[3 + d]); clo[d] = dec(ctr->bounds[6 + d]); chi[d] = dec(ctr->bounds[9 + d]); }
  // pieces have other centres than their triangles: if anything was cut, the centroid box is the one outlier_mark and outlier_clip measured over the references that stay
  // and the pieces (was: centroid_reset + centroid_bounds_guarded, a pass of its own over the compacted array: 45 us of a 4.75 ms commit)
  if (ctr->outlierPieces != 0u) {
    uint32_t c2[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    for (uint32_t r = 0; r < Counters::STRIPES; r++) for (int k = 0; k < 6; k++) { const uint32_t v = ctr->stripe[r].cb2[k]; c2[k] = k < 3 ? min(c2[k], v) : max(c2[k], v); }
    for (int d = 0; d < 3; d++) { clo[d] = dec(c2[d]); chi[d] = dec(c2[3 + d]); ctr->bounds[6 + d] = c2[d]; ctr->bounds[9 + d] = c2[3 + d]; }
  }
  ctr->rootArea = n ? fmaf(ghi[0] - glo[0], (ghi[1] - glo[1]) + (ghi[2] - glo[2]), (ghi[1] - glo[1]) * (ghi[2] - glo[2])) : 0.0f;
  ctr->numSegsNext = 0; ctr->numChunks = 0; ctr->numSmall = 0; ctr->numSegs = 0; ctr->topLevels = 0;
  if (n == 0u) return;
  BNode rootB{}; for (int d = 0; d < 3; d++) { rootB.lo[d] = glo[d]; rootB.hi[d] = ghi[d]; } rootB.begin = 0; rootB.end = n; rootB.left = rootB.right = NIL; rootB.splitSah = __builtin_inff();
  bnodes[0] = rootB;
  if (n > smallThreshold) {
    Seg s0{}; s0.begin = 0; s0.end = n; s0.bnode = 0; for (int d = 0; d < 3; d++) { s0.cmin[d] = clo[d]; s0.cmax[d] = chi[d]; }
    segs0[0] = s0; ctr->numSegs = 1;
  } else {
    SmallEntry se{}; se.begin = 0; se.end = n; se.bnode = 0; se.buf = 0; for (int d = 0; d < 3; d++) { se.cmin[d] = clo[d]; se.cmax[d] = chi[d]; }
    small[0] = se; ctr->numSmall = 1;
  }
}
