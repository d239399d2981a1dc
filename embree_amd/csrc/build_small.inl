// build_small.inl -- K3: sub-trees finished by one wavefront in LDS.
// Part of build.hip (included inside its anonymous namespace); see the header of build.hip for the pipeline.
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

// A sub-tree root of <= MICRO triangles, parked by the large mode until the wave has finished its larger sets
struct MicroRoot { uint32_t begin, nbuf, bnode; float cmin[3], cmax[3]; };   // nbuf = n | buf << 8
#ifndef MI355_SMALL_ROOTS
#define MI355_SMALL_ROOTS 32
#endif
#ifndef MI355_SMALL_STACK
#define MI355_SMALL_STACK 24
#endif
constexpr uint32_t MICRO_ROOTS = MI355_SMALL_ROOTS;               // (a set of <= 1024 triangles leaves at most 31 of them, one of <= 512 at most 15; the large mode flushes when the list is full)
constexpr uint32_t SMALL_STACK = MI355_SMALL_STACK;               // (the wave goes on with the smaller child and pushes the larger: <= log2(small_threshold / 64) <= 10 entries for any threshold the host lets through)

// R: per-wave LDS scratch of 64 * W words (W = 32 words per triangle when min_leaf >= 2, 48 for min_leaf = 1):
//   the bins are seven PLANES of 64 * W / 8 words -- lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, count -- and the segment that starts at lane b owns the slots
//   [b * W / 8, ...) of every plane, one slot per (axis, bin): slot = b * W / 8 + axis * nb + bin (3 * nb <= n * W / 8 for every splittable n).  A lane's
//   seven atomics go to seven planes, and within a plane the lanes of one instruction are spread over consecutive words: with the record layout this
//   replaces ([axis][bin][8 words] at R + b * W) every atomic of an instruction fell on the same 4 of the 32 banks, 58 % of the kernel's LDS cycles were
//   bank conflicts and the LDS pipe was busy half of the time (profiles/r03_pmc_small_build.md).  Once the candidates are evaluated the same memory holds -- all as planes of 64 words, word j of segment b at
//   base + j * 64 + b -- the split records (15 planes at R), the exchange buffer the partition moves the triangles through (11 planes at R + 960) and
//   the centroid bounds of the NEXT level's segments (6 planes at CB = R + 1664: cleared once the bins are dead, filled by the partition, read at the
//   top of the next level before the bins are cleared again).
// The wave works on SEVERAL sub-trees at a time: the lanes [0, nAct) hold the triangles of all segments that still split,
// segment after segment; a segment that has become a leaf writes its ids and leaves (the partition squeezes its lanes out),
// and whenever a parked root fits into the free lanes it is taken in.  One sub-tree at a time left 58 % of the lanes of a
// level pass idle (profiles/r01_build_history.md: 26.9 of 64 lanes on average over the 6.5 levels of a 43-triangle root).
// (The workgroup is ONE wavefront: -DSM_LDS_SYNC makes the barriers of the micro mode wait for the LDS queue only, not for the global stores of node
// records and ids -- measured: no difference, the kernel is bound by instruction issue, not by waiting.)
#ifdef SM_LDS_SYNC
#define MICRO_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define MICRO_SYNC() __syncthreads()
#endif
template <uint32_t W>
__device__ void micro_flush(uint32_t* R, unsigned long long* s_key, const MicroRoot* roots, uint32_t numRoots,
                            const PrimRef* bufA, const PrimRef* bufB, BNode* bnodes, uint2* finalIds, Counters* ctr, const Params& prm, uint32_t lane) {
  PrimRef p{};
  uint32_t segB = 0, segE = 0, node = 0, gsb = 0;                 // my segment: lanes [segB, segE), binary node, where it begins in the id array
  uint32_t nAct = 0, leaves = 0;                                  // (uniform)
  uint32_t pending = numRoots >= 32u ? 0xFFFFFFFFu : (1u << numRoots) - 1u;
  const uint32_t myRootN = lane < numRoots ? (roots[lane].nbuf & 0xFFu) : 0xFFFFu;
  uint32_t* const CB = R + 1664u;
  const uint32_t addBlk = (1u << prm.shift) - 1u;
  for (;;) {
    // ---- take in parked roots while one fits (first fit)
    while (pending) {
      const unsigned long long fm = __ballot(((pending >> (lane & 31u)) & 1u) != 0u && lane < 32u && myRootN <= 64u - nAct);
      if (fm == 0ull) break;
      const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_ctzll(fm));
      pending &= ~(1u << k);
      const MicroRoot r = roots[k];
      const uint32_t n = r.nbuf & 0xFFu;
      const PrimRef* src = (r.nbuf >> 8) ? bufB : bufA;
      if (n <= prm.minLeaf) {                                     // a leaf as it is (the reference never splits sets of <= minLeafSize, bvh_builder_sah.h:253)
        if (lane < n) { const PrimRef q = load_prim(src + r.begin + lane); finalIds[r.begin + lane] = make_uint2(q.geom & 0x07FFFFFFu, q.prim); }
        if (lane == 0u) ((uint4*)(bnodes + r.bnode))[2] = make_uint4(NIL, NIL, __float_as_uint(__builtin_inff()), 0u);
        leaves++;
        continue;
      }
      if (lane >= nAct && lane < nAct + n) { p = load_prim(src + r.begin + (lane - nAct)); segB = nAct; segE = nAct + n; node = r.bnode; gsb = r.begin; }
      if (lane == nAct) for (int d = 0; d < 3; d++) { CB[d * 64 + nAct] = zlo(r.cmin[d]); CB[(3 + d) * 64 + nAct] = zhi(r.cmax[d]); }
      nAct += n;
    }
    if (nAct == 0u) break;
    const bool act = lane < nAct;                                 // every lane below nAct sits in a segment that splits
#ifdef SM_STATS
    if (lane == 0u) { atomicAdd(&ctr->padC[0], 1u); atomicAdd(&ctr->padC[1], nAct); }
#endif
    MICRO_SYNC();                                             // (the new roots' centroid bounds)
    // ---- L0: bin mapping of my segment (BinMapping, heuristic_binning.h:46-55); clear bins and keys
    const uint32_t n = segE - segB;
    float ofs[3] = {0, 0, 0}, scale[3] = {0, 0, 0}; uint32_t nb = 4;
    if (act) {
      float cmin[3], cmax[3];
      for (int d = 0; d < 3; d++) { cmin[d] = unzlo(CB[d * 64 + segB]); cmax[d] = unzhi(CB[(3 + d) * 64 + segB]); }
      const Mapping m = make_mapping(n, cmin, cmax);
      for (int d = 0; d < 3; d++) { ofs[d] = m.ofs[d]; scale[d] = m.scale[d]; }
      nb = m.nb;
    }
    MICRO_SYNC();                                             // everybody has read the centroid bounds and is done with the exchange buffer
#pragma unroll
    for (uint32_t i = 0; i < (W == 32u ? 7u : 12u); i++) ((uint4*)R)[i * 64u + lane] = make_uint4(0u, 0u, 0u, 0u);   // W = 32: the seven planes are the first 7 KB
    s_key[lane] = ~0ull;
    MICRO_SYNC();
    // ---- L1: bin (BinInfoT::bin, heuristic_binning.h:210-257)
    constexpr uint32_t SS = W / 8u, P = 64u * SS;                // slots per lane, words per plane
    uint32_t* const sb = R + segB * SS;
    {
      // Lanes of an instruction that hit the SAME word cost the LDS ~3 cycles each (profiles/r03_lds_atomics.md), and neighbouring triangles of a segment usually
      // share a bin: the even lane of a pair whose odd neighbour is in the same segment and bin hands its box over (quad_perm [1,0,3,2]) and stays out.
      uint32_t z[6] = {zlo(p.lo[0]), zlo(p.lo[1]), zlo(p.lo[2]), zhi(p.hi[0]), zhi(p.hi[1]), zhi(p.hi[2])}, zo[6];
      for (int k = 0; k < 6; k++) zo[k] = dpp_u<0xB1, 0xF>(z[k], z[k]);
      for (int d = 0; d < 3; d++) {
        const uint32_t b = act ? (uint32_t)bin_clamped(p.lo[d] + p.hi[d], ofs[d], scale[d], nb) : 0u;
        const uint32_t key = act ? ((segB << 8) | b) : (0xFFFF0000u | lane);
        const bool same = dpp_u<0xB1, 0xF>(key, key) == key, taker = same && (lane & 1u) != 0u, giver = same && (lane & 1u) == 0u;
        if (act && !giver) {
          uint32_t* e = sb + (uint32_t)d * nb + b;
          atomicMax(&e[0], taker ? max(z[0], zo[0]) : z[0]); atomicMax(&e[P], taker ? max(z[1], zo[1]) : z[1]); atomicMax(&e[2u * P], taker ? max(z[2], zo[2]) : z[2]);
          atomicMax(&e[3u * P], taker ? max(z[3], zo[3]) : z[3]); atomicMax(&e[4u * P], taker ? max(z[4], zo[4]) : z[4]); atomicMax(&e[5u * P], taker ? max(z[5], zo[5]) : z[5]);
          atomicAdd(&e[6u * P], taker ? 2u : 1u);
        }
      }
    }
    MICRO_SYNC();
    // ---- L2: candidates (BinInfoT::best :339-386): lane j of a segment evaluates candidates j, j + n, ...;
    //      candidate c = axis * (nb - 1) + (pos - 1), so the minimum of (sah, c) is the reference's choice
    float bestSah = __builtin_inff(); uint32_t bestC = NIL, bestNL = 0;
    float bl[3] = {0, 0, 0}, bh[3] = {0, 0, 0}, rl[3] = {0, 0, 0}, rh[3] = {0, 0, 0};
    if (act && nb == 4u) {
      // the common case (n < 20): lane j of the segment sweeps axis j once -- suffix bounds S1..S3, then a running prefix; 4 bins are read
      // once (8 x 16 bytes) instead of once per candidate
      for (uint32_t axis = lane - segB; axis < 3u; axis += n) {
        if (sel3(axis, scale[0], scale[1], scale[2]) == 0.0f) continue;          // mapping.invalid(dim) :375
        const uint32_t* e = sb + axis * 4u;                                     // plane k holds word k of the four bins side by side
        uint32_t q[7][4];
#pragma unroll
        for (int k = 0; k < 7; k++) {
          if (W == 32u) { const uint4 x = *(const uint4*)(e + k * P); q[k][0] = x.x; q[k][1] = x.y; q[k][2] = x.z; q[k][3] = x.w; }   // (16-byte aligned: 4 slots per lane)
          else { const uint2 x = *(const uint2*)(e + k * P), y = *(const uint2*)(e + k * P + 2u); q[k][0] = x.x; q[k][1] = x.y; q[k][2] = y.x; q[k][3] = y.y; }
        }
        float lo[4][3], hi[4][3]; uint32_t cn[4];                             // (an empty bin holds zeros: they decode to quiet NaNs, which vmin / vmax drop)
#pragma unroll
        for (int b = 0; b < 4; b++) {
          cn[b] = q[6][b];
          lo[b][0] = unzlo(q[0][b]); lo[b][1] = unzlo(q[1][b]); lo[b][2] = unzlo(q[2][b]);
          hi[b][0] = unzhi(q[3][b]); hi[b][1] = unzhi(q[4][b]); hi[b][2] = unzhi(q[5][b]);
        }
        float slo[4][3], shi[4][3]; uint32_t sn[4];                            // suffix: bins pos..3
        for (int d = 0; d < 3; d++) { slo[3][d] = lo[3][d]; shi[3][d] = hi[3][d]; } sn[3] = cn[3];
#pragma unroll
        for (int b = 2; b >= 1; b--) { for (int d = 0; d < 3; d++) { slo[b][d] = vmin(lo[b][d], slo[b + 1][d]); shi[b][d] = vmax(hi[b][d], shi[b + 1][d]); } sn[b] = cn[b] + sn[b + 1]; }
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
          for (int d = 0; d < 3; d++) { llo[d] = vmin(llo[d], lo[pos][d]); lhi[d] = vmax(lhi[d], hi[pos][d]); }
          lN += cn[pos];
        }
      }
    } else if (act) {
#ifdef SM_STATS
#endif
      const uint32_t nb1 = nb - 1u, ncand = 3u * nb1;
      for (uint32_t c = lane - segB; c < ncand; c += n) {
        const uint32_t axis = (c >= nb1 ? 1u : 0u) + (c >= 2u * nb1 ? 1u : 0u), pos = c - axis * nb1 + 1u;
        if (sel3(axis, scale[0], scale[1], scale[2]) == 0.0f) continue;          // mapping.invalid(dim) :375
        // both sides are merged as they lie in the bins (atomicMax encodings, zero = nothing) and decoded once
        uint32_t zl[6] = {0u, 0u, 0u, 0u, 0u, 0u}, zr[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        uint32_t lN = 0, rN = 0;
        const uint32_t* e = sb + axis * nb;
        for (uint32_t b = 0; b < nb; b++) {
          const bool isL = b < pos;
          const uint32_t v[6] = {e[b], e[P + b], e[2u * P + b], e[3u * P + b], e[4u * P + b], e[5u * P + b]}, cnt = e[6u * P + b];
          for (int k = 0; k < 6; k++) { zl[k] = max(zl[k], isL ? v[k] : 0u); zr[k] = max(zr[k], isL ? 0u : v[k]); }
          lN += isL ? cnt : 0u; rN += isL ? 0u : cnt;
        }
        if (lN == 0u || rN == 0u) continue;
        float llo[3], lhi[3], rlo[3], rhi[3];
        for (int d = 0; d < 3; d++) { llo[d] = unzlo(zl[d]); lhi[d] = unzhi(zl[3 + d]); rlo[d] = unzlo(zr[d]); rhi[d] = unzhi(zr[3 + d]); }
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
    MICRO_SYNC();                                             // bins are dead from here on: R now holds split records + exchange buffer + next centroid bounds
    const unsigned long long win = act ? s_key[segB] : 0ull;
    for (int k = 0; k < 6; k++) CB[k * 64 + lane] = 0u;          // (zero = the identity of the atomicMax encodings)
    const bool fb = act && win == ~0ull;                          // no valid candidate -> median split (split_template :144-147)
    uint32_t* const rec = R + segB;                               // word j of my segment's record: rec[j * 64] (planes: neighbouring segments, neighbouring banks)
    if (act && !fb && key == win) {
      const uint32_t nb1 = nb - 1u, axis = (bestC >= nb1 ? 1u : 0u) + (bestC >= 2u * nb1 ? 1u : 0u), pos = bestC - axis * nb1 + 1u;
      rec[0] = axis | (pos << 8); rec[64] = bestNL; rec[128] = __float_as_uint(bestSah);
      for (int d = 0; d < 3; d++) { rec[(3 + d) * 64] = __float_as_uint(bl[d]); rec[(6 + d) * 64] = __float_as_uint(bh[d]); rec[(9 + d) * 64] = __float_as_uint(rl[d]); rec[(12 + d) * 64] = __float_as_uint(rh[d]); }
    }
    if (fb && lane == segB) {
      rec[0] = 1u << 16; rec[64] = n >> 1; rec[128] = __float_as_uint(__builtin_inff());      // (begin + end) / 2 - begin
      for (int k = 3; k < 15; k++) rec[k * 64] = 0u;
    }
    MICRO_SYNC();
    if (__ballot(fb) != 0ull) {                                   // child geometry bounds of a median split: reduce over the triangles
      if (fb) {
        const uint32_t o = lane < segB + rec[64] ? 3u : 9u;
        for (int d = 0; d < 3; d++) { atomicMax(&rec[(o + d) * 64u], zlo(p.lo[d])); atomicMax(&rec[(o + 3u + d) * 64u], zhi(p.hi[d])); }
      }
      MICRO_SYNC();
    }
    // ---- L4: partition (heuristic_binning_array_aligned.h:150-176): new lane of my triangle, child centroid bounds, node records.
    //      A child of <= min_leaf triangles is a leaf: its triangles write their ids and leave; the others close ranks.
    bool left = false; uint32_t nL = 0;
    if (act) {
      const uint32_t w0 = rec[0], dim = w0 & 3u, pos = (w0 >> 8) & 0xFFu; nL = rec[64];
      const float c2 = sel3(dim, p.lo[0] + p.hi[0], p.lo[1] + p.hi[1], p.lo[2] + p.hi[2]);
      left = (w0 >> 16) ? (lane < segB + nL) : (bin_unsafe(c2, sel3(dim, ofs[0], ofs[1], ofs[2]), sel3(dim, scale[0], scale[1], scale[2])) < (int)pos);
    }
    // the partition permutes inside a segment's lanes: as a PLACE, lane x belongs to the left child iff x < segB + nL
    const unsigned long long gone = __ballot(act && (lane < segB + nL ? nL : n - nL) <= prm.minLeaf);
    // my rank among the lefts / rights of my segment: lanes-below counts (v_mbcnt) of the wave's ballot, minus what the segment's first lane counts
    const unsigned long long lb = __ballot(act && left);
    const uint32_t mbL = __builtin_amdgcn_mbcnt_hi((uint32_t)(lb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lb, 0u));
    const uint32_t mbG = __builtin_amdgcn_mbcnt_hi((uint32_t)(gone >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)gone, 0u));
    const uint32_t rankL = mbL - (uint32_t)__shfl((int)mbL, (int)segB, 64), rankR = (lane - segB) - rankL;
    bool leafHead = false;
    const uint32_t below = (uint32_t)__shfl((int)mbG, (int)(left ? segB : segB + nL) & 63, 64);   // places below my child that were given up (whole segments)
    if (act) {
      const uint32_t nSegB = left ? segB : segB + nL, cs = left ? nL : n - nL, nNode = left ? node + 1u : node + 2u * nL, ngsb = left ? gsb : gsb + nL;
      const uint32_t npos = left ? segB + rankL : segB + nL + rankR;
      if (cs <= prm.minLeaf) {
        finalIds[ngsb + (npos - nSegB)] = make_uint2(p.geom & 0x07FFFFFFu, p.prim);   // (top 5 bits: split budget of spatial-split builds, build_spatial.inl)
        leafHead = npos == nSegB;
        if (leafHead) ((uint4*)(bnodes + nNode))[2] = make_uint4(NIL, NIL, __float_as_uint(__builtin_inff()), 0u);
      } else {
        const uint32_t cB = nSegB - below, cpos = npos - below;
        for (int d = 0; d < 3; d++) { const float cc = p.lo[d] + p.hi[d]; atomicMax(&CB[d * 64 + cB], zlo(cc)); atomicMax(&CB[(3 + d) * 64 + cB], zhi(cc)); }
        uint32_t* X = R + 960u + cpos;
        X[0] = __float_as_uint(p.lo[0]); X[64] = __float_as_uint(p.lo[1]); X[128] = __float_as_uint(p.lo[2]); X[192] = p.geom;
        X[256] = __float_as_uint(p.hi[0]); X[320] = __float_as_uint(p.hi[1]); X[384] = __float_as_uint(p.hi[2]); X[448] = p.prim;
        X[512] = cB | ((cB + cs) << 8); X[576] = nNode; X[640] = ngsb;
      }
      if (lane == segB) {                                        // one lane per segment: my links, my children's boxes and ranges
        float cb[12];
        for (int k = 0; k < 12; k++) cb[k] = __uint_as_float(rec[(3 + k) * 64]);
        if ((rec[0] >> 16) != 0u) for (int k = 0; k < 12; k++) cb[k] = (k % 6) < 3 ? unzlo(rec[(3 + k) * 64]) : unzhi(rec[(3 + k) * 64]);   // (median split: reduced with the atomicMax encodings)
        const uint32_t L = node + 1u, Rr = node + 2u * nL;
        ((uint4*)(bnodes + node))[2] = make_uint4(L, Rr, rec[128], 0u);
        ((float4*)(bnodes + L))[0] = make_float4(cb[0], cb[1], cb[2], __uint_as_float(gsb));
        ((float4*)(bnodes + L))[1] = make_float4(cb[3], cb[4], cb[5], __uint_as_float(gsb + nL));
        ((float4*)(bnodes + Rr))[0] = make_float4(cb[6], cb[7], cb[8], __uint_as_float(gsb + nL));
        ((float4*)(bnodes + Rr))[1] = make_float4(cb[9], cb[10], cb[11], __uint_as_float(gsb + n));
      }
    }
    leaves += (uint32_t)__popcll(__ballot(leafHead));
    nAct -= (uint32_t)__popcll(gone);
    MICRO_SYNC();
    // ---- L5: pick up the triangle that moved to my lane
    if (lane < nAct) {
      const uint32_t* X = R + 960u + lane;
      p.lo[0] = __uint_as_float(X[0]); p.lo[1] = __uint_as_float(X[64]); p.lo[2] = __uint_as_float(X[128]); p.geom = X[192];
      p.hi[0] = __uint_as_float(X[256]); p.hi[1] = __uint_as_float(X[320]); p.hi[2] = __uint_as_float(X[384]); p.prim = X[448];
      segB = X[512] & 0xFFu; segE = X[512] >> 8; node = X[576]; gsb = X[640];
    }
  }
  if (lane == 0u && leaves) atomicAdd(&ctr->stripe[blockIdx.x % Counters::STRIPES].numBLeaves, leaves);
  MICRO_SYNC();
}

// The order the sub-trees are taken in: LARGEST FIRST.  A sub-tree is one wavefront's for 100 - 500 us and the list is ~3.8 of them per wave slot of the chip; taken in the
// order the top phase happened to append them, the kernel ends with a few 500-triangle sets that started last while the other slots stand empty (2.5 of 3.5 waves per SIMD
// resident on average over the kernel, profiles/r06_pmc_small_build.md).  One workgroup sorts the list's INDICES by size class (128 classes, counting sort in LDS); the order
// within a class is whatever the atomics give -- no tree depends on which wave builds which sub-tree (implicit numbering, disjoint ranges).
__global__ __launch_bounds__(1024) void small_order(const SmallEntry* entries, const Counters* ctr, uint32_t* order, uint32_t maxSmall, uint32_t small) {
  __shared__ uint32_t s_cnt[128], s_off[128];
  const uint32_t tid = threadIdx.x, num = min(ctr->numSmall, maxSmall);
  const uint32_t* size = (const uint32_t*)(entries + maxSmall);   // the entries' sizes, side by side (top_emit_child_at)
  if (tid < 128u) s_cnt[tid] = 0u;
  __syncthreads();
  for (uint32_t i = tid; i < num; i += 1024u) atomicAdd(&s_cnt[min(127u, (size[i] * 127u) / small)], 1u);
  __syncthreads();
  if (tid < 128u) { uint32_t run = 0u; for (uint32_t c = tid + 1u; c < 128u; c++) run += s_cnt[c]; s_off[tid] = run; }   // (the larger classes come first)
  __syncthreads();
  for (uint32_t i = tid; i < num; i += 1024u) order[atomicAdd(&s_off[min(127u, (size[i] * 127u) / small)], 1u)] = i;
}

#ifdef MI355_SMALL_WAVES                                      /* A/B: a register budget for this many waves per SIMD (tools/build_variant.sh) */
__attribute__((amdgpu_waves_per_eu(MI355_SMALL_WAVES, MI355_SMALL_WAVES)))
#endif
__global__ __launch_bounds__(64) void small_build(const SmallEntry* entries, PrimRef* bufA, PrimRef* bufB, BNode* bnodes,
                                                  uint2* finalIds, Counters* ctr, Params prm, uint32_t W, const uint32_t* order) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_R[];   // max(BINS_WORDS, 64 * W) words: bins / micro scratch
  __shared__ SplitResult s_res;
  __shared__ uint32_t s_acc[2][12];
  __shared__ StackEntry s_stack[SMALL_STACK];
  __shared__ unsigned long long s_key[64];
  __shared__ MicroRoot s_roots[MICRO_ROOTS];
  uint32_t numRoots = 0;
  uint32_t* const s_bins = s_R;
  const uint32_t lane = threadIdx.x;
  if (blockIdx.x >= ctr->numSmall) return;                       // the grid is an upper bound (the host does not read the list's length back)
  const SmallEntry e0 = entries[order ? order[blockIdx.x] : blockIdx.x];
  StackEntry cur; cur.begin = e0.begin; cur.end = e0.end; cur.bnode = e0.bnode; cur.buf = e0.buf;
  for (int d = 0; d < 3; d++) { cur.cmin[d] = e0.cmin[d]; cur.cmax[d] = e0.cmax[d]; }
  uint32_t sp = 0;
  bool done = false;
  uint32_t iter = 0;
#ifdef SM_TIME
  unsigned long long tm[4] = {0, 0, 0, 0}, t0;
#define SM_T0() t0 = __builtin_readcyclecounter()
#define SM_T(k) do { const unsigned long long t1_ = __builtin_readcyclecounter(); tm[k] += t1_ - t0; t0 = t1_; } while (0)
#else
#define SM_T0()
#define SM_T(k)
#endif
  while (!done) {
  for (; iter < (1u << 20); iter++) {         // the cap is a safety net only: <= 2*small_threshold iterations are possible
    const uint32_t n = cur.end - cur.begin;
    PrimRef* src = cur.buf ? bufB : bufA;
    PrimRef* dst = cur.buf ? bufA : bufB;
    if (n <= MICRO) {                                            // parked: the micro mode takes the roots of this sub-tree in together (micro_flush)
      if (lane == 0) {
        MicroRoot r; r.begin = cur.begin; r.nbuf = n | (cur.buf << 8); r.bnode = cur.bnode;
        for (int d = 0; d < 3; d++) { r.cmin[d] = cur.cmin[d]; r.cmax[d] = cur.cmax[d]; }
        s_roots[numRoots] = r;
      }
      numRoots++;
      if (sp == 0) { done = true; break; }
      cur = s_stack[--sp];
      __syncthreads();
      if (numRoots == MICRO_ROOTS) break;                        // (only with small_threshold > 1024)
      continue;
    }
    SM_T0();
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
    SM_T(0);
    sah_best_wave(s_bins, m, prm.shift, &s_res, lane);
    __syncthreads();
    SM_T(1);
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
    SM_T(2);
  }
  if (iter >= (1u << 20)) done = true;
  __syncthreads();
  SM_T0();
  if (W == 32u) micro_flush<32u>(s_R, s_key, s_roots, numRoots, bufA, bufB, bnodes, finalIds, ctr, prm, lane);
  else micro_flush<48u>(s_R, s_key, s_roots, numRoots, bufA, bufB, bnodes, finalIds, ctr, prm, lane);
  numRoots = 0;
  SM_T(3);
  }
#ifdef SM_TIME
  if (lane == 0u) for (int k = 0; k < 4; k++) atomicAdd(&ctr->smTime[k], tm[k]);
#endif
}
