// build_sort.inl -- the radix sort of the Morton build (RTC_BUILD_QUALITY_LOW).
// Part of build.hip (included inside its anonymous namespace); see the header of build.hip for the pipeline.
// ------------------------------------------------------------------------------------ radix sort (63-bit key, 32-bit payload)
// The reference sorts its Morton codes with its own radix_sort_u32 (kernels/builders/bvh_builder_morton.h:439 -> common/algorithms/parallel_sort.h);
// until round 6 this file's caller went through hipcub::DeviceRadixSort (the one library kernel of any path, 0.49 ms of a 3.1 ms commit).  This is the same
// sort -- stable, least significant digit first, so equal codes keep their index order -- written for this machine and this key:
//   * 9-bit digits: 7 passes cover the 63 bits exactly (a library sort of 64-bit keys takes 8 passes of 8 bits);
//   * ONE kernel per pass, every key read once and written once ("onesweep"): a workgroup takes a tile of 4096 keys, ranks them (below), publishes the
//     tile's 512 digit counts, and finds the counts of the tiles before it by looking BACK at what they published -- a count ("aggregate") or already a
//     running sum ("prefix") -- instead of a histogram pass + scan in front of every scatter.  Tiles are handed out by a ticket, so the tiles a workgroup
//     waits for have all started; the published words are 64-bit {pass tag, flag, count} written and read with device-scope atomics (the XCDs' L2s are not
//     coherent with each other), so one array serves all seven passes without being cleared in between;
//   * the digit histogram a pass needs IN FRONT (where digit d's keys begin in the output) is counted by the pass before it while it holds the keys in
//     registers (the first one by morton_keys): no histogram kernel;
//   * ranking = where a key goes among the keys of its digit in its tile, in index order: a wavefront finds the lanes that hold the same digit with nine
//     ballots (gfx950 has no match instruction), the lowest ... highest lanes of a group get consecutive ranks behind what the wave has counted for that
//     digit so far (a per-wave LDS counter row; LDS operations of one wave execute in order, so no barrier between the rounds), the waves' rows are summed
//     per digit afterwards;
//   * the tile is put in order in LDS first and written out from there: a thread writes key j of the sorted tile, neighbours write neighbours (runs of one
//     digit are contiguous in the output), instead of 4096 scattered 12-byte stores.
// Payloads of the first pass are the indices themselves (never read).  Deterministic: where a key lands depends on the keys alone.
#ifndef MI355_RS_KPT
#define MI355_RS_KPT 8                      /* keys per thread: a tile is 512 x this many keys (A/B: tools/build_variant.sh) */
#endif
constexpr uint32_t RS_BITS = 9u, RS_RADIX = 1u << RS_BITS, RS_PASSES = 7u, RS_THREADS = 512u, RS_KPT = MI355_RS_KPT, RS_TILE = RS_THREADS * RS_KPT, RS_WAVES = RS_THREADS / 64u;
constexpr uint32_t RS_AGG = 1u, RS_PREFIX = 2u;                 // flags of a published word (0 = nothing yet): this tile's count / the count of this tile and all before it
constexpr uint32_t RS_HIST_WORDS = RS_PASSES * RS_RADIX + 8u;   // the seven digit histograms + the tile tickets of the passes (one memset)
static_assert(RS_THREADS == RS_RADIX, "one thread per digit");
static_assert(RS_BITS * RS_PASSES == 63u, "the passes cover the 63 bits of a Morton code");

__device__ __forceinline__ uint32_t rs_digit(unsigned long long key, uint32_t pass) { return (uint32_t)(key >> (pass * RS_BITS)) & (RS_RADIX - 1u); }
__device__ __forceinline__ void rs_publish(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long rs_peek(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// exclusive scan over the 512 threads of a workgroup (DPP wave scans + the eight wave totals)
__device__ __forceinline__ uint32_t rs_block_exclusive_scan(uint32_t v, uint32_t* s_w, uint32_t tid) {
  const uint32_t lane = tid & 63u, w = tid >> 6;
  const uint32_t incl = wave_incl_scan_u32(v);
  if (lane == 63u) s_w[w] = incl;
  __syncthreads();
  uint32_t add = 0u;
#pragma unroll
  for (uint32_t i = 0; i < RS_WAVES; i++) { const uint32_t t = s_w[i]; add += i < w ? t : 0u; }
  __syncthreads();
  return add + incl - v;
}

// Morton codes of a tile of references (bvh_builder_morton.h: 21 bits per axis of the centroid inside the centroid bounds) + the histogram of their lowest digit
__global__ __launch_bounds__(256) void morton_keys(const PrimRef* prims, uint32_t n, float3 cmin, float3 cscale, unsigned long long* keys, uint32_t* hist0) {
  __shared__ uint32_t s_h[RS_RADIX];
  const uint32_t tid = threadIdx.x, base = blockIdx.x * RS_TILE;
  s_h[tid] = 0u; s_h[tid + 256u] = 0u;
  __syncthreads();
#pragma unroll 4
  for (uint32_t k = 0; k < RS_TILE / 256u; k++) {
    const uint32_t i = base + k * 256u + tid;
    if (i >= n) break;
    const PrimRef p = load_prim(prims + i);
    const float fx = ((p.lo[0] + p.hi[0]) - cmin.x) * cscale.x, fy = ((p.lo[1] + p.hi[1]) - cmin.y) * cscale.y, fz = ((p.lo[2] + p.hi[2]) - cmin.z) * cscale.z;
    const uint32_t ix = (uint32_t)fminf(fmaxf(fx, 0.0f), 2097151.0f), iy = (uint32_t)fminf(fmaxf(fy, 0.0f), 2097151.0f), iz = (uint32_t)fminf(fmaxf(fz, 0.0f), 2097151.0f);
    const unsigned long long key = spread21(ix) | (spread21(iy) << 1) | (spread21(iz) << 2);
    keys[i] = key;
    atomicAdd(&s_h[rs_digit(key, 0u)], 1u);
  }
  __syncthreads();
  for (uint32_t d = tid; d < RS_RADIX; d += 256u) { const uint32_t c = s_h[d]; if (c) atomicAdd(hist0 + d, c); }
}

// (the first digit's histogram of keys that are already there: mi355_sort_keys63, the sort on its own)
__global__ __launch_bounds__(256) void sort_hist0(const unsigned long long* keys, uint32_t n, uint32_t* hist0) {
  __shared__ uint32_t s_h[RS_RADIX];
  const uint32_t tid = threadIdx.x, base = blockIdx.x * RS_TILE;
  s_h[tid] = 0u; s_h[tid + 256u] = 0u;
  __syncthreads();
  for (uint32_t k = 0; k < RS_TILE / 256u; k++) { const uint32_t i = base + k * 256u + tid; if (i < n) atomicAdd(&s_h[rs_digit(keys[i], 0u)], 1u); }
  __syncthreads();
  for (uint32_t d = tid; d < RS_RADIX; d += 256u) { const uint32_t c = s_h[d]; if (c) atomicAdd(hist0 + d, c); }
}

// One pass: keysIn / valsIn (valsIn == nullptr: the payload of element i is i) -> keysOut / valsOut in the order of digit `pass`, stable.
// hist: the seven histograms; hist[pass] is complete (the kernel before this one counted it), hist[pass + 1] is counted here.  ticket[pass] hands out the tiles.
__global__ __launch_bounds__(RS_THREADS) void radix_pass(const unsigned long long* __restrict__ keysIn, const uint32_t* __restrict__ valsIn,
                                                         unsigned long long* __restrict__ keysOut, uint32_t* __restrict__ valsOut, uint32_t n, uint32_t pass,
                                                         uint32_t* hist, unsigned long long* status, uint32_t* ticket) {
  __shared__ unsigned long long s_keys[RS_TILE];
  __shared__ uint32_t s_vals[RS_TILE];
  __shared__ uint32_t s_whist[RS_WAVES][RS_RADIX];               // per wave and digit: keys counted so far -> (after the ranking) where the wave's keys of that digit begin among the tile's
  __shared__ uint32_t s_off[RS_RADIX], s_lstart[RS_RADIX], s_next[RS_RADIX], s_w[RS_WAVES], s_tile;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  if (tid == 0u) s_tile = atomicAdd(ticket + pass, 1u);
#pragma unroll
  for (uint32_t i = 0; i < RS_WAVES; i++) s_whist[i][tid] = 0u;
  s_next[tid] = 0u;
  __syncthreads();
  const uint32_t tile = s_tile, tbase = tile * RS_TILE, nv = min(RS_TILE, n - tbase);
  const bool last = pass + 1u == RS_PASSES;
  // ---- load: wave w owns elements [w * 512, (w + 1) * 512) of the tile, round r of it the 64 consecutive ones from r * 64
  unsigned long long key[RS_KPT]; uint32_t val[RS_KPT], rank[RS_KPT];
#pragma unroll
  for (uint32_t r = 0; r < RS_KPT; r++) {
    const uint32_t li = w * (64u * RS_KPT) + r * 64u + lane;
    key[r] = 0ull; val[r] = 0u;
    if (li < nv) { key[r] = keysIn[tbase + li]; val[r] = valsIn ? valsIn[tbase + li] : tbase + li; }
  }
  // ---- rank inside the wave, round after round; count the next pass's digits on the way
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (uint32_t r = 0; r < RS_KPT; r++) {
    const bool valid = w * (64u * RS_KPT) + r * 64u + lane < nv;
    const uint32_t d = valid ? rs_digit(key[r], pass) : 0u;
    // the lanes that hold my digit: for every bit, the lanes whose bit equals mine = ~(ballot ^ x) with x = all ones if my bit is set (v_bfe_i32), in 32-bit halves
    // (one three-input bit operation per half and bit; a select between ballot and ~ballot on 64-bit lane masks was six instructions per bit)
    const unsigned long long vm = __ballot(valid);
    uint32_t mlo = (uint32_t)vm, mhi = (uint32_t)(vm >> 32);
#pragma unroll
    for (uint32_t b = 0; b < RS_BITS; b++) {
      const uint32_t x = (uint32_t)(-(int)((d >> b) & 1u));
      const unsigned long long bal = __ballot(x != 0u);
      mlo &= ((uint32_t)bal ^ ~x); mhi &= ((uint32_t)(bal >> 32) ^ ~x);
    }
    const unsigned long long m = ((unsigned long long)mhi << 32) | mlo;
    const uint32_t before = s_whist[w][d];                        // (every lane of a group reads the same word; the group's highest lane then moves it on)
    rank[r] = before + (uint32_t)__popcll(m & lt);
    if (valid && (m >> lane) == 1ull) s_whist[w][d] = before + (uint32_t)__popcll(m);
    if (valid && !last) atomicAdd(&s_next[rs_digit(key[r], pass + 1u)], 1u);
  }
  __syncthreads();
  // ---- thread = digit: the waves' counts in wave order, the tile's count published, the counts of the tiles before it looked up
  uint32_t cnt = 0u;
#pragma unroll
  for (uint32_t i = 0; i < RS_WAVES; i++) { const uint32_t t = s_whist[i][tid]; s_whist[i][tid] = cnt; cnt += t; }
  const unsigned long long tag = (unsigned long long)(pass + 1u) << 32;
  unsigned long long* const mine = status + (size_t)tile * RS_RADIX + tid;
  uint32_t excl = 0u;
#ifdef MI355_RS_SKIP_LOOKBACK                                   /* (timing experiment: wrong places, no waiting) */
  if (true) excl = tile * cnt;
  else
#endif
  if (tile == 0u) rs_publish(mine, tag | ((unsigned long long)RS_PREFIX << 30) | cnt);
  else {
    rs_publish(mine, tag | ((unsigned long long)RS_AGG << 30) | cnt);
    // four tiles back at a time (their words are in flight together: a walk of one dependent load per tile was 12 us of a 60 us pass)
    bool found = false;
    for (uint32_t t = tile; t != 0u && !found;) {
      const uint32_t nw = min(4u, t);
      unsigned long long v[4];
#pragma unroll
      for (uint32_t i = 0; i < 4u; i++) v[i] = i < nw ? rs_peek(status + (size_t)(t - 1u - i) * RS_RADIX + tid) : 0ull;
#pragma unroll
      for (uint32_t i = 0; i < 4u; i++) {
        if (i < nw && !found) {
          const unsigned long long* p = status + (size_t)(t - 1u - i) * RS_RADIX + tid;
          while ((uint32_t)(v[i] >> 32) != pass + 1u) { __builtin_amdgcn_s_sleep(1); v[i] = rs_peek(p); }   // (that tile has started -- it drew its ticket before this one -- and publishes before it waits for anybody)
          excl += (uint32_t)v[i] & 0x3FFFFFFFu;
          found = (((uint32_t)v[i] >> 30) & 3u) == RS_PREFIX;
        }
      }
      t -= nw;
    }
    rs_publish(mine, tag | ((unsigned long long)RS_PREFIX << 30) | (excl + cnt));
  }
  if (!last) { const uint32_t c2 = s_next[tid]; if (c2) atomicAdd(hist + (pass + 1u) * RS_RADIX + tid, c2); }
  const uint32_t lstart = rs_block_exclusive_scan(cnt, s_w, tid);                                   // where digit d begins in the sorted tile
  const uint32_t gbase = rs_block_exclusive_scan(hist[pass * RS_RADIX + tid], s_w, tid);            // where digit d begins in the output
  s_lstart[tid] = lstart; s_off[tid] = gbase + excl - lstart;
  __syncthreads();
  // ---- the tile in order, in LDS
#pragma unroll
  for (uint32_t r = 0; r < RS_KPT; r++) {
    if (w * (64u * RS_KPT) + r * 64u + lane < nv) {
      const uint32_t d = rs_digit(key[r], pass), pos = s_lstart[d] + s_whist[w][d] + rank[r];
      s_keys[pos] = key[r]; s_vals[pos] = val[r];
    }
  }
  __syncthreads();
  // ---- ... and out: sorted position j of the tile goes to s_off[its digit] + j
#pragma unroll
  for (uint32_t k = 0; k < RS_KPT; k++) {
    const uint32_t j = k * RS_THREADS + tid;
    if (j < nv) {
      const unsigned long long kk = s_keys[j];
      const uint32_t dst = s_off[rs_digit(kk, pass)] + j;
#ifdef MI355_RS_SKIP_WRITE                                      /* (timing experiment) */
      if (dst == 0xFFFFFFFFu)
#endif
      { keysOut[dst] = kk; valsOut[dst] = s_vals[j]; }
    }
  }
}
