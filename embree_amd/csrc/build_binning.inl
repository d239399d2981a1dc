// build_binning.inl -- bin mapping, binning helpers (rows / runs), SAH sweep of one wavefront.
// Part of build.hip (included inside its anonymous namespace); see the header of build.hip for the pipeline.
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

__device__ __forceinline__ uint32_t wave_uadd63(uint32_t v) { asm volatile(MI355_WAVE_REDUCE63("v_add_u32_dpp") : "+v"(v)); return v; }

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

// History of top_bin's inner loop (profiles/r01_build_history.md): wave-uniform runs (above, still what one wavefront of small_build uses) 129 us per full pass;
// a lane-private current bin per axis 110 us; DPP row aggregation -- the 16 lanes of a row form two groups, lowest and highest bin, reduced with row_shr steps, lane 15 issues the
// atomics -- 92 us, 70 us with the DPP operands folded into v_min / v_max (round 3).  All of them pay ~100+ VALU instructions per reference and axis to keep lanes off the same word;
// round 4 replaced them by private copies of the bins (below): ~52 us.

// Private copies of the bins instead of combining lanes in registers: lane l of a wave bins into copy l mod BIN_COPIES, so consecutive references -- which share a bin more
// often than not -- do not meet on a word, and an atomic instruction of 64 lanes finds at least BIN_COPIES different words on as many banks
// (the copies are an odd number of words apart).  10-14 cycles per instruction (profiles/r03_lds_atomics.md) with 16 copies, a little more with 8, which let more workgroups share a CU (measured: 8 is faster) and ~25 VALU instructions per
// reference and axis, where the row aggregation of bins_add_rows spends ~110 VALU instructions to get the same-word lanes down to two per row: the kernels
// that bin whole chunks were bound by exactly those (top_bin: 70 us per pass over the crown stand-in = 4.76 M references).  The copies are folded once per
// workgroup; min / max / count do not care in which order.
#ifndef MI355_BIN_COPIES
#define MI355_BIN_COPIES 8
#endif
constexpr uint32_t BIN_COPIES = MI355_BIN_COPIES, COPY_STRIDE = BINS_WORDS + 1u;
__device__ __forceinline__ void bins_clear_copies(uint32_t* bins, uint32_t tid, uint32_t nthreads) {
  for (uint32_t w = tid; w < (uint32_t)BINS_WORDS; w += nthreads) {
    const uint32_t k = w % BINW, v = k < 3 ? ENC_POS_INF : (k < 6 ? ENC_NEG_INF : 0u);
#pragma unroll
    for (uint32_t c = 0; c < BIN_COPIES; c++) bins[c * COPY_STRIDE + w] = v;
  }
}
// (round 6) PAIRS first: lanes 2k and 2k + 1 hold consecutive references, which share a bin more often than not -- then the odd lane bins the union of the two boxes
// with a count of 2 and the even lane stays out (one quad_perm exchange of the six box words for all three axes, one of the bin index per axis).  The two lanes of a
// pair work on the SAME copy (copy = lane / 2 mod BIN_COPIES), so what is left to meet on a word are lanes 16 references apart, four per copy instead of eight: the
// same-word lanes of an atomic instruction are what top_bin waits for (~3 cycles each, profiles/r03_lds_atomics.md).  min / max / add: the bins do not change by a bit.
// EVERY lane of the wave calls it (valid = holds a reference): the exchange reads the neighbour's registers.
#ifndef MI355_BIN_PAIRS
#define MI355_BIN_PAIRS 1
#endif
__device__ __forceinline__ void bins_add_copies(uint32_t* bins, const Mapping& m, const PrimRef& p, bool valid, uint32_t lane) {
  uint32_t c[6];
  for (int k = 0; k < 3; k++) { c[k] = enc(p.lo[k]); c[3 + k] = enc(p.hi[k]); }
#if MI355_BIN_PAIRS
  uint32_t u[6];                                                // the pair's box
  for (int k = 0; k < 3; k++) { u[k] = min(c[k], dpp_u<0xB1, 0xF>(c[k], c[k])); u[3 + k] = max(c[3 + k], dpp_u<0xB1, 0xF>(c[3 + k], c[3 + k])); }
  const bool odd = (lane & 1u) != 0u;
  uint32_t* mine = bins + ((lane >> 1) & (BIN_COPIES - 1u)) * COPY_STRIDE;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const uint32_t b = valid ? (uint32_t)bin_clamped(p.lo[d] + p.hi[d], m.ofs[d], m.scale[d], m.nb) : 0xFFFFFFFFu - lane;   // (a lane without a reference pairs with nobody)
    const bool same = dpp_u<0xB1, 0xF>(b, b) == b;
    if (valid && !(same && !odd)) {
      uint32_t* e = mine + (d * NBINS + b) * BINW;
      atomicMin(&e[0], same ? u[0] : c[0]); atomicMin(&e[1], same ? u[1] : c[1]); atomicMin(&e[2], same ? u[2] : c[2]);
      atomicMax(&e[3], same ? u[3] : c[3]); atomicMax(&e[4], same ? u[4] : c[4]); atomicMax(&e[5], same ? u[5] : c[5]);
      atomicAdd(&e[6], same ? 2u : 1u);
    }
  }
#else
  if (!valid) return;
  uint32_t* mine = bins + (lane & (BIN_COPIES - 1u)) * COPY_STRIDE;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const uint32_t b = (uint32_t)bin_clamped(p.lo[d] + p.hi[d], m.ofs[d], m.scale[d], m.nb);
    uint32_t* e = mine + (d * NBINS + b) * BINW;
    atomicMin(&e[0], c[0]); atomicMin(&e[1], c[1]); atomicMin(&e[2], c[2]);
    atomicMax(&e[3], c[3]); atomicMax(&e[4], c[4]); atomicMax(&e[5], c[5]);
    atomicAdd(&e[6], 1u);
  }
#endif
}
// every word of copy 0 becomes the fold of its copies (the caller puts a barrier on either side)
__device__ __forceinline__ void bins_fold_copies(uint32_t* bins, uint32_t tid, uint32_t nthreads) {
  for (uint32_t w = tid; w < (uint32_t)BINS_WORDS; w += nthreads) {
    const uint32_t k = w % BINW; uint32_t v = bins[w];
#pragma unroll
    for (uint32_t c = 1; c < BIN_COPIES; c++) { const uint32_t x = bins[c * COPY_STRIDE + w]; v = k < 3 ? min(v, x) : (k < 6 ? max(v, x) : v + x); }
    bins[w] = v;
  }
}

struct SplitResult { float sah; int dim, pos; uint32_t nL; float llo[3], lhi[3], rlo[3], rhi[3]; };

// BinInfoT::best (heuristic_binning.h:339-386) by ONE wavefront, one axis per pass: lanes 0-31 hold the axis' 32 bins in order, lanes 32-63 hold them
// REVERSED, so ONE inclusive prefix scan over 32-lane halves -- four row_shr steps and a row_bcast:15, DPP folded into the min / max / add -- leaves
// "everything left of the plane" in the lower half and "everything right of it" in the upper half.  Lane b - 1 prices the left side of candidate b,
// lane 63 - b its right side; lane b combines them (two lane shifts and one bpermute per axis).  The reference's "first strict minimum per axis, then
// first better axis" is the lexicographic minimum of (sah, axis, pos).  Result lands in `res` (LDS).
// (The first version let every candidate loop over all bins: ~1900 instructions per lane.  The second scanned two axes per pass with __shfl_up / __shfl_down:
// 140 ds_bpermute round trips per call, 15 % of small_build's wave cycles and the stretch of top_local where three of its four waves wait.)
#define MI355_SCAN7(CTRL) \
  "v_min_f32_dpp %0, %0, %0 " CTRL "\n v_min_f32_dpp %1, %1, %1 " CTRL "\n v_min_f32_dpp %2, %2, %2 " CTRL "\n" \
  "v_max_f32_dpp %3, %3, %3 " CTRL "\n v_max_f32_dpp %4, %4, %4 " CTRL "\n v_max_f32_dpp %5, %5, %5 " CTRL "\n v_add_u32_dpp %6, %6, %6 " CTRL "\n"
__device__ __forceinline__ void scan32_box(float (&lo)[3], float (&hi)[3], uint32_t& n) {      // inclusive, over each half of the wave; a lane without a source keeps its value
  asm volatile("s_nop 1\n"
               MI355_SCAN7("row_shr:1 row_mask:0xf bank_mask:0xf") MI355_SCAN7("row_shr:2 row_mask:0xf bank_mask:0xf")
               MI355_SCAN7("row_shr:4 row_mask:0xf bank_mask:0xf") MI355_SCAN7("row_shr:8 row_mask:0xf bank_mask:0xf")
               MI355_SCAN7("row_bcast:15 row_mask:0xa bank_mask:0xf")
               : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(n));
}
#undef MI355_SCAN7
__device__ __forceinline__ float lane_shr1(float v) { return __uint_as_float(dpp_u<0x138, 0xF>(__float_as_uint(v), __float_as_uint(v))); }   // wave_shr:1 (lane 0 keeps its value)
__device__ void sah_best_wave(const uint32_t* bins, const Mapping& m, uint32_t shift, SplitResult* res, uint32_t lane) {
  const bool upper = lane >= 32u;
  const uint32_t b = upper ? 63u - lane : lane;                   // my bin
  const uint32_t add = (1u << shift) - 1u;
  unsigned long long bestKey = ~0ull;
  float klo[3][3], khi[3][3]; uint32_t kn[3];                      // my scan results per axis: the winner's two lanes write them out
#pragma unroll
  for (uint32_t axis = 0; axis < 3u; axis++) {
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    uint32_t n = 0;
    if (b < m.nb) {
      const uint32_t* e = bins + (axis * NBINS + b) * BINW;
      n = e[6];
      if (n) for (int d = 0; d < 3; d++) { lo[d] = dec(e[d]); hi[d] = dec(e[3 + d]); }
    }
    scan32_box(lo, hi, n);
    for (int d = 0; d < 3; d++) { klo[axis][d] = lo[d]; khi[axis][d] = hi[d]; } kn[axis] = n;
    // my side of a candidate: lower lane b = left side of candidate b + 1, upper lane 63 - b = right side of candidate b
    const float A = half_area3(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]);
    const float cntf = (float)((n + add) >> shift);
    const float rprod = n ? A * cntf : -1.0f;                      // (an empty side is never selected)
    const float lA = lane_shr1(A), lcnt = lane_shr1(cntf);         // candidate b: its left side comes from lane b - 1 ...
    const uint32_t lN = dpp_u<0x138, 0xF>(n, n);
    const float rp = __shfl(rprod, (int)(63u - lane), 64);         // ... its right side from lane 63 - b
    const bool cand = !upper && b != 0u && b < m.nb && sel3(axis, m.scale[0], m.scale[1], m.scale[2]) != 0.0f && lN != 0u && rp >= 0.0f;   // mapping.invalid(dim) :375, pos != 0 :379
    if (cand) {
      const float sah = fmaf(lA, lcnt, rp);                        // :367
      const unsigned long long key = ((unsigned long long)__float_as_uint(sah) << 32) | ((axis << 5) | b);   // sah >= 0: its bit pattern is order preserving
      if (key < bestKey) bestKey = key;
    }
  }
  unsigned long long k = bestKey;
  for (int o = 32; o >= 1; o >>= 1) { const unsigned long long other = __shfl_xor(k, o, 64); k = other < k ? other : k; }
  if (lane == 0) { res->sah = __builtin_inff(); res->dim = -1; res->pos = 0; res->nL = 0; }
  if (k != ~0ull) {                                                // (wave-uniform)
    const uint32_t axis = (uint32_t)(k >> 5) & 3u, pos = (uint32_t)k & 31u;
    float lo[3], hi[3]; uint32_t n;
    for (int d = 0; d < 3; d++) { lo[d] = axis == 0u ? klo[0][d] : (axis == 1u ? klo[1][d] : klo[2][d]); hi[d] = axis == 0u ? khi[0][d] : (axis == 1u ? khi[1][d] : khi[2][d]); }
    n = axis == 0u ? kn[0] : (axis == 1u ? kn[1] : kn[2]);
    if (lane == pos) { res->sah = __uint_as_float((uint32_t)(k >> 32)); res->dim = (int)axis; res->pos = (int)pos; }
    if (lane == pos - 1u) { res->nL = n; for (int d = 0; d < 3; d++) { res->llo[d] = lo[d]; res->lhi[d] = hi[d]; } }
    if (lane == 63u - pos) { for (int d = 0; d < 3; d++) { res->rlo[d] = lo[d]; res->rhi[d] = hi[d]; } }
  }
}
