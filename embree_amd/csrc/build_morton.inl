// build_morton.inl -- RTC_BUILD_QUALITY_LOW: Morton-code build.
// Part of build.hip (included inside its anonymous namespace); see the header of build.hip for the pipeline.
// ---------------------------------------------------------------------------- fast build (RTC_BUILD_QUALITY_LOW)
// The reference answers RTC_BUILD_QUALITY_LOW with its Morton builder (kernels/builders/bvh_builder_morton.h:  63-bit codes of the
// centroids, radix sort, recursive splits at the highest differing bit; selected per mesh by the two-level builder, kernels/bvh/
// bvh_builder_twolevel.cpp, kernels/common/scene.cpp:195-206).  The GPU formulation of the same tree: sort the 63-bit codes, then
// every internal node finds its own range and split from the codes alone (Karras 2012: the split of a range is where the common
// prefix of the codes is shortest; ties between equal codes are broken by the index), and the boxes are propagated from the leaves
// with one atomic flag per node.  The result is a binary tree in the BNode format, so the wide collapse, quantisation and leaf
// layout are the ones of the SAH build; only the decisions differ.  Leaf j is BNode (n-1)+j, internal node i is BNode i, root = 0.
__device__ __forceinline__ unsigned long long spread21(uint32_t v) {   // 21 bits -> every third bit
  unsigned long long x = v & 0x1FFFFFull;
  x = (x | x << 32) & 0x1F00000000FFFFull; x = (x | x << 16) & 0x1F0000FF0000FFull; x = (x | x << 8) & 0x100F00F00F00F00Full;
  x = (x | x << 4) & 0x10C30C30C30C30C3ull; x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
// (morton_keys: build_sort.inl -- it also counts the histogram of the sort's first digit)
__global__ __launch_bounds__(256) void morton_gather(const PrimRef* src, const uint32_t* order, uint32_t n, PrimRef* dst, uint2* finalIds) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const PrimRef p = load_prim(src + order[i]);
  store_prim(dst + i, p);
  finalIds[i] = make_uint2(p.geom & 0x07FFFFFFu, p.prim);
}
// length of the common prefix of the (code, index) pairs i and j; -1 outside the array
__device__ __forceinline__ int lbvh_delta(const unsigned long long* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const unsigned long long a = keys[i], b = keys[j];
  return a != b ? __clzll((long long)(a ^ b)) : 64 + __clz(i ^ j);
}
__global__ __launch_bounds__(256) void lbvh_hierarchy(const unsigned long long* keys, uint32_t n, BNode* bnodes, uint32_t* parent) {
  const int i = (int)(blockIdx.x * 256u + threadIdx.x), N = (int)n;
  if (i >= N - 1) return;
  const int d = lbvh_delta(keys, N, i, i + 1) - lbvh_delta(keys, N, i, i - 1) >= 0 ? 1 : -1;
  const int dmin = lbvh_delta(keys, N, i, i - d);
  int lmax = 2;
  while (lbvh_delta(keys, N, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1) if (lbvh_delta(keys, N, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = lbvh_delta(keys, N, i, j);
  int sft = 0;
  for (int t = (l + 1) >> 1; ; t = (t + 1) >> 1) { if (lbvh_delta(keys, N, i, i + (sft + t) * d) > dnode) sft += t; if (t == 1) break; }
  const int gamma = i + sft * d + min(d, 0);
  const int first = min(i, j), last = max(i, j);
  const uint32_t left = gamma == first ? (uint32_t)(N - 1 + gamma) : (uint32_t)gamma;
  const uint32_t right = gamma + 1 == last ? (uint32_t)(N - 1 + gamma + 1) : (uint32_t)(gamma + 1);
  ((uint4*)(bnodes + i))[2] = make_uint4(left, right, __float_as_uint(__builtin_inff()), 0u);   // splitSah = inf: <= max_leaf triangles always form a leaf slot
  ((uint32_t*)(bnodes + i))[3] = (uint32_t)first; ((uint32_t*)(bnodes + i))[7] = (uint32_t)last + 1u;
  parent[left] = (uint32_t)i; parent[right] = (uint32_t)i;
}
// Boxes from the leaves up: the second child to arrive at a node (atomic flag) merges the two child boxes and goes on.  The
// two children are usually processed by different CUs, often on different XCDs, whose L2s are not coherent with each other: the
// boxes are therefore written and read with system-scope (sc0 sc1) 16-byte accesses, which go through to memory, and a store is made to
// complete (s_waitcnt vmcnt(0)) before the flag is touched -- the "sc0 sc1 on both sides" hand-off of MI355X_MICROARCH.md; a
// __threadfence() per step would write back the whole L2 each time (microseconds) and a plain load may return a stale line.
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_sys(void* p, v4f v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory"); }
// the two 16-byte halves (lo|begin, hi|end) of two BNodes, system scope, one wait for the four loads
__device__ __forceinline__ void ld_boxes_sys(const BNode* x, const BNode* y, v4f& xl, v4f& xh, v4f& yl, v4f& yh) {
  asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc0 sc1\n\t"
               "global_load_dwordx4 %2, %5, off sc0 sc1\n\tglobal_load_dwordx4 %3, %5, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(xl), "=&v"(xh), "=&v"(yl), "=&v"(yh) : "v"(x), "v"(y) : "memory");
}
// (round 6) The first levels stay inside the workgroup.  The leaves are in code order, so an internal node whose range lies within the 256 leaves of one workgroup has
// both children there as well, and its index (it is one end of its own range) is one of the workgroup's 256: the two children meet at an LDS flag and hand their boxes
// over in LDS -- no device atomic, no system-scope access, and the second to arrive carries the merged box on in registers.  Only a node whose range crosses a workgroup
// boundary takes the global protocol above (its children's boxes are stored system-scope when their threads leave the local phase): ~1 node in 100.  The kernel was the
// largest of a LOW commit, 859 us of 3.1 ms for 4.76 M triangles (profiles/r06_bench_kernel_stats.md), every node two system-scope stores, four loads and an atomic.
// Boxes are min / max: the tree does not change by a bit.
__global__ __launch_bounds__(256) void lbvh_bounds(const PrimRef* prims, uint32_t n, BNode* bnodes, const uint32_t* parent, uint32_t* flags, Counters* ctr) {
  __shared__ uint32_t s_flag[256];
  __shared__ float s_cb[256][2][6];                             // per local internal node: the box each child brought
  const uint32_t tid = threadIdx.x, bs = blockIdx.x * 256u, be = min(bs + 256u, n), j = bs + tid;
  s_flag[tid] = 0u;
  __syncthreads();
  if (j >= n) return;
  const PrimRef p = load_prim(prims + j);
  uint32_t id = n - 1u + j, w3 = j, w7 = j + 1u;                 // the node this thread carries, its range words
  float lo[3] = {p.lo[0], p.lo[1], p.lo[2]}, hi[3] = {p.hi[0], p.hi[1], p.hi[2]};
  ((float4*)(bnodes + id))[0] = make_float4(lo[0], lo[1], lo[2], __uint_as_float(w3));
  ((float4*)(bnodes + id))[1] = make_float4(hi[0], hi[1], hi[2], __uint_as_float(w7));
  ((uint4*)(bnodes + id))[2] = make_uint4(NIL, NIL, __float_as_uint(__builtin_inff()), 0u);   // links: only read by later kernels
  if (j == 0u) ctr->stripe[0].numBLeaves = n;
  bool local = true;
  while (id != 0u) {
    const uint32_t par = parent[id];
    const uint32_t* pw = (const uint32_t*)(bnodes + par);       // links and range: written by lbvh_hierarchy, never changed here
    const uint32_t l = pw[8], r = pw[9], first = pw[3], end = pw[7];
    if (local && first >= bs && end <= be) {                    // both children are this workgroup's: meet in LDS
      const uint32_t k = par - bs, side = l == id ? 0u : 1u;
      for (int d = 0; d < 3; d++) { s_cb[k][side][d] = lo[d]; s_cb[k][side][3 + d] = hi[d]; }
      __threadfence_block();
      if (atomicAdd(&s_flag[k], 1u) == 0u) return;              // first to arrive: the sibling's thread takes over
      __threadfence_block();
      for (int d = 0; d < 3; d++) { lo[d] = fminf(lo[d], s_cb[k][side ^ 1u][d]); hi[d] = fmaxf(hi[d], s_cb[k][side ^ 1u][3 + d]); }
      w3 = first; w7 = end;
      ((float4*)(bnodes + par))[0] = make_float4(lo[0], lo[1], lo[2], __uint_as_float(w3));
      ((float4*)(bnodes + par))[1] = make_float4(hi[0], hi[1], hi[2], __uint_as_float(w7));
    } else {
      if (local) {                                              // leaving the workgroup: what I carry must be in memory for whoever merges it
        v4f a = {lo[0], lo[1], lo[2], __uint_as_float(w3)}, b = {hi[0], hi[1], hi[2], __uint_as_float(w7)};
        st16_sys(bnodes + id, a); st16_sys((char*)(bnodes + id) + 16, b);
        local = false;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my box is in memory before my arrival is announced
      if (atomicAdd(&flags[par], 1u) == 0u) return;             // first to arrive: the sibling's thread takes over
      v4f al, ah, bl, bh;
      ld_boxes_sys(bnodes + l, bnodes + r, al, ah, bl, bh);
      lo[0] = fminf(al.x, bl.x); lo[1] = fminf(al.y, bl.y); lo[2] = fminf(al.z, bl.z); hi[0] = fmaxf(ah.x, bh.x); hi[1] = fmaxf(ah.y, bh.y); hi[2] = fmaxf(ah.z, bh.z);
      w3 = first; w7 = end;
      v4f a = {lo[0], lo[1], lo[2], __uint_as_float(w3)}, b = {hi[0], hi[1], hi[2], __uint_as_float(w7)};
      st16_sys(bnodes + par, a); st16_sys((char*)(bnodes + par) + 16, b);
    }
    id = par;
  }
}
