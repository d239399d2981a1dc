// build_collapse.inl -- which binary nodes become the children of an 8-wide node: a dynamic programme over the binary tree.
// Part of build.hip (included inside its anonymous namespace); see the header of build.hip for the pipeline.
//
// The reference forms a wide node greedily: split the child with the largest half-area until there are 8 (BuilderT::recurse, kernels/builders/
// bvh_builder_sah.h:247-272), then decides leaf-vs-split by SAH per child.  On this machine a node visit costs the same five 16-byte loads and ~200 VALU
// instructions whether two or eight of its slots are used, and with leaves of <= 3 triangles the greedy rule leaves 37 % of the slots of the crown stand-in
// empty (695 k nodes of 5.06 children for 4.76 M triangles: the sub-trees a node is rooted at have whatever size the top-down recursion left).  Which cut
// of the binary tree a wide node takes is therefore decided by cost, bottom-up, the way Ylitie, Karras and Laine collapse a binary tree into compressed
// wide nodes ("Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", HPG 2017, section 4.1): for every binary node n and every
// i = 1..7, C(n, i) = the cheapest way to represent the sub-tree of n as a forest of at most i wide-tree roots (inner nodes or leaf slots),
//     C(n, 1) = min( leaf(n), A_n * c_node + D(n, 8) ),   leaf(n) = A_n * P_n * c_tri if P_n <= max_leaf, else inf
//     C(n, i) = min( D(n, i), C(n, i - 1) ),              D(n, j) = min over 0 < k < j of C(left, k) + C(right, j - k)
// with A = half area, P = triangles.  8 floats per binary node: C(n, 1..7) and a word that says whether C(n, 1) is the leaf.  The collapse (wide_plan,
// build_wide.inl) then walks the decisions top-down: the children of a wide node rooted at n are what D(n, 8) distributes.
//
// The table is filled from the binary leaves up in ONE launch: a thread per triangle position starts at the binary leaf that begins there, and at every
// parent the second child to arrive (an atomic counter) combines the two child tables and goes on -- the scheme of lbvh_bounds (build_morton.inl), with the
// same system-scope accesses: the two children usually finish on different XCDs, whose L2s are not coherent with each other.
__device__ __forceinline__ void ld_dp_sys(const float* x, const float* y, v4f& x0, v4f& x1, v4f& y0, v4f& y1) {
  asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc0 sc1\n\t"
               "global_load_dwordx4 %2, %5, off sc0 sc1\n\tglobal_load_dwordx4 %3, %5, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(x0), "=&v"(x1), "=&v"(y0), "=&v"(y1) : "v"(x), "v"(y) : "memory");
}
constexpr uint32_t DP_LEAF = 1u;                               // word 7 of a table: C(n, 1) is the leaf

__global__ __launch_bounds__(256) void collapse_dp(const uint32_t* leafNode, uint32_t lbvhN, const BNode* bnodes, const uint32_t* parent, uint32_t* flags, float* dp,
                                                   Params prm, uint32_t maxB, uint32_t limit) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= limit) return;
  uint32_t id = leafNode ? leafNode[j] : lbvhN - 1u + j;        // SAH builds: the binary leaf that begins at position j (NIL elsewhere: the array is cleared to NIL before every build); Morton build: one leaf per triangle
  if (id >= maxB) return;
  {
    const BNode b = load_bnode(bnodes + id);
    const float A = bnode_area(b);
    const float c = prm.dpTri * (A * (float)(b.end - b.begin));   // a binary leaf is never split: it is a leaf slot whatever its size
    v4f t0 = {c, c, c, c}, t1 = {c, c, c, __uint_as_float(DP_LEAF)};
    st16_sys(dp + 8ull * id, t0); st16_sys(dp + 8ull * id + 4, t1);
  }
  for (uint32_t step = 0; id != 0u && step < 4096u; step++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // my table is in memory before my arrival is announced
    const uint32_t par = parent[id];
    if (par >= maxB) return;                                    // (only on a tree whose top phase did not finish: that commit is repeated on the stepwise path)
    if (atomicAdd(&flags[par], 1u) == 0u) return;               // first to arrive: the sibling's thread takes over
    const BNode b = load_bnode(bnodes + par);                   // box, range and links were written by earlier kernels
    if (b.left >= maxB || b.right >= maxB) return;
    v4f l0, l1, r0, r1;
    ld_dp_sys(dp + 8ull * b.left, dp + 8ull * b.right, l0, l1, r0, r1);
    const float L[8] = {0.0f, l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z}, R[8] = {0.0f, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z};
    float D[9];
#pragma unroll
    for (int jj = 2; jj <= 8; jj++) {
      float m = __builtin_inff();
#pragma unroll
      for (int k = 1; k < jj; k++) if (k <= 7 && jj - k <= 7) m = fminf(m, L[k] + R[jj - k]);
      D[jj] = m;
    }
    const float A = bnode_area(b);
    const uint32_t P = b.end - b.begin;
    const float cInner = fmaf(A, prm.dpNode, D[8]);
    const float cLeaf = P <= prm.maxLeaf ? prm.dpTri * (A * (float)P) : __builtin_inff();
    float C[8];
    C[1] = fminf(cLeaf, cInner);
#pragma unroll
    for (int i = 2; i <= 7; i++) C[i] = fminf(D[i], C[i - 1]);
    v4f t0 = {C[1], C[2], C[3], C[4]}, t1 = {C[5], C[6], C[7], __uint_as_float(cLeaf <= cInner ? DP_LEAF : 0u)};
    st16_sys(dp + 8ull * par, t0); st16_sys(dp + 8ull * par + 4, t1);
    id = par;
  }
}

// ---- top-down side, used by wide_plan.  Tables are read with plain loads: they were written by an earlier kernel.
__device__ __forceinline__ bool dp_is_leaf(const float* dp, uint32_t node) { return (__float_as_uint(dp[8ull * node + 7]) & DP_LEAF) != 0u; }
// the cheapest way to hand `budget` (2..8) roots to the two children of a binary node: k for the left child (1..7), D = its cost
__device__ __forceinline__ uint32_t dp_distribute(const float* dp, uint32_t l, uint32_t r, uint32_t budget, float& D) {
  const float* tl = dp + 8ull * l; const float* tr = dp + 8ull * r;
  float best = __builtin_inff(); uint32_t bk = 1u;
  for (uint32_t k = 1u; k < budget; k++) {
    if (k > 7u || budget - k > 7u) continue;
    const float v = tl[k - 1u] + tr[budget - k - 1u];
    if (v < best) { best = v; bk = k; }
  }
  D = best;
  return bk;
}
