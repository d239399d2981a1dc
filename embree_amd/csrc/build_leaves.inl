// build_leaves.inl -- K5: leaf records; refit; node rebasing for instanced scenes.
// Part of build.hip (included inside its anonymous namespace); see the header of build.hip for the pipeline.
// --------------------------------------------------------------------------------- K5 tri_records
// (round 6) Four records per thread -- record k * 256 + i of a tile of 1024 is thread i's -- with the ids, then the geometry entries and indices, then the vertices of all
// four in flight together: one record per thread was a chain of four dependent round trips (id, geometry table, indices, vertices), 130 us for 228 MB written.
// (round 6) copyBlocks != 0: the FIRST copyBlocks workgroups copy the finished node array out of the build arena into the tree's own buffer, which the host sized from what
// the last commit of this kind needed -- the copy used to wait for the host to learn the node count (a round trip of ~30 us with the GPU idle, then a 24 us copy behind this
// kernel); now it runs beside the record gathers, which leave the memory pipes idle.  More nodes than the buffer holds: nothing is copied, the host copies as before.
__global__ __launch_bounds__(256) void tri_records(const uint2* finalIds, uint32_t n, const GeomDesc* geoms, TriRec* out, uint32_t robust, const Counters* ctr,
                                                   const uint4* nodeSrc, uint4* nodeDst, uint32_t nodeCap, uint32_t copyBlocks) {
  if (blockIdx.x < copyBlocks) {
    const uint32_t numNodes = ctr->numWide;
    if (numNodes > nodeCap) return;
    const uint32_t words = numNodes * 5u, stride = copyBlocks * 256u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < words; i += 4u * stride) {   // (four 16-byte loads in flight)
      uint4 x[4];
#pragma unroll
      for (uint32_t k = 0; k < 4u; k++) if (i + k * stride < words) x[k] = nodeSrc[i + k * stride];
#pragma unroll
      for (uint32_t k = 0; k < 4u; k++) if (i + k * stride < words) nodeDst[i + k * stride] = x[k];
    }
    return;
  }
  const uint32_t block = blockIdx.x - copyBlocks;
  // ctr: the grid is an upper bound, the number of leaf records is on the device: what the wide collapse has numbered so far.  Of a tree deeper than the levels
  // enqueued with this launch that is a part only -- the ids behind it are whatever the arena held (a device fault, found when tree buffers began to be
  // recycled); the host runs the remaining levels and launches this kernel again
  const uint32_t count = ctr ? ctr->numTrisOut : n;
  const uint32_t base = block * 1024u + threadIdx.x;
  if (block * 1024u >= count) return;
  uint2 id[4]; bool on[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { const uint32_t i = base + (uint32_t)k * 256u; on[k] = i < count; id[k] = make_uint2(0u, 0u); if (on[k]) id[k] = finalIds[i]; }
  uint32_t i0[4], i1[4], i2[4], pid[4], vs[4], gid[4], msk[4]; const char* vb[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    i0[k] = i1[k] = i2[k] = 0u; pid[k] = 0u; vs[k] = 0u; gid[k] = 0u; msk[k] = 0u; vb[k] = nullptr;
    if (on[k]) { const GeomDesc g = geoms[id[k].x]; prim_indices(g, id[k].y, i0[k], i1[k], i2[k], pid[k]); vs[k] = g.vstride; vb[k] = g.verts; gid[k] = g.geomID; msk[k] = g.mask; }
  }
  float a[4][3], b[4][3], c[4][3];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    for (int d = 0; d < 3; d++) a[k][d] = b[k][d] = c[k][d] = 0.0f;
    if (on[k]) {
      const float* pa = (const float*)(vb[k] + (size_t)i0[k] * vs[k]);
      const float* pb = (const float*)(vb[k] + (size_t)i1[k] * vs[k]);
      const float* pc = (const float*)(vb[k] + (size_t)i2[k] * vs[k]);
      for (int d = 0; d < 3; d++) { a[k][d] = pa[d]; b[k][d] = pb[d]; c[k][d] = pc[d]; }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (!on[k]) continue;
    float4* o = (float4*)(out + base + (uint32_t)k * 256u);
    // pid: quads: quad index, bit 31 = second half (cleared again when a hit is written)
    if (robust) {   // TriangleMv: the three vertices (kernels/geometry/trianglev.h), same 48-byte record
      o[0] = make_float4(a[k][0], a[k][1], a[k][2], b[k][0]);
      o[1] = make_float4(b[k][1], b[k][2], c[k][0], c[k][1]);
      o[2] = make_float4(c[k][2], __uint_as_float(pid[k]), __uint_as_float(gid[k]), __uint_as_float(msk[k]));
    } else {        // TriangleM ctor: e1 = v0 - v1, e2 = v2 - v0 (kernels/geometry/triangle.h:40-41)
      o[0] = make_float4(a[k][0], a[k][1], a[k][2], a[k][0] - b[k][0]);
      o[1] = make_float4(a[k][1] - b[k][1], a[k][2] - b[k][2], c[k][0] - a[k][0], c[k][1] - a[k][1]);
      o[2] = make_float4(c[k][2] - a[k][2], __uint_as_float(pid[k]), __uint_as_float(gid[k]), __uint_as_float(msk[k]));
    }
  }
}

// --------------------------------------------------------------------------------- refit (RTC_BUILD_QUALITY_REFIT, kernels/bvh/bvh_refit.cpp)
// The topology of the tree stays; the triangle records are rewritten from the moved vertices (tri_records) and the boxes are
// recomputed bottom-up, one launch per level of the wide tree (nodes are numbered breadth first, so a level is a contiguous range).
// Eight lanes per node as in wide_emit: lane = child slot; a leaf slot bounds its <= 3 triangles from the vertex buffers (the same
// min/max as primref_gen), an inner slot takes the exact box its child wrote one launch earlier; the node is re-quantised with the
// builder's own routine.  A triangle that has become invalid (non-finite / huge coordinate) raises *flag: the caller rebuilds.
__global__ __launch_bounds__(64) void refit_level(CNode* nodes, float4* boxes, const uint2* ids, const GeomDesc* geoms, uint32_t first, uint32_t count, uint32_t* flag) {
  __shared__ uint32_t s_node[8][20];
  const uint32_t lane = threadIdx.x, s = lane & 7u, g = lane >> 3;
  for (uint32_t base = blockIdx.x * 8u; base < count; base += gridDim.x * 8u) {
    const uint32_t t = base + g; const bool valid = t < count;
    const uint32_t node = first + (valid ? t : 0u);
    __syncthreads();
    if (valid && s < 5u) ((uint4*)&s_node[g][0])[s] = ((const uint4*)(nodes + node))[s];
    __syncthreads();
    const uint8_t* nbr = (const uint8_t*)&s_node[g][0];
    const uint32_t meta = valid ? nbr[24 + s] : 0u, imask = s_node[g][3] >> 24, childBase = s_node[g][4], triBase = s_node[g][5];
    const bool has = meta != 0u, inner = has && ((imask >> s) & 1u) != 0u;
    float lo[3], hi[3], olo[3], ohi[3];
    for (int d = 0; d < 3; d++) { lo[d] = __builtin_inff(); hi[d] = -__builtin_inff(); }
    if (inner) {
      const uint32_t c = childBase + (uint32_t)__popc(imask & ((1u << s) - 1u));
      const float4 a = boxes[2u * c], b = boxes[2u * c + 1u];
      lo[0] = a.x; lo[1] = a.y; lo[2] = a.z; hi[0] = b.x; hi[1] = b.y; hi[2] = b.z;
    } else if (has) {
      const uint32_t cnt = (uint32_t)__popc(meta >> 5), t0 = triBase + (meta & 31u);
      bool ok = true;
      for (uint32_t k = 0; k < cnt; k++) {
        const uint2 id = ids[t0 + k];
        const GeomDesc gd = geoms[id.x];
        uint32_t i0, i1, i2, pid;
        prim_indices(gd, id.y, i0, i1, i2, pid);
        const float* a = (const float*)(gd.verts + (size_t)i0 * gd.vstride);
        const float* b = (const float*)(gd.verts + (size_t)i1 * gd.vstride);
        const float* c = (const float*)(gd.verts + (size_t)i2 * gd.vstride);
        for (int d = 0; d < 3; d++) {
          const float x = a[d], y = b[d], z = c[d];
          ok = ok && valid_f(x) && valid_f(y) && valid_f(z);
          lo[d] = fminf(lo[d], fminf(fminf(x, y), z)); hi[d] = fmaxf(hi[d], fmaxf(fmaxf(x, y), z));
        }
        if (gd.quad) {
          const uint32_t* q = (const uint32_t*)(gd.idx + (size_t)(id.y >> 1) * gd.istride);
          const float* o4 = (const float*)(gd.verts + (size_t)((id.y & 1u) ? q[0] : q[2]) * gd.vstride);
          ok = ok && valid_f(o4[0]) && valid_f(o4[1]) && valid_f(o4[2]);
        }
      }
      if (!ok) { atomicOr(flag, 1u); for (int d = 0; d < 3; d++) { lo[d] = 0.0f; hi[d] = 0.0f; } }
    }
    for (int d = 0; d < 3; d++) { olo[d] = grp_min(lo[d]); ohi[d] = grp_max(hi[d]); }
    uint32_t ex[3], qa[3], qb[3];
    quantise_slots(has, lane, lo, hi, olo, ohi, ex, qa, qb);
    __syncthreads();
    uint8_t* nb = (uint8_t*)&s_node[g][0];
    for (int d = 0; d < 3; d++) { nb[32 + d * 8 + s] = (uint8_t)qa[d]; nb[56 + d * 8 + s] = (uint8_t)qb[d]; }
    if (s == 0u) {
      s_node[g][0] = __float_as_uint(olo[0]); s_node[g][1] = __float_as_uint(olo[1]); s_node[g][2] = __float_as_uint(olo[2]);
      s_node[g][3] = ex[0] | (ex[1] << 8) | (ex[2] << 16) | (imask << 24);
      if (valid) { boxes[2u * node] = make_float4(olo[0], olo[1], olo[2], 0.0f); boxes[2u * node + 1u] = make_float4(ohi[0], ohi[1], ohi[2], 0.0f); }
    }
    __syncthreads();
    if (valid && s < 5u) ((uint4*)(nodes + node))[s] = ((const uint4*)&s_node[g][0])[s];
  }
}

// --------------------------------------------------------------------------------- instanced scenes: object trees copied behind the top tree
// One thread per node: the 80 bytes are copied, child / triangle base indices moved by the object's offset in the combined arrays.
__global__ __launch_bounds__(256) void rebase_nodes(const uint4* src, uint4* dst, uint32_t n, uint32_t nodeOfs, uint32_t triOfs) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  uint4 w0 = src[5u * i], w1 = src[5u * i + 1u];
  w1.x += nodeOfs; w1.y += triOfs;
  dst[5u * i] = w0; dst[5u * i + 1u] = w1; dst[5u * i + 2u] = src[5u * i + 2u]; dst[5u * i + 3u] = src[5u * i + 3u]; dst[5u * i + 4u] = src[5u * i + 4u];
}
