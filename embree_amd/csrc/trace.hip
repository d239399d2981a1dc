// trace.hip -- closest-hit / any-hit traversal kernels for gfx950 (CDNA4, wave64).
//
// Replaces, on the GPU, the reference's single-ray hot loop
//   BVHNIntersector1<8,BVH_AN1,false,ArrayIntersector1<TriangleMIntersector1Moeller<4,true>>>
//     ::intersect   kernels/bvh/bvh_intersector1.cpp:32-114
//     ::occluded    kernels/bvh/bvh_intersector1.cpp:117-197
// with its pieces intersectNode<8> (kernels/bvh/node_intersector1.h:484-531), traverseClosestHit
// (kernels/bvh/bvh_traverser1.h:311-433), MoellerTrumboreIntersector1 (kernels/geometry/
// triangle_intersector_moeller.h:69-111) and Intersect1EpilogM / Occluded1EpilogM
// (kernels/geometry/intersector_epilog.h:235-368).
//
// MI355X mapping -- "one ray per octet":
//   The reference runs one ray across the 8 lanes of an AVX register (8 children tested at once,
//   4-8 triangles at once).  A 64-lane wavefront is 8 such units: lanes 8k..8k+7 form an *octet*
//   that owns one ray; lane j tests child j of the 8-wide node / triangle j of the leaf.
//     * node fetch  = each octet reads ONE 128-byte line (16 B header broadcast + 12 B per lane):
//                     coalesced, 1 line per ray-step instead of 64 divergent lines per instruction
//     * child order = each hit lane ranks its entry distance against the other 7 with DPP
//                     quad_perm / row_half_mirror moves (no LDS, no sorting network, no branches);
//                     rank 0 is descended, ranks 1.. are scattered to the stack in one ds_write
//     * stack       = per-octet, 32 entries in LDS (+96 spill entries in HBM), popped with a
//                     broadcast ds_read; entries carry the entry distance for culling
//     * leaves      = up to 8 triangles tested in parallel, octet-min picks the winner
//     * persistent threads: a wave pulls chunks of rays from a global counter and re-fills
//                     finished octets immediately (ballot + popcount), so a long ray stalls 7
//                     neighbours at most, never 63
//   All arithmetic is fp32; the triangle test keeps the reference's FMA pattern (compiled with
//   -ffp-contract=off so only the explicit fmaf()s fuse).  MFMA is not used: there is no dense
//   contraction anywhere on this path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bvh_common.h"
#include "../../include/embree_amd_hip.h"
#include "internal.h"

namespace {

constexpr int BLOCK = 256;                 // 4 waves
constexpr int OCT_PER_BLOCK = BLOCK / 8;   // 32 rays in flight per block
constexpr int STACK_LDS = 32;              // entries per octet kept in LDS
constexpr int STACK_ROW = STACK_LDS + 1;   // +1 entry pad: octets pop different banks
constexpr int STACK_GLB = 96;              // spill entries per octet in HBM
constexpr int CHUNK = 16;                  // rays a wave reserves per atomic (2 per octet)
constexpr uint32_t DONE = 0xFFFFFFFEu;     // octet has no current ray
constexpr uint32_t ITER_CAP = 1u << 24;

// ---- DPP moves inside an 8-lane octet -------------------------------------------------------
#define QUAD_PERM(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
constexpr int DPP_X1 = QUAD_PERM(1, 0, 3, 2);  // lane ^ 1
constexpr int DPP_X2 = QUAD_PERM(2, 3, 0, 1);  // lane ^ 2
constexpr int DPP_X3 = QUAD_PERM(3, 2, 1, 0);  // lane ^ 3
constexpr int DPP_HM = 0x141;                  // row_half_mirror: lane -> 7 - lane (within 8)

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __uint_as_float(dpp_u<CTRL>(__float_as_uint(v))); }

__device__ __forceinline__ float oct_min(float v) {
  v = fminf(v, dpp_f<DPP_X1>(v));
  v = fminf(v, dpp_f<DPP_X2>(v));
  v = fminf(v, dpp_f<DPP_HM>(v));
  return v;
}
// number of the other 7 lanes of the octet whose key is smaller than mine
__device__ __forceinline__ uint32_t oct_rank(uint32_t key) {
  const uint32_t k7 = dpp_u<DPP_HM>(key);
  uint32_t r = 0;
  r += dpp_u<DPP_X1>(key) < key;
  r += dpp_u<DPP_X2>(key) < key;
  r += dpp_u<DPP_X3>(key) < key;
  r += k7 < key;
  r += dpp_u<DPP_X1>(k7) < key;
  r += dpp_u<DPP_X2>(k7) < key;
  r += dpp_u<DPP_X3>(k7) < key;
  return r;
}

__device__ __forceinline__ float rcp_nr(float a) {  // v_rcp_f32 + one Newton step (reference: RCPPS + Newton, vfloat4_sse2.h:304)
  float r = __builtin_amdgcn_rcpf(a);
  return fmaf(r, fmaf(-a, r, 1.0f), r);
}
__device__ __forceinline__ float xor_sign(float a, uint32_t s) { return __uint_as_float(__float_as_uint(a) ^ s); }

struct TraceArgs {
  const QNode* nodes;
  const TriRec* tris;
  uint32_t root;
  char* rays;            // AoS records
  uint32_t count;
  uint32_t stride;
  uint32_t* counter;     // global ray cursor (zeroed before launch)
  uint2* spill;          // [gridDim.x * OCT_PER_BLOCK][STACK_GLB]
  unsigned long long* stats;  // optional counters
};

template <bool ANY, bool STATS>
__global__ __launch_bounds__(BLOCK) void trace_kernel(TraceArgs a) {
  __shared__ uint2 s_stack[OCT_PER_BLOCK * STACK_ROW];

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t sub = tid & 7u;          // child / triangle slot of this lane
  const uint32_t oct = tid >> 3;          // octet within the block
  const uint32_t octShift = lane & ~7u;   // position of this octet's byte in a 64-bit ballot
  uint2* const myStack = s_stack + oct * STACK_ROW;
  uint2* const mySpill = a.spill + ((size_t)blockIdx.x * OCT_PER_BLOCK + oct) * STACK_GLB;

  // wave-uniform ray chunk
  uint32_t chunkNext = 0, chunkEnd = 0;
  bool exhausted = false;

  // per-octet state (replicated in its 8 lanes)
  uint32_t cur = DONE, sp = 0, rayIdx = 0, rmask = 0;
  float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, rdx = 0, rdy = 0, rdz = 0;
  float tnear = 0, tnearTrav = 0, tfar = 0;
  bool retired = false, haveHit = false;
  uint32_t winLane = 0;
  // per-lane record of the last hit this lane committed
  float hNgx = 0, hNgy = 0, hNgz = 0, hu = 0, hv = 0;
  uint32_t hprim = 0, hgeom = 0;
  // statistics (per lane, only sub==0 counts octet events)
  uint32_t stNodes = 0, stLeaves = 0, stTris = 0, stRays = 0, stSpill = 0, stDepth = 0;

  // the iteration cap is a safety net only (a corrupt tree must not hang the GPU); ~1e3-1e5 iterations are normal
  for (uint32_t iter = 0; iter < ITER_CAP; iter++) {
    // ------------------------------------------------------------------ refill idle octets (all 64 lanes are active here)
    {
      bool idle = (cur == DONE) && !retired;
      unsigned long long idleMask = __ballot(idle && sub == 0);
      while (idleMask != 0ull) {
        if (chunkNext >= chunkEnd) {
          if (exhausted) break;
          uint32_t base = 0;
          if (lane == 0u) base = atomicAdd(a.counter, (uint32_t)CHUNK);
          base = __builtin_amdgcn_readfirstlane(base);
          if (base >= a.count) { exhausted = true; break; }
          chunkNext = base;
          chunkEnd = min(base + (uint32_t)CHUNK, a.count);
        }
        const uint32_t avail = chunkEnd - chunkNext;
        const uint32_t myRank = (uint32_t)__popcll(idleMask & ((1ull << octShift) - 1ull));
        if (idle && myRank < avail) {
          rayIdx = chunkNext + myRank;
          const float4* rp = (const float4*)(a.rays + (size_t)rayIdx * a.stride);
          const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
          ox = r0.x; oy = r0.y; oz = r0.z; tnear = r0.w;
          dx = r1.x; dy = r1.y; dz = r1.z;
          tfar = r2.x; rmask = __float_as_uint(r2.y);
          // TravRay: rdir = rcp_safe(dir) (|d| < 1e-18 -> +1e-18), tnear/tfar clamped to >= 0 for traversal
          // kernels/bvh/node_intersector1.h:29-57, common/math/vec3fa.h:167-172, bvh_intersector1.cpp:65
          rdx = rcp_nr(fabsf(dx) < 1e-18f ? 1e-18f : dx);
          rdy = rcp_nr(fabsf(dy) < 1e-18f ? 1e-18f : dy);
          rdz = rcp_nr(fabsf(dz) < 1e-18f ? 1e-18f : dz);
          tnearTrav = fmaxf(tnear, 0.0f);
          sp = 0; haveHit = false; winLane = 0;
          cur = a.root;
          if (a.root == MI355_EMPTY_REF) cur = DONE;           // empty scene: nothing is written
          if (ANY && tfar < 0.0f) cur = DONE;                   // already occluded, bvh_intersector1.cpp:128
          if (STATS && sub == 0) stRays++;
          idle = (cur == DONE);                                 // a ray that finished instantly frees the octet again
        }
        chunkNext += min(avail, (uint32_t)__popcll(idleMask));
        idleMask = __ballot(idle && sub == 0);
      }
      if (idle) retired = true;
    }
    if (__ballot(!retired) == 0ull) break;

    bool needPop = false;
    if (!retired && cur != DONE) {
      if (!mi355_is_leaf(cur)) {
        // -------------------------------------------------------------- inner node: 8 children, one per lane
        const char* np = (const char*)(a.nodes + cur);
        const float4 hdr = *(const float4*)np;                       // org.xyz, exps (same address in all 8 lanes)
        const uint32_t* cp = (const uint32_t*)(np + 16 + 12 * sub);  // my child: 12 B
        const uint32_t w0 = cp[0], w1 = cp[1], cref = cp[2];
        const uint32_t ex = __float_as_uint(hdr.w);
        const float sx = __uint_as_float((ex & 0xFFu) << 23);
        const float sy = __uint_as_float(((ex >> 8) & 0xFFu) << 23);
        const float sz = __uint_as_float(((ex >> 16) & 0xFFu) << 23);
        // t(q) = (org + q*s - O) * rdir = q*(s*rdir) + (org-O)*rdir
        const float ax = sx * rdx, ay = sy * rdy, az = sz * rdz;
        const float bx = (hdr.x - ox) * rdx, by = (hdr.y - oy) * rdy, bz = (hdr.z - oz) * rdz;
        const float t0x = fmaf((float)(w0 & 0xFFu), ax, bx), t1x = fmaf((float)(w0 >> 24), ax, bx);
        const float t0y = fmaf((float)((w0 >> 8) & 0xFFu), ay, by), t1y = fmaf((float)(w1 & 0xFFu), ay, by);
        const float t0z = fmaf((float)((w0 >> 16) & 0xFFu), az, bz), t1z = fmaf((float)((w1 >> 8) & 0xFFu), az, bz);
        const float tN = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tnearTrav));
        const float tF = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), fmaxf(tfar, 0.0f)));
        const bool hit = (tN <= tF) && (cref != MI355_EMPTY_REF);
        if (STATS && sub == 0) stNodes++;
        const uint32_t hitBits = (uint32_t)(__ballot(hit) >> octShift) & 0xFFu;
        if (hitBits == 0u) {
          needPop = true;
        } else {
          // order hit children by entry distance (ties: lower slot first); the low 3 bits of the key are
          // the slot, so keys are unique and <= the true distance (still a valid culling bound)
          const uint32_t key = hit ? ((__float_as_uint(tN) & ~7u) | sub) : 0xFFFFFFFFu;
          const uint32_t rank = oct_rank(key);
          const uint32_t nhit = (uint32_t)__popc(hitBits);
          // nearest child next; the others go to the stack, farthest deepest (traverseClosestHit,
          // kernels/bvh/bvh_traverser1.h:311-433; any-hit keeps the same order instead of index order)
          if (hit && rank != 0u) {
            const uint32_t slot = sp + (nhit - 1u - rank);
            const uint2 e = make_uint2(cref, key);
            if (slot < (uint32_t)STACK_LDS) myStack[slot] = e;
            else if (slot < (uint32_t)(STACK_LDS + STACK_GLB)) { mySpill[slot - STACK_LDS] = e; if (STATS) stSpill++; }
          }
          // broadcast the nearest child's ref: it is the hit lane with rank 0
          const uint32_t first = (uint32_t)__builtin_ctz((uint32_t)(__ballot(hit && rank == 0u) >> octShift) & 0xFFu);
          cur = __shfl(cref, (int)(octShift + first), 64);
          sp = min(sp + nhit - 1u, (uint32_t)(STACK_LDS + STACK_GLB));
          if (STATS) stDepth = max(stDepth, sp);
        }
      } else {
        // -------------------------------------------------------------- leaf: up to 8 triangles per round
        const uint32_t first = mi355_leaf_first(cur), cnt = mi355_leaf_count(cur);
        if (STATS && sub == 0) stLeaves++;
        bool occluded = false;
        for (uint32_t base = 0; base < cnt; base += 8u) {
          const uint32_t j = base + sub;
          const bool tv = j < cnt;
          const float4* tp = (const float4*)(a.tris + first + (tv ? j : 0u));
          const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
          if (STATS && tv) stTris++;
          const float v0x = q0.x, v0y = q0.y, v0z = q0.z;
          const float e1x = q0.w, e1y = q1.x, e1z = q1.y;
          const float e2x = q1.z, e2y = q1.w, e2z = q2.x;
          const uint32_t tprim = __float_as_uint(q2.y), tgeom = __float_as_uint(q2.z), tmask = __float_as_uint(q2.w);
          // Moeller-Trumbore, same operation order and FMA placement as the reference
          // (triangle_intersector_moeller.h:79-108; cross/dot: common/math/vec3.h:204,209)
          const float Ngx = fmaf(e2y, e1z, -(e2z * e1y));
          const float Ngy = fmaf(e2z, e1x, -(e2x * e1z));
          const float Ngz = fmaf(e2x, e1y, -(e2y * e1x));
          const float Cx = v0x - ox, Cy = v0y - oy, Cz = v0z - oz;
          const float Rx = fmaf(Cy, dz, -(Cz * dy));
          const float Ry = fmaf(Cz, dx, -(Cx * dz));
          const float Rz = fmaf(Cx, dy, -(Cy * dx));
          const float den = fmaf(Ngx, dx, fmaf(Ngy, dy, Ngz * dz));
          const float absDen = fabsf(den);
          const uint32_t sgn = __float_as_uint(den) & 0x80000000u;
          const float U = xor_sign(fmaf(Rx, e2x, fmaf(Ry, e2y, Rz * e2z)), sgn);
          const float V = xor_sign(fmaf(Rx, e1x, fmaf(Ry, e1y, Rz * e1z)), sgn);
          const float T = xor_sign(fmaf(Ngx, Cx, fmaf(Ngy, Cy, Ngz * Cz)), sgn);
          bool ok = tv && (den != 0.0f) && (U >= 0.0f) && (V >= 0.0f) && (U + V <= absDen);
          ok = ok && (absDen * tnear < T) && (T <= absDen * tfar);    // strict at tnear, inclusive at tfar
          ok = ok && ((tmask & rmask) != 0u);                            // EMBREE_RAY_MASK, intersector_epilog.h:256-262
          if (ANY) {
            if ((__ballot(ok) >> octShift) & 0xFFull) { occluded = true; break; }
          } else {
            const float rcpd = rcp_nr(absDen);
            const float t = T * rcpd;
            const float tc = ok ? t : __builtin_inff();
            const float tmin = oct_min(tc);
            const uint32_t winBits = (uint32_t)(__ballot(ok && tc == tmin) >> octShift) & 0xFFu;
            if (winBits != 0u) {                                         // select_min: lowest lane among the minimum
              const uint32_t w = (uint32_t)__builtin_ctz(winBits);
              tfar = tmin; winLane = w; haveHit = true;
              if (sub == w) { hNgx = Ngx; hNgy = Ngy; hNgz = Ngz; hu = U * rcpd; hv = V * rcpd; hprim = tprim; hgeom = tgeom; }
            }
          }
        }
        if (ANY && occluded) {
          if (sub == 0) *(float*)(a.rays + (size_t)rayIdx * a.stride + 32) = -__builtin_inff();
          cur = DONE;
        } else {
          needPop = true;
        }
      }

      // ---------------------------------------------------------------- pop (with distance culling)
      while (needPop) {
        if (sp == 0u) {
          // ray finished: the lane that committed the last hit writes the record
          if (!ANY && haveHit && sub == winLane) {
            char* rp = a.rays + (size_t)rayIdx * a.stride;
            *(float*)(rp + 32) = tfar;
            *(float4*)(rp + 48) = make_float4(hNgx, hNgy, hNgz, hu);
            *(uint4*)(rp + 64) = make_uint4(__float_as_uint(hv), hprim, hgeom, MI355_EMPTY_REF);
            *(uint32_t*)(rp + 80) = MI355_EMPTY_REF;
          }
          cur = DONE;
          needPop = false;
        } else {
          sp--;
          const uint2 e = (sp < (uint32_t)STACK_LDS) ? myStack[sp] : mySpill[sp - STACK_LDS];
          if (ANY || !(__uint_as_float(e.y) > tfar)) { cur = e.x; needPop = false; }   // pop skips dist > ray.tfar (:79)
        }
      }
    }
  }

  if (STATS) {
    atomicAdd(&a.stats[0], (unsigned long long)stNodes);
    atomicAdd(&a.stats[1], (unsigned long long)stLeaves);
    atomicAdd(&a.stats[2], (unsigned long long)stTris);
    atomicAdd(&a.stats[3], (unsigned long long)stRays);
    atomicAdd(&a.stats[4], (unsigned long long)stSpill);
    atomicMax(&a.stats[5], (unsigned long long)stDepth);
  }
}

// ---- packet adaptor: SoA RTCRayHitK / RTCRayK <-> the AoS records the trace kernels consume ----
// (RayHitK::get/set kernels/common/ray.h:283-376; packet calls never touch lanes whose valid[i] != -1)
struct PacketArgs { const int* valid; char* packets; uint32_t K, numPackets; size_t packetStride; char* aos; uint32_t* index; uint32_t* numActive; };

__global__ void packet_gather(PacketArgs p, int withHit) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.K * p.numPackets) return;
  const uint32_t pk = i / p.K, k = i % p.K;
  if (p.valid && p.valid[i] != -1) return;
  const uint32_t slot = atomicAdd(p.numActive, 1u);
  p.index[slot] = i;
  const uint32_t* src = (const uint32_t*)(p.packets + (size_t)pk * p.packetStride);
  uint32_t* dst = (uint32_t*)(p.aos + (size_t)slot * (withHit ? 96 : 48));
  for (int f = 0; f < 12; f++) dst[f] = src[f * p.K + k];
  if (withHit) for (int f = 0; f < 9; f++) dst[12 + f] = src[(12 + f) * p.K + k];
}
__global__ void packet_scatter(PacketArgs p, int withHit) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= *p.numActive) return;
  const uint32_t i = p.index[slot];
  const uint32_t pk = i / p.K, k = i % p.K;
  uint32_t* dst = (uint32_t*)(p.packets + (size_t)pk * p.packetStride);
  const uint32_t* src = (const uint32_t*)(p.aos + (size_t)slot * (withHit ? 96 : 48));
  dst[8 * p.K + k] = src[8];   // tfar
  if (withHit && src[12 + 6] != MI355_EMPTY_REF)
    for (int f = 0; f < 9; f++) dst[(12 + f) * p.K + k] = src[12 + f];
}

}  // namespace

// ------------------------------------------------------------------------------------- host side
namespace mi355 {

static int launch_trace(Bvh* b, void* d_rays, uint32_t count, size_t stride, bool any, hipStream_t s, uint64_t* statsOut,
                        hipEvent_t evStart = nullptr, hipEvent_t evStop = nullptr) {
  if (count == 0) return 0;
  if (stride < (any ? 48u : 96u) || (stride & 15u) || ((uintptr_t)d_rays & 15u)) return set_error(hipErrorInvalidValue, "ray array must be 16-byte aligned with a 16-byte-multiple stride");
  HIP_TRY(hipSetDevice(b->device));
  // persistent grid: enough blocks to fill the chip (8 blocks of 256 threads per CU), never more than the rays need
  const uint32_t maxBlocks = (uint32_t)b->numCUs * 8u;
  uint32_t blocks = (count + OCT_PER_BLOCK - 1) / OCT_PER_BLOCK;
  if (blocks > maxBlocks) blocks = maxBlocks;
  TraceScratch* sc = b->scratch_for(s);
  if (!sc) return set_error(hipErrorOutOfMemory, "trace scratch allocation failed");
  HIP_TRY(hipMemsetAsync(sc->counter, 0, sizeof(uint32_t), s));
  TraceArgs a;
  a.nodes = (const QNode*)b->d_nodes; a.tris = (const TriRec*)b->d_tris; a.root = b->root;
  a.rays = (char*)d_rays; a.count = count; a.stride = (uint32_t)stride;
  a.counter = sc->counter; a.spill = (uint2*)sc->spill; a.stats = nullptr;
  if (statsOut) {
    HIP_TRY(hipMemsetAsync(sc->stats, 0, 8 * sizeof(uint64_t), s));
    a.stats = (unsigned long long*)sc->stats;
    if (any) hipLaunchKernelGGL((trace_kernel<true, true>), dim3(blocks), dim3(BLOCK), 0, s, a);
    else     hipLaunchKernelGGL((trace_kernel<false, true>), dim3(blocks), dim3(BLOCK), 0, s, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(statsOut, sc->stats, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
  }
  if (evStart) HIP_TRY(hipEventRecord(evStart, s));
  if (any) hipLaunchKernelGGL((trace_kernel<true, false>), dim3(blocks), dim3(BLOCK), 0, s, a);
  else     hipLaunchKernelGGL((trace_kernel<false, false>), dim3(blocks), dim3(BLOCK), 0, s, a);
  if (evStop) HIP_TRY(hipEventRecord(evStop, s));
  HIP_TRY(hipGetLastError());
  return 0;
}

size_t trace_spill_bytes(int numCUs) { return (size_t)numCUs * 8u * OCT_PER_BLOCK * STACK_GLB * sizeof(uint2); }

static int launch_packets(Bvh* b, const int* d_valid, void* d_pk, uint32_t K, uint32_t n, size_t pstride, bool any, hipStream_t s) {
  if (n == 0) return 0;
  if (K != 4 && K != 8 && K != 16) return set_error(hipErrorInvalidValue, "packet size must be 4, 8 or 16");
  HIP_TRY(hipSetDevice(b->device));
  const size_t rec = any ? 48 : 96, total = (size_t)K * n;
  char* aos = nullptr; uint32_t* index = nullptr; uint32_t* nact = nullptr;
  HIP_TRY(hipMallocAsync((void**)&aos, total * rec, s));
  HIP_TRY(hipMallocAsync((void**)&index, total * 4 + 16, s));
  nact = index + total;
  HIP_TRY(hipMemsetAsync(nact, 0, 4, s));
  PacketArgs p{d_valid, (char*)d_pk, K, n, pstride, aos, index, nact};
  const uint32_t g = (uint32_t)((total + 255) / 256);
  hipLaunchKernelGGL(packet_gather, dim3(g), dim3(256), 0, s, p, any ? 0 : 1);
  uint32_t active = 0;
  HIP_TRY(hipMemcpyAsync(&active, nact, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  int rc = launch_trace(b, aos, active, rec, any, s, nullptr);
  if (rc == 0 && active) hipLaunchKernelGGL(packet_scatter, dim3(g), dim3(256), 0, s, p, any ? 0 : 1);
  hipFreeAsync(aos, s); hipFreeAsync(index, s);
  return rc;
}

}  // namespace mi355

extern "C" {
int mi355_trace_closest(mi355_bvh_t bvh, void* d, uint32_t n, size_t stride, void* stream) {
  return mi355::launch_trace((mi355::Bvh*)bvh, d, n, stride, false, (hipStream_t)stream, nullptr);
}
int mi355_trace_any(mi355_bvh_t bvh, void* d, uint32_t n, size_t stride, void* stream) {
  return mi355::launch_trace((mi355::Bvh*)bvh, d, n, stride, true, (hipStream_t)stream, nullptr);
}
int mi355_trace_timed(mi355_bvh_t bvh, void* d, uint32_t n, size_t stride, int any_hit, void* stream, void* ev_start, void* ev_stop) {
  return mi355::launch_trace((mi355::Bvh*)bvh, d, n, stride, any_hit != 0, (hipStream_t)stream, nullptr, (hipEvent_t)ev_start, (hipEvent_t)ev_stop);
}
int mi355_trace_stats(mi355_bvh_t bvh, void* d, uint32_t n, size_t stride, int any_hit, uint64_t out[8]) {
  for (int i = 0; i < 8; i++) out[i] = 0;
  return mi355::launch_trace((mi355::Bvh*)bvh, d, n, stride, any_hit != 0, nullptr, out);
}
int mi355_trace_closest_packet(mi355_bvh_t bvh, const int* v, void* d, uint32_t K, uint32_t n, size_t ps, void* stream) {
  return mi355::launch_packets((mi355::Bvh*)bvh, v, d, K, n, ps, false, (hipStream_t)stream);
}
int mi355_trace_any_packet(mi355_bvh_t bvh, const int* v, void* d, uint32_t K, uint32_t n, size_t ps, void* stream) {
  return mi355::launch_packets((mi355::Bvh*)bvh, v, d, K, n, ps, true, (hipStream_t)stream);
}
}
