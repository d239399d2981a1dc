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
// MI355X mapping -- one ray per lane, persistent waves, triangle tests pooled per wave:
//   * a wave owns 64 rays; every lane walks its own ray through the 8-wide compressed tree of bvh_common.h.
//     A node visit is five 16-byte loads per lane (80-byte node) and ~200 VALU instructions for the wave
//     (8 slab tests on dequantised planes: v_cvt_f32_ubyte per plane, v_pk_fma_f32 per plane pair, v_max3/v_min3 per child);
//     a triangle visit is three 16-byte loads and the reference's Moeller-Trumbore arithmetic.
//   * traversal order needs no sort: children are stored in the slot matching their octant, so the hit bits of
//     a node, XOR-ed with the ray's octant, are already front to back.  The hits of one node are a 32-bit word
//     (8 inner-node bits, 24 triangle bits); the per-lane stack holds {child base, hit word} pairs = one entry
//     per tree level at most, 9 entries per lane in LDS ([entry][lane] so that lanes never bank-conflict),
//     deeper levels spill to HBM (sized from the depth the builder reports, so it can never overflow).
//   * the kernel is VALU-issue bound (profiles/r01_pmc_trace.md), so what matters is how many lanes each wave
//     instruction serves.  Triangle tests therefore do not run in the lane that found them: triangle bits go
//     to a per-wave LDS ring and are tested 64 at a time by all lanes (see trace_kernel_q below).
//   * persistent threads: rays are handed out in blocks of REFILL_MIN through 8 cache-line-separated cursors; one wave per
//     workgroup (waves never synchronise), so a CU slot frees as soon as one wave is done.
//   * tail: once the cursors are dry, lanes without a ray take pending sub-trees of the rays that are left (step 1b).
//   All arithmetic is fp32; the triangle test keeps the reference's operation order and FMA placement (compiled
//   with -ffp-contract=off so only the explicit fmaf()s fuse).  MFMA is not used: no dense contraction here.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <utility>
#include <vector>
#include "bvh_common.h"
#include "../../include/embree_amd_hip.h"
#include "internal.h"

namespace {

#ifndef MI355_TRACE_BLOCK
#define MI355_TRACE_BLOCK 64
#endif
constexpr int BLOCK = MI355_TRACE_BLOCK;   // one wave per workgroup (waves never synchronise): a CU slot frees as soon as ONE wave is done, which lets the next batch in earlier (+2.7 % with 4 batches in flight vs 256)
constexpr int MAX_BLOCKS_PER_CU = MI355_MAX_BLOCKS_PER_CU * 256 / BLOCK;
constexpr uint32_t ITER_CAP = 1u << 24;    // safety net only: a corrupt tree must not hang the GPU (env MI355_TRACE_ITER_CAP lowers it for the test of the flag below)
// Neither safety net may fail silently: a wave that runs into the iteration cap, or a lane whose stack would outgrow its spill area, raises a word
// in host-visible memory (TraceScratch::status); the blocking entry points turn it into RTC_ERROR_UNKNOWN, mi355_trace_status() reads it for device-pointer callers.
constexpr uint32_t STATUS_ITER_CAP = 0, STATUS_SPILL = 1, STATUS_COHERENT = 2;   // words of TraceScratch::status (64 bytes of host-mapped memory); COHERENT: 1 = the last packet sample kept its packets, 2 = it did not
constexpr uint32_t REFILL_MIN_DEFAULT = 16;  // rays are handed out in blocks of this many, once that many lanes are free (env MI355_REFILL_MIN); 32 before finished rays went to the done queue

__device__ __forceinline__ float rcp_nr(float a) {  // v_rcp_f32 + one Newton step (reference: RCPPS + Newton, vfloat4_sse2.h:304)
  float r = __builtin_amdgcn_rcpf(a);
  return fmaf(r, fmaf(-a, r, 1.0f), r);
}
// set bits of a lane mask below this lane: v_mbcnt_lo + v_mbcnt_hi (the compiler does not find them in popcount(m & ((1 << lane) - 1)): two v_and + two v_bcnt and two registers for the constant)
__device__ __forceinline__ uint32_t rank_below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__device__ __forceinline__ float xor_sign(float a, uint32_t s) { return __uint_as_float(__float_as_uint(a) ^ s); }
#ifndef MI355_NO_UNDEF                                          /* (-DMI355_NO_UNDEF: the registers are zeroed instead -- a variant for the bisection of profiles/r06_device_filter.md) */
#define MI355_UNDEF4(n) asm volatile("" : "=v"(n.x), "=v"(n.y), "=v"(n.z), "=v"(n.w))
#define MI355_UNDEF2(n) asm volatile("" : "=v"(n.x), "=v"(n.y))
#else
#define MI355_UNDEF4(n) do { n.x = 0; n.y = 0; n.z = 0; n.w = 0; } while (0)
#define MI355_UNDEF2(n) do { n.x = 0; n.y = 0; } while (0)
#endif
// "all four words of this load are used": the plain kernels read only e2.z and the mask from a triangle record's third 16 bytes, and the compiler then fetches them as TWO
// one-word loads -- four (lane, load) pairs per triangle test instead of three, on a kernel that runs into the CU's address path once its instruction count is down (round 5:
// 352 -> 306 lane-accesses per ray, profiles/r05_trace.md)
#define MI355_KEEP4(n) asm volatile("" : "+v"(n.x), "+v"(n.y), "+v"(n.z), "+v"(n.w))
template <int J> __device__ __forceinline__ float ubyte(uint32_t w) { return (float)((w >> (8 * J)) & 0xFFu); }   // v_cvt_f32_ubyteJ

struct TraceArgs {
  const uint4* nodes;    // CNode[] as 5 x uint4
  const float4* tris;    // TriRec[] as 3 x float4
  uint32_t hasRoot;
  char* rays;            // AoS records
  uint32_t count;
  uint32_t stride;
  uint32_t* counter;     // global ray cursor (zeroed before launch)
  uint2* spill;          // [gridDim.x * BLOCK][spillPerLane]
  uint32_t spillPerLane;
  uint32_t refillMin, pushRounds, numCursors, drainWaiters;   // refillMin and numCursors are powers of two (gShift, cShift: their logarithms): the hand-out arithmetic is shifts and 32-bit adds
  uint32_t gShift, cShift;
  uint32_t iterCap, helpers;
  uint32_t staticRays;   // 0: rays are handed out through the cursors.  R (a power of two <= 64): a SMALL batch -- no more rays than lane slots in the grid -- wave w owns rays [w R, (w + 1) R): no cursor, no
                         // atomic, no reserve-ahead; lanes R .. 63 (and every lane whose ray is done) are tail helpers from the first iteration on (step 1b)
  volatile uint32_t* status;  // host-mapped: [STATUS_ITER_CAP], [STATUS_SPILL] set to 1 when a safety net dropped work
  unsigned long long* stats;  // optional counters
  uint32_t* touch; uint32_t touchTriWord;   // counting build only: one bit per node (from word 0) and per triangle record (from word touchTriWord), set when it is fetched: the UNIQUE bytes a launch needs
  const float4* insts;   // INST kernels: InstRec[] as 4 x float4 (world2local vx,vy,vz,p | root node, instID, mask, flags)
  const uint4* rules;    // device-side filter rules, 48 B per geometry (+ bit arrays behind them), or nullptr
  unsigned long long filterFn; void* filterCtx; uint32_t filterEnforce;   // FILT == 2: address of a __device__ filter function (RTCIntersectArguments::filter of a *Device query), its context, RTC_RAY_QUERY_FLAG_INVOKE_ARGUMENT_FILTER
  const uint32_t* deferList; const uint32_t* deferCount;   // second pass of a RTC_RAY_QUERY_FLAG_COHERENT query: the packets (64 consecutive rays each) the packet kernel gave up on; nullptr otherwise
};

// slab test of the 4 children whose quantised planes sit in one dword per plane; returns their contribution to the hit word.
// The reference's fast node test compares exact fp32 planes, t = plane * rdir - org * rdir (intersectNode<8>, node_intersector1.h:484-531).  Here a plane
// distance is q * (scale * rdir) + (org_node - org_ray) * rdir: the plane itself is never rounded (quantise_slots keeps the EXACT plane outside the
// child's geometry), but the two coefficients are, by up to 2^-22 (255 |a| + |b|) together with the FMA's own rounding and the 2 ulp of rcp_nr.  That
// bound is taken off every near distance and added to every far distance (bn* / bf* below), so the test never loses a box that the exact arithmetic
// accepts -- a ray through the exact corner of a box (cube corner of tutorials/triangle_geometry), a 40-unit pipe triangle seen from inside its node,
// a scene 1e5 away from the origin.  Conservative = never wrong, only (very slightly) slower.  Measured on the 12.7 M triangle powerplant stand-in:
// the reference's own fast mode loses 55 of 2^20 hits that its robust mode finds; this test loses none (tests/test_gpu_round2.py).
typedef float f2 __attribute__((ext_vector_type(2)));   // v_pk_fma_f32: two fp32 FMAs per VALU issue on gfx950
struct SlabCoef { f2 sxy, szx, syz, bxy, bzx, byz; };  // plane distance = q * scale + base, paired (near x, near y) (near z, far x) (far y, far z)
// (round 5) Lane masks of the node step in SGPR pairs, selects in the VOP3 form, the hit word's shifts as SDWA instructions that pick their bytes themselves:
// the compiler spent v_bfe + v_lshrrev + v_lshlrev per child on what v_lshlrev_b32_sdwa does in one (205 -> 199 VALU instructions per node step, +0.6 %).
// (The VOP2 form of v_cndmask with its implicit VCC measures 23 cycles in tools/valu_bench.hip when nothing writes VCC -- and 4.3 like every other form behind a
// v_cmp, which is how the node step uses it: profiles/r05_valu_issue_costs.txt.  No gain from avoiding VCC; the SGPR pairs only keep VCC free for the scheduler.)
#ifndef MI355_SEL_SGPR
#define MI355_SEL_SGPR 1
#endif
typedef unsigned long long lanemask_t;
__device__ __forceinline__ lanemask_t cmp_neg_s(float a) { lanemask_t m; asm("v_cmp_gt_f32_e64 %0, 0, %1" : "=s"(m) : "v"(a)); return m; }   // a < 0
__device__ __forceinline__ lanemask_t cmp_le_s(float a, float b) { lanemask_t m; asm("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ __forceinline__ uint32_t sel_s(uint32_t ifClear, uint32_t ifSet, lanemask_t m) { uint32_t r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(ifClear), "v"(ifSet), "s"(m)); return r; }
__device__ __forceinline__ uint32_t sel0_s(uint32_t ifSet, lanemask_t m) { uint32_t r; asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(ifSet), "s"(m)); return r; }
// byte J of `bits` shifted left by the low five bits of byte J of `index`
#define MI355_SHL_BYTES(J, DST, INDEX, BITS) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #J " src1_sel:BYTE_" #J : "=v"(DST) : "v"(INDEX), "v"(BITS))
#if MI355_SEL_SGPR
#define MI355_CHILD_HIT(J) { uint32_t sh; MI355_SHL_BYTES(J, sh, bitIndex4, childBits4); hits |= sel0_s(sh, cmp_le_s(tN, tF)); }
#else
#define MI355_CHILD_HIT(J) { const uint32_t cb = (childBits4 >> (8 * J)) & 0xFFu, bi = (bitIndex4 >> (8 * J)) & 0x1Fu; hits |= (tN <= tF) ? (cb << bi) : 0u; }
#endif
__device__ __forceinline__ uint32_t test4(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t fx, uint32_t fy, uint32_t fz, uint32_t meta4,
                                          uint32_t octinv4, const SlabCoef& k, float tmin0, float tmax0) {
  // meta byte: inner = 001 11sss (bits 3 and 4 set), leaf = ccc ooooo with offset <= 23, empty = 0
  const uint32_t isInner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
  const uint32_t innerMask4 = (isInner4 >> 1) - (isInner4 >> 4);      // 0x07 in the bytes of inner slots (0x08 - 0x01 per byte: v_mul_lo_u32 is a quarter-rate instruction)
  const uint32_t bitIndex4 = meta4 ^ (octinv4 & innerMask4);          // low 5 bits: position in the hit word
  const uint32_t childBits4 = (meta4 >> 5) & 0x07070707u;             // inner: 1, leaf: unary triangle count, empty: 0
  uint32_t hits = 0;
#define MI355_CHILD(J)                                                                                         \
  {                                                                                                            \
    f2 a, b, c;                                                                                                \
    a.x = ubyte<J>(nx); a.y = ubyte<J>(ny); b.x = ubyte<J>(nz); b.y = ubyte<J>(fx); c.x = ubyte<J>(fy); c.y = ubyte<J>(fz); \
    a = __builtin_elementwise_fma(a, k.sxy, k.bxy);   /* tnx, tny */                                           \
    b = __builtin_elementwise_fma(b, k.szx, k.bzx);   /* tnz, tfx */                                           \
    c = __builtin_elementwise_fma(c, k.syz, k.byz);   /* tfy, tfz */                                           \
    const float tN = fmaxf(fmaxf(a.x, a.y), fmaxf(b.x, tmin0));                                                \
    const float tF = fminf(fminf(b.y, c.x), fminf(c.y, tmax0));                                                \
    MI355_CHILD_HIT(J)                                                                                         \
  }
  MI355_CHILD(0) MI355_CHILD(1) MI355_CHILD(2) MI355_CHILD(3)
#undef MI355_CHILD
  return hits;
}

// RTC_SCENE_FLAG_ROBUST: intersectNodeRobust (kernels/bvh/node_intersector1.h:539-554) = (plane - org) * rdir_near|far with
// rdir_near/far = rdir * (1 -+ 3 ulp) (TravRayBase<N,true>, :98-121).  plane - org is a single correctly rounded operation, so
// the distance has a RELATIVE error of 1.5 ulp and the 3-ulp factors make the test conservative -- which the fast path's
// q * (scale * rdir) + (org_node - org) * rdir is not (its error is relative to the node origin's distance, not to t).  The
// plane itself is decoded with the same fmaf(q, scale, org) the builder used to verify that the quantised box contains the
// child (wide_emit), so decoded planes never cut into the geometry.
__device__ __forceinline__ uint32_t test4_robust(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t fx, uint32_t fy, uint32_t fz, uint32_t meta4, uint32_t octinv4,
                                                 float scx, float scy, float scz, float nox, float noy, float noz, float ox, float oy, float oz,
                                                 float rnx, float rny, float rnz, float rfx, float rfy, float rfz, float tmin0, float tmax0) {
  const uint32_t isInner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
  const uint32_t innerMask4 = (isInner4 >> 1) - (isInner4 >> 4);
  const uint32_t bitIndex4 = meta4 ^ (octinv4 & innerMask4);
  const uint32_t childBits4 = (meta4 >> 5) & 0x07070707u;
  uint32_t hits = 0;
#define MI355_CHILD(J)                                                                                         \
  {                                                                                                            \
    const float tnx = (fmaf(ubyte<J>(nx), scx, nox) - ox) * rnx, tny = (fmaf(ubyte<J>(ny), scy, noy) - oy) * rny, tnz = (fmaf(ubyte<J>(nz), scz, noz) - oz) * rnz; \
    const float tfx = (fmaf(ubyte<J>(fx), scx, nox) - ox) * rfx, tfy = (fmaf(ubyte<J>(fy), scy, noy) - oy) * rfy, tfz = (fmaf(ubyte<J>(fz), scz, noz) - oz) * rfz; \
    const float tN = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin0));                                                \
    const float tF = fminf(fminf(tfx, tfy), fminf(tfz, tmax0));                                                \
    MI355_CHILD_HIT(J)                                                                                         \
  }
  MI355_CHILD(0) MI355_CHILD(1) MI355_CHILD(2) MI355_CHILD(3)
#undef MI355_CHILD
  return hits;
}

// ---- triangle tests.  Record = three float4: fast scenes (v0, e1 = v0-v1, e2 = v2-v0), robust scenes (v0, v1, v2); then primID, geomID, mask.
struct TriOut { float t, u, v, Ngx, Ngy, Ngz; };
// Moeller-Trumbore, same operation order and FMA placement as the reference
// (triangle_intersector_moeller.h:79-108; cross/dot: common/math/vec3.h:204,209); FINISH also produces t,u,v,Ng (Intersect1EpilogM, intersector_epilog.h:235-300)
template <bool FINISH>
__device__ __forceinline__ bool tri_moeller(const float4 q0, const float4 q1, const float4 q2, float ox, float oy, float oz, float dx, float dy, float dz,
                                            float tnear, float tfar, TriOut& o, bool quad2 = false) {
  const float v0x = q0.x, v0y = q0.y, v0z = q0.z;
  const float e1x = q0.w, e1y = q1.x, e1z = q1.y;
  const float e2x = q1.z, e2y = q1.w, e2z = q2.x;
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
  bool ok = (den != 0.0f) && (U >= 0.0f) && (V >= 0.0f) && (U + V <= absDen);
  ok = ok && (absDen * tnear < T) && (T <= absDen * tfar);      // strict at tnear, inclusive at tfar
  if (FINISH || ok) {
    const float rcpd = rcp_nr(absDen);
    o.t = T * rcpd;
    if (FINISH) {
      // second half (v2,v1,v3) of a quad: U,V <- absDen-V, absDen-U and Ng <- -Ng (quad_intersector_moeller.h:205-207, AVX path)
      const float Uq = quad2 ? absDen - V : U, Vq = quad2 ? absDen - U : V, sg = quad2 ? -1.0f : 1.0f;
      o.u = Uq * rcpd; o.v = Vq * rcpd; o.Ngx = Ngx * sg; o.Ngy = Ngy * sg; o.Ngz = Ngz * sg;
    }
  }
  return ok;
}
// modified Pluecker test (triangle_intersector_pluecker.h:68-118), stable_triangle_normal (common/math/vec3.h:210-222),
// PlueckerHitM::finalize (:26-33); watertight along shared edges.  Same operation order as the reference.
template <bool FINISH>
__device__ __forceinline__ bool tri_pluecker(const float4 q0, const float4 q1, const float4 q2, float ox, float oy, float oz, float dx, float dy, float dz,
                                             float tnear, float tfar, TriOut& o, bool quad2 = false) {
  const float v0x = q0.x - ox, v0y = q0.y - oy, v0z = q0.z - oz;
  const float v1x = q0.w - ox, v1y = q1.x - oy, v1z = q1.y - oz;
  const float v2x = q1.z - ox, v2y = q1.w - oy, v2z = q2.x - oz;
  const float e0x = v2x - v0x, e0y = v2y - v0y, e0z = v2z - v0z;
  const float e1x = v0x - v1x, e1y = v0y - v1y, e1z = v0z - v1z;
  const float e2x = v1x - v2x, e2y = v1y - v2y, e2z = v1z - v2z;
#define MI355_EDGE(ex, ey, ez, sx, sy, sz) fmaf(fmaf(ey, sz, -(ez * sy)), dx, fmaf(fmaf(ez, sx, -(ex * sz)), dy, fmaf(ex, sy, -(ey * sx)) * dz))
  const float U = MI355_EDGE(e0x, e0y, e0z, (v2x + v0x), (v2y + v0y), (v2z + v0z));
  const float V = MI355_EDGE(e1x, e1y, e1z, (v0x + v1x), (v0y + v1y), (v0z + v1z));
  const float W = MI355_EDGE(e2x, e2y, e2z, (v1x + v2x), (v1y + v2y), (v1z + v2z));
#undef MI355_EDGE
  const float UVW = (U + V) + W;
  const float eps = 1.1920929e-07f * fabsf(UVW);
  bool ok = (fminf(fminf(U, V), W) >= -eps) || (fmaxf(fmaxf(U, V), W) <= eps);
  const float abx = e0z * e1y, aby = e0x * e1z, abz = e0y * e1x;
  const float bcx = e1z * e2y, bcy = e1x * e2z, bcz = e1y * e2x;
  const float Ngx = fabsf(abx) < fabsf(bcx) ? fmaf(e0y, e1z, -abx) : fmaf(e1y, e2z, -bcx);
  const float Ngy = fabsf(aby) < fabsf(bcy) ? fmaf(e0z, e1x, -aby) : fmaf(e1z, e2x, -bcy);
  const float Ngz = fabsf(abz) < fabsf(bcz) ? fmaf(e0x, e1y, -abz) : fmaf(e1x, e2y, -bcz);
  const float dn = fmaf(Ngx, dx, fmaf(Ngy, dy, Ngz * dz)), den = dn + dn;
  const float tt = fmaf(v0x, Ngx, fmaf(v0y, Ngy, v0z * Ngz)), T = tt + tt;
  const float t = rcp_nr(den) * T;
  ok = ok && (tnear <= t) && (t <= tfar) && (den != 0.0f);      // inclusive at both ends
  o.t = t;
  if (FINISH) {
    const float rcpUVW = fabsf(UVW) < 1e-18f ? 0.0f : rcp_nr(UVW);
    const float u = fminf(U * rcpUVW, 1.0f), v = fminf(V * rcpUVW, 1.0f), sg = quad2 ? -1.0f : 1.0f;
    // second half of a quad: u,v <- 1-v, 1-u and Ng <- -Ng (QuadHitPlueckerM::finalize, quad_intersector_pluecker.h:33-50, AVX path)
    o.u = quad2 ? 1.0f - v : u; o.v = quad2 ? 1.0f - u : v; o.Ngx = sg * Ngx; o.Ngy = sg * Ngy; o.Ngz = sg * Ngz;
  }
  return ok;
}

// ---- device-side filter rules (include/embree4/rtcore.h rtcSetGeometryFilterRule_mi355).  The reference runs filter CALLBACKS inside the traversal for every
// potential hit and goes on when one says no (runIntersectionFilter1 / runOcclusionFilter1, kernels/geometry/filter.h:14-80, called from Intersect1EpilogM /
// Occluded1EpilogM, intersector_epilog.h:235-368).  A host function cannot run in a HIP kernel; what can is a small fixed set of RULES per geometry, evaluated
// where the reference calls the callback: after the triangle test and the mask test accepted a candidate, before it is published.  A rule is a pure function
// of (primID, geomID, t, u, v), so "closest accepted hit" / "any accepted hit" do not depend on the order candidates are met in.  48 bytes per geometry:
//   w0 kinds | apply << 8   w1 modulus   w2 remainder   w3 primID factor | geomID factor << 16     w4-7 tmin tmax umax vmax     w8 bit array offset (words)  w9 bits
constexpr uint32_t RULE_MODULO = 1u, RULE_BITS = 2u, RULE_TWINDOW = 4u, RULE_UV = 8u, RULE_APPLY_INTERSECT = 1u << 8, RULE_APPLY_OCCLUDED = 2u << 8;
constexpr uint32_t RULE_ARG_FILTER = 1u << 16;   // w0 bit 16: rtcSetGeometryEnableFilterFunctionFromArguments (w10, w11: the geometry's user pointer, handed to the filter function)
template <bool ANY, bool ROBUST>
__device__ __forceinline__ bool rule_accepts(const uint4* rules, uint32_t ruleIdx, const float4 q0, const float4 q1, const float4 q2, float t,
                                             float ox, float oy, float oz, float dx, float dy, float dz) {
  const uint4 r0 = rules[(size_t)ruleIdx * 3u];
  if ((r0.x & 0xFFu) == 0u || !(r0.x & (ANY ? RULE_APPLY_OCCLUDED : RULE_APPLY_INTERSECT))) return true;   // (bits 0-7: the rule kinds; bit 16 is RULE_ARG_FILTER)
  const uint4 r1 = rules[(size_t)ruleIdx * 3u + 1u], r2 = rules[(size_t)ruleIdx * 3u + 2u];
  const uint32_t pidRaw = __float_as_uint(q2.y), pid = pidRaw & 0x7FFFFFFFu, gid = __float_as_uint(q2.z);
  bool reject = false;
  if ((r0.x & RULE_MODULO) && r0.y != 0u) reject = ((pid * (r0.w & 0xFFFFu) + gid * (r0.w >> 16)) % r0.y) == r0.z;
  if ((r0.x & RULE_BITS) && pid < r2.y) reject = reject || ((((const uint32_t*)rules)[r2.x + (pid >> 5)] >> (pid & 31u)) & 1u) != 0u;
  if (r0.x & RULE_TWINDOW) reject = reject || !(t >= __uint_as_float(r1.x) && t <= __uint_as_float(r1.y));
  if (r0.x & RULE_UV) {                                         // u, v as the hit record would carry them (second half of a quad included)
    TriOut w;
    if (ROBUST) tri_pluecker<true>(q0, q1, q2, ox, oy, oz, dx, dy, dz, 0.0f, 0.0f, w, (pidRaw >> 31) != 0u);
    else tri_moeller<true>(q0, q1, q2, ox, oy, oz, dx, dy, dz, 0.0f, 0.0f, w, (pidRaw >> 31) != 0u);
    reject = reject || w.u > __uint_as_float(r1.z) || w.v > __uint_as_float(r1.w);
  }
  return !reject;
}

// ---- device filter FUNCTIONS (round 5): the address of a __device__ function in RTCIntersectArguments::filter / RTCOccludedArguments::filter of a *Device query is called
// for every candidate hit of a geometry that enabled it (rtcSetGeometryEnableFilterFunctionFromArguments) or of every geometry (RTC_RAY_QUERY_FLAG_INVOKE_ARGUMENT_FILTER),
// where the reference's GPU path calls its function pointer: runIntersectionFilter1SYCL / runOcclusionFilter1SYCL, kernels/geometry/filter_sycl.h:12-120 -- after the
// geometry's own rules, before the candidate is published.  The callee sees what the reference hands over: RTCFilterFunctionNArguments with N = 1, valid[0] = -1,
// ray.tfar = the candidate's distance, the full hit (Ng, u, v, primID, geomID, instID[0]); clearing valid[0] rejects the candidate.  ABI: include/embree4/rtcore.h,
// INTEGRATION.md (the function lives in the CALLER's code object; both are compiled for gfx950 by the same compiler: one calling convention, one address space).
struct DevRay { float org_x, org_y, org_z, tnear, dir_x, dir_y, dir_z, time, tfar; uint32_t mask, id, flags; };                 // RTCRay (rtcore_ray.h:19-36)
struct DevHit { float Ng_x, Ng_y, Ng_z, u, v; uint32_t primID, geomID, instID0, instPrimID0; };                                  // RTCHit (rtcore_ray.h:39-53), RTC_MAX_INSTANCE_LEVEL_COUNT = 1
struct DevFilterArgs { int* valid; void* geometryUserPtr; void* context; DevRay* ray; DevHit* hit; unsigned int N; };          // RTCFilterFunctionNArguments (rtcore_common.h:318-326)
typedef void (*DevFilterFn)(const DevFilterArgs*);
// (inlined into the kernel ON PURPOSE: as a function of its own, the kernel's register allocation trusted what THIS function is seen to clobber -- inter-procedural register
// allocation -- and kept values in v0-v7 / s0-s3 across it, which the caller's function, compiled elsewhere, is free to overwrite: the robust kernels lost their rays.  With the
// indirect call in the kernel itself the full clobber set of the calling convention applies.)
template <bool ROBUST>
__device__ __forceinline__ bool call_device_filter(unsigned long long fn, void* ctx, void* userPtr, const float4 q0, const float4 q1, const float4 q2,
                                                float ox, float oy, float oz, float dx, float dy, float dz, float tnear, uint32_t mask, uint32_t rayId, uint32_t instID) {
  TriOut w;
  const uint32_t pid = __float_as_uint(q2.y);
  if (ROBUST) tri_pluecker<true>(q0, q1, q2, ox, oy, oz, dx, dy, dz, 0.0f, 0.0f, w, (pid >> 31) != 0u);
  else tri_moeller<true>(q0, q1, q2, ox, oy, oz, dx, dy, dz, 0.0f, 0.0f, w, (pid >> 31) != 0u);
  DevRay ray{ox, oy, oz, tnear, dx, dy, dz, 0.0f, w.t, mask, rayId, 0u};
  DevHit hit{w.Ngx, w.Ngy, w.Ngz, w.u, w.v, pid & 0x7FFFFFFFu, __float_as_uint(q2.z), instID, instID == MI355_EMPTY_REF ? MI355_EMPTY_REF : 0u};
  int valid = -1;
  DevFilterArgs args{&valid, userPtr, ctx, &ray, &hit, 1u};
#ifndef MI355_FPTR_NOCALL                     /* (debugging: everything but the call itself) */
  ((DevFilterFn)fn)(&args);
#endif
  return valid != 0;
}

// =============================================================================================
// Triangle tests leave the lane that found them.
//   If a lane that hits leaf slots had to run its triangle tests itself before opening the next node, the node block and
//   the triangle block would each run with 30-40 % of the lanes (measured: 695 Mrays/s, profiles/r01_trace_history.md).
//   Instead the hit word's triangle bits are expanded into a per-wave LDS ring of {triangle, owner lane} pairs and the lane
//   goes on traversing; whenever 64 pairs are queued the whole wave tests them, one pair per lane, fetching the owner's ray
//   with ds_bpermute.  An accepted hit is published with one 64-bit LDS atomic-min on best[owner] = {t bits : triangle index};
//   owners re-read their best[] entry each iteration (that is their current tfar).  A ray retires once its traversal is
//   done AND the ring has drained past its last pair; u/v/Ng/IDs are recomputed from the winning triangle at that point.
//   Closest hit = minimum t over ALL accepted candidates, ties to the lower triangle index: independent of scheduling.
#ifndef MI355_QSTACK_LDS
#define MI355_QSTACK_LDS 9
#endif
constexpr int QSTACK_LDS = MI355_QSTACK_LDS;   // stack entries per lane in LDS
#ifndef MI355_QCAP
#define MI355_QCAP 256
#endif
constexpr uint32_t QCAP = MI355_QCAP;      // ring capacity (pairs) per wave
constexpr uint32_t PUSH_ROUNDS_DEFAULT = 8;  // triangle bits a lane may queue per iteration (the rest waits one iteration: the lane cannot open a node; env MI355_PUSH_ROUNDS).  5 while every
                                             // bit cost the wave a queueing round; with the scan of step 5 a lane with more bits costs only its own loop passes (lanes blocked 7 % -> 2 %)
constexpr uint32_t NUM_CURSORS = 8;        // ray cursors per launch (one per XCD)
constexpr uint32_t CURSOR_STRIDE = 64;     // words between cursors: each one in its own 256-byte block
constexpr uint32_t EXIT_WORD = NUM_CURSORS * CURSOR_STRIDE;   // behind the cursors: waves of the running launch that have left (the last one zeroes the cursors for the next launch)

#ifndef MI355_TRACE_ATTR
#define MI355_TRACE_ATTR
#endif
#ifndef MI355_TRI_PREFETCH
#define MI355_TRI_PREFETCH 1
#endif
#ifndef MI355_PUSH_SCAN
#define MI355_PUSH_SCAN 1
#endif
constexpr uint32_t NO_INST = 0xFFFFFFFFu;
// InstanceIntersector1 (kernels/geometry/instance_intersector.cpp:26-31): org' = xfmPoint(world2local, org), dir' = xfmVector(world2local, dir),
// nested FMAs exactly like common/math/affinespace.h:102 and linearspace3.h:159; tnear / tfar (and so every t) are unchanged.
__device__ __forceinline__ void xfm_ray(const float4 m0, const float4 m1, const float4 m2, float& ox, float& oy, float& oz, float& dx, float& dy, float& dz) {
  const float px = ox, py = oy, pz = oz, vx = dx, vy = dy, vz = dz;
  ox = fmaf(px, m0.x, fmaf(py, m0.w, fmaf(pz, m1.z, m2.y)));
  oy = fmaf(px, m0.y, fmaf(py, m1.x, fmaf(pz, m1.w, m2.z)));
  oz = fmaf(px, m0.z, fmaf(py, m1.y, fmaf(pz, m2.x, m2.w)));
  dx = fmaf(vx, m0.x, fmaf(vy, m0.w, vz * m1.z));
  dy = fmaf(vx, m0.y, fmaf(vy, m1.x, vz * m1.w));
  dz = fmaf(vx, m0.z, fmaf(vy, m1.y, vz * m2.x));
}
// TravRay: rdir = rcp_safe(dir) (|d| < 1e-18 -> +1e-18)  kernels/bvh/node_intersector1.h:29-57, common/math/vec3fa.h:167-172;
// robust (TravRayBase<N,true>, :98-121): a true division, then 3 ulp down / up.  A ray travelling towards +x meets the children on the
// -x side first: priority of slot s = s ^ octinv.
template <bool ROBUST>
__device__ __forceinline__ void setup_rdir(float dx, float dy, float dz, float& rdx, float& rdy, float& rdz, float& rfx, float& rfy, float& rfz, uint32_t& octinv4) {
  if (ROBUST) {
    const float rx = 1.0f / (fabsf(dx) < 1e-18f ? 1e-18f : dx), ry = 1.0f / (fabsf(dy) < 1e-18f ? 1e-18f : dy), rz = 1.0f / (fabsf(dz) < 1e-18f ? 1e-18f : dz);
    const float down = 1.0f - 3.0f * 1.1920929e-07f, up = 1.0f + 3.0f * 1.1920929e-07f;
    rdx = down * rx; rdy = down * ry; rdz = down * rz; rfx = up * rx; rfy = up * ry; rfz = up * rz;
  } else {
    rdx = rcp_nr(fabsf(dx) < 1e-18f ? 1e-18f : dx);
    rdy = rcp_nr(fabsf(dy) < 1e-18f ? 1e-18f : dy);
    rdz = rcp_nr(fabsf(dz) < 1e-18f ? 1e-18f : dz);
  }
  octinv4 = ((rdx < 0.0f ? 0u : 1u) | (rdy < 0.0f ? 0u : 2u) | (rdz < 0.0f ? 0u : 4u)) * 0x01010101u;
}
// INST (scenes with RTC_GEOMETRY_TYPE_INSTANCE, one level): nodes[] / tris[] hold the top tree over the instances' world boxes followed by the trees of
// the instanced scenes (indices rebased), a.insts the instance records.  A "triangle" of the top tree is an instance: a lane that finds one
// pushes what is left of its node, transforms ITS ray into the instance (the world ray is re-read from the ray array on the way out), and walks
// the object's tree with the same loop; ring pairs are tested with the owner's CURRENT ray, so a lane changes space only after the ring has
// passed its last pair.  The winning triangle's instance is remembered per lane (the key in best[] changed while inside) and the hit is
// recomputed in that instance's space when the ray retires.  Tail helpers (1b) only take sub-trees that lie inside an instance.
// FILT: the scene has device-side filter rules (rule_accepts above).  A template parameter, not a run-time test: the rule code (a second, finishing triangle
// test for the u/v cut-off) costs 6 - 12 VGPRs, which takes the robust kernels from 128 to 134 = from four to three waves per SIMD for every scene.
// FILT == 2: rules + a device filter FUNCTION (call_device_filter above): an indirect call inside the triangle block -- a stack in scratch memory and the register budget of a
// callee the compiler cannot see; only the queries that pass a function pay for it (profiles/r05_device_filter.md).
// SMALL: the static launch shape of small batches (a.staticRays rays per wave, see in front of the loop) is an instantiation of its own: with the pre-started rays in the
// same code the loop of the large-batch kernel came out 23 VALU instructions longer (+4 % wave instructions per launch by the counters, same answers) -- the ray state then
// enters the loop with values the compiler has to carry instead of constants.
template <bool ANY, bool STATS, bool ROBUST, bool INST, int FILT, bool SMALL = false>
__global__ __launch_bounds__(BLOCK, FILT == 2 ? 4 : 1) MI355_TRACE_ATTR void trace_kernel_q(TraceArgs a) {
  __shared__ uint2 s_stack[BLOCK / 64][QSTACK_LDS][64];
  __shared__ __attribute__((aligned(QCAP * 8))) uint2 s_queue[BLOCK / 64][QCAP];   // (aligned to its size: a ring position is one v_and away from its address)
  __shared__ unsigned long long s_best[BLOCK / 64][64];
  __shared__ uint32_t s_pend[BLOCK / 64][64], s_lastT[BLOCK / 64][64];   // per ray slot: helper sub-trees in flight, last ring ticket pushed by helpers
  __shared__ uint2 s_done[BLOCK / 64][INST ? 96 : 64];                    // finished closest-hit rays {ray index, winning triangle}: their hit records are written 64 at a time (flush_done);
                                                                          // INST: + 64 words, the instance the winning triangle lies in (no separate array: 8192 B per wave is the budget of 5 waves / SIMD)

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  uint2* const stk = &s_stack[tid >> 6][0][lane];
  uint2* const queue = &s_queue[tid >> 6][0];
  unsigned long long* const best = &s_best[tid >> 6][0];
  uint32_t* const pend = &s_pend[tid >> 6][0];
  uint32_t* const lastT = &s_lastT[tid >> 6][0];
  uint2* const done = &s_done[tid >> 6][0];
  uint32_t* const doneInst = (uint32_t*)&s_done[tid >> 6][INST ? 64 : 0];   // (only touched by INST kernels)
  uint32_t dCount = 0;                                                    // wave-uniform: entries in the done queue
  // Writes the hit records of the queued rays, one ray per lane: t, u, v, Ng are recomputed from the ray (re-read from the ray array: its lane has moved on)
  // and the winning triangle with the arithmetic of the test in step 4 (Intersect1EpilogM, intersector_epilog.h:235-300).  A finished ray costs its lane two LDS
  // words at retire time; this block runs with >= 75 % of the lanes instead of the 10-50 % that are retiring in any one iteration.
  auto flush_done = [&]() {
    if (lane < dCount) {
      const uint2 e = done[lane];
      char* rp = a.rays + (size_t)e.x * a.stride;
      const float4 r0 = ((const float4*)rp)[0], r1 = ((const float4*)rp)[1];
      const float4* tp = a.tris + (size_t)e.y * 3u;
      const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
      float hx = r0.x, hy = r0.y, hz = r0.z, gx = r1.x, gy = r1.y, gz = r1.z;
      uint32_t hitInst = MI355_EMPTY_REF, hitInstPrim = MI355_EMPTY_REF;
      if (INST) {
        const uint32_t bi = doneInst[lane];
        if (bi != NO_INST) {
          const float4* ip = a.insts + (size_t)bi * 4u;
          const float4 i0 = ip[0], i1 = ip[1], i2 = ip[2], i3 = ip[3];
          if ((__float_as_uint(i3.w) & 1u) == 0u) { xfm_ray(i0, i1, i2, hx, hy, hz, gx, gy, gz); hitInst = __float_as_uint(i3.y); hitInstPrim = 0u; }   // the hit lies in an instance: its space, its id (instPrimID 0: instance_stack.h:19-50)
        }
      }
      TriOut w;
      const uint32_t pid = __float_as_uint(q2.y);            // bit 31: second triangle of a quad (tri_records, build.hip)
      if (ROBUST) tri_pluecker<true>(q0, q1, q2, hx, hy, hz, gx, gy, gz, 0.0f, 0.0f, w, (pid >> 31) != 0u);
      else tri_moeller<true>(q0, q1, q2, hx, hy, hz, gx, gy, gz, 0.0f, 0.0f, w, (pid >> 31) != 0u);
      *(float*)(rp + 32) = w.t;
      *(float4*)(rp + 48) = make_float4(w.Ngx, w.Ngy, w.Ngz, w.u);
      *(uint4*)(rp + 64) = make_uint4(__float_as_uint(w.v), pid & 0x7FFFFFFFu, __float_as_uint(q2.z), hitInst);
      *(uint32_t*)(rp + 80) = hitInstPrim;
    }
    dCount = 0;
  };
  uint2* const spill = a.spill + (size_t)(blockIdx.x * BLOCK + tid) * a.spillPerLane;

  bool active = false, travDone = false, exhausted = false;
  uint32_t owner = lane; bool helper = false, helpersUsed = false;     // tail: a lane without a ray traverses a sub-tree of another lane's ray (see 1b)
  uint32_t rayIdx = 0, rmask = 0, octinv4 = 0, sp = 0, lastTicket = 0;
  uint32_t ngBase = 0, ngHits = 0, tgBase = 0, tgHits = 0;
  uint32_t qHead = 0, qTail = 0;                                         // wave-uniform ring cursors (monotonic)
  uint32_t cursor = blockIdx.x % a.numCursors, dryCursors = 0;            // wave-uniform: which ray cursor this wave pulls from
  uint32_t resV = 0; bool resValid = false;                              // the block reserved ahead (lane 0 holds the atomic's result)
  float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, rdx = 0, rdy = 0, rdz = 0, tnear = 0, tnearTrav = 0, tfar = 0;
  float tfar0 = 0;                                                       // the ray's OWN tfar (tfar follows the best hit): what a triangle candidate is tested against, see step 4
  float rfx = 0, rfy = 0, rfz = 0;                                       // ROBUST: rdir_far (rdx.. hold rdir_near)
  uint32_t inst = NO_INST, topSp = 0, bestInst = NO_INST, entryLo = 0, entryHi = 0;   // INST: instance the lane is in, stack depth at entry, instance of the best hit, best[] key at entry
  uint32_t stNodes = 0, stTris = 0, stRays = 0, stSpill = 0, stDepth = 0, stIter = 0, stNodeBlk = 0, stTriBlk = 0;
  uint32_t stIdle = 0, stWaitBatch = 0, stWaitDrain = 0, stBlocked = 0, stEmpty = 0, stCulled = 0;
  unsigned long long stRefillClk = 0, stLoopClk = 0, stNodeClk = 0; uint32_t stRefillEv = 0;      // STATS: shader clocks (s_memtime) inside the hand-out block / the whole loop / the node step, hand-out events
  const unsigned long long stClk0 = STATS ? __builtin_readcyclecounter() : 0ull;

  // The ray cursors are zero when a launch starts because the LAST wave of the launch before it left them so (a hipMemsetAsync in front of every launch was a
  // second node on the stream per query: rtcIntersect1 pays it per ray).  A wave leaves only when every cursor is dry, so the wave that counts itself out last
  // knows nobody reads them any more.
  auto wave_exit = [&]() {
    // (every cursor atomic of this wave has been PERFORMED before it counts itself out -- also the block reserved ahead whose answer nobody waited for: atomics
    // on different words may reach L2 in any order, and one that lands after the last wave's reset leaves a cursor at 1: the next launch skips 16 rays)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0u && !SMALL) {                                 // (a static launch never touched a cursor: nothing to reset, nobody to count)
      if (atomicAdd(a.counter + EXIT_WORD, 1u) == gridDim.x * (BLOCK / 64u) - 1u) {
        for (uint32_t c = 0; c < NUM_CURSORS; c++) atomicExch(a.counter + c * CURSOR_STRIDE, 0u);
        atomicExch(a.counter + EXIT_WORD, 0u);
      }
    }
  };
  // a lane takes a ray (SMALL kernels, in front of the loop; the same lines as step 1d): TravRay set-up (setup_rdir), the root as "the one hit child of a virtual node", its best[] / helper words
  auto start_ray = [&](const uint32_t newIdx, const float4 r0, const float4 r1, const float4 r2) {
    rayIdx = newIdx;
    ox = r0.x; oy = r0.y; oz = r0.z; tnear = r0.w;
    dx = r1.x; dy = r1.y; dz = r1.z;
    tfar = r2.x; tfar0 = r2.x; rmask = __float_as_uint(r2.y);
    setup_rdir<ROBUST>(dx, dy, dz, rdx, rdy, rdz, rfx, rfy, rfz, octinv4);
    tnearTrav = fmaxf(tnear, 0.0f);                               // tnear/tfar clamped to >= 0 for traversal (bvh_intersector1.cpp:65)
    if (INST) { inst = NO_INST; bestInst = NO_INST; }
    sp = 0; ngBase = 0; ngHits = 0x80000000u; tgBase = 0; tgHits = 0;   // "the root is the one hit child of a virtual node"
    lastTicket = qHead; travDone = false;
    best[lane] = ((unsigned long long)__float_as_uint(tfar) << 32) | 0xFFFFFFFFull;
    pend[lane] = 0u; lastT[lane] = qHead;
    active = a.hasRoot != 0u && !(ANY && tfar < 0.0f);            // empty scene / already occluded (bvh_intersector1.cpp:128)
    if (STATS) stRays++;
  };
  const uint32_t rayCount = a.deferCount ? *a.deferCount * 64u : a.count;   // (wave-uniform) rays to hand out: all of them, or those of the deferred packets
  if (rayCount == 0u) return;                                               // (the pass behind a packet launch none of whose packets gave up: nobody touches a cursor, nobody counts
                                                                            // itself out -- 4096 waves leaving at once through one atomic word cost the Cornell box 40 % of its coherent rate)
  const uint32_t totalBlocks = (rayCount + a.refillMin - 1u) >> a.gShift;      // (wave-uniform) blocks of refillMin rays in this launch
  // SMALL batches (round 6; SURVEY 8e: one 2^20-ray batch over 8 GPUs is 2^17 rays per launch).  A launch with no more rays than the grid has lane slots used to run
  // through the same hand-out: every wave took its 64 rays with one atomic, found the eight cursors dry one atomic round trip after the other (that is when `exhausted`
  // rises and the tail helpers may start), and counted itself out through a ninth -- and with 64 rays per wave half of the SIMDs' wave slots stayed empty.  Such a launch
  // is all ramp and tail, so: the host picks R = rays per wave so that the grid fills every wave slot (launch_trace_locked), wave w owns rays [w R, (w + 1) R), nobody
  // touches a cursor, and the lanes without a ray help from the first iteration on.  Same rays, same candidates, same minimum: the hit records are the bytes of a large batch.
  if (SMALL) {
    const uint32_t idx = blockIdx.x * (BLOCK / 64u) * a.staticRays + (tid >> 6) * a.staticRays + lane;
    if (lane < a.staticRays && idx < rayCount) {
      const float4* rp = (const float4*)(a.rays + (size_t)idx * a.stride);
      start_ray(idx, rp[0], rp[1], rp[2]);
    }
    exhausted = true;
  }
  uint32_t iter = 0;
  for (; iter < a.iterCap; iter++) {
    // ------------------------------------------------------------------ 0. a full batch of queued pairs is waiting: issue the loads of its triangle records now, so that
    // their round trip overlaps steps 1 - 3a instead of being waited for in step 4 (the node loads of 3a already overlap step 4)
    const bool pre = MI355_TRI_PREFETCH && (qTail - qHead) >= 64u;          // wave-uniform
    uint2 pe; float4 pq0, pq1, pq2;
    MI355_UNDEF2(pe); MI355_UNDEF4(pq0); MI355_UNDEF4(pq1); MI355_UNDEF4(pq2);
    if (pre) {
      pe = queue[(qHead + lane) & (QCAP - 1u)];
      const float4* tp = a.tris + (size_t)pe.x * 3u;
      pq0 = tp[0]; pq1 = tp[1]; pq2 = tp[2];                               // (all of pq2: MI355_KEEP4 where it is consumed, step 4 -- here it would wait for the load)
    }
    // ------------------------------------------------------------------ 1. retire finished rays, hand out new ones
    // Ray indices are handed out in blocks of G = refillMin consecutive rays.  Block B belongs to cursor B % numCursors, so
    // neighbouring blocks go to different cursors = different XCDs (a wave pulls from cursor blockIdx % 8 = its XCD): one
    // global atomic word saturates at ~88 dequeues/us (MI355X_MICROARCH.md "dequeue"), eight do not -- provided each
    // cursor sits in its own cache line (eight cursors inside one 32-byte sector measured 30 % SLOWER than one cursor).
    // A wave always holds ONE block reserved ahead of time (the atomic for the next block is issued while the rays of this
    // one are being loaded), and the loads of the retiring rays' winning triangles are issued together with the loads of the
    // new rays, so a refill costs one memory round trip, not three.  A wave whose cursor runs dry moves on to the next
    // cursor; every ray is handed out exactly once; a wave never exits while it holds a block that exists.
    {
      bool retirable = active && !helper && travDone && (int)(qHead - lastTicket) >= 0;
      if (helpersUsed && retirable) retirable = pend[lane] == 0u && (int)(qHead - lastT[lane]) >= 0;   // helpers done and their pairs tested
      const unsigned long long freeMask = __ballot(retirable || !active);
      const bool anyBusy = __ballot(active && !retirable) != 0ull;
      if ((uint32_t)__popcll(freeMask) >= a.refillMin || !anyBusy) {
        const uint32_t G = a.refillMin;
        const unsigned long long stT0 = STATS ? __builtin_readcyclecounter() : 0ull;
        if (STATS && lane == 0u) stRefillEv++;
        // (a) retiring rays: an occluded ray gets its tfar = -inf right away, a closest hit goes to the done queue {ray, winning triangle}
        {
          uint32_t htri = MI355_EMPTY_REF;
          if (retirable) htri = (uint32_t)best[lane];
          const bool hasHit = retirable && htri != MI355_EMPTY_REF;
          if (ANY) { if (hasHit) *(float*)(a.rays + (size_t)rayIdx * a.stride + 32) = -__builtin_inff(); }   // Occluded1EpilogM: tfar = -inf
          else {
            const unsigned long long hm = __ballot(hasHit);
            const uint32_t k = (uint32_t)__popcll(hm);
            if (dCount + k > 64u) flush_done();                  // room for this batch (wave-uniform)
            if (hasHit) {
              const uint32_t pos = dCount + rank_below(hm);
              done[pos] = make_uint2(rayIdx, htri);
              if (INST) doneInst[pos] = bestInst;
            }
            dCount += k;
          }
          if (retirable) active = false;
        }
        bool more = true;
        while (more) {
          // (b) claim the reserved block for free lanes
          const unsigned long long freeLanes = __ballot(!active);
          const bool canGrab = !exhausted && (uint32_t)__popcll(freeLanes) >= G;
          bool got = false; uint32_t newIdx = 0; float4 r0, r1, r2;
          MI355_UNDEF4(r0); MI355_UNDEF4(r1); MI355_UNDEF4(r2);               // (only read under `got`: no twelve v_mov per pass to give them a value)
          if (canGrab) {
            while (!exhausted) {
              // A wave without a reserved block (launch start: all 64 lanes free) takes as many blocks as it has room for with ONE atomic: 4096 waves asking
              // for four blocks one after the other is 16384 atomics on eight words, ~20 us in which nobody traverses.  Blocks v, v + 1, ... of a cursor are
              // not neighbours in the ray array (cursors interleave), which is as good as any other order.
              uint32_t take = 1u;
              if (!resValid) { take = max(1u, (uint32_t)__popcll(freeLanes) >> a.gShift); if (lane == 0u) resV = atomicAdd(a.counter + cursor * CURSOR_STRIDE, take); resValid = true; }
              const uint32_t base = __builtin_amdgcn_readfirstlane(resV);
              // 32-bit arithmetic throughout (the launch refuses more than 0xFFF00000 rays; a dry cursor is over-asked by a few blocks per wave at most): block
              // (base << cShift) + cursor of the batch, rays block << gShift ...  (64-bit products and two divisions by run-time values were a tenth of this block)
              const uint32_t block = (base << a.cShift) + cursor;
              resValid = false;
              if (block >= totalBlocks) {                                 // this cursor is dry: try the next one
                cursor = (cursor + 1u) & (a.numCursors - 1u);
                if (++dryCursors >= a.numCursors) exhausted = true;
                continue;
              }
              const uint32_t rank = rank_below(freeLanes);
              const uint32_t myRay = ((((base + (rank >> a.gShift)) << a.cShift) + cursor) << a.gShift) + (rank & (G - 1u));   // block rank / G of this grab, ray rank % G of it
              got = ((freeLanes >> lane) & 1ull) != 0ull && rank < (take << a.gShift) && myRay < rayCount;
              if (got && a.deferList) { const uint32_t q = myRay; newIdx = a.deferList[q >> 6] * 64u + (q & 63u); got = newIdx < a.count; }   // (the last packet of a batch may be ragged)
              else if (got) newIdx = myRay;
              if (got) {
                const float4* rp = (const float4*)(a.rays + (size_t)newIdx * a.stride);
                r0 = rp[0]; r1 = rp[1]; r2 = rp[2];
              }
              if (lane == 0u) resV = atomicAdd(a.counter + cursor * CURSOR_STRIDE, 1u);   // reserve the next block while these loads fly
              resValid = true;
              break;
            }
          }
          // (d) start the new rays
          if (got) {
            rayIdx = newIdx;
            ox = r0.x; oy = r0.y; oz = r0.z; tnear = r0.w;
            dx = r1.x; dy = r1.y; dz = r1.z;
            tfar = r2.x; tfar0 = r2.x; rmask = __float_as_uint(r2.y);
            setup_rdir<ROBUST>(dx, dy, dz, rdx, rdy, rdz, rfx, rfy, rfz, octinv4);
            tnearTrav = fmaxf(tnear, 0.0f);                               // tnear/tfar clamped to >= 0 for traversal (bvh_intersector1.cpp:65)
            if (INST) { inst = NO_INST; bestInst = NO_INST; }
            sp = 0; ngBase = 0; ngHits = 0x80000000u; tgBase = 0; tgHits = 0;   // "the root is the one hit child of a virtual node"
            lastTicket = qHead; travDone = false;
            best[lane] = ((unsigned long long)__float_as_uint(tfar) << 32) | 0xFFFFFFFFull;
            pend[lane] = 0u; lastT[lane] = qHead;
            active = a.hasRoot != 0u && !(ANY && tfar < 0.0f);            // empty scene / already occluded (bvh_intersector1.cpp:128)
            if (STATS) stRays++;
          }
          more = canGrab && !exhausted;                                   // more free lanes than one block (launch start, small G)
        }
        if (STATS) stRefillClk += __builtin_readcyclecounter() - stT0;
      }
      if (__ballot(active) == 0ull) { if (exhausted) break; else continue; }
    }
    // ------------------------------------------------------------------ 1b. tail: lanes without a ray help the rays that are left
    // Once the cursors are dry a wave keeps paying full instruction cost for its last, longest rays (22 % of the lane-iterations of a
    // lone 2^20-ray launch belong to lanes without a ray) and the launch cannot end before its longest ray has walked its nodes one
    // after the other.  So a free lane takes the top stack entry (= the nearest pending sub-trees) of a lane that still traverses,
    // copies that lane's ray and traverses the entry on its behalf: hits go to best[owner] through the ring exactly like the
    // owner's own (ring pairs carry the owner, the testers fetch the ray from the owner's registers), the owner retires when its own
    // traversal is done, pend[owner] == 0 and the ring has passed the last ticket any of its helpers drew.  The result is the same
    // minimum over all accepted candidates; only the order in which sub-trees are visited changes.
    if (exhausted && a.helpers) {
      const unsigned long long freeM = __ballot(!active);
      // INST: only sub-trees INSIDE an instance are given away (entries above the depth at which the donor entered it): the helper copies the donor's
      // object-space ray and never changes space; the donor stays in the instance until its helpers are done (step 2)
      const bool canGive = active && !travDone && sp > (INST ? topSp : 0u) && sp <= (uint32_t)QSTACK_LDS && !(ANY && (uint32_t)best[owner] != MI355_EMPTY_REF) && (!INST || inst != NO_INST);
      const unsigned long long giveM = __ballot(canGive);
      if (freeM != 0ull && giveM != 0ull) {
        helpersUsed = true;
        const uint32_t k = min((uint32_t)__popcll(freeM), (uint32_t)__popcll(giveM));
        const bool gives = canGive && rank_below(giveM) < k;
        uint2 e = make_uint2(0u, 0u);
        if (gives) { sp--; e = stk[sp * 64u]; }
        const uint32_t rf = rank_below(freeM);
        const bool takes = !active && rf < k;
        // lane of the rf-th giver: rf-th set bit of giveM
        uint32_t n = takes ? rf : 0u, pos = 0u, w = (uint32_t)giveM, c = (uint32_t)__popc(w);
        if (n >= c) { n -= c; pos = 32u; w = (uint32_t)(giveM >> 32); }
        c = (uint32_t)__popc(w & 0xFFFFu); if (n >= c) { n -= c; pos += 16u; w >>= 16; }
        c = (uint32_t)__popc(w & 0xFFu);   if (n >= c) { n -= c; pos += 8u;  w >>= 8; }
        c = (uint32_t)__popc(w & 0xFu);    if (n >= c) { n -= c; pos += 4u;  w >>= 4; }
        c = (uint32_t)__popc(w & 0x3u);    if (n >= c) { n -= c; pos += 2u;  w >>= 2; }
        if (n >= (w & 1u)) pos += 1u;
        const int src = (int)(pos & 63u);
        const uint32_t gx = (uint32_t)__shfl((int)e.x, src, 64), gy = (uint32_t)__shfl((int)e.y, src, 64), gown = (uint32_t)__shfl((int)owner, src, 64);
        const float h0 = __shfl(ox, src, 64), h1 = __shfl(oy, src, 64), h2 = __shfl(oz, src, 64), h3 = __shfl(dx, src, 64), h4 = __shfl(dy, src, 64), h5 = __shfl(dz, src, 64);
        const float h6 = __shfl(rdx, src, 64), h7 = __shfl(rdy, src, 64), h8 = __shfl(rdz, src, 64), h9 = __shfl(tnear, src, 64), h10 = __shfl(tnearTrav, src, 64);
        const uint32_t hm = (uint32_t)__shfl((int)rmask, src, 64), ho = (uint32_t)__shfl((int)octinv4, src, 64);
        const uint32_t hinst = INST ? (uint32_t)__shfl((int)inst, src, 64) : 0u;
        float h11 = 0, h12 = 0, h13 = 0;
        if (ROBUST) { h11 = __shfl(rfx, src, 64); h12 = __shfl(rfy, src, 64); h13 = __shfl(rfz, src, 64); }
        if (takes) {
          ox = h0; oy = h1; oz = h2; dx = h3; dy = h4; dz = h5; rdx = h6; rdy = h7; rdz = h8; tnear = h9; tnearTrav = h10; rmask = hm; octinv4 = ho;
          if (ROBUST) { rfx = h11; rfy = h12; rfz = h13; }
          owner = gown; helper = true; active = true; travDone = false;
          if (INST) { inst = hinst; topSp = 0u; }
          sp = 0; ngBase = gx; ngHits = gy; tgBase = 0; tgHits = 0; lastTicket = qHead;
          atomicAdd(&pend[owner], 1u);
        }
      }
    }
    if (STATS && lane == 0u) stIter++;

    // ------------------------------------------------------------------ 2. current tfar = what the testers published; pop / finish traversal
    bool waitDrain = false;                                     // INST: this lane waits for the ring before it may change space
    if (active && !travDone) {
      const unsigned long long b = best[owner];
      tfar = __uint_as_float((uint32_t)(b >> 32));
      bool finished = false;
      if (ANY && (uint32_t)b != MI355_EMPTY_REF) { finished = true; tgHits = 0; ngHits = 0; sp = 0; lastTicket = qHead; }   // occluded: nothing left to wait for
      else if (tgHits == 0u && ngHits <= 0x00FFFFFFu) {
        bool mayPop = true;
        if (INST && inst != NO_INST && sp == topSp && !helper) {   // the instance's sub-tree is done: back to world space once my queued pairs are tested (and my helpers' too)
          if ((int)(qHead - lastTicket) >= 0 && (!helpersUsed || (pend[lane] == 0u && (int)(qHead - lastT[lane]) >= 0))) {
            if ((uint32_t)b != entryLo || (uint32_t)(b >> 32) != entryHi) bestInst = inst;
            inst = NO_INST;
            const float4* rp = (const float4*)(a.rays + (size_t)rayIdx * a.stride);
            const float4 r0 = rp[0], r1 = rp[1];
            ox = r0.x; oy = r0.y; oz = r0.z; dx = r1.x; dy = r1.y; dz = r1.z;
            setup_rdir<ROBUST>(dx, dy, dz, rdx, rdy, rdz, rfx, rfy, rfz, octinv4);
          } else { waitDrain = true; mayPop = false; }
        }
        if (mayPop) {                                              // (a lane that has just left an instance pops in the same iteration)
          if (sp != 0u) {
            sp--;
            uint2 e = stk[min(sp, (uint32_t)(QSTACK_LDS - 1)) * 64u];
            // (an LDS read of its own: the compiler otherwise selects between the two ADDRESSES and issues one flat_load -- slower than ds_read for the LDS case, and a flat
            // load is waited for with vmcnt(0), i.e. together with the triangle records step 0 has just asked for: their round trip no longer overlapped steps 2 - 3a)
            asm volatile("" : "+v"(e.x), "+v"(e.y));
            if (__builtin_expect(sp >= (uint32_t)QSTACK_LDS, 0)) { e = spill[sp - QSTACK_LDS]; asm volatile("" : "+v"(e.x), "+v"(e.y)); }   // (waited for in here, not with vmcnt(0) where the paths join)
            if (INST && e.y <= 0x00FFFFFFu) { tgBase = e.x; tgHits = e.y; }   // instances of a top node that are still to be visited
            else { ngBase = e.x; ngHits = e.y; }
          } else finished = true;
        }
      }
      if (finished) {
        if (helper) {                                             // hand the sub-tree back: my tickets first, then my share of pend
          atomicMax(&lastT[owner], lastTicket); atomicSub(&pend[owner], 1u);
          active = false; helper = false; owner = lane; if (INST) inst = NO_INST;
        } else travDone = true;                                    // lastTicket already names this ray's last queued pair
      }
    }

    // ------------------------------------------------------------------ 2b. INST: in world space the "triangle" bits of a node are instances: enter the first one
    if (INST && active && !travDone && inst == NO_INST && tgHits != 0u) {
      const uint32_t k = (uint32_t)__builtin_ctz(tgHits);
      tgHits &= tgHits - 1u;
      if (ngHits > 0x00FFFFFFu) {                                  // what is left of the node: its inner children ...
        const uint2 e = make_uint2(ngBase, ngHits);
        if (sp < (uint32_t)QSTACK_LDS) stk[sp * 64u] = e;
        else { if (sp - QSTACK_LDS < a.spillPerLane) spill[sp - QSTACK_LDS] = e; else a.status[STATUS_SPILL] = 1u; if (STATS) stSpill++; }
        sp++;
      }
      if (tgHits != 0u) {                                          // ... and its other instances (an entry with no inner-child bits)
        const uint2 e = make_uint2(tgBase, tgHits);
        if (sp < (uint32_t)QSTACK_LDS) stk[sp * 64u] = e;
        else { if (sp - QSTACK_LDS < a.spillPerLane) spill[sp - QSTACK_LDS] = e; else a.status[STATUS_SPILL] = 1u; if (STATS) stSpill++; }
        sp++;
      }
      if (STATS) stDepth = max(stDepth, sp);
      ngHits = 0u; tgHits = 0u;
      const uint32_t ii = __float_as_uint(a.tris[(size_t)(tgBase + k) * 3u + 2u].y);   // the top tree's leaf record: primID = index into insts[]
      const float4* ip = a.insts + (size_t)ii * 4u;
      const float4 m0 = ip[0], m1 = ip[1], m2 = ip[2], m3 = ip[3];
      if ((__float_as_uint(m3.z) & rmask) != 0u) {                 // ray mask test, instance_intersector.cpp:19-23
        if ((__float_as_uint(m3.w) & 1u) == 0u) {                  // (bit 0: the scene's own geometry, no transform)
          xfm_ray(m0, m1, m2, ox, oy, oz, dx, dy, dz);
          setup_rdir<ROBUST>(dx, dy, dz, rdx, rdy, rdz, rfx, rfy, rfz, octinv4);
        }
        const unsigned long long b = best[lane];
        inst = ii; topSp = sp; entryLo = (uint32_t)b; entryHi = (uint32_t)(b >> 32);
        ngBase = __float_as_uint(m3.x); ngHits = 0x80000000u;      // the object's root, as at ray start
      }
    }

    // ------------------------------------------------------------------ 3a. node step, first half: pick the child, issue its loads
    const bool doNode = active && !travDone && tgHits == 0u && ngHits > 0x00FFFFFFu;
    uint4 n0, n1, n2, n3, n4;
    uint32_t nodeIdx = 0u;                                      // (lanes that open no node fetch the root: see below)
    if (doNode) {
      const uint32_t bit = 31u - (uint32_t)__clz((int)ngHits);
      ngHits &= ~(1u << bit);
      if (ngHits > 0x00FFFFFFu) {
        const uint2 e = make_uint2(ngBase, ngHits);
        if (sp < (uint32_t)QSTACK_LDS) stk[sp * 64u] = e;
        else { if (sp - QSTACK_LDS < a.spillPerLane) spill[sp - QSTACK_LDS] = e; else a.status[STATUS_SPILL] = 1u; if (STATS) stSpill++; }
        sp++;
        if (STATS) stDepth = max(stDepth, sp);
      }
      const uint32_t slot = (bit ^ octinv4) & 7u;
      const uint32_t rel = (uint32_t)__popc(ngHits & ~(0xFFFFFFFFu << slot));
      nodeIdx = ngBase + rel;
    }
    // The five node loads are issued here and consumed in 3b, behind the triangle block -- by EVERY lane, whether it opens a node or not: s_waitcnt counts loads, not
    // registers, so the triangle block can only wait for "all but the five youngest loads" (= its prefetched records, step 0) if those five are issued on every path.  Under
    // `if (doNode)` the compiler has to assume they may be missing and waits with vmcnt(0): the node round trip then started only behind the triangle tests (rounds 1 - 4).
    // (a lane without a node reads the root: 64 lanes, one address)
    {
      const uint4* np = a.nodes + (size_t)nodeIdx * 5u;
      n0 = np[0]; n1 = np[1]; n2 = np[2]; n3 = np[3]; n4 = np[4];
    }

    // ------------------------------------------------------------------ 4. test queued pairs (queued in earlier iterations), 64 at a time (fewer only when nothing else can run)
    const bool anyTraversing = __ballot(active && !travDone && !waitDrain) != 0ull;
    bool usePre = pre;
    for (;;) {
      const uint32_t count = qTail - qHead;
      if (count == 0u) break;
      if (count < 64u && anyTraversing && (!INST || (uint32_t)__popcll(__ballot(waitDrain)) < a.drainWaiters)) break;   // INST: lanes that wait to leave an instance force a partial batch
      const uint32_t n = min(count, 64u);
      const bool mine = lane < n;
      if (!usePre) pe = queue[(qHead + (mine ? lane : 0u)) & (QCAP - 1u)];
      const uint2 e = pe;
      const int owner = (int)e.y;
      // the owner's ray (all 64 lanes execute the permutes; lanes without a pair read pair 0's owner and drop the result)
      const float gox = __shfl(ox, owner, 64), goy = __shfl(oy, owner, 64), goz = __shfl(oz, owner, 64);
      const float gdx = __shfl(dx, owner, 64), gdy = __shfl(dy, owner, 64), gdz = __shfl(dz, owner, 64);
      const float gtnear = __shfl(tnear, owner, 64);
      const uint32_t grmask = (uint32_t)__shfl((int)rmask, owner, 64);
      // A candidate is tested against the ray's OWN far limit, not against the best hit so far: the reference's form of the test, T <= absDen * tfar, is not
      // monotone in the rounded t = T * rcp(absDen), so with the current best as tfar a second triangle at exactly the same t (a shared edge hit exactly) was
      // accepted or not depending on which of the two was tested first -- one ray in 2^20 changed its primID between launches.  Every candidate in front of
      // the ray's own limit now reaches the atomic-min, which takes the minimum of (t bits, triangle index): the answer no longer depends on the order.
      const float gtfar0 = __shfl(tfar0, owner, 64);
      const uint32_t ginst = (INST && FILT) ? (uint32_t)__shfl((int)inst, owner, 64) : NO_INST;   // (rules of an instanced scene's geometries sit behind that instance's base)
      const uint32_t grayIdx = FILT == 2 ? (uint32_t)__shfl((int)rayIdx, owner, 64) : 0u;       // (what the filter function finds in ray.id: the index of the ray in the batch)
      if (STATS && lane == 0u) stTriBlk++;
      // (two copies of the test, one per source of the record: where the paths joined the registers were merged with v_mov -- and the compiler's s_waitcnt at the join has
      // to serve both: vmcnt(0).  Apart, the prefetched path waits for "all but the five node loads of 3a", the other one for its own loads.)
      auto test_pair = [&](const float4 q0, const float4 q1, float4 q2) {
        MI355_KEEP4(q2);
        if (STATS) { stTris++; if (a.touch) atomicOr(&a.touch[a.touchTriWord + (e.x >> 5)], 1u << (e.x & 31u)); }
        const uint32_t tmask = __float_as_uint(q2.w);
        TriOut w;
        bool ok = ROBUST ? tri_pluecker<false>(q0, q1, q2, gox, goy, goz, gdx, gdy, gdz, gtnear, gtfar0, w)
                         : tri_moeller<false>(q0, q1, q2, gox, goy, goz, gdx, gdy, gdz, gtnear, gtfar0, w);
        ok = ok && ((tmask & grmask) != 0u);                           // EMBREE_RAY_MASK, intersector_epilog.h:256-262
        uint32_t finst = MI355_EMPTY_REF;                              // FILT == 2: what the function finds in hit.instID[0] (instanced scenes: the id of the instance the owner is in)
        if (FILT == 2 && INST && ginst != NO_INST) { const float4 i3 = a.insts[(size_t)ginst * 4u + 3u]; if ((__float_as_uint(i3.w) & 1u) == 0u) finst = __float_as_uint(i3.y); }
        if (FILT && ok && a.rules) {                                   // device-side filter rule of the candidate's geometry: where the reference calls the filter callback
          uint32_t ri = __float_as_uint(q2.z);
          if (INST && ginst != NO_INST) ri += __float_as_uint(a.insts[(size_t)ginst * 4u + 3u].w) >> 8;
          ok = rule_accepts<ANY, ROBUST>(a.rules, ri, q0, q1, q2, w.t, gox, goy, goz, gdx, gdy, gdz);
          if (FILT == 2 && ok && a.filterFn != 0ull) {                  // ... and the filter FUNCTION of the query (filter_sycl.h:31-43: the geometry enabled it, or the query enforces it)
            const uint4 r0 = a.rules[(size_t)ri * 3u];
            if (a.filterEnforce != 0u || (r0.x & RULE_ARG_FILTER) != 0u) {
              const uint4 r2 = a.rules[(size_t)ri * 3u + 2u];
              ok = call_device_filter<ROBUST>(a.filterFn, a.filterCtx, (void*)(((unsigned long long)r2.w << 32) | r2.z), q0, q1, q2, gox, goy, goz, gdx, gdy, gdz, gtnear, grmask, grayIdx, finst);
            }
          }
        } else if (FILT == 2 && ok && a.filterFn != 0ull && a.filterEnforce != 0u) {   // (no rule table: no geometry enabled the function; an enforcing query calls it for all of them)
          ok = call_device_filter<ROBUST>(a.filterFn, a.filterCtx, nullptr, q0, q1, q2, gox, goy, goz, gdx, gdy, gdz, gtnear, grmask, grayIdx, finst);
        }
        if (ok) atomicMin(&best[owner], ((unsigned long long)__float_as_uint(w.t + 0.0f) << 32) | e.x);   // + 0: a hit at -0 must not sort as a huge key
      };
      if (mine) {
        if (usePre) { asm volatile("; prefetched records"); test_pair(pq0, pq1, pq2); }
        else {                                                            // a second batch in one iteration, or a partial one: no prefetch
          asm volatile("; records fetched here");
          const float4* tp = a.tris + (size_t)e.x * 3u;
          const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
          test_pair(q0, q1, q2);
        }
      }
      qHead += n; usePre = false;
    }

    // ------------------------------------------------------------------ 3b. node step, second half: 8 slab tests
    const unsigned long long stN0 = STATS ? __builtin_readcyclecounter() : 0ull;
    if (doNode) {
      if (STATS) { stNodes++; if (a.touch) atomicOr(&a.touch[nodeIdx >> 5], 1u << (nodeIdx & 31u)); }
      const float adx = __uint_as_float((n0.w & 0xFFu) << 23) * rdx;
      const float ady = __uint_as_float(((n0.w >> 8) & 0xFFu) << 23) * rdy;
      const float adz = __uint_as_float(((n0.w >> 16) & 0xFFu) << 23) * rdz;
      const float bx = (__uint_as_float(n0.x) - ox) * rdx, by = (__uint_as_float(n0.y) - oy) * rdy, bz = (__uint_as_float(n0.z) - oz) * rdz;
#if MI355_SEL_SGPR
      const lanemask_t sx = cmp_neg_s(rdx), sy = cmp_neg_s(rdy), sz = cmp_neg_s(rdz);
      const uint32_t nx0 = sel_s(n2.x, n3.z, sx), nx1 = sel_s(n2.y, n3.w, sx), fx0 = sel_s(n3.z, n2.x, sx), fx1 = sel_s(n3.w, n2.y, sx);
      const uint32_t ny0 = sel_s(n2.z, n4.x, sy), ny1 = sel_s(n2.w, n4.y, sy), fy0 = sel_s(n4.x, n2.z, sy), fy1 = sel_s(n4.y, n2.w, sy);
      const uint32_t nz0 = sel_s(n3.x, n4.z, sz), nz1 = sel_s(n3.y, n4.w, sz), fz0 = sel_s(n4.z, n3.x, sz), fz1 = sel_s(n4.w, n3.y, sz);
#else
      const bool sx = rdx < 0.0f, sy = rdy < 0.0f, sz = rdz < 0.0f;
      const uint32_t nx0 = sx ? n3.z : n2.x, nx1 = sx ? n3.w : n2.y, fx0 = sx ? n2.x : n3.z, fx1 = sx ? n2.y : n3.w;
      const uint32_t ny0 = sy ? n4.x : n2.z, ny1 = sy ? n4.y : n2.w, fy0 = sy ? n2.z : n4.x, fy1 = sy ? n2.w : n4.y;
      const uint32_t nz0 = sz ? n4.z : n3.x, nz1 = sz ? n4.w : n3.y, fz0 = sz ? n3.x : n4.z, fz1 = sz ? n3.y : n4.w;
#endif
      const float tmax0 = fmaxf(tfar, 0.0f);
      uint32_t hits;
      if (ROBUST) {
        const float scx = __uint_as_float((n0.w & 0xFFu) << 23), scy = __uint_as_float(((n0.w >> 8) & 0xFFu) << 23), scz = __uint_as_float(((n0.w >> 16) & 0xFFu) << 23);
        const float nox = __uint_as_float(n0.x), noy = __uint_as_float(n0.y), noz = __uint_as_float(n0.z);
        hits = test4_robust(nx0, ny0, nz0, fx0, fy0, fz0, n1.z, octinv4, scx, scy, scz, nox, noy, noz, ox, oy, oz, rdx, rdy, rdz, rfx, rfy, rfz, tnearTrav, tmax0) |
               test4_robust(nx1, ny1, nz1, fx1, fy1, fz1, n1.w, octinv4, scx, scy, scz, nox, noy, noz, ox, oy, oz, rdx, rdy, rdz, rfx, rfy, rfz, tnearTrav, tmax0);
      } else {
      // error bound of a plane distance q * a + b (see test4): off the near side, onto the far side
      const float ex = fmaf(fabsf(adx), 255.0f, fabsf(bx)) * 0x1p-21f, ey = fmaf(fabsf(ady), 255.0f, fabsf(by)) * 0x1p-21f, ez = fmaf(fabsf(adz), 255.0f, fabsf(bz)) * 0x1p-21f;
      SlabCoef k;
      k.sxy.x = adx; k.sxy.y = ady; k.szx.x = adz; k.szx.y = adx; k.syz.x = ady; k.syz.y = adz;
      k.bxy.x = bx - ex; k.bxy.y = by - ey; k.bzx.x = bz - ez; k.bzx.y = bx + ex; k.byz.x = by + ey; k.byz.y = bz + ez;
      hits = test4(nx0, ny0, nz0, fx0, fy0, fz0, n1.z, octinv4, k, tnearTrav, tmax0) |
             test4(nx1, ny1, nz1, fx1, fy1, fz1, n1.w, octinv4, k, tnearTrav, tmax0);
      if (STATS && hits == 0u) {
        // a visit that finds no child: is it a box the ray enters without entering any of its children (inherent to boxes), or does every child it enters lie
        // behind the hit found since the entry was pushed (what a distance kept with the stack entry could skip)?  The same test against the ray's OWN limit tells.
        const float tmaxOwn = fmaxf(tfar0, 0.0f);
        if ((test4(nx0, ny0, nz0, fx0, fy0, fz0, n1.z, octinv4, k, tnearTrav, tmaxOwn) | test4(nx1, ny1, nz1, fx1, fy1, fz1, n1.w, octinv4, k, tnearTrav, tmaxOwn)) != 0u) stCulled++;
      }
      }
      ngBase = n1.x; ngHits = (hits & 0xFF000000u) | (n0.w >> 24);
      tgBase = n1.y; tgHits = hits & 0x00FFFFFFu;
      if (STATS && hits == 0u) stEmpty++;
    }
    if (STATS) {
      stNodeClk += __builtin_readcyclecounter() - stN0;
      const bool anyNode = __ballot(doNode) != 0ull;
      if (lane == 0u && anyNode) stNodeBlk++;
      // lane-iteration census: where do the lanes that are NOT opening a node spend this iteration?
      if (!active) stIdle++; else if (travDone) { if ((int)(qHead - lastTicket) >= 0) stWaitBatch++; else stWaitDrain++; } else if (!doNode) stBlocked++;
    }

    // ------------------------------------------------------------------ 5. queue triangle bits
#if MI355_PUSH_SCAN
    // Every lane queues its triangle bits (at most pushRounds of them per iteration) behind those of the lanes below it: the places come from ONE inclusive prefix scan over
    // the wave (six DPP adds).  Before, a round queued one pair per lane -- ballot, popcount, rank, ~15 wave instructions -- and some lane holds five or more bits in nearly
    // every iteration: 5 rounds = 74 of the ~420 VALU instructions of an iteration, each serving ~14 lanes (profiles/r05_trace.md).  The loop below is 8 per pair of the busiest lane.
    // The order of the pairs in the ring changes (lane by lane instead of round by round); the result is the minimum over all candidates whatever the order.
    {
      const bool has = tgHits != 0u && (!INST || inst != NO_INST);       // only active, traversing lanes hold triangle bits (INST: in world space they are instances, see 2b)
      if (__ballot(has) != 0ull) {
        uint32_t c = has ? min((uint32_t)__popc(tgHits), a.pushRounds) : 0u;
        uint32_t incl = c;
        asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                     "s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                     "s_nop 1\n v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 1\n" : "+v"(incl));
        uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t room = QCAP - (qTail - qHead);
        if (total > room) {                                               // rare (launch start: 64 rays meet the same first leaves): the lanes whose pairs still fit queue theirs, the
          const uint32_t m = (uint32_t)__popcll(__ballot(incl <= room));  // others wait -- incl grows with the lane number, so those are lanes 0 .. m - 1.  (All or nothing would be a
          total = m != 0u ? (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(m - 1u)) : 0u;   // deadlock: a ring below 64 pairs is not drained while lanes traverse.)  m == 0: the ring
          if (lane >= m) c = 0u;                                          // holds more than 250 pairs and step 4 of the next iteration drains it.
        }
        uint32_t p8 = (qTail + incl - c) << 3;                            // byte position of this lane's first pair (the ring is aligned to its size: one v_and per address)
        if (c != 0u) lastTicket = qTail + incl;                           // one behind this lane's last pair
#pragma unroll 1
        for (uint32_t k = c; k != 0u; k--) {
          const uint32_t bitk = (uint32_t)__builtin_ctz(tgHits);
          tgHits &= tgHits - 1u;
          *(uint2*)((char*)queue + (p8 & (QCAP * 8u - 1u))) = make_uint2(tgBase + bitk, owner);
          p8 += 8u;
        }
        qTail += total;
      }
    }
#else
    for (uint32_t r = 0; r < a.pushRounds; r++) {
      const bool has = tgHits != 0u && (!INST || inst != NO_INST);       // only active, traversing lanes hold triangle bits (INST: in world space they are instances, see 2b)
      const unsigned long long m = __ballot(has);
      if (m == 0ull) break;
      const uint32_t n = (uint32_t)__popcll(m);
      if (QCAP - (qTail - qHead) < n) break;                             // no room: the drain below makes some, the bits wait
      if (has) {
        const uint32_t k = (uint32_t)__builtin_ctz(tgHits);
        tgHits &= tgHits - 1u;
        const uint32_t pos = qTail + rank_below(m);
        queue[pos & (QCAP - 1u)] = make_uint2(tgBase + k, owner);
        lastTicket = pos + 1u;
      }
      qTail += n;
    }
#endif
  }

  if (!ANY && dCount != 0u) flush_done();                        // the last finished rays
  if (iter >= a.iterCap) a.status[STATUS_ITER_CAP] = 1u;        // left the loop through the cap, not through "no rays left": results are incomplete
  wave_exit();
  if (STATS) {
    atomicAdd(&a.stats[0], (unsigned long long)stNodes);
    atomicAdd(&a.stats[1], (unsigned long long)stTris);
    atomicAdd(&a.stats[2], (unsigned long long)stRays);
    atomicAdd(&a.stats[3], (unsigned long long)stSpill);
    atomicMax(&a.stats[4], (unsigned long long)stDepth);
    atomicAdd(&a.stats[5], (unsigned long long)stIter);
    atomicAdd(&a.stats[6], (unsigned long long)stNodeBlk);
    atomicAdd(&a.stats[7], (unsigned long long)stTriBlk);
    atomicAdd(&a.stats[8], (unsigned long long)stIdle);
    atomicAdd(&a.stats[9], (unsigned long long)stWaitBatch);
    atomicAdd(&a.stats[10], (unsigned long long)stWaitDrain);
    atomicAdd(&a.stats[11], (unsigned long long)stBlocked);
    atomicAdd(&a.stats[12], (unsigned long long)stEmpty);
    atomicAdd(&a.stats[13], (unsigned long long)stCulled);
    if (lane == 0u) {
      atomicAdd(&a.stats[14], stRefillClk); atomicAdd(&a.stats[15], (unsigned long long)stRefillEv);
      atomicAdd(&a.stats[16], __builtin_readcyclecounter() - stClk0); atomicAdd(&a.stats[17], stNodeClk);
    }
  }
}

// =============================================================================================
// RTC_RAY_QUERY_FLAG_COHERENT: the 64 rays of a wavefront walk the tree TOGETHER.
//   The reference's coherent path (BVHNIntersectorKHybrid::intersectCoherent, kernels/bvh/bvh_intersector_hybrid.cpp:374-533) traverses a packet with a
//   shared stack and tests every node against all rays of the packet at once.  Here a packet is one wavefront = 64 CONSECUTIVE rays of the batch: the
//   stack lives per wave in LDS ({node, mask of the lanes that entered its box}), a node is fetched ONCE per packet through the scalar cache (the node index
//   is wave-uniform) instead of once per lane, every lane of the mask runs the 8 slab tests on it, the triangles of the leaf slots some lane entered are
//   fetched once and tested by all those lanes, and each lane keeps its own best hit in registers (no ring, no LDS atomics).  Children are visited in the
//   front-to-back order of the packet's first ray.  What a lane reports is the minimum of (t, triangle index) over all accepted candidates exactly as in
//   trace_kernel_q, so the two kernels give bit-identical answers; which one is faster depends on the rays: a packet pays ~200 VALU instructions per node
//   ANY of its rays visits, so it wins when neighbouring rays share most of their path (primary rays of a moderately tessellated scene, shadow rays towards
//   one light) and loses on incoherent batches -- the flag is the application's promise, as in the reference.
constexpr int PSTACK = 128;                 // stack entries per packet: <= 7 siblings left behind per level
struct PacketTraceArgs { const uint4* nodes; const float4* tris; uint32_t hasRoot; char* rays; uint32_t count, stride; uint32_t* deferList; uint32_t* deferCount; volatile uint32_t* status; uint32_t minServed; const uint4* rules;
                         uint32_t part, bailAbove, verdictAbove; };   // part: 0 = every packet, 1 = the sample (every PACKET_SAMPLE-th), 2 = the others; bailAbove: part 2 defers everything when the sample deferred more than this
constexpr uint32_t PACKET_SAMPLE = 32;

template <bool ANY, bool ROBUST>
__global__ __launch_bounds__(64) void trace_packet_kernel(PacketTraceArgs a) {
  __shared__ uint4 s_stk[PSTACK];                               // {node, lane mask lo, lane mask hi, -}
  const uint32_t lane = threadIdx.x;
  const uint4* __restrict__ nodes = a.nodes;
  const float4* __restrict__ tris = a.tris;
  // packets are dealt round-robin to the resident waves (no cursor: a single atomic word hands out ~88 packets per microsecond, which alone would take 0.19 ms
  // for the 16384 packets of a 2^20-ray batch -- more than the whole Cornell-box launch)
  // A batch whose packets do not stay together costs the packet attempt on top of the per-lane traversal (crown stand-in, primary rays: 1.06 instead of
  // 1.46 Grays/s).  Large batches are therefore traced in two launches: every 32nd packet first (part 1), then the rest (part 2) -- which looks at how
  // many packets of the sample gave up and, if that is more than a quarter, hands all of its packets to the per-lane kernel at once.
  // The decision must be the same for every block of the launch (the two modes split the packet index space differently): it is taken on a SNAPSHOT of the
  // sample's count (deferCount[1], written by the LAST block of the sample launch, see the end of this kernel), which no block of part 2 writes -- the live
  // count grows while part 2 runs.  (A 4-byte device-to-device copy between the two launches did the same and cost the Cornell box 40 % of its rate.)
  if (a.part == 2u && a.deferCount[1] > a.bailAbove) {
    // (64 packets per atomic: one word takes ~88 appends per microsecond, 15,000 single appends would cost more than the packets they save)
    for (uint32_t j0 = blockIdx.x * 64u;; j0 += gridDim.x * 64u) {
      const uint32_t j = j0 + lane, pk = j + j / (PACKET_SAMPLE - 1u) + 1u;
      const bool mine = (unsigned long long)pk * 64ull < a.count;
      const unsigned long long mm = __ballot(mine);
      if (mm == 0ull) break;
      uint32_t base = 0u;
      if (lane == 0u) base = atomicAdd(a.deferCount, (uint32_t)__popcll(mm));
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      if (mine) a.deferList[base + rank_below(mm)] = pk;
    }
    return;
  }
  for (uint32_t j = blockIdx.x;; j += gridDim.x) {
    const uint32_t pk = a.part == 0u ? j : (a.part == 1u ? j * PACKET_SAMPLE : j + j / (PACKET_SAMPLE - 1u) + 1u);
    const unsigned long long first = (unsigned long long)pk * 64ull;
    if (first >= a.count) break;
    const uint32_t idx = (uint32_t)first + lane;
    const bool valid = idx < a.count;
    char* const rp = a.rays + (size_t)(valid ? idx : (uint32_t)first) * a.stride;
    const float4 r0 = ((const float4*)rp)[0], r1 = ((const float4*)rp)[1], r2 = ((const float4*)rp)[2];
    const float ox = r0.x, oy = r0.y, oz = r0.z, tnear = r0.w, dx = r1.x, dy = r1.y, dz = r1.z;
    const uint32_t rmask = __float_as_uint(r2.y);
    float rdx, rdy, rdz, rfx = 0, rfy = 0, rfz = 0; uint32_t octinv4;
    setup_rdir<ROBUST>(dx, dy, dz, rdx, rdy, rdz, rfx, rfy, rfz, octinv4);
    const float tnearTrav = fmaxf(tnear, 0.0f);
    float bestT = r2.x; uint32_t bestTri = MI355_EMPTY_REF;
    const float tfar0 = r2.x;                                    // candidates are tested against the ray's own limit (see trace_kernel_q, step 4)
    bool alive = valid && a.hasRoot != 0u && !(ANY && bestT < 0.0f);
    unsigned long long am = __ballot(alive);
    if (am == 0ull) continue;
    const uint32_t pOct = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)(octinv4 & 7u), __builtin_ctzll(am), 64));   // the packet's order: its first ray's
    uint32_t sp = 0u;
    uint32_t cur = 0u; unsigned long long curMask = am;
    bool haveCur = true;
    // A packet whose rays do not stay together is not worth finishing: every node ANY ray visits costs the whole wave its 200 instructions.  Every 4 node
    // visits the packet looks at how many lanes those 4 visits served; fewer than 48 of 64 on average (env MI355_PACKET_MIN_LANES; a visit of the per-lane
    // kernel serves ~47 lanes for about the same instructions) and it gives up, leaves its rays untouched and puts
    // itself on the deferred list, which the per-lane kernel (trace_kernel_q) traces right behind this launch.  The reference's hybrid traversal switches to
    // single-ray traversal the same way when few rays of a packet are active (BVHNIntersectorKHybrid, bvh_intersector_hybrid.h:33-37: switchThreshold).
    uint32_t visits = 0u, served = 0u; bool gaveUp = false;
    for (uint32_t guard = 0; guard < (1u << 22); guard++) {
      if (!haveCur) {
        if (sp == 0u) break;
        sp--;
        const uint4 e = s_stk[sp];
        cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x);
        curMask = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)e.z) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
        if (ANY) curMask &= __ballot(alive);
        if (curMask == 0ull) continue;
      }
      haveCur = false;
      visits++; served += (uint32_t)__popcll(curMask);
      if ((visits & 3u) == 0u) {
        if (served < a.minServed) { gaveUp = true; break; }
        served = 0u;
      }
      // ---- the node, once per packet
      const uint4* np = nodes + (size_t)cur * 5u;
      const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
      const bool in = ((curMask >> lane) & 1ull) != 0ull && (!ANY || alive);
      uint32_t hitBits = 0u;
      if (in) {
        const float scx = __uint_as_float((n0.w & 0xFFu) << 23), scy = __uint_as_float(((n0.w >> 8) & 0xFFu) << 23), scz = __uint_as_float(((n0.w >> 16) & 0xFFu) << 23);
        const float nox = __uint_as_float(n0.x), noy = __uint_as_float(n0.y), noz = __uint_as_float(n0.z);
        const bool sx = rdx < 0.0f, sy = rdy < 0.0f, sz = rdz < 0.0f;
        const float tmax0 = fmaxf(bestT, 0.0f);
        // fast mode: plane distance = q * (scale * rdir) + (org_node - org_ray) * rdir with the per-axis error bound of test4; robust mode: test4_robust's arithmetic
        const float adx = scx * rdx, ady = scy * rdy, adz = scz * rdz;
        const float bx = (nox - ox) * rdx, by = (noy - oy) * rdy, bz = (noz - oz) * rdz;
        const float ex = fmaf(fabsf(adx), 255.0f, fabsf(bx)) * 0x1p-21f, ey = fmaf(fabsf(ady), 255.0f, fabsf(by)) * 0x1p-21f, ez = fmaf(fabsf(adz), 255.0f, fabsf(bz)) * 0x1p-21f;
#define MI355_PCHILD(J, QLX, QLY, QLZ, QHX, QHY, QHZ)                                                          \
        {                                                                                                      \
          const float lx = ubyte<(J) & 3>(QLX), ly = ubyte<(J) & 3>(QLY), lz = ubyte<(J) & 3>(QLZ);              \
          const float hx = ubyte<(J) & 3>(QHX), hy = ubyte<(J) & 3>(QHY), hz = ubyte<(J) & 3>(QHZ);              \
          float tN, tF;                                                                                        \
          if (ROBUST) {                                                                                        \
            const float nxq = sx ? hx : lx, fxq = sx ? lx : hx, nyq = sy ? hy : ly, fyq = sy ? ly : hy, nzq = sz ? hz : lz, fzq = sz ? lz : hz; \
            const float tnx = (fmaf(nxq, scx, nox) - ox) * rdx, tny = (fmaf(nyq, scy, noy) - oy) * rdy, tnz = (fmaf(nzq, scz, noz) - oz) * rdz; \
            const float tfx = (fmaf(fxq, scx, nox) - ox) * rfx, tfy = (fmaf(fyq, scy, noy) - oy) * rfy, tfz = (fmaf(fzq, scz, noz) - oz) * rfz; \
            tN = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tnearTrav)); tF = fminf(fminf(tfx, tfy), fminf(tfz, tmax0)); \
          } else {                                                                                             \
            const float ax = fmaf(lx, adx, bx), cx = fmaf(hx, adx, bx), ay = fmaf(ly, ady, by), cy = fmaf(hy, ady, by), az = fmaf(lz, adz, bz), cz = fmaf(hz, adz, bz); \
            const float tnx = fminf(ax, cx) - ex, tfx = fmaxf(ax, cx) + ex, tny = fminf(ay, cy) - ey, tfy = fmaxf(ay, cy) + ey, tnz = fminf(az, cz) - ez, tfz = fmaxf(az, cz) + ez; \
            tN = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tnearTrav)); tF = fminf(fminf(tfx, tfy), fminf(tfz, tmax0)); \
          }                                                                                                    \
          hitBits |= (tN <= tF) ? (1u << (J)) : 0u;                                                            \
        }
        MI355_PCHILD(0, n2.x, n2.z, n3.x, n3.z, n4.x, n4.z) MI355_PCHILD(1, n2.x, n2.z, n3.x, n3.z, n4.x, n4.z)
        MI355_PCHILD(2, n2.x, n2.z, n3.x, n3.z, n4.x, n4.z) MI355_PCHILD(3, n2.x, n2.z, n3.x, n3.z, n4.x, n4.z)
        MI355_PCHILD(4, n2.y, n2.w, n3.y, n3.w, n4.y, n4.w) MI355_PCHILD(5, n2.y, n2.w, n3.y, n3.w, n4.y, n4.w)
        MI355_PCHILD(6, n2.y, n2.w, n3.y, n3.w, n4.y, n4.w) MI355_PCHILD(7, n2.y, n2.w, n3.y, n3.w, n4.y, n4.w)
#undef MI355_PCHILD
      }
      const uint32_t imask = n0.w >> 24;
      const unsigned long long metaAll = ((unsigned long long)n1.w << 32) | n1.z;   // meta byte of slot s = bits 8s .. 8s+7
      // ---- leaf slots some lane entered: ALL their triangles (<= 24) are fetched in one round trip, lane j loading the j-th of them (nearest slot first:
      // far limits shrink early), then handed to the lanes that entered the slot one after the other with v_readlane (the triangle index is wave-uniform).
      // (One scalar load per triangle inside the loop exposed a memory round trip per triangle: 3.6 instead of 5.5 Grays/s on the Cornell box.)
      uint32_t T = 0u, myTri = 0u, mySlot = 0u;
      for (int k = 7; k >= 0; k--) {
        const uint32_t s_ = ((uint32_t)k ^ pOct) & 7u;
        if ((imask >> s_) & 1u) continue;
        const uint32_t meta = (uint32_t)(metaAll >> (8u * s_)) & 0xFFu;
        if (meta == 0u) continue;
        if (__ballot(((hitBits >> s_) & 1u) != 0u) == 0ull) continue;
        const uint32_t cnt = (uint32_t)__popc(meta >> 5), firstTri = n1.y + (meta & 31u);
        if (lane >= T && lane < T + cnt) { myTri = firstTri + (lane - T); mySlot = s_; }
        T += cnt;
      }
      if (T != 0u) {
        float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0;
        if (lane < T) { const float4* tp = tris + (size_t)myTri * 3u; q0 = tp[0]; q1 = tp[1]; q2 = tp[2]; }
        for (uint32_t t = 0; t < T; t++) {
#define MI355_BCAST(x) __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(x), (int)t))
          const float4 u0 = make_float4(MI355_BCAST(q0.x), MI355_BCAST(q0.y), MI355_BCAST(q0.z), MI355_BCAST(q0.w));
          const float4 u1 = make_float4(MI355_BCAST(q1.x), MI355_BCAST(q1.y), MI355_BCAST(q1.z), MI355_BCAST(q1.w));
          const float4 u2 = make_float4(MI355_BCAST(q2.x), MI355_BCAST(q2.y), MI355_BCAST(q2.z), MI355_BCAST(q2.w));
#undef MI355_BCAST
          const uint32_t ti = (uint32_t)__builtin_amdgcn_readlane((int)myTri, (int)t), sl = (uint32_t)__builtin_amdgcn_readlane((int)mySlot, (int)t);
          if (((hitBits >> sl) & 1u) != 0u && (!ANY || alive)) {
            TriOut w;
            bool ok = ROBUST ? tri_pluecker<false>(u0, u1, u2, ox, oy, oz, dx, dy, dz, tnear, tfar0, w)
                             : tri_moeller<false>(u0, u1, u2, ox, oy, oz, dx, dy, dz, tnear, tfar0, w);
            ok = ok && ((__float_as_uint(u2.w) & rmask) != 0u);
            if (ok && a.rules) ok = rule_accepts<ANY, ROBUST>(a.rules, __float_as_uint(u2.z), u0, u1, u2, w.t, ox, oy, oz, dx, dy, dz);
            if (ok) {
              const float tt = w.t + 0.0f;
              if (ANY) { bestTri = ti; alive = false; }
              else if (tt < bestT || (tt == bestT && ti < bestTri)) { bestT = tt; bestTri = ti; }   // minimum of (t, triangle index): trace_kernel_q's atomicMin key
            }
          }
        }
      }
      if (ANY && __ballot(alive) == 0ull) break;
      // ---- inner children some lane entered: far ones first onto the stack, the nearest is opened next
      for (int k = 0; k < 8; k++) {
        const uint32_t s_ = ((uint32_t)k ^ pOct) & 7u;
        if (!((imask >> s_) & 1u)) continue;
        const unsigned long long m = __ballot(((hitBits >> s_) & 1u) != 0u && (!ANY || alive));
        if (m == 0ull) continue;
        const uint32_t child = n1.x + (uint32_t)__popc(imask & ((1u << s_) - 1u));
        if (haveCur) {                                            // what was "next" so far is farther than this one: it goes onto the stack
          if (sp < (uint32_t)PSTACK) { if (lane == 0u) s_stk[sp] = make_uint4(cur, (uint32_t)curMask, (uint32_t)(curMask >> 32), 0u); sp++; }
          else a.status[STATUS_SPILL] = 1u;
        }
        cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)child); curMask = m; haveCur = true;
      }
    }
    if (gaveUp) {                                                // nothing has been written: the rays are as they came
      if (lane == 0u) a.deferList[atomicAdd(a.deferCount, 1u)] = pk;
      continue;
    }
    // ---- results
    if (valid && bestTri != MI355_EMPTY_REF) {
      if (ANY) *(float*)(rp + 32) = -__builtin_inff();           // Occluded1EpilogM: tfar = -inf
      else {
        const float4* tp = tris + (size_t)bestTri * 3u;
        const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
        TriOut w;
        const uint32_t pid = __float_as_uint(q2.y);
        if (ROBUST) tri_pluecker<true>(q0, q1, q2, ox, oy, oz, dx, dy, dz, 0.0f, 0.0f, w, (pid >> 31) != 0u);
        else tri_moeller<true>(q0, q1, q2, ox, oy, oz, dx, dy, dz, 0.0f, 0.0f, w, (pid >> 31) != 0u);
        *(float*)(rp + 32) = w.t;
        *(float4*)(rp + 48) = make_float4(w.Ngx, w.Ngy, w.Ngz, w.u);
        *(uint4*)(rp + 64) = make_uint4(__float_as_uint(w.v), pid & 0x7FFFFFFFu, __float_as_uint(q2.z), MI355_EMPTY_REF);
        *(uint32_t*)(rp + 80) = MI355_EMPTY_REF;
      }
    }
  }
  if (a.part == 1u && lane == 0u) {                              // the sample's verdict, frozen by the last block to leave (its appends above returned: they are performed)
    if (atomicAdd(a.deferCount + 2, 1u) == gridDim.x - 1u) {
      const uint32_t gaveUp = atomicAdd(a.deferCount, 0u);
      a.deferCount[1] = gaveUp; atomicExch(a.deferCount + 2, 0u);
      a.status[STATUS_COHERENT] = gaveUp > a.verdictAbove ? 2u : 1u;   // what the host remembers for the next coherent query on this (tree, stream): launch_trace_coherent
    }
  }
}

// ---- packet adaptor: SoA RTCRayHitK / RTCRayK <-> the AoS records the trace kernels consume ----
// (RayHitK::get/set kernels/common/ray.h:283-376; packet calls never touch lanes whose valid[i] != -1, InactiveRaysTest verify.cpp:3553)
// Lane i of the packet array becomes AoS record i, active or not: an inactive lane is copied with tnear = +inf, tfar = -inf (it cannot enter the root's
// box, so it costs one node visit and finds nothing) and is skipped on the way back.  No compaction, hence no counter to read back between the
// kernels: the three launches are enqueued back to back.
struct PacketArgs { const int* valid; char* packets; uint32_t K, numPackets; size_t packetStride; char* aos; };

__global__ void packet_gather(PacketArgs p, int withHit) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.K * p.numPackets) return;
  const uint32_t pk = i / p.K, k = i % p.K;
  const bool on = !p.valid || p.valid[i] == -1;
  const uint32_t* src = (const uint32_t*)(p.packets + (size_t)pk * p.packetStride);
  uint32_t* dst = (uint32_t*)(p.aos + (size_t)i * (withHit ? 96 : 48));
  for (int f = 0; f < 12; f++) dst[f] = src[f * p.K + k];
  if (!on) { dst[3] = 0x7F800000u; dst[8] = 0xFF800000u; }          // tnear = +inf, tfar = -inf
  if (withHit) for (int f = 0; f < 9; f++) dst[12 + f] = src[(12 + f) * p.K + k];
}
__global__ void packet_scatter(PacketArgs p, int withHit) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.K * p.numPackets) return;
  if (p.valid && p.valid[i] != -1) return;
  const uint32_t pk = i / p.K, k = i % p.K;
  uint32_t* dst = (uint32_t*)(p.packets + (size_t)pk * p.packetStride);
  const uint32_t* src = (const uint32_t*)(p.aos + (size_t)i * (withHit ? 96 : 48));
  dst[8 * p.K + k] = src[8];   // tfar
  if (withHit && src[12 + 6] != MI355_EMPTY_REF)
    for (int f = 0; f < 9; f++) dst[(12 + f) * p.K + k] = src[12 + f];
}

}  // namespace

#ifdef MI355_FPTR_TU
// ------------------------------------------------------------------------------------- trace_fptr.hip: this file again, for the FILT == 2 instantiations only
// The kernels that CALL a device filter function are compiled in a translation unit of their own (embree_amd/csrc/trace_fptr.hip = this file with MI355_FPTR_TU defined)
// WITHOUT the record prefetch of step 0 (-DMI355_TRI_PREFETCH=0, embree_amd/build.py).  Round 5 found that above -O1 the loop around the indirect call loses its rays or
// faults with ANY callee and compiled this unit at -O1; round 6 bisected the kernel's hand-written pieces at -O3 on the GPU (profiles/r06_device_filter.md): SGPR lane masks,
// the DPP scan and the "undefined register" asm are innocent, the prefetched triangle records in flight across the call are not.  The ordinary kernels keep the prefetch
// (they call nothing); everything below this block (the host side) exists once, in trace.hip.
namespace mi355 {
template <bool INST> static void* fptr_kernel_i(bool any, bool robust) {
  if (robust) return any ? (void*)trace_kernel_q<true, false, true, INST, 2> : (void*)trace_kernel_q<false, false, true, INST, 2>;
  return any ? (void*)trace_kernel_q<true, false, false, INST, 2> : (void*)trace_kernel_q<false, false, false, INST, 2>;
}
void* fptr_kernel(bool any, bool robust, bool inst) { return inst ? fptr_kernel_i<true>(any, robust) : fptr_kernel_i<false>(any, robust); }   // (round 6: scenes with instances too)
}  // namespace mi355
#else
// ------------------------------------------------------------------------------------- host side
namespace mi355 {
void* fptr_kernel(bool any, bool robust, bool inst);            // the FILT == 2 kernels (device filter functions): trace_fptr.hip

typedef void (*TraceFn)(TraceArgs);
static uint32_t log2floor_u32(uint32_t v) { uint32_t l = 0; while ((2u << l) <= v) l++; return l; }
static uint32_t env_u32(const char* name, uint32_t def, uint32_t lo, uint32_t hi) {
  const char* e = getenv(name); if (!e) return def;
  const long v = atol(e); return v < (long)lo || v > (long)hi ? def : (uint32_t)v;
}
template <bool INST, int FILT, bool SMALL = false> static TraceFn pick_kernel_if(bool any, bool robust) {
  if (robust) return any ? trace_kernel_q<true, false, true, INST, FILT, SMALL> : trace_kernel_q<false, false, true, INST, FILT, SMALL>;
  return any ? trace_kernel_q<true, false, false, INST, FILT, SMALL> : trace_kernel_q<false, false, false, INST, FILT, SMALL>;
}
static TraceFn pick_small_kernel(bool any, bool robust, bool inst, bool filt) {      // the static launch shape (no filter function, no counting build)
  if (inst) return filt ? pick_kernel_if<true, true, true>(any, robust) : pick_kernel_if<true, false, true>(any, robust);
  return filt ? pick_kernel_if<false, true, true>(any, robust) : pick_kernel_if<false, false, true>(any, robust);
}
template <bool INST> static TraceFn pick_stats_i(bool any, bool robust) {            // the counting build always knows the rules (it is not the measured path)
  if (robust) return any ? trace_kernel_q<true, true, true, INST, true> : trace_kernel_q<false, true, true, INST, true>;
  return any ? trace_kernel_q<true, true, false, INST, true> : trace_kernel_q<false, true, false, INST, true>;
}
static TraceFn pick_kernel(bool any, bool stats, bool robust, bool inst, bool filt, bool fptr = false) {
  if (fptr) return (TraceFn)fptr_kernel(any, robust, inst);      // (a filter function of the query: compiled in trace_fptr.hip)
  if (stats) return inst ? pick_stats_i<true>(any, robust) : pick_stats_i<false>(any, robust);
  if (inst) return filt ? pick_kernel_if<true, true>(any, robust) : pick_kernel_if<true, false>(any, robust);
  return filt ? pick_kernel_if<false, true>(any, robust) : pick_kernel_if<false, false>(any, robust);
}
// persistent grid = exactly the blocks that are resident at once (a larger grid would run a second, ragged round)
static uint32_t resident_blocks(Bvh* b, TraceFn fn) {
  static std::mutex m; static std::map<std::pair<int, TraceFn>, uint32_t> cache;
  std::lock_guard<std::mutex> lk(m);
  auto key = std::make_pair(b->device, fn);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int perCU = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (const void*)fn, BLOCK, 0) != hipSuccess || perCU < 1) perCU = 4;
  if (perCU > MAX_BLOCKS_PER_CU) perCU = MAX_BLOCKS_PER_CU;
  const char* e = getenv("MI355_TRACE_BLOCKS_PER_CU");
  if (e && atoi(e) > 0 && atoi(e) <= MAX_BLOCKS_PER_CU) perCU = atoi(e);
  return cache[key] = (uint32_t)b->numCUs * (uint32_t)perCU;
}

uint32_t trace_spill_per_lane(uint32_t depth) { return depth + 2u > (uint32_t)QSTACK_LDS ? depth + 2u - (uint32_t)QSTACK_LDS : 0u; }   // one entry per level at most
size_t trace_spill_bytes(int numCUs, uint32_t depth) {
  return (size_t)numCUs * MAX_BLOCKS_PER_CU * BLOCK * trace_spill_per_lane(depth) * sizeof(uint2) + 1024;   // + slack: after a (flagged) overflow a lane still pops what it believes it pushed
}

// (callers hold sc->enqueue of the stream's scratch: see launch_trace / launch_trace_coherent)
struct FilterCall { unsigned long long fn = 0; void* ctx = nullptr; uint32_t enforce = 0; };   // a __device__ filter function of the caller (mi355_trace_query_filtered)
static int launch_trace_locked(Bvh* b, TraceScratch* sc, void* d_rays, uint32_t count, size_t stride, bool any, hipStream_t s, uint64_t* statsOut,
                               hipEvent_t evStart = nullptr, hipEvent_t evStop = nullptr, const uint32_t* deferList = nullptr, const uint32_t* deferCount = nullptr,
                               const FilterCall* fc = nullptr) {
  if (count == 0) return 0;
  if (stride < (any ? 48u : 96u) || (stride & 15u) || ((uintptr_t)d_rays & 15u)) return set_error(hipErrorInvalidValue, "ray array must be 16-byte aligned with a 16-byte-multiple stride");
  if (count > 0xFFF00000u) return set_error(hipErrorInvalidValue, "more than 0xFFF00000 rays in one launch (the hand-out arithmetic is 32-bit)");
  HIP_TRY(hipSetDevice(b->device));
  TraceFn fn = pick_kernel(any, statsOut != nullptr, b->robust, b->d_insts != nullptr, b->d_rules != nullptr, fc && fc->fn != 0ull);
  const uint32_t maxBlocks = resident_blocks(b, fn);
  uint32_t blocks = (count + BLOCK - 1) / BLOCK;
  if (blocks > maxBlocks) blocks = maxBlocks;
  // The launch shape follows the ray count (round 6; the sweep behind the numbers: profiles/r06_batch_sweep.md).  More rays than lane slots: the persistent grid with the
  // cursor hand-out, as before.  A batch that fits into HALF the resident waves at 16 rays each or more (count <= 2^16 on 256 CUs): static ownership, R rays per wave with
  // R the smallest power of two >= 16 that keeps the grid at two waves per SIMD or fewer -- a launch of 2^12 .. 2^15 rays is ONE ray's dependent chain of node
  // fetches long (84 us for 4096 rays), and 48 helper lanes per wave from the first iteration on shorten that chain (2^15 rays: 136 -> 106 us).  More waves than two per
  // SIMD do NOT help: an iteration costs a wave the same ~420 instructions whether 16 or 64 of its lanes hold a ray, and two waves already keep a SIMD's issue port busy
  // (2^17 rays: 4096 waves x 32 rays 186 us, 2048 x 64 171 us).  MI355_STATIC_RAYS=0 switches the static shape off, =16/32/64 fixes R for every batch that fits.
  static const uint32_t staticEnv = env_u32("MI355_STATIC_RAYS", 1, 0, 64);
  static const uint32_t smallFrac8 = env_u32("MI355_SMALL_FRAC8", 8, 1, 8);       // A/B: a batch with fewer rays than lane slots runs on this many eighths of the waves it could fill (the rest is refill)
  uint32_t staticRays = 0u;
  if (!deferList && !statsOut && !(fc && fc->fn) && BLOCK == 64 && (uint64_t)count <= (uint64_t)maxBlocks * 64u) {
    if (staticEnv == 1u) {
      uint32_t R = 16u; while (R < 64u && (uint64_t)(maxBlocks / 2u) * R < count) R <<= 1;
      if ((uint64_t)(maxBlocks / 2u) * R >= count && R < 64u) staticRays = R;
    } else if (staticEnv >= 16u) staticRays = 1u << log2floor_u32(staticEnv);
    if (staticRays) { blocks = (count + staticRays - 1u) / staticRays; if (blocks > maxBlocks) { staticRays = 0u; blocks = (count + BLOCK - 1) / BLOCK; } }
    if (!staticRays && smallFrac8 < 8u) { blocks = (uint32_t)(((uint64_t)blocks * smallFrac8 + 7u) / 8u); if (blocks == 0u) blocks = 1u; }
  }
  TraceArgs a;                                                 // (the ray cursors are zero: the last wave of the previous launch on this stream left them so, see wave_exit)
  a.nodes = (const uint4*)b->d_nodes; a.tris = (const float4*)b->d_tris; a.hasRoot = b->root != MI355_EMPTY_REF ? 1u : 0u;
  a.rays = (char*)d_rays; a.count = count; a.stride = (uint32_t)stride; a.insts = (const float4*)b->d_insts; a.deferList = deferList; a.deferCount = deferCount; a.rules = (const uint4*)b->d_rules;
  a.counter = sc->counter; a.spill = (uint2*)sc->spill; a.spillPerLane = trace_spill_per_lane(b->info.depth); a.stats = nullptr;
  a.filterFn = fc ? fc->fn : 0ull; a.filterCtx = fc ? fc->ctx : nullptr; a.filterEnforce = fc ? fc->enforce : 0u;
  a.staticRays = staticRays;
  if (staticRays) fn = pick_small_kernel(any, b->robust, b->d_insts != nullptr, b->d_rules != nullptr);   // (the SMALL instantiation: same code, rays started in front of the loop)
  static const uint32_t refillMin = env_u32("MI355_REFILL_MIN", REFILL_MIN_DEFAULT, 1, 64);
  static const uint32_t pushRounds = env_u32("MI355_PUSH_ROUNDS", PUSH_ROUNDS_DEFAULT, 1, 24);
  static const uint32_t numCursors = env_u32("MI355_NUM_CURSORS", NUM_CURSORS, 1, NUM_CURSORS);
  static const uint32_t drainWaiters = env_u32("MI355_DRAIN_WAITERS", 3, 1, 65);
  auto log2floor = [](uint32_t v) { uint32_t l = 0; while ((2u << l) <= v) l++; return l; };
  a.gShift = log2floor(refillMin); a.cShift = log2floor(numCursors);
  { static bool warned = false;                                  // (an environment value that is not a power of two used to be rounded down without a word)
    if (!warned && ((1u << a.gShift) != refillMin || (1u << a.cShift) != numCursors)) {
      warned = true; fprintf(stderr, "[mi355] MI355_REFILL_MIN / MI355_NUM_CURSORS must be powers of two: using %u / %u\n", 1u << a.gShift, 1u << a.cShift); } }
  // The cursors are zero because the last wave of the previous launch on this (tree, stream) left them so (wave_exit).  A launch that did NOT run to its end -- it raised the
  // iteration-cap word, or the application reset the device under it -- may have left them anywhere: after a raised status word the cursors are zeroed on the stream before the next
  // launch (mi355_trace_status, below); MI355_TRACE_MEMSET=1 zeroes them in front of EVERY launch, as rounds 1 - 3 did (A/B and debugging: ADVICE r04).
  static const bool memsetAlways = env_u32("MI355_TRACE_MEMSET", 0, 0, 1) != 0u;
  if (memsetAlways || sc->cursorsDirty) { HIP_TRY(hipMemsetAsync(sc->counter, 0, (EXIT_WORD + 1u) * sizeof(uint32_t), s)); sc->cursorsDirty = false; }
  a.refillMin = 1u << a.gShift; a.numCursors = 1u << a.cShift; a.pushRounds = pushRounds; a.drainWaiters = drainWaiters;
  { const char* e = getenv("MI355_TRACE_ITER_CAP"); const long v = e ? atol(e) : 0; a.iterCap = v > 0 && v < (long)ITER_CAP ? (uint32_t)v : ITER_CAP; }
  a.status = sc->statusDev;
  { const char* e = getenv("MI355_TRACE_HELPERS"); a.helpers = e && atoi(e) == 0 ? 0u : 1u; }   // tail helpers (step 1b) on unless MI355_TRACE_HELPERS=0
  a.touch = nullptr; a.touchTriWord = 0u;
  if (statsOut) {
    HIP_TRY(hipMemsetAsync(sc->stats, 0, 32 * sizeof(uint64_t), s));
    a.stats = (unsigned long long*)sc->stats;
    // the unique nodes / triangle records this launch fetches (out[18], out[19]): a bit each, set by the counting kernel -- the COMPULSORY bytes of the launch, what a
    // perfect cache in front of the memory would still have to read once (bench.py: roofline.compulsory_bytes)
    const size_t nodeWords = ((size_t)b->info.num_nodes + 31u) / 32u + 1u, triWords = ((size_t)b->info.num_triangles + 31u) / 32u + 1u;
    uint32_t* d_touch = nullptr;
    if (hipMalloc((void**)&d_touch, (nodeWords + triWords) * 4u) == hipSuccess) {
      HIP_TRY(hipMemsetAsync(d_touch, 0, (nodeWords + triWords) * 4u, s));
      a.touch = d_touch; a.touchTriWord = (uint32_t)nodeWords;
    } else (void)hipGetLastError();
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(BLOCK), 0, s, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(statsOut, sc->stats, 32 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (d_touch) {
      std::vector<uint32_t> h(nodeWords + triWords);
      const int ce = mi355_memcpy_d2h(h.data(), d_touch, h.size() * 4u);   // (staged: the GPU never maps pageable host memory)
      hipFree(d_touch);
      if (ce) return ce;
      uint64_t un = 0, ut = 0;
      for (size_t i = 0; i < nodeWords; i++) un += (uint64_t)__builtin_popcount(h[i]);
      for (size_t i = nodeWords; i < h.size(); i++) ut += (uint64_t)__builtin_popcount(h[i]);
      statsOut[18] = un; statsOut[19] = ut;
    }
    return 0;
  }
  if (evStart) HIP_TRY(hipEventRecord(evStart, s));
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(BLOCK), 0, s, a);
  if (evStop) HIP_TRY(hipEventRecord(evStop, s));
  HIP_TRY(hipGetLastError());
  return 0;
}

static int launch_trace(Bvh* b, void* d_rays, uint32_t count, size_t stride, bool any, hipStream_t s, uint64_t* statsOut,
                        hipEvent_t evStart = nullptr, hipEvent_t evStop = nullptr, const FilterCall* fc = nullptr) {
  if (count == 0) return 0;
  HIP_TRY(hipSetDevice(b->device));
  TraceScratch* sc = b->scratch_for(s);
  if (!sc) return set_error(hipErrorOutOfMemory, "trace scratch allocation failed");
  std::lock_guard<std::mutex> enqueueLock(*sc->enqueue);       // rtcIntersect* are thread safe: another thread's reset must not slip between my reset and my kernel
  return launch_trace_locked(b, sc, d_rays, count, stride, any, s, statsOut, evStart, evStop, nullptr, nullptr, fc);
}

typedef void (*PacketFn)(PacketTraceArgs);
static int launch_trace_coherent(Bvh* b, void* d_rays, uint32_t count, size_t stride, bool any, hipStream_t s, bool noMemory = false) {
  if (count == 0) return 0;
  if (stride < (any ? 48u : 96u) || (stride & 15u) || ((uintptr_t)d_rays & 15u)) return set_error(hipErrorInvalidValue, "ray array must be 16-byte aligned with a 16-byte-multiple stride");
  HIP_TRY(hipSetDevice(b->device));
  const PacketFn fn = b->robust ? (any ? trace_packet_kernel<true, true> : trace_packet_kernel<false, true>) : (any ? trace_packet_kernel<true, false> : trace_packet_kernel<false, false>);
  static std::mutex m; static std::map<std::pair<int, PacketFn>, uint32_t> cache;
  uint32_t maxBlocks;
  { std::lock_guard<std::mutex> lk(m);
    auto key = std::make_pair(b->device, fn); auto it = cache.find(key);
    if (it == cache.end()) {
      int perCU = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (const void*)fn, 64, 0) != hipSuccess || perCU < 1) perCU = 8;
      if (perCU > 32) perCU = 32;
      it = cache.emplace(key, (uint32_t)b->numCUs * (uint32_t)perCU).first;
    }
    maxBlocks = it->second; }
  uint32_t blocks = (count + 63u) / 64u;
  if (blocks > maxBlocks) blocks = maxBlocks;
  TraceScratch* sc = b->scratch_for(s);
  if (!sc) return set_error(hipErrorOutOfMemory, "trace scratch allocation failed");
  const size_t packets = ((size_t)count + 63u) / 64u, need = (64u + packets) * sizeof(uint32_t);
  uint32_t* defer = nullptr;
  // ONE lock over the packet launches and the per-lane pass behind them: the deferred list belongs to (tree, stream), and a second thread's coherent query
  // on the same stream (all host queries use the null stream) must not reset or regrow it between my packets and my second pass
  std::lock_guard<std::mutex> enqueueLock(*sc->enqueue);
  // What the sample of the LAST large coherent query on this (tree, stream) said is still in the host-mapped status word: "its packets do not stay together"
  // (crown stand-in, primary rays: the sample launch, the list of packets that gave up and the indirection through it cost 10 % of the per-lane rate).  Then
  // this query goes to the per-lane kernel as it is; every 16th one is sampled again (a camera that moved into the open keeps its packets).  Results do not
  // depend on the path.
  // (Only after THREE samples in a row said so -- a sample that finds its packets together resets the count: an application that alternates coherent and incoherent
  // batches on one scene keeps sampling, because sending a coherent batch to the per-lane kernel costs far more than a sample does.)
  static const bool remember = env_u32("MI355_PACKET_REMEMBER", 1, 0, 1) != 0u;
  static const uint32_t sampleMin = env_u32("MI355_PACKET_SAMPLE_MIN", 1024, 0, 0x7FFFFFFF);   // packets: smaller batches are traced in one launch
  if (remember && !noMemory && packets >= sampleMin && packets >= 4u * PACKET_SAMPLE) {
    const uint32_t said = sc->statusHost[STATUS_COHERENT];     // the last sampled launch's verdict, once it has run (0: not yet, or taken already)
    if (said) { sc->statusHost[STATUS_COHERENT] = 0u; sc->divergeStreak = said == 2u ? sc->divergeStreak + 1u : 0u; }
    if (sc->divergeStreak >= 3u && (++sc->coherentCalls & 15u) != 0u)
      return launch_trace_locked(b, sc, d_rays, count, stride, any, s, nullptr, nullptr, nullptr);
  }
  {
    if (sc->deferCap < need) {                                  // (stream order keeps earlier launches' use of the old list apart: wait for them before it goes)
      if (sc->defer) { HIP_TRY(hipStreamSynchronize(s)); HIP_TRY(hipFree(sc->defer)); sc->defer = nullptr; sc->deferCap = 0; }
      const size_t cap = need < 65536 ? 65536 : need + need / 4;
      { const int rc = mi355_malloc_retry(b->device, cap, (void**)&sc->defer); if (rc) return rc; } sc->deferCap = cap;
    }
    defer = sc->defer;
    HIP_TRY(hipMemsetAsync(defer, 0, 3 * sizeof(uint32_t), s));   // [0] packets that gave up, [1] the sample's count as part 2 sees it, [2] blocks of the sample launch that have left
    PacketTraceArgs a;
    a.nodes = (const uint4*)b->d_nodes; a.tris = (const float4*)b->d_tris; a.hasRoot = b->root != MI355_EMPTY_REF ? 1u : 0u;
    a.rays = (char*)d_rays; a.count = count; a.stride = (uint32_t)stride; a.deferCount = defer; a.deferList = defer + 64; a.status = sc->statusDev; a.rules = (const uint4*)b->d_rules;
    static const uint32_t minLanes = env_u32("MI355_PACKET_MIN_LANES", 48, 0, 64);
    a.minServed = 4u * minLanes;
    a.verdictAbove = 0xFFFFFFFFu;
    if (packets >= sampleMin && packets >= 4u * PACKET_SAMPLE) {
      const uint32_t nSample = (uint32_t)((packets + PACKET_SAMPLE - 1u) / PACKET_SAMPLE), nRest = (uint32_t)packets - nSample;
      a.verdictAbove = nSample / 4u;
      a.part = 1u; a.bailAbove = 0xFFFFFFFFu;
      hipLaunchKernelGGL(fn, dim3(nSample < maxBlocks ? nSample : maxBlocks), dim3(64), 0, s, a);
      a.part = 2u; a.bailAbove = nSample / 4u;
      hipLaunchKernelGGL(fn, dim3(nRest < maxBlocks ? nRest : maxBlocks), dim3(64), 0, s, a);
    } else {
      a.part = 0u; a.bailAbove = 0xFFFFFFFFu;
      hipLaunchKernelGGL(fn, dim3(blocks), dim3(64), 0, s, a);
    }
    HIP_TRY(hipGetLastError()); }
  // the packets that gave up (their rays untouched), traced per lane right behind: the count stays on the device, a launch that finds none ends at once
  return launch_trace_locked(b, sc, d_rays, count, stride, any, s, nullptr, nullptr, nullptr, defer + 64, defer);
}

static int launch_packets(Bvh* b, const int* d_valid, void* d_pk, uint32_t K, uint32_t n, size_t pstride, bool any, hipStream_t s) {
  if (n == 0) return 0;
  if (K != 4 && K != 8 && K != 16) return set_error(hipErrorInvalidValue, "packet size must be 4, 8 or 16");
  HIP_TRY(hipSetDevice(b->device));
  const size_t rec = any ? 48 : 96, total = (size_t)K * n;
  if (total > 0xFFFFFFFFull) return set_error(hipErrorInvalidValue, "too many packets for one call");
  TraceScratch* sc = b->scratch_for(s);
  if (!sc) return set_error(hipErrorOutOfMemory, "trace scratch allocation failed");
  char* aos = nullptr;
  { std::lock_guard<std::mutex> lk(*sc->enqueue);               // the AoS staging area of this stream grows on demand and is reused by later calls (stream order keeps them apart)
    if (sc->pktCap < total * rec) {
      if (sc->pkt) { HIP_TRY(hipStreamSynchronize(s)); HIP_TRY(hipFree(sc->pkt)); sc->pkt = nullptr; sc->pktCap = 0; }
      const size_t cap = total * rec < 65536 ? 65536 : total * rec;
      { const int rc = mi355_malloc_retry(b->device, cap, &sc->pkt); if (rc) return rc; } sc->pktCap = cap;
    }
    aos = (char*)sc->pkt; }
  PacketArgs p{d_valid, (char*)d_pk, K, n, pstride, aos};
  const uint32_t g = (uint32_t)((total + 255) / 256);
  hipLaunchKernelGGL(packet_gather, dim3(g), dim3(256), 0, s, p, any ? 0 : 1);
  const int rc = launch_trace(b, aos, (uint32_t)total, rec, any, s, nullptr);
  if (rc == 0) hipLaunchKernelGGL(packet_scatter, dim3(g), dim3(256), 0, s, p, any ? 0 : 1);
  HIP_TRY(hipGetLastError());
  return rc;
}

}  // namespace mi355

extern "C" {
int mi355_trace_prepare(mi355_bvh_t bvh, void* stream) {      // allocates the per-stream traversal scratch now instead of inside the first launch on that stream
  mi355::Bvh* b = (mi355::Bvh*)bvh; HIP_TRY(hipSetDevice(b->device));
  return b->scratch_for((hipStream_t)stream) ? 0 : mi355::set_error(hipErrorOutOfMemory, "trace scratch allocation failed");
}
int mi355_trace_status(mi355_bvh_t bvh, void* stream, uint32_t* out) {
  mi355::Bvh* b = (mi355::Bvh*)bvh; HIP_TRY(hipSetDevice(b->device));
  mi355::TraceScratch* sc = b->scratch_for((hipStream_t)stream);
  if (!sc) return mi355::set_error(hipErrorOutOfMemory, "trace scratch allocation failed");
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  const uint32_t v = (sc->statusHost[0] ? MI355_TRACE_ITER_CAP_HIT : 0u) | (sc->statusHost[1] ? MI355_TRACE_STACK_OVERFLOW : 0u);
  sc->statusHost[0] = 0u; sc->statusHost[1] = 0u;
  if (v & MI355_TRACE_ITER_CAP_HIT) sc->cursorsDirty = true;    // waves left through the cap: nobody knows where the ray cursors stand (launch_trace_locked zeroes them before the next launch)
  if (out) *out = v;
  return 0;
}
int mi355_trace_closest(mi355_bvh_t bvh, void* d, uint32_t n, size_t stride, void* stream) {
  return mi355::launch_trace((mi355::Bvh*)bvh, d, n, stride, false, (hipStream_t)stream, nullptr);
}
int mi355_trace_any(mi355_bvh_t bvh, void* d, uint32_t n, size_t stride, void* stream) {
  return mi355::launch_trace((mi355::Bvh*)bvh, d, n, stride, true, (hipStream_t)stream, nullptr);
}
int mi355_trace_query(mi355_bvh_t bvh, void* d, uint32_t n, size_t stride, int any_hit, uint32_t query_flags, void* stream) {
  mi355::Bvh* b = (mi355::Bvh*)bvh;
  // RTC_RAY_QUERY_FLAG_COHERENT: the wave-packet kernel (not for scenes with instances: a packet cannot change space lane by lane)
  if ((query_flags & MI355_QUERY_COHERENT) && !b->d_insts) return mi355::launch_trace_coherent(b, d, n, stride, any_hit != 0, (hipStream_t)stream, (query_flags & MI355_QUERY_COHERENT_NO_MEMORY) != 0u);
  return mi355::launch_trace(b, d, n, stride, any_hit != 0, (hipStream_t)stream, nullptr);
}
int mi355_trace_query_filtered(mi355_bvh_t bvh, void* d, uint32_t n, size_t stride, int any_hit, uint32_t query_flags, uint64_t filter_fn, void* filter_ctx, void* stream) {
  mi355::FilterCall fc; fc.fn = filter_fn; fc.ctx = filter_ctx; fc.enforce = (query_flags & MI355_QUERY_INVOKE_ARGUMENT_FILTER) ? 1u : 0u;
  // (a query with a filter function goes to the per-lane kernel whatever RTC_RAY_QUERY_FLAG_COHERENT says: the packet kernel does not call functions)
  return mi355::launch_trace((mi355::Bvh*)bvh, d, n, stride, any_hit != 0, (hipStream_t)stream, nullptr, nullptr, nullptr, filter_fn ? &fc : nullptr);
}
int mi355_trace_timed(mi355_bvh_t bvh, void* d, uint32_t n, size_t stride, int any_hit, void* stream, void* ev_start, void* ev_stop) {
  return mi355::launch_trace((mi355::Bvh*)bvh, d, n, stride, any_hit != 0, (hipStream_t)stream, nullptr, (hipEvent_t)ev_start, (hipEvent_t)ev_stop);
}
int mi355_trace_stats(mi355_bvh_t bvh, void* d, uint32_t n, size_t stride, int any_hit, uint64_t out[32]) {
  for (int i = 0; i < 32; i++) out[i] = 0;
  return mi355::launch_trace((mi355::Bvh*)bvh, d, n, stride, any_hit != 0, nullptr, out);
}
int mi355_trace_closest_packet(mi355_bvh_t bvh, const int* v, void* d, uint32_t K, uint32_t n, size_t ps, void* stream) {
  return mi355::launch_packets((mi355::Bvh*)bvh, v, d, K, n, ps, false, (hipStream_t)stream);
}
int mi355_trace_any_packet(mi355_bvh_t bvh, const int* v, void* d, uint32_t K, uint32_t n, size_t ps, void* stream) {
  return mi355::launch_packets((mi355::Bvh*)bvh, v, d, K, n, ps, true, (hipStream_t)stream);
}
}
#endif  // MI355_FPTR_TU
