// bvh_common.h -- data layout of the MI355X BVH (shared by build.hip, trace.hip and the host).
//
// HBM layout (all arrays 128-B aligned, 32-bit indices instead of the reference's tagged 64-bit
// pointers, kernels/bvh/bvh_node_ref.h:59-241, so the tree is position independent):
//
//   QNode[num_nodes]   128 B each = exactly one L2 line, never straddling two.
//       header 16 B : quantisation origin (3 x f32) + 3 biased exponents + child count
//       child  12 B x 8 : {lo.xyz, hi.xyz as u8, 2 spare bytes, ref u32}  -- lane j of an
//                     8-lane "octet" reads child j, so one octet fetches the node as one
//                     coalesced 112-B read (the reference's AABBNode_t is 256 B of fp32 planes,
//                     kernels/bvh/bvh_node_aabb.h:216-221; its QuantizedNode 136 B, bvh_node_qaabb.h).
//       Decoded plane = origin + q * 2^(e-127); lower planes rounded down, upper planes rounded up
//       (conservative like QuantizedBaseNode_t::init_dim, bvh_node_qaabb.h:42-85).
//   TriRec[num_tris]   48 B each, in leaf order: v0, e1 = v0-v1, e2 = v2-v0 (the reference's
//                     TriangleM<4> stores the same three vectors SoA, kernels/geometry/triangle.h:40-41),
//                     primID, geomID, geometry mask.  Lane j of an octet reads triangle j of the leaf.
//   ref (u32)          bit31 = 0: inner node index.  bit31 = 1: leaf, bits 30..5 first TriRec, bits 4..0 count-1.
//                     0xFFFFFFFF = empty slot / empty scene.
#pragma once
#include <stdint.h>

#define MI355_EMPTY_REF 0xFFFFFFFFu
#define MI355_LEAF_BIT  0x80000000u
#define MI355_MAX_LEAF  32u

struct alignas(128) QNode {
  float org[3];
  uint8_t exp[3];      // biased exponents: plane = org + q * as_float(exp << 23)
  uint8_t count;       // number of used child slots (slots are filled from 0)
  uint32_t child[8][3];// 12 B per child: word0 = lo.x | lo.y<<8 | lo.z<<16 | hi.x<<24, word1 = hi.y | hi.z<<8, word2 = ref
  uint32_t pad[4];
};
static_assert(sizeof(QNode) == 128, "QNode must be one 128-byte line");

struct alignas(16) TriRec {
  float v0[3];
  float e1[3];
  float e2[3];
  uint32_t primID, geomID, mask;
};
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 bytes");

static inline __host__ __device__ uint32_t mi355_leaf_ref(uint32_t first, uint32_t count) {
  return MI355_LEAF_BIT | (first << 5) | (count - 1u);
}
static inline __host__ __device__ bool mi355_is_leaf(uint32_t ref) { return (ref & MI355_LEAF_BIT) != 0u; }
static inline __host__ __device__ uint32_t mi355_leaf_first(uint32_t ref) { return (ref & 0x7FFFFFFFu) >> 5; }
static inline __host__ __device__ uint32_t mi355_leaf_count(uint32_t ref) { return (ref & 31u) + 1u; }
