// bvh_common.h -- data layout of the MI355X BVH (shared by build.hip, trace.hip and the host).
//
// HBM layout (32-bit indices instead of the reference's tagged 64-bit pointers, kernels/bvh/bvh_node_ref.h:59-241,
// so the tree is position independent and can be copied / broadcast as two flat arrays):
//
//   CNode[num_nodes]   80 B each, 16-B aligned = five 16-byte loads per visit (one ray per lane; the
//                     memory pipeline charges per 16-byte lane-load, profiles/r01_pmc_trace_octet.md).
//       word 0-2   quantisation origin (f32 x 3) = lower corner of the node
//       word 3     ex | ey << 8 | ez << 16 | imask << 24   (biased exponents: plane = org + q * 2^(e-127);
//                  imask bit s = child slot s is an inner node)
//       word 4     childBase: index of the first inner child; inner children are consecutive in slot order
//       word 5     triBase:   index of the first TriRec of this node's leaf children (<= 24, consecutive)
//       word 6-7   meta[8]:   empty slot 0; inner slot (1 << 5) | (24 + s); leaf slot (unary count << 5) | offset,
//                  unary count = 1, 3, 7 for 1, 2, 3 triangles, offset = first triangle relative to triBase
//       word 8-19  qlo_x[8] qlo_y[8] qlo_z[8] qhi_x[8] qhi_y[8] qhi_z[8]  (u8; lower planes rounded down,
//                  upper planes rounded up: conservative like QuantizedBaseNode_t::init_dim, bvh_node_qaabb.h:42-85)
//       The reference's AABBNode_t<8> is 256 B of fp32 planes + 8 x 8-B refs (kernels/bvh/bvh_node_aabb.h:216-221);
//       its QuantizedNode 136 B (bvh_node_qaabb.h).  Children sit in the slot whose octant (bit0 = +x, bit1 = +y,
//       bit2 = +z side of the node centre) matches their position best, so that "slot XOR ray octant" is a
//       front-to-back order without any distance sort (the reference sorts by tNear, bvh_traverser1.h:311-433).
//   TriRec[num_tris]   48 B each, grouped per node: v0, e1 = v0-v1, e2 = v2-v0 (the reference's TriangleM<4> stores
//                     the same three vectors SoA, kernels/geometry/triangle.h:40-41), primID, geomID, geometry mask.
//   The root is node 0 (a scene of <= 3 triangles still gets one node with a single leaf slot).
#pragma once
#include <stdint.h>

#define MI355_EMPTY_REF 0xFFFFFFFFu
#define MI355_MAX_LEAF  3u            /* triangles per leaf slot (unary count in 3 bits) */
#define MI355_NODE_WORDS 20

struct alignas(16) CNode {
  float org[3];
  uint8_t exp[3];
  uint8_t imask;
  uint32_t childBase;
  uint32_t triBase;
  uint8_t meta[8];
  uint8_t qlo[3][8];
  uint8_t qhi[3][8];
};
static_assert(sizeof(CNode) == 80, "CNode must be 80 bytes");

struct alignas(16) TriRec {
  float v0[3];
  float e1[3];
  float e2[3];
  uint32_t primID, geomID, mask;
};
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 bytes");
