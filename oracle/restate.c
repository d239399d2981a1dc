/* oracle/restate.c -- TEST INFRASTRUCTURE.  Never linked, loaded or called by the
 * product path (embree_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker.
 *
 * A scalar, plain-C restatement of the reference's algorithm for the hot path
 * (Embree 4.4.1, triangle meshes, default MEDIUM build quality, AVX2 code path):
 *   commit    : PrimRef generation -> binned-SAH BVH8 -> Triangle4 leaf blocks
 *   intersect : single-ray stack traversal, Moeller-Trumbore, closest-hit epilog
 *   occluded  : any-hit traversal
 * Every function cites the reference file:line it follows (paths relative to the
 * reference root).  Arithmetic follows the reference's AVX2 contraction pattern:
 * madd/msub are single-rounded FMAs (fmaf), everything else is rounded per op
 * (compile with -ffp-contract=off), rcp is RCPPS + one Newton step.
 *
 * Parity is PINNED: tests/test_oracle.py checks this file against the
 * reference's own known-answer tests (TriangleHitTest, verify.cpp:2462-2547;
 * tutorials/minimal/minimal.cpp) and against outputs of the real reference
 * (oracle/_ref, built by oracle/ref.mk) on the same scenes and rays.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <xmmintrin.h>

#define API __attribute__((visibility("default")))
#define INVALID_ID 0xFFFFFFFFu

/* ---- ABI structs: include/embree4/rtcore_ray.h:11-52 (default ABI) ---- */
typedef struct { float org_x, org_y, org_z, tnear, dir_x, dir_y, dir_z, time, tfar; uint32_t mask, id, flags; } Ray;
typedef struct { float Ng_x, Ng_y, Ng_z, u, v; uint32_t primID, geomID, instID, instPrimID, pad[3]; } Hit;
typedef struct { Ray ray; Hit hit; } RayHit;

/* ---- builder data ---- */
typedef struct { float lo[3]; uint32_t geomID; float hi[3]; uint32_t primID; } PrimRef; /* kernels/builders/primref.h:11-107 */
typedef struct { float lo[3], hi[3]; } Box;
typedef struct { Box geom, cent; size_t begin, end; } Set; /* PrimInfoRange, kernels/builders/priminfo.h:77-162 */

/* BVH8 inner node: kernels/bvh/bvh_node_aabb.h:12-222 (SoA planes + 8 refs) */
typedef struct { float lower[3][8], upper[3][8]; int32_t child[8]; } Node8;
/* Triangle4 leaf block: kernels/geometry/triangle.h:13-156 (v0, e1=v0-v1, e2=v2-v0) */
typedef struct { float v0[3][4], e1[3][4], e2[3][4], v3[3][4]; uint32_t geomID[4], primID[4]; } Tri4;   /* quad scenes: QuadMv<4> = v0,v1,v2,v3 (kernels/geometry/quadv.h) */

typedef struct { const float* v; uint32_t nv; const uint32_t* t; uint32_t nt; uint32_t mask; float* vown; uint32_t* town; int quad; } Mesh;   /* quad: 4 indices per primitive */

typedef struct {
  Mesh* mesh; uint32_t nmesh, cmesh;
  PrimRef* prims; size_t nprims;
  Node8* nodes; size_t nnodes, cnodes;
  Tri4* blocks; size_t nblocks, cblocks;
  int32_t root;            /* >=0 inner node, <0 leaf, EMPTY if nothing valid */
  Box bounds;
  double sah;              /* sum(area(node))/area(root) style statistic, see ora_stats */
  uint64_t stat_nodes, stat_leaves, stat_blocks; /* traversal visit counters (STAT3, kernels/common/stat.h:9-19) */
  int quads;               /* this tree is the scene's QUAD accel (BVH8Quad4v, scene.cpp:276-320): the reference builds one BVH per geometry type and
                              queries them in turn (AccelN, kernels/common/acceln.cpp:44-50); oracle/restate.py keeps one Scene per type */
  uint32_t ctxInstID, ctxInstPrimID; /* RTCRayQueryContext::instID[0] / instPrimID[0] while this scene is queried through an instance (instance_stack.h:19-50); INVALID otherwise */
  int robust;              /* RTC_SCENE_FLAG_ROBUST: Triangle4v leaves (v0,v1,v2) + Pluecker test + conservative node test (scene.cpp:180-188) */
} Scene;

#define EMPTY_REF INT32_MIN
/* leaf ref: -(1 + (blockStart*8 + (numBlocks-1)))  (NodeRefPtr keeps count in low bits, bvh_node_ref.h:225-229; max 7 blocks :92) */
static int32_t enc_leaf(size_t start, size_t num) { return -(int32_t)(1 + start * 8 + (num - 1)); }
static void dec_leaf(int32_t r, size_t* start, size_t* num) { uint32_t x = (uint32_t)(-(r + 1)); *start = x >> 3; *num = (x & 7) + 1; }

/* ---- helpers ---- */
static float fminf_(float a, float b) { return a < b ? a : b; } /* SSE min/max operand order is irrelevant here (no NaNs in valid prims) */
static float fmaxf_(float a, float b) { return a > b ? a : b; }
static void box_empty(Box* b) { for (int k = 0; k < 3; k++) { b->lo[k] = INFINITY; b->hi[k] = -INFINITY; } }
static void box_extend(Box* b, const float* lo, const float* hi) { for (int k = 0; k < 3; k++) { b->lo[k] = fminf_(b->lo[k], lo[k]); b->hi[k] = fmaxf_(b->hi[k], hi[k]); } }
static void box_extend_pt(Box* b, const float* p) { box_extend(b, p, p); }
/* halfArea(d) = madd(d.x,(d.y+d.z),d.y*d.z)  common/math/vec3fa.h:349 */
static float half_area(const Box* b) {
  float dx = b->hi[0] - b->lo[0], dy = b->hi[1] - b->lo[1], dz = b->hi[2] - b->lo[2];
  return fmaf(dx, dy + dz, dy * dz);
}
/* rcp: RCPPS + Newton  r + r*(1 - a*r)  common/simd/vfloat4_sse2.h:304-321 (AVX2 branch: fnmadd, fmadd) */
static float rcp_nr(float a) {
  float r = _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(a)));
  return fmaf(r, fmaf(-a, r, 1.0f), r);
}
static float xorf(float a, uint32_t s) { uint32_t u; memcpy(&u, &a, 4); u ^= s; memcpy(&a, &u, 4); return a; }
static uint32_t fbits(float a) { uint32_t u; memcpy(&u, &a, 4); return u; }

/* ------------------------------------------------------------------ scene */
API void ora_set_robust(Scene* s, int robust) { s->robust = robust; }   /* rtcSetSceneFlags(RTC_SCENE_FLAG_ROBUST), before ora_commit */
API Scene* ora_new(void) { Scene* s = (Scene*)calloc(1, sizeof(Scene)); s->root = EMPTY_REF; s->ctxInstID = s->ctxInstPrimID = INVALID_ID; return s; }

static void free_build(Scene* s) {
  free(s->prims); free(s->nodes); free(s->blocks);
  s->prims = NULL; s->nodes = NULL; s->blocks = NULL;
  s->nprims = s->nnodes = s->cnodes = s->nblocks = s->cblocks = 0; s->root = EMPTY_REF;
}
API void ora_free(Scene* s) {
  if (!s) return;
  free_build(s);
  for (uint32_t i = 0; i < s->nmesh; i++) { free(s->mesh[i].vown); free(s->mesh[i].town); }
  free(s->mesh); free(s);
}
/* copies the arrays (like rtcSetNewGeometryBuffer); geomID = attach order (Scene::bind, kernels/common/scene.cpp:717-741) */
API uint32_t ora_add_mesh(Scene* s, const float* v, uint32_t nv, const uint32_t* t, uint32_t nt, uint32_t mask) {
  if (s->nmesh == s->cmesh) { s->cmesh = s->cmesh ? 2 * s->cmesh : 8; s->mesh = (Mesh*)realloc(s->mesh, s->cmesh * sizeof(Mesh)); }
  Mesh* m = &s->mesh[s->nmesh];
  m->vown = (float*)malloc((size_t)nv * 12 + 16); m->town = (uint32_t*)malloc((size_t)nt * 12 + 16);
  if (nv) memcpy(m->vown, v, (size_t)nv * 12);
  if (nt) memcpy(m->town, t, (size_t)nt * 12);
  m->v = m->vown; m->nv = nv; m->t = m->town; m->nt = nt; m->mask = mask; m->quad = 0;
  return s->nmesh++;
}
/* RTC_GEOMETRY_TYPE_QUAD: uint4 indices (kernels/common/scene_quad_mesh.h) */
API uint32_t ora_add_quads(Scene* s, const float* v, uint32_t nv, const uint32_t* q, uint32_t nq, uint32_t mask) {
  if (s->nmesh == s->cmesh) { s->cmesh = s->cmesh ? 2 * s->cmesh : 8; s->mesh = (Mesh*)realloc(s->mesh, s->cmesh * sizeof(Mesh)); }
  Mesh* m = &s->mesh[s->nmesh];
  m->vown = (float*)malloc((size_t)nv * 12 + 16); m->town = (uint32_t*)malloc((size_t)nq * 16 + 16);
  if (nv) memcpy(m->vown, v, (size_t)nv * 12);
  if (nq) memcpy(m->town, q, (size_t)nq * 16);
  m->v = m->vown; m->nv = nv; m->t = m->town; m->nt = nq; m->mask = mask; m->quad = 1; s->quads = 1;
  return s->nmesh++;
}

/* ------------------------------------------------------- PrimRef generation */
/* isvalid: -FLT_LARGE < x < FLT_LARGE, FLT_LARGE = 1.844e18  common/math/vec3fa.h:304, constants.h:21 */
static int valid_f(float x) { return x > -1.844E18f && x < 1.844E18f; }
/* TriangleMesh::buildBounds kernels/common/scene_triangle_mesh.h:195-215, createPrimRefArray :293-305 */
static void gen_primrefs(Scene* s, Set* all) {
  size_t total = 0;
  for (uint32_t g = 0; g < s->nmesh; g++) total += s->mesh[g].nt;
  s->prims = (PrimRef*)malloc((total + 1) * sizeof(PrimRef));
  box_empty(&all->geom); box_empty(&all->cent);
  size_t k = 0;
  for (uint32_t g = 0; g < s->nmesh; g++) {
    const Mesh* m = &s->mesh[g];
    for (uint32_t j = 0; j < m->nt && m->quad; j++) {   /* QuadMesh::buildBounds kernels/common/scene_quad_mesh.h:170-195: box of the four vertices */
      const uint32_t* q = m->t + 4 * (size_t)j;
      if (q[0] >= m->nv || q[1] >= m->nv || q[2] >= m->nv || q[3] >= m->nv) continue;
      const float* vv[4] = { m->v + 3 * (size_t)q[0], m->v + 3 * (size_t)q[1], m->v + 3 * (size_t)q[2], m->v + 3 * (size_t)q[3] };
      int ok = 1;
      for (int c = 0; c < 4; c++) for (int d = 0; d < 3; d++) ok &= valid_f(vv[c][d]);
      if (!ok) continue;
      PrimRef* p = &s->prims[k++];
      for (int d = 0; d < 3; d++) { p->lo[d] = fminf_(fminf_(vv[0][d], vv[1][d]), fminf_(vv[2][d], vv[3][d])); p->hi[d] = fmaxf_(fmaxf_(vv[0][d], vv[1][d]), fmaxf_(vv[2][d], vv[3][d])); }
      p->geomID = g; p->primID = j;
      float c2[3] = { p->lo[0] + p->hi[0], p->lo[1] + p->hi[1], p->lo[2] + p->hi[2] };
      box_extend(&all->geom, p->lo, p->hi); box_extend_pt(&all->cent, c2);
    }
    for (uint32_t j = 0; j < m->nt && !m->quad; j++) {
      const uint32_t* tri = m->t + 3 * (size_t)j;
      if (tri[0] >= m->nv || tri[1] >= m->nv || tri[2] >= m->nv) continue;
      const float *a = m->v + 3 * (size_t)tri[0], *b = m->v + 3 * (size_t)tri[1], *c = m->v + 3 * (size_t)tri[2];
      int ok = 1;
      for (int d = 0; d < 3; d++) ok &= valid_f(a[d]) & valid_f(b[d]) & valid_f(c[d]);
      if (!ok) continue;
      PrimRef* p = &s->prims[k++];
      for (int d = 0; d < 3; d++) { p->lo[d] = fminf_(fminf_(a[d], b[d]), c[d]); p->hi[d] = fmaxf_(fmaxf_(a[d], b[d]), c[d]); }
      p->geomID = g; p->primID = j;
      /* PrimInfo::add_center2: centroid proxy = lower+upper, never halved (priminfo.h:46-52) */
      float c2[3] = { p->lo[0] + p->hi[0], p->lo[1] + p->hi[1], p->lo[2] + p->hi[2] };
      box_extend(&all->geom, p->lo, p->hi); box_extend_pt(&all->cent, c2);
    }
  }
  s->nprims = k; all->begin = 0; all->end = k;
}

/* ------------------------------------------------------------ binned SAH */
#define BINS 32 /* NUM_OBJECT_BINS kernels/builders/bvh_builder_sah.h:10 */
typedef struct { size_t num; float ofs[3], scale[3]; } BinMapping;
typedef struct { float sah; int dim, pos; BinMapping map; } Split;

/* BinMapping(pinfo)  kernels/builders/heuristic_binning.h:46-55 */
static void mapping_init(BinMapping* m, const Set* set) {
  size_t n = set->end - set->begin;
  size_t num = (size_t)(4.0f + 0.05f * (float)n);
  m->num = num < BINS ? num : BINS;
  for (int d = 0; d < 3; d++) {
    float diag = fmaxf_(1E-34f, set->cent.hi[d] - set->cent.lo[d]);
    m->scale[d] = diag > 1E-34f ? (0.99f * (float)m->num) / diag : 0.0f;
    m->ofs[d] = set->cent.lo[d];
  }
}
/* BinMapping::bin (floor + clamp) :62-72 ; bin_unsafe (no clamp) :75-77 */
static int bin_unsafe(const BinMapping* m, const PrimRef* p, int d) {
  float c2 = p->lo[d] + p->hi[d];
  return (int)floorf((c2 - m->ofs[d]) * m->scale[d]);
}
static int bin_safe(const BinMapping* m, const PrimRef* p, int d) {
  int i = bin_unsafe(m, p, d);
  if (i < 0) i = 0;
  if (i > (int)m->num - 1) i = (int)m->num - 1;
  return i;
}
/* HeuristicArrayBinningSAH::find = BinInfoT::bin + ::best  heuristic_binning.h:210-257, 339-386 */
static Split sah_find(const Scene* s, const Set* set, int block_shift) {
  Split sp; mapping_init(&sp.map, set);
  const BinMapping* m = &sp.map;
  static __thread Box bb[BINS][3];
  static __thread uint32_t cnt[BINS][3];
  for (size_t i = 0; i < m->num; i++) for (int d = 0; d < 3; d++) { box_empty(&bb[i][d]); cnt[i][d] = 0; }
  for (size_t i = set->begin; i < set->end; i++) {
    const PrimRef* p = &s->prims[i];
    for (int d = 0; d < 3; d++) { int b = bin_safe(m, p, d); box_extend(&bb[b][d], p->lo, p->hi); cnt[b][d]++; }
  }
  float rA[BINS][3]; uint32_t rC[BINS][3];
  uint32_t c[3] = { 0, 0, 0 }; Box bx[3]; for (int d = 0; d < 3; d++) box_empty(&bx[d]);
  for (size_t i = m->num - 1; i > 0; i--)
    for (int d = 0; d < 3; d++) { c[d] += cnt[i][d]; rC[i][d] = c[d]; box_extend(&bx[d], bb[i][d].lo, bb[i][d].hi); rA[i][d] = half_area(&bx[d]); }
  uint32_t add = (1u << block_shift) - 1;
  float bestS[3] = { INFINITY, INFINITY, INFINITY }; int bestP[3] = { 0, 0, 0 };
  for (int d = 0; d < 3; d++) { c[d] = 0; box_empty(&bx[d]); }
  for (size_t i = 1; i < m->num; i++)
    for (int d = 0; d < 3; d++) {
      c[d] += cnt[i - 1][d]; box_extend(&bx[d], bb[i - 1][d].lo, bb[i - 1][d].hi);
      float lA = half_area(&bx[d]);
      uint32_t lN = (c[d] + add) >> block_shift, rN = (rC[i][d] + add) >> block_shift;
      float sah = fmaf(lA, (float)lN, rA[i][d] * (float)rN); /* madd(lArea,lCount,rArea*rCount) :367 */
      if (sah < bestS[d]) { bestS[d] = sah; bestP[d] = (int)i; }
    }
  sp.sah = INFINITY; sp.dim = -1; sp.pos = 0;
  for (int d = 0; d < 3; d++) {
    if (m->scale[d] == 0.0f) continue; /* mapping.invalid(dim) :375 */
    if (bestS[d] < sp.sah && bestP[d] != 0) { sp.dim = d; sp.pos = bestP[d]; sp.sah = bestS[d]; }
  }
  return sp;
}
static int cmp_id64(const void* a, const void* b) { /* PrimRef::operator<  primref.h:88-96 */
  const PrimRef *p = (const PrimRef*)a, *q = (const PrimRef*)b;
  uint64_t x = ((uint64_t)p->primID << 32) + p->geomID, y = ((uint64_t)q->primID << 32) + q->geomID;
  return x < y ? -1 : x > y;
}
static void deterministic_order(Scene* s, const Set* set) { qsort(s->prims + set->begin, set->end - set->begin, sizeof(PrimRef), cmp_id64); }
static void set_bounds(const Scene* s, Set* o, size_t b, size_t e) {
  o->begin = b; o->end = e; box_empty(&o->geom); box_empty(&o->cent);
  for (size_t i = b; i < e; i++) {
    const PrimRef* p = &s->prims[i];
    float c2[3] = { p->lo[0] + p->hi[0], p->lo[1] + p->hi[1], p->lo[2] + p->hi[2] };
    box_extend(&o->geom, p->lo, p->hi); box_extend_pt(&o->cent, c2);
  }
}
/* performFallbackSplit  heuristic_binning_array_aligned.h:50-65 */
static void split_fallback(const Scene* s, const Set* set, Set* l, Set* r) {
  size_t center = (set->begin + set->end) / 2;
  set_bounds(s, l, set->begin, center); set_bounds(s, r, center, set->end);
}
/* split_template  heuristic_binning_array_aligned.h:141-176 (serial_partitioning: in-place two-pointer) */
static void sah_split(Scene* s, const Split* sp, const Set* set, Set* l, Set* r) {
  if (sp->dim == -1) { deterministic_order(s, set); split_fallback(s, set, l, r); return; }
  size_t i = set->begin, j = set->end;
  while (1) {
    while (i < j && bin_unsafe(&sp->map, &s->prims[i], sp->dim) < sp->pos) i++;
    while (i < j && !(bin_unsafe(&sp->map, &s->prims[j - 1], sp->dim) < sp->pos)) j--;
    if (i >= j) break;
    PrimRef t = s->prims[i]; s->prims[i] = s->prims[j - 1]; s->prims[j - 1] = t;
    i++; j--;
  }
  set_bounds(s, l, set->begin, i); set_bounds(s, r, i, set->end);
}

/* ---------------------------------------------------------- node / leaf */
static int32_t new_node(Scene* s) {
  if (s->nnodes == s->cnodes) { s->cnodes = s->cnodes ? 2 * s->cnodes : 1024; s->nodes = (Node8*)realloc(s->nodes, s->cnodes * sizeof(Node8)); }
  Node8* n = &s->nodes[s->nnodes];
  for (int i = 0; i < 8; i++) { /* AABBNode::clear: empty bounds + emptyNode  bvh_node_aabb.h:84-91 */
    for (int d = 0; d < 3; d++) { n->lower[d][i] = INFINITY; n->upper[d][i] = -INFINITY; }
    n->child[i] = EMPTY_REF;
  }
  return (int32_t)s->nnodes++;
}
/* CreateLeaf kernels/bvh/bvh_builder_sah.cpp:32-55 + TriangleM::fill kernels/geometry/triangle.h:98-120, ctor :40-41 */
static int32_t create_leaf(Scene* s, const Set* set) {
  size_t n = set->end - set->begin, items = (n + 3) / 4, start = s->nblocks;
  if (s->nblocks + items > s->cblocks) { s->cblocks = 2 * s->cblocks + items + 1024; s->blocks = (Tri4*)realloc(s->blocks, s->cblocks * sizeof(Tri4)); }
  size_t b = set->begin;
  for (size_t k = 0; k < items; k++) {
    Tri4* t = &s->blocks[s->nblocks++];
    memset(t, 0, sizeof(*t));
    for (int i = 0; i < 4; i++) { t->geomID[i] = INVALID_ID; t->primID[i] = INVALID_ID; }
    for (int i = 0; i < 4 && b < set->end; i++, b++) {
      const PrimRef* p = &s->prims[b];
      const Mesh* m = &s->mesh[p->geomID];
      if (m->quad) {                                            /* Quad4v::fill kernels/geometry/quadv.h */
        const uint32_t* q = m->t + 4 * (size_t)p->primID;
        for (int d = 0; d < 3; d++) { t->v0[d][i] = m->v[3 * (size_t)q[0] + d]; t->e1[d][i] = m->v[3 * (size_t)q[1] + d]; t->e2[d][i] = m->v[3 * (size_t)q[2] + d]; t->v3[d][i] = m->v[3 * (size_t)q[3] + d]; }
        t->geomID[i] = p->geomID; t->primID[i] = p->primID;
        continue;
      }
      const uint32_t* tri = m->t + 3 * (size_t)p->primID;
      const float *v0 = m->v + 3 * (size_t)tri[0], *v1 = m->v + 3 * (size_t)tri[1], *v2 = m->v + 3 * (size_t)tri[2];
      if (s->robust) for (int d = 0; d < 3; d++) { t->v0[d][i] = v0[d]; t->e1[d][i] = v1[d]; t->e2[d][i] = v2[d]; }   /* TriangleMv: the vertices themselves, kernels/geometry/trianglev.h */
      else for (int d = 0; d < 3; d++) { t->v0[d][i] = v0[d]; t->e1[d][i] = v0[d] - v1[d]; t->e2[d][i] = v2[d] - v0[d]; }
      t->geomID[i] = p->geomID; t->primID[i] = p->primID;
    }
  }
  return enc_leaf(start, items);
}

/* Settings: BVH8Triangle4SceneBuilderSAH = (sahBlockSize 4, intCost 1, minLeaf 4, maxLeaf inf -> 4*7)  bvh_builder_sah.cpp:467;
   branchingFactor 8, maxDepth 40  bvh_builder.cpp:18-19; travCost 1  builders/bvh_builder_sah.h:17 */
enum { LOG_BLOCK = 2, MIN_LEAF = 4, MAX_LEAF = 28, BRANCH = 8, MAX_DEPTH = 40, MIN_LARGE_LEAF_LEVELS = 8 };

static void set_child(Scene* s, int32_t node, int i, const Set* c, int32_t ref) {
  Node8* n = &s->nodes[node];
  for (int d = 0; d < 3; d++) { n->lower[d][i] = c->geom.lo[d]; n->upper[d][i] = c->geom.hi[d]; }
  n->child[i] = ref;
}
/* BuilderT::createLargeLeaf  builders/bvh_builder_sah.h:149-212 */
static int32_t create_large_leaf(Scene* s, const Set* cur, int depth) {
  size_t n = cur->end - cur->begin;
  if (n <= MAX_LEAF) return create_leaf(s, cur);
  Set ch[BRANCH]; size_t nch = 1; ch[0] = *cur;
  do {
    size_t best = (size_t)-1, bestSize = 0;
    for (size_t i = 0; i < nch; i++) {
      size_t sz = ch[i].end - ch[i].begin;
      if (sz <= MAX_LEAF) continue;
      if (sz > bestSize) { bestSize = sz; best = i; }
    }
    if (best == (size_t)-1) break;
    Set l, r; split_fallback(s, &ch[best], &l, &r);
    ch[best] = ch[nch - 1]; ch[nch - 1] = l; ch[nch] = r; nch++;
  } while (nch < BRANCH);
  int32_t node = new_node(s);
  for (size_t i = 0; i < nch; i++) { int32_t ref = create_large_leaf(s, &ch[i], depth + 1); set_child(s, node, (int)i, &ch[i], ref); }
  return node;
}
static int cmp_set_desc(const void* a, const void* b) { /* std::greater<BuildRecord> by size  :275 */
  size_t x = ((const Set*)a)->end - ((const Set*)a)->begin, y = ((const Set*)b)->end - ((const Set*)b)->begin;
  return x > y ? -1 : x < y;
}
/* BuilderT::recurse  builders/bvh_builder_sah.h:214-308 */
static int32_t recurse(Scene* s, Set* cur, int depth) {
  size_t n = cur->end - cur->begin;
  Split sp = sah_find(s, cur, LOG_BLOCK);
  float A = half_area(&cur->geom);
  float leafSAH = 1.0f * (A * (float)((n + 3) >> 2));   /* intCost*leafSAH(logBlockSize)  priminfo.h:149-151 */
  float splitSAH = 1.0f * A + 1.0f * sp.sah;
  if (n <= MIN_LEAF || depth + MIN_LARGE_LEAF_LEVELS >= MAX_DEPTH || (n <= MAX_LEAF && leafSAH <= splitSAH)) {
    deterministic_order(s, cur);
    return create_large_leaf(s, cur, depth);
  }
  Set ch[BRANCH]; size_t nch = 2;
  sah_split(s, &sp, cur, &ch[0], &ch[1]);
  while (nch < BRANCH) {
    float bestArea = -INFINITY; int best = -1;
    for (size_t i = 0; i < nch; i++) {
      if (ch[i].end - ch[i].begin <= MIN_LEAF) continue;
      float a = half_area(&ch[i].geom);
      if (a > bestArea) { best = (int)i; bestArea = a; }
    }
    if (best == -1) break;
    Split sp2 = sah_find(s, &ch[best], LOG_BLOCK);
    Set l, r; sah_split(s, &sp2, &ch[best], &l, &r);
    ch[best] = l; ch[nch] = r; nch++;
  }
  qsort(ch, nch, sizeof(Set), cmp_set_desc);
  int32_t node = new_node(s);
  for (size_t i = 0; i < nch; i++) { int32_t ref = recurse(s, &ch[i], depth + 1); set_child(s, node, (int)i, &ch[i], ref); }
  return node;
}

/* BVHNBuilderSAH::build  kernels/bvh/bvh_builder_sah.cpp:112-193 */
API void ora_commit(Scene* s) {
  free_build(s);
  Set all; gen_primrefs(s, &all);
  s->bounds = all.geom;
  if (s->nprims == 0) { s->root = EMPTY_REF; return; }
  s->root = recurse(s, &all, 1);
}
API void ora_bounds(const Scene* s, float* lo3hi3) { memcpy(lo3hi3, &s->bounds, 24); }
API void ora_counts(const Scene* s, uint64_t* out4) { out4[0] = s->nprims; out4[1] = s->nnodes; out4[2] = s->nblocks; out4[3] = (uint64_t)(s->root >= 0); }
API void ora_visit_stats(Scene* s, uint64_t* out3, int reset) {
  out3[0] = s->stat_nodes; out3[1] = s->stat_leaves; out3[2] = s->stat_blocks;
  if (reset) s->stat_nodes = s->stat_leaves = s->stat_blocks = 0;
}

/* --------------------------------------------------------------- traversal */
typedef struct { float org[3], dir[3], rdir[3], org_rdir[3], rdir_near[3], rdir_far[3]; int nearIsUpper[3]; float tnear, tfar; } TravRay;
/* TravRayBase<N,false>  kernels/bvh/node_intersector1.h:29-57; rcp_safe  common/math/vec3fa.h:167-172 */
static void travray_init(TravRay* t, const Ray* r) {
  const float o[3] = { r->org_x, r->org_y, r->org_z }, d[3] = { r->dir_x, r->dir_y, r->dir_z };
  for (int k = 0; k < 3; k++) {
    t->org[k] = o[k]; t->dir[k] = d[k];
    float z = fabsf(d[k]) < 1E-18f ? 1E-18f : d[k]; /* zero_fix: +min_rcp_input regardless of sign */
    t->rdir[k] = rcp_nr(z);
    t->org_rdir[k] = o[k] * t->rdir[k];
    t->nearIsUpper[k] = !(t->rdir[k] >= 0.0f);
  }
  t->tnear = fmaxf_(r->tnear, 0.0f); t->tfar = fmaxf_(r->tfar, 0.0f); /* bvh_intersector1.cpp:65 */
}
/* TravRayBase<N,true>  node_intersector1.h:98-121: rdir = 1/zero_fix(dir) (a true division), rdir_near/far = rdir * (1 -+ 3 ulp) */
static void travray_init_robust(TravRay* t, const Ray* r) {
  const float o[3] = { r->org_x, r->org_y, r->org_z }, d[3] = { r->dir_x, r->dir_y, r->dir_z };
  const float round_down = 1.0f - 3.0f * FLT_EPSILON, round_up = 1.0f + 3.0f * FLT_EPSILON;
  for (int k = 0; k < 3; k++) {
    t->org[k] = o[k]; t->dir[k] = d[k];
    float z = fabsf(d[k]) < 1E-18f ? 1E-18f : d[k];
    float rd = 1.0f / z;
    t->rdir[k] = rd; t->rdir_near[k] = round_down * rd; t->rdir_far[k] = round_up * rd;
    t->nearIsUpper[k] = !(t->rdir_near[k] >= 0.0f);
  }
  t->tnear = fmaxf_(r->tnear, 0.0f); t->tfar = fmaxf_(r->tfar, 0.0f);
}
/* intersectNodeRobust  node_intersector1.h:539-554: (plane - org) * rdir_near|far */
static unsigned node_test_robust(const Node8* n, const TravRay* t, float dist[8]) {
  unsigned mask = 0;
  for (int i = 0; i < 8; i++) {
    float tn[3], tf[3];
    for (int k = 0; k < 3; k++) {
      float pn = t->nearIsUpper[k] ? n->upper[k][i] : n->lower[k][i];
      float pf = t->nearIsUpper[k] ? n->lower[k][i] : n->upper[k][i];
      tn[k] = (pn - t->org[k]) * t->rdir_near[k];
      tf[k] = (pf - t->org[k]) * t->rdir_far[k];
    }
    float tNear = fmaxf_(fmaxf_(tn[0], tn[1]), fmaxf_(tn[2], t->tnear));
    float tFar = fminf_(fminf_(tf[0], tf[1]), fminf_(tf[2], t->tfar));
    dist[i] = tNear;
    if (tNear <= tFar) mask |= 1u << i;
  }
  return mask;
}
/* intersectNode<8> AVX2 branch  node_intersector1.h:484-531: t = msub(plane, rdir, org_rdir) */
static unsigned node_test(const Node8* n, const TravRay* t, float dist[8]) {
  unsigned mask = 0;
  for (int i = 0; i < 8; i++) {
    float tn[3], tf[3];
    for (int k = 0; k < 3; k++) {
      float pn = t->nearIsUpper[k] ? n->upper[k][i] : n->lower[k][i];
      float pf = t->nearIsUpper[k] ? n->lower[k][i] : n->upper[k][i];
      tn[k] = fmaf(pn, t->rdir[k], -t->org_rdir[k]);
      tf[k] = fmaf(pf, t->rdir[k], -t->org_rdir[k]);
    }
    float tNear = fmaxf_(fmaxf_(tn[0], tn[1]), fmaxf_(tn[2], t->tnear));
    float tFar = fminf_(fminf_(tf[0], tf[1]), fminf_(tf[2], t->tfar));
    dist[i] = tNear;
    if (tNear <= tFar) mask |= 1u << i;
  }
  return mask;
}

/* MoellerTrumboreIntersector1<4>::intersect  kernels/geometry/triangle_intersector_moeller.h:69-111 on one lane */
typedef struct { int valid; float T, U, V, absDen, Ng[3]; } MTHit;
static void mt_lane(const Tri4* b, int i, const Ray* ray, MTHit* h) {
  const float v0[3] = { b->v0[0][i], b->v0[1][i], b->v0[2][i] };
  const float e1[3] = { b->e1[0][i], b->e1[1][i], b->e1[2][i] };
  const float e2[3] = { b->e2[0][i], b->e2[1][i], b->e2[2][i] };
  const float O[3] = { ray->org_x, ray->org_y, ray->org_z }, D[3] = { ray->dir_x, ray->dir_y, ray->dir_z };
  /* Ng = cross(e2,e1); cross(a,b) = (msub(a.y,b.z,a.z*b.y), msub(a.z,b.x,a.x*b.z), msub(a.x,b.y,a.y*b.x))  common/math/vec3.h:209 */
  float Ng[3] = { fmaf(e2[1], e1[2], -(e2[2] * e1[1])), fmaf(e2[2], e1[0], -(e2[0] * e1[2])), fmaf(e2[0], e1[1], -(e2[1] * e1[0])) };
  float C[3] = { v0[0] - O[0], v0[1] - O[1], v0[2] - O[2] };
  float R[3] = { fmaf(C[1], D[2], -(C[2] * D[1])), fmaf(C[2], D[0], -(C[0] * D[2])), fmaf(C[0], D[1], -(C[1] * D[0])) };
  /* dot(a,b) = madd(a.x,b.x,madd(a.y,b.y,a.z*b.z))  vec3.h:204 */
  float den = fmaf(Ng[0], D[0], fmaf(Ng[1], D[1], Ng[2] * D[2]));
  float absDen = fabsf(den);
  uint32_t sgn = fbits(den) & 0x80000000u;
  float U = xorf(fmaf(R[0], e2[0], fmaf(R[1], e2[1], R[2] * e2[2])), sgn);
  float V = xorf(fmaf(R[0], e1[0], fmaf(R[1], e1[1], R[2] * e1[2])), sgn);
  h->valid = (den != 0.0f) && (U >= 0.0f) && (V >= 0.0f) && (U + V <= absDen);
  float T = xorf(fmaf(Ng[0], C[0], fmaf(Ng[1], C[1], Ng[2] * C[2])), sgn);
  h->valid = h->valid && (absDen * ray->tnear < T) && (T <= absDen * ray->tfar);
  h->T = T; h->U = U; h->V = V; h->absDen = absDen; h->Ng[0] = Ng[0]; h->Ng[1] = Ng[1]; h->Ng[2] = Ng[2];
}

/* PlueckerIntersector1<4>::intersect  kernels/geometry/triangle_intersector_pluecker.h:68-118 on one lane of a TriangleMv block
   (fields e1/e2 of Tri4 hold v1/v2 in robust scenes); PlueckerHitM::finalize :26-33.  h->T carries t itself, h->absDen = 1. */
typedef struct { int valid; float t, u, v, Ng[3]; } PLHit;
static void cross3(const float* a, const float* b, float* o) { o[0] = fmaf(a[1], b[2], -(a[2] * b[1])); o[1] = fmaf(a[2], b[0], -(a[0] * b[2])); o[2] = fmaf(a[0], b[1], -(a[1] * b[0])); }
static float dot3(const float* a, const float* b) { return fmaf(a[0], b[0], fmaf(a[1], b[1], a[2] * b[2])); }
static void pl_lane(const Tri4* b, int i, const Ray* ray, PLHit* h) {
  const float O[3] = { ray->org_x, ray->org_y, ray->org_z }, D[3] = { ray->dir_x, ray->dir_y, ray->dir_z };
  float v0[3], v1[3], v2[3], e0[3], e1[3], e2[3], s0[3], s1[3], s2[3], c[3];
  for (int d = 0; d < 3; d++) { v0[d] = b->v0[d][i] - O[d]; v1[d] = b->e1[d][i] - O[d]; v2[d] = b->e2[d][i] - O[d]; }
  for (int d = 0; d < 3; d++) { e0[d] = v2[d] - v0[d]; e1[d] = v0[d] - v1[d]; e2[d] = v1[d] - v2[d]; s0[d] = v2[d] + v0[d]; s1[d] = v0[d] + v1[d]; s2[d] = v1[d] + v2[d]; }
  cross3(e0, s0, c); const float U = dot3(c, D);
  cross3(e1, s1, c); const float V = dot3(c, D);
  cross3(e2, s2, c); const float W = dot3(c, D);
  const float UVW = (U + V) + W;
  const float eps = FLT_EPSILON * fabsf(UVW);
  const float mn = fminf_(fminf_(U, V), W), mx = fmaxf_(fmaxf_(U, V), W);
  int valid = (mn >= -eps) || (mx <= eps);
  /* stable_triangle_normal(e0,e1,e2)  common/math/vec3.h:210-222 */
  const float ab_x = e0[2] * e1[1], ab_y = e0[0] * e1[2], ab_z = e0[1] * e1[0];
  const float bc_x = e1[2] * e2[1], bc_y = e1[0] * e2[2], bc_z = e1[1] * e2[0];
  const float cab[3] = { fmaf(e0[1], e1[2], -ab_x), fmaf(e0[2], e1[0], -ab_y), fmaf(e0[0], e1[1], -ab_z) };
  const float cbc[3] = { fmaf(e1[1], e2[2], -bc_x), fmaf(e1[2], e2[0], -bc_y), fmaf(e1[0], e2[1], -bc_z) };
  const float Ng[3] = { fabsf(ab_x) < fabsf(bc_x) ? cab[0] : cbc[0], fabsf(ab_y) < fabsf(bc_y) ? cab[1] : cbc[1], fabsf(ab_z) < fabsf(bc_z) ? cab[2] : cbc[2] };
  const float dn = dot3(Ng, D), den = dn + dn;
  const float tt = dot3(v0, Ng), T = tt + tt;
  const float t = rcp_nr(den) * T;
  valid = valid && (ray->tnear <= t) && (t <= ray->tfar) && (den != 0.0f);
  const float rcpUVW = fabsf(UVW) < 1E-18f ? 0.0f : rcp_nr(UVW);
  h->valid = valid; h->t = t; h->u = fminf_(U * rcpUVW, 1.0f); h->v = fminf_(V * rcpUVW, 1.0f);
  h->Ng[0] = Ng[0]; h->Ng[1] = Ng[1]; h->Ng[2] = Ng[2];
}

/* QuadMIntersector1MoellerTrumbore<4> / QuadMIntersector1Pluecker<4>, AVX specialisations (kernels/geometry/quad_intersector_moeller.h:179-216,
   quad_intersector_pluecker.h:198-217): eight lanes = triangles (v0,v1,v3) of the four quads, then triangles (v2,v1,v3) with flag set;
   flagged lanes: U,V <- absDen-V, absDen-U and Ng <- -Ng (fast), u,v <- 1-v, 1-u and Ng <- -Ng (robust, QuadHitPlueckerM::finalize :33-50).
   Fills t/u/v/Ng/valid for lane l (0..7) of block b. */
typedef struct { int valid; float t, u, v, Ng[3]; } QLane;
static void quad_lane(const Scene* s, const Tri4* b, int l, const Ray* ray, QLane* o) {
  const int i = l & 3, flag = l >> 2;
  Tri4 tmp; memset(&tmp, 0, sizeof(tmp));
  const float (*A)[4] = flag ? b->e2 : b->v0;                   /* first vertex: v0 or v2 */
  if (s->robust) {
    for (int d = 0; d < 3; d++) { tmp.v0[d][0] = A[d][i]; tmp.e1[d][0] = b->e1[d][i]; tmp.e2[d][0] = b->v3[d][i]; }
    PLHit p; pl_lane(&tmp, 0, ray, &p);
    o->valid = p.valid; o->t = p.t;
    o->u = flag ? 1.0f - p.v : p.u; o->v = flag ? 1.0f - p.u : p.v;
    for (int d = 0; d < 3; d++) o->Ng[d] = flag ? -1.0f * p.Ng[d] : 1.0f * p.Ng[d];
  } else {
    for (int d = 0; d < 3; d++) { tmp.v0[d][0] = A[d][i]; tmp.e1[d][0] = A[d][i] - b->e1[d][i]; tmp.e2[d][0] = b->v3[d][i] - A[d][i]; }   /* e1 = v0-v1, e2 = v2-v0 */
    MTHit h; mt_lane(&tmp, 0, ray, &h);
    const float U = flag ? h.absDen - h.V : h.U, V = flag ? h.absDen - h.U : h.V;
    const float r = rcp_nr(h.absDen);
    o->valid = h.valid; o->t = h.T * r; o->u = U * r; o->v = V * r;
    for (int d = 0; d < 3; d++) o->Ng[d] = flag ? h.Ng[d] * -1.0f : h.Ng[d] * 1.0f;
  }
  if (b->geomID[i] == INVALID_ID) o->valid = 0;
}
static void quad_leaf_intersect(Scene* s, RayHit* rh, size_t start, size_t num) {
  for (size_t k = 0; k < num; k++) {
    const Tri4* b = &s->blocks[start + k];
    s->stat_blocks++;
    QLane q[8]; int valid[8], any = 0;
    for (int l = 0; l < 8; l++) { quad_lane(s, b, l, &rh->ray, &q[l]); valid[l] = q[l].valid; any |= valid[l]; }
    if (!any) continue;
    for (;;) {                                                  /* Intersect1EpilogM<8,true>: select_min over the 8 lanes, lowest lane on equal t */
      int best = -1; float bt = INFINITY;
      for (int l = 0; l < 8; l++) if (valid[l] && q[l].t < bt) bt = q[l].t;
      for (int l = 0; l < 8; l++) if (valid[l] && q[l].t == bt) { best = l; break; }
      if (best < 0) for (int l = 0; l < 8; l++) if (valid[l]) { best = l; break; }
      if (best < 0) break;
      const uint32_t g = b->geomID[best & 3];
      if ((s->mesh[g].mask & rh->ray.mask) == 0) { valid[best] = 0; continue; }
      rh->ray.tfar = q[best].t;
      rh->hit.Ng_x = q[best].Ng[0]; rh->hit.Ng_y = q[best].Ng[1]; rh->hit.Ng_z = q[best].Ng[2];
      rh->hit.u = q[best].u; rh->hit.v = q[best].v;
      rh->hit.primID = b->primID[best & 3]; rh->hit.geomID = g;
      rh->hit.instID = s->ctxInstID; rh->hit.instPrimID = s->ctxInstPrimID;
      break;
    }
  }
}
static int quad_leaf_occluded(Scene* s, const Ray* ray, size_t start, size_t num) {
  for (size_t k = 0; k < num; k++) {
    const Tri4* b = &s->blocks[start + k];
    s->stat_blocks++;
    for (int l = 0; l < 8; l++) {
      QLane q; quad_lane(s, b, l, ray, &q);
      if (q.valid && (s->mesh[b->geomID[l & 3]].mask & ray->mask) != 0) return 1;
    }
  }
  return 0;
}

/* ArrayIntersector1::intersect (kernels/geometry/intersector_iterators.h:23-28) over the leaf's blocks,
   each block = MoellerTrumbore x4 + Intersect1EpilogM<4,true> (kernels/geometry/intersector_epilog.h:235-300) */
static void leaf_intersect(Scene* s, RayHit* rh, size_t start, size_t num) {
  for (size_t k = 0; k < num; k++) {
    const Tri4* b = &s->blocks[start + k];
    s->stat_blocks++;
    MTHit h[4]; float t[4], u[4], v[4]; int valid[4], any = 0;
    if (s->robust) {
      for (int i = 0; i < 4; i++) {
        PLHit p; pl_lane(b, i, &rh->ray, &p);
        valid[i] = p.valid; any |= valid[i]; t[i] = p.t; u[i] = p.u; v[i] = p.v;
        h[i].Ng[0] = p.Ng[0]; h[i].Ng[1] = p.Ng[1]; h[i].Ng[2] = p.Ng[2];
      }
      if (!any) continue;
    } else {
    for (int i = 0; i < 4; i++) { mt_lane(b, i, &rh->ray, &h[i]); valid[i] = h[i].valid; any |= valid[i]; }
    if (!any) continue;
    for (int i = 0; i < 4; i++) { /* finalize(): t,u,v = T,U,V * rcp(absDen)  triangle_intersector_moeller.h:29-36 */
      float r = rcp_nr(h[i].absDen);
      t[i] = h[i].T * r; u[i] = h[i].U * r; v[i] = h[i].V * r;
    }
    }
    for (;;) {
      /* select_min(valid, vt): lowest lane among the minimum  common/simd/vfloat4_sse2.h:759 */
      int best = -1; float bt = INFINITY;
      for (int i = 0; i < 4; i++) if (valid[i] && t[i] < bt) { bt = t[i]; }
      for (int i = 0; i < 4; i++) if (valid[i] && t[i] == bt) { best = i; break; }
      if (best < 0) for (int i = 0; i < 4; i++) if (valid[i]) { best = i; break; } /* any(valid_min) ? valid_min : valid */
      if (best < 0) break;
      uint32_t g = b->geomID[best];
      if ((s->mesh[g].mask & rh->ray.mask) == 0) { valid[best] = 0; continue; } /* EMBREE_RAY_MASK :256-262 */
      rh->ray.tfar = t[best];
      rh->hit.Ng_x = h[best].Ng[0]; rh->hit.Ng_y = h[best].Ng[1]; rh->hit.Ng_z = h[best].Ng[2];
      rh->hit.u = u[best]; rh->hit.v = v[best];
      rh->hit.primID = b->primID[best]; rh->hit.geomID = g;
      rh->hit.instID = s->ctxInstID; rh->hit.instPrimID = s->ctxInstPrimID; /* copy of the default context, :295-298 */
      break;
    }
  }
}
/* Occluded1EpilogM  intersector_epilog.h:304-368 */
static int leaf_occluded(Scene* s, const Ray* ray, size_t start, size_t num) {
  for (size_t k = 0; k < num; k++) {
    const Tri4* b = &s->blocks[start + k];
    s->stat_blocks++;
    for (int i = 0; i < 4; i++) {
      int valid;
      if (s->robust) { PLHit p; pl_lane(b, i, ray, &p); valid = p.valid && b->geomID[i] != INVALID_ID; }
      else { MTHit h; mt_lane(b, i, ray, &h); valid = h.valid; }
      if (valid && (s->mesh[b->geomID[i]].mask & ray->mask) != 0) return 1;
    }
  }
  return 0;
}

typedef struct { int32_t ref; uint32_t dist; } StackItem; /* StackItemT  kernels/common/stack_item.h:11-109 */
#define STACK 600 /* 1+(N-1)*maxDepth+3  bvh_intersector1.h:26 */

/* BVHNIntersector1::intersect  kernels/bvh/bvh_intersector1.cpp:32-114 */
static void intersect1(Scene* s, RayHit* rh) {
  if (s->root == EMPTY_REF) return;
  TravRay tr; if (s->robust) travray_init_robust(&tr, &rh->ray); else travray_init(&tr, &rh->ray);
  StackItem stack[STACK]; int sp = 1;
  stack[0].ref = s->root; stack[0].dist = fbits(-INFINITY);
  while (sp > 0) {
    sp--;
    int32_t cur = stack[sp].ref;
    float d; memcpy(&d, &stack[sp].dist, 4);
    if (d > rh->ray.tfar) continue;
    int popped = 0;
    while (cur >= 0) {
      const Node8* n = &s->nodes[cur];
      float dist[8];
      s->stat_nodes++;
      unsigned mask = s->robust ? node_test_robust(n, &tr, dist) : node_test(n, &tr, dist);
      if (mask == 0) { popped = 1; break; }
      /* traverseClosestHit  kernels/bvh/bvh_traverser1.h:311-433: nearest child next, rest pushed far -> near */
      StackItem hit[8]; int nh = 0;
      for (int i = 0; i < 8; i++) if (mask & (1u << i)) { hit[nh].ref = n->child[i]; hit[nh].dist = fbits(dist[i]); nh++; }
      if (nh == 1) { cur = hit[0].ref; continue; }
      if (nh == 2) { /* :337-341: d0<d1 ? (push c1, go c0) : (push c0, go c1); unsigned compare of the float bits */
        if (hit[0].dist < hit[1].dist) { stack[sp++] = hit[1]; cur = hit[0].ref; }
        else { stack[sp++] = hit[0]; cur = hit[1].ref; }
        continue;
      }
      /* >=3: sort so that the closest ends on top (stack_item.h sort3/sort4/sort: descending dist bottom->top) */
      for (int a = 1; a < nh; a++) { StackItem x = hit[a]; int b2 = a - 1; while (b2 >= 0 && hit[b2].dist < x.dist) { hit[b2 + 1] = hit[b2]; b2--; } hit[b2 + 1] = x; }
      for (int a = 0; a < nh - 1; a++) stack[sp++] = hit[a];
      cur = hit[nh - 1].ref;
    }
    if (popped) continue;
    if (cur == EMPTY_REF) continue;
    size_t start, num; dec_leaf(cur, &start, &num);
    s->stat_leaves++;
    if (s->quads) quad_leaf_intersect(s, rh, start, num); else leaf_intersect(s, rh, start, num);
    tr.tfar = rh->ray.tfar; /* :105 */
  }
}
/* BVHNIntersector1::occluded  bvh_intersector1.cpp:117-197 + traverseAnyHit bvh_traverser1.h:435-463 */
static void occluded1(Scene* s, Ray* ray) {
  if (s->root == EMPTY_REF) return;
  if (ray->tfar < 0.0f) return;
  TravRay tr; if (s->robust) travray_init_robust(&tr, ray); else travray_init(&tr, ray);
  int32_t stack[STACK]; int sp = 1; stack[0] = s->root;
  while (sp > 0) {
    int32_t cur = stack[--sp];
    int popped = 0;
    while (cur >= 0) {
      const Node8* n = &s->nodes[cur];
      float dist[8];
      s->stat_nodes++;
      unsigned mask = s->robust ? node_test_robust(n, &tr, dist) : node_test(n, &tr, dist);
      if (mask == 0) { popped = 1; break; }
      int last = -1;
      for (int i = 0; i < 8; i++) if (mask & (1u << i)) { if (last >= 0) stack[sp++] = n->child[last]; last = i; }
      cur = n->child[last];
    }
    if (popped) continue;
    if (cur == EMPTY_REF) continue;
    size_t start, num; dec_leaf(cur, &start, &num);
    s->stat_leaves++;
    if (s->quads ? quad_leaf_occluded(s, ray, start, num) : leaf_occluded(s, ray, start, num)) { ray->tfar = -INFINITY; break; }
  }
}

/* API wrappers: rtcIntersect1 / rtcOccluded1 looped over an AoS array (kernels/common/rtcore.cpp:599,918) */
API void ora_intersect1(Scene* s, RayHit* rh, uint32_t M) { for (uint32_t i = 0; i < M; i++) intersect1(s, &rh[i]); }
API void ora_occluded1(Scene* s, Ray* r, uint32_t M) { for (uint32_t i = 0; i < M; i++) occluded1(s, &r[i]); }

/* ------------------------------------------------------------------ instances (RTC_GEOMETRY_TYPE_INSTANCE, one level)
 * Instance::setTransform / commit (kernels/common/scene_instance.cpp:105,150-153): world2local0 = rcp(local2world[0]);
 * rcp(AffineSpace) = (il = rcp(l), -(il * p)) (common/math/affinespace.h:83); LinearSpace3::inverse = adjoint() / det()
 * (linearspace3.h:44-51), cross = msub(a0,b0,a1*b1) shuffled (vec3fa.h:334-341), dot = DPPS 0x7F (:325-327).
 * InstanceIntersector1::intersect / occluded (kernels/geometry/instance_intersector.cpp:15-68): ray mask test, push the instance id
 * (fails when a level is already open: RTC_MAX_INSTANCE_LEVEL_COUNT = 1), org' = xfmPoint(w2l, org), dir' = xfmVector(w2l, dir)
 * (affinespace.h:102-103, linearspace3.h:159: nested madd), the object's accels are queried with the transformed ray (tnear, tfar and
 * therefore every t are unchanged), org/dir restored.  The hit keeps the object-space Ng and gets instID[0] / instPrimID[0] = id / 0.
 * The reference finds the instances whose bounds a ray enters with a BVH4 over xfmBounds(local2world, object bounds)
 * (Instance::bounds, scene_instance.h:64-69); here they are visited in id order with the same box test as a cull -- the set of
 * candidate hits is the same, only the order (= which of two hits at the identical t is kept) can differ. */
typedef struct { Scene* tri; Scene* quad; float l2w[12], w2l[12]; uint32_t mask, id; Box bounds; int valid; } Inst;
typedef struct { Inst* inst; uint32_t n, cap; Box bounds; } InstSet;

static void cross3f(const float* a, const float* b, float* o) {     /* vec3fa.h:334-341 */
  o[0] = fmaf(a[1], b[2], -(a[2] * b[1])); o[1] = fmaf(a[2], b[0], -(a[0] * b[2])); o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
static float dpps3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + 0.0f); }   /* _mm_dp_ps(a,b,0x7F) */
static void affine_rcp(const float* m, float* o) {
  const float *vx = m, *vy = m + 3, *vz = m + 6, *p = m + 9;
  float c0[3], c1[3], c2[3];
  cross3f(vy, vz, c0); cross3f(vz, vx, c1); cross3f(vx, vy, c2);   /* adjoint = (c0, c1, c2) transposed */
  const float det = dpps3(vx, c0);
  float il[9];                                                     /* columns of the inverse */
  for (int r = 0; r < 3; r++) { il[0 + r] = (r == 0 ? c0[0] : r == 1 ? c1[0] : c2[0]) / det; il[3 + r] = (r == 0 ? c0[1] : r == 1 ? c1[1] : c2[1]) / det; il[6 + r] = (r == 0 ? c0[2] : r == 1 ? c1[2] : c2[2]) / det; }
  memcpy(o, il, 36);
  for (int r = 0; r < 3; r++) o[9 + r] = -fmaf(p[0], il[0 + r], fmaf(p[1], il[3 + r], p[2] * il[6 + r]));   /* -(il * p), linearspace3.h:150 */
}
static void xfm_point(const float* m, const float* p, float* o) { for (int r = 0; r < 3; r++) o[r] = fmaf(p[0], m[0 + r], fmaf(p[1], m[3 + r], fmaf(p[2], m[6 + r], m[9 + r]))); }
static void xfm_vector(const float* m, const float* v, float* o) { for (int r = 0; r < 3; r++) o[r] = fmaf(v[0], m[0 + r], fmaf(v[1], m[3 + r], v[2] * m[6 + r])); }

API InstSet* ora_inst_new(void) { InstSet* s = (InstSet*)calloc(1, sizeof(InstSet)); box_empty(&s->bounds); return s; }
API void ora_inst_free(InstSet* s) { if (s) { free(s->inst); free(s); } }
/* tri / quad: the object scene's two accels (either may be NULL), already committed; l2w: vx, vy, vz, p */
API void ora_inst_add(InstSet* s, Scene* tri, Scene* quad, const float* l2w, uint32_t mask, uint32_t id) {
  if (s->n == s->cap) { s->cap = s->cap ? 2 * s->cap : 16; s->inst = (Inst*)realloc(s->inst, s->cap * sizeof(Inst)); }
  Inst* I = &s->inst[s->n++];
  I->tri = tri; I->quad = quad; I->mask = mask; I->id = id;
  memcpy(I->l2w, l2w, 48); affine_rcp(l2w, I->w2l);
  Box ob; box_empty(&ob);
  if (tri && tri->nprims) box_extend(&ob, tri->bounds.lo, tri->bounds.hi);
  if (quad && quad->nprims) box_extend(&ob, quad->bounds.lo, quad->bounds.hi);
  box_empty(&I->bounds);
  for (int c = 0; c < 8; c++) {                                   /* xfmBounds, affinespace.h:106-118 */
    const float p[3] = { (c & 4) ? ob.hi[0] : ob.lo[0], (c & 2) ? ob.hi[1] : ob.lo[1], (c & 1) ? ob.hi[2] : ob.lo[2] };
    float q[3]; xfm_point(I->l2w, p, q); box_extend_pt(&I->bounds, q);
  }
  I->valid = 1;
  for (int k = 0; k < 3; k++) if (!(valid_f(I->bounds.lo[k]) && valid_f(I->bounds.hi[k]) && I->bounds.lo[k] <= I->bounds.hi[k])) I->valid = 0;   /* Instance::buildBounds: isvalid(b) */
  if (I->valid) box_extend(&s->bounds, I->bounds.lo, I->bounds.hi);
}
API void ora_inst_bounds(const InstSet* s, float* lo3hi3) { memcpy(lo3hi3, &s->bounds, 24); }

static int inst_box_hit(const Inst* I, const Ray* r, float tfar) {    /* the cull only: slab test against the instance's world box */
  const float o[3] = { r->org_x, r->org_y, r->org_z }, d[3] = { r->dir_x, r->dir_y, r->dir_z };
  float t0 = fmaxf_(r->tnear, 0.0f), t1 = tfar;
  for (int k = 0; k < 3; k++) {
    const float rd = 1.0f / (fabsf(d[k]) < 1E-18f ? 1E-18f : d[k]);
    float a = (I->bounds.lo[k] - o[k]) * rd, b = (I->bounds.hi[k] - o[k]) * rd;
    if (a > b) { float t = a; a = b; b = t; }
    a *= a > 0 ? 1.0f - 4.0f * FLT_EPSILON : 1.0f + 4.0f * FLT_EPSILON; b *= b > 0 ? 1.0f + 4.0f * FLT_EPSILON : 1.0f - 4.0f * FLT_EPSILON;
    if (a > t0) t0 = a;
    if (b < t1) t1 = b;
  }
  return t0 <= t1;
}
static void inst_enter(const Inst* I, Ray* r, float* save) {
  save[0] = r->org_x; save[1] = r->org_y; save[2] = r->org_z; save[3] = r->dir_x; save[4] = r->dir_y; save[5] = r->dir_z;
  float o[3], d[3]; xfm_point(I->w2l, save, o); xfm_vector(I->w2l, save + 3, d);
  r->org_x = o[0]; r->org_y = o[1]; r->org_z = o[2]; r->dir_x = d[0]; r->dir_y = d[1]; r->dir_z = d[2];
}
static void inst_leave(Ray* r, const float* save) { r->org_x = save[0]; r->org_y = save[1]; r->org_z = save[2]; r->dir_x = save[3]; r->dir_y = save[4]; r->dir_z = save[5]; }
static void inst_ctx(const Inst* I, uint32_t id, uint32_t prim) { if (I->tri) { I->tri->ctxInstID = id; I->tri->ctxInstPrimID = prim; } if (I->quad) { I->quad->ctxInstID = id; I->quad->ctxInstPrimID = prim; } }

API void ora_inst_intersect1(InstSet* s, RayHit* rh, uint32_t M) {
  for (uint32_t i = 0; i < M; i++) for (uint32_t k = 0; k < s->n; k++) {
    const Inst* I = &s->inst[k];
    if (!I->valid || (rh[i].ray.mask & I->mask) == 0) continue;
    if (!inst_box_hit(I, &rh[i].ray, rh[i].ray.tfar)) continue;
    float save[6]; inst_enter(I, &rh[i].ray, save); inst_ctx(I, I->id, 0);
    if (I->tri) intersect1(I->tri, &rh[i]);
    if (I->quad) intersect1(I->quad, &rh[i]);
    inst_ctx(I, INVALID_ID, INVALID_ID); inst_leave(&rh[i].ray, save);
  }
}
API void ora_inst_occluded1(InstSet* s, Ray* r, uint32_t M) {
  for (uint32_t i = 0; i < M; i++) for (uint32_t k = 0; k < s->n && !(r[i].tfar < 0.0f); k++) {
    const Inst* I = &s->inst[k];
    if (!I->valid || (r[i].mask & I->mask) == 0) continue;
    if (!inst_box_hit(I, &r[i], r[i].tfar)) continue;
    float save[6]; inst_enter(I, &r[i], save);
    if (I->tri) occluded1(I->tri, &r[i]);
    if (I->quad && !(r[i].tfar < 0.0f)) occluded1(I->quad, &r[i]);
    inst_leave(&r[i], save);
  }
}

/* Tie classification helper: t of ray i against ONE named triangle,
   computed with the same Moeller-Trumbore arithmetic; NaN if that triangle is not hit in (tnear, tfar_in]. */
API void ora_triangle_t(Scene* s, const RayHit* rh, const uint32_t* geomID, const uint32_t* primID, float* t_out, uint32_t M) {
  for (uint32_t i = 0; i < M; i++) {
    t_out[i] = NAN;
    uint32_t g = geomID[i], p = primID[i];
    if (g >= s->nmesh || p >= s->mesh[g].nt) continue;
    const Mesh* m = &s->mesh[g];
    if (m->quad) {                                              /* a quad: the nearer of its two triangles */
      const uint32_t* q = m->t + 4 * (size_t)p;
      if (q[0] >= m->nv || q[1] >= m->nv || q[2] >= m->nv || q[3] >= m->nv) continue;
      Tri4 b; memset(&b, 0, sizeof(b));
      for (int d = 0; d < 3; d++) { b.v0[d][0] = m->v[3 * (size_t)q[0] + d]; b.e1[d][0] = m->v[3 * (size_t)q[1] + d]; b.e2[d][0] = m->v[3 * (size_t)q[2] + d]; b.v3[d][0] = m->v[3 * (size_t)q[3] + d]; }
      b.geomID[0] = g; b.primID[0] = p;
      for (int l = 0; l < 8; l += 4) { QLane ql; quad_lane(s, &b, l, &rh[i].ray, &ql); if (ql.valid && !(ql.t >= t_out[i])) t_out[i] = ql.t; }
      continue;
    }
    const uint32_t* tri = m->t + 3 * (size_t)p;
    if (tri[0] >= m->nv || tri[1] >= m->nv || tri[2] >= m->nv) continue;
    Tri4 b; memset(&b, 0, sizeof(b));
    const float *v0 = m->v + 3 * (size_t)tri[0], *v1 = m->v + 3 * (size_t)tri[1], *v2 = m->v + 3 * (size_t)tri[2];
    if (s->robust) {
      for (int d = 0; d < 3; d++) { b.v0[d][0] = v0[d]; b.e1[d][0] = v1[d]; b.e2[d][0] = v2[d]; }
      PLHit ph; pl_lane(&b, 0, &rh[i].ray, &ph);
      if (ph.valid) t_out[i] = ph.t;
      continue;
    }
    for (int d = 0; d < 3; d++) { b.v0[d][0] = v0[d]; b.e1[d][0] = v0[d] - v1[d]; b.e2[d][0] = v2[d] - v0[d]; }
    MTHit h; mt_lane(&b, 0, &rh[i].ray, &h);
    if (h.valid) t_out[i] = h.T * rcp_nr(h.absDen);
  }
}
