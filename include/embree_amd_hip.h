/* embree_amd_hip.h -- the thin C ABI between the host object model and the HIP kernels.
 *
 * Plain pointers and sizes only (no C++ / torch types).  Everything that touches the GPU
 * goes through these entry points; include/embree4/rtcore.h (the Embree-4 C API) is built
 * on top of them in embree_amd/csrc/rtcore_api.cpp.  Both sets live in libembree4_mi355.so.
 *
 * What each entry point replaces in the reference (paths relative to the reference root):
 *   mi355_bvh_build        BVHNBuilderSAH<8,Triangle4>::build      kernels/bvh/bvh_builder_sah.cpp:112-193
 *                          (createPrimRefArray builders/primrefgen.cpp:35-57, BuilderT::recurse
 *                          builders/bvh_builder_sah.h:214-308, CreateLeaf bvh_builder_sah.cpp:32-55)
 *   mi355_trace_closest    BVHNIntersector1<8,..>::intersect       kernels/bvh/bvh_intersector1.cpp:32-114
 *   mi355_trace_any        BVHNIntersector1<8,..>::occluded        kernels/bvh/bvh_intersector1.cpp:117-197
 *   mi355_trace_*_packet   BVHNIntersectorKHybrid::intersect/occluded  kernels/bvh/bvh_intersector_hybrid.cpp:106-369,600-783
 *   mi355_trace_stats      STAT3 counters                          kernels/common/stat.h:9-19
 * All functions return 0 on success or a hipError_t value (>0); mi355_last_error() gives text.
 */
#ifndef EMBREE_AMD_HIP_H
#define EMBREE_AMD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_API __attribute__((visibility("default")))

typedef struct mi355_bvh* mi355_bvh_t;

/* One triangle mesh, buffers RESIDENT ON THE DEVICE (the host object model uploads them in
   rtcCommitGeometry, or the application passes device memory directly).
   Layout rules are the reference's (TriangleMesh::setBuffer, kernels/common/scene_triangle_mesh.cpp:35-80):
   vertices float3 with byte stride >= 12 (4-byte aligned), indices uint3 with byte stride >= 12. */
typedef struct mi355_mesh {
  const void* d_vertices; size_t vertex_stride; uint32_t num_vertices;
  const void* d_indices;  size_t index_stride;  uint32_t num_triangles;
  uint32_t geom_id;       /* value reported in RTCHit.geomID */
  uint32_t mask;          /* geometry mask, tested against RTCRay.mask (default 1, kernels/common/geometry.cpp:48) */
  uint32_t quads;         /* 1: RTC_GEOMETRY_TYPE_QUAD -- indices are uint4 (stride >= 16), num_triangles counts QUADS.  A quad (v0,v1,v2,v3) is the
                             triangle pair (v0,v1,v3), (v2,v1,v3) of the reference's AVX quad intersectors (kernels/geometry/
                             quad_intersector_moeller.h:179-216); hits report the quad's index as primID and the quad's u,v */
  uint32_t reserved;
} mi355_mesh;

typedef struct mi355_build_params {
  uint32_t sah_block_shift;  /* SAH cost counts ceil(n / 2^shift) leaf blocks; reference: 2 (Triangle4).  default 0 */
  uint32_t min_leaf;         /* never split sets of <= min_leaf triangles; reference: 4.            default 2, max 3 */
  uint32_t max_leaf;         /* largest leaf slot; reference: 28 (7 Triangle4 blocks); encoding limit 3.  default 3 */
  uint32_t small_threshold;  /* sub-trees of <= this many triangles are finished by one wavefront in LDS.   default 512 (1024 until round 6) */
  float    trav_cost;        /* reference travCost = 1 */
  float    int_cost;         /* reference intCost  = 1 */
  uint32_t robust;           /* RTC_SCENE_FLAG_ROBUST (kernels/common/scene.cpp:180-188): leaves keep v0,v1,v2; traversal uses the
                                conservative node test (node_intersector1.h:539-554) and the Pluecker triangle test
                                (triangle_intersector_pluecker.h:68-118).  default 0 */
  uint32_t quality;          /* 0 = RTC_BUILD_QUALITY_MEDIUM (binned SAH, the default); 1 = RTC_BUILD_QUALITY_LOW: Morton-code build like the
                                reference's fast builder (kernels/builders/bvh_builder_morton.h), same node and leaf layout;
                                2 = RTC_BUILD_QUALITY_HIGH: SAH build with spatial splits inside the recursion (the reference's default high-quality
                                builder, kernels/builders/heuristic_spatial_array.h), or, with presplits = 1, large triangles pre-split into several
                                references with clipped boxes before a plain SAH build (kernels/builders/primrefgen_presplit.h); no refit data is kept */
  float    split_factor;     /* quality 2 only: references may grow to split_factor * triangles (reference: max_spatial_split_replications = 1.2).  default 1.2 */
  uint32_t refit;            /* 1: keep what mi355_bvh_refit needs (8 B per triangle: the leaf order, and the level table).  default 0 */
  uint32_t presplits;        /* quality 2 only: 1 = pre-split instead of splitting inside the recursion (reference: device config "presplits=1",
                                kernels/common/state.cpp:88,443).  default 0 */
  uint32_t top_splits;       /* quality 0 (MEDIUM) only: 1 = the few references that dwarf all others (walls, a ground plane) are cut into grid pieces before the
                                build, on the device (embree_amd/csrc/build_presplit.inl, outlier_*): the tree is then built by object splits over a few more
                                references.  The reference's MEDIUM builder never cuts a triangle; hits do not depend on it; leaf records of a cut triangle
                                exist more than once (mi355_bvh_info.num_presplit).  Off whenever params.refit is set.  default 1 */
  uint32_t top_split_min;    /* reserved */
  float    top_split_rel;    /* an outlier's box area is at least this many times the mean box area ...  default 32 */
  float    top_split_cell;   /* ... and it is longer than this fraction of the scene's largest extent, which is also the grid its pieces are cut on
                                (at most 32 cells per axis; nothing is cut if the cells of all outliers exceed N / 16 + 65536).  default 1/8 */
} mi355_build_params;

typedef struct mi355_bvh_info {
  uint64_t num_triangles;    /* valid triangles in the tree (invalid ones are skipped like the reference) */
  uint64_t num_nodes;        /* 8-wide quantised inner nodes (80 B each) */
  uint64_t num_leaves;
  uint64_t num_binary_nodes; /* intermediate binary SAH tree */
  uint64_t bytes_nodes, bytes_triangles;
  float    bounds_lower[3], bounds_upper[3];
  float    sah;              /* (sum inner area*travCost + sum leaf area*blocks*intCost) / root area */
  float    build_ms;         /* GPU time of the last build (hipEvent), excluding host->device uploads */
  uint32_t root_ref, top_levels, max_leaf, depth;
  uint64_t bytes_refit;      /* extra device memory kept for mi355_bvh_refit (0 unless built with params.refit) */
  uint32_t num_refits;       /* refits since the build; build_ms is the GPU time of the last build OR refit */
  uint32_t num_presplit;     /* quality 2: references added by pre-splitting (num_triangles counts references = leaf records) */
  uint32_t num_launches;     /* kernel launches of the last build */
  uint32_t num_host_syncs;   /* host round trips (stream synchronisations) of the last build: 2 on the default path (counters, final copy) */
  uint32_t build_attempts;   /* 1; 2 = the launch sequence learned from the last commit of this size was too short for this scene, the commit ran again with the blind margins;
                                + 1 if those were exceeded too and the stepwise path built the tree */
  uint32_t reserved0;
} mi355_bvh_info;

MI355_API void mi355_default_build_params(mi355_build_params* p);
MI355_API const char* mi355_last_error(void);
MI355_API int mi355_device_count(void);
MI355_API int mi355_device_name(int device, char* out, size_t n);

/* Build: blocking. `stream` is a hipStream_t or NULL. */
MI355_API int mi355_bvh_build(int device, const mi355_mesh* meshes, uint32_t num_meshes,
                              const mi355_build_params* params, void* stream, mi355_bvh_t* out);
MI355_API void mi355_bvh_destroy(mi355_bvh_t bvh);
/* Scenes with RTC_GEOMETRY_TYPE_INSTANCE, one level (the reference: kernels/common/scene_instance.h, kernels/geometry/instance_intersector.cpp).
   `own` = the flat tree of the scene's own triangles / quads (or NULL), `instances` = the instance geometries: the flat tree of the instanced scene,
   local2world as the reference's AffineSpace3fa (vx, vy, vz, p), the geometry id reported in RTCHit.instID[0], the geometry mask.  The result holds a
   top tree over the instances' world boxes and a COPY of every distinct tree (node / triangle indices rebased), so it stays valid when `own` or an
   object tree is destroyed or rebuilt -- and has to be built again to see their changes, like the reference needs a new commit.  Queries use the same
   entry points; hits inside an instance report object-space Ng, instID[0] = inst_id, instPrimID[0] = 0.  All trees must share params->robust. */
typedef struct mi355_instance {
  mi355_bvh_t object;
  float       local2world[12];
  uint32_t    inst_id, mask;
} mi355_instance;
MI355_API int mi355_bvh_build_instanced(int device, mi355_bvh_t own, const mi355_instance* instances, uint32_t num_instances,
                                        const mi355_build_params* params, void* stream, mi355_bvh_t* out);
/* Refit (RTC_BUILD_QUALITY_REFIT; the reference: kernels/bvh/bvh_refit.cpp, BVHNRefitT): the vertices moved, the topology did not.
   `meshes` must list the same geometries (ids, primitive counts, types) in the same order as at the build; vertex pointers, strides and
   masks are taken anew.  Triangle records are rewritten and the node boxes recomputed bottom-up, in place; blocking.
   Returns 0 on success; MI355_REFIT_IMPOSSIBLE when the tree was built without params.refit, the mesh list differs or the build had skipped invalid
   triangles (the tree is untouched); MI355_REFIT_BROKEN when a triangle has become invalid (the boxes are half rewritten: the tree is UNUSABLE and the
   caller must build again); other values: HIP errors. */
#define MI355_REFIT_IMPOSSIBLE (-2)
#define MI355_REFIT_BROKEN (-3)
MI355_API int mi355_bvh_refit(mi355_bvh_t bvh, const mi355_mesh* meshes, uint32_t num_meshes, void* stream);
/* A tree of mi355_bvh_build_instanced whose instances only MOVED (or changed mask / id): same instances in the same order naming the same objects.  The top tree
   keeps its topology and is refitted over the new world boxes, the instance records get their new world2local; the object trees are not touched and nothing
   is concatenated again (the reference refits / rebuilds only the top level of its two-level scenes, kernels/bvh/bvh_refit.cpp, bvh_builder_twolevel.cpp).
   Returns 0, or MI355_REFIT_IMPOSSIBLE when the list is not a move of what the tree was built from (the tree is untouched: build again). */
/* hipMalloc for everything this library and its host layer allocate next to the trees: on hipErrorOutOfMemory the node / triangle arrays of destroyed trees that
   the build arena keeps for the next commit (up to 1 GiB, see mi355_release_build_scratch) are returned to the driver and the allocation is tried once more. */
MI355_API int mi355_malloc_retry(int device, size_t bytes, void** d);
MI355_API int mi355_bvh_refit_instanced(mi355_bvh_t bvh, const mi355_instance* instances, uint32_t num_instances, void* stream);
/* Build scratch (prim refs, binary tree, work lists) is kept per device between commits, and so are the node / triangle arrays of up to four destroyed
   trees (the next commit of a similar size takes them over instead of paying hipFree + hipMalloc); this returns all of it to the driver. */
MI355_API void mi355_release_build_scratch(int device);
MI355_API int mi355_bvh_get_info(mi355_bvh_t bvh, mi355_bvh_info* info);
/* Device-side filter rules of a FLAT tree (the reference: filter callbacks run inside the traversal, kernels/geometry/filter.h:14-80, intersector_epilog.h:
   235-368; a host function cannot run in a HIP kernel, a rule can).  `words` = num_geoms entries of 12 words, indexed by geometry id --
     w0 kinds (1 modulo, 2 bit array, 4 distance window, 8 u/v cut-off) | apply << 8 (1 rtcIntersect*, 2 rtcOccluded*)
     w1 modulus m, w2 remainder r, w3 a | b << 16:  reject if (primID * a + geomID * b) % m == r
     w4 tmin, w5 tmax (floats): reject unless tmin <= t <= tmax          w6 umax, w7 vmax: reject if u > umax or v > vmax
     w8 offset (in words from the start of `words`) and w9 length (bits) of a bit array: reject if bit primID is set          w10, w11 reserved
   -- followed by the bit arrays.  num_words = 0 removes the rules.  An instanced tree (mi355_bvh_build_instanced) takes the rules its object trees carry at
   the time it is built.  Blocking; must not overlap queries on the tree. */
MI355_API int mi355_bvh_set_filter_rules(mi355_bvh_t bvh, const uint32_t* words, size_t num_words, uint32_t num_geoms);
/* Copies the tree to host memory for validation (tests): nodes = num_nodes*80 B, tris = num_triangles*48 B. */
MI355_API int mi355_bvh_download(mi355_bvh_t bvh, void* nodes, size_t nodes_bytes, void* tris, size_t tris_bytes);

/* Ray queries on DEVICE-resident AoS arrays (RTCRayHit = 96 B / RTCRay = 48 B records, byte_stride apart).
   Asynchronous on `stream`.  Per-ray contract = rtcIntersect1 / rtcOccluded1.  count <= 0xFFF00000 per launch (32-bit hand-out arithmetic; hipErrorInvalidValue beyond). */
/* Optional: create the traversal scratch of `stream` (ray cursors, stack spill area) ahead of the first launch on it (which would otherwise allocate it). */
MI355_API int mi355_trace_prepare(mi355_bvh_t bvh, void* stream);
MI355_API int mi355_trace_closest(mi355_bvh_t bvh, void* d_rayhit, uint32_t count, size_t byte_stride, void* stream);
MI355_API int mi355_trace_any(mi355_bvh_t bvh, void* d_ray, uint32_t count, size_t byte_stride, void* stream);
/* The same two launches with the query flags of RTCIntersectArguments / RTCOccludedArguments (include/embree4/rtcore.h RTCRayQueryFlags).
   MI355_QUERY_COHERENT (= RTC_RAY_QUERY_FLAG_COHERENT, the reference: BVHNIntersectorKHybrid::intersectCoherent, kernels/bvh/
   bvh_intersector_hybrid.cpp:374-533): consecutive rays are expected to take similar paths; the launch then uses the wave-packet kernel, in which
   the 64 rays of a wavefront walk the tree TOGETHER (one node fetch for all of them).  Results are those of the incoherent kernels. */
#define MI355_QUERY_COHERENT 0x10000u
/* with MI355_QUERY_COHERENT: every large coherent query samples its packets (every 32nd packet first) before it decides between the packet and the per-lane kernel, whatever
   earlier queries on this (tree, stream) found -- rtcNewDevice("coherent_memory=0").  Default: three samples in a row that found their packets apart are remembered and the
   next 15 coherent queries go to the per-lane kernel unsampled (what a renderer that passes the flag for incoherent batches wants; the answers are the same either way). */
#define MI355_QUERY_COHERENT_NO_MEMORY 0x40000000u
MI355_API int mi355_trace_query(mi355_bvh_t bvh, void* d_rays, uint32_t count, size_t byte_stride, int any_hit, uint32_t query_flags, void* stream);
/* The same query with a filter FUNCTION on the device (the reference's GPU path: runIntersectionFilter1SYCL / runOcclusionFilter1SYCL, kernels/geometry/filter_sycl.h:12-120,
   the function pointer of RTCIntersectArguments::filter / RTCOccludedArguments::filter called from inside the traversal).  filter_fn is the ADDRESS of a
       __device__ void f(const struct RTCFilterFunctionNArguments* args)          (N = 1; clear args->valid[0] to reject the candidate)
   in a gfx950 code object loaded in this process (the caller's own .so / hipModule: take the address on the device, e.g. a __device__ variable initialised with it and read
   back with hipMemcpyFromSymbol); filter_ctx becomes args->context.  The function is called for every candidate hit that passed the geometry's mask test and rules -- of the
   geometries that enabled it (mi355_bvh_set_filter_rules, w0 bit 16) or, with MI355_QUERY_INVOKE_ARGUMENT_FILTER (= RTC_RAY_QUERY_FLAG_INVOKE_ARGUMENT_FILTER), of every
   geometry.  Scenes with instances: hipErrorNotSupported.  The callee runs on the traversal kernel's wave: keep it small (it is compiled to the budget of 128 VGPRs the
   kernel runs with: declare it __attribute__((amdgpu_waves_per_eu(4,4))) if it is large); INTEGRATION.md, "device filter functions". */
#define MI355_QUERY_INVOKE_ARGUMENT_FILTER 0x2u
MI355_API int mi355_trace_query_filtered(mi355_bvh_t bvh, void* d_rays, uint32_t count, size_t byte_stride, int any_hit, uint32_t query_flags,
                                         uint64_t filter_fn, void* filter_ctx, void* stream);
/* Same launch with a HIP event recorded on `stream` immediately before and after the traversal kernel
   (after the 4-byte cursor reset), so that the interval is the kernel alone.  any_hit selects the kernel. */
MI355_API int mi355_trace_timed(mi355_bvh_t bvh, void* d_rays, uint32_t count, size_t byte_stride, int any_hit,
                                void* stream, void* ev_start, void* ev_stop);
/* The traversal kernels have two safety nets that drop work instead of hanging or corrupting memory: a per-wave iteration cap (a corrupt tree must not
   hang the GPU) and a bound on the per-lane stack spill area.  Neither fails silently: the kernel raises a word in host-visible memory.  This call
   synchronises `stream`, returns the OR of the flags raised by launches on it since the last call, and clears them.  The blocking host-array entry
   points (rtcIntersect1M, ...) check it themselves and record RTC_ERROR_UNKNOWN (the reference: RTC_CATCH_END -> RTC_ERROR_UNKNOWN for anything
   unexpected, kernels/common/rtcore.h:23-49). */
#define MI355_TRACE_ITER_CAP_HIT   1u
#define MI355_TRACE_STACK_OVERFLOW 2u
MI355_API int mi355_trace_status(mi355_bvh_t bvh, void* stream, uint32_t* out_flags);
/* SoA packets RTCRayHitK / RTCRayK (K = 4, 8, 16) on the device; d_valid = K ints per packet (-1 = active)
   or NULL for all-active; num_packets packets, packet_stride bytes apart. */
MI355_API int mi355_trace_closest_packet(mi355_bvh_t bvh, const int* d_valid, void* d_rayhitK, uint32_t K,
                                         uint32_t num_packets, size_t packet_stride, void* stream);
MI355_API int mi355_trace_any_packet(mi355_bvh_t bvh, const int* d_valid, void* d_rayK, uint32_t K,
                                     uint32_t num_packets, size_t packet_stride, void* stream);
/* Counting build of the same kernels (blocking): out[0]=inner nodes visited (80 B each), out[1]=triangle records
   fetched (48 B each), out[2]=rays, out[3]=stack entries spilled to global memory, out[4]=max stack depth,
   out[5]=wave loop iterations, out[6]=node-step blocks executed (per wave), out[7]=triangle-step blocks executed
   (per wave); out[0] / (64 * out[6]) is the SIMD utilisation of the node step; out[8..11] = lane-iterations spent
   without a ray / waiting for the retire batch / waiting for the triangle queue to drain / blocked on unqueued
   triangle bits; out[12] = node visits that found no child, out[14] = shader clocks (summed over the waves) spent in the ray hand-out block,
   out[15] = hand-out events, out[16] = shader clocks of the whole loop, out[17] = of the node step; out[18] / out[19] = DISTINCT nodes / triangle records fetched (a bit
   per record, set by the counting kernel: the compulsory bytes of the launch).  any_hit != 0 selects the occlusion kernel.
   The rays ARE traced (results written). */
MI355_API int mi355_trace_stats(mi355_bvh_t bvh, void* d_rays, uint32_t count, size_t byte_stride, int any_hit,
                                uint64_t out[32]);

/* ---- multi-GPU: sharded ray batches, results gathered over RCCL / xGMI (the reference has no multi-process code, so there is no
   reference function these replace: they implement BASELINE.json's north_star, "ray batches shard embarrassingly across the 8 GPUs of one node with the
   BVH replicated and hits gathered over RCCL/xGMI").  One process per GPU; every process commits the same scene (the build is deterministic: identical
   trees, nothing to broadcast) and traces its contiguous ray range (embree_amd/shard.py shard_range).  The 128-byte id made by rank 0 has to reach the
   other ranks through the host's own channel (bench.py: torch.distributed / gloo).  librccl.so.1 is loaded on first use of a mi355_comm_* call. */
typedef struct mi355_comm* mi355_comm_t;
#define MI355_COMM_ID_BYTES 128
MI355_API int  mi355_comm_unique_id(void* out_id128);                                             /* ncclGetUniqueId (rank 0) */
MI355_API int  mi355_comm_init(int device, const void* id128, int world, int rank, mi355_comm_t* out);   /* ncclCommInitRank: collective over all ranks, blocking */
MI355_API void mi355_comm_destroy(mi355_comm_t comm);
/* every rank contributes bytes_per_rank bytes at d_send; d_recv (world * bytes_per_rank bytes, rank order) is filled on every rank / on `root` only.
   Asynchronous on `stream` (a hipStream_t); all ranks must issue the same sequence of calls. */
MI355_API int  mi355_comm_allgather(mi355_comm_t comm, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream);
MI355_API int  mi355_comm_gather(mi355_comm_t comm, const void* d_send, void* d_recv, size_t bytes_per_rank, int root, void* stream);
/* The fields a query writes, squeezed out of the AoS records so that only results travel:
   closest hit: 32 B per ray = { tfar, u, v, primID | geomID, Ng_x, Ng_y, Ng_z } (two 16-byte halves); occlusion: 4 B per ray = tfar (-inf = occluded). */
MI355_API int  mi355_pack_hits(const void* d_rayhit, uint32_t count, size_t byte_stride, void* d_out, void* stream);
MI355_API int  mi355_pack_occluded(const void* d_ray, uint32_t count, size_t byte_stride, void* d_out, void* stream);
/* closest hits of scenes with instances: 48 B per ray = the 32 bytes above + { instID[0], instPrimID[0], 0, 0 } (RTCHit, include/embree4/rtcore.h) */
MI355_API int  mi355_pack_hits_inst(const void* d_rayhit, uint32_t count, size_t byte_stride, void* d_out, void* stream);
/* The way UP of a host-array query that sends only what the kernels read: `count` packed 48-byte RTCRay parts (org, tnear, dir, time, tfar, mask, id, flags) are put at
   d_records + i * byte_stride; closest != 0: the record's geomID becomes RTC_INVALID_GEOMETRY_ID (a miss stays recognisable on the way down).  With mi355_pack_hits / _inst /
   mi355_pack_occluded on the way down a host array crosses the link with 48 + 32 (48, 4) bytes per ray instead of 96 (48) each way (rtcIntersect1M / rtcOccluded1M of
   include/embree4/rtcore.h on host memory; the reference reads the caller's memory in place, kernels/common/rtcore.cpp). */
MI355_API int  mi355_unpack_rays(const void* d_packed, uint32_t count, void* d_records, size_t byte_stride, int closest, void* stream);
/* hipStreamWaitEvent: work enqueued on `stream` after this call waits for `event` (a handle of mi355_event_create) -- how a gather on a communication stream
   is ordered behind the traversal of its batch while the next batch is traced (bench.py) */
MI355_API int  mi355_stream_wait_event(void* stream, void* event);
MI355_API int  mi355_stream_query(void* stream);
/* The radix sort of the RTC_BUILD_QUALITY_LOW (Morton) build on its own -- the reference's counterpart is radix_sort_u32, kernels/builders/bvh_builder_morton.h:439
   (common/algorithms/parallel_sort.h).  n 63-bit keys in device memory (bit 63 clear) -> d_keys_sorted (n x uint64) and d_index_sorted (n x uint32: where each came from);
   equal keys keep their order (stable).  *ms (optional): time of the seven passes.  Blocking; the source array is not written. */
MI355_API int  mi355_sort_keys63(int device, const void* d_keys, void* d_keys_sorted, void* d_index_sorted, uint32_t n, float* ms);
/* What a streaming kernel reaches on this GPU (the achievable figure beside the 8 TB/s vendor peak): out[0] = device-to-device copy,
   bytes read + written per second; out[1] = read only; GB/s, best of `reps` passes over `bytes` (use >= 1 GiB: the Infinity Cache holds 256 MB). Blocking. */
MI355_API int  mi355_measure_bandwidth(int device, size_t bytes, int reps, double out[2]);             /* hipStreamQuery: 0 = idle, 1 = work pending, < 0 = error */
/* the host link of `device`: `bytes` up from pinned host memory and `bytes` down, alone and both at once (two streams = the two copy engines), best of `reps`:
 * out[0] = upload GB/s, out[1] = download GB/s, out[2] = ms for both directions at once -- the floor of a host-array query of that size (bench.py end_to_end.link_floor_ms).
 * No counterpart in the reference (its rays never leave host memory). */
MI355_API int  mi355_measure_host_link(int device, size_t bytes, int reps, double out[3]);

/* raw device memory helpers for hosts without a HIP binding (ctypes tests / bench) */
MI355_API int mi355_malloc(int device, size_t bytes, void** d_ptr);
MI355_API int mi355_free(void* d_ptr);
MI355_API int mi355_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
MI355_API int mi355_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
MI355_API int mi355_synchronize(void* stream);            /* hipStreamSynchronize (NULL = default stream) */
MI355_API int mi355_device_synchronize(int device);         /* hipDeviceSynchronize */
MI355_API int mi355_memcpy_d2d_async(void* d_dst, const void* d_src, size_t bytes, void* stream);
MI355_API int mi355_stream_create(int device, void** stream);
MI355_API int mi355_stream_destroy(void* stream);
/* HIP events for timing a region ON THE STREAM THE KERNELS RUN ON (bench.py roofline leg) */
MI355_API int mi355_event_create(void** event);
MI355_API int mi355_event_record(void* event, void* stream);
MI355_API int mi355_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on `stop` */
MI355_API int mi355_event_destroy(void* event);

#ifdef __cplusplus
}
#endif
#endif
