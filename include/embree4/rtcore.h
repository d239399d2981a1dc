/* embree4/rtcore.h -- C ABI of the MI355X-native ray-tracing core.
 *
 * This is the drop-in boundary: the subset of the Embree 4.4.1
 * C API that the triangle-mesh hot path needs, declared from scratch in ONE
 * header.  Every name, enum value and struct layout below is binary compatible
 * with the reference's default build (all geometry options on, instance-array
 * on => sizeof(RTCHit)==48, sizeof(RTCRayHit)==96, one instance level), so an
 * application compiled against the reference's headers links and runs against
 * libembree4_mi355.so unchanged.  Each declaration cites the reference
 * interface it replaces as  [ref: file:line]  (paths relative to the
 * reference's include/embree4/).
 *
 * Declared and built beyond plain triangle meshes: quad meshes, one level of
 * instancing, filter callbacks (host-array entry points) and device-side filter
 * rules, rtcInterpolate.  Entry points of the reference outside that path
 * (curves, subdivision, user geometry, point queries, collision, motion blur)
 * are not declared here; the shared library still exports every one of the
 * reference's 154 rtc* symbols, the undeclared ones as stubs that record
 * RTC_ERROR_INVALID_OPERATION -- what a reference build with those features
 * compiled out does (kernels/common/rtcore.cpp:1553-1555).
 *
 * Extension (required for a GPU: one host call per ray cannot feed 256 CUs):
 * the batched calls rtcIntersect1M / rtcOccluded1M (modelled on the Embree-3
 * stream API removed in 4.0, reference README.md:1434-1441) and their
 * device-resident forms rtcIntersect1MDevice / rtcOccluded1MDevice, declared
 * at the end of this file.
 */
#ifndef EMBREE4_MI355_RTCORE_H
#define EMBREE4_MI355_RTCORE_H

#include <stddef.h>
#include <stdbool.h>
#include <sys/types.h>

#if defined(__cplusplus)
#  define RTC_API extern "C" __attribute__((visibility("default")))
#  define RTC_OPTIONAL_ARGUMENT = nullptr
#else
#  define RTC_API __attribute__((visibility("default")))
#  define RTC_OPTIONAL_ARGUMENT
#endif
#define RTC_NAMESPACE_BEGIN
#define RTC_NAMESPACE_END
#define RTC_NAMESPACE_USE
#define RTC_ALIGN(n) __attribute__((aligned(n)))
#define RTC_FORCEINLINE inline __attribute__((always_inline))

/* [ref: kernels/rtcore_config.h.in:10-21] */
#define RTC_VERSION_MAJOR 4
#define RTC_VERSION_MINOR 4
#define RTC_VERSION_PATCH 1
#define RTC_VERSION 40401
#define RTC_VERSION_STRING "4.4.1"
#define RTC_MAX_INSTANCE_LEVEL_COUNT 1
#define RTC_GEOMETRY_INSTANCE_ARRAY
#define RTC_MIN_WIDTH 0

/* [ref: rtcore_common.h:52-55] */
#define RTC_INVALID_GEOMETRY_ID ((unsigned int)-1)
#define RTC_MAX_TIME_STEP_COUNT 129

/* ------------------------------------------------------------------ handles */
/* opaque, intrusively reference counted [ref: rtcore_device.h:11-12,
   rtcore_buffer.h:43, rtcore_scene.h:10-11] */
typedef struct RTCDeviceTy*      RTCDevice;
typedef struct RTCSceneTy*       RTCScene;
typedef struct RTCGeometryTy*    RTCGeometry;
typedef struct RTCBufferTy*      RTCBuffer;
typedef struct RTCTraversableTy* RTCTraversable;

/* -------------------------------------------------------------------- enums */
/* [ref: rtcore_common.h:58-150] only the families a triangle mesh can use are
   named; values are the reference's. */
enum RTCFormat {
  RTC_FORMAT_UNDEFINED = 0,
  RTC_FORMAT_UCHAR = 0x1001, RTC_FORMAT_UCHAR2, RTC_FORMAT_UCHAR3, RTC_FORMAT_UCHAR4,
  RTC_FORMAT_CHAR  = 0x2001, RTC_FORMAT_CHAR2,  RTC_FORMAT_CHAR3,  RTC_FORMAT_CHAR4,
  RTC_FORMAT_USHORT= 0x3001, RTC_FORMAT_USHORT2,RTC_FORMAT_USHORT3,RTC_FORMAT_USHORT4,
  RTC_FORMAT_SHORT = 0x4001, RTC_FORMAT_SHORT2, RTC_FORMAT_SHORT3, RTC_FORMAT_SHORT4,
  RTC_FORMAT_UINT  = 0x5001, RTC_FORMAT_UINT2,  RTC_FORMAT_UINT3,  RTC_FORMAT_UINT4,
  RTC_FORMAT_INT   = 0x6001, RTC_FORMAT_INT2,   RTC_FORMAT_INT3,   RTC_FORMAT_INT4,
  RTC_FORMAT_ULLONG= 0x7001, RTC_FORMAT_ULLONG2,RTC_FORMAT_ULLONG3,RTC_FORMAT_ULLONG4,
  RTC_FORMAT_LLONG = 0x8001, RTC_FORMAT_LLONG2, RTC_FORMAT_LLONG3, RTC_FORMAT_LLONG4,
  RTC_FORMAT_FLOAT = 0x9001, RTC_FORMAT_FLOAT2, RTC_FORMAT_FLOAT3, RTC_FORMAT_FLOAT4,
  RTC_FORMAT_FLOAT5, RTC_FORMAT_FLOAT6, RTC_FORMAT_FLOAT7, RTC_FORMAT_FLOAT8,
  RTC_FORMAT_FLOAT9, RTC_FORMAT_FLOAT10, RTC_FORMAT_FLOAT11, RTC_FORMAT_FLOAT12,
  RTC_FORMAT_FLOAT13, RTC_FORMAT_FLOAT14, RTC_FORMAT_FLOAT15, RTC_FORMAT_FLOAT16,
  /* matrix formats accepted by rtcSetGeometryTransform [ref: rtcore_common.h:128-147] */
  RTC_FORMAT_FLOAT3X4_ROW_MAJOR = 0x9134, RTC_FORMAT_FLOAT3X4_COLUMN_MAJOR = 0x9234, RTC_FORMAT_FLOAT4X4_COLUMN_MAJOR = 0x9244,
  RTC_FORMAT_GRID = 0xA001,
  RTC_FORMAT_QUATERNION_DECOMPOSITION = 0xB001
};

/* [ref: rtcore_common.h:153-159] */
enum RTCBuildQuality {
  RTC_BUILD_QUALITY_LOW = 0, RTC_BUILD_QUALITY_MEDIUM = 1,
  RTC_BUILD_QUALITY_HIGH = 2, RTC_BUILD_QUALITY_REFIT = 3
};

/* [ref: rtcore_common.h:179-298] the triangle path only distinguishes these */
enum RTCFeatureFlags {
  RTC_FEATURE_FLAG_NONE = 0,
  RTC_FEATURE_FLAG_MOTION_BLUR = 1 << 0,
  RTC_FEATURE_FLAG_TRIANGLE = 1 << 1,
  RTC_FEATURE_FLAG_QUAD = 1 << 2,
  RTC_FEATURE_FLAG_GRID = 1 << 3,
  RTC_FEATURE_FLAG_SUBDIVISION = 1 << 4,
  RTC_FEATURE_FLAG_INSTANCE = 1 << 23,
  RTC_FEATURE_FLAG_FILTER_FUNCTION_IN_ARGUMENTS = 1 << 24,
  RTC_FEATURE_FLAG_FILTER_FUNCTION_IN_GEOMETRY = 1 << 25,
  RTC_FEATURE_FLAG_FILTER_FUNCTION = (1 << 24) | (1 << 25),
  RTC_FEATURE_FLAG_USER_GEOMETRY_CALLBACK_IN_ARGUMENTS = 1 << 26,
  RTC_FEATURE_FLAG_USER_GEOMETRY_CALLBACK_IN_GEOMETRY = 1 << 27,
  RTC_FEATURE_FLAG_USER_GEOMETRY = (1 << 26) | (1 << 27),
  RTC_FEATURE_FLAG_32_BIT_RAY_MASK = 1 << 28,
  RTC_FEATURE_FLAG_INSTANCE_ARRAY = 1 << 29,
  RTC_FEATURE_FLAG_ALL = 0xffffffff
};

/* [ref: rtcore_common.h:301-310] */
enum RTCRayQueryFlags {
  RTC_RAY_QUERY_FLAG_NONE = 0,
  RTC_RAY_QUERY_FLAG_INVOKE_ARGUMENT_FILTER = 1 << 1,
  RTC_RAY_QUERY_FLAG_INCOHERENT = 0 << 16,
  RTC_RAY_QUERY_FLAG_COHERENT = 1 << 16
};

/* [ref: rtcore_device.h:49-81] */
enum RTCDeviceProperty {
  RTC_DEVICE_PROPERTY_VERSION = 0,
  RTC_DEVICE_PROPERTY_VERSION_MAJOR = 1,
  RTC_DEVICE_PROPERTY_VERSION_MINOR = 2,
  RTC_DEVICE_PROPERTY_VERSION_PATCH = 3,
  RTC_DEVICE_PROPERTY_NATIVE_RAY4_SUPPORTED = 32,
  RTC_DEVICE_PROPERTY_NATIVE_RAY8_SUPPORTED = 33,
  RTC_DEVICE_PROPERTY_NATIVE_RAY16_SUPPORTED = 34,
  RTC_DEVICE_PROPERTY_BACKFACE_CULLING_SPHERES_ENABLED = 62,
  RTC_DEVICE_PROPERTY_BACKFACE_CULLING_CURVES_ENABLED = 63,
  RTC_DEVICE_PROPERTY_RAY_MASK_SUPPORTED = 64,
  RTC_DEVICE_PROPERTY_BACKFACE_CULLING_ENABLED = 65,
  RTC_DEVICE_PROPERTY_FILTER_FUNCTION_SUPPORTED = 66,
  RTC_DEVICE_PROPERTY_IGNORE_INVALID_RAYS_ENABLED = 67,
  RTC_DEVICE_PROPERTY_COMPACT_POLYS_ENABLED = 68,
  RTC_DEVICE_PROPERTY_TRIANGLE_GEOMETRY_SUPPORTED = 96,
  RTC_DEVICE_PROPERTY_QUAD_GEOMETRY_SUPPORTED = 97,
  RTC_DEVICE_PROPERTY_SUBDIVISION_GEOMETRY_SUPPORTED = 98,
  RTC_DEVICE_PROPERTY_CURVE_GEOMETRY_SUPPORTED = 99,
  RTC_DEVICE_PROPERTY_USER_GEOMETRY_SUPPORTED = 100,
  RTC_DEVICE_PROPERTY_POINT_GEOMETRY_SUPPORTED = 101,
  RTC_DEVICE_PROPERTY_TASKING_SYSTEM = 128,
  RTC_DEVICE_PROPERTY_JOIN_COMMIT_SUPPORTED = 129,
  RTC_DEVICE_PROPERTY_PARALLEL_COMMIT_SUPPORTED = 130,
  RTC_DEVICE_PROPERTY_CPU_DEVICE = 140,
  RTC_DEVICE_PROPERTY_SYCL_DEVICE = 141,
  /* extension: 1 on this library (HIP device behind the API) */
  RTC_DEVICE_PROPERTY_HIP_DEVICE = 142,
  RTC_DEVICE_PROPERTY_GPU_COUNT = 143,             /* extension: GPUs behind this device (rtcNewDevice("gpus=N")) */
  RTC_DEVICE_PROPERTY_GPU_OF_REPLICA_0 = 1000      /* extension: + k = the HIP device ordinal replica k of this device lives on (k < GPU_COUNT): where shard k of
                                                      rtcIntersect1MDeviceSharded / rtcOccluded1MDeviceSharded must be allocated */
};

/* [ref: rtcore_device.h:90-100] */
enum RTCError {
  RTC_ERROR_NONE = 0, RTC_ERROR_UNKNOWN = 1, RTC_ERROR_INVALID_ARGUMENT = 2,
  RTC_ERROR_INVALID_OPERATION = 3, RTC_ERROR_OUT_OF_MEMORY = 4,
  RTC_ERROR_UNSUPPORTED_CPU = 5, RTC_ERROR_CANCELLED = 6,
  RTC_ERROR_LEVEL_ZERO_RAYTRACING_SUPPORT_MISSING = 7
};

/* [ref: rtcore_buffer.h:12-40] */
enum RTCBufferType {
  RTC_BUFFER_TYPE_INDEX = 0, RTC_BUFFER_TYPE_VERTEX = 1,
  RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE = 2, RTC_BUFFER_TYPE_NORMAL = 3,
  RTC_BUFFER_TYPE_TANGENT = 4, RTC_BUFFER_TYPE_NORMAL_DERIVATIVE = 5,
  RTC_BUFFER_TYPE_GRID = 8, RTC_BUFFER_TYPE_FACE = 16, RTC_BUFFER_TYPE_LEVEL = 17,
  RTC_BUFFER_TYPE_EDGE_CREASE_INDEX = 18, RTC_BUFFER_TYPE_EDGE_CREASE_WEIGHT = 19,
  RTC_BUFFER_TYPE_VERTEX_CREASE_INDEX = 20, RTC_BUFFER_TYPE_VERTEX_CREASE_WEIGHT = 21,
  RTC_BUFFER_TYPE_HOLE = 22, RTC_BUFFER_TYPE_TRANSFORM = 23, RTC_BUFFER_TYPE_FLAGS = 32
};

/* [ref: rtcore_geometry.h:18-53] TRIANGLE, QUAD and INSTANCE are the types this library
   builds; every other value makes rtcNewGeometry record RTC_ERROR_INVALID_OPERATION. */
enum RTCGeometryType {
  RTC_GEOMETRY_TYPE_TRIANGLE = 0, RTC_GEOMETRY_TYPE_QUAD = 1, RTC_GEOMETRY_TYPE_GRID = 2,
  RTC_GEOMETRY_TYPE_SUBDIVISION = 8,
  RTC_GEOMETRY_TYPE_CONE_LINEAR_CURVE = 15, RTC_GEOMETRY_TYPE_ROUND_LINEAR_CURVE = 16,
  RTC_GEOMETRY_TYPE_FLAT_LINEAR_CURVE = 17,
  RTC_GEOMETRY_TYPE_ROUND_BEZIER_CURVE = 24, RTC_GEOMETRY_TYPE_FLAT_BEZIER_CURVE = 25,
  RTC_GEOMETRY_TYPE_NORMAL_ORIENTED_BEZIER_CURVE = 26,
  RTC_GEOMETRY_TYPE_ROUND_BSPLINE_CURVE = 32, RTC_GEOMETRY_TYPE_FLAT_BSPLINE_CURVE = 33,
  RTC_GEOMETRY_TYPE_NORMAL_ORIENTED_BSPLINE_CURVE = 34,
  RTC_GEOMETRY_TYPE_ROUND_HERMITE_CURVE = 40, RTC_GEOMETRY_TYPE_FLAT_HERMITE_CURVE = 41,
  RTC_GEOMETRY_TYPE_NORMAL_ORIENTED_HERMITE_CURVE = 42,
  RTC_GEOMETRY_TYPE_SPHERE_POINT = 50, RTC_GEOMETRY_TYPE_DISC_POINT = 51,
  RTC_GEOMETRY_TYPE_ORIENTED_DISC_POINT = 52,
  RTC_GEOMETRY_TYPE_ROUND_CATMULL_ROM_CURVE = 58, RTC_GEOMETRY_TYPE_FLAT_CATMULL_ROM_CURVE = 59,
  RTC_GEOMETRY_TYPE_NORMAL_ORIENTED_CATMULL_ROM_CURVE = 60,
  RTC_GEOMETRY_TYPE_USER = 120, RTC_GEOMETRY_TYPE_INSTANCE = 121,
  RTC_GEOMETRY_TYPE_INSTANCE_ARRAY = 122
};

/* [ref: rtcore_scene.h:21-29] */
enum RTCSceneFlags {
  RTC_SCENE_FLAG_NONE = 0, RTC_SCENE_FLAG_DYNAMIC = 1 << 0, RTC_SCENE_FLAG_COMPACT = 1 << 1,
  RTC_SCENE_FLAG_ROBUST = 1 << 2, RTC_SCENE_FLAG_FILTER_FUNCTION_IN_ARGUMENTS = 1 << 3,
  RTC_SCENE_FLAG_PREFETCH_USM_SHARED_ON_GPU = 1 << 4
};

/* ------------------------------------------------------------ ray / hit data */
/* [ref: rtcore_ray.h:11-27] 48 bytes */
struct RTC_ALIGN(16) RTCRay {
  float org_x, org_y, org_z, tnear;
  float dir_x, dir_y, dir_z, time;
  float tfar; unsigned int mask, id, flags;
};
/* [ref: rtcore_ray.h:30-45] 48 bytes in the default ABI (instPrimID present) */
struct RTC_ALIGN(16) RTCHit {
  float Ng_x, Ng_y, Ng_z;   /* unnormalised (v1-v0)x(v2-v0) */
  float u, v;
  unsigned int primID, geomID;
  unsigned int instID[RTC_MAX_INSTANCE_LEVEL_COUNT];
  unsigned int instPrimID[RTC_MAX_INSTANCE_LEVEL_COUNT];
};
/* [ref: rtcore_ray.h:48-52] 96 bytes */
struct RTCRayHit { struct RTCRay ray; struct RTCHit hit; };

/* SoA packets [ref: rtcore_ray.h:55-184]; K = 4, 8, 16, alignment 16/32/64 */
#define RTC__DECL_PACKET(K, A)                                                        \
  struct RTC_ALIGN(A) RTCRay##K {                                                     \
    float org_x[K], org_y[K], org_z[K], tnear[K];                                     \
    float dir_x[K], dir_y[K], dir_z[K], time[K];                                      \
    float tfar[K]; unsigned int mask[K], id[K], flags[K]; };                          \
  struct RTC_ALIGN(A) RTCHit##K {                                                     \
    float Ng_x[K], Ng_y[K], Ng_z[K], u[K], v[K];                                      \
    unsigned int primID[K], geomID[K];                                                \
    unsigned int instID[RTC_MAX_INSTANCE_LEVEL_COUNT][K];                             \
    unsigned int instPrimID[RTC_MAX_INSTANCE_LEVEL_COUNT][K]; };                      \
  struct RTCRayHit##K { struct RTCRay##K ray; struct RTCHit##K hit; };
RTC__DECL_PACKET(4, 16)
RTC__DECL_PACKET(8, 32)
RTC__DECL_PACKET(16, 64)

/* [ref: rtcore_common.h:162-173] */
struct RTC_ALIGN(16) RTCBounds {
  float lower_x, lower_y, lower_z, align0;
  float upper_x, upper_y, upper_z, align1;
};

/* ----------------------------------------------------------- query arguments */
/* [ref: rtcore_common.h:335-361] */
struct RTCRayQueryContext {
  unsigned int instID[RTC_MAX_INSTANCE_LEVEL_COUNT];
  unsigned int instPrimID[RTC_MAX_INSTANCE_LEVEL_COUNT];
};
RTC_FORCEINLINE void rtcInitRayQueryContext(struct RTCRayQueryContext* c) {
  c->instID[0] = RTC_INVALID_GEOMETRY_ID; c->instPrimID[0] = RTC_INVALID_GEOMETRY_ID;
}

/* Filter callbacks [ref: rtcore_common.h:308-324, rtcore_ray.h:221-261, kernels/geometry/filter.h:14-80] are HOST functions.  The host-array entry points
   (rtcIntersect1/4/8/16/1M, rtcOccluded...) run them between launches: the traversal finds the closest candidate, the callbacks (N = 1) accept or reject it,
   a rejected ray is traced on from behind the candidate -- the closest ACCEPTED hit is what the reference returns too.  The device-pointer entry points
   record RTC_ERROR_INVALID_OPERATION for a non-NULL filter; user-geometry callbacks (intersect / occluded) always do. */
struct RTCRayN; struct RTCHitN;
struct RTCFilterFunctionNArguments {
  int* valid;
  void* geometryUserPtr;
  struct RTCRayQueryContext* context;
  struct RTCRayN* ray;
  struct RTCHitN* hit;
  unsigned int N;
};
struct RTCIntersectFunctionNArguments;
struct RTCOccludedFunctionNArguments;
typedef void (*RTCFilterFunctionN)(const struct RTCFilterFunctionNArguments*);
typedef void (*RTCIntersectFunctionN)(const struct RTCIntersectFunctionNArguments*);
typedef void (*RTCOccludedFunctionN)(const struct RTCOccludedFunctionNArguments*);
/* structure-of-arrays accessors of a ray / hit packet of size N, lane i [ref: rtcore_ray.h:221-261]: in C++ the reference's names returning references
   (RTCRayN_tfar(ray, N, i) = ...), in C pointer forms (*RTCRayN_tfar_ptr(ray, N, i) = ...) */
#if defined(__cplusplus)
#define RTC_SOA_FIELD(S, name, type, k) RTC_FORCEINLINE type& S##_##name(S* p, unsigned int N, unsigned int i) { return ((type*)p)[(k) * N + i]; }
#else
#define RTC_SOA_FIELD(S, name, type, k) RTC_FORCEINLINE type* S##_##name##_ptr(struct S* p, unsigned int N, unsigned int i) { return (type*)p + (k) * N + i; }
#endif
RTC_SOA_FIELD(RTCRayN, org_x, float, 0) RTC_SOA_FIELD(RTCRayN, org_y, float, 1) RTC_SOA_FIELD(RTCRayN, org_z, float, 2) RTC_SOA_FIELD(RTCRayN, tnear, float, 3)
RTC_SOA_FIELD(RTCRayN, dir_x, float, 4) RTC_SOA_FIELD(RTCRayN, dir_y, float, 5) RTC_SOA_FIELD(RTCRayN, dir_z, float, 6) RTC_SOA_FIELD(RTCRayN, time, float, 7)
RTC_SOA_FIELD(RTCRayN, tfar, float, 8) RTC_SOA_FIELD(RTCRayN, mask, unsigned int, 9) RTC_SOA_FIELD(RTCRayN, id, unsigned int, 10) RTC_SOA_FIELD(RTCRayN, flags, unsigned int, 11)
RTC_SOA_FIELD(RTCHitN, Ng_x, float, 0) RTC_SOA_FIELD(RTCHitN, Ng_y, float, 1) RTC_SOA_FIELD(RTCHitN, Ng_z, float, 2) RTC_SOA_FIELD(RTCHitN, u, float, 3) RTC_SOA_FIELD(RTCHitN, v, float, 4)
RTC_SOA_FIELD(RTCHitN, primID, unsigned int, 5) RTC_SOA_FIELD(RTCHitN, geomID, unsigned int, 6)
#if defined(__cplusplus)
RTC_FORCEINLINE unsigned int& RTCHitN_instID(RTCHitN* p, unsigned int N, unsigned int i, unsigned int level) { return ((unsigned int*)p)[(7 + level) * N + i]; }
RTC_FORCEINLINE unsigned int& RTCHitN_instPrimID(RTCHitN* p, unsigned int N, unsigned int i, unsigned int level) { return ((unsigned int*)p)[(7 + RTC_MAX_INSTANCE_LEVEL_COUNT + level) * N + i]; }   /* [ref: rtcore_ray.h:220] */
#else
RTC_FORCEINLINE unsigned int* RTCHitN_instID_ptr(struct RTCHitN* p, unsigned int N, unsigned int i, unsigned int level) { return (unsigned int*)p + (7 + level) * N + i; }
RTC_FORCEINLINE unsigned int* RTCHitN_instPrimID_ptr(struct RTCHitN* p, unsigned int N, unsigned int i, unsigned int level) { return (unsigned int*)p + (7 + RTC_MAX_INSTANCE_LEVEL_COUNT + level) * N + i; }
#endif
#undef RTC_SOA_FIELD

/* [ref: rtcore_scene.h:34-58] */
struct RTCIntersectArguments {
  enum RTCRayQueryFlags flags;
  enum RTCFeatureFlags feature_mask;
  struct RTCRayQueryContext* context;
  RTCFilterFunctionN filter;
  RTCIntersectFunctionN intersect;
};
RTC_FORCEINLINE void rtcInitIntersectArguments(struct RTCIntersectArguments* a) {
  a->flags = RTC_RAY_QUERY_FLAG_INCOHERENT; a->feature_mask = RTC_FEATURE_FLAG_ALL;
  a->context = NULL; a->filter = NULL; a->intersect = NULL;
}
/* [ref: rtcore_scene.h:61-86] */
struct RTCOccludedArguments {
  enum RTCRayQueryFlags flags;
  enum RTCFeatureFlags feature_mask;
  struct RTCRayQueryContext* context;
  RTCFilterFunctionN filter;
  RTCOccludedFunctionN occluded;
};
RTC_FORCEINLINE void rtcInitOccludedArguments(struct RTCOccludedArguments* a) {
  a->flags = RTC_RAY_QUERY_FLAG_INCOHERENT; a->feature_mask = RTC_FEATURE_FLAG_ALL;
  a->context = NULL; a->filter = NULL; a->occluded = NULL;
}

/* ------------------------------------------------------------------- device */
/* [ref: rtcore_device.h:16,38,41,84,87,103-121]
   config string: "k=v,k=v" as the reference (kernels/common/state.cpp:224).
   Honoured keys: verbose, benchmark, gpu=<ordinal>, max_leaf=<n>, leaf_block=<n>;
   CPU-only keys (threads, isa, tri_accel, ...) are accepted and ignored. */
RTC_API RTCDevice rtcNewDevice(const char* config);
RTC_API void rtcRetainDevice(RTCDevice device);
RTC_API void rtcReleaseDevice(RTCDevice device);
RTC_API ssize_t rtcGetDeviceProperty(RTCDevice device, enum RTCDeviceProperty prop);
RTC_API void rtcSetDeviceProperty(RTCDevice device, const enum RTCDeviceProperty prop, ssize_t value);
RTC_API const char* rtcGetErrorString(enum RTCError error);
RTC_API enum RTCError rtcGetDeviceError(RTCDevice device);
RTC_API const char* rtcGetDeviceLastErrorMessage(RTCDevice device);
typedef void (*RTCErrorFunction)(void* userPtr, enum RTCError code, const char* str);
RTC_API void rtcSetDeviceErrorFunction(RTCDevice device, RTCErrorFunction error, void* userPtr);
typedef bool (*RTCMemoryMonitorFunction)(void* ptr, ssize_t bytes, bool post);
RTC_API void rtcSetDeviceMemoryMonitorFunction(RTCDevice device, RTCMemoryMonitorFunction memoryMonitor, void* userPtr);

/* ------------------------------------------------------------------- buffer */
/* [ref: rtcore_buffer.h:46,52,69,77,80] */
RTC_API RTCBuffer rtcNewBuffer(RTCDevice device, size_t byteSize);
RTC_API RTCBuffer rtcNewSharedBuffer(RTCDevice device, void* ptr, size_t byteSize);
RTC_API void* rtcGetBufferData(RTCBuffer buffer);
RTC_API void rtcRetainBuffer(RTCBuffer buffer);
RTC_API void rtcReleaseBuffer(RTCBuffer buffer);

/* ----------------------------------------------------------------- geometry */
/* [ref: rtcore_geometry.h:130-207] */
RTC_API RTCGeometry rtcNewGeometry(RTCDevice device, enum RTCGeometryType type);
RTC_API void rtcRetainGeometry(RTCGeometry geometry);
RTC_API void rtcReleaseGeometry(RTCGeometry geometry);
RTC_API void rtcCommitGeometry(RTCGeometry geometry);
RTC_API void rtcEnableGeometry(RTCGeometry geometry);
RTC_API void rtcDisableGeometry(RTCGeometry geometry);
RTC_API void rtcSetGeometryTimeStepCount(RTCGeometry geometry, unsigned int timeStepCount);
RTC_API void rtcSetGeometryVertexAttributeCount(RTCGeometry geometry, unsigned int vertexAttributeCount);
RTC_API void rtcSetGeometryMask(RTCGeometry geometry, unsigned int mask);
/* Vertex data interpolation at (u, v) of a triangle / quad, on the host [ref: rtcore_geometry.h:284-387] */
struct RTCInterpolateArguments {
  RTCGeometry geometry; unsigned int primID; float u; float v; enum RTCBufferType bufferType; unsigned int bufferSlot;
  float* P; float* dPdu; float* dPdv; float* ddPdudu; float* ddPdvdv; float* ddPdudv; unsigned int valueCount;
};
RTC_API void rtcInterpolate(const struct RTCInterpolateArguments* args);
struct RTCInterpolateNArguments {
  RTCGeometry geometry; const void* valid; const unsigned int* primIDs; const float* u; const float* v; unsigned int N;
  enum RTCBufferType bufferType; unsigned int bufferSlot;
  float* P; float* dPdu; float* dPdv; float* ddPdudu; float* ddPdvdv; float* ddPdudv; unsigned int valueCount;
};
RTC_API void rtcInterpolateN(const struct RTCInterpolateNArguments* args);
/* RTC_GEOMETRY_TYPE_INSTANCE, one level, one time step [ref: rtcore_geometry.h: rtcSetGeometryInstancedScene, rtcSetGeometryTransform, rtcGetGeometryTransform] */
RTC_API void rtcSetGeometryInstancedScene(RTCGeometry geometry, RTCScene scene);
RTC_API void rtcSetGeometryTransform(RTCGeometry geometry, unsigned int timeStep, enum RTCFormat format, const void* xfm);
RTC_API void rtcGetGeometryTransform(RTCGeometry geometry, float time, enum RTCFormat format, void* xfm);
RTC_API void rtcSetGeometryBuildQuality(RTCGeometry geometry, enum RTCBuildQuality quality);
RTC_API void rtcSetGeometryBuffer(RTCGeometry geometry, enum RTCBufferType type, unsigned int slot,
                                  enum RTCFormat format, RTCBuffer buffer, size_t byteOffset,
                                  size_t byteStride, size_t itemCount);
RTC_API void rtcSetSharedGeometryBuffer(RTCGeometry geometry, enum RTCBufferType type, unsigned int slot,
                                        enum RTCFormat format, const void* ptr, size_t byteOffset,
                                        size_t byteStride, size_t itemCount);
/* device-resident geometry: dptr is HIP device memory on the device's GPU; no upload happens at commit.
   [ref: rtcore_geometry.h:175, the reference's entry point for its own GPU (SYCL) device] */
RTC_API void rtcSetSharedGeometryBufferHostDevice(RTCGeometry geometry, enum RTCBufferType type, unsigned int slot,
                                                  enum RTCFormat format, const void* ptr, const void* dptr,
                                                  size_t byteOffset, size_t byteStride, size_t itemCount);
RTC_API void* rtcSetNewGeometryBuffer(RTCGeometry geometry, enum RTCBufferType type, unsigned int slot,
                                      enum RTCFormat format, size_t byteStride, size_t itemCount);
RTC_API void* rtcGetGeometryBufferData(RTCGeometry geometry, enum RTCBufferType type, unsigned int slot);
RTC_API void rtcUpdateGeometryBuffer(RTCGeometry geometry, enum RTCBufferType type, unsigned int slot);
RTC_API void rtcSetGeometryUserData(RTCGeometry geometry, void* ptr);
RTC_API void* rtcGetGeometryUserData(RTCGeometry geometry);
/* host callbacks: run by the host-array entry points between launches (the closest candidate of every ray is
   offered, rejected rays are traced on); the device-pointer entry points record RTC_ERROR_INVALID_OPERATION when
   a callback would have to run (a host function cannot run in a HIP kernel) -- see rtcSetGeometryFilterRule_mi355
   below for rules that run inside the kernel.  [ref: rtcore_geometry.h:150-165] */
RTC_API void rtcSetGeometryIntersectFilterFunction(RTCGeometry geometry, RTCFilterFunctionN filter);
RTC_API void rtcSetGeometryOccludedFilterFunction(RTCGeometry geometry, RTCFilterFunctionN filter);
RTC_API void rtcSetGeometryEnableFilterFunctionFromArguments(RTCGeometry geometry, bool enable);
/* Extension: a filter RULE that runs inside the traversal kernels, where the reference calls the filter callbacks [ref: kernels/geometry/filter.h:14-80,
   intersector_epilog.h:235-368].  A candidate hit of this geometry is rejected (the ray goes on) if ANY enabled part of the rule says so:
     RTC_FILTER_RULE_MODULO           (primID * primFactor + geomID * geomFactor) % modulus == remainder
     RTC_FILTER_RULE_PRIMITIVE_BITS   bit primID of `bits` (numBits bits, copied at rtcCommitScene) is set -- a per-primitive alpha mask
     RTC_FILTER_RULE_DISTANCE_WINDOW  t outside [tmin, tmax]
     RTC_FILTER_RULE_UV_CUTOFF        u > umax or v > vmax
   `apply` says which queries run it.  Works for every entry point (the device-pointer ones included) and inside instanced scenes; takes effect with the next
   rtcCommitScene of the scene(s) the geometry is attached to.  rule = NULL removes it.  Host callbacks, where set, run after it on the host-array entry points. */
enum RTCFilterRuleKind { RTC_FILTER_RULE_MODULO = 1, RTC_FILTER_RULE_PRIMITIVE_BITS = 2, RTC_FILTER_RULE_DISTANCE_WINDOW = 4, RTC_FILTER_RULE_UV_CUTOFF = 8 };
enum RTCFilterRuleApply { RTC_FILTER_RULE_APPLY_INTERSECT = 1, RTC_FILTER_RULE_APPLY_OCCLUDED = 2 };
struct RTCFilterRule {
  unsigned int kinds;                      /* OR of RTCFilterRuleKind */
  unsigned int apply;                      /* OR of RTCFilterRuleApply */
  unsigned int modulus, remainder, primFactor, geomFactor;   /* factors < 65536 */
  float tmin, tmax, umax, vmax;
  const unsigned int* bits; unsigned int numBits;
};
RTC_API void rtcSetGeometryFilterRule(RTCGeometry geometry, const struct RTCFilterRule* rule);

/* -------------------------------------------------------------------- scene */
/* [ref: rtcore_scene.h:89-150] */
RTC_API RTCScene rtcNewScene(RTCDevice device);
RTC_API RTCDevice rtcGetSceneDevice(RTCScene scene);
RTC_API void rtcRetainScene(RTCScene scene);
RTC_API void rtcReleaseScene(RTCScene scene);
RTC_API RTCTraversable rtcGetSceneTraversable(RTCScene scene);
RTC_API unsigned int rtcAttachGeometry(RTCScene scene, RTCGeometry geometry);
RTC_API void rtcAttachGeometryByID(RTCScene scene, RTCGeometry geometry, unsigned int geomID);
RTC_API void rtcDetachGeometry(RTCScene scene, unsigned int geomID);
RTC_API RTCGeometry rtcGetGeometry(RTCScene scene, unsigned int geomID);
RTC_API RTCGeometry rtcGetGeometryThreadSafe(RTCScene scene, unsigned int geomID);
RTC_API void rtcCommitScene(RTCScene scene);       /* builds the BVH on the GPU; blocking */
RTC_API void rtcJoinCommitScene(RTCScene scene);
typedef bool (*RTCProgressMonitorFunction)(void* ptr, double n);
RTC_API void rtcSetSceneProgressMonitorFunction(RTCScene scene, RTCProgressMonitorFunction progress, void* ptr);
RTC_API void rtcSetSceneBuildQuality(RTCScene scene, enum RTCBuildQuality quality);
RTC_API void rtcSetSceneFlags(RTCScene scene, enum RTCSceneFlags flags);
RTC_API enum RTCSceneFlags rtcGetSceneFlags(RTCScene scene);
RTC_API void rtcGetSceneBounds(RTCScene scene, struct RTCBounds* bounds_o);

/* ------------------------------------------------------------- ray queries */
/* [ref: rtcore_scene.h:170-179, 208-217, 260-300]
   Host pointers.  Each call is a (tiny) GPU launch: correct, re-entrant, slow.
   The measured path is the batched extension below. */
RTC_API void rtcIntersect1(RTCScene scene, struct RTCRayHit* rayhit, struct RTCIntersectArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcIntersect4(const int* valid, RTCScene scene, struct RTCRayHit4* rayhit, struct RTCIntersectArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcIntersect8(const int* valid, RTCScene scene, struct RTCRayHit8* rayhit, struct RTCIntersectArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcIntersect16(const int* valid, RTCScene scene, struct RTCRayHit16* rayhit, struct RTCIntersectArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcOccluded1(RTCScene scene, struct RTCRay* ray, struct RTCOccludedArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcOccluded4(const int* valid, RTCScene scene, struct RTCRay4* ray, struct RTCOccludedArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcOccluded8(const int* valid, RTCScene scene, struct RTCRay8* ray, struct RTCOccludedArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcOccluded16(const int* valid, RTCScene scene, struct RTCRay16* ray, struct RTCOccludedArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcTraversableIntersect1(RTCTraversable t, struct RTCRayHit* rayhit, struct RTCIntersectArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcTraversableIntersect4(const int* valid, RTCTraversable t, struct RTCRayHit4* rayhit, struct RTCIntersectArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcTraversableIntersect8(const int* valid, RTCTraversable t, struct RTCRayHit8* rayhit, struct RTCIntersectArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcTraversableIntersect16(const int* valid, RTCTraversable t, struct RTCRayHit16* rayhit, struct RTCIntersectArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcTraversableOccluded1(RTCTraversable t, struct RTCRay* ray, struct RTCOccludedArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcTraversableOccluded4(const int* valid, RTCTraversable t, struct RTCRay4* ray, struct RTCOccludedArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcTraversableOccluded8(const int* valid, RTCTraversable t, struct RTCRay8* ray, struct RTCOccludedArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcTraversableOccluded16(const int* valid, RTCTraversable t, struct RTCRay16* ray, struct RTCOccludedArguments* args RTC_OPTIONAL_ARGUMENT);

/* ------------------------------------------------ batched extension (new) */
/* M rays, array-of-structs, `byteStride` bytes between consecutive elements
   (>= sizeof element, multiple of 16).  Same per-ray contract as
   rtcIntersect1 / rtcOccluded1 (reference doc/src/api/rtcIntersect1.md,
   rtcOccluded1.md): a miss leaves the element untouched, a closest hit writes
   ray.tfar + the RTCHit members, an occluded ray gets tfar = -inf.
   Blocking; thread-safe on a committed scene. */
RTC_API void rtcIntersect1M(RTCScene scene, struct RTCRayHit* rayhit, unsigned int M, size_t byteStride,
                            struct RTCIntersectArguments* args RTC_OPTIONAL_ARGUMENT);
RTC_API void rtcOccluded1M(RTCScene scene, struct RTCRay* ray, unsigned int M, size_t byteStride,
                           struct RTCOccludedArguments* args RTC_OPTIONAL_ARGUMENT);
/* Device-resident forms: `rayhit`/`ray` are HIP device pointers on the scene's
   GPU, `stream` is a hipStream_t (NULL = the default stream).  Asynchronous:
   returns after enqueueing; results are ordered on `stream`.  One call takes at
   most 0xFFF00000 (4,293,918,720) rays: the kernels' hand-out arithmetic is
   32-bit; a larger M records RTC_ERROR_INVALID_ARGUMENT (split the batch).
   Device filter FUNCTIONS (extension, off by default: rtcNewDevice("device_filter_functions=1")): args->filter of these two calls may then be the ADDRESS of a
       __device__ void f(const struct RTCFilterFunctionNArguments* args)      (gfx950, the caller's own code object; N = 1, args->valid[0] = -1 on entry)
   which the traversal kernel calls where the reference's GPU path calls its function pointer (kernels/geometry/filter_sycl.h:12-120): for every candidate hit -- after the
   ray-mask test and the geometry's filter rule -- of the geometries that called rtcSetGeometryEnableFilterFunctionFromArguments, or of all geometries with
   RTC_RAY_QUERY_FLAG_INVOKE_ARGUMENT_FILTER; args->ray->tfar is the candidate's distance, args->hit the full hit, args->geometryUserPtr / args->context what the
   application set (device-accessible memory, if the function dereferences them).  Clearing args->valid[0] rejects the candidate and the traversal goes on.  Requires
   args->feature_mask to contain RTC_FEATURE_FLAG_FILTER_FUNCTION_IN_ARGUMENTS (the default mask does), a device over ONE GPU and a scene without instances.  Without the
   config key a non-NULL filter in these calls records RTC_ERROR_INVALID_OPERATION (a host function cannot run on the GPU).  How to take the address: INTEGRATION.md. */
RTC_API void rtcIntersect1MDevice(RTCScene scene, void* rayhit, unsigned int M, size_t byteStride,
                                  struct RTCIntersectArguments* args, void* stream);
RTC_API void rtcOccluded1MDevice(RTCScene scene, void* ray, unsigned int M, size_t byteStride,
                                 struct RTCOccludedArguments* args, void* stream);
/* One RTCDevice over N GPUs (rtcNewDevice("gpus=N")), rays resident where they are traced: shard k (counts[k] records at the HIP device pointer rayhits[k],
   memory of the GPU RTC_DEVICE_PROPERTY_GPU_OF_REPLICA_0 + k names) is traced by replica k on streams[k] (a hipStream_t of that GPU; streams == NULL or
   streams[k] == NULL = that GPU's default stream).  Nothing crosses xGMI: the committed tree exists once per GPU, every shard stays on its GPU, and what
   the consumer needs afterwards travels in whatever form it chooses (include/embree_amd_hip.h: mi355_pack_hits + mi355_comm_gather).  numShards <= GPU_COUNT;
   a shard with counts[k] == 0 is skipped.  Asynchronous, ordered on the shards' streams.  (The single-pointer forms above keep the array on the first GPU and
   move 1 - 1/N of it over xGMI and back per call.) */
RTC_API void rtcIntersect1MDeviceSharded(RTCScene scene, unsigned int numShards, void* const* rayhits, const unsigned int* counts, size_t byteStride,
                                         struct RTCIntersectArguments* args, void* const* streams);
RTC_API void rtcOccluded1MDeviceSharded(RTCScene scene, unsigned int numShards, void* const* rays, const unsigned int* counts, size_t byteStride,
                                        struct RTCOccludedArguments* args, void* const* streams);

#endif /* EMBREE4_MI355_RTCORE_H */
