// shard.hip -- multi-GPU result exchange for sharded ray batches (BASELINE.json north_star: "ray batches shard
// embarrassingly across the 8 GPUs of one node with the BVH replicated and hits gathered over RCCL/xGMI").
//
// The reference has no multi-process code at all: a host application that wants N GPUs runs one process per GPU, every process commits the
// same scene (the GPU build is deterministic, so the trees are bit-identical and nothing has to be broadcast), traces its contiguous ray
// range [g*M/G, (g+1)*M/G) and hands the results to whoever consumes them.  This file is that last step when the consumer is a GPU:
//   * pack kernels squeeze the fields a query WRITES out of the AoS ray records (RTCRayHit: tfar, Ng, u, v, primID, geomID = 32 B of the
//     96-byte record; RTCRay after rtcOccluded: tfar = 4 B of 48), so that only results travel;
//   * a thin wrapper over RCCL (librccl.so.1, loaded on first use so that single-GPU applications never map its 570 MB) gathers the packed
//     shards on every GPU (ncclAllGather) or on one (ncclGather).  xGMI is point to point (7 links per GPU): a gather to one root uses the
//     root's 7 links in parallel; an all-gather is a ring and costs (G-1)/G of the total per link, so it is meant for the 4-byte occlusion
//     results (64 MB for 16 Mi shadow rays), not for full hit records.
// The rendezvous (who is rank r, the 128-byte ncclUniqueId) is the host's business: bench.py / embree_amd/shard.py pass the id around with
// torch.distributed (gloo).  Plain C ABI, declared in include/embree_amd_hip.h.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>
#include <mutex>
#include <string>
#include "../../include/embree_amd_hip.h"
#include "internal.h"

namespace {

// ---- result records -----------------------------------------------------------------------------------------------------------------
// RTCRayHit (include/embree4/rtcore.h): tfar at byte 32, Ng_x..u at 48..63, v, primID, geomID at 64..75.  One lane per ray, two 16-byte stores.
__global__ void pack_hits_kernel(const char* rays, uint32_t count, uint32_t stride, uint4* out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const char* r = rays + (size_t)i * stride;
    const uint32_t tfar = *(const uint32_t*)(r + 32);
    const uint4 a = *(const uint4*)(r + 48);            // Ng_x, Ng_y, Ng_z, u
    const uint4 b = *(const uint4*)(r + 64);            // v, primID, geomID, instID[0]
    out[2 * (size_t)i] = make_uint4(tfar, a.w, b.x, b.y);       // tfar, u, v, primID
    out[2 * (size_t)i + 1] = make_uint4(b.z, a.x, a.y, a.z);    // geomID, Ng
  }
}
// scenes with instances: + instID[0], instPrimID[0] (bytes 76..83 of the record) = 48 B per ray, three 16-byte stores
__global__ void pack_hits_inst_kernel(const char* rays, uint32_t count, uint32_t stride, uint4* out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const char* r = rays + (size_t)i * stride;
    const uint32_t tfar = *(const uint32_t*)(r + 32);
    const uint4 a = *(const uint4*)(r + 48);            // Ng_x, Ng_y, Ng_z, u
    const uint4 b = *(const uint4*)(r + 64);            // v, primID, geomID, instID[0]
    const uint32_t instPrim = *(const uint32_t*)(r + 80);
    out[3 * (size_t)i] = make_uint4(tfar, a.w, b.x, b.y);
    out[3 * (size_t)i + 1] = make_uint4(b.z, a.x, a.y, a.z);
    out[3 * (size_t)i + 2] = make_uint4(b.w, instPrim, 0u, 0u);
  }
}
// RTCRay after an occlusion query: only tfar changed (-inf = occluded)
__global__ void pack_occluded_kernel(const char* rays, uint32_t count, uint32_t stride, uint32_t* out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
    out[i] = *(const uint32_t*)(rays + (size_t)i * stride + 32);
}

// The way UP of a host-array query that sends only what the kernels read (rtcore_api.cpp, staged_query): 48 bytes per ray -- org, tnear, dir, time, tfar, mask, id, flags --
// arrive packed and are put where the traversal expects them, at `stride`; a closest-hit record also gets geomID = RTC_INVALID_GEOMETRY_ID, which is how the way down tells a
// miss (nothing to write back: the reference leaves the caller's hit fields alone) from a hit.
__global__ void unpack_rays_kernel(const uint4* packed, uint32_t count, char* recs, uint32_t stride, uint32_t closest) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const uint4 a = packed[3 * (size_t)i], b = packed[3 * (size_t)i + 1], c = packed[3 * (size_t)i + 2];
    char* r = recs + (size_t)i * stride;
    *(uint4*)r = a; *(uint4*)(r + 16) = b; *(uint4*)(r + 32) = c;
    if (closest) *(uint32_t*)(r + 72) = 0xFFFFFFFFu;
  }
}

// ---- achievable HBM bandwidth of THIS box ("also measure an on-device copy/read kernel and report the fraction against both").  Grid-stride
// 16-byte accesses, 8 independent loads in flight per lane, far more workgroups than CUs; the buffers are larger than the 256 MB Infinity Cache.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void bw_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
    for (int k = 0; k < 8; k++) __builtin_nontemporal_store(v[k], dst + i + k * stride);
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void bw_read_kernel(const u32x4* __restrict__ src, size_t n16, uint32_t* sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
    for (int k = 0; k < 8; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  }
  for (; i < n16; i += stride) acc ^= src[i].x;
  if (acc == 0x9E3779B9u) *sink = acc;                        // (keeps the loads alive; practically never taken)
}

// ---- RCCL, bound at run time ------------------------------------------------------------------------------------------------------------
typedef struct { char internal[128]; } rcclUniqueId;        // ncclUniqueId, rccl.h: NCCL_UNIQUE_ID_BYTES = 128
typedef void* rcclComm;
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(rcclUniqueId*) = nullptr;
  int (*CommInitRank)(rcclComm*, int, rcclUniqueId, int) = nullptr;
  int (*CommDestroy)(rcclComm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, rcclComm, hipStream_t) = nullptr;
  int (*Gather)(const void*, void*, size_t, int, int, rcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
};
Rccl* rccl() {
  static std::mutex m; static Rccl* r = nullptr;
  std::lock_guard<std::mutex> lk(m);
  if (r) return r;
  r = new Rccl;
  // The ROCm installation's own RCCL first, by PATH: a process that has imported PyTorch already holds PyTorch's private librccl.so.1, which is bound to
  // PyTorch's private copy of the HIP runtime -- a second runtime that does not know this library's device allocations.  A dlopen by soname would return
  // that copy; a dlopen by path loads the ROCm one, whose libamdhip64.so.7 is the runtime this library is already bound to.
  std::string rocm = std::string(getenv("ROCM_PATH") ? getenv("ROCM_PATH") : "/opt/rocm") + "/lib/librccl.so.1";
  const char* names[] = {getenv("MI355_RCCL_LIB"), rocm.c_str(), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
  for (const char* n : names) { if (!n) continue; r->lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (r->lib) break; r->err = dlerror(); }
  if (!r->lib) return r;
  auto sym = [&](const char* s) { void* p = dlsym(r->lib, s); if (!p) r->err = std::string("missing RCCL symbol ") + s; return p; };
  r->GetUniqueId = (int (*)(rcclUniqueId*))sym("ncclGetUniqueId");
  r->CommInitRank = (int (*)(rcclComm*, int, rcclUniqueId, int))sym("ncclCommInitRank");
  r->CommDestroy = (int (*)(rcclComm))sym("ncclCommDestroy");
  r->AllGather = (int (*)(const void*, void*, size_t, int, rcclComm, hipStream_t))sym("ncclAllGather");
  r->Gather = (int (*)(const void*, void*, size_t, int, int, rcclComm, hipStream_t))sym("ncclGather");
  r->GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
  if (!r->GetUniqueId || !r->CommInitRank || !r->CommDestroy || !r->AllGather || !r->Gather) { dlclose(r->lib); r->lib = nullptr; }
  return r;
}
int rccl_fail(const char* what, int rc) {
  Rccl* r = rccl();
  std::string m = std::string(what) + ": " + (r->lib && r->GetErrorString ? r->GetErrorString(rc) : r->err.c_str());
  return mi355::set_error(hipErrorUnknown, m.c_str());
}
constexpr int RCCL_INT8 = 0;   // ncclInt8 = ncclChar = 0 (rccl.h): everything travels as bytes

}  // namespace

struct mi355_comm { rcclComm comm; int device, world, rank; };

extern "C" {

int mi355_pack_hits(const void* d_rayhit, uint32_t count, size_t stride, void* d_out, void* stream) {
  if (count == 0) return 0;
  if (stride < 96 || (stride & 15u) || ((uintptr_t)d_rayhit & 15u) || ((uintptr_t)d_out & 15u)) return mi355::set_error(hipErrorInvalidValue, "mi355_pack_hits: 16-byte aligned RTCRayHit records expected");
  const uint32_t blocks = (count + 255u) / 256u < 4096u ? (count + 255u) / 256u : 4096u;
  hipLaunchKernelGGL(pack_hits_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const char*)d_rayhit, count, (uint32_t)stride, (uint4*)d_out);
  HIP_TRY(hipGetLastError());
  return 0;
}
int mi355_pack_hits_inst(const void* d_rayhit, uint32_t count, size_t stride, void* d_out, void* stream) {
  if (count == 0) return 0;
  if (stride < 96 || (stride & 15u) || ((uintptr_t)d_rayhit & 15u) || ((uintptr_t)d_out & 15u)) return mi355::set_error(hipErrorInvalidValue, "mi355_pack_hits_inst: 16-byte aligned RTCRayHit records expected");
  const uint32_t blocks = (count + 255u) / 256u < 4096u ? (count + 255u) / 256u : 4096u;
  hipLaunchKernelGGL(pack_hits_inst_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const char*)d_rayhit, count, (uint32_t)stride, (uint4*)d_out);
  HIP_TRY(hipGetLastError());
  return 0;
}
int mi355_unpack_rays(const void* d_packed, uint32_t count, void* d_records, size_t stride, int closest, void* stream) {
  if (count == 0) return 0;
  if (stride < (closest ? 96u : 48u) || (stride & 15u) || ((uintptr_t)d_records & 15u) || ((uintptr_t)d_packed & 15u)) return mi355::set_error(hipErrorInvalidValue, "mi355_unpack_rays: 16-byte aligned records expected");
  const uint32_t blocks = (count + 255u) / 256u < 4096u ? (count + 255u) / 256u : 4096u;
  hipLaunchKernelGGL(unpack_rays_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)d_packed, count, (char*)d_records, (uint32_t)stride, closest ? 1u : 0u);
  HIP_TRY(hipGetLastError());
  return 0;
}
int mi355_stream_wait_event(void* stream, void* event) { HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0)); return 0; }
int mi355_pack_occluded(const void* d_ray, uint32_t count, size_t stride, void* d_out, void* stream) {
  if (count == 0) return 0;
  if (stride < 48 || (stride & 3u)) return mi355::set_error(hipErrorInvalidValue, "mi355_pack_occluded: RTCRay records expected");
  const uint32_t blocks = (count + 255u) / 256u < 4096u ? (count + 255u) / 256u : 4096u;
  hipLaunchKernelGGL(pack_occluded_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const char*)d_ray, count, (uint32_t)stride, (uint32_t*)d_out);
  HIP_TRY(hipGetLastError());
  return 0;
}

int mi355_comm_unique_id(void* out128) {
  Rccl* r = rccl();
  if (!r->lib) return rccl_fail("RCCL is not available", 0);
  rcclUniqueId id; memset(&id, 0, sizeof(id));
  const int rc = r->GetUniqueId(&id);
  if (rc) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(out128, &id, sizeof(id));
  return 0;
}
int mi355_comm_init(int device, const void* id128, int world, int rank, mi355_comm_t* out) {
  *out = nullptr;
  Rccl* r = rccl();
  if (!r->lib) return rccl_fail("RCCL is not available", 0);
  if (world < 1 || rank < 0 || rank >= world) return mi355::set_error(hipErrorInvalidValue, "mi355_comm_init: rank outside the world");
  HIP_TRY(hipSetDevice(device));
  rcclUniqueId id; memcpy(&id, id128, sizeof(id));
  rcclComm c = nullptr;
  const int rc = r->CommInitRank(&c, world, id, rank);
  if (rc) return rccl_fail("ncclCommInitRank", rc);
  *out = new mi355_comm{c, device, world, rank};
  return 0;
}
void mi355_comm_destroy(mi355_comm_t c) {
  if (!c) return;
  Rccl* r = rccl();
  if (r->lib && c->comm) { hipSetDevice(c->device); r->CommDestroy(c->comm); }
  delete c;
}
int mi355_comm_allgather(mi355_comm_t c, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream) {
  if (!c) return mi355::set_error(hipErrorInvalidValue, "mi355_comm_allgather: no communicator");
  HIP_TRY(hipSetDevice(c->device));
  const int rc = rccl()->AllGather(d_send, d_recv, bytes_per_rank, RCCL_INT8, c->comm, (hipStream_t)stream);
  return rc ? rccl_fail("ncclAllGather", rc) : 0;
}
int mi355_comm_gather(mi355_comm_t c, const void* d_send, void* d_recv, size_t bytes_per_rank, int root, void* stream) {
  if (!c) return mi355::set_error(hipErrorInvalidValue, "mi355_comm_gather: no communicator");
  HIP_TRY(hipSetDevice(c->device));
  const int rc = rccl()->Gather(d_send, d_recv, bytes_per_rank, RCCL_INT8, root, c->comm, (hipStream_t)stream);
  return rc ? rccl_fail("ncclGather", rc) : 0;
}
// Measures what a streaming kernel reaches on this GPU: out[0] = copy (bytes read + bytes written per second), out[1] = read only, GB/s, best of `reps`.
int mi355_measure_bandwidth(int device, size_t bytes, int reps, double out[2]) {
  HIP_TRY(hipSetDevice(device));
  if (bytes < (1u << 20)) bytes = 1u << 20;
  bytes &= ~(size_t)4095;
  char *a = nullptr, *b = nullptr; uint32_t* sink = nullptr;
  HIP_TRY(hipMalloc((void**)&a, bytes)); 
  if (hipMalloc((void**)&b, bytes) != hipSuccess) { hipFree(a); return mi355::set_error(hipErrorOutOfMemory, "mi355_measure_bandwidth"); }
  if (hipMalloc((void**)&sink, 64) != hipSuccess) { hipFree(a); hipFree(b); return mi355::set_error(hipErrorOutOfMemory, "mi355_measure_bandwidth"); }
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t n16 = bytes / 16;
  const uint32_t blocks = 256u * 16u;
  double bestCopy = 0.0, bestRead = 0.0;
  for (int r = 0; r < reps + 1; r++) {
    float ms = 0;
    hipEventRecord(e0, nullptr); hipLaunchKernelGGL(bw_copy_kernel, dim3(blocks), dim3(256), 0, nullptr, (const u32x4*)a, (u32x4*)b, n16); hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    if (r && ms > 0) { const double g = 2.0 * (double)bytes / (ms * 1e-3) / 1e9; if (g > bestCopy) bestCopy = g; }
    hipEventRecord(e0, nullptr); hipLaunchKernelGGL(bw_read_kernel, dim3(blocks), dim3(256), 0, nullptr, (const u32x4*)a, n16, sink); hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    if (r && ms > 0) { const double g = (double)bytes / (ms * 1e-3) / 1e9; if (g > bestRead) bestRead = g; }
  }
  const hipError_t le = hipGetLastError();
  hipEventDestroy(e0); hipEventDestroy(e1); hipFree(a); hipFree(b); hipFree(sink);
  if (le != hipSuccess) return mi355::set_error(le, "mi355_measure_bandwidth");
  out[0] = bestCopy; out[1] = bestRead;
  return 0;
}
// What the host link allows for a host-array query (bench.py's end_to_end leg is read against it): `bytes` from pinned host memory to the device and `bytes` back,
// each direction alone and both at once on two streams (the two copy engines), best of `reps`.  out[0] = upload GB/s, out[1] = download GB/s, out[2] = milliseconds for
// both directions at once = the floor of rtcIntersect1M on an array of that size with a perfect pipeline (profiles/r04_host_link.md).
int mi355_measure_host_link(int device, size_t bytes, int reps, double out[3]) {
  HIP_TRY(hipSetDevice(device));
  if (bytes < (1u << 20)) bytes = 1u << 20;
  char *hUp = nullptr, *hDown = nullptr, *dUp = nullptr, *dDown = nullptr;
  hipStream_t s0 = nullptr, s1 = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  hipError_t err = hipHostMalloc((void**)&hUp, bytes, hipHostMallocDefault);
  if (err == hipSuccess) err = hipHostMalloc((void**)&hDown, bytes, hipHostMallocDefault);
  if (err == hipSuccess) err = hipMalloc((void**)&dUp, bytes);
  if (err == hipSuccess) err = hipMalloc((void**)&dDown, bytes);
  if (err == hipSuccess) err = hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  if (err == hipSuccess) err = hipEventCreate(&e2);
  double up = 0.0, down = 0.0, both = 1e30;
  if (err == hipSuccess) {
    memset(hUp, 1, bytes); (void)hipMemset(dDown, 2, bytes);
    for (int r = 0; r < reps + 1 && err == hipSuccess; r++) {
      float ms = 0;
      hipEventRecord(e0, s0); hipMemcpyAsync(dUp, hUp, bytes, hipMemcpyHostToDevice, s0); hipEventRecord(e1, s0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      if (r && ms > 0) { const double g = (double)bytes / (ms * 1e-3) / 1e9; if (g > up) up = g; }
      hipEventRecord(e0, s1); hipMemcpyAsync(hDown, dDown, bytes, hipMemcpyDeviceToHost, s1); hipEventRecord(e1, s1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      if (r && ms > 0) { const double g = (double)bytes / (ms * 1e-3) / 1e9; if (g > down) down = g; }
      hipDeviceSynchronize();
      hipEventRecord(e0, s0); hipStreamWaitEvent(s1, e0, 0);                                  // both directions start together
      hipMemcpyAsync(dUp, hUp, bytes, hipMemcpyHostToDevice, s0); hipMemcpyAsync(hDown, dDown, bytes, hipMemcpyDeviceToHost, s1);
      hipEventRecord(e2, s1); hipStreamWaitEvent(s0, e2, 0); hipEventRecord(e1, s0);            // ... and e1 is behind both
      err = hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      if (r && ms > 0 && ms < both) both = ms;
    }
  }
  if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); if (e2) hipEventDestroy(e2);
  if (s0) hipStreamDestroy(s0); if (s1) hipStreamDestroy(s1);
  if (hUp) hipHostFree(hUp); if (hDown) hipHostFree(hDown); if (dUp) hipFree(dUp); if (dDown) hipFree(dDown);
  if (err != hipSuccess) { (void)hipGetLastError(); return mi355::set_error(err, "mi355_measure_host_link"); }
  out[0] = up; out[1] = down; out[2] = both < 1e29 ? both : 0.0;
  return 0;
}
int mi355_stream_query(void* stream) {                           // 0 = everything enqueued on the stream has finished, 1 = still running, < 0 = error
  const hipError_t e = hipStreamQuery((hipStream_t)stream);
  if (e == hipSuccess) return 0;
  if (e == hipErrorNotReady) { (void)hipGetLastError(); return 1; }
  return -mi355::set_error(e, "hipStreamQuery");
}

}
