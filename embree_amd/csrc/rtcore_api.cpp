// rtcore_api.cpp -- the Embree-4 C API (include/embree4/rtcore.h) on top of the HIP core.
//
// Mirrors, for triangle meshes only, the reference's API shim and object model:
//   kernels/common/rtcore.cpp   entry points, argument defaults, error capture (RTC_CATCH_BEGIN/END, rtcore.h:23-68)
//   kernels/common/device.cpp   per-device, per-thread, first-error-wins error state (:265-330), config string
//   kernels/common/scene.cpp    attach/detach with lowest-free-ID pool (:717-741), commit state machine (:919-1043)
//   kernels/common/scene_triangle_mesh.cpp  buffer rules (:35-80)
//   kernels/common/geometry.cpp default mask 1 (:48), enable/disable, commit
// Errors never cross the ABI as exceptions; unsupported features record RTC_ERROR_INVALID_OPERATION exactly
// like a reference build with the feature compiled out (rtcore.cpp:1553-1555).
// There is NO CPU fallback: every ray query and every commit runs on the GPU or reports an error.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <map>
#include <mutex>
#include <set>
#include <exception>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include <emmintrin.h>
#include "../../include/embree4/rtcore.h"
#include "../../include/embree_amd_hip.h"

namespace {

struct rtc_error {
  RTCError code; std::string msg;
  rtc_error(RTCError c, const char* m) : code(c), msg(m) {}
};
#define THROW(code, msg) throw rtc_error(code, msg)

struct ErrState { RTCError error = RTC_ERROR_NONE; std::string msg; };
thread_local ErrState g_threadError;                       // errors without a device (failed rtcNewDevice)

struct RefCounted {
  std::atomic<int> refs{1};
  virtual ~RefCounted() {}
  void retain() { refs.fetch_add(1); }
  void release() { if (refs.fetch_sub(1) == 1) delete this; }
};

struct Device : RefCounted {
  int gpu = 0;                                               // first GPU (config key gpu=)
  // "gpus=N": ONE RTCDevice over N GPUs of the node (the north star's "ray batches shard embarrassingly across the 8 GPUs of one node with the BVH
  // replicated"; the reference shows a GPU device sitting behind the same RTCDevice in Scene::commit_task, kernels/common/scene.cpp:866-872).  Every
  // geometry buffer and every committed tree exists once per GPU (the build is deterministic: bit-identical replicas, nothing to broadcast); the batched
  // queries split their ray range contiguously over the replicas, one host thread per GPU, and every shard's results go straight into the caller's
  // array (no collective when the consumer is the host).  gpus[k] = HIP ordinal of replica k; "gpu_oversubscribe=1" lets replicas share a
  // GPU when the box has fewer than N (how the sharded path is tested on a 1-GPU box).
  std::vector<int> gpus{0};
  int wantGpus = 1; bool oversubscribe = false;
  unsigned shardMin = 16384;                                 // host-array / device-array batches of fewer rays than this per replica stay on replica 0 (config key shard_min)
  int verbose = 0;
  bool benchmark = false;
  unsigned instanceRefitMax = 32;                            // config key instance_refit_max=<n>: after n refits in a row of the top tree over moved instances, the next commit rebuilds it
  bool noInstanceRefit = false;                              // config key instance_refit=0: moved instances rebuild the top tree and concatenate the object trees again (A/B)
  bool coherentMemory = true;                                // config key coherent_memory=0: RTC_RAY_QUERY_FLAG_COHERENT queries never skip their packet sample on the strength of earlier queries (MI355_QUERY_COHERENT_NO_MEMORY)
  bool hostRegister = false;                                 // config key host_register=1: large host ray arrays are registered with the device for the duration of a query (zero-copy transfers; rounds 3 - 5) instead of staged
  bool hostInPlace = false;                                  // config key host_in_place=1: large host arrays are traced where they lie (registered + mapped), see replica_query
  bool deviceFilters = false;                                // config key device_filter_functions=1: RTCIntersectArguments::filter / RTCOccludedArguments::filter of the *Device entry points
                                                             // is the address of a __device__ function (include/embree4/rtcore.h, "device filter functions"); off: a non-NULL filter there is an error
  bool pollSmall = true;                                     // config key small_poll=0: a blocking single-ray call sleeps on the stream instead of polling it (A/B)
  bool smallInPlace = true;                                  // config key small_in_place=0: small host queries go through device staging like the others (A/B)
  bool packedLink = true;                                    // config key packed_link=0: host-array queries send whole records both ways (rounds 1 - 5) instead of 48 bytes up and only the written fields down
  unsigned pipelineMin = 262144, pipelineChunk = 131072;   // host-array queries of at least pipelineMin rays are cut into chunks of pipelineChunk rays (config keys host_pipeline_min / host_pipeline_chunk)
  mi355_build_params build;
  RTCErrorFunction errorFn = nullptr; void* errorFnPtr = nullptr;
  RTCMemoryMonitorFunction memFn = nullptr; void* memFnPtr = nullptr;
  std::mutex errMutex;
  std::map<size_t, ErrState> errors;                     // per thread (keyed by a thread-local token)
  std::string name;
  static size_t threadToken() { static thread_local char t; return (size_t)&t; }
  ErrState& err() { std::lock_guard<std::mutex> lk(errMutex); return errors[threadToken()]; }
  // the build arena of a GPU is shared by all devices on it (a commit takes it for its duration) and goes back to the driver with the LAST of them
  static std::mutex& countMutex() { static std::mutex m; return m; }
  static std::map<int, int>& liveOnGpu() { static std::map<int, int> c; return c; }
  bool counted = false;
  void countIn() { std::lock_guard<std::mutex> lk(countMutex()); for (int g : gpus) liveOnGpu()[g]++; counted = true; }
  ~Device() override {
    if (!counted) return;
    for (int g : gpus) {
      bool last; { std::lock_guard<std::mutex> lk(countMutex()); last = --liveOnGpu()[g] == 0; }
      if (last) mi355_release_build_scratch(g);
    }
  }
  size_t numReplicas() const { return gpus.size(); }
  // Device::memoryMonitor (kernels/common/device.cpp:332-345): every allocation the library keeps on the caller's behalf (buffers it owns, their
  // device copies, the committed BVH) is announced with +bytes before and -bytes after; a callback answering false fails the request with
  // RTC_ERROR_OUT_OF_MEMORY.  Build scratch is transient and not announced.
  void memoryMonitor(ssize_t bytes, bool post) {
    if (memFn && bytes != 0 && !memFn(memFnPtr, bytes, post) && bytes > 0) THROW(RTC_ERROR_OUT_OF_MEMORY, "memory monitor forced termination");
  }
};

void process_error(Device* dev, RTCError code, const char* str) {     // Device::process_error, device.cpp:312-330
  if (!dev) { if (g_threadError.error == RTC_ERROR_NONE) { g_threadError.error = code; g_threadError.msg = str ? str : ""; } return; }
  if (dev->verbose >= 1) fprintf(stderr, "Embree(MI355X): %s%s%s%s\n", rtcGetErrorString(code), str ? ", (" : "", str ? str : "", str ? ")" : "");
  if (dev->errorFn) dev->errorFn(dev->errorFnPtr, code, str);
  ErrState& e = dev->err();
  if (e.error == RTC_ERROR_NONE) { e.error = code; if (str && *str) e.msg = str; }
}

#define CATCH_BEGIN try {
#define CATCH_END(dev)                                                                         \
  } catch (const rtc_error& e) { process_error(dev, e.code, e.msg.c_str());                    \
  } catch (const std::bad_alloc&) { process_error(dev, RTC_ERROR_OUT_OF_MEMORY, "out of memory"); \
  } catch (const std::exception& e) { process_error(dev, RTC_ERROR_UNKNOWN, e.what());          \
  } catch (...) { process_error(dev, RTC_ERROR_UNKNOWN, "unknown exception caught"); }

void hip_check(hipError_t e, const char* what) {
  if (e == hipSuccess) return;
  std::string m = std::string(what) + ": " + hipGetErrorString(e);
  THROW(e == hipErrorOutOfMemory ? RTC_ERROR_OUT_OF_MEMORY : RTC_ERROR_UNKNOWN, m.c_str());
}
void core_check(int rc, const char* what) {
  if (rc == 0) return;
  std::string m = std::string(what) + ": " + mi355_last_error();
  THROW(rc == (int)hipErrorOutOfMemory ? RTC_ERROR_OUT_OF_MEMORY : RTC_ERROR_UNKNOWN, m.c_str());
}


// ---- host memory never meets the GPU directly (round 6) --------------------------------------------------------------------------------------------------
// Rounds 3 - 5 let the GPU work on the CALLER's host arrays: rtcIntersect1M / rtcOccluded1M registered the ray array with the device for the duration of the call
// (hipHostRegister: zero-copy uploads and downloads), geometry buffers and small batches went through hipMemcpy on pageable memory (which the runtime pins chunk by
// chunk above a size of its own).  Either way the GPU's view of those pages is a "userptr" mapping that the kernel driver has to keep in step with whatever the CPU side
// does to the pages -- NUMA balancing, huge-page collapse, compaction, a neighbouring free() that trims the heap.  On the round's boxes that goes wrong about once in ten
// runs of the GPU suite: "Memory access fault by GPU node-2 on address 0x5abc3ebda000. Reason: Unknown" -- a page-aligned address in the process's brk heap, inside or next
// to a ray array, in the middle of a host-array query or a download; the process is aborted by the runtime.  Measured with round 5's library as well (1 of 12 suite runs;
// this round's: 4 of 38; profiles/r06_host_memory_fault.md), so it is not new, it was luck.  The GPU now only ever touches memory this library allocated itself:
// hipHostMalloc'ed staging buffers (really pinned, not userptr), filled and drained by the CPU -- several threads for large arrays (copy_pool below).  "host_register=1"
// in the device config restores the registration of the caller's array (faster on a quiet box: the link instead of the CPU's memcpy is the limit).
struct CopyPool {                                              // a few threads that do nothing but memcpy: one 96 MB array at ~10 GB/s per thread is 10 ms, the link moves it in 2
  // (MI355_COPY_THREADS: 4 helpers + the caller.  On boxes that give the process 16 CPUs of 256 the best count moves with whatever else runs there: 2^20 closest-hit rays
  // through the packed link took 3.0 / 3.3 / 3.7 / 3.9 ms with 2 / 4 / 6 / 10 helpers in a process of its own (gpurun_out/r06zw, r06zx) and 3.5 / 3.0 / 3.4 ms with
  // 2 / 6 / 12 inside bench.py, whose reference worker pool is alive (r06zy); whole records: 4.3 ms either way)
  static constexpr size_t MIN_PART = (size_t)1 << 20;
  typedef void (*RangeFn)(void* ctx, size_t begin, size_t end);   // (a job of run(): items [begin, end) of whatever ctx describes)
  struct Job { char* dst; const char* src; size_t n; std::atomic<int>* left; RangeFn fn = nullptr; void* ctx = nullptr; };
  std::mutex mtx; std::condition_variable cv; std::vector<Job> jobs; std::vector<std::thread> workers; bool stop = false;
  void start(unsigned n) {
    for (unsigned i = 0; i < n; i++) workers.emplace_back([this]() {
      for (;;) {
        Job j;
        { std::unique_lock<std::mutex> lk(mtx); cv.wait(lk, [this]() { return stop || !jobs.empty(); }); if (stop && jobs.empty()) return; j = jobs.back(); jobs.pop_back(); }
        if (j.fn) j.fn(j.ctx, (size_t)j.dst, (size_t)j.dst + j.n); else memcpy(j.dst, j.src, j.n);
        j.left->fetch_sub(1, std::memory_order_release);
      }
    });
  }
  ~CopyPool() { { std::lock_guard<std::mutex> lk(mtx); stop = true; } cv.notify_all(); for (auto& t : workers) t.join(); }
  void copy(void* dst, const void* src, size_t n) {
    static const unsigned want = []() { const char* e = getenv("MI355_COPY_THREADS"); const long v = e ? atol(e) : 4; return (unsigned)(v < 0 ? 0 : (v > 32 ? 32 : v)); }();
    if (n < 2 * MIN_PART || want == 0) { memcpy(dst, src, n); return; }
    { std::lock_guard<std::mutex> lk(mtx); if (workers.empty()) start(want); }
    size_t parts = n / MIN_PART; if (parts > workers.size() + 1) parts = workers.size() + 1;
    const size_t per = ((n / parts) + 63) & ~(size_t)63;
    std::atomic<int> left{0};
    size_t ofs = per;                                           // (the caller copies the first part itself)
    { std::lock_guard<std::mutex> lk(mtx);
      for (; ofs < n; ofs += per) { left.fetch_add(1, std::memory_order_relaxed); jobs.push_back({(char*)dst + ofs, (const char*)src + ofs, n - ofs < per ? n - ofs : per, &left}); } }
    cv.notify_all();
    memcpy(dst, src, per < n ? per : n);
    while (left.load(std::memory_order_acquire) != 0) std::this_thread::yield();
  }
  // fn(ctx, begin, end) over the items [0, n), `bytesPerItem` of memory traffic each, cut like copy() cuts its bytes (the caller takes the first part itself)
  void run(RangeFn fn, void* ctx, size_t n, size_t bytesPerItem) {
    static const unsigned want = []() { const char* e = getenv("MI355_COPY_THREADS"); const long v = e ? atol(e) : 4; return (unsigned)(v < 0 ? 0 : (v > 32 ? 32 : v)); }();
    if (n * bytesPerItem < 2 * MIN_PART || want == 0) { fn(ctx, 0, n); return; }
    { std::lock_guard<std::mutex> lk(mtx); if (workers.empty()) start(want); }
    size_t parts = n * bytesPerItem / MIN_PART; if (parts > workers.size() + 1) parts = workers.size() + 1;
    const size_t per = (n + parts - 1) / parts;
    std::atomic<int> left{0};
    { std::lock_guard<std::mutex> lk(mtx);
      for (size_t b = per; b < n; b += per) { left.fetch_add(1, std::memory_order_relaxed); Job j{(char*)b, nullptr, n - b < per ? n - b : per, &left, fn, ctx}; jobs.push_back(j); } }
    cv.notify_all();
    fn(ctx, 0, per < n ? per : n);
    while (left.load(std::memory_order_acquire) != 0) std::this_thread::yield();
  }
};
static CopyPool& copy_pool() { static CopyPool* p = new CopyPool; return *p; }   // (never destroyed: worker threads must not be joined from a static destructor at exit)

// Pinned staging of one GPU for everything that is NOT a ray query (geometry buffers at commit; mi355_* helpers have their own): two buffers, so that the CPU fills one
// while the link empties the other.  Blocking, serialised per GPU.
struct HostStage {
  static constexpr size_t PIECE = (size_t)8 << 20;
  std::mutex mtx; char* buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; hipStream_t st = nullptr;
  void ensure(int gpu) {
    hip_check(hipSetDevice(gpu), "hipSetDevice");
    if (st) return;
    hip_check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate(host staging)");
    for (int k = 0; k < 2; k++) { void* h = nullptr; hip_check(hipHostMalloc(&h, PIECE, hipHostMallocPortable), "hipHostMalloc(host staging)"); buf[k] = (char*)h;
                                  hip_check(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming), "hipEventCreate"); }
  }
  void h2d(int gpu, void* dev, const void* host, size_t bytes) {
    std::lock_guard<std::mutex> lk(mtx); ensure(gpu);
    hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");   // (what the blocking hipMemcpy this replaces did: earlier work on the destination is over)
    size_t c = 0;
    for (size_t ofs = 0; ofs < bytes; ofs += PIECE, c++) {
      const size_t n = bytes - ofs < PIECE ? bytes - ofs : PIECE; const int k = (int)(c & 1u);
      if (c >= 2) hip_check(hipEventSynchronize(ev[k]), "hipEventSynchronize");
      copy_pool().copy(buf[k], (const char*)host + ofs, n);
      hip_check(hipMemcpyAsync((char*)dev + ofs, buf[k], n, hipMemcpyHostToDevice, st), "hipMemcpyAsync(staged H2D)");
      hip_check(hipEventRecord(ev[k], st), "hipEventRecord");
    }
    hip_check(hipStreamSynchronize(st), "hipStreamSynchronize");
  }
};
static HostStage& host_stage(int gpu) {
  static std::mutex m; static std::map<int, HostStage*> all;
  std::lock_guard<std::mutex> lk(m);
  HostStage*& h = all[gpu]; if (!h) h = new HostStage; return *h;
}

struct Buffer : RefCounted {
  Device* device; size_t bytes; char* host = nullptr; bool ownsHost = false;
  char* dev = nullptr; bool ownsDev = false; bool devDirty = true;   // the copy on the device's first GPU (or the application's own device memory there)
  std::vector<char*> peers;                                  // "gpus=N": the copies on replicas 1 .. N-1 (always owned)
  char* devAt(size_t k) const { return k == 0 ? dev : (k - 1 < peers.size() ? peers[k - 1] : nullptr); }
  Buffer(Device* d, size_t n, void* shared, void* sharedDev = nullptr) : device(d), bytes(n) {
    d->retain();
    if (shared) host = (char*)shared;
    else {
      try { d->memoryMonitor((ssize_t)n, false); } catch (...) { d->release(); throw; }
      host = (char*)aligned_alloc(64, ((n + 16 + 63) / 64) * 64); if (!host) { d->memoryMonitor(-(ssize_t)n, true); d->release(); throw std::bad_alloc(); } ownsHost = true; memset(host, 0, n);
    }
    if (sharedDev) { dev = (char*)sharedDev; devDirty = false; }
  }
  ~Buffer() override {
    if (ownsHost) { free(host); device->memoryMonitor(-(ssize_t)bytes, true); }
    if (ownsDev && dev) { hipSetDevice(device->gpu); hipFree(dev); device->memoryMonitor(-(ssize_t)bytes, true); }
    for (size_t k = 0; k < peers.size(); k++) if (peers[k]) { hipSetDevice(device->gpus[k + 1]); hipFree(peers[k]); device->memoryMonitor(-(ssize_t)bytes, true); }
    device->release();
  }
  void upload() {                                           // the reference's SYCL path copies in rtcCommitBuffer/rtcCommitGeometry too
    const bool sharedDevMem = dev && !ownsDev;               // the application's own device memory on the first GPU: its contents are the application's business
    if (devDirty || !dev) {
      hip_check(hipSetDevice(device->gpu), "hipSetDevice");
      if (!dev) {
        device->memoryMonitor((ssize_t)bytes, false);
        if (mi355_malloc_retry(device->gpu, bytes + 16, (void**)&dev) != 0) { dev = nullptr; device->memoryMonitor(-(ssize_t)bytes, true); THROW(RTC_ERROR_OUT_OF_MEMORY, "hipMalloc(geometry buffer)"); }
        ownsDev = true;
      }
      if (bytes) host_stage(device->gpu).h2d(device->gpu, dev, host, bytes);   // (staged: the GPU never reads the application's pages, see "host memory never meets the GPU")
    } else if (!sharedDevMem && peers.size() + 1 == device->numReplicas()) return;
    // replicas: from the host copy, or -- memory the application shared as device memory -- from the first GPU, peer to peer (xGMI)
    if (peers.size() + 1 < device->numReplicas()) peers.resize(device->numReplicas() - 1, nullptr);
    for (size_t k = 0; k < peers.size(); k++) {
      const int g = device->gpus[k + 1];
      hip_check(hipSetDevice(g), "hipSetDevice");
      if (!peers[k]) {
        device->memoryMonitor((ssize_t)bytes, false);
        if (mi355_malloc_retry(g, bytes + 16, (void**)&peers[k]) != 0) { peers[k] = nullptr; device->memoryMonitor(-(ssize_t)bytes, true); THROW(RTC_ERROR_OUT_OF_MEMORY, "hipMalloc(geometry buffer replica)"); }
      }
      if (!bytes) continue;
      if (sharedDevMem) hip_check(hipMemcpyPeer(peers[k], g, dev, device->gpu, bytes), "hipMemcpyPeer(geometry buffer)");
      else host_stage(g).h2d(g, peers[k], host, bytes);
    }
    hip_check(hipSetDevice(device->gpu), "hipSetDevice");
    devDirty = false;
  }
};

struct BufferView { Buffer* buf = nullptr; size_t offset = 0, stride = 0; unsigned num = 0; };

struct Scene;
static std::atomic<unsigned long long> g_commitSerial{0};
static std::atomic<unsigned long long> g_geomSerial{0};
struct Geometry : RefCounted {
  Device* device; RTCGeometryType type;
  const unsigned long long serial = ++g_geomSerial;          // process-unique: what a tree was built from is remembered by serial, never by address (a freed geometry's address comes back)
  BufferView vertices, indices;
  std::map<unsigned, Buffer*> attribs;                     // vertex attributes: kept alive for the caller, unused by the kernels (rtcInterpolate reads them on the host)
  std::map<unsigned, BufferView> attribViews;
  unsigned mask = 1;                                        // Geometry ctor, geometry.cpp:48
  bool enabled = true, modified = true, committed = false;
  RTCBuildQuality quality = RTC_BUILD_QUALITY_MEDIUM;
  Scene* object = nullptr;                           // RTC_GEOMETRY_TYPE_INSTANCE: the instanced scene (retained) and local2world as vx, vy, vz, p
  float l2w[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
  unsigned topoCounter = 0, dataCounter = 0;                // bumped when buffers are (re)bound / the index data changes; when vertex data or the mask changes
  void* userPtr = nullptr;
  RTCFilterFunctionN intersectFilter = nullptr, occludedFilter = nullptr;   // host callbacks: run between launches by the host-array entry points (filtered_query)
  bool argFilter = false;                                   // rtcSetGeometryEnableFilterFunctionFromArguments
  bool hasRule = false; RTCFilterRule rule{}; std::vector<unsigned> ruleBits; unsigned ruleCounter = 0;   // device-side filter rule (rtcSetGeometryFilterRule): goes to the GPU with the next rtcCommitScene
  std::atomic<int> attached{0};
  Geometry(Device* d, RTCGeometryType t) : device(d), type(t) { d->retain(); }
  ~Geometry() override;
  void dtor_body() {
    if (vertices.buf) vertices.buf->release();
    if (indices.buf) indices.buf->release();
    for (auto& kv : attribs) kv.second->release();
    device->release();
  }
  void setBuffer(RTCBufferType t, unsigned slot, RTCFormat fmt, Buffer* b, size_t off, size_t stride, size_t num) {
    if (((size_t)(b->host) + off) & 3 || (stride & 3)) THROW(RTC_ERROR_INVALID_OPERATION, "data must be 4 bytes aligned");
    if (num > 0xFFFFFFFFull) THROW(RTC_ERROR_INVALID_ARGUMENT, "buffer too large");
    if (stride > 0xFFFFFFFFull) THROW(RTC_ERROR_INVALID_ARGUMENT, "stride too large");
    BufferView* v = nullptr;
    if (t == RTC_BUFFER_TYPE_VERTEX) {
      if (fmt != RTC_FORMAT_FLOAT3) THROW(RTC_ERROR_INVALID_OPERATION, "invalid vertex buffer format");
      if (stride * num > 16ull * 1024 * 1024 * 1024) THROW(RTC_ERROR_INVALID_OPERATION, "vertex buffer can be at most 16GB large");
      if (slot != 0) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid vertex buffer slot");
      v = &vertices;
    } else if (t == RTC_BUFFER_TYPE_INDEX) {
      if (slot != 0) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid buffer slot");
      if (fmt != (type == RTC_GEOMETRY_TYPE_QUAD ? RTC_FORMAT_UINT4 : RTC_FORMAT_UINT3)) THROW(RTC_ERROR_INVALID_OPERATION, "invalid index buffer format");
      v = &indices;
    } else if (t == RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE) {
      if (fmt < RTC_FORMAT_FLOAT || fmt > RTC_FORMAT_FLOAT16) THROW(RTC_ERROR_INVALID_OPERATION, "invalid vertex attribute buffer format");
      b->retain();                                          // accepted and kept alive, unused on the GPU path (no rtcInterpolate)
      if (attribs.count(slot)) attribs[slot]->release();
      attribs[slot] = b;
      { BufferView av; av.buf = b; av.offset = off; av.stride = stride; av.num = (unsigned)num; attribViews[slot] = av; }
      return;
    } else THROW(RTC_ERROR_INVALID_ARGUMENT, "unknown buffer type");
    const size_t elem = (t == RTC_BUFFER_TYPE_INDEX && type == RTC_GEOMETRY_TYPE_QUAD) ? 16 : 12;
    // (no "stride >= element size": the reference accepts elements that overlap -- BufferStrideTest, tutorials/verify/verify.cpp:995-1008, binds UINT4 quad indices with a
    // 12-byte stride -- and so do the kernels here, which read element i at offset + i * stride whatever the stride)
    if (num && off + stride * (num - 1) + elem > b->bytes) THROW(RTC_ERROR_INVALID_ARGUMENT, "buffer too small for view");
    b->retain();
    if (v->buf) v->buf->release();
    v->buf = b; v->offset = off; v->stride = stride; v->num = (unsigned)num;
    modified = true; committed = false; topoCounter++;
  }
};

// Ray staging is kept per calling thread (Device::threadToken()).  The workers a sharded host query starts for replicas 1 .. N-1 are fresh threads every call:
// keyed by their own thread-local address every call could leave another staging buffer behind.  They carry the APPLICATION thread's key instead.
static thread_local size_t t_stagingKey = 0;
static inline size_t staging_key() { return t_stagingKey ? t_stagingKey : Device::threadToken(); }

// What a scene keeps on ONE GPU.  A device over N GPUs ("gpus=N") commits N bit-identical replicas (the build is deterministic) and shards ray batches over them.
struct Replica {
  int gpu = 0;
  mi355_bvh_t bvh = nullptr; ssize_t bvhBytes = 0;          // what queries traverse (== flat unless the scene has instances)
  mi355_bvh_t flat = nullptr; ssize_t flatBytes = 0;        // the tree of this scene's own triangles / quads: what an instance of this scene refers to
  // host-pointer query staging (device memory), one per calling thread
  static constexpr int PIPE = 4;
  std::mutex mtx;
  hipStream_t pipe[PIPE] = {nullptr, nullptr, nullptr, nullptr};   // large host-array queries: upload, download, two compute streams (pipelined_query)
  std::vector<hipEvent_t> pipeEvents;                       // ... and two events per chunk
  std::mutex pipeMtx;                                       // one pipelined query per replica at a time: the four streams, the events and the streams' status words are shared
  char* pinUp[2] = {nullptr, nullptr}; char* pinDown[2] = {nullptr, nullptr}; size_t pinCap = 0;   // ... and the pinned staging of the chunks (staged_query): two on the way up, two on the way down
  hipEvent_t pinUpEv[2] = {nullptr, nullptr}, pinDownEv[2] = {nullptr, nullptr};
  char* packUp[2] = {nullptr, nullptr}; char* packDown[2] = {nullptr, nullptr}; size_t packCap = 0; hipEvent_t packUpEv[2] = {nullptr, nullptr};   // ... and, on the device, the packed chunks of a query that sends only what is read / written (staged_query)
  struct Staging { char* d = nullptr; size_t cap = 0; char* h = nullptr; char* hd = nullptr; };   // h / hd: 4 KiB of pinned host memory and its device address
  std::map<size_t, Staging> staging;
  static constexpr size_t SMALL_BYTES = 4096;               // queries of up to this many bytes (rtcIntersect1 .. a few dozen rays) are traced in place in pinned host memory
  // One blocking rtcIntersect1 call used to be hipMemcpy up + launch + hipMemcpy down + status read = four round trips (77 us); the kernel now reads the ray
  // from and writes the hit to a pinned, device-mapped buffer of the calling thread: one launch and one wait.
  char* stage_host(char** devAddr) {
    std::lock_guard<std::mutex> lk(mtx);
    Staging& s = staging[staging_key()];
    if (!s.h) {
      hip_check(hipSetDevice(gpu), "hipSetDevice");
      void* h = nullptr; void* d = nullptr;
      hip_check(hipHostMalloc(&h, SMALL_BYTES, hipHostMallocMapped | hipHostMallocPortable), "hipHostMalloc(small-query staging)");
      hip_check(hipHostGetDevicePointer(&d, h, 0), "hipHostGetDevicePointer");
      s.h = (char*)h; s.hd = (char*)d;
    }
    *devAddr = s.hd;
    return s.h;
  }
  hipStream_t shardStream = nullptr; hipEvent_t shardIn = nullptr, shardOut = nullptr;   // device-array queries sharded over the replicas (sharded_device_query)
  // Small host queries (rtcIntersect1 ... a few dozen rays, traced in place in pinned memory) run on a NON-BLOCKING stream of their own: on the legacy null stream a poll
  // (hipStreamQuery) and the launch itself have to look at every blocking stream of the process first -- 49 us per call inside bench.py, which holds a dozen streams, against
  // 33 us in a process that holds none.  Their inputs are copied from host memory at call time and the committed tree does not change under a query: nothing they could be
  // ordered against lives on the null stream.
  hipStream_t smallStream = nullptr;
  hipStream_t small_stream() {
    std::lock_guard<std::mutex> lk(mtx);
    if (!smallStream) { hip_check(hipSetDevice(gpu), "hipSetDevice"); hip_check(hipStreamCreateWithFlags(&smallStream, hipStreamNonBlocking), "hipStreamCreate(small queries)"); }
    return smallStream;
  }
  char* stage(size_t bytes) {
    std::lock_guard<std::mutex> lk(mtx);
    Staging& s = staging[staging_key()];
    if (s.cap < bytes) {
      hip_check(hipSetDevice(gpu), "hipSetDevice");
      if (s.d) hipFree(s.d);
      s.cap = bytes < 4096 ? 4096 : bytes + bytes / 4;
      s.d = nullptr;
      core_check(mi355_malloc_retry(gpu, s.cap, (void**)&s.d), "hipMalloc(ray staging)");
    }
    return s.d;
  }
  void release_all(Device* device) {
    hipSetDevice(gpu);
    if (bvh && bvh != flat) { mi355_bvh_destroy(bvh); device->memoryMonitor(-bvhBytes, true); }
    if (flat) { mi355_bvh_destroy(flat); device->memoryMonitor(-flatBytes, true); }
    bvh = flat = nullptr;
    for (auto& kv : staging) { if (kv.second.d) hipFree(kv.second.d); if (kv.second.h) hipHostFree(kv.second.h); }
    for (int k = 0; k < PIPE; k++) if (pipe[k]) hipStreamDestroy(pipe[k]);
    for (int k = 0; k < 2; k++) { if (pinUp[k]) hipHostFree(pinUp[k]); if (pinDown[k]) hipHostFree(pinDown[k]); if (pinUpEv[k]) hipEventDestroy(pinUpEv[k]); if (pinDownEv[k]) hipEventDestroy(pinDownEv[k]); pinUp[k] = pinDown[k] = nullptr; }
    for (hipEvent_t e : pipeEvents) hipEventDestroy(e);
    for (int k = 0; k < 2; k++) { if (packUp[k]) hipFree(packUp[k]); if (packDown[k]) hipFree(packDown[k]); if (packUpEv[k]) hipEventDestroy(packUpEv[k]); packUp[k] = packDown[k] = nullptr; packUpEv[k] = nullptr; } packCap = 0;
    if (shardStream) hipStreamDestroy(shardStream);
    if (smallStream) hipStreamDestroy(smallStream);
    if (shardIn) hipEventDestroy(shardIn);
    if (shardOut) hipEventDestroy(shardOut);
  }
};

// runs fn(k) for every replica, one host thread per GPU (replica 0 on the calling thread); the first exception is rethrown on the caller
template <typename F> static void for_each_replica(size_t n, F fn) {
  if (n == 1) { fn((size_t)0); return; }
  std::vector<std::exception_ptr> err(n);
  std::vector<std::thread> th;
  const size_t callerKey = staging_key();
  for (size_t k = 1; k < n; k++) th.emplace_back([&, k, callerKey]() { t_stagingKey = callerKey; try { fn(k); } catch (...) { err[k] = std::current_exception(); } });
  try { fn((size_t)0); } catch (...) { err[0] = std::current_exception(); }
  for (auto& t : th) t.join();
  for (size_t k = 0; k < n; k++) if (err[k]) std::rethrow_exception(err[k]);
}

struct Scene : RefCounted {
  Device* device;
  std::mutex mtx;
  std::map<unsigned, Geometry*> geoms;
  RTCSceneFlags flags = RTC_SCENE_FLAG_NONE; RTCBuildQuality quality = RTC_BUILD_QUALITY_MEDIUM;
  bool committed = false, modified = true;
  const unsigned long long serial = ++g_geomSerial;        // (instances remember the scene they were built over by this, not by address)
  std::vector<std::unique_ptr<Replica>> reps;               // one per GPU of the device
  struct BuiltFrom { unsigned id; unsigned long long g; RTCBuildQuality q; unsigned topo, data, rule; };   // what the current tree was built from: decides rebuild vs refit vs nothing to do; g = Geometry::serial
  std::vector<BuiltFrom> builtFrom; unsigned builtFlags = 0;
  struct InstFrom { unsigned id; unsigned long long g; unsigned long long object; unsigned topo, data; unsigned long long objSerial; };   // g = Geometry::serial, object = Scene::serial
  std::vector<InstFrom> builtInst;
  unsigned instRefitsInARow = 0;                              // commits in a row that refitted the top tree over moved instances (bounded: instance_refit_max)
  unsigned long long commitSerial = 0;                       // changes with every commit that built or refitted something (instances of this scene notice)
  RTCBounds bounds;
  RTCProgressMonitorFunction progress = nullptr; void* progressPtr = nullptr;
  Scene(Device* d) : device(d) {
    d->retain(); setEmptyBounds();
    for (int g : d->gpus) { reps.emplace_back(new Replica); reps.back()->gpu = g; }
  }
  void setEmptyBounds() {
    bounds.lower_x = bounds.lower_y = bounds.lower_z = INFINITY; bounds.upper_x = bounds.upper_y = bounds.upper_z = -INFINITY;
    bounds.align0 = bounds.align1 = 0;
  }
  ~Scene() override {
    for (auto& kv : geoms) { kv.second->attached--; kv.second->release(); }
    for (auto& r : reps) r->release_all(device);
    hipSetDevice(device->gpu);
    device->release();
  }
  mi355_bvh_t bvh0() const { return reps[0]->bvh; }
  mi355_bvh_t flat0() const { return reps[0]->flat; }
  void commit() {
    std::lock_guard<std::mutex> lk(mtx);
    for (auto& kv : geoms) {                                 // std::map => ascending geomID
      Geometry* g = kv.second;
      if (!g->enabled) continue;
      if (!g->vertices.buf || !g->indices.buf) continue;     // a mesh without buffers has no primitives
      if (!g->committed) THROW(RTC_ERROR_INVALID_OPERATION, "geometry attached to the scene was modified but not committed");
    }
    auto meshes_of = [&](size_t k) {                         // the mesh table as replica k sees it (its own copies of the buffers)
      std::vector<mi355_mesh> meshes;
      for (auto& kv : geoms) {
        Geometry* g = kv.second;
        if (!g->enabled || !g->vertices.buf || !g->indices.buf) continue;
        mi355_mesh m;
        m.d_vertices = g->vertices.buf->devAt(k) + g->vertices.offset; m.vertex_stride = g->vertices.stride; m.num_vertices = g->vertices.num;
        m.d_indices = g->indices.buf->devAt(k) + g->indices.offset; m.index_stride = g->indices.stride; m.num_triangles = g->indices.num;
        m.geom_id = kv.first; m.mask = g->mask; m.quads = g->type == RTC_GEOMETRY_TYPE_QUAD ? 1u : 0u; m.reserved = 0;
        meshes.push_back(m);
      }
      return meshes;
    };
    if (progress && !progress(progressPtr, 0.0)) THROW(RTC_ERROR_CANCELLED, "progress monitor forced termination");
    mi355_build_params bp = device->build;
    bp.robust = (flags & RTC_SCENE_FLAG_ROBUST) ? 1u : 0u;   // scene.cpp:180-188: robust scenes get Triangle4v leaves + the Pluecker intersector
    if (quality == RTC_BUILD_QUALITY_LOW) bp.quality = 1u;    // scene.cpp:195-206: low quality = the Morton builder
    if (quality == RTC_BUILD_QUALITY_HIGH) bp.quality = 2u;   // high quality = the presplit SAH builder (bvh_builder_sah_spatial.cpp:93-125)
    std::vector<BuiltFrom> from; bool wantRefit = (flags & RTC_SCENE_FLAG_DYNAMIC) != 0;
    for (auto& kv : geoms) {
      Geometry* g = kv.second;
      if (!g->enabled || !g->vertices.buf || !g->indices.buf) continue;
      from.push_back({kv.first, g->serial, g->quality, g->topoCounter, g->dataCounter, g->ruleCounter});
      wantRefit = wantRefit || g->quality == RTC_BUILD_QUALITY_REFIT;
    }
    bp.refit = wantRefit ? 1u : 0u;
    const unsigned nowFlags = (bp.robust ? 1u : 0u) | (bp.quality << 1) | (bp.refit << 8);
    // Scene::commit returns at once when nothing was modified since the last commit (kernels/common/scene.cpp:831, isModified()): same geometries, same
    // buffer bindings / index data / vertex data / masks / transforms (the per-geometry counters), same enable state, same flags and quality.  Instances
    // also look at the scene they refer to: a re-committed object scene has a new commit number.
    std::vector<InstFrom> instFrom;
    for (auto& kv : geoms) {
      Geometry* g = kv.second;
      if (g->type != RTC_GEOMETRY_TYPE_INSTANCE || !g->enabled || !g->object) continue;
      instFrom.push_back({kv.first, g->serial, g->object->serial, g->topoCounter, g->dataCounter, g->object->commitSerial});
    }
    const bool haveTree = committed && reps[0]->bvh != nullptr, haveFlat = committed && reps[0]->flat != nullptr;
    if (haveTree && !modified && nowFlags == builtFlags && from.size() == builtFrom.size() && instFrom.size() == builtInst.size()) {   // (attach / detach set `modified`)
      bool same = true;
      for (size_t i = 0; same && i < from.size(); i++) { const BuiltFrom &a = from[i], &b = builtFrom[i]; same = a.id == b.id && a.g == b.g && a.topo == b.topo && a.data == b.data && a.rule == b.rule; }
      for (size_t i = 0; same && i < instFrom.size(); i++) { const InstFrom &a = instFrom[i], &b = builtInst[i]; same = a.id == b.id && a.g == b.g && a.object == b.object && a.topo == b.topo && a.data == b.data && a.objSerial == b.objSerial; }
      if (same) { if (progress) progress(progressPtr, 1.0); return; }
    }
    // Refit instead of rebuild (the reference: BVHNRefitT for RTC_BUILD_QUALITY_REFIT meshes of a dynamic scene, kernels/bvh/bvh_refit.cpp):
    // same geometries with the same buffer bindings and index data, and every geometry whose vertices / mask changed asks for REFIT.
    bool refit = haveFlat && from.size() == builtFrom.size() && nowFlags == builtFlags && !from.empty();
    for (size_t i = 0; refit && i < from.size(); i++) {
      const BuiltFrom &a = from[i], &b = builtFrom[i];
      refit = a.id == b.id && a.g == b.g && a.topo == b.topo && (a.data == b.data || a.q == RTC_BUILD_QUALITY_REFIT);
    }
    bool keepFlat = false;
    {                                                        // the scene's own triangles / quads are what they were: only instances moved, keep the flat tree
      bool flatSame = haveFlat && from.size() == builtFrom.size() && nowFlags == builtFlags;
      for (size_t i = 0; flatSame && i < from.size(); i++) { const BuiltFrom &a = from[i], &b = builtFrom[i]; flatSame = a.id == b.id && a.g == b.g && a.topo == b.topo && a.data == b.data; }
      if (flatSame) { keepFlat = true; refit = false; }
    }
    if (refit) { mi355_bvh_info fi; mi355_bvh_get_info(reps[0]->flat, &fi); refit = fi.bytes_refit != 0; }   // the tree was built to be refitted
    // ---- instances (RTC_GEOMETRY_TYPE_INSTANCE, one level): top tree over their world boxes + copies of the instanced scenes' flat trees
    struct InstGeom { Geometry* g; unsigned id; };
    std::vector<InstGeom> instGeoms;
    for (auto& kv : geoms) {
      Geometry* g = kv.second;
      if (g->type != RTC_GEOMETRY_TYPE_INSTANCE || !g->enabled || !g->object) continue;
      if (!g->committed) THROW(RTC_ERROR_INVALID_OPERATION, "geometry attached to the scene was modified but not committed");
      if (g->object == this) THROW(RTC_ERROR_INVALID_OPERATION, "a scene cannot instance itself");
      if (!g->object->committed || !g->object->flat0()) THROW(RTC_ERROR_INVALID_OPERATION, "the instanced scene has to be committed before the scene that instances it");
      instGeoms.push_back({g, kv.first});
    }
    // ---- device-side filter rules (rtcSetGeometryFilterRule): the table every replica uploads
    std::vector<uint32_t> ruleTable; uint32_t ruleGeoms = geoms.empty() ? 0u : geoms.rbegin()->first + 1u;
    { bool any = false;
      // (device filter FUNCTIONS, config device_filter_functions=1: a geometry that enabled the argument filter says so in bit 16 of its first word and brings its user pointer)
      const bool fptr = device->deviceFilters;
      for (auto& kv : geoms) any = any || (kv.second->enabled && ((kv.second->hasRule && kv.second->rule.kinds != 0u) || (fptr && kv.second->argFilter)));
      if (any) {
        ruleTable.assign((size_t)ruleGeoms * 12u, 0u);
        for (auto& kv : geoms) {
          Geometry* g = kv.second;
          const bool hasRule = g->hasRule && g->rule.kinds != 0u, argF = fptr && g->argFilter;
          if (!g->enabled || !(hasRule || argF)) continue;
          const RTCFilterRule q = hasRule ? g->rule : RTCFilterRule{};
          uint32_t e[12] = {0}; float f[4] = {q.tmin, q.tmax, q.umax, q.vmax};
          if (argF) { e[0] |= 1u << 16; const unsigned long long up = (unsigned long long)(uintptr_t)g->userPtr; e[10] = (uint32_t)up; e[11] = (uint32_t)(up >> 32); }
          e[0] |= (q.kinds & 0xFFu) | ((q.apply & 3u) << 8); e[1] = q.modulus; e[2] = q.remainder; e[3] = (q.primFactor & 0xFFFFu) | ((q.geomFactor & 0xFFFFu) << 16);
          memcpy(&e[4], f, 16);
          if ((q.kinds & RTC_FILTER_RULE_PRIMITIVE_BITS) && !g->ruleBits.empty()) { e[8] = (uint32_t)ruleTable.size(); e[9] = q.numBits; ruleTable.insert(ruleTable.end(), g->ruleBits.begin(), g->ruleBits.end()); }
          else e[0] &= ~(uint32_t)RTC_FILTER_RULE_PRIMITIVE_BITS;
          memcpy(&ruleTable[(size_t)kv.first * 12u], e, 48);
        }
      }
    }
    // ---- instances that only MOVED (rtcSetGeometryTransform / mask; same instances, same objects, objects not re-committed, the scene's own geometry and rules as
    // they were): the top tree is refitted in place and the records get their new transforms -- no top build, no copy of the object trees (mi355_bvh_refit_instanced)
    bool instMoveOnly = keepFlat && !instGeoms.empty() && haveTree && nowFlags == builtFlags && instFrom.size() == builtInst.size() && !device->noInstanceRefit;
    for (size_t i = 0; instMoveOnly && i < instFrom.size(); i++) { const InstFrom &a = instFrom[i], &b = builtInst[i]; instMoveOnly = a.id == b.id && a.g == b.g && a.object == b.object && a.topo == b.topo && a.objSerial == b.objSerial; }
    for (size_t i = 0; instMoveOnly && i < from.size(); i++) instMoveOnly = from[i].rule == builtFrom[i].rule;
    //  A refit keeps the top tree's SHAPE: instances that swap places or scatter leave it worse with every commit, and nothing bounded that.  Every
    // `instance_refit_max`-th commit in a row that would refit builds the top tree anew instead (default 32; the reference rebuilds its top level on every commit of a
    // dynamic two-level scene, bvh_builder_twolevel.cpp -- the refit is this library's shortcut, so the bound errs on the side of rebuilding).
    if (instMoveOnly && instRefitsInARow >= device->instanceRefitMax) instMoveOnly = false;
    // ---- every replica does the same thing on its own GPU, side by side (one host thread per GPU; the build is deterministic, so the replicas come out bit-identical)
    std::vector<int> didRefit(reps.size(), 0);
    std::atomic<bool> refitBroken{false};
    try {
    for_each_replica(reps.size(), [&](size_t k) {
      Replica& r = *reps[k];
      hip_check(hipSetDevice(r.gpu), "hipSetDevice");
      const std::vector<mi355_mesh> meshes = meshes_of(k);
      mi355_bvh_info info;
      bool done = keepFlat;
      if (refit) {
        const int rc = mi355_bvh_refit(r.flat, meshes.data(), (uint32_t)meshes.size(), nullptr);
        if (rc == 0) { done = true; didRefit[k] = 1; }
        else if (rc != MI355_REFIT_IMPOSSIBLE) { refitBroken = true; if (rc != MI355_REFIT_BROKEN) core_check(rc, "BVH refit"); }   // a refit that stopped half way leaves no usable tree
      }
      if (!done) {
        mi355_bvh_t nb = nullptr;
        core_check(mi355_bvh_build(r.gpu, meshes.data(), (uint32_t)meshes.size(), &bp, nullptr, &nb), "BVH build");
        mi355_bvh_get_info(nb, &info);
        const ssize_t newBytes = (ssize_t)(info.bytes_nodes + info.bytes_triangles + info.bytes_refit);
        try { device->memoryMonitor(newBytes, false); } catch (...) { mi355_bvh_destroy(nb); throw; }   // the old tree stays in place
        if (r.flat) { mi355_bvh_destroy(r.flat); device->memoryMonitor(-r.flatBytes, true); if (r.bvh == r.flat) r.bvh = nullptr; }
        r.flat = nb; r.flatBytes = newBytes;
      }
      // device-side filter rules of this scene's geometries: one 12-word entry per geometry id + the bit arrays (mi355_bvh_set_filter_rules)
      core_check(mi355_bvh_set_filter_rules(r.flat, ruleTable.empty() ? nullptr : ruleTable.data(), ruleTable.size(), ruleGeoms), "filter rules");
      std::vector<mi355_instance> insts;
      for (const InstGeom& ig : instGeoms) {
        mi355_instance in; in.object = ig.g->object->reps[k]->flat;   // a second level inside the object is dropped, like the reference does at RTC_MAX_INSTANCE_LEVEL_COUNT = 1 (instance_stack.h:36-47)
        memcpy(in.local2world, ig.g->l2w, sizeof(in.local2world)); in.inst_id = ig.id; in.mask = ig.g->mask;
        insts.push_back(in);
      }
      if (instMoveOnly && r.bvh && r.bvh != r.flat) {
        const int rc = mi355_bvh_refit_instanced(r.bvh, insts.data(), (uint32_t)insts.size(), nullptr);
        if (rc == 0) { didRefit[k] = 1; return; }
        if (rc != MI355_REFIT_IMPOSSIBLE && rc != MI355_REFIT_BROKEN) core_check(rc, "instanced BVH refit");   // (impossible / broken: the tree is built anew below)
      }
      if (r.bvh && r.bvh != r.flat) { mi355_bvh_destroy(r.bvh); device->memoryMonitor(-r.bvhBytes, true); r.bvhBytes = 0; }
      r.bvh = r.flat;
      if (!instGeoms.empty()) {
        mi355_bvh_t nb = nullptr;
        core_check(mi355_bvh_build_instanced(r.gpu, r.flat, insts.data(), (uint32_t)insts.size(), &bp, nullptr, &nb), "instanced BVH build");
        mi355_bvh_get_info(nb, &info);
        const ssize_t newBytes = (ssize_t)(info.bytes_nodes + info.bytes_triangles);
        try { device->memoryMonitor(newBytes, false); } catch (...) { mi355_bvh_destroy(nb); throw; }
        r.bvh = nb; r.bvhBytes = newBytes;
      }
    });
    } catch (...) {
      // a refit that stopped half way and a rebuild that failed leave no usable tree; and with several replicas a failure on ONE GPU (out of memory, the memory
      // monitor's veto) leaves the others with the NEW tree: sharded queries would then answer from different geometry depending on the shard a ray falls in
      // -- the scene counts as not committed until a commit goes through on every GPU
      if (refitBroken || reps.size() > 1) committed = false;
      throw;
    }
    builtFrom = from; builtFlags = nowFlags;
    mi355_bvh_info info; mi355_bvh_get_info(reps[0]->bvh, &info);
    setEmptyBounds();
    if (info.num_triangles) {
      bounds.lower_x = info.bounds_lower[0]; bounds.lower_y = info.bounds_lower[1]; bounds.lower_z = info.bounds_lower[2];
      bounds.upper_x = info.bounds_upper[0]; bounds.upper_y = info.bounds_upper[1]; bounds.upper_z = info.bounds_upper[2];
    }
    if (device->verbose >= 2 || device->benchmark)           // BVHN::postBuild prints BENCHMARK_BUILD, bvh.cpp:175-179
      printf("%s %.3f ms %.3f Mprims/s sah %.4f nodes %llu tris %llu bytes %llu\n", (keepFlat || didRefit[0]) ? "BENCHMARK_REFIT" : "BENCHMARK_BUILD", info.build_ms,
             info.build_ms > 0 ? info.num_triangles / (info.build_ms * 1e3) : 0.0, info.sah, (unsigned long long)info.num_nodes,
             (unsigned long long)info.num_triangles, (unsigned long long)(info.bytes_nodes + info.bytes_triangles));
    if (progress) progress(progressPtr, 1.0);
    builtInst = instFrom;
    { bool all = !instGeoms.empty(); for (int d : didRefit) all = all && d != 0; instRefitsInARow = (all && instMoveOnly) ? instRefitsInARow + 1u : 0u; }
    committed = true; modified = false; commitSerial = ++g_commitSerial;
  }
};

Geometry::~Geometry() { if (object) object->release(); dtor_body(); }

Device* dev_of(RTCDevice h) { if (!h) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid argument"); return (Device*)h; }
Scene* scene_of(RTCScene h) { if (!h) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid argument"); return (Scene*)h; }
Geometry* geom_of(RTCGeometry h) { if (!h) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid argument"); return (Geometry*)h; }

// State::parseString (kernels/common/state.cpp:224): comma/space separated key=value list
void parse_config(Device* d, const char* cfg) {
  if (!cfg) return;
  std::string s(cfg); size_t i = 0;
  while (i < s.size()) {
    size_t j = s.find_first_of(", ", i); if (j == std::string::npos) j = s.size();
    std::string tok = s.substr(i, j - i); i = j + 1;
    if (tok.empty()) continue;
    size_t eq = tok.find('='); std::string k = tok.substr(0, eq), v = eq == std::string::npos ? "" : tok.substr(eq + 1);
    if (k == "verbose") d->verbose = atoi(v.c_str());
    else if (k == "benchmark") d->benchmark = atoi(v.c_str()) != 0;
    else if (k == "gpu") d->gpu = atoi(v.c_str());
    else if (k == "gpus") d->wantGpus = atoi(v.c_str());                                              // one RTCDevice over N GPUs: replicated BVH, sharded ray batches
    else if (k == "gpu_oversubscribe") d->oversubscribe = atoi(v.c_str()) != 0;
    else if (k == "shard_min") d->shardMin = atol(v.c_str()) >= 1 ? (unsigned)atol(v.c_str()) : 1u;
    else if (k == "max_leaf" || k == "max_triangles_per_leaf") d->build.max_leaf = (uint32_t)atoi(v.c_str());
    else if (k == "min_leaf") d->build.min_leaf = (uint32_t)atoi(v.c_str());
    else if (k == "leaf_block_shift") d->build.sah_block_shift = (uint32_t)atoi(v.c_str());
    else if (k == "small_threshold") d->build.small_threshold = (uint32_t)atoi(v.c_str());
    else if (k == "quality") d->build.quality = (v == "low" || v == "1") ? 1u : (v == "high" || v == "2") ? 2u : 0u;
    else if (k == "presplits") d->build.presplits = atoi(v.c_str()) != 0 ? 1u : 0u;                    // state.cpp:443
    else if (k == "max_spatial_split_replications") d->build.split_factor = (float)atof(v.c_str());   // state.cpp:437
    else if (k == "host_pipeline_min") d->pipelineMin = (unsigned)atol(v.c_str());
    else if (k == "small_in_place") d->smallInPlace = atoi(v.c_str()) != 0;
    else if (k == "small_poll") d->pollSmall = atoi(v.c_str()) != 0;
    else if (k == "device_filter_functions") d->deviceFilters = atoi(v.c_str()) != 0;
    else if (k == "host_pipeline_chunk") d->pipelineChunk = atol(v.c_str()) >= 1024 ? (unsigned)atol(v.c_str()) : 1024u;
    else if (k == "int_cost") d->build.int_cost = (float)atof(v.c_str());
    else if (k == "top_splits") d->build.top_splits = atoi(v.c_str()) != 0 ? 1u : 0u;                 // MEDIUM builds: references that dwarf all others are cut into grid pieces first
    else if (k == "top_split_min") d->build.top_split_min = (uint32_t)atol(v.c_str());
    else if (k == "instance_refit") d->noInstanceRefit = atoi(v.c_str()) == 0;
    else if (k == "instance_refit_max") d->instanceRefitMax = (unsigned)atol(v.c_str());
    else if (k == "coherent_memory") d->coherentMemory = atoi(v.c_str()) != 0;
    else if (k == "packed_link") d->packedLink = atoi(v.c_str()) != 0;
    else if (k == "host_register") d->hostRegister = atoi(v.c_str()) != 0;                           // the caller's ray arrays are hipHostRegister'ed for the duration of a query (see "host memory never meets the GPU")
    else if (k == "host_in_place") d->hostInPlace = atoi(v.c_str()) != 0;                             // rtcIntersect1M / rtcOccluded1M on large host arrays: trace them in place over the host link
    else if (k == "top_split_rel") d->build.top_split_rel = (float)atof(v.c_str());
    else if (k == "top_split_cell") d->build.top_split_cell = (float)atof(v.c_str());
    else if (k == "trav_cost") d->build.trav_cost = (float)atof(v.c_str());
    // CPU-only keys of the reference (threads, isa, tri_accel, hugepages, ...) are accepted and ignored
  }
}

static std::atomic<bool> g_anyFilterEver{false};            // some geometry was given a filter callback at some point: host queries then look at their scene's geometries
// user-geometry callbacks cannot exist here (no user geometries); filter callbacks are HOST functions: the host-array entry points run them between
// launches (filtered_query), the device-pointer entry points cannot
void check_query_args(const RTCFilterFunctionN filter, const void* callback, bool hostEntry = false) {
  if (callback) THROW(RTC_ERROR_INVALID_OPERATION, "user-geometry callbacks are not supported (no user geometries on the GPU path)");
  if (filter && !hostEntry) THROW(RTC_ERROR_INVALID_OPERATION, "a filter callback is a host function: use the host-array entry points (rtcIntersect1/4/8/16/1M), not the device-pointer ones "
                                                               "(or create the device with device_filter_functions=1 and pass the address of a __device__ function)");
}
mi355_bvh_t committed_bvh(Scene* s, size_t k = 0) {
  if (!s->committed || !s->reps[k]->bvh) THROW(RTC_ERROR_INVALID_OPERATION, "scene not committed");   // missing_rtcCommit, scene.cpp:66
  return s->reps[k]->bvh;
}

// the kernels' safety nets (iteration cap, stack bound) drop work rather than hang; a blocking query that ran into one reports it instead of returning a wrong answer
static void check_trace_status(mi355_bvh_t b, hipStream_t q) {
  uint32_t flags = 0;
  core_check(mi355_trace_status(b, q, &flags), "trace status");
  if (flags & MI355_TRACE_ITER_CAP_HIT) THROW(RTC_ERROR_UNKNOWN, "traversal stopped at its iteration cap: results are incomplete");
  if (flags & MI355_TRACE_STACK_OVERFLOW) THROW(RTC_ERROR_UNKNOWN, "traversal stack overflow: results are incomplete");
}
static int trace_launch(const Scene* s, mi355_bvh_t b, void* d, unsigned n, size_t stride, bool any, unsigned qflags, hipStream_t q) {
  if ((qflags & MI355_QUERY_COHERENT) && !s->device->coherentMemory) qflags |= MI355_QUERY_COHERENT_NO_MEMORY;   // rtcNewDevice("coherent_memory=0")
  return mi355_trace_query(b, d, n, stride, any ? 1 : 0, qflags, q);
}

// Large host arrays: the caller's array is pinned for the duration of the call and cut into chunks; ONE stream uploads them one after the other, ONE
// downloads them, the traversal of chunk k runs on one of two compute streams between "chunk k is up" and "chunk k may go down" (an event each).  Both
// directions of the link are busy all the time that way (tests/gpu_pcie.py: upload + download streams alone 2.4 ms for 96 MB each way; chunks that go round
// k streams, each doing its own H2D - kernel - D2H: 3.3 ms and up, the copy engines then serve one direction at a time for long stretches).
// Returns false when the array cannot be pinned (the plain path takes over).  `pinned`: the caller has pinned the whole array already (sharded queries).
// The default since round 6: the same pipeline -- upload stream, two compute streams, download stream, an event between each -- but what the link reads and writes are
// four pinned buffers of this replica (two chunks on the way up, two on the way down); the calling thread fills and drains them with the CPU (copy_pool: several threads
// for a 12 MB chunk) while the GPU works on the chunks in between.  The caller's array is never registered and never seen by the GPU.
// (round 6, last session) What crosses the link is what the kernels read and write, not the records: 48 bytes per ray on the way up (the RTCRay part, packed by the CPU while
// it fills the pinned buffer, put back at `stride` by mi355_unpack_rays), and on the way down the fields a query writes -- 32 bytes per ray of a closest-hit query (48 with
// instances), 4 of an occlusion query (mi355_pack_hits / _inst / _occluded) -- which the CPU writes into the caller's records of the rays that HIT; a miss leaves the caller's
// record alone, as the reference does.  192 -> 80 MB over the link for 2^20 closest-hit rays, 96 -> 52 for occlusion rays, and half the bytes through the CPU's caches.
struct PackCtx { char* packed; char* recs; size_t stride; };
static void pack_rays_range(void* c, size_t b, size_t e) {           // caller's records -> 48 packed bytes each
  const PackCtx& x = *(const PackCtx*)c;
  static const bool stream = !(getenv("MI355_PACK_NT") && atoi(getenv("MI355_PACK_NT")) == 0);
  if (stream && !((uintptr_t)x.packed & 15u)) {
    // streaming stores: the pinned buffer is read next by the copy engine, not by this core -- written around the caches its lines are not fetched first (48 of the
    // 192 bytes per ray the packing moved through memory) and do not evict the caller's records
    for (size_t i = b; i < e; i++) {
      const __m128i* src = (const __m128i*)(x.recs + i * x.stride); __m128i* dst = (__m128i*)(x.packed + i * 48u);
      const __m128i a0 = _mm_loadu_si128(src), a1 = _mm_loadu_si128(src + 1), a2 = _mm_loadu_si128(src + 2);
      _mm_stream_si128(dst, a0); _mm_stream_si128(dst + 1, a1); _mm_stream_si128(dst + 2, a2);
    }
    _mm_sfence();                                              // (in order before whatever tells the copy engine to start)
    return;
  }
  for (size_t i = b; i < e; i++) memcpy(x.packed + i * 48u, x.recs + i * x.stride, 48u);
}
static void scatter_hits_range(void* c, size_t b, size_t e) {        // { tfar, u, v, primID | geomID, Ng } of the rays that hit -> the caller's RTCRayHit
  const PackCtx& x = *(const PackCtx*)c;
  for (size_t i = b; i < e; i++) {
    const uint32_t* p = (const uint32_t*)(x.packed + i * 32u);
    if (p[4] == RTC_INVALID_GEOMETRY_ID) continue;
    uint32_t* r = (uint32_t*)(x.recs + i * x.stride);
    r[8] = p[0]; r[12] = p[5]; r[13] = p[6]; r[14] = p[7]; r[15] = p[1]; r[16] = p[2]; r[17] = p[3]; r[18] = p[4]; r[19] = RTC_INVALID_GEOMETRY_ID; r[20] = RTC_INVALID_GEOMETRY_ID;   // (no instances: instID[0] / instPrimID[0] as the kernel writes them)
  }
}
static void scatter_hits_inst_range(void* c, size_t b, size_t e) {   // ... + { instID[0], instPrimID[0] }
  const PackCtx& x = *(const PackCtx*)c;
  for (size_t i = b; i < e; i++) {
    const uint32_t* p = (const uint32_t*)(x.packed + i * 48u);
    if (p[4] == RTC_INVALID_GEOMETRY_ID) continue;
    uint32_t* r = (uint32_t*)(x.recs + i * x.stride);
    r[8] = p[0]; r[12] = p[5]; r[13] = p[6]; r[14] = p[7]; r[15] = p[1]; r[16] = p[2]; r[17] = p[3]; r[18] = p[4]; r[19] = p[8]; r[20] = p[9];
  }
}
static void scatter_occluded_range(void* c, size_t b, size_t e) {    // tfar = -inf of the rays that are occluded -> the caller's RTCRay
  const PackCtx& x = *(const PackCtx*)c;
  for (size_t i = b; i < e; i++) { const uint32_t t = ((const uint32_t*)x.packed)[i]; if (t == 0xFF800000u) ((uint32_t*)(x.recs + i * x.stride))[8] = t; }
}
static void staged_query_packed(Scene* s, Replica& r, char* data, char* d, unsigned M, size_t stride, bool any, unsigned qflags) {
  std::lock_guard<std::mutex> pipeLock(r.pipeMtx);
  mi355_bvh_t b = r.bvh;
  const bool inst = r.bvh != r.flat;                            // (a scene with instances reports instID[0] / instPrimID[0] as well)
  const size_t down = any ? 4u : (inst ? 48u : 32u);
  const unsigned chunk = M < s->device->pipelineMin ? M : s->device->pipelineChunk, nchunks = (M + chunk - 1u) / chunk;
  const size_t upBytes = (size_t)chunk * 48u, downBytes = (size_t)chunk * down;
  hipStream_t up, dn, comp[2];
  std::vector<hipEvent_t> ev;
  { std::lock_guard<std::mutex> lk(r.mtx);
    for (int k = 0; k < Replica::PIPE; k++) if (!r.pipe[k]) hip_check(hipStreamCreateWithFlags(&r.pipe[k], hipStreamNonBlocking), "hipStreamCreate");
    while (r.pipeEvents.size() < 2u * (size_t)nchunks) { hipEvent_t e; hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"); r.pipeEvents.push_back(e); }
    if (r.pinCap < upBytes) {                                    // (the pinned buffers of the whole-record path serve: 48 bytes per ray is what its occlusion queries need at least)
      for (int k = 0; k < 2; k++) { if (r.pinUp[k]) hipHostFree(r.pinUp[k]); if (r.pinDown[k]) hipHostFree(r.pinDown[k]); r.pinUp[k] = r.pinDown[k] = nullptr; }
      r.pinCap = 0;
      const size_t cap = upBytes + upBytes / 4 + 4096;
      for (int k = 0; k < 2; k++) { void* h = nullptr; hip_check(hipHostMalloc(&h, cap, hipHostMallocPortable), "hipHostMalloc(ray staging)"); r.pinUp[k] = (char*)h;
                                    h = nullptr; hip_check(hipHostMalloc(&h, cap, hipHostMallocPortable), "hipHostMalloc(ray staging)"); r.pinDown[k] = (char*)h; }
      r.pinCap = cap;
    }
    if (r.packCap < upBytes) {
      hip_check(hipSetDevice(r.gpu), "hipSetDevice");
      for (int k = 0; k < 2; k++) { if (r.packUp[k]) hipFree(r.packUp[k]); if (r.packDown[k]) hipFree(r.packDown[k]); r.packUp[k] = r.packDown[k] = nullptr; }
      r.packCap = 0;
      const size_t cap = upBytes + upBytes / 4 + 4096;
      for (int k = 0; k < 2; k++) { core_check(mi355_malloc_retry(r.gpu, cap, (void**)&r.packUp[k]), "hipMalloc(packed rays)"); core_check(mi355_malloc_retry(r.gpu, cap, (void**)&r.packDown[k]), "hipMalloc(packed results)"); }
      r.packCap = cap;
    }
    for (int k = 0; k < 2; k++) { if (!r.pinUpEv[k]) hip_check(hipEventCreateWithFlags(&r.pinUpEv[k], hipEventDisableTiming), "hipEventCreate");
                                  if (!r.pinDownEv[k]) hip_check(hipEventCreateWithFlags(&r.pinDownEv[k], hipEventDisableTiming), "hipEventCreate");
                                  if (!r.packUpEv[k]) hip_check(hipEventCreateWithFlags(&r.packUpEv[k], hipEventDisableTiming), "hipEventCreate"); }
    up = r.pipe[0]; dn = r.pipe[1]; comp[0] = r.pipe[2]; comp[1] = r.pipe[3]; ev = r.pipeEvents; }
  (void)downBytes;
  auto span = [&](unsigned c, size_t& ofs, unsigned& n) { const unsigned first = c * chunk; n = M - first < chunk ? M - first : chunk; ofs = (size_t)first * stride; };
  auto drain = [&](unsigned c) {                               // chunk c's results have come down into their pinned buffer: into the caller's records of the rays that hit
    size_t ofs; unsigned n; span(c, ofs, n);
    hip_check(hipEventSynchronize(r.pinDownEv[c & 1u]), "hipEventSynchronize");
    PackCtx x{r.pinDown[c & 1u], data + ofs, stride};
    copy_pool().run(any ? scatter_occluded_range : (inst ? scatter_hits_inst_range : scatter_hits_range), &x, n, down + 44u);
  };
  for (unsigned c = 0; c < nchunks; c++) {
    size_t ofs; unsigned n; span(c, ofs, n);
    const unsigned k = c & 1u; hipStream_t q = comp[k];
    if (c >= 2) hip_check(hipEventSynchronize(r.pinUpEv[k]), "hipEventSynchronize");     // (the upload of chunk c - 2 has left this buffer)
    { PackCtx x{r.pinUp[k], data + ofs, stride}; copy_pool().run(pack_rays_range, &x, n, 96u); }
    if (c >= 2) hip_check(hipStreamWaitEvent(up, r.packUpEv[k], 0), "hipStreamWaitEvent");   // (chunk c - 2's packed rays have been put into their records: packUp[k] may be written again)
    hip_check(hipMemcpyAsync(r.packUp[k], r.pinUp[k], (size_t)n * 48u, hipMemcpyHostToDevice, up), "hipMemcpyAsync(rays H2D)");
    hip_check(hipEventRecord(r.pinUpEv[k], up), "hipEventRecord");
    hip_check(hipEventRecord(ev[2u * c], up), "hipEventRecord"); hip_check(hipStreamWaitEvent(q, ev[2u * c], 0), "hipStreamWaitEvent");
    core_check(mi355_unpack_rays(r.packUp[k], n, d + ofs, stride, any ? 0 : 1, q), "unpack rays");
    hip_check(hipEventRecord(r.packUpEv[k], q), "hipEventRecord");
    core_check(trace_launch(s, b, d + ofs, n, stride, any, qflags, q), "trace");
    if (c >= 2) { hip_check(hipStreamWaitEvent(q, r.pinDownEv[k], 0), "hipStreamWaitEvent"); }   // (chunk c - 2's results have left packDown[k])
    core_check(any ? mi355_pack_occluded(d + ofs, n, stride, r.packDown[k], q) : (inst ? mi355_pack_hits_inst(d + ofs, n, stride, r.packDown[k], q) : mi355_pack_hits(d + ofs, n, stride, r.packDown[k], q)), "pack results");
    hip_check(hipEventRecord(ev[2u * c + 1u], q), "hipEventRecord"); hip_check(hipStreamWaitEvent(dn, ev[2u * c + 1u], 0), "hipStreamWaitEvent");
    if (c >= 2) drain(c - 2);                                   // (its pinned buffer is the one chunk c comes down into)
    hip_check(hipMemcpyAsync(r.pinDown[k], r.packDown[k], (size_t)n * down, hipMemcpyDeviceToHost, dn), "hipMemcpyAsync(results D2H)");
    hip_check(hipEventRecord(r.pinDownEv[k], dn), "hipEventRecord");
  }
  if (nchunks >= 2) drain(nchunks - 2);
  drain(nchunks - 1);
  check_trace_status(b, comp[0]); if (nchunks > 1) check_trace_status(b, comp[1]);
}

static void staged_query(Scene* s, Replica& r, char* data, char* d, unsigned M, size_t stride, bool any, unsigned qflags) {
  static const bool envPacked = !(getenv("MI355_PACKED_LINK") && atoi(getenv("MI355_PACKED_LINK")) == 0);   // (A/B without touching the device config)
  if (envPacked && s->device->packedLink && M >= 1024u && !(stride & 15u)) { staged_query_packed(s, r, data, d, M, stride, any, qflags); return; }
  std::lock_guard<std::mutex> pipeLock(r.pipeMtx);
  mi355_bvh_t b = r.bvh;
  const size_t rec = any ? 48 : 96;
  const unsigned chunk = M < s->device->pipelineMin ? M : s->device->pipelineChunk, nchunks = (M + chunk - 1u) / chunk;
  const size_t chunkBytes = (size_t)(chunk - 1) * stride + rec;
  hipStream_t up, down, comp[2];
  std::vector<hipEvent_t> ev;
  { std::lock_guard<std::mutex> lk(r.mtx);
    for (int k = 0; k < Replica::PIPE; k++) if (!r.pipe[k]) hip_check(hipStreamCreateWithFlags(&r.pipe[k], hipStreamNonBlocking), "hipStreamCreate");
    while (r.pipeEvents.size() < 2u * (size_t)nchunks) { hipEvent_t e; hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"); r.pipeEvents.push_back(e); }
    if (r.pinCap < chunkBytes) {
      for (int k = 0; k < 2; k++) { if (r.pinUp[k]) hipHostFree(r.pinUp[k]); if (r.pinDown[k]) hipHostFree(r.pinDown[k]); r.pinUp[k] = r.pinDown[k] = nullptr; }
      r.pinCap = 0;
      const size_t cap = chunkBytes + chunkBytes / 4 + 4096;
      for (int k = 0; k < 2; k++) { void* h = nullptr; hip_check(hipHostMalloc(&h, cap, hipHostMallocPortable), "hipHostMalloc(ray staging)"); r.pinUp[k] = (char*)h;
                                    h = nullptr; hip_check(hipHostMalloc(&h, cap, hipHostMallocPortable), "hipHostMalloc(ray staging)"); r.pinDown[k] = (char*)h; }
      r.pinCap = cap;
    }
    for (int k = 0; k < 2; k++) { if (!r.pinUpEv[k]) hip_check(hipEventCreateWithFlags(&r.pinUpEv[k], hipEventDisableTiming), "hipEventCreate");
                                  if (!r.pinDownEv[k]) hip_check(hipEventCreateWithFlags(&r.pinDownEv[k], hipEventDisableTiming), "hipEventCreate"); }
    up = r.pipe[0]; down = r.pipe[1]; comp[0] = r.pipe[2]; comp[1] = r.pipe[3]; ev = r.pipeEvents; }
  auto span = [&](unsigned c, size_t& ofs, size_t& nb, unsigned& n) { const unsigned first = c * chunk; n = M - first < chunk ? M - first : chunk; ofs = (size_t)first * stride; nb = (size_t)(n - 1) * stride + rec; };
  auto drain = [&](unsigned c) {                               // chunk c has come down into its pinned buffer: hand it to the caller's array
    size_t ofs, nb; unsigned n; span(c, ofs, nb, n);
    hip_check(hipEventSynchronize(r.pinDownEv[c & 1u]), "hipEventSynchronize");
    copy_pool().copy(data + ofs, r.pinDown[c & 1u], nb);
  };
  for (unsigned c = 0; c < nchunks; c++) {
    size_t ofs, nb; unsigned n; span(c, ofs, nb, n);
    const unsigned k = c & 1u; hipStream_t q = comp[k];
    if (c >= 2) hip_check(hipEventSynchronize(r.pinUpEv[k]), "hipEventSynchronize");     // (the upload of chunk c - 2 has left this buffer)
    copy_pool().copy(r.pinUp[k], data + ofs, nb);
    hip_check(hipMemcpyAsync(d + ofs, r.pinUp[k], nb, hipMemcpyHostToDevice, up), "hipMemcpyAsync(rays H2D)");
    hip_check(hipEventRecord(r.pinUpEv[k], up), "hipEventRecord");
    hip_check(hipEventRecord(ev[2u * c], up), "hipEventRecord"); hip_check(hipStreamWaitEvent(q, ev[2u * c], 0), "hipStreamWaitEvent");
    core_check(trace_launch(s, b, d + ofs, n, stride, any, qflags, q), "trace");
    hip_check(hipEventRecord(ev[2u * c + 1u], q), "hipEventRecord"); hip_check(hipStreamWaitEvent(down, ev[2u * c + 1u], 0), "hipStreamWaitEvent");
    if (c >= 2) drain(c - 2);                                   // (its pinned buffer is the one chunk c comes down into)
    hip_check(hipMemcpyAsync(r.pinDown[k], d + ofs, nb, hipMemcpyDeviceToHost, down), "hipMemcpyAsync(rays D2H)");
    hip_check(hipEventRecord(r.pinDownEv[k], down), "hipEventRecord");
  }
  if (nchunks >= 2) drain(nchunks - 2);
  drain(nchunks - 1);
  check_trace_status(b, comp[0]); if (nchunks > 1) check_trace_status(b, comp[1]);
}

static bool pipelined_query(Scene* s, Replica& r, char* data, char* d, unsigned M, size_t stride, bool any, unsigned qflags, size_t bytes, bool pinned) {
  if (!pinned && hipHostRegister(data, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
  struct Unpin { void* p; ~Unpin() { if (p) hipHostUnregister(p); } } unpin{pinned ? nullptr : data};
  std::lock_guard<std::mutex> pipeLock(r.pipeMtx);          // (another thread's pipelined query would wait on my events and could read my status words)
  mi355_bvh_t b = r.bvh;
  const size_t rec = any ? 48 : 96;
  const unsigned chunk = s->device->pipelineChunk, nchunks = (M + chunk - 1u) / chunk;
  hipStream_t up, down, comp[2];
  std::vector<hipEvent_t> ev;
  { std::lock_guard<std::mutex> lk(r.mtx);
    for (int k = 0; k < Replica::PIPE; k++) if (!r.pipe[k]) hip_check(hipStreamCreateWithFlags(&r.pipe[k], hipStreamNonBlocking), "hipStreamCreate");
    while (r.pipeEvents.size() < 2u * (size_t)nchunks) { hipEvent_t e; hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"); r.pipeEvents.push_back(e); }
    up = r.pipe[0]; down = r.pipe[1]; comp[0] = r.pipe[2]; comp[1] = r.pipe[3]; ev = r.pipeEvents; }
  unsigned c = 0;
  for (unsigned first = 0; first < M; first += chunk, c++) {
    const unsigned n = M - first < chunk ? M - first : chunk;
    const size_t ofs = (size_t)first * stride, nb = (size_t)(n - 1) * stride + rec;
    hipStream_t q = comp[c & 1u];
    hip_check(hipMemcpyAsync(d + ofs, data + ofs, nb, hipMemcpyHostToDevice, up), "hipMemcpyAsync(rays H2D)");
    hip_check(hipEventRecord(ev[2u * c], up), "hipEventRecord"); hip_check(hipStreamWaitEvent(q, ev[2u * c], 0), "hipStreamWaitEvent");
    core_check(trace_launch(s, b, d + ofs, n, stride, any, qflags, q), "trace");
    hip_check(hipEventRecord(ev[2u * c + 1u], q), "hipEventRecord"); hip_check(hipStreamWaitEvent(down, ev[2u * c + 1u], 0), "hipStreamWaitEvent");
    hip_check(hipMemcpyAsync(data + ofs, d + ofs, nb, hipMemcpyDeviceToHost, down), "hipMemcpyAsync(rays D2H)");
  }
  hip_check(hipStreamSynchronize(down), "hipStreamSynchronize");   // (everything else was waited for by the downloads)
  check_trace_status(b, comp[0]); if (c > 1) check_trace_status(b, comp[1]);
  return true;
}

// bytes a buffer view really touches: offset + stride * (n - 1) + element size (TriangleMesh::setBuffer, scene_triangle_mesh.cpp:35-80; the reference reads
// exactly that plus its documented 16-byte vertex padding) -- NOT n * stride: with an interleaved layout the last element ends before the last stride does
static size_t format_bytes(RTCFormat fmt) {
  if (fmt >= RTC_FORMAT_FLOAT && fmt <= RTC_FORMAT_FLOAT16) return 4u * (size_t)(fmt - RTC_FORMAT_FLOAT + 1);
  if (fmt >= RTC_FORMAT_UINT && fmt <= RTC_FORMAT_UINT4) return 4u * (size_t)(fmt - RTC_FORMAT_UINT + 1);
  return 16;
}
static size_t view_bytes(RTCFormat fmt, size_t stride, size_t n) { return n ? (n - 1) * stride + format_bytes(fmt) : 0; }

// ---- filter callbacks (rtcSetGeometryIntersectFilterFunction / OccludedFilterFunction, RTCIntersectArguments::filter).  The reference calls them inside the
// traversal for every potential hit (runIntersectionFilter1 / runOcclusionFilter1, kernels/geometry/filter.h:14-80; Intersect1EpilogM, intersector_epilog.h:
// 235-300) and goes on when one says no.  A host function cannot run in a HIP kernel, so here the traversal finds the CLOSEST candidate, the host asks the
// callbacks (geometry filter first, then the argument filter if the geometry enabled it or RTC_RAY_QUERY_FLAG_INVOKE_ARGUMENT_FILTER is set; N = 1, ray.tfar =
// the candidate's distance while they run), and a ray whose candidate was rejected is traced again from just behind it.  The closest ACCEPTED hit -- what
// the reference returns -- comes out the same; differences: a callback sees candidates in distance order and only the closest ones (the reference: in
// traversal order, possibly farther ones first), a second triangle at exactly a rejected distance is skipped with it, and occlusion queries with filters
// cost a closest-hit search per round.  Scenes with instances: the filter is the one of the instanced scene's geometry (instID[0] names the instance).
static bool scene_has_filters(Scene* s, const RTCFilterFunctionN argFilter, unsigned flags, bool any, bool top = true) {
  for (auto& kv : s->geoms) {
    Geometry* g = kv.second;
    if (any ? g->occludedFilter != nullptr : g->intersectFilter != nullptr) return true;
    if (argFilter && g->type != RTC_GEOMETRY_TYPE_INSTANCE && (g->argFilter || (flags & RTC_RAY_QUERY_FLAG_INVOKE_ARGUMENT_FILTER))) return true;
    // the geometries of an instanced scene carry their own filters (one level, like the traversal)
    if (top && g->type == RTC_GEOMETRY_TYPE_INSTANCE && g->enabled && g->object && g->object != s && scene_has_filters(g->object, argFilter, flags, any, false)) return true;
  }
  return false;
}
static void plain_query(Scene* s, void* data, unsigned M, size_t stride, bool any, unsigned qflags);
static void filtered_query(Scene* s, void* data, unsigned M, size_t stride, bool any, RTCFilterFunctionN argFilter, unsigned qflags, RTCRayQueryContext* uctx) {
  RTCRayQueryContext defctx; rtcInitRayQueryContext(&defctx);
  RTCRayQueryContext* ctx = uctx ? uctx : &defctx;
  std::vector<RTCRayHit> work(M);                            // the rays still searching, as closest-hit records
  std::vector<unsigned> who(M);                              // ... and whose they are
  for (unsigned i = 0; i < M; i++) {
    const char* src = (const char*)data + (size_t)i * stride;
    memcpy(&work[i].ray, src, sizeof(RTCRay)); memset(&work[i].hit, 0, sizeof(RTCHit));
    work[i].hit.geomID = RTC_INVALID_GEOMETRY_ID; work[i].hit.primID = RTC_INVALID_GEOMETRY_ID; work[i].hit.instID[0] = RTC_INVALID_GEOMETRY_ID;
    who[i] = i;
  }
  unsigned n = M;
  for (unsigned round = 0; n != 0u; round++) {
    if (round >= 4096u) THROW(RTC_ERROR_UNKNOWN, "filter callbacks rejected 4096 candidates in a row along one ray");
    plain_query(s, work.data(), n, sizeof(RTCRayHit), false, 0u);
    unsigned m = 0;
    for (unsigned k = 0; k < n; k++) {
      RTCRayHit& w = work[k];
      if (w.hit.geomID == RTC_INVALID_GEOMETRY_ID) continue;  // nothing (left) on this ray: the caller's record stays as it is
      char* dst = (char*)data + (size_t)who[k] * stride;
      // whose filter: a hit inside an instance names the instance in instID[0] and the geometry of the INSTANCED scene in geomID (instance_intersector.cpp:26-60);
      // the callback sees the instance in context->instID[0] as well, like the reference's traversal has it while inside the instance
      Scene* owner = s;
      if (w.hit.instID[0] != RTC_INVALID_GEOMETRY_ID) {
        auto ii = s->geoms.find(w.hit.instID[0]);
        owner = (ii != s->geoms.end() && ii->second->type == RTC_GEOMETRY_TYPE_INSTANCE) ? ii->second->object : nullptr;
      }
      Geometry* g = nullptr;
      if (owner) { auto it = owner->geoms.find(w.hit.geomID); if (it != owner->geoms.end()) g = it->second; }
      const unsigned ctxInst = ctx->instID[0], ctxPrim = ctx->instPrimID[0];
      ctx->instID[0] = w.hit.instID[0]; ctx->instPrimID[0] = w.hit.instPrimID[0];
      struct Restore { RTCRayQueryContext* c; unsigned a, b; ~Restore() { c->instID[0] = a; c->instPrimID[0] = b; } } restore{ctx, ctxInst, ctxPrim};
      int valid = -1;
      RTCRayHit cand = w;                                     // the callbacks may change the hit and shorten tfar
      RTCFilterFunctionNArguments fa; fa.valid = &valid; fa.geometryUserPtr = g ? g->userPtr : nullptr; fa.context = ctx;
      fa.ray = (RTCRayN*)&cand.ray; fa.hit = (RTCHitN*)&cand.hit; fa.N = 1;
      const RTCFilterFunctionN gf = g ? (any ? g->occludedFilter : g->intersectFilter) : nullptr;
      if (gf) gf(&fa);
      if (valid != 0 && argFilter && g && (g->argFilter || (qflags & RTC_RAY_QUERY_FLAG_INVOKE_ARGUMENT_FILTER))) argFilter(&fa);
      if (valid != 0) {                                       // accepted
        if (any) ((RTCRay*)dst)->tfar = -INFINITY;
        else { ((RTCRayHit*)dst)->ray.tfar = cand.ray.tfar; ((RTCRayHit*)dst)->hit = cand.hit; }
        continue;
      }
      // rejected: search on behind it, up to the ray's own tfar
      const float t = w.ray.tfar;
      RTCRayHit nx; memcpy(&nx.ray, (const char*)data + (size_t)who[k] * stride, sizeof(RTCRay));
      // fast scenes: Moeller-Trumbore is strict at tnear, so restarting AT t leaves the rejected hit out; robust scenes: the Pluecker test is inclusive
      // at both ends (triangle_intersector_pluecker.h:104), the same triangle would come back and the callback would see it twice: restart one ulp behind
      const bool robust = (s->flags & RTC_SCENE_FLAG_ROBUST) != 0;
      nx.ray.tnear = t > nx.ray.tnear ? (robust ? nextafterf(t, INFINITY) : t) : nextafterf(nx.ray.tnear, INFINITY);
      if (w.ray.tnear >= nx.ray.tnear) nx.ray.tnear = nextafterf(w.ray.tnear, INFINITY);
      memset(&nx.hit, 0, sizeof(RTCHit)); nx.hit.geomID = RTC_INVALID_GEOMETRY_ID; nx.hit.primID = RTC_INVALID_GEOMETRY_ID; nx.hit.instID[0] = RTC_INVALID_GEOMETRY_ID;
      work[m] = nx; who[m] = who[k]; m++;
    }
    n = m;
  }
}

// host-pointer AoS query on ONE replica: upload, trace, download
static void replica_query(Scene* s, size_t k, char* data, unsigned M, size_t stride, bool any, unsigned qflags, bool pinned) {
  Replica& r = *s->reps[k];
  mi355_bvh_t b = committed_bvh(s, k);
  hip_check(hipSetDevice(r.gpu), "hipSetDevice");
  const size_t rec = any ? 48 : 96;
  const size_t bytes = (size_t)(M - 1) * stride + rec;
  if (bytes <= Replica::SMALL_BYTES && s->device->smallInPlace) {
    char* hd = nullptr; char* h = r.stage_host(&hd);
    memcpy(h, data, bytes);
    const hipStream_t sq = s->device->pollSmall ? r.small_stream() : nullptr;
    core_check(trace_launch(s, b, hd, M, stride, any, qflags, sq), "trace");
    // (round 5) The launch of a single ray takes ~25 us; hipStreamSynchronize puts the thread to sleep on the completion signal and is woken by an interrupt, which adds
    // 10 - 25 us to every call (36 us minimum, 51 us median on the driver's box in round 4).  A thread that waits for one ray polls instead -- hipStreamQuery reads the
    // signal without sleeping -- for at most ~200 us, after which it falls back to the blocking wait (a long kernel ahead of it in the stream, a preempted process).
    if (s->device->pollSmall) {
      const auto t0 = std::chrono::steady_clock::now();
      while (hipStreamQuery(sq) == hipErrorNotReady) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
      }
      (void)hipGetLastError();                                 // (hipErrorNotReady is not an error)
    }
    uint32_t flags = 0;
    core_check(mi355_trace_status(b, sq, &flags), "trace status");   // (waits for the launch: the status words are read after it)
    memcpy(data, h, bytes);
    if (flags & MI355_TRACE_ITER_CAP_HIT) THROW(RTC_ERROR_UNKNOWN, "traversal stopped at its iteration cap: results are incomplete");
    if (flags & MI355_TRACE_STACK_OVERFLOW) THROW(RTC_ERROR_UNKNOWN, "traversal stack overflow: results are incomplete");
    return;
  }
  // "host_in_place=1" (or MI355_HOST_IN_PLACE=1): large host arrays are traced where they lie -- the caller's array is registered with the device (pinned + mapped) and the
  // kernel reads the 48 bytes of every ray and writes the <= 52 bytes of every hit over the host link itself, instead of 96 bytes each way through a staging
  // copy.  Off by default until measured against the chunked pipeline below on the round's box (bench.py end_to_end; tests/gpu_hostpath.py).
  static const bool envInPlace = getenv("MI355_HOST_IN_PLACE") && atoi(getenv("MI355_HOST_IN_PLACE")) != 0;
  if ((envInPlace || s->device->hostInPlace) && M >= s->device->pipelineMin) {   // (host_in_place registers the caller's array by definition)
    const bool reg = pinned || hipHostRegister(data, bytes, hipHostRegisterMapped) == hipSuccess;
    if (!reg) (void)hipGetLastError();
    void* dp = nullptr;
    if (reg && hipHostGetDevicePointer(&dp, data, 0) == hipSuccess && dp) {
      struct Unpin { void* p; ~Unpin() { if (p) hipHostUnregister(p); } } unpin{pinned ? nullptr : data};
      std::lock_guard<std::mutex> pipeLock(r.pipeMtx);
      core_check(trace_launch(s, b, dp, M, stride, any, qflags, nullptr), "trace");
      check_trace_status(b, nullptr);                        // (waits for the launch)
      return;
    }
    (void)hipGetLastError();
    if (reg && !pinned) hipHostUnregister(data);
  }
  char* d = r.stage(bytes);
  if (!s->device->hostRegister) { staged_query(s, r, data, d, M, stride, any, qflags); return; }
  if (M >= s->device->pipelineMin && pipelined_query(s, r, data, d, M, stride, any, qflags, bytes, pinned)) return;
  hip_check(hipMemcpy(d, data, bytes, hipMemcpyHostToDevice), "hipMemcpy(rays H2D)");
  core_check(trace_launch(s, b, d, M, stride, any, qflags, nullptr), "trace");
  hip_check(hipMemcpy(data, d, bytes, hipMemcpyDeviceToHost), "hipMemcpy(rays D2H)");
  check_trace_status(b, nullptr);
}
// contiguous ray range of replica k out of n ([k M / n, (k + 1) M / n); embree_amd/shard.py shard_range is the same formula)
static inline unsigned shard_begin(unsigned M, size_t k, size_t n) { return (unsigned)(((unsigned long long)M * k) / n); }
// host-pointer AoS query: on one GPU, or -- device over N GPUs, batch large enough -- the ray range split contiguously over the replicas, one host thread per GPU,
// every shard uploaded to / traced on / downloaded from its own GPU straight into the caller's array (no collective: the consumer is the host)
static void plain_query(Scene* s, void* data, unsigned M, size_t stride, bool any, unsigned qflags = 0) {
  const size_t n = s->reps.size();
  if (n == 1 || M < s->device->shardMin * n) { replica_query(s, 0, (char*)data, M, stride, any, qflags, false); return; }
  const size_t rec = any ? 48 : 96, bytes = (size_t)(M - 1) * stride + rec;
  const bool pinned = s->device->hostRegister && hipHostRegister(data, bytes, hipHostRegisterPortable) == hipSuccess;    // "host_register=1": once, for all GPUs (portable: every device's copies may use it)
  if (!pinned) (void)hipGetLastError();
  struct Unpin { void* p; ~Unpin() { if (p) hipHostUnregister(p); } } unpin{pinned ? data : nullptr};
  for_each_replica(n, [&](size_t k) {
    const unsigned lo = shard_begin(M, k, n), hi = shard_begin(M, k + 1, n);
    if (hi > lo) replica_query(s, k, (char*)data + (size_t)lo * stride, hi - lo, stride, any, qflags, pinned);
  });
  hip_check(hipSetDevice(s->device->gpu), "hipSetDevice");
}
// device-array query on a device over N GPUs: the array lives on the first GPU; shards 1 .. N-1 travel peer to peer (xGMI) to their replicas' staging
// areas and back, each on its replica's own stream, fenced against the caller's stream by events: asynchronous like the single-GPU form.
static void sharded_device_query(Scene* s, char* d, unsigned M, size_t stride, bool any, unsigned qflags, hipStream_t stream) {
  const size_t n = s->reps.size(), rec = any ? 48 : 96;
  Replica& r0 = *s->reps[0];
  std::lock_guard<std::mutex> enqueueLock(r0.pipeMtx);      // the fence events are per replica: one thread enqueues a sharded query at a time
  hip_check(hipSetDevice(r0.gpu), "hipSetDevice");
  { std::lock_guard<std::mutex> lk(r0.mtx); if (!r0.shardIn) hip_check(hipEventCreateWithFlags(&r0.shardIn, hipEventDisableTiming), "hipEventCreate"); }
  hip_check(hipEventRecord(r0.shardIn, stream), "hipEventRecord");          // the rays are ready on the caller's stream from here on
  for (size_t k = 1; k < n; k++) {
    Replica& r = *s->reps[k];
    const unsigned lo = shard_begin(M, k, n), hi = shard_begin(M, k + 1, n);
    if (hi <= lo) continue;
    const size_t nb = (size_t)(hi - lo - 1) * stride + rec;
    hip_check(hipSetDevice(r.gpu), "hipSetDevice");
    { std::lock_guard<std::mutex> lk(r.mtx);
      if (!r.shardStream) hip_check(hipStreamCreateWithFlags(&r.shardStream, hipStreamNonBlocking), "hipStreamCreate");
      if (!r.shardOut) hip_check(hipEventCreateWithFlags(&r.shardOut, hipEventDisableTiming), "hipEventCreate"); }
    char* st = r.stage(nb);
    hip_check(hipStreamWaitEvent(r.shardStream, r0.shardIn, 0), "hipStreamWaitEvent");
    hip_check(hipMemcpyPeerAsync(st, r.gpu, d + (size_t)lo * stride, r0.gpu, nb, r.shardStream), "hipMemcpyPeerAsync(shard out)");
    core_check(trace_launch(s, committed_bvh(s, k), st, hi - lo, stride, any, qflags, r.shardStream), "trace");
    hip_check(hipMemcpyPeerAsync(d + (size_t)lo * stride, r0.gpu, st, r.gpu, nb, r.shardStream), "hipMemcpyPeerAsync(shard back)");
    hip_check(hipEventRecord(r.shardOut, r.shardStream), "hipEventRecord");
  }
  hip_check(hipSetDevice(r0.gpu), "hipSetDevice");
  const unsigned hi0 = shard_begin(M, 1, n);
  if (hi0) core_check(trace_launch(s, committed_bvh(s, 0), d, hi0, stride, any, qflags, stream), "trace");
  for (size_t k = 1; k < n; k++) if (s->reps[k]->shardOut && shard_begin(M, k + 1, n) > shard_begin(M, k, n)) hip_check(hipStreamWaitEvent(stream, s->reps[k]->shardOut, 0), "hipStreamWaitEvent");
}
// device arrays that already live where they are traced: shard k on replica k's GPU, on the caller's stream of that GPU; nothing travels
static void sharded_pointer_query(Scene* s, unsigned numShards, void* const* d, const unsigned* counts, size_t stride, bool any, unsigned qflags, void* const* streams) {
  if (!d || !counts) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid argument");
  if (numShards > s->reps.size()) THROW(RTC_ERROR_INVALID_ARGUMENT, "more shards than GPUs behind this device (RTC_DEVICE_PROPERTY_GPU_COUNT)");
  committed_bvh(s);
  for (unsigned k = 0; k < numShards; k++) {
    if (counts[k] == 0u) continue;
    if (!d[k]) THROW(RTC_ERROR_INVALID_ARGUMENT, "shard pointer is NULL");
    hip_check(hipSetDevice(s->reps[k]->gpu), "hipSetDevice");
    core_check(trace_launch(s, committed_bvh(s, k), d[k], counts[k], stride, any, qflags, streams ? (hipStream_t)streams[k] : nullptr), "trace");
  }
  hip_check(hipSetDevice(s->device->gpu), "hipSetDevice");
}
// filterFn: the address of a __device__ filter function (device_filter_functions=1), 0 = none; the reference's GPU path calls the pointer it finds in the arguments the same
// way (filter_sycl.h:31-43, under RTC_FEATURE_FLAG_FILTER_FUNCTION_IN_ARGUMENTS)
static void device_query(Scene* s, void* d, unsigned M, size_t stride, bool any, unsigned qflags, void* stream, uint64_t filterFn = 0, void* filterCtx = nullptr) {
  if (M == 0) { committed_bvh(s); return; }
  const size_t n = s->reps.size();
  if (filterFn) {
    if (n > 1) THROW(RTC_ERROR_INVALID_OPERATION, "device filter functions need a device over ONE GPU (a function address belongs to one GPU's code object)");
    core_check(mi355_trace_query_filtered(committed_bvh(s), d, M, stride, any ? 1 : 0, qflags, filterFn, filterCtx, (hipStream_t)stream), "trace");
    return;
  }
  if (n > 1 && M >= s->device->shardMin * n) { sharded_device_query(s, (char*)d, M, stride, any, qflags, (hipStream_t)stream); return; }
  core_check(trace_launch(s, committed_bvh(s), d, M, stride, any, qflags, (hipStream_t)stream), "trace");
}
void host_query(Scene* s, void* data, unsigned M, size_t stride, bool any, RTCFilterFunctionN argFilter = nullptr, unsigned qflags = 0, RTCRayQueryContext* uctx = nullptr) {
  if (M == 0) return;
  committed_bvh(s);
  const size_t rec = any ? 48 : 96;
  if (stride < rec) THROW(RTC_ERROR_INVALID_ARGUMENT, "byteStride smaller than the ray record");
  hip_check(hipSetDevice(s->device->gpu), "hipSetDevice");
  const bool repack = (stride & 15) || ((size_t)data & 15);
  if (repack) THROW(RTC_ERROR_INVALID_ARGUMENT, "ray records must be 16-byte aligned (include/embree4/rtcore.h)");
  if ((g_anyFilterEver.load(std::memory_order_relaxed) || argFilter) && scene_has_filters(s, argFilter, qflags, any)) { filtered_query(s, data, M, stride, any, argFilter, qflags, uctx); return; }
  plain_query(s, data, M, stride, any, qflags);
}
// rtcIntersect4/8/16, rtcOccluded4/8/16 on a host packet: the K lanes are turned into AoS records on the host (RayHitK::get / set, kernels/common/ray.h:283-376),
// the active ones go through the batch path as one launch, and only active lanes are written back (InactiveRaysTest, verify.cpp:3553).
void host_packet_query(Scene* s, const int* valid, void* packet, unsigned K, bool any, RTCFilterFunctionN argFilter = nullptr, unsigned qflags = 0, RTCRayQueryContext* uctx = nullptr) {
  if (!valid || !packet) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid argument");
  const unsigned nf = any ? 12u : 21u, recw = any ? 12u : 24u;
  alignas(16) uint32_t aos[16 * 24];
  unsigned lane[16], n = 0;
  const uint32_t* pk = (const uint32_t*)packet;
  for (unsigned k = 0; k < K; k++) {
    if (valid[k] != -1) continue;
    uint32_t* r = aos + (size_t)n * recw;
    for (unsigned f = 0; f < nf; f++) r[f] = pk[f * K + k];
    for (unsigned f = nf; f < recw; f++) r[f] = 0u;
    lane[n++] = k;
  }
  if (n == 0) { committed_bvh(s); return; }
  host_query(s, aos, n, recw * 4, any, argFilter, qflags, uctx);
  uint32_t* out = (uint32_t*)packet;
  for (unsigned j = 0; j < n; j++) {
    const uint32_t* r = aos + (size_t)j * recw; const unsigned k = lane[j];
    out[8 * K + k] = r[8];                                   // tfar
    if (!any && r[12 + 6] != RTC_INVALID_GEOMETRY_ID) for (unsigned f = 12; f < 21; f++) out[f * K + k] = r[f];
  }
}

}  // namespace

// ============================================================================================ device
RTC_API RTCDevice rtcNewDevice(const char* config) {
  Device* d = nullptr;
  CATCH_BEGIN
  d = new Device;
  mi355_default_build_params(&d->build);
  parse_config(d, config);
  const int n = mi355_device_count();
  if (n <= 0) THROW(RTC_ERROR_UNSUPPORTED_CPU, "no HIP device found: this library has no CPU fallback");
  if (d->gpu < 0 || d->gpu >= n) THROW(RTC_ERROR_INVALID_ARGUMENT, "gpu ordinal out of range");
  if (d->wantGpus < 1 || d->wantGpus > 64) THROW(RTC_ERROR_INVALID_ARGUMENT, "gpus= out of range");
  if (d->wantGpus > n && !d->oversubscribe) THROW(RTC_ERROR_INVALID_ARGUMENT, "gpus= asks for more GPUs than this node has (gpu_oversubscribe=1 lets replicas share a GPU)");
  d->gpus.clear();
  for (int k = 0; k < d->wantGpus; k++) d->gpus.push_back((d->gpu + k) % n);                          // replica k lives on GPU (gpu + k) mod #GPUs
  for (size_t k = 1; k < d->gpus.size(); k++) {                                                        // shards of device-array queries travel peer to peer
    if (d->gpus[k] == d->gpu) continue;
    int can = 0; if (hipDeviceCanAccessPeer(&can, d->gpu, d->gpus[k]) == hipSuccess && can) { hipSetDevice(d->gpu); if (hipDeviceEnablePeerAccess(d->gpus[k], 0) != hipSuccess) (void)hipGetLastError(); }
    can = 0; if (hipDeviceCanAccessPeer(&can, d->gpus[k], d->gpu) == hipSuccess && can) { hipSetDevice(d->gpus[k]); if (hipDeviceEnablePeerAccess(d->gpu, 0) != hipSuccess) (void)hipGetLastError(); }
  }
  hipSetDevice(d->gpu);
  char nm[256]; if (mi355_device_name(d->gpu, nm, sizeof(nm)) == 0) d->name = nm;
  d->countIn();
  if (d->verbose >= 1) printf("Embree(MI355X) %s on %s\n", RTC_VERSION_STRING, d->name.c_str());
  return (RTCDevice)d;
  CATCH_END(nullptr)
  if (d) d->release();
  return nullptr;
}
RTC_API void rtcRetainDevice(RTCDevice h) { CATCH_BEGIN dev_of(h)->retain(); CATCH_END((Device*)h) }
RTC_API void rtcReleaseDevice(RTCDevice h) { CATCH_BEGIN dev_of(h)->release(); CATCH_END(nullptr) }
RTC_API ssize_t rtcGetDeviceProperty(RTCDevice h, enum RTCDeviceProperty prop) {
  CATCH_BEGIN
  dev_of(h);
  if ((int)prop >= (int)RTC_DEVICE_PROPERTY_GPU_OF_REPLICA_0 && (size_t)((int)prop - (int)RTC_DEVICE_PROPERTY_GPU_OF_REPLICA_0) < ((Device*)h)->gpus.size())
    return (ssize_t)((Device*)h)->gpus[(size_t)((int)prop - (int)RTC_DEVICE_PROPERTY_GPU_OF_REPLICA_0)];
  switch (prop) {
    case RTC_DEVICE_PROPERTY_VERSION: return RTC_VERSION;
    case RTC_DEVICE_PROPERTY_VERSION_MAJOR: return RTC_VERSION_MAJOR;
    case RTC_DEVICE_PROPERTY_VERSION_MINOR: return RTC_VERSION_MINOR;
    case RTC_DEVICE_PROPERTY_VERSION_PATCH: return RTC_VERSION_PATCH;
    case RTC_DEVICE_PROPERTY_NATIVE_RAY4_SUPPORTED: case RTC_DEVICE_PROPERTY_NATIVE_RAY8_SUPPORTED:
    case RTC_DEVICE_PROPERTY_NATIVE_RAY16_SUPPORTED: return 1;
    case RTC_DEVICE_PROPERTY_RAY_MASK_SUPPORTED: return 1;
    case RTC_DEVICE_PROPERTY_TRIANGLE_GEOMETRY_SUPPORTED: case RTC_DEVICE_PROPERTY_QUAD_GEOMETRY_SUPPORTED: return 1;
    case RTC_DEVICE_PROPERTY_HIP_DEVICE: return 1;
    case RTC_DEVICE_PROPERTY_GPU_COUNT: return (ssize_t)((Device*)h)->gpus.size();
    // What is BUILT answers 1, like the reference's default configuration does for the same feature (kernels/common/device.cpp:480-600):
    //  * filter functions: rtcSetGeometryIntersect/OccludedFilterFunction and the argument filter run for every entry point (host callbacks between launches, rules
    //    and device functions inside the kernels) -- EMBREE_FILTER_FUNCTION, device.cpp:515.  tutorials/verify registers its intersection_filter group on this answer.
    //  * rtcJoinCommitScene: Scene::commit holds the scene's lock and an unmodified scene returns at once, so threads that join find the tree built when their call
    //    returns (device.cpp:566-572: 1 with the internal tasking system).
    case RTC_DEVICE_PROPERTY_FILTER_FUNCTION_SUPPORTED: case RTC_DEVICE_PROPERTY_JOIN_COMMIT_SUPPORTED: return 1;
    // What is NOT built, or is off in the reference's default configuration too (backface culling, compact polys, ignore-invalid-rays: CMake options that default to OFF),
    // answers 0; TASKING_SYSTEM 0 = "internal" (the GPU is the worker pool); commits of one device are serialised by the build arena: no parallel commit.
    case RTC_DEVICE_PROPERTY_BACKFACE_CULLING_ENABLED: case RTC_DEVICE_PROPERTY_BACKFACE_CULLING_CURVES_ENABLED:
    case RTC_DEVICE_PROPERTY_BACKFACE_CULLING_SPHERES_ENABLED:
    case RTC_DEVICE_PROPERTY_IGNORE_INVALID_RAYS_ENABLED: case RTC_DEVICE_PROPERTY_COMPACT_POLYS_ENABLED:
    case RTC_DEVICE_PROPERTY_SUBDIVISION_GEOMETRY_SUPPORTED:
    case RTC_DEVICE_PROPERTY_CURVE_GEOMETRY_SUPPORTED: case RTC_DEVICE_PROPERTY_USER_GEOMETRY_SUPPORTED:
    case RTC_DEVICE_PROPERTY_POINT_GEOMETRY_SUPPORTED: case RTC_DEVICE_PROPERTY_TASKING_SYSTEM:
    case RTC_DEVICE_PROPERTY_PARALLEL_COMMIT_SUPPORTED:
    case RTC_DEVICE_PROPERTY_CPU_DEVICE: case RTC_DEVICE_PROPERTY_SYCL_DEVICE: return 0;
    default: THROW(RTC_ERROR_INVALID_ARGUMENT, "unknown readable property");
  }
  CATCH_END((Device*)h)
  return 0;
}
RTC_API void rtcSetDeviceProperty(RTCDevice h, const enum RTCDeviceProperty, ssize_t) {
  CATCH_BEGIN dev_of(h); THROW(RTC_ERROR_INVALID_ARGUMENT, "unknown writable property"); CATCH_END((Device*)h)
}
RTC_API const char* rtcGetErrorString(enum RTCError e) {
  switch (e) {
    case RTC_ERROR_NONE: return "No error";
    case RTC_ERROR_UNKNOWN: return "Unknown error";
    case RTC_ERROR_INVALID_ARGUMENT: return "Invalid argument";
    case RTC_ERROR_INVALID_OPERATION: return "Invalid operation";
    case RTC_ERROR_OUT_OF_MEMORY: return "Out of memory";
    case RTC_ERROR_UNSUPPORTED_CPU: return "Unsupported CPU";
    case RTC_ERROR_CANCELLED: return "Cancelled";
    case RTC_ERROR_LEVEL_ZERO_RAYTRACING_SUPPORT_MISSING: return "Level Zero raytracing support missing";
    default: return "Invalid error code";
  }
}
RTC_API enum RTCError rtcGetDeviceError(RTCDevice h) {        // returns and clears (device.cpp:273-279)
  if (!h) { RTCError e = g_threadError.error; g_threadError.error = RTC_ERROR_NONE; return e; }
  ErrState& e = ((Device*)h)->err(); RTCError c = e.error; e.error = RTC_ERROR_NONE; return c;
}
RTC_API const char* rtcGetDeviceLastErrorMessage(RTCDevice h) {
  if (!h) return g_threadError.msg.c_str();
  return ((Device*)h)->err().msg.c_str();
}
RTC_API void rtcSetDeviceErrorFunction(RTCDevice h, RTCErrorFunction fn, void* p) { CATCH_BEGIN Device* d = dev_of(h); d->errorFn = fn; d->errorFnPtr = p; CATCH_END((Device*)h) }
RTC_API void rtcSetDeviceMemoryMonitorFunction(RTCDevice h, RTCMemoryMonitorFunction fn, void* p) { CATCH_BEGIN Device* d = dev_of(h); d->memFn = fn; d->memFnPtr = p; CATCH_END((Device*)h) }

// ============================================================================================ buffer
RTC_API RTCBuffer rtcNewBuffer(RTCDevice h, size_t bytes) {
  CATCH_BEGIN return (RTCBuffer) new Buffer(dev_of(h), bytes, nullptr); CATCH_END((Device*)h) return nullptr;
}
RTC_API RTCBuffer rtcNewSharedBuffer(RTCDevice h, void* ptr, size_t bytes) {
  CATCH_BEGIN if (!ptr && bytes) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid argument"); return (RTCBuffer) new Buffer(dev_of(h), bytes, ptr); CATCH_END((Device*)h) return nullptr;
}
RTC_API void* rtcGetBufferData(RTCBuffer b) { if (!b) return nullptr; return ((Buffer*)b)->host; }
RTC_API void rtcRetainBuffer(RTCBuffer b) { if (b) ((Buffer*)b)->retain(); }
RTC_API void rtcReleaseBuffer(RTCBuffer b) { if (b) ((Buffer*)b)->release(); }

// ========================================================================================== geometry
RTC_API RTCGeometry rtcNewGeometry(RTCDevice h, enum RTCGeometryType type) {
  CATCH_BEGIN
  Device* d = dev_of(h);
  if (type != RTC_GEOMETRY_TYPE_TRIANGLE && type != RTC_GEOMETRY_TYPE_QUAD && type != RTC_GEOMETRY_TYPE_INSTANCE)
    THROW(RTC_ERROR_INVALID_OPERATION, "only RTC_GEOMETRY_TYPE_TRIANGLE, RTC_GEOMETRY_TYPE_QUAD and RTC_GEOMETRY_TYPE_INSTANCE are supported by the MI355X core");
  return (RTCGeometry) new Geometry(d, type);
  CATCH_END((Device*)h)
  return nullptr;
}
#define GEOM_DEV(h) ((h) ? ((Geometry*)(h))->device : nullptr)
RTC_API void rtcRetainGeometry(RTCGeometry h) { CATCH_BEGIN geom_of(h)->retain(); CATCH_END(GEOM_DEV(h)) }
RTC_API void rtcReleaseGeometry(RTCGeometry h) { CATCH_BEGIN geom_of(h)->release(); CATCH_END(nullptr) }
RTC_API void rtcCommitGeometry(RTCGeometry h) {
  CATCH_BEGIN
  Geometry* g = geom_of(h);
  if (g->vertices.buf) g->vertices.buf->upload();
  if (g->indices.buf) g->indices.buf->upload();
  g->committed = true;
  CATCH_END(GEOM_DEV(h))
}
RTC_API void rtcEnableGeometry(RTCGeometry h) { CATCH_BEGIN geom_of(h)->enabled = true; CATCH_END(GEOM_DEV(h)) }
RTC_API void rtcDisableGeometry(RTCGeometry h) { CATCH_BEGIN geom_of(h)->enabled = false; CATCH_END(GEOM_DEV(h)) }
RTC_API void rtcSetGeometryTimeStepCount(RTCGeometry h, unsigned n) {
  CATCH_BEGIN geom_of(h); if (n != 1) THROW(RTC_ERROR_INVALID_OPERATION, "motion blur is not supported by the MI355X core"); CATCH_END(GEOM_DEV(h))
}
RTC_API void rtcSetGeometryVertexAttributeCount(RTCGeometry h, unsigned) { CATCH_BEGIN geom_of(h); CATCH_END(GEOM_DEV(h)) }
RTC_API void rtcSetGeometryMask(RTCGeometry h, unsigned mask) { CATCH_BEGIN Geometry* g = geom_of(h); g->mask = mask; g->committed = false; g->dataCounter++; CATCH_END(GEOM_DEV(h)) }
RTC_API void rtcSetGeometryBuildQuality(RTCGeometry h, enum RTCBuildQuality q) {
  CATCH_BEGIN Geometry* g = geom_of(h);
  if (q != RTC_BUILD_QUALITY_LOW && q != RTC_BUILD_QUALITY_MEDIUM && q != RTC_BUILD_QUALITY_HIGH && q != RTC_BUILD_QUALITY_REFIT) THROW(RTC_ERROR_INVALID_OPERATION, "invalid build quality");
  g->quality = q;                                           // REFIT: a commit after a vertex update refits the tree instead of rebuilding it (Scene::commit)
  CATCH_END(GEOM_DEV(h))
}
RTC_API void rtcSetGeometryBuffer(RTCGeometry h, enum RTCBufferType type, unsigned slot, enum RTCFormat fmt, RTCBuffer buffer,
                                  size_t byteOffset, size_t byteStride, size_t itemCount) {
  CATCH_BEGIN
  Geometry* g = geom_of(h); if (!buffer) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid argument");
  Buffer* b = (Buffer*)buffer;
  if (g->device != b->device) THROW(RTC_ERROR_INVALID_ARGUMENT, "inputs are from different devices");
  g->setBuffer(type, slot, fmt, b, byteOffset, byteStride, itemCount);
  CATCH_END(GEOM_DEV(h))
}
RTC_API void rtcSetSharedGeometryBuffer(RTCGeometry h, enum RTCBufferType type, unsigned slot, enum RTCFormat fmt, const void* ptr,
                                        size_t byteOffset, size_t byteStride, size_t itemCount) {
  CATCH_BEGIN
  Geometry* g = geom_of(h);
  Buffer* b = new Buffer(g->device, view_bytes(fmt, byteStride, itemCount), (char*)ptr + byteOffset);
  try { g->setBuffer(type, slot, fmt, b, 0, byteStride, itemCount); } catch (...) { b->release(); throw; }
  b->release();
  CATCH_END(GEOM_DEV(h))
}
// device-resident geometry: the reference's own entry point for GPU devices (rtcore.cpp, rtcSetSharedGeometryBufferHostDevice):
// dptr is HIP device memory on the device's GPU; no upload happens in rtcCommitGeometry.
RTC_API void rtcSetSharedGeometryBufferHostDevice(RTCGeometry h, enum RTCBufferType type, unsigned slot, enum RTCFormat fmt,
                                                  const void* ptr, const void* dptr, size_t byteOffset, size_t byteStride, size_t itemCount) {
  CATCH_BEGIN
  Geometry* g = geom_of(h);
  if (!dptr) THROW(RTC_ERROR_INVALID_ARGUMENT, "device pointer may not be NULL");
  Buffer* b = new Buffer(g->device, view_bytes(fmt, byteStride, itemCount), ptr ? (char*)ptr + byteOffset : (char*)16, (char*)dptr + byteOffset);
  try { g->setBuffer(type, slot, fmt, b, 0, byteStride, itemCount); } catch (...) { b->release(); throw; }
  b->release();
  CATCH_END(GEOM_DEV(h))
}
RTC_API void* rtcSetNewGeometryBuffer(RTCGeometry h, enum RTCBufferType type, unsigned slot, enum RTCFormat fmt, size_t byteStride, size_t itemCount) {
  CATCH_BEGIN
  Geometry* g = geom_of(h);
  size_t bytes = itemCount * byteStride;
  if (type == RTC_BUFFER_TYPE_VERTEX || type == RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE) bytes += (16 - (byteStride % 16)) % 16;   // rtcore.cpp:1932-1935
  Buffer* b = new Buffer(g->device, bytes, nullptr);
  try { g->setBuffer(type, slot, fmt, b, 0, byteStride, itemCount); } catch (...) { b->release(); throw; }
  void* p = b->host; b->release();
  return p;
  CATCH_END(GEOM_DEV(h))
  return nullptr;
}
RTC_API void* rtcGetGeometryBufferData(RTCGeometry h, enum RTCBufferType type, unsigned slot) {
  CATCH_BEGIN
  Geometry* g = geom_of(h); if (slot != 0) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid buffer slot");
  BufferView* v = type == RTC_BUFFER_TYPE_VERTEX ? &g->vertices : type == RTC_BUFFER_TYPE_INDEX ? &g->indices : nullptr;
  if (!v) THROW(RTC_ERROR_INVALID_ARGUMENT, "unknown buffer type");
  return v->buf ? v->buf->host + v->offset : nullptr;
  CATCH_END(GEOM_DEV(h))
  return nullptr;
}
RTC_API void rtcUpdateGeometryBuffer(RTCGeometry h, enum RTCBufferType type, unsigned slot) {
  CATCH_BEGIN
  Geometry* g = geom_of(h); if (slot != 0) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid buffer slot");
  BufferView* v = type == RTC_BUFFER_TYPE_VERTEX ? &g->vertices : type == RTC_BUFFER_TYPE_INDEX ? &g->indices : nullptr;
  if (!v) THROW(RTC_ERROR_INVALID_ARGUMENT, "unknown buffer type");
  if (v->buf && v->buf->ownsDev) v->buf->devDirty = true;
  if (v->buf && !v->buf->dev) v->buf->devDirty = true;
  g->modified = true; g->committed = false;
  if (type == RTC_BUFFER_TYPE_VERTEX) g->dataCounter++; else g->topoCounter++;
  CATCH_END(GEOM_DEV(h))
}
// ---- RTC_GEOMETRY_TYPE_INSTANCE (kernels/common/scene_instance.cpp; rtcore.cpp:1408-1495 for the matrix formats)
RTC_API void rtcSetGeometryInstancedScene(RTCGeometry h, RTCScene sc) {
  CATCH_BEGIN Geometry* g = geom_of(h);
  if (g->type != RTC_GEOMETRY_TYPE_INSTANCE) THROW(RTC_ERROR_INVALID_OPERATION, "operation not supported for this geometry");
  Scene* s = scene_of(sc);
  if (s->device != g->device) THROW(RTC_ERROR_INVALID_ARGUMENT, "inputs are from different devices");
  s->retain(); if (g->object) g->object->release();
  g->object = s; g->committed = false; g->topoCounter++;
  CATCH_END(GEOM_DEV(h))
}
RTC_API void rtcSetGeometryTransform(RTCGeometry h, unsigned timeStep, enum RTCFormat fmt, const void* xfm) {
  CATCH_BEGIN Geometry* g = geom_of(h);
  if (g->type != RTC_GEOMETRY_TYPE_INSTANCE) THROW(RTC_ERROR_INVALID_OPERATION, "operation not supported for this geometry");
  if (!xfm) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid argument");
  if (timeStep != 0) THROW(RTC_ERROR_INVALID_OPERATION, "motion blur is not supported by the MI355X core");
  const float* x = (const float*)xfm; float* m = g->l2w;       // vx, vy, vz, p
  if (fmt == RTC_FORMAT_FLOAT3X4_ROW_MAJOR) { for (int c = 0; c < 4; c++) for (int r = 0; r < 3; r++) m[3 * c + r] = x[4 * r + c]; }
  else if (fmt == RTC_FORMAT_FLOAT3X4_COLUMN_MAJOR) memcpy(m, x, 48);
  else if (fmt == RTC_FORMAT_FLOAT4X4_COLUMN_MAJOR) { for (int c = 0; c < 4; c++) for (int r = 0; r < 3; r++) m[3 * c + r] = x[4 * c + r]; }
  else THROW(RTC_ERROR_INVALID_OPERATION, "invalid matrix format");
  g->committed = false; g->dataCounter++;
  CATCH_END(GEOM_DEV(h))
}
RTC_API void rtcGetGeometryTransform(RTCGeometry h, float, enum RTCFormat fmt, void* out) {
  CATCH_BEGIN Geometry* g = geom_of(h);
  if (g->type != RTC_GEOMETRY_TYPE_INSTANCE) THROW(RTC_ERROR_INVALID_OPERATION, "operation not supported for this geometry");
  if (!out) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid argument");
  const float* m = g->l2w; float* x = (float*)out;
  if (fmt == RTC_FORMAT_FLOAT3X4_ROW_MAJOR) { for (int c = 0; c < 4; c++) for (int r = 0; r < 3; r++) x[4 * r + c] = m[3 * c + r]; }
  else if (fmt == RTC_FORMAT_FLOAT3X4_COLUMN_MAJOR) memcpy(x, m, 48);
  else if (fmt == RTC_FORMAT_FLOAT4X4_COLUMN_MAJOR) { for (int c = 0; c < 4; c++) { for (int r = 0; r < 3; r++) x[4 * c + r] = m[3 * c + r]; x[4 * c + 3] = c == 3 ? 1.0f : 0.0f; } }
  else THROW(RTC_ERROR_INVALID_OPERATION, "invalid matrix format");
  CATCH_END(GEOM_DEV(h))
}
//  With device filter functions the user pointer of a geometry that enabled the argument filter travels to the GPU in the rule table (words 10 / 11 of its entry,
// written at commit): a changed pointer counts as a changed rule, so that the next rtcCommitScene uploads it instead of returning early on an "unmodified" scene.  (The reference
// applies setUserData at once, geometry.cpp:137; the host filter path here reads g->userPtr live.  For a __device__ filter function: rtcCommitScene after rtcSetGeometryUserData.)
RTC_API void rtcSetGeometryUserData(RTCGeometry h, void* p) {
  CATCH_BEGIN Geometry* g = geom_of(h);
  if (g->userPtr != p && g->device->deviceFilters && g->argFilter) g->ruleCounter++;
  g->userPtr = p;
  CATCH_END(GEOM_DEV(h))
}
RTC_API void* rtcGetGeometryUserData(RTCGeometry h) { CATCH_BEGIN return geom_of(h)->userPtr; CATCH_END(GEOM_DEV(h)) return nullptr; }
// Filter callbacks are host functions; the host-array entry points run them between launches (filtered_query above).  A geometry tells the scenes it is
// attached to lazily: the flag below is looked at by every host query.
RTC_API void rtcSetGeometryIntersectFilterFunction(RTCGeometry h, RTCFilterFunctionN f) { CATCH_BEGIN geom_of(h)->intersectFilter = f; if (f) g_anyFilterEver = true; CATCH_END(GEOM_DEV(h)) }
RTC_API void rtcSetGeometryOccludedFilterFunction(RTCGeometry h, RTCFilterFunctionN f) { CATCH_BEGIN geom_of(h)->occludedFilter = f; if (f) g_anyFilterEver = true; CATCH_END(GEOM_DEV(h)) }
RTC_API void rtcSetGeometryFilterRule(RTCGeometry h, const struct RTCFilterRule* rule) {
  CATCH_BEGIN Geometry* g = geom_of(h);
  if (rule) {
    if (rule->primFactor > 0xFFFFu || rule->geomFactor > 0xFFFFu) THROW(RTC_ERROR_INVALID_ARGUMENT, "filter rule factors must be below 65536");
    if ((rule->kinds & RTC_FILTER_RULE_PRIMITIVE_BITS) && rule->numBits && !rule->bits) THROW(RTC_ERROR_INVALID_ARGUMENT, "filter rule without its bit array");
    g->rule = *rule; g->hasRule = true;
    g->ruleBits.clear();
    if ((rule->kinds & RTC_FILTER_RULE_PRIMITIVE_BITS) && rule->numBits) g->ruleBits.assign(rule->bits, rule->bits + (rule->numBits + 31u) / 32u);   // copied: the caller's array may go
    g->rule.bits = nullptr;
  } else { g->hasRule = false; g->rule = RTCFilterRule{}; g->ruleBits.clear(); }
  g->ruleCounter++;
  CATCH_END(GEOM_DEV(h))
}
RTC_API void rtcSetGeometryEnableFilterFunctionFromArguments(RTCGeometry h, bool enable) {
  CATCH_BEGIN Geometry* g = geom_of(h);
  if (g->argFilter != enable && g->device->deviceFilters) g->ruleCounter++;   // (with device filter functions the flag travels to the GPU in the rule table: the next commit uploads it)
  g->argFilter = enable;
  CATCH_END(GEOM_DEV(h))
}

// ============================================================================================= scene
#define SCENE_DEV(h) ((h) ? ((Scene*)(h))->device : nullptr)
RTC_API RTCScene rtcNewScene(RTCDevice h) { CATCH_BEGIN return (RTCScene) new Scene(dev_of(h)); CATCH_END((Device*)h) return nullptr; }
RTC_API RTCDevice rtcGetSceneDevice(RTCScene h) { CATCH_BEGIN Scene* s = scene_of(h); s->device->retain(); return (RTCDevice)s->device; CATCH_END(SCENE_DEV(h)) return nullptr; }
RTC_API void rtcRetainScene(RTCScene h) { CATCH_BEGIN scene_of(h)->retain(); CATCH_END(SCENE_DEV(h)) }
RTC_API void rtcReleaseScene(RTCScene h) { CATCH_BEGIN scene_of(h)->release(); CATCH_END(nullptr) }
RTC_API RTCTraversable rtcGetSceneTraversable(RTCScene h) { return (RTCTraversable)h; }
RTC_API unsigned int rtcAttachGeometry(RTCScene h, RTCGeometry hg) {
  CATCH_BEGIN
  Scene* s = scene_of(h); Geometry* g = geom_of(hg);
  if (s->device != g->device) THROW(RTC_ERROR_INVALID_ARGUMENT, "inputs are from different devices");
  std::lock_guard<std::mutex> lk(s->mtx);
  unsigned id = 0; for (auto& kv : s->geoms) { if (kv.first != id) break; id++; }     // IDPool: lowest free ID (scene.cpp:717-741)
  g->retain(); g->attached++; s->geoms[id] = g; s->modified = true;
  return id;
  CATCH_END(SCENE_DEV(h))
  return RTC_INVALID_GEOMETRY_ID;
}
RTC_API void rtcAttachGeometryByID(RTCScene h, RTCGeometry hg, unsigned id) {
  CATCH_BEGIN
  Scene* s = scene_of(h); Geometry* g = geom_of(hg);
  if (id == RTC_INVALID_GEOMETRY_ID) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid geometry ID");
  if (s->device != g->device) THROW(RTC_ERROR_INVALID_ARGUMENT, "inputs are from different devices");
  std::lock_guard<std::mutex> lk(s->mtx);
  if (s->geoms.count(id)) THROW(RTC_ERROR_INVALID_OPERATION, "trying to bind geometry to already reserved ID");
  g->retain(); g->attached++; s->geoms[id] = g; s->modified = true;
  CATCH_END(SCENE_DEV(h))
}
RTC_API void rtcDetachGeometry(RTCScene h, unsigned id) {
  CATCH_BEGIN
  Scene* s = scene_of(h);
  std::lock_guard<std::mutex> lk(s->mtx);
  auto it = s->geoms.find(id);
  if (it == s->geoms.end()) THROW(RTC_ERROR_INVALID_OPERATION, "invalid geometry ID");
  it->second->attached--; it->second->release(); s->geoms.erase(it); s->modified = true;
  CATCH_END(SCENE_DEV(h))
}
RTC_API RTCGeometry rtcGetGeometry(RTCScene h, unsigned id) {
  CATCH_BEGIN
  Scene* s = scene_of(h); std::lock_guard<std::mutex> lk(s->mtx);
  auto it = s->geoms.find(id); if (it == s->geoms.end()) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid geometry ID");
  return (RTCGeometry)it->second;
  CATCH_END(SCENE_DEV(h))
  return nullptr;
}
RTC_API RTCGeometry rtcGetGeometryThreadSafe(RTCScene h, unsigned id) { return rtcGetGeometry(h, id); }
RTC_API void rtcCommitScene(RTCScene h) { CATCH_BEGIN scene_of(h)->commit(); CATCH_END(SCENE_DEV(h)) }
RTC_API void rtcJoinCommitScene(RTCScene h) { rtcCommitScene(h); }      // the GPU does the work; joining threads have nothing to add
RTC_API void rtcSetSceneProgressMonitorFunction(RTCScene h, RTCProgressMonitorFunction f, void* p) { CATCH_BEGIN Scene* s = scene_of(h); s->progress = f; s->progressPtr = p; CATCH_END(SCENE_DEV(h)) }
RTC_API void rtcSetSceneBuildQuality(RTCScene h, enum RTCBuildQuality q) {
  CATCH_BEGIN Scene* s = scene_of(h);
  if (q != RTC_BUILD_QUALITY_LOW && q != RTC_BUILD_QUALITY_MEDIUM && q != RTC_BUILD_QUALITY_HIGH) THROW(RTC_ERROR_INVALID_OPERATION, "invalid build quality");
  s->quality = q;
  CATCH_END(SCENE_DEV(h))
}
RTC_API void rtcSetSceneFlags(RTCScene h, enum RTCSceneFlags f) { CATCH_BEGIN scene_of(h)->flags = f; CATCH_END(SCENE_DEV(h)) }
RTC_API enum RTCSceneFlags rtcGetSceneFlags(RTCScene h) { CATCH_BEGIN return scene_of(h)->flags; CATCH_END(SCENE_DEV(h)) return RTC_SCENE_FLAG_NONE; }
RTC_API void rtcGetSceneBounds(RTCScene h, struct RTCBounds* o) {
  CATCH_BEGIN Scene* s = scene_of(h); if (!o) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid destination pointer");
  if (!s->committed) THROW(RTC_ERROR_INVALID_OPERATION, "scene not committed");
  *o = s->bounds;
  CATCH_END(SCENE_DEV(h))
}

// ============================================================================================ queries
RTC_API void rtcIntersect1(RTCScene h, struct RTCRayHit* rh, struct RTCIntersectArguments* a) {
  CATCH_BEGIN Scene* s = scene_of(h); if (a) check_query_args(a->filter, (const void*)a->intersect, true); host_query(s, rh, 1, sizeof(RTCRayHit), false, a ? a->filter : nullptr, a ? (unsigned)a->flags : 0u, a ? a->context : nullptr); CATCH_END(SCENE_DEV(h))
}
RTC_API void rtcOccluded1(RTCScene h, struct RTCRay* r, struct RTCOccludedArguments* a) {
  CATCH_BEGIN Scene* s = scene_of(h); if (a) check_query_args(a->filter, (const void*)a->occluded, true); host_query(s, r, 1, sizeof(RTCRay), true, a ? a->filter : nullptr, a ? (unsigned)a->flags : 0u, a ? a->context : nullptr); CATCH_END(SCENE_DEV(h))
}
#define PACKET_ENTRY(K)                                                                                                     \
  RTC_API void rtcIntersect##K(const int* valid, RTCScene h, struct RTCRayHit##K* rh, struct RTCIntersectArguments* a) {    \
    CATCH_BEGIN Scene* s = scene_of(h); if (a) check_query_args(a->filter, (const void*)a->intersect, true);                \
    host_packet_query(s, valid, rh, K, false, a ? a->filter : nullptr, a ? (unsigned)a->flags : 0u, a ? a->context : nullptr); CATCH_END(SCENE_DEV(h)) }                                                    \
  RTC_API void rtcOccluded##K(const int* valid, RTCScene h, struct RTCRay##K* r, struct RTCOccludedArguments* a) {          \
    CATCH_BEGIN Scene* s = scene_of(h); if (a) check_query_args(a->filter, (const void*)a->occluded, true);                 \
    host_packet_query(s, valid, r, K, true, a ? a->filter : nullptr, a ? (unsigned)a->flags : 0u, a ? a->context : nullptr); CATCH_END(SCENE_DEV(h)) }                                                      \
  RTC_API void rtcTraversableIntersect##K(const int* valid, RTCTraversable t, struct RTCRayHit##K* rh, struct RTCIntersectArguments* a) { rtcIntersect##K(valid, (RTCScene)t, rh, a); } \
  RTC_API void rtcTraversableOccluded##K(const int* valid, RTCTraversable t, struct RTCRay##K* r, struct RTCOccludedArguments* a) { rtcOccluded##K(valid, (RTCScene)t, r, a); }
PACKET_ENTRY(4)
PACKET_ENTRY(8)
PACKET_ENTRY(16)
RTC_API void rtcTraversableIntersect1(RTCTraversable t, struct RTCRayHit* rh, struct RTCIntersectArguments* a) { rtcIntersect1((RTCScene)t, rh, a); }
RTC_API void rtcTraversableOccluded1(RTCTraversable t, struct RTCRay* r, struct RTCOccludedArguments* a) { rtcOccluded1((RTCScene)t, r, a); }

RTC_API void rtcIntersect1M(RTCScene h, struct RTCRayHit* rh, unsigned M, size_t stride, struct RTCIntersectArguments* a) {
  CATCH_BEGIN Scene* s = scene_of(h); if (a) check_query_args(a->filter, (const void*)a->intersect, true); host_query(s, rh, M, stride, false, a ? a->filter : nullptr, a ? (unsigned)a->flags : 0u, a ? a->context : nullptr); CATCH_END(SCENE_DEV(h))
}
RTC_API void rtcOccluded1M(RTCScene h, struct RTCRay* r, unsigned M, size_t stride, struct RTCOccludedArguments* a) {
  CATCH_BEGIN Scene* s = scene_of(h); if (a) check_query_args(a->filter, (const void*)a->occluded, true); host_query(s, r, M, stride, true, a ? a->filter : nullptr, a ? (unsigned)a->flags : 0u, a ? a->context : nullptr); CATCH_END(SCENE_DEV(h))
}
RTC_API void rtcIntersect1MDevice(RTCScene h, void* d_rh, unsigned M, size_t stride, struct RTCIntersectArguments* a, void* stream) {
  CATCH_BEGIN Scene* s = scene_of(h);
  const bool fptr = a && a->filter && s->device->deviceFilters && (a->feature_mask & RTC_FEATURE_FLAG_FILTER_FUNCTION_IN_ARGUMENTS);
  if (a) check_query_args(fptr ? nullptr : a->filter, (const void*)a->intersect);
  device_query(s, d_rh, M, stride, false, a ? (unsigned)a->flags : 0u, stream, fptr ? (uint64_t)(uintptr_t)a->filter : 0ull, fptr ? (void*)a->context : nullptr); CATCH_END(SCENE_DEV(h))
}
RTC_API void rtcOccluded1MDevice(RTCScene h, void* d_r, unsigned M, size_t stride, struct RTCOccludedArguments* a, void* stream) {
  CATCH_BEGIN Scene* s = scene_of(h);
  const bool fptr = a && a->filter && s->device->deviceFilters && (a->feature_mask & RTC_FEATURE_FLAG_FILTER_FUNCTION_IN_ARGUMENTS);
  if (a) check_query_args(fptr ? nullptr : a->filter, (const void*)a->occluded);
  device_query(s, d_r, M, stride, true, a ? (unsigned)a->flags : 0u, stream, fptr ? (uint64_t)(uintptr_t)a->filter : 0ull, fptr ? (void*)a->context : nullptr); CATCH_END(SCENE_DEV(h))
}
RTC_API void rtcIntersect1MDeviceSharded(RTCScene h, unsigned n, void* const* d_rh, const unsigned* counts, size_t stride, struct RTCIntersectArguments* a, void* const* streams) {
  CATCH_BEGIN Scene* s = scene_of(h); if (a) check_query_args(a->filter, (const void*)a->intersect);
  sharded_pointer_query(s, n, d_rh, counts, stride, false, a ? (unsigned)a->flags : 0u, streams); CATCH_END(SCENE_DEV(h))
}
RTC_API void rtcOccluded1MDeviceSharded(RTCScene h, unsigned n, void* const* d_r, const unsigned* counts, size_t stride, struct RTCOccludedArguments* a, void* const* streams) {
  CATCH_BEGIN Scene* s = scene_of(h); if (a) check_query_args(a->filter, (const void*)a->occluded);
  sharded_pointer_query(s, n, d_r, counts, stride, true, a ? (unsigned)a->flags : 0u, streams); CATCH_END(SCENE_DEV(h))
}
// extension used by tests/bench: the core BVH handle behind a committed scene (NULL if not committed)
extern "C" __attribute__((visibility("default"))) mi355_bvh_t rtcGetSceneBVH_mi355(RTCScene h) { return h ? ((Scene*)h)->bvh0() : nullptr; }
// ... and of replica k of a device over several GPUs (NULL beyond the last)
extern "C" __attribute__((visibility("default"))) mi355_bvh_t rtcGetSceneReplicaBVH_mi355(RTCScene h, unsigned k) { return h && k < ((Scene*)h)->reps.size() ? ((Scene*)h)->reps[k]->bvh : nullptr; }

// ================================================================= entry points outside the triangle path
// exported so that applications linking the full Embree API resolve; each records RTC_ERROR_INVALID_OPERATION
#define UNSUPPORTED_GEOM(name, ...) RTC_API void name(RTCGeometry h, ##__VA_ARGS__) { process_error(GEOM_DEV(h), RTC_ERROR_INVALID_OPERATION, #name " is not supported by the MI355X triangle core"); }
UNSUPPORTED_GEOM(rtcSetGeometryTimeRange, float, float)
UNSUPPORTED_GEOM(rtcSetGeometryMaxRadiusScale, float)
UNSUPPORTED_GEOM(rtcSetGeometryPointQueryFunction, void*)
UNSUPPORTED_GEOM(rtcSetGeometryUserPrimitiveCount, unsigned)
UNSUPPORTED_GEOM(rtcSetGeometryBoundsFunction, void*, void*)
UNSUPPORTED_GEOM(rtcSetGeometryIntersectFunction, void*)
UNSUPPORTED_GEOM(rtcSetGeometryOccludedFunction, void*)
UNSUPPORTED_GEOM(rtcSetGeometryTessellationRate, float)
UNSUPPORTED_GEOM(rtcSetGeometryTopologyCount, unsigned)
UNSUPPORTED_GEOM(rtcSetGeometrySubdivisionMode, unsigned, int)
UNSUPPORTED_GEOM(rtcSetGeometryVertexAttributeTopology, unsigned, unsigned)
UNSUPPORTED_GEOM(rtcSetGeometryDisplacementFunction, void*)
RTC_API bool rtcPointQuery(RTCScene h, void*, void*, void*, void*) { process_error(SCENE_DEV(h), RTC_ERROR_INVALID_OPERATION, "rtcPointQuery is not supported by the MI355X triangle core"); return false; }
RTC_API void rtcCollide(RTCScene h, RTCScene, void*, void*) { process_error(SCENE_DEV(h), RTC_ERROR_INVALID_OPERATION, "rtcCollide is not supported by the MI355X triangle core"); }
// rtcInterpolate / rtcInterpolateN: host arithmetic on the host copies of the buffers, as in the reference (TriangleMesh::interpolate_impl,
// kernels/common/scene_triangle_mesh.h:49-100: P = madd(w, p0, madd(u, p1, v * p2)), dPdu = p1 - p0, dPdv = p2 - p0, second derivatives 0;
// QuadMesh::interpolate_impl, scene_quad_mesh.h:55-112: the triangle (p0,p1,p3) for u + v <= 1, else (p2,p3,p1) with 1-u, 1-v and flipped derivatives).
static void interpolate_one(Geometry* g, const RTCInterpolateArguments* a) {
  if (g->type != RTC_GEOMETRY_TYPE_TRIANGLE && g->type != RTC_GEOMETRY_TYPE_QUAD) THROW(RTC_ERROR_INVALID_OPERATION, "operation not supported for this geometry");
  const BufferView* src = nullptr;
  if (a->bufferType == RTC_BUFFER_TYPE_VERTEX) { if (a->bufferSlot != 0) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid buffer slot"); src = &g->vertices; }
  else if (a->bufferType == RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE) { auto it = g->attribViews.find(a->bufferSlot); if (it == g->attribViews.end()) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid buffer slot"); src = &it->second; }
  else THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid buffer type");
  if (!src->buf || !g->indices.buf) THROW(RTC_ERROR_INVALID_OPERATION, "geometry has no such buffer");
  if ((size_t)src->buf->host < 4096 || (size_t)g->indices.buf->host < 4096) THROW(RTC_ERROR_INVALID_OPERATION, "rtcInterpolate needs host-visible buffers (this buffer was shared as device memory only)");
  if (a->primID >= g->indices.num) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid primitive");
  const unsigned* idx = (const unsigned*)(g->indices.buf->host + g->indices.offset + (size_t)a->primID * g->indices.stride);
  const char* base = src->buf->host + src->offset;
  const bool quad = g->type == RTC_GEOMETRY_TYPE_QUAD;
  const unsigned nidx = quad ? 4u : 3u;
  for (unsigned k = 0; k < nidx; k++) if (idx[k] >= src->num) THROW(RTC_ERROR_INVALID_ARGUMENT, "vertex index out of range");
  const float *p0 = (const float*)(base + (size_t)idx[0] * src->stride), *p1 = (const float*)(base + (size_t)idx[1] * src->stride), *p2 = (const float*)(base + (size_t)idx[2] * src->stride);
  float u = a->u, v = a->v; bool left = true;
  const float *q0 = p0, *q1 = p1, *q2 = p2;
  if (quad) {
    const float* p3 = (const float*)(base + (size_t)idx[3] * src->stride);
    left = u + v <= 1.0f;
    q0 = left ? p0 : p2; q1 = left ? p1 : p3; q2 = left ? p3 : p1;
    if (!left) { u = 1.0f - u; v = 1.0f - v; }
  }
  const float w = 1.0f - u - v;
  for (unsigned i = 0; i < a->valueCount; i++) {
    if (a->P) a->P[i] = fmaf(w, q0[i], fmaf(u, q1[i], v * q2[i]));
    if (a->dPdu) { a->dPdu[i] = left ? q1[i] - q0[i] : q0[i] - q1[i]; a->dPdv[i] = left ? q2[i] - q0[i] : q0[i] - q2[i]; }
    if (a->ddPdudu) { a->ddPdudu[i] = 0.0f; a->ddPdvdv[i] = 0.0f; a->ddPdudv[i] = 0.0f; }
  }
}
RTC_API void rtcInterpolate(const struct RTCInterpolateArguments* a) {
  Geometry* g = a ? (Geometry*)a->geometry : nullptr;
  CATCH_BEGIN if (!a) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid argument"); interpolate_one(geom_of(a->geometry), a); CATCH_END(g ? g->device : nullptr)
}
RTC_API void rtcInterpolateN(const struct RTCInterpolateNArguments* a) {   // Geometry::interpolateN, kernels/common/geometry.cpp:163-235: outputs are SoA, value j of point i at [j * N + i]
  Geometry* g = a ? (Geometry*)a->geometry : nullptr;
  CATCH_BEGIN
  if (!a) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid argument");
  if (a->valueCount > 256) THROW(RTC_ERROR_INVALID_OPERATION, "maximally 256 floating point values can be interpolated per vertex");
  float P[256], du[256], dv[256], duu[256], dvv[256], duv[256];
  const int* valid = (const int*)a->valid;
  for (unsigned i = 0; i < a->N; i++) {
    if (valid && !valid[i]) continue;
    RTCInterpolateArguments one{a->geometry, a->primIDs[i], a->u[i], a->v[i], a->bufferType, a->bufferSlot, a->P ? P : nullptr, a->dPdu ? du : nullptr, a->dPdu ? dv : nullptr,
                                a->ddPdudu ? duu : nullptr, a->ddPdudu ? dvv : nullptr, a->ddPdudu ? duv : nullptr, a->valueCount};
    interpolate_one(geom_of(a->geometry), &one);
    for (unsigned j = 0; j < a->valueCount; j++) {
      if (a->P) a->P[(size_t)j * a->N + i] = P[j];
      if (a->dPdu) { a->dPdu[(size_t)j * a->N + i] = du[j]; a->dPdv[(size_t)j * a->N + i] = dv[j]; }
      if (a->ddPdudu) { a->ddPdudu[(size_t)j * a->N + i] = duu[j]; a->ddPdvdv[(size_t)j * a->N + i] = dvv[j]; a->ddPdudv[(size_t)j * a->N + i] = duv[j]; }
    }
  }
  CATCH_END(g ? g->device : nullptr)
}
RTC_API void rtcGetSceneLinearBounds(RTCScene h, void*) { process_error(SCENE_DEV(h), RTC_ERROR_INVALID_OPERATION, "rtcGetSceneLinearBounds is not supported by the MI355X triangle core"); }

// ---- the rest of the reference library's export table (oracle/_ref/libembree4.so exports 154 rtc* symbols): an application linked against libembree4
// resolves every one of them here.  The host/device buffer calls of Embree 4.4 map onto this library's device copies; what lies outside the triangle /
// quad / instance path records RTC_ERROR_INVALID_OPERATION like an Embree built without that feature.
RTC_API RTCBuffer rtcNewBufferHostDevice(RTCDevice h, size_t bytes) { return rtcNewBuffer(h, bytes); }
RTC_API RTCBuffer rtcNewSharedBufferHostDevice(RTCDevice h, void* ptr, size_t bytes) { return rtcNewSharedBuffer(h, ptr, bytes); }
RTC_API void rtcCommitBuffer(RTCBuffer b) {                  // host -> device copy (rtcore_buffer.h:51)
  if (!b) return;
  Buffer* buf = (Buffer*)b;
  CATCH_BEGIN if (buf->ownsDev || !buf->dev) buf->devDirty = true; buf->upload(); CATCH_END(buf->device)
}
RTC_API void* rtcGetBufferDataDevice(RTCBuffer b) { if (!b) return nullptr; Buffer* buf = (Buffer*)b; return buf->dev; }
RTC_API void* rtcGetGeometryBufferDataDevice(RTCGeometry h, enum RTCBufferType type, unsigned slot) {
  CATCH_BEGIN
  Geometry* g = geom_of(h); if (slot != 0) THROW(RTC_ERROR_INVALID_ARGUMENT, "invalid buffer slot");
  BufferView* v = type == RTC_BUFFER_TYPE_VERTEX ? &g->vertices : type == RTC_BUFFER_TYPE_INDEX ? &g->indices : nullptr;
  if (!v) THROW(RTC_ERROR_INVALID_ARGUMENT, "unknown buffer type");
  return v->buf && v->buf->dev ? v->buf->dev + v->offset : nullptr;
  CATCH_END(GEOM_DEV(h))
  return nullptr;
}
RTC_API void rtcSetNewGeometryBufferHostDevice(RTCGeometry h, enum RTCBufferType type, unsigned slot, enum RTCFormat fmt, size_t byteStride, size_t itemCount, void** ptr, void** dptr) {
  void* p = rtcSetNewGeometryBuffer(h, type, slot, fmt, byteStride, itemCount);
  if (ptr) *ptr = p;
  if (dptr) {                                               // the device copy exists from now on; rtcCommitGeometry / rtcCommitBuffer fill it
    *dptr = nullptr;
    CATCH_BEGIN
    Geometry* g = geom_of(h);
    BufferView* v = type == RTC_BUFFER_TYPE_VERTEX ? &g->vertices : type == RTC_BUFFER_TYPE_INDEX ? &g->indices : nullptr;
    if (p && v && v->buf) { v->buf->upload(); v->buf->devDirty = true; *dptr = v->buf->dev + v->offset; }
    CATCH_END(GEOM_DEV(h))
  }
}
RTC_API void rtcGetGeometryTransformEx(RTCGeometry h, unsigned, float time, enum RTCFormat fmt, void* xfm) { rtcGetGeometryTransform(h, time, fmt, xfm); }
RTC_API void rtcGetGeometryTransformFromScene(RTCScene h, unsigned geomID, float time, enum RTCFormat fmt, void* xfm) {
  RTCGeometry g = rtcGetGeometry(h, geomID); if (g) rtcGetGeometryTransform(g, time, fmt, xfm);
}
RTC_API void rtcGetGeometryTransformFromTraversable(RTCTraversable t, unsigned geomID, float time, enum RTCFormat fmt, void* xfm) { rtcGetGeometryTransformFromScene((RTCScene)t, geomID, time, fmt, xfm); }
RTC_API void* rtcGetGeometryUserDataFromScene(RTCScene h, unsigned geomID) { RTCGeometry g = rtcGetGeometry(h, geomID); return g ? rtcGetGeometryUserData(g) : nullptr; }
RTC_API void* rtcGetGeometryUserDataFromTraversable(RTCTraversable t, unsigned geomID) { return rtcGetGeometryUserDataFromScene((RTCScene)t, geomID); }
UNSUPPORTED_GEOM(rtcSetGeometryInstancedScenes, RTCScene*, size_t)
UNSUPPORTED_GEOM(rtcSetGeometryTransformQuaternion, unsigned, const void*)
#define UNSUPPORTED_VOID(name) RTC_API void name(void) { process_error(nullptr, RTC_ERROR_INVALID_OPERATION, #name " is not supported by the MI355X triangle core"); }
#define UNSUPPORTED_ZERO(T, name) RTC_API T name(void) { process_error(nullptr, RTC_ERROR_INVALID_OPERATION, #name " is not supported by the MI355X triangle core"); return (T)0; }
// (C symbols carry no signature: the callers' arguments are simply not read)
UNSUPPORTED_VOID(rtcForwardIntersect1) UNSUPPORTED_VOID(rtcForwardIntersect4) UNSUPPORTED_VOID(rtcForwardIntersect8) UNSUPPORTED_VOID(rtcForwardIntersect16)
UNSUPPORTED_VOID(rtcForwardIntersect1Ex) UNSUPPORTED_VOID(rtcForwardIntersect4Ex) UNSUPPORTED_VOID(rtcForwardIntersect8Ex) UNSUPPORTED_VOID(rtcForwardIntersect16Ex)
UNSUPPORTED_VOID(rtcForwardOccluded1) UNSUPPORTED_VOID(rtcForwardOccluded4) UNSUPPORTED_VOID(rtcForwardOccluded8) UNSUPPORTED_VOID(rtcForwardOccluded16)
UNSUPPORTED_VOID(rtcForwardOccluded1Ex) UNSUPPORTED_VOID(rtcForwardOccluded4Ex) UNSUPPORTED_VOID(rtcForwardOccluded8Ex) UNSUPPORTED_VOID(rtcForwardOccluded16Ex)
UNSUPPORTED_VOID(rtcTraversableForwardIntersect1) UNSUPPORTED_VOID(rtcTraversableForwardIntersect4) UNSUPPORTED_VOID(rtcTraversableForwardIntersect8) UNSUPPORTED_VOID(rtcTraversableForwardIntersect16)
UNSUPPORTED_VOID(rtcTraversableForwardIntersect1Ex) UNSUPPORTED_VOID(rtcTraversableForwardIntersect4Ex) UNSUPPORTED_VOID(rtcTraversableForwardIntersect8Ex) UNSUPPORTED_VOID(rtcTraversableForwardIntersect16Ex)
UNSUPPORTED_VOID(rtcTraversableForwardOccluded1) UNSUPPORTED_VOID(rtcTraversableForwardOccluded4) UNSUPPORTED_VOID(rtcTraversableForwardOccluded8) UNSUPPORTED_VOID(rtcTraversableForwardOccluded16)
UNSUPPORTED_VOID(rtcTraversableForwardOccluded1Ex) UNSUPPORTED_VOID(rtcTraversableForwardOccluded4Ex) UNSUPPORTED_VOID(rtcTraversableForwardOccluded8Ex) UNSUPPORTED_VOID(rtcTraversableForwardOccluded16Ex)
UNSUPPORTED_ZERO(bool, rtcPointQuery4) UNSUPPORTED_ZERO(bool, rtcPointQuery8) UNSUPPORTED_ZERO(bool, rtcPointQuery16)
UNSUPPORTED_ZERO(bool, rtcTraversablePointQuery) UNSUPPORTED_ZERO(bool, rtcTraversablePointQuery4) UNSUPPORTED_ZERO(bool, rtcTraversablePointQuery8) UNSUPPORTED_ZERO(bool, rtcTraversablePointQuery16)
UNSUPPORTED_VOID(rtcInvokeIntersectFilterFromGeometry) UNSUPPORTED_VOID(rtcInvokeOccludedFilterFromGeometry)
UNSUPPORTED_ZERO(unsigned, rtcGetGeometryFirstHalfEdge) UNSUPPORTED_ZERO(unsigned, rtcGetGeometryFace) UNSUPPORTED_ZERO(unsigned, rtcGetGeometryNextHalfEdge)
UNSUPPORTED_ZERO(unsigned, rtcGetGeometryPreviousHalfEdge) UNSUPPORTED_ZERO(unsigned, rtcGetGeometryOppositeHalfEdge)
UNSUPPORTED_ZERO(void*, rtcNewBVH) UNSUPPORTED_ZERO(void*, rtcBuildBVH) UNSUPPORTED_ZERO(void*, rtcThreadLocalAlloc) UNSUPPORTED_VOID(rtcMakeStaticBVH) UNSUPPORTED_VOID(rtcRetainBVH) UNSUPPORTED_VOID(rtcReleaseBVH)
