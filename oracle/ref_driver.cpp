// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE (never linked or loaded by the product path).
//
// A small C-ABI shim around the REAL reference library built by oracle/ref.mk
// (oracle/_ref/libembree4.so).  It is compiled against the reference's own
// headers where they lie under /root/reference/include (no copy) and lets the
// Python tests / bench.py drive the reference through ctypes:
//   * as the parity checker for the HIP path (tests/, __graft_entry__.smoke),
//   * as the "reference" CPU baseline (bench.py cpu_baseline leg): worker
//     threads of a PERSISTENT, pinned pool (started before the clock) loop
//     rtcIntersect1 / rtcOccluded1 over contiguous 1024-ray blocks with
//     FTZ|DAZ set, exactly the blocking of the reference's own
//     ParallelIntersectBenchmark (tutorials/verify/verify.cpp:5728-5755) and
//     the MXCSR advice of README.md:10319-10340.
// Built into oracle/_ref/libref_driver.so (git-ignored, travels with gpurun).
#include <embree4/rtcore.h>
#include <xmmintrin.h>
#include <pmmintrin.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sched.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace {
struct RefScene {
  RTCDevice device = nullptr;
  RTCScene scene = nullptr;
};
double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// A persistent worker pool: the clock of a measurement must not contain the creation of 255 threads (a 2^20-ray job lasts ~10 ms).  The reference's own
// ParallelIntersectBenchmark (tutorials/verify/verify.cpp:5728-5755) runs on the already-running threads of its tasking system; this is the same with
// std::thread: workers are started (and pinned, one per hardware thread, FTZ|DAZ set: README.md:10319-10340) BEFORE the clock, sleep on a generation
// counter between jobs, and a job is handed out in 1024-ray blocks through one atomic cursor (verify.cpp:5733).
struct Pool {
  std::vector<std::thread> workers;
  std::vector<int> cpus;                                       // the hardware threads this process may run on
  std::atomic<unsigned> gen{0}, next{0}, done{0}, quit{0};
  std::mutex m, jobs; std::condition_variable cv;
  std::function<void(unsigned, unsigned)> body;
  unsigned M = 0, nblocks = 0, active = 0;
  static const unsigned BLOCK = 1024;
  void work() {
    for (;;) {
      const unsigned b = next.fetch_add(1, std::memory_order_relaxed);
      if (b >= nblocks) break;
      const unsigned lo = b * BLOCK, hi = lo + BLOCK < M ? lo + BLOCK : M;
      body(lo, hi);
    }
  }
  void loop(unsigned index) {
    _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);
    _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_ON);
    unsigned seen = 0;
    for (;;) {
      unsigned g; int spins = 0;
      while ((g = gen.load(std::memory_order_acquire)) == seen) {   // spin for a moment (back-to-back jobs), then sleep (the GPU tests share this host)
        if (quit.load(std::memory_order_relaxed)) return;
        if (++spins < 20000) { _mm_pause(); continue; }
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return gen.load(std::memory_order_acquire) != seen || quit.load() != 0u; });
      }
      seen = g;
      if (index < active) work();
      done.fetch_add(1, std::memory_order_release);
    }
  }
  void ensure(unsigned n) {                                   // n workers besides the caller
    if (cpus.empty()) {
      cpu_set_t set; CPU_ZERO(&set);
      if (sched_getaffinity(0, sizeof(set), &set) == 0) for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &set)) cpus.push_back(c);
      if (cpus.empty()) cpus.push_back(0);
    }
    while (workers.size() < n) {
      const unsigned i = (unsigned)workers.size();
      workers.emplace_back([this, i] { loop(i); });
      cpu_set_t set; CPU_ZERO(&set); CPU_SET(cpus[(i + 1) % cpus.size()], &set);
      pthread_setaffinity_np(workers.back().native_handle(), sizeof(set), &set);
    }
  }
  void dispatch() {
    next.store(0); done.store(0);
    { std::lock_guard<std::mutex> lk(m); gen.fetch_add(1, std::memory_order_release); }
    cv.notify_all();
  }
  void wait_all() { const unsigned all = (unsigned)workers.size(); while (done.load(std::memory_order_acquire) < all) _mm_pause(); }
  double run(unsigned M_, int threads, std::function<void(unsigned, unsigned)> f) {
    std::lock_guard<std::mutex> one(jobs);
    if (threads < 1) threads = 1;
    ensure((unsigned)threads - 1);                            // (outside the clock)
    _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);
    _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_ON);
    body = std::move(f); active = (unsigned)threads - 1;
    M = 0; nblocks = 0; dispatch(); wait_all();               // an empty job first: every worker is awake and spinning when the clock starts
    M = M_; nblocks = (M + BLOCK - 1) / BLOCK;
    const double t0 = now();
    dispatch();
    work();
    wait_all();
    return now() - t0;
  }
  ~Pool() { quit.store(1); { std::lock_guard<std::mutex> lk(m); gen.fetch_add(1); } cv.notify_all(); for (auto& t : workers) t.join(); }
};
Pool& pool() { static Pool p; return p; }
template <typename F>
double run_blocks(unsigned M, int threads, F&& body) { return pool().run(M, threads, std::function<void(unsigned, unsigned)>(body)); }
}  // namespace

extern "C" {

__attribute__((visibility("default"))) void* refd_new(const char* cfg) {
  RefScene* s = new RefScene;
  s->device = rtcNewDevice(cfg);
  if (!s->device) { delete s; return nullptr; }
  s->scene = rtcNewScene(s->device);
  return s;
}

__attribute__((visibility("default"))) void refd_set_flags(void* h, int flags, int quality) {
  RefScene* s = (RefScene*)h;
  rtcSetSceneFlags(s->scene, (RTCSceneFlags)flags);
  rtcSetSceneBuildQuality(s->scene, (RTCBuildQuality)quality);
}

// copies the arrays into library-owned buffers (rtcSetNewGeometryBuffer pads them)
__attribute__((visibility("default"))) unsigned refd_add_mesh(void* h, const float* verts, unsigned nv,
                                                              const unsigned* idx, unsigned nt, unsigned mask) {
  RefScene* s = (RefScene*)h;
  RTCGeometry g = rtcNewGeometry(s->device, RTC_GEOMETRY_TYPE_TRIANGLE);
  float* v = (float*)rtcSetNewGeometryBuffer(g, RTC_BUFFER_TYPE_VERTEX, 0, RTC_FORMAT_FLOAT3, 12, nv);
  unsigned* t = (unsigned*)rtcSetNewGeometryBuffer(g, RTC_BUFFER_TYPE_INDEX, 0, RTC_FORMAT_UINT3, 12, nt);
  if (v && nv) memcpy(v, verts, (size_t)nv * 12);
  if (t && nt) memcpy(t, idx, (size_t)nt * 12);
  rtcSetGeometryMask(g, mask);
  rtcCommitGeometry(g);
  unsigned id = rtcAttachGeometry(s->scene, g);
  rtcReleaseGeometry(g);
  return id;
}

// RTC_GEOMETRY_TYPE_QUAD: index buffer of uint4 (rtcore_geometry.h), same vertex buffer layout
__attribute__((visibility("default"))) unsigned refd_add_quads(void* h, const float* verts, unsigned nv,
                                                               const unsigned* idx, unsigned nq, unsigned mask) {
  RefScene* s = (RefScene*)h;
  RTCGeometry g = rtcNewGeometry(s->device, RTC_GEOMETRY_TYPE_QUAD);
  float* v = (float*)rtcSetNewGeometryBuffer(g, RTC_BUFFER_TYPE_VERTEX, 0, RTC_FORMAT_FLOAT3, 12, nv);
  unsigned* t = (unsigned*)rtcSetNewGeometryBuffer(g, RTC_BUFFER_TYPE_INDEX, 0, RTC_FORMAT_UINT4, 16, nq);
  if (v && nv) memcpy(v, verts, (size_t)nv * 12);
  if (t && nq) memcpy(t, idx, (size_t)nq * 16);
  rtcSetGeometryMask(g, mask);
  rtcCommitGeometry(g);
  unsigned id = rtcAttachGeometry(s->scene, g);
  rtcReleaseGeometry(g);
  return id;
}

// a second scene on the SAME device (rtcRetainDevice), to be instanced by the first: RTC_GEOMETRY_TYPE_INSTANCE needs both on one device
__attribute__((visibility("default"))) void* refd_new_object(void* parent) {
  RefScene* p = (RefScene*)parent;
  RefScene* s = new RefScene;
  s->device = p->device; rtcRetainDevice(s->device);
  s->scene = rtcNewScene(s->device);
  return s;
}

// rtcNewGeometry(INSTANCE) + rtcSetGeometryInstancedScene + rtcSetGeometryTransform(FLOAT3X4_COLUMN_MAJOR: vx, vy, vz, p)
// (tutorials/instanced_geometry/instanced_geometry_device.cpp:145-160 is the idiom); the object scene must be committed by the caller
__attribute__((visibility("default"))) unsigned refd_add_instance(void* h, void* object, const float* xfm12, unsigned mask) {
  RefScene* s = (RefScene*)h;
  RTCGeometry g = rtcNewGeometry(s->device, RTC_GEOMETRY_TYPE_INSTANCE);
  rtcSetGeometryInstancedScene(g, ((RefScene*)object)->scene);
  rtcSetGeometryTimeStepCount(g, 1);
  rtcSetGeometryTransform(g, 0, RTC_FORMAT_FLOAT3X4_COLUMN_MAJOR, xfm12);
  rtcSetGeometryMask(g, mask);
  rtcCommitGeometry(g);
  unsigned id = rtcAttachGeometry(s->scene, g);
  rtcReleaseGeometry(g);
  return id;
}

__attribute__((visibility("default"))) double refd_commit(void* h) {
  RefScene* s = (RefScene*)h;
  double t0 = now();
  rtcCommitScene(s->scene);
  return now() - t0;
}

__attribute__((visibility("default"))) int refd_error(void* h) {
  RefScene* s = (RefScene*)h;
  return (int)rtcGetDeviceError(s ? s->device : nullptr);
}

__attribute__((visibility("default"))) void refd_bounds(void* h, float* out8) {
  RefScene* s = (RefScene*)h;
  RTCBounds b;
  rtcGetSceneBounds(s->scene, &b);
  memcpy(out8, &b, 32);
}

__attribute__((visibility("default"))) double refd_intersect1(void* h, RTCRayHit* rh, unsigned M, int threads) {
  RefScene* s = (RefScene*)h;
  return run_blocks(M, threads, [&](unsigned lo, unsigned hi) {
    for (unsigned i = lo; i < hi; i++) rtcIntersect1(s->scene, &rh[i]);
  });
}

__attribute__((visibility("default"))) double refd_occluded1(void* h, RTCRay* r, unsigned M, int threads) {
  RefScene* s = (RefScene*)h;
  return run_blocks(M, threads, [&](unsigned lo, unsigned hi) {
    for (unsigned i = lo; i < hi; i++) rtcOccluded1(s->scene, &r[i]);
  });
}

// The CPU-baseline job of bench.py: `tiles` copies of the M records, one after the other (verify.cpp:5923-5983 runs 16 Mi rays), in a buffer the POOL fills --
// first touch by the threads that will trace it, like the reference's benchmark, whose rays are made inside its parallel tasks.  (A 1.5 GB numpy array written
// by one Python thread sits on one NUMA node: 256 threads then traced it at 20 Mrays/s instead of 200.)  Returns the seconds of the traced pass only.
// mode bit 0: static ranges (worker i fills AND traces records [i n / T, (i + 1) n / T): every page is traced by the thread that touched it first) instead of the
// 1024-ray blocks handed out dynamically; bit 1: madvise(MADV_HUGEPAGE) on the buffer.
__attribute__((visibility("default"))) double refd_run_tiled_mode(void* h, const void* rays, unsigned M, unsigned tiles, int any, int threads, unsigned mode);
__attribute__((visibility("default"))) double refd_run_tiled(void* h, const void* rays, unsigned M, unsigned tiles, int any, int threads) { return refd_run_tiled_mode(h, rays, M, tiles, any, threads, 0u); }
__attribute__((visibility("default"))) double refd_run_tiled_mode(void* h, const void* rays, unsigned M, unsigned tiles, int any, int threads, unsigned mode) {
  RefScene* s = (RefScene*)h;
  const size_t rec = any ? sizeof(RTCRay) : sizeof(RTCRayHit);
  const unsigned long long total64 = (unsigned long long)M * tiles;
  if (M == 0 || tiles == 0 || total64 > 0xFFFFFFFFull) return -1.0;
  const unsigned total = (unsigned)total64;
  char* buf = (char*)aligned_alloc(2u << 20, (((size_t)total * rec) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1));
  if (!buf) return -1.0;
  if (mode & 2u) madvise(buf, (size_t)total * rec, MADV_HUGEPAGE);
  const char* src = (const char*)rays;
  if (mode & 1u) {
    const unsigned T = (unsigned)(threads < 1 ? 1 : threads);
    // one "block" per thread: run_blocks over T * 1024 pseudo-records, block b = thread slice b (each worker takes about one; a slice is filled and traced as a unit)
    auto slice = [&](unsigned b, unsigned& lo, unsigned& hi) { lo = (unsigned)((unsigned long long)total * b / T); hi = (unsigned)((unsigned long long)total * (b + 1) / T); };
    run_blocks(T * 1024u, threads, [&](unsigned blo, unsigned) { unsigned lo, hi; slice(blo / 1024u, lo, hi); for (unsigned i = lo; i < hi; i++) memcpy(buf + (size_t)i * rec, src + (size_t)(i % M) * rec, rec); });
    const double dt = run_blocks(T * 1024u, threads, [&](unsigned blo, unsigned) { unsigned lo, hi; slice(blo / 1024u, lo, hi);
      if (any) for (unsigned i = lo; i < hi; i++) rtcOccluded1(s->scene, (RTCRay*)(buf + (size_t)i * rec));
      else for (unsigned i = lo; i < hi; i++) rtcIntersect1(s->scene, (RTCRayHit*)(buf + (size_t)i * rec)); });
    free(buf);
    return dt;
  }
  run_blocks(total, threads, [&](unsigned lo, unsigned hi) { for (unsigned i = lo; i < hi; i++) memcpy(buf + (size_t)i * rec, src + (size_t)(i % M) * rec, rec); });
  const double dt = any ? run_blocks(total, threads, [&](unsigned lo, unsigned hi) { for (unsigned i = lo; i < hi; i++) rtcOccluded1(s->scene, (RTCRay*)(buf + (size_t)i * rec)); })
                        : run_blocks(total, threads, [&](unsigned lo, unsigned hi) { for (unsigned i = lo; i < hi; i++) rtcIntersect1(s->scene, (RTCRayHit*)(buf + (size_t)i * rec)); });
  free(buf);
  return dt;
}

// packet entry points, driven from AoS input for convenience: gathers K rays
// into an RTCRayHitK, calls rtcIntersectK with all-valid mask (-1 = active,
// kernels/bvh/bvh_intersector_hybrid.cpp:127), scatters back.
#define REFD_PACKET(K, ALIGN)                                                                     \
  __attribute__((visibility("default"))) double refd_intersect##K(void* h, RTCRayHit* rh,         \
                                                                  unsigned M, int threads) {      \
    RefScene* s = (RefScene*)h;                                                                   \
    return run_blocks(M, threads, [&](unsigned lo, unsigned hi) {                                 \
      for (unsigned b = lo; b < hi; b += K) {                                                     \
        alignas(ALIGN) RTCRayHit##K p;                                                            \
        alignas(ALIGN) int valid[K];                                                              \
        for (unsigned k = 0; k < K; k++) {                                                        \
          unsigned i = b + k;                                                                     \
          valid[k] = i < hi ? -1 : 0;                                                             \
          const RTCRayHit& r = rh[i < hi ? i : hi - 1];                                           \
          p.ray.org_x[k] = r.ray.org_x; p.ray.org_y[k] = r.ray.org_y; p.ray.org_z[k] = r.ray.org_z; \
          p.ray.tnear[k] = r.ray.tnear;                                                           \
          p.ray.dir_x[k] = r.ray.dir_x; p.ray.dir_y[k] = r.ray.dir_y; p.ray.dir_z[k] = r.ray.dir_z; \
          p.ray.time[k] = r.ray.time; p.ray.tfar[k] = r.ray.tfar; p.ray.mask[k] = r.ray.mask;     \
          p.ray.id[k] = r.ray.id; p.ray.flags[k] = r.ray.flags;                                   \
          p.hit.geomID[k] = r.hit.geomID; p.hit.primID[k] = r.hit.primID;                         \
          p.hit.instID[0][k] = r.hit.instID[0];                                                   \
        }                                                                                         \
        rtcIntersect##K(valid, s->scene, &p);                                                     \
        for (unsigned k = 0; k < K && b + k < hi; k++) {                                          \
          RTCRayHit& r = rh[b + k];                                                               \
          if (p.hit.geomID[k] == RTC_INVALID_GEOMETRY_ID) continue;                               \
          r.ray.tfar = p.ray.tfar[k];                                                             \
          r.hit.Ng_x = p.hit.Ng_x[k]; r.hit.Ng_y = p.hit.Ng_y[k]; r.hit.Ng_z = p.hit.Ng_z[k];     \
          r.hit.u = p.hit.u[k]; r.hit.v = p.hit.v[k];                                             \
          r.hit.primID = p.hit.primID[k]; r.hit.geomID = p.hit.geomID[k];                         \
          r.hit.instID[0] = p.hit.instID[0][k];                                                   \
        }                                                                                         \
      }                                                                                           \
    });                                                                                           \
  }
REFD_PACKET(4, 16)
REFD_PACKET(8, 32)


// General packet driver: rtcIntersectK / rtcOccludedK (K = 4, 8, 16) of the REAL library over an AoS array, with a per-ray
// valid flag (valid[i] != 0 -> lane active = -1, else 0: InactiveRaysTest, tutorials/verify/verify.cpp:3553) so that the GPU
// packet entry points can be compared with the reference's packet code path (BVHNIntersectorKHybrid, bvh_intersector_hybrid.cpp:106-369;
// rtcIntersect16 on an AVX2 build takes the library's own single-ray fallback, rtcore.cpp:889-897).  Returns seconds.
}  // extern "C"
namespace {
template <int K> struct PK;
template <> struct PK<4>  { typedef RTCRayHit4 RH;  typedef RTCRay4 R;  static void isect(const int* v, RTCScene s, RH* p) { rtcIntersect4(v, s, p); }  static void occl(const int* v, RTCScene s, R* p) { rtcOccluded4(v, s, p); } };
template <> struct PK<8>  { typedef RTCRayHit8 RH;  typedef RTCRay8 R;  static void isect(const int* v, RTCScene s, RH* p) { rtcIntersect8(v, s, p); }  static void occl(const int* v, RTCScene s, R* p) { rtcOccluded8(v, s, p); } };
template <> struct PK<16> { typedef RTCRayHit16 RH; typedef RTCRay16 R; static void isect(const int* v, RTCScene s, RH* p) { rtcIntersect16(v, s, p); } static void occl(const int* v, RTCScene s, R* p) { rtcOccluded16(v, s, p); } };
template <int K, typename RayK> void load_ray(RayK& p, unsigned k, const RTCRay& r) {
  p.org_x[k] = r.org_x; p.org_y[k] = r.org_y; p.org_z[k] = r.org_z; p.tnear[k] = r.tnear;
  p.dir_x[k] = r.dir_x; p.dir_y[k] = r.dir_y; p.dir_z[k] = r.dir_z; p.time[k] = r.time;
  p.tfar[k] = r.tfar; p.mask[k] = r.mask; p.id[k] = r.id; p.flags[k] = r.flags;
}
template <int K> double packet_run(RefScene* s, void* rays, unsigned M, size_t stride, const int* valid, int any, int threads) {
  return run_blocks(M, threads, [&](unsigned lo, unsigned hi) {
    for (unsigned b = lo; b < hi; b += K) {
      alignas(64) typename PK<K>::RH p;
      alignas(64) int v[K];
      memset(&p, 0, sizeof(p));
      for (unsigned k = 0; k < K; k++) {
        const unsigned i = b + k;
        v[k] = (i < hi && (!valid || valid[i])) ? -1 : 0;
        const char* rec = (const char*)rays + (size_t)(i < hi ? i : hi - 1) * stride;
        load_ray<K>(p.ray, k, *(const RTCRay*)rec);
        if (!any) { const RTCRayHit& rh = *(const RTCRayHit*)rec; p.hit.geomID[k] = rh.hit.geomID; p.hit.primID[k] = rh.hit.primID; p.hit.instID[0][k] = rh.hit.instID[0]; }
      }
      if (any) PK<K>::occl(v, s->scene, &p.ray); else PK<K>::isect(v, s->scene, &p);
      for (unsigned k = 0; k < K && b + k < hi; k++) {
        char* rec = (char*)rays + (size_t)(b + k) * stride;
        // inactive lanes: the packet's tfar / hit.geomID ARE copied back (so that the caller sees whether the library touched them), the fields this
        // driver zero-filled are not
        if (any) { ((RTCRay*)rec)->tfar = p.ray.tfar[k]; continue; }
        if (v[k] != -1) { RTCRayHit& r = *(RTCRayHit*)rec; r.ray.tfar = p.ray.tfar[k]; r.hit.primID = p.hit.primID[k]; r.hit.geomID = p.hit.geomID[k]; r.hit.instID[0] = p.hit.instID[0][k]; continue; }
        RTCRayHit& r = *(RTCRayHit*)rec;
        r.ray.tfar = p.ray.tfar[k];
        r.hit.Ng_x = p.hit.Ng_x[k]; r.hit.Ng_y = p.hit.Ng_y[k]; r.hit.Ng_z = p.hit.Ng_z[k];
        r.hit.u = p.hit.u[k]; r.hit.v = p.hit.v[k];
        r.hit.primID = p.hit.primID[k]; r.hit.geomID = p.hit.geomID[k]; r.hit.instID[0] = p.hit.instID[0][k];
      }
    }
  });
}
}  // namespace
extern "C" {
__attribute__((visibility("default"))) double refd_packet(void* h, int K, int any, void* rays, unsigned M, const int* valid, int threads) {
  RefScene* s = (RefScene*)h;
  const size_t stride = any ? sizeof(RTCRay) : sizeof(RTCRayHit);
  if (K == 4) return packet_run<4>(s, rays, M, stride, valid, any, threads);
  if (K == 8) return packet_run<8>(s, rays, M, stride, valid, any, threads);
  if (K == 16) return packet_run<16>(s, rays, M, stride, valid, any, threads);
  return -1.0;
}

// ---- filter callbacks: two fixed rules, installed on the REAL library, so that the GPU path's host-side filter loop (rtcore_api.cpp: filtered_query) can be
// compared with the reference's in-traversal filters (kernels/geometry/filter.h).  tests/test_gpu_round2.py installs the same rules through its own callbacks.
//   geometry rule : reject a hit whose primID is a multiple of 3, or whose u is above 0.7
//   argument rule : reject a hit with (primID + 2 * geomID) % 5 == 1
static void refd_rule_geometry(const RTCFilterFunctionNArguments* a) {
  for (unsigned i = 0; i < a->N; i++) {
    if (a->valid[i] != -1) continue;
    if (RTCHitN_primID(a->hit, a->N, i) % 3u == 0u || RTCHitN_u(a->hit, a->N, i) > 0.7f) a->valid[i] = 0;
  }
}
static void refd_rule_argument(const RTCFilterFunctionNArguments* a) {
  for (unsigned i = 0; i < a->N; i++) {
    if (a->valid[i] != -1) continue;
    if ((RTCHitN_primID(a->hit, a->N, i) + 2u * RTCHitN_geomID(a->hit, a->N, i)) % 5u == 1u) a->valid[i] = 0;
  }
}
// mode bit 0: geometry rule as intersect filter, bit 1: as occluded filter, bit 2: the geometries accept the argument filter; on geometries 0 .. ngeom-1
__attribute__((visibility("default"))) void refd_set_filters(void* h, unsigned ngeom, unsigned mode) {
  RefScene* s = (RefScene*)h;
  for (unsigned id = 0; id < ngeom; id++) {
    RTCGeometry g = rtcGetGeometry(s->scene, id);
    if (!g) continue;
    rtcSetGeometryIntersectFilterFunction(g, (mode & 1u) ? refd_rule_geometry : nullptr);
    rtcSetGeometryOccludedFilterFunction(g, (mode & 2u) ? refd_rule_geometry : nullptr);
    rtcSetGeometryEnableFilterFunctionFromArguments(g, (mode & 4u) != 0u);
    rtcCommitGeometry(g);
  }
  rtcCommitScene(s->scene);
}
// rtcIntersect1 / rtcOccluded1 with RTCIntersectArguments: argRule != 0 passes the argument rule as args.filter, flags as given (e.g. INVOKE_ARGUMENT_FILTER)
__attribute__((visibility("default"))) double refd_intersect1_args(void* h, RTCRayHit* rh, unsigned M, int threads, int argRule, unsigned flags) {
  RefScene* s = (RefScene*)h;
  return run_blocks(M, threads, [&](unsigned lo, unsigned hi) {
    RTCIntersectArguments a; rtcInitIntersectArguments(&a);
    a.flags = (RTCRayQueryFlags)flags; a.filter = argRule ? refd_rule_argument : nullptr;
    for (unsigned i = lo; i < hi; i++) rtcIntersect1(s->scene, &rh[i], &a);
  });
}
__attribute__((visibility("default"))) double refd_occluded1_args(void* h, RTCRay* r, unsigned M, int threads, int argRule, unsigned flags) {
  RefScene* s = (RefScene*)h;
  return run_blocks(M, threads, [&](unsigned lo, unsigned hi) {
    RTCOccludedArguments a; rtcInitOccludedArguments(&a);
    a.flags = (RTCRayQueryFlags)flags; a.filter = argRule ? refd_rule_argument : nullptr;
    for (unsigned i = lo; i < hi; i++) rtcOccluded1(s->scene, &r[i], &a);
  });
}

__attribute__((visibility("default"))) void refd_free(void* h) {
  RefScene* s = (RefScene*)h;
  if (!s) return;
  if (s->scene) rtcReleaseScene(s->scene);
  if (s->device) rtcReleaseDevice(s->device);
  delete s;
}

__attribute__((visibility("default"))) unsigned refd_sizeof_rayhit() { return (unsigned)sizeof(RTCRayHit); }
__attribute__((visibility("default"))) unsigned refd_hw_threads() { return std::thread::hardware_concurrency(); }
// which single-ISA build of the reference this driver is linked to (RTC_DEVICE_PROPERTY_NATIVE_RAY16_SUPPORTED is 1 only for the AVX-512 build, rtcore.cpp / device.cpp:getProperty)
__attribute__((visibility("default"))) int refd_native16(void* h) { return (int)rtcGetDeviceProperty(((RefScene*)h)->device, RTC_DEVICE_PROPERTY_NATIVE_RAY16_SUPPORTED); }
}
