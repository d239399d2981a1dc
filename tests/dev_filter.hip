// tests/dev_filter.hip -- TEST INFRASTRUCTURE: an application's __device__ filter function, compiled on its own into tests/golden/_bin/libdevfilter.so
// (hipcc --offload-arch=gfx950 -shared -fPIC; __graft_entry__.build()), the way a user of rtcIntersect1MDevice with device_filter_functions=1 would ship one.
// The function is the ARGUMENT rule of oracle/ref_driver.cpp (refd_rule_argument: reject a hit with (primID + 2 * geomID) % 5 == 1), so that the real reference running that
// rule as a host callback inside its traversal is the checker (tests/test_gpu_round5.py).  It also counts its calls through args->context (a device counter).
#include <hip/hip_runtime.h>
#include <stdint.h>
// RTCFilterFunctionNArguments with N = 1 (include/embree4/rtcore.h; rtcore_common.h:318-326 of the reference): SoA of one lane = the plain RTCRay / RTCHit layouts
struct FilterArgs { int* valid; void* geometryUserPtr; void* context; float* ray; uint32_t* hit; unsigned int N; };
struct Counters { unsigned long long calls, rejected, userPtrSum; };

extern "C" __device__ void devfilter_argument_rule(const FilterArgs* a) {
  if (a->valid[0] != -1) return;
  const uint32_t primID = a->hit[5], geomID = a->hit[6];        // RTCHit: Ng_x Ng_y Ng_z u v primID geomID instID[0]
  Counters* c = (Counters*)a->context;
  if (c) { atomicAdd(&c->calls, 1ull); atomicAdd(&c->userPtrSum, (unsigned long long)(uintptr_t)a->geometryUserPtr); }
  if ((primID + 2u * geomID) % 5u == 1u) { a->valid[0] = 0; if (c) atomicAdd(&c->rejected, 1ull); }
}
// the address, as the device sees it (a function address is a device-side constant: it is stored in a __device__ variable by the loader and read back)
typedef void (*FilterFn)(const FilterArgs*);
__device__ FilterFn g_devfilter_argument_rule = devfilter_argument_rule;

// (debugging aids of round 5: what exactly a cross-code-object callee may do -- tests/gpu_devfilter.py WHICH=1|2|3)
extern "C" __device__ void devfilter_empty(const FilterArgs*) {}
extern "C" __device__ void devfilter_read_only(const FilterArgs* a) { if (a->valid[0] == -1 && a->hit[5] == 0xFFFFFFF0u) a->valid[0] = 0; }
extern "C" __device__ void devfilter_reject_no_counter(const FilterArgs* a) { if (a->valid[0] == -1 && (a->hit[5] + 2u * a->hit[6]) % 5u == 1u) a->valid[0] = 0; }
__device__ FilterFn g_devfilter_table[4] = {devfilter_argument_rule, devfilter_empty, devfilter_read_only, devfilter_reject_no_counter};
extern "C" __attribute__((visibility("default"))) uint64_t devfilter_address_of(int which) {
  FilterFn t[4] = {nullptr, nullptr, nullptr, nullptr};
  if (hipMemcpyFromSymbol(t, HIP_SYMBOL(g_devfilter_table), sizeof(t)) != hipSuccess) return 0;
  return (uint64_t)(uintptr_t)t[which & 3];
}

extern "C" __attribute__((visibility("default"))) uint64_t devfilter_address(void) {
  FilterFn p = nullptr;
  if (hipMemcpyFromSymbol(&p, HIP_SYMBOL(g_devfilter_argument_rule), sizeof(p)) != hipSuccess) return 0;
  return (uint64_t)(uintptr_t)p;
}
