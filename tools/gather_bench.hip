// gather_bench.hip -- micro-benchmark behind DESIGN.md §"why one ray per octet": how many dependent,
// incoherent 128-byte node visits per second does the MI355X memory pipeline sustain when
//   mode 0: every LANE chases its own node and reads it with L 16-byte loads (lane-per-ray layout), or
//   mode 1: every OCTET (8 lanes) chases one node and each lane reads 16 bytes of it (octet layout).
// The next index depends on the loaded data, like a BVH traversal.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE, int L>
__global__ __launch_bounds__(256) void chase(const float4* __restrict__ nodes, uint32_t numNodes, uint32_t steps, uint32_t* out) {
  const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  uint32_t idx = (MODE == 0 ? tid : (tid >> 3)) * 2654435761u % numNodes;
  const uint32_t sub = threadIdx.x & 7;
  uint32_t acc = 0;
  for (uint32_t s = 0; s < steps; s++) {
    const float4* p = nodes + (size_t)idx * 8;      // 128 B node = 8 x float4
    uint32_t h = 0;
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < L; k++) { float4 v = p[k]; h ^= __float_as_uint(v.x) + __float_as_uint(v.w); }
    } else {
      float4 v = p[sub];
      h = __float_as_uint(v.x) + __float_as_uint(v.w);
      h ^= __shfl_xor(h, 1); h ^= __shfl_xor(h, 2); h ^= __shfl_xor(h, 4);   // octet-uniform
    }
    acc += h;
    idx = (idx * 1664525u + 1013904223u + h) % numNodes;
  }
  out[tid] = acc;
}

int main(int argc, char** argv) {
  const uint32_t mb = argc > 1 ? atoi(argv[1]) : 24;         // node array size in MiB
  const uint32_t numNodes = mb * (1u << 20) / 128, steps = 2000;
  std::vector<float> h((size_t)numNodes * 32);
  for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i * 2654435761u % 1000003u);
  float4* d; uint32_t* out;
  hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int blocks = 256 * 8;                                   // 8 blocks x 256 threads per CU = 32 waves/CU
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, void (*k)(const float4*, uint32_t, uint32_t, uint32_t*), double visitsPerThreadStep) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, numNodes, 100u, out);
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, numNodes, steps, out); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    const double visits = (double)blocks * 256 * steps * visitsPerThreadStep;
    printf("GATHER %-34s %4u MiB: %8.3f ms  %8.2f G node-visits/s  (%.1f cycles/visit/CU at 2.4 GHz)\n", name, mb, ms, visits / ms / 1e6,
           ms * 1e-3 * 2.4e9 * 256 / visits);
  };
  run("lane-per-ray 5x16B (80 B node)", chase<0, 5>, 1.0);
  run("lane-per-ray 8x16B (128 B node)", chase<0, 8>, 1.0);
  run("lane-per-ray 3x16B (48 B tri)", chase<0, 3>, 1.0);
  run("octet 1x16B/lane (128 B node)", chase<1, 1>, 1.0 / 8);
  return 0;
}
