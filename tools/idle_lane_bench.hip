// idle_lane_bench.hip -- what do the lanes of a wave cost the CU's address path when they do NOT need a node?  (round 6; profiles/r06_idle_lane_loads.md)
// The traversal kernel issues its five node loads from EVERY lane (step 3a: a lane without a node reads the root, so that s_waitcnt can count loads); ~20 % of the lanes of a
// node step are such lanes.  Modes, 64 lanes x 5 x 16-byte loads per iteration, nodes of 80 bytes at random places of an array that fits the L2:
//   0: every lane reads its own node                         (100 % scattered)
//   1: 4 of 5 lanes read their own node, every 5th the root  (what the kernel does)
//   2: 4 of 5 lanes read their own node, every 5th is masked off (EXEC) for the loads
//   3: as 1, but the idle lanes are the LAST 13 lanes of the wave (whole quads idle) -- 4: as 2 with those lanes
//   5: every lane its own node, but the 64 lanes of a wave-load stay inside ONE 2 MB window (chosen at random per iteration): as many lines from beyond the L2 as mode 0, one
//      translation per wave-load instead of up to 64 -- what address translation costs a scattered gather (would a tree laid out sub-tree by sub-tree pay?)
// Build: hipcc --offload-arch=gfx950 -O3 tools/idle_lane_bench.hip -o /tmp/idle_lane_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(64) void k(const uint4* __restrict__ nodes, uint32_t numNodes, uint32_t iters, uint32_t* out) {
  const uint32_t tid = blockIdx.x * 64 + threadIdx.x, lane = threadIdx.x;
  const bool idle = (MODE == 1 || MODE == 2) ? (lane % 5u == 4u) : (MODE >= 3 ? lane >= 51u : false);
  uint32_t s = tid * 2654435761u + 12345u, acc = 0;
  for (uint32_t i = 0; i < iters; i++) {
    s = s * 1664525u + 1013904223u;
    uint32_t idx = (s >> 8) % numNodes;
    if ((MODE == 1 || MODE == 3) && idle) idx = 0;
    if (MODE == 5) { const uint32_t win = 26214u, nw = numNodes / win;               // 26214 nodes x 80 B = 2 MB
      uint32_t ws = (blockIdx.x * 2654435761u + i * 40503u) * 1664525u + 1013904223u; idx = nw ? ((ws >> 8) % nw) * win + (s >> 8) % win : idx; }
    if ((MODE == 2 || MODE == 4) && idle) continue;            // (divergent: the loads below run with these lanes masked off)
    const uint4* p = nodes + (size_t)idx * 5u;
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    acc ^= a.x + b.y + c.z + d.w + e.x;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const uint32_t numNodes = argc > 1 ? atoi(argv[1]) : 32768;   // x 80 B = 2.6 MB: inside one XCD's L2
  const uint32_t iters = 2000, waves = 256 * 16;
  uint4* d; uint32_t* o;
  hipMalloc(&d, (size_t)numNodes * 80); hipMemset(d, 1, (size_t)numNodes * 80); hipMalloc(&o, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[6] = {"all lanes scattered", "1 in 5 reads the root", "1 in 5 masked off", "last 13 lanes read the root", "last 13 lanes masked off", "scattered inside a 2 MB window"};
  for (int rep = 0; rep < 2; rep++)
  for (int m = 0; m < 6; m++) {
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
      hipEventRecord(e0);
      switch (m) { case 0: k<0><<<waves, 64>>>(d, numNodes, iters, o); break; case 1: k<1><<<waves, 64>>>(d, numNodes, iters, o); break; case 2: k<2><<<waves, 64>>>(d, numNodes, iters, o); break;
                   case 3: k<3><<<waves, 64>>>(d, numNodes, iters, o); break; case 4: k<4><<<waves, 64>>>(d, numNodes, iters, o); break; default: k<5><<<waves, 64>>>(d, numNodes, iters, o); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double waveLoads = (double)waves * iters * 5;
    printf("IDLE nodes=%u mode %d (%-28s): %.3f ms, %.1f G wave-loads/s, %.2f clocks per wave-load per CU at 2.4 GHz\n", numNodes, m, names[m], best, waveLoads / best / 1e6, best * 1e-3 * 2.4e9 * 256 / waveLoads);
  }
  return 0;
}
