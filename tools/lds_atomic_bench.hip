// LDS atomic throughput on gfx950: cycles per ds_max_u32 wave-instruction for different lane -> address patterns (what bounds the binning kernels).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o tools/lds_atomic_bench && tools/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE> __global__ __launch_bounds__(256) void bench(unsigned long long* out, uint32_t iters) {
  __shared__ uint32_t s[4096];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  for (uint32_t i = tid; i < 4096u; i += 256u) s[i] = 0u;
  __syncthreads();
  uint32_t idx;
  if (MODE == 0) idx = 0u;                         // all lanes one word
  else if (MODE == 1) idx = (lane >> 4) * 7u;      // 4 words, 16 lanes each
  else if (MODE == 2) idx = (lane >> 2) * 7u;      // 16 words, 4 lanes each
  else if (MODE == 3) idx = lane;                  // 64 consecutive words
  else if (MODE == 4) idx = lane * 8u;             // 64 words, stride 8: 4 banks
  else idx = lane * 7u;                            // 64 words, stride 7
  idx += wave * 1024u;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (uint32_t i = 0; i < iters; i++) { atomicMax(&s[idx], i + lane); atomicMax(&s[idx + 1u], i ^ lane); atomicMax(&s[idx + 2u], i + 3u); atomicMax(&s[idx + 3u], i); }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  if (s[idx] == 0xFFFFFFFFu) out[0] = 0;
}
template <int MODE> void run(const char* what, int blocksPerCU) {
  unsigned long long* d; hipMalloc(&d, 4096 * 8);
  const uint32_t iters = 2000; const int blocks = 256 * blocksPerCU;
  hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipDeviceSynchronize();
  hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipDeviceSynchronize();
  unsigned long long h[4096]; hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (int i = 0; i < blocks; i++) sum += (double)h[i];
  // per CU: blocksPerCU * 4 waves * iters * 4 atomic wave-instructions in (sum / blocks) cycles (the counter runs at 100 MHz on some parts: reported raw)
  printf("%-40s %d blocks/CU: %.1f counter ticks per wave-instruction per CU\n", what, blocksPerCU, (sum / blocks) / ((double)blocksPerCU * 4 * iters * 4));
  hipFree(d);
}
int main() {
  for (int b : {1, 2}) {
    run<0>("64 lanes -> 1 word", b); run<1>("64 lanes -> 4 words (16 each)", b); run<2>("64 lanes -> 16 words (4 each)", b);
    run<3>("64 lanes -> 64 consecutive words", b); run<4>("64 lanes -> 64 words, stride 8", b); run<5>("64 lanes -> 64 words, stride 7", b);
  }
  return 0;
}
