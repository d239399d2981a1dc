// tools/fetch_calib.hip -- what does rocprofv3's FETCH_SIZE count for SCATTERED 16-byte loads on gfx950?  (the x2 correction of MI355X_MICROARCH.md is
// calibrated for wide coalesced streaming reads only; the traversal kernel's node and triangle fetches are 16-byte loads at unrelated addresses.)
// Three kernels over one 4 GiB buffer (16 x the Infinity Cache), each with a KNOWN number of bytes asked for and of distinct 64 B / 128 B lines touched:
//   stream16   every lane reads consecutive 16-byte pieces (the guide's calibration case)             bytes = 4 GiB
//   scatter16  every lane reads ONE 16-byte piece at a pseudo-random 16-byte-aligned address            bytes = loads x 16; lines touched ~ loads (4 GiB / 64 B = 64 Mi lines, 4 Mi loads)
//   node80     every lane reads five consecutive 16-byte pieces of a random 80-byte record (a CNode)    bytes = records x 80; 64-byte lines touched = 2 or 3 per record
// Run:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- tools/fetch_calib      (then FETCH_SIZE x 1024 / bytes per kernel = the factor; tools/r05f.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void stream16(const u32x4* p, size_t n16, uint32_t* sink) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const u32x4 v = p[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}
__global__ void scatter16(const u32x4* p, size_t n16, size_t loads, uint32_t* sink) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < loads; i += (size_t)gridDim.x * blockDim.x) { const u32x4 v = p[mix(i + 1) % n16]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}
__global__ void node80(const u32x4* p, size_t nrec, size_t recs, uint32_t* sink) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < recs; i += (size_t)gridDim.x * blockDim.x) {
    const u32x4* q = p + (mix(i + 7) % nrec) * 5;
    const u32x4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
    acc.x ^= a.x ^ b.x ^ c.x ^ d.x ^ e.x; acc.y ^= a.y ^ e.w;
  }
  if ((acc.x ^ acc.y) == 0x12345678u) *sink = 1;
}
int main() {
  const size_t bytes = 4ull << 30, n16 = bytes / 16, loads = 4u << 20, recs = 2u << 20;
  u32x4* p; uint32_t* sink;
  if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(p, 1, bytes); hipDeviceSynchronize();
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(stream16, dim3(256 * 16), dim3(256), 0, 0, p, n16, sink);
    hipLaunchKernelGGL(scatter16, dim3(256 * 16), dim3(256), 0, 0, p, n16, loads, sink);
    hipLaunchKernelGGL(node80, dim3(256 * 16), dim3(256), 0, 0, p, n16 / 5, recs, sink);
    hipDeviceSynchronize();
  }
  printf("CALIB stream16 bytes %zu | scatter16 loads %zu bytes %zu | node80 records %zu bytes %zu\n", bytes, loads, loads * 16, recs, recs * 80);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
