// Host link probe for the host-array entry points (rtcIntersect1M on a pageable array): what pinning, linear copies, 2-D copies of the ray / hit halves cost.
//   hipcc --offload-arch=gfx950 -O2 tools/pcie_probe.hip -o tools/pcie_probe && tools/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main() {
  const size_t n = 1u << 20, rec = 96, bytes = n * rec;
  char* h = (char*)aligned_alloc(4096, bytes); memset(h, 1, bytes);
  char* d; CK(hipMalloc(&d, bytes));
  hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  for (int rep = 0; rep < 3; rep++) {
    double t0 = now(); CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); double t1 = now(); CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); double t2 = now();
    printf("pageable  H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) * 1e-9, (t2 - t1) * 1e3, bytes / (t2 - t1) * 1e-9);
  }
  for (int rep = 0; rep < 3; rep++) {
    double t0 = now(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); double t1 = now();
    CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t2 = now();
    CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); double t3 = now();
    CK(hipMemcpyAsync(d, h, bytes / 2, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(h + bytes / 2, d + bytes / 2, bytes / 2, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2)); double t4 = now();
    CK(hipMemcpy2DAsync(d, rec, h, rec, 48, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t5 = now();
    CK(hipMemcpy2DAsync(h + 48, rec, d + 48, rec, 48, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); double t6 = now();
    CK(hipHostUnregister(h)); double t7 = now();
    printf("register %.2f ms | pinned H2D %.2f ms (%.1f GB/s) D2H %.2f ms (%.1f GB/s) | both ways at once, half each: %.2f ms | 2-D width 48: H2D %.2f ms D2H %.2f ms | unregister %.2f ms\n",
           (t1 - t0) * 1e3, (t2 - t1) * 1e3, bytes / (t2 - t1) * 1e-9, (t3 - t2) * 1e3, bytes / (t3 - t2) * 1e-9, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3, (t7 - t6) * 1e3);
  }
  // a pinned bounce buffer filled by host threads (what a staging pipeline without hipHostRegister would do)
  char* p; CK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
  for (int threads : {1, 4, 8, 16}) {
    double best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      double t0 = now();
      std::vector<std::thread> th;
      for (int k = 0; k < threads; k++) th.emplace_back([=] { const size_t a = bytes * k / threads, b = bytes * (k + 1) / threads; memcpy(p + a, h + a, b - a); });
      for (auto& t : th) t.join();
      best = std::min(best, now() - t0);
    }
    printf("host memcpy pageable -> pinned, %2d threads: %.2f ms (%.1f GB/s)\n", threads, best * 1e3, bytes / best * 1e-9);
  }
  return 0;
}
