// hipMalloc / hipFree cost by size on one MI355X (the 180 GB kernel matrix of BASELINE configs[4] spends 3.5-5 s outside
// its kernels):  hipcc --offload-arch=gfx950 tools/malloc_probe.hip -o /tmp/malloc_probe && /tmp/malloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(nullptr);
  for (int rep = 0; rep < 2; ++rep)
    for (double gb : {8.0, 32.0, 64.0, 128.0, 160.0, 180.0, 200.0}) {
      void* p = nullptr;
      size_t bytes = (size_t)(gb * 1e9);
      double t0 = now();
      hipError_t e = hipMalloc(&p, bytes);
      double t1 = now();
      if (e != hipSuccess) { printf("%6.0f GB: hipMalloc failed (%s)\n", gb, hipGetErrorString(e)); continue; }
      hipMemsetAsync(p, 0, bytes, 0);
      hipDeviceSynchronize();
      double t2 = now();
      hipFree(p);
      double t3 = now();
      printf("rep %d %6.0f GB: hipMalloc %.3f s, memset %.3f s (%.0f GB/s), hipFree %.3f s\n", rep, gb, t1 - t0, t2 - t1, gb / (t2 - t1), t3 - t2);
    }
  return 0;
}
