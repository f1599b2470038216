// Cost of getting large device buffers on one MI355X, three ways (the 127 / 180 GB kernel matrices of BASELINE configs[3] /
// [4] spent 3.5-6.5 s in hipMalloc in round 3):
//   malloc   hipMalloc / hipFree
//   pool     hipMallocAsync / hipFreeAsync on a pool that never releases (release threshold = max)
//   vmm      one virtual-address reservation, physical memory created and mapped in chunks (hipMemCreate / hipMemMap /
//            hipMemSetAccess): grow-only arena -- a second, larger request only creates the difference
// Every buffer is touched (memset) so that lazily backed memory would show up.
//   hipcc --offload-arch=gfx950 tools/malloc_probe.hip -o /tmp/malloc_probe && /tmp/malloc_probe [malloc|pool|vmm ...]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                                            \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      printf("  %s -> %s\n", #x, hipGetErrorString(e_));                                 \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

static const double kSizes[] = {32.0, 127.0, 180.0, 64.0, 180.0};  // GB, in the order the bench asks for them

static int touch(void* p, size_t bytes, double* gbs) {
  double t0 = now();
  CK(hipMemsetAsync(p, 0, bytes, 0));
  CK(hipDeviceSynchronize());
  *gbs = bytes / 1e9 / (now() - t0);
  return 0;
}

static int run_malloc() {
  printf("== hipMalloc / hipFree\n");
  for (double gb : kSizes) {
    void* p = nullptr;
    size_t bytes = (size_t)(gb * 1e9);
    double t0 = now();
    CK(hipMalloc(&p, bytes));
    double t1 = now(), gbs;
    if (touch(p, bytes, &gbs)) return 1;
    double t2 = now();
    CK(hipFree(p));
    printf("  %5.0f GB: hipMalloc %.3f s, memset %.0f GB/s, hipFree %.3f s\n", gb, t1 - t0, gbs, now() - t2);
  }
  return 0;
}

static int run_pool() {
  printf("== hipMallocAsync / hipFreeAsync, pool release threshold = max\n");
  hipMemPool_t pool;
  CK(hipDeviceGetDefaultMemPool(&pool, 0));
  uint64_t thr = UINT64_MAX;
  CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
  for (double gb : kSizes) {
    void* p = nullptr;
    size_t bytes = (size_t)(gb * 1e9);
    double t0 = now();
    CK(hipMallocAsync(&p, bytes, 0));
    CK(hipStreamSynchronize(0));
    double t1 = now(), gbs;
    if (touch(p, bytes, &gbs)) return 1;
    double t2 = now();
    CK(hipFreeAsync(p, 0));
    CK(hipStreamSynchronize(0));
    printf("  %5.0f GB: hipMallocAsync %.3f s, memset %.0f GB/s, hipFreeAsync %.3f s\n", gb, t1 - t0, gbs, now() - t2);
  }
  return 0;
}

static int run_vmm(double chunk_gb) {
  printf("== VMM grow-only arena, chunk %.2f GB\n", chunk_gb);
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  size_t chunk = ((size_t)(chunk_gb * 1e9) + gran - 1) / gran * gran;
  size_t va_bytes = ((size_t)280e9 + chunk - 1) / chunk * chunk;
  void* base = nullptr;
  double t0 = now();
  CK(hipMemAddressReserve(&base, va_bytes, 0, nullptr, 0));
  printf("  granularity %zu B, reserve %.0f GB of address space: %.3f s\n", gran, va_bytes / 1e9, now() - t0);
  std::vector<hipMemGenericAllocationHandle_t> handles;
  size_t mapped = 0;
  hipMemAccessDesc acc;
  memset(&acc, 0, sizeof(acc));
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (double gb : kSizes) {
    size_t bytes = (size_t)(gb * 1e9);
    double t_create = 0, t_map = 0, t_acc = 0;
    size_t grown = 0;
    while (mapped < bytes) {
      hipMemGenericAllocationHandle_t h;
      double a = now();
      CK(hipMemCreate(&h, chunk, &prop, 0));
      double b = now();
      CK(hipMemMap((char*)base + mapped, chunk, 0, h, 0));
      double c = now();
      CK(hipMemSetAccess((char*)base + mapped, chunk, &acc, 1));
      double d = now();
      t_create += b - a;
      t_map += c - b;
      t_acc += d - c;
      handles.push_back(h);
      mapped += chunk;
      grown += chunk;
    }
    double gbs;
    if (touch(base, bytes, &gbs)) return 1;
    printf("  %5.0f GB: grew by %6.1f GB: create %.3f s, map %.3f s, set-access %.3f s; memset %.0f GB/s\n", gb, grown / 1e9,
           t_create, t_map, t_acc, gbs);
  }
  double t1 = now();
  for (size_t i = 0; i < handles.size(); ++i) {
    CK(hipMemUnmap((char*)base + i * chunk, chunk));
    CK(hipMemRelease(handles[i]));
  }
  CK(hipMemAddressFree(base, va_bytes));
  printf("  unmap + release everything: %.3f s\n", now() - t1);
  return 0;
}

int main(int argc, char** argv) {
  hipFree(nullptr);
  bool all = argc < 2;
  for (int i = 1; i < argc || all; ++i) {
    const char* m = all ? "" : argv[i];
    if (all || !strcmp(m, "malloc")) run_malloc();
    if (all || !strcmp(m, "pool")) run_pool();
    if (all || !strcmp(m, "vmm")) {
      run_vmm(1.0);
      run_vmm(8.0);
    }
    if (all) break;
  }
  return 0;
}
