// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 for NC accumulators visited in runs of R
// consecutive MFMAs on the same accumulator (inline asm, VGPR form, no compiler-inserted copies).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mf(d4& c, double a, double b) {
  asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int NC, int R>
__global__ void __launch_bounds__(256, 2) chain_kernel(double* out, int iters, double a0, double b0) {
  d4 acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 64 / (NC * R); ++rep)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < R; ++r) mf(acc[c], a, b);
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
  double s = 0;
#pragma unroll
  for (int c = 0; c < NC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NC, int R>
void run(double* d, int nwg) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 5000;
  hipLaunchKernelGGL((chain_kernel<NC, R>), dim3(nwg), dim3(256), 0, 0, d, 100, 1.0, 1.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL((chain_kernel<NC, R>), dim3(nwg), dim3(256), 0, 0, d, iters, 1.0, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double n = (double)iters * 64;  // MFMAs per wave
  printf("acc=%2d run=%2d wgs=%d: %.3f ms, %.1f TF\n", NC, R, nwg, ms, n * nwg * 4 * 2048.0 / (ms * 1e-3) / 1e12);
}
int main() {
  double* d; hipMalloc(&d, 1024 * 256 * 8);
  for (int rep = 0; rep < 2; ++rep) {
    run<1, 16>(d, 512); run<16, 1>(d, 512); run<16, 2>(d, 512); run<16, 4>(d, 512); run<4, 1>(d, 512); run<2, 1>(d, 512);
    run<1, 16>(d, 256); run<16, 1>(d, 256); run<16, 4>(d, 256);
  }
  return 0;
}
