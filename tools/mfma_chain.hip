// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 as a function of the number of independent
// accumulator chains per wavefront (1 wave per SIMD, 256 workgroups of 256 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NC>
__global__ void __launch_bounds__(256, 1) chain_kernel(double* out, int iters, double a0, double b0) {
  d4 acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16 / NC; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < NC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NC>
void run(double* d, int nwg) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL(chain_kernel<NC>, dim3(nwg), dim3(256), 0, 0, d, 100, 1.0, 1.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(chain_kernel<NC>, dim3(nwg), dim3(256), 0, 0, d, iters, 1.0, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double n = (double)iters * 16;  // MFMAs per wave
  printf("chains=%2d wgs=%d: %.3f ms, %.1f ns per MFMA per wave, %.1f TF\n", NC, nwg, ms, ms * 1e6 / n,
         n * nwg * 4 * 2048.0 / (ms * 1e-3) / 1e12);
}
int main() {
  double* d; hipMalloc(&d, 1024 * 256 * 8);
  run<1>(d, 256); run<2>(d, 256); run<4>(d, 256); run<8>(d, 256); run<16>(d, 256);
  run<1>(d, 512); run<2>(d, 512);
  return 0;
}
