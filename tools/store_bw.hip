// Write-bandwidth probe for the kernel-matrix store pattern on MI355X (63000^2 fp64).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef double d2 __attribute__((ext_vector_type(2)));
// mode 0: block (j, iy) plain; mode 1: XCD remap (consecutive j on one XCD); mode 2: WG loops over 8 consecutive j
__global__ void seg_kernel(double* K, int64_t ld, int N3, int M, int ichunk, int mode, int nj) {
  int64_t b = blockIdx.x;
  int64_t j, iy; int jrep = 1;
  if (mode == 0) { j = b % nj; iy = b / nj; }
  else if (mode == 1) { int64_t x = b & 7, l = b >> 3; int64_t per = nj / 8; j = x * per + (l % per); iy = l / per; }
  else { int64_t x = b & 7, l = b >> 3; int64_t groups = nj / 8; int64_t per = groups / 8; j = (x * per + (l % per)) * 8; iy = l / per; jrep = 8; }
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int64_t i0 = iy * ichunk;
  for (int64_t i = i0; i < i0 + ichunk && i < M; ++i)
    for (int jj = 0; jj < jrep; ++jj)
      for (int r = wave; r < N3; r += nw)
        if (lane < N3) K[(i * N3 + r) * ld + (j + jj) * N3 + lane] = (double)lane;
}
// non-temporal variants (streaming stores that should not allocate in L2 / MALL)
__global__ void seg_nt_kernel(double* K, int64_t ld, int N3, int M, int ichunk, int nj) {
  int64_t b = blockIdx.x;
  int64_t x = b & 7, l = b >> 3; int64_t groups = nj / 8; int64_t per = groups / 8; int64_t j = (x * per + (l % per)) * 8; int64_t iy = l / per;
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int64_t i0 = iy * ichunk;
  for (int64_t i = i0; i < i0 + ichunk && i < M; ++i)
    for (int jj = 0; jj < 8; ++jj)
      for (int r = wave; r < N3; r += nw)
        if (lane < N3) __builtin_nontemporal_store((double)lane, &K[(i * N3 + r) * ld + (j + jj) * N3 + lane]);
}
__global__ void row8_nt_kernel(double* K, int64_t ld, int64_t n) {
  int64_t r = blockIdx.x; double* row = K + r * ld;
  for (int64_t c = threadIdx.x; c < n; c += blockDim.x) __builtin_nontemporal_store(1.0, &row[c]);
}
// mode 3: one wavefront per row point i (4 per workgroup), walks j in groups of J; inside a group the loop
// is row-outer / j-inner, so J adjacent 504-byte segments of one matrix row are written back to back
__global__ void seg_rowmajor_kernel(double* K, int64_t ld, int N3, int M, int J, int jchunk) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t b = blockIdx.x; int64_t x = b & 7, l = b >> 3;
  int64_t nchunks = (M + jchunk - 1) / jchunk;      // j chunks
  int64_t per = nchunks / 8 > 0 ? nchunks / 8 : 1;  // chunks per XCD
  int64_t jc = x * per + (l % per), ig = l / per;
  int64_t i = ig * 4 + wave;
  if (i >= M || jc >= nchunks) return;
  for (int64_t j0 = jc * jchunk; j0 < (jc + 1) * jchunk && j0 < M; j0 += J)
    for (int r = 0; r < N3; ++r)
      for (int jj = 0; jj < J; ++jj)
        if (lane < N3 && j0 + jj < M) K[(i * N3 + r) * ld + (j0 + jj) * N3 + lane] = (double)lane;
}
// mode 4: workgroup of W wavefronts for one row point i; wavefront w owns block column j0 + w and all
// wavefronts walk the rows together, so W adjacent segments of one matrix row are written at the same time
__global__ void seg_wavecols_kernel(double* K, int64_t ld, int N3, int M, int jchunk, int sync) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
  int64_t b = blockIdx.x; int64_t x = b & 7, l = b >> 3;
  int64_t nchunks = (M + jchunk - 1) / jchunk;
  int64_t per = nchunks / 8 > 0 ? nchunks / 8 : 1;
  int64_t jc = x * per + (l % per), i = l / per;
  if (i >= M || jc >= nchunks) return;
  for (int64_t j0 = jc * jchunk; j0 < (jc + 1) * jchunk && j0 < M; j0 += W) {
    for (int r = 0; r < N3; ++r)
      if (lane < N3 && j0 + wave < M) K[(i * N3 + r) * ld + (j0 + wave) * N3 + lane] = (double)lane;
    if (sync) __syncthreads();
  }
}
// pairs of block columns, 16-byte stores, XCD remap
__global__ void seg2_kernel(double* K, int64_t ld, int N3, int M, int ichunk, int nj2) {
  int64_t b = blockIdx.x; int64_t x = b & 7, l = b >> 3; int64_t per = nj2 / 8;
  int64_t j2 = x * per + (l % per), iy = l / per;
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int64_t i0 = iy * ichunk; d2 v = {1.0, 2.0};
  for (int64_t i = i0; i < i0 + ichunk && i < M; ++i)
    for (int r = wave; r < N3; r += nw)
      if (lane < N3) *reinterpret_cast<d2*>(&K[(i * N3 + r) * ld + j2 * 2 * N3 + 2 * lane]) = v;
}
__global__ void row8_kernel(double* K, int64_t ld, int64_t n) {
  int64_t r = blockIdx.x; double* row = K + r * ld;
  for (int64_t c = threadIdx.x; c < n; c += blockDim.x) row[c] = 1.0;
}
// mode 5 (round 2): 64-column STRIPS.  Wavefront = 64 (or 128 with 16-byte stores) consecutive global columns, walks
// ichunk row points x N3 rows; W wavefronts of a workgroup own adjacent strips (W*512 contiguous bytes per matrix row),
// optionally in lockstep per row point; xcd = 1 places consecutive workgroups of a row chunk on one XCD.
template <int VW>
__global__ void strip_kernel(double* K, int64_t ld, int64_t n, int N3, int M, int ichunk, int sync, int xcd, int64_t nsg) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
  int64_t b = blockIdx.x, sg, iy;
  if (xcd) { int64_t x = b & 7, l = b >> 3; int64_t per = (nsg + 7) / 8; sg = x * per + (l % per); iy = l / per; if (sg >= nsg) return; }
  else { sg = b % nsg; iy = b / nsg; }
  int64_t c = ((sg * W + wave) * 64 + lane) * VW;
  int64_t i0 = iy * ichunk;
  for (int64_t i = i0; i < i0 + ichunk && i < M; ++i) {
    double* dst = K + (i * N3) * ld + c;
    if (c < n)
      for (int r = 0; r < N3; ++r, dst += ld) {
        if (VW == 1) *dst = (double)lane; else { d2 v = {1.0, 2.0}; *reinterpret_cast<d2*>(dst) = v; }
      }
    if (sync) __syncthreads();
  }
}
#define T(name, call) do { hipEventRecord(a); call; hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); printf("%-28s: %.2f ms  %.0f GB/s\n", name, ms, gb / ms * 1e3); } while (0)
int main() {
  int N3 = 63, M = 1000; int64_t n = (int64_t)N3 * M, ld = (n + 15) / 16 * 16;
  double* K; hipMalloc(&K, n * ld * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float ms;
  double gb = 8.0 * n * n / 1e9;
  int ic = 32; int ny = (M + ic - 1) / ic;
  for (int rep = 0; rep < 2; ++rep) {
    T("seg plain", (seg_kernel<<<dim3(M * ny), 448>>>(K, ld, N3, M, ic, 0, M)));
    T("seg xcd-remap", (seg_kernel<<<dim3(M * ny), 448>>>(K, ld, N3, M, ic, 1, M)));
    T("seg xcd-remap 8j/WG", (seg_kernel<<<dim3(M / 8 * ny), 448>>>(K, ld, N3, M, ic, 2, M)));
    T("seg pairs 16B xcd-remap", (seg2_kernel<<<dim3(M / 2 * ny), 448>>>(K, ld, N3, M, ic, M / 2)));
    T("rows 8B", (row8_kernel<<<(unsigned)n, 256>>>(K, ld, n)));
    T("rows 8B nontemporal", (row8_nt_kernel<<<(unsigned)n, 256>>>(K, ld, n)));
    T("seg 8j/WG nontemporal", (seg_nt_kernel<<<dim3(M / 8 * ny), 448>>>(K, ld, N3, M, ic, M)));
    for (int W = 4; W <= 16; W *= 2) for (int sy = 0; sy < 2; ++sy) {
      char nm[64]; int jchunk = 64; int64_t nch = (M + jchunk - 1) / jchunk;
      snprintf(nm, 64, "WG/i, wave/j, W=%d sync=%d", W, sy);
      T(nm, (seg_wavecols_kernel<<<dim3((unsigned)(nch * M)), 64 * W>>>(K, ld, N3, M, jchunk, sy)));
    }
    for (int vw = 1; vw <= 2; ++vw) for (int W = 1; W <= 16; W *= 2) for (int sy = 0; sy < 2; ++sy) for (int xc = 0; xc < 2; ++xc) {
      if (W == 1 && sy) continue;
      if (rep == 0) continue;
      char nm[64]; int64_t nsg = (n + 64 * vw * W - 1) / (64 * vw * W); int64_t per = (nsg + 7) / 8;
      int64_t nb = xc ? per * 8 * ny : nsg * ny;
      snprintf(nm, 64, "strip %dB W=%d sync=%d xcd=%d", 8 * vw, W, sy, xc);
      if (vw == 1) T(nm, (strip_kernel<1><<<dim3((unsigned)nb), 64 * W>>>(K, ld, n, N3, M, ic, sy, xc, nsg)));
      else T(nm, (strip_kernel<2><<<dim3((unsigned)nb), 64 * W>>>(K, ld, n, N3, M, ic, sy, xc, nsg)));
    }
    for (int J = 1; J <= 8; J *= 2) {
      char nm[64]; int jchunk = 40; int64_t nch = (M + jchunk - 1) / jchunk;
      snprintf(nm, 64, "wave/i row-outer J=%d", J);
      T(nm, (seg_rowmajor_kernel<<<dim3((unsigned)(nch * (M / 4))), 256>>>(K, ld, N3, M, J, jchunk)));
    }
  }
  return 0;
}
