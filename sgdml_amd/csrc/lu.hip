// LU fallback of the analytic solve on gfx950.
//
// Replaces scipy.linalg.solve(K, y) as called by Analytic.solve when the Cholesky factorisation raised
// (sgdml/solvers/analytic.py:101-114; LAPACK dgesv = dgetrf + dgetrs underneath): A = -K + lam I with the
// FULL matrix (both triangles), P A = L U with partial pivoting, x = U^-1 L^-1 P y, alphas = -x.
//
// Storage: row-major, in place (unit lower L below the diagonal, U on and above), one pivot index per column.
// Right-looking blocked algorithm, panel width NB = 64:
//   per panel column c:  lu_pivot_kernel   one workgroup: argmax |A[r][c]|, r >= c (first maximum, like idamax);
//                                          interchange of rows c and p inside the panel's columns
//                        lu_column_kernel  one thread per row r > c: l = A[r][c] / A[c][c], rank-1 update of the
//                                          row's remaining panel columns
//   per panel:           lu_apply_pivots_kernel  the panel's row interchanges on all columns outside the panel
//                        lu_trsm_kernel    U12 = L11^-1 A12 (one thread per column, L11 from LDS); also writes
//                                          U12^T so that the trailing update is the NT GEMM of chol.hip
//                        gemm_nt_sub       A22 -= L21 U12  on v_mfma_f64_16x16x4_f64
// Row interchanges are cheap in row-major storage (contiguous rows); the pivot search reads a strided column
// (one 64-byte sector per row: 4 MB per column at n = 63 000, served by L2 / Infinity Cache).
// 2 n^3 / 3 of the flops are in gemm_nt_sub at K = 64; this is a fallback path, its speed is secondary to its
// semantics (the reference's LU on the host is two orders of magnitude slower).
#include <math.h>

#include "common.h"

#define LU_NB 64

// ---- pivot search + panel-local interchange -------------------------------------------------------------
__global__ void __launch_bounds__(1024) lu_pivot_kernel(double* __restrict__ A, int64_t ld, int64_t n, int64_t c,
                                                        int64_t k0, int nb, int64_t* __restrict__ piv,
                                                        int* __restrict__ info) {
  __shared__ double smax[16];
  __shared__ int64_t sidx[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double best = -1.0;
  int64_t bidx = c;
  for (int64_t r = c + tid; r < n; r += 1024) {
    const double v = fabs(A[r * ld + c]);
    if (v > best) {  // strict: the first maximum wins within a thread (rows ascend)
      best = v;
      bidx = r;
    }
  }
  // wave reduction: larger value wins, ties go to the smaller row index (LAPACK idamax)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_xor(best, off, 64);
    const int64_t oi = __shfl_xor(bidx, off, 64);
    if (ov > best || (ov == best && oi < bidx)) {
      best = ov;
      bidx = oi;
    }
  }
  if (lane == 0) {
    smax[wv] = best;
    sidx[wv] = bidx;
  }
  __syncthreads();
  if (wv == 0) {
    best = lane < 16 ? smax[lane] : -1.0;
    bidx = lane < 16 ? sidx[lane] : c;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      const double ov = __shfl_xor(best, off, 64);
      const int64_t oi = __shfl_xor(bidx, off, 64);
      if (ov > best || (ov == best && oi < bidx)) {
        best = ov;
        bidx = oi;
      }
    }
    if (lane == 0) {
      sidx[0] = bidx;
      smax[0] = best;
      piv[c] = bidx;
      if (!(best > 0.0)) atomicCAS(info, 0, (int)(c + 1));  // exactly singular (or NaN): dgetrf's info > 0
    }
  }
  __syncthreads();
  const int64_t p = sidx[0];
  if (p != c && tid < nb) {
    const double a = A[c * ld + k0 + tid], b = A[p * ld + k0 + tid];
    A[c * ld + k0 + tid] = b;
    A[p * ld + k0 + tid] = a;
  }
}

// ---- multipliers of column c and rank-1 update of the rest of the panel ---------------------------------
__global__ void __launch_bounds__(256) lu_column_kernel(double* __restrict__ A, int64_t ld, int64_t n, int64_t c,
                                                        int64_t k0, int nb) {
  __shared__ double urow[LU_NB];
  const int j = (int)(c - k0);
  if (threadIdx.x < nb) urow[threadIdx.x] = A[c * ld + k0 + threadIdx.x];
  __syncthreads();
  const double pivot = urow[j];
  if (pivot == 0.0) return;
  const int64_t r = c + 1 + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  double* row = A + r * ld + k0;
  const double l = row[j] / pivot;
  row[j] = l;
  for (int jj = j + 1; jj < nb; ++jj) row[jj] -= l * urow[jj];
}

// ---- the panel's interchanges on the columns outside the panel -------------------------------------------
__global__ void __launch_bounds__(256) lu_apply_pivots_kernel(double* __restrict__ A, int64_t ld, int64_t n,
                                                              int64_t k0, int nb, const int64_t* __restrict__ piv) {
  int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= k0) col += nb;  // skip the panel's own columns
  if (col >= n) return;
  for (int j = 0; j < nb; ++j) {
    const int64_t c = k0 + j, p = piv[c];
    if (p != c) {
      const double a = A[c * ld + col], b = A[p * ld + col];
      A[c * ld + col] = b;
      A[p * ld + col] = a;
    }
  }
}

// ---- U12 = L11^-1 A12 (unit lower L11, nb x nb), one thread per column; U12^T to scratch ------------------
__global__ void __launch_bounds__(256) lu_trsm_kernel(double* __restrict__ A, int64_t ld, int64_t n, int64_t k0,
                                                      int nb, double* __restrict__ UT) {
  __shared__ double L11[LU_NB * LU_NB];
  for (int e = threadIdx.x; e < LU_NB * LU_NB; e += 256) {
    const int r = e / LU_NB, q = e % LU_NB;
    L11[e] = (r < nb && q < r) ? A[(k0 + r) * ld + k0 + q] : 0.0;
  }
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;  // column index inside A12
  const int64_t col = k0 + nb + t;
  if (col >= n) return;
  double x[LU_NB];
#pragma unroll
  for (int r = 0; r < LU_NB; ++r) x[r] = (r < nb) ? A[(k0 + r) * ld + col] : 0.0;
#pragma unroll
  for (int r = 1; r < LU_NB; ++r) {
    double s = x[r];
#pragma unroll
    for (int q = 0; q < r; ++q) s -= L11[r * LU_NB + q] * x[q];
    x[r] = s;
  }
#pragma unroll
  for (int r = 0; r < LU_NB; ++r)
    if (r < nb) {
      A[(k0 + r) * ld + col] = x[r];
      UT[t * LU_NB + r] = x[r];
    }
}

// ---- triangular solves with the factors (dgetrs) -----------------------------------------------------------
// forward, unit lower: per 64-block solve the diagonal block in one wavefront, then b[r] -= L[r, J] z_J below
__global__ void __launch_bounds__(256) lu_fwd_kernel(const double* __restrict__ A, int64_t ld, int64_t n, int64_t c0,
                                                     int w, double* __restrict__ b, double* __restrict__ z_out) {
  __shared__ double Ls[64 * 65];
  __shared__ double z[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, q = e & 63;
    Ls[r * 65 + q] = (r < w && q < r) ? A[(c0 + r) * ld + c0 + q] : 0.0;
  }
  __syncthreads();
  if (wave == 0) {
    double bi = lane < w ? b[c0 + lane] : 0.0;
    double sol = 0.0;
    for (int q = 0; q < 64; ++q) {  // unit diagonal: z_q = b_q
      const double zq = __shfl(bi, q, 64);
      if (lane == q) sol = zq;
      bi -= Ls[lane * 65 + q] * zq;  // rows r > q (entries with q >= r are zero)
    }
    z[lane] = sol;
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid < w) z_out[c0 + tid] = z[tid];  // separate vector: other workgroups still read b[c0..]
  const double zl = lane < w ? z[lane] : 0.0;
  for (int64_t r = c0 + w + (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
    double v = lane < w ? A[r * ld + c0 + lane] * zl : 0.0;
    v = wave_sum(v);
    if (lane == 0) b[r] -= v;
  }
}

// backward, upper U: from the bottom, solve the diagonal block, then z[r] -= U[r, I] x_I for the rows above
__global__ void __launch_bounds__(256) lu_bwd_kernel(const double* __restrict__ A, int64_t ld, int64_t c0, int w,
                                                     double* __restrict__ b, double* __restrict__ x_out) {
  __shared__ double Us[64 * 65];
  __shared__ double x[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, q = e & 63;
    double v = (r == q) ? 1.0 : 0.0;
    if (r < w && q < w && q >= r) v = A[(c0 + r) * ld + c0 + q];
    Us[r * 65 + q] = v;
  }
  __syncthreads();
  if (wave == 0) {
    double bi = lane < w ? b[c0 + lane] : 0.0;
    double sol = 0.0;
    for (int q = 63; q >= 0; --q) {
      const double t = bi / Us[lane * 65 + lane];  // meaningful on lane q
      const double xq = __shfl(t, q, 64);
      if (lane == q) sol = xq;
      bi -= Us[lane * 65 + q] * xq;  // rows r < q
    }
    x[lane] = sol;
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid < w) x_out[c0 + tid] = x[tid];
  const double xl = lane < w ? x[lane] : 0.0;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < c0; r += (int64_t)gridDim.x * 4) {
    double v = lane < w ? A[r * ld + c0 + lane] * xl : 0.0;
    v = wave_sum(v);
    if (lane == 0) b[r] -= v;
  }
}

// A <- -A + lam I on the whole matrix
__global__ void __launch_bounds__(256) negate_shift_full_kernel(double* __restrict__ A, int64_t n, int64_t ld,
                                                                double lam) {
  const int64_t r = blockIdx.x;
  double* row = A + r * ld;
  for (int64_t c = threadIdx.x; c < n; c += 256) {
    double v = -row[c];
    if (c == r) v += lam;
    row[c] = v;
  }
}

int lu_factor_device(gdml_ctx* ctx, double* A, int64_t n, int64_t ld, int64_t* d_piv, int* info_out) {
  hipStream_t st = ctx->stream;
  HIP_CHECK(ctx, hipMemsetAsync(ctx->d_info, 0, sizeof(int), st));
  double* UT = nullptr;
  GDML_TRY(ctx_slot(ctx, 6, n * LU_NB * 8, &UT));
  for (int64_t k0 = 0; k0 < n; k0 += LU_NB) {
    const int nb = (int)((n - k0 < LU_NB) ? n - k0 : LU_NB);
    for (int j = 0; j < nb; ++j) {
      const int64_t c = k0 + j;
      hipLaunchKernelGGL(lu_pivot_kernel, dim3(1), dim3(1024), 0, st, A, ld, n, c, k0, nb, d_piv, ctx->d_info);
      const int64_t below = n - c - 1;
      if (below > 0)
        hipLaunchKernelGGL(lu_column_kernel, dim3((unsigned)ceil_div(below, 256)), dim3(256), 0, st, A, ld, n, c, k0,
                           nb);
      ctx->launch_counter += 2;
    }
    if (n - nb > 0)
      hipLaunchKernelGGL(lu_apply_pivots_kernel, dim3((unsigned)ceil_div(n - nb, 256)), dim3(256), 0, st, A, ld, n, k0,
                         nb, d_piv);
    const int64_t right = n - k0 - nb;
    if (right > 0) {
      hipLaunchKernelGGL(lu_trsm_kernel, dim3((unsigned)ceil_div(right, 256)), dim3(256), 0, st, A, ld, n, k0, nb, UT);
      ctx->launch_counter += 2;
      // A22 -= L21 U12:  C[M x N] -= A[M x K] B[N x K]^T with B = U12^T
      GDML_TRY(launch_gemm_nt_sub(ctx, st, A + (k0 + nb) * ld + k0, ld, UT, LU_NB, A + (k0 + nb) * ld + k0 + nb, ld,
                                  right, right, nb, 0));
    }
    HIP_CHECK(ctx, hipGetLastError());
  }
  int info = 0;
  HIP_CHECK(ctx, hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_CHECK(ctx, hipStreamSynchronize(st));
  if (info_out) *info_out = info;
  return GDML_OK;
}

extern "C" int gdml_lu_solve(gdml_ctx* ctx, double lam, const double* y, int64_t n, double* alphas_out, int* info) {
  if (!ctx || !y || !alphas_out) return GDML_ERR_INVALID;
  if (info) *info = 0;
  if (!ctx->K || ctx->K_rows != ctx->K_cols)
    return gdml_fail(ctx, GDML_ERR_STATE, "gdml_lu_solve: assemble the full square K first (gdml_assemble_K, all columns)");
  if (ctx->K_factored || ctx->K_is_A || ctx->K_destroyed)
    return gdml_fail(ctx, GDML_ERR_STATE,
                     "gdml_lu_solve: needs the full un-negated matrix of gdml_assemble_K (a Cholesky attempt destroys it; "
                     "gdml_assemble_A writes only the lower blocks): assemble again");
  if (n != ctx->K_rows) return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_lu_solve: n mismatch");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int64_t ld = ctx->K_ld;
  void* buf = nullptr;
  GDML_TRY(ctx_alloc(ctx, &buf, 4 * n * 8));
  int64_t* d_piv = (int64_t*)buf;
  double* d_b = (double*)buf + n;
  double* d_z = d_b + n;
  double* d_x = d_z + n;
  int rc = GDML_OK, inf = 0;
  auto body = [&]() -> int {
    phase_begin(ctx);
    hipLaunchKernelGGL(negate_shift_full_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, ctx->K, n, ld, lam);
    ctx->K_destroyed = true;  // whatever happens below, the buffer no longer holds K
    GDML_TRY(lu_factor_device(ctx, ctx->K, n, ld, d_piv, &inf));
    GDML_TRY(phase_end(ctx, "factor"));
    if (inf != 0) return GDML_OK;
    // dgetrs: x = U^-1 L^-1 P y
    std::vector<int64_t> piv((size_t)n);
    HIP_CHECK(ctx, hipMemcpyAsync(piv.data(), d_piv, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<double> b(y, y + n);
    for (int64_t c = 0; c < n; ++c) {
      const int64_t p = piv[(size_t)c];
      if (p != c) std::swap(b[(size_t)c], b[(size_t)p]);
    }
    HIP_CHECK(ctx, hipMemcpyAsync(d_b, b.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // b is a local vector
    phase_begin(ctx);
    for (int64_t c0 = 0; c0 < n; c0 += 64) {
      const int w = (int)((n - c0 < 64) ? n - c0 : 64);
      int grid = (int)((n - c0 - w + 3) / 4);
      grid = grid < 1 ? 1 : (grid > 1024 ? 1024 : grid);
      hipLaunchKernelGGL(lu_fwd_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->K, ld, n, c0, w, d_b, d_z);
    }
    for (int64_t c0 = ((n - 1) / 64) * 64; c0 >= 0; c0 -= 64) {
      const int w = (int)((n - c0 < 64) ? n - c0 : 64);
      int grid = (int)((c0 + 3) / 4);
      grid = grid < 1 ? 1 : (grid > 1024 ? 1024 : grid);
      hipLaunchKernelGGL(lu_bwd_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->K, ld, c0, w, d_z, d_x);
    }
    HIP_CHECK(ctx, hipGetLastError());
    GDML_TRY(phase_end(ctx, "solve"));
    HIP_CHECK(ctx, hipMemcpyAsync(alphas_out, d_x, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < n; ++i) alphas_out[i] = -alphas_out[i];  // analytic.py:112: alphas = -x
    return GDML_OK;
  };
  rc = body();
  ctx->K_factored = false;  // an LU, not a Cholesky factor: gdml_chol_solve must not use it
  ctx->K_destroyed = true;  // the buffer stays allocated for reuse, but holds nothing a later call may read
  int rc2 = ctx_free(ctx, buf);
  if (rc != GDML_OK) return rc;
  if (rc2 != GDML_OK) return rc2;
  if (info) *info = inf;
  if (inf != 0)
    return gdml_fail(ctx, GDML_ERR_NOT_PD, "Matrix is singular (zero pivot in column %d of the LU factorisation)", inf);
  return GDML_OK;
}
