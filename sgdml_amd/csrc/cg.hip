// Nystroem-preconditioned conjugate gradients on the device.
//
// Replaces sgdml/solvers/iterative.py: _nystroem_cholesky_factor (:208-351), _cho_factor_stable
// (:414-471), the preconditioner operator (:83-142), the kernel operator (:144-206, predict.hip)
// and scipy.sparse.linalg.cg as called at :740-752.
//
// Device layout: the (n+m) x m matrix of gdml_assemble_K(GDML_COLS_INDEX, alloc_extra_rows = m):
// rows [0,n) = K_nm (un-negated), rows [n,n+m) = work area for the m x m blocks (K_mm, then
// K_nm^T K_nm + lam I), exactly like the reference's K_nmm (iterative.py:237-253).  After the
// factorisation rows [0,n) hold X = (L^-1 K_mn)^T, the preconditioner is P v = (X X^T v - v)/lam.
#include <math.h>

#include "common.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// dst rows n..n+m <- -K[idx[q], :]           (iterative.py:250)
__global__ void __launch_bounds__(256) gather_neg_rows_kernel(double* __restrict__ K, int64_t ld,
                                                              const int64_t* __restrict__ idx,
                                                              int64_t m, int64_t n) {
  const int64_t q = blockIdx.x;
  const double* src = K + idx[q] * ld;
  double* dst = K + (n + q) * ld;
  for (int64_t c = threadIdx.x; c < m; c += 256) dst[c] = -src[c];
}

__global__ void __launch_bounds__(256) add_diag_kernel(double* __restrict__ A, int64_t ld, int64_t m,
                                                       double v) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < m) A[t * ld + t] += v;
}

// ------------------------------------------------------------------------------------------
// C[m x m] (lower tiles) = X^T X, X is n x m row-major (the contraction runs over the ROWS of X: K_nm^T K_nm of the
// Nystroem factor, iterative.py:293, and the Gram matrices of the CholeskyQR branch).  fp64 MFMA, 128 x 128 tiles, 16
// rows of X per k-tile, double-buffered LDS.
//
// Both operand tiles of a k-tile are 16 x 128 blocks of X with contiguous rows, so everything moves in 16-byte pieces:
// one check-free global_load_dwordx4 per chunk from a wave-uniform base (tile corner + k offset in SGPRs) plus the
// thread's constant byte offset, one ds_write_b128 per chunk into an unpadded [k][128] image, and one ds_read_b128 per
// PAIR of MFMA operand blocks: the 16 x 16 operand blocks of a wave's 64 tile columns are taken as column pairs
// (block 2 h + e, lane index x  <->  tile column 32 h + 2 x + e), so that a lane's two operands of blocks 2h, 2h + 1 are
// adjacent in LDS and the 8 lanes an LDS cycle serves read 128 contiguous bytes (no bank conflict, no padding).  The
// relabelling only changes which tile column an accumulator belongs to (undone in the epilogue).  Tiles that reach past
// m and the last, partial k-tile take guarded loads.  Round 3's kernel (8-byte bounds-checked loads inside the k loop,
// 8-byte LDS layout, 222 VGPRs) ran at 0.73 of the fp64-MFMA peak.
// ------------------------------------------------------------------------------------------
#define TT 128
#define TBK 16

typedef const __attribute__((address_space(1))) char* tn_gcptr;
typedef const __attribute__((address_space(1))) d2* tn_gd2ptr;

template <bool FULL>
__device__ __forceinline__ void tn_load(const double* __restrict__ X, int64_t ld, int64_t n, int64_t m, int64_t k0, int64_t c0,
                                        int tid, const unsigned (&off)[4], d2 (&r)[4]) {
  if (FULL) {  // interior tile, whole k-tile: wave-uniform base + 32-bit lane offsets
    const uint64_t p = reinterpret_cast<uint64_t>(X + k0 * ld + c0);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    tn_gcptr b = (tn_gcptr)(((uint64_t)hi << 32) | lo);
#pragma unroll
    for (int s = 0; s < 4; ++s) r[s] = *(tn_gd2ptr)(b + off[s]);
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int cidx = tid + 256 * s;
      const int kr = cidx >> 6, cc = (cidx & 63) * 2;
      const int64_t gk = k0 + kr, gc = c0 + cc;
      d2 v = {0.0, 0.0};
      if (gk < n) {
        if (gc < m) v.x = X[gk * ld + gc];
        if (gc + 1 < m) v.y = X[gk * ld + gc + 1];
      }
      r[s] = v;
    }
  }
}
__device__ __forceinline__ void tn_store(double* __restrict__ S, int tid, const d2 (&r)[4]) {
  d2* S2 = reinterpret_cast<d2*>(S);
#pragma unroll
  for (int s = 0; s < 4; ++s) S2[tid + 256 * s] = r[s];  // chunk (kr, cc) sits at doubles kr * 128 + cc = 2 * cidx
}

template <bool FULL>
__device__ __forceinline__ void syrk_tn_tile(const double* __restrict__ X, int64_t ldx, int64_t n, int64_t m,
                                             double* __restrict__ C, int64_t ldc, int64_t row0, int64_t col0,
                                             double (*lds)[2][TBK * TT], int accumulate) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lk = lane >> 4;
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  unsigned off[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int cidx = tid + 256 * s;
    off[s] = (unsigned)(((int64_t)(cidx >> 6) * ldx + (cidx & 63) * 2) * 8);
  }
  const int64_t nk_full = FULL ? n / TBK : 0;          // k-tiles loaded without checks
  const int64_t nk = (n + TBK - 1) / TBK;
  d2 ra[4], rb[4];
  if (nk_full > 0) {
    tn_load<true>(X, ldx, n, m, 0, row0, tid, off, ra);
    tn_load<true>(X, ldx, n, m, 0, col0, tid, off, rb);
  } else {
    tn_load<false>(X, ldx, n, m, 0, row0, tid, off, ra);
    tn_load<false>(X, ldx, n, m, 0, col0, tid, off, rb);
  }
  tn_store(lds[0][0], tid, ra);
  tn_store(lds[0][1], tid, rb);
  __syncthreads();
  for (int64_t kt = 0; kt < nk; ++kt) {
    const int cur = (int)(kt & 1);
    if (kt + 1 < nk) {
      if (kt + 1 < nk_full) {
        tn_load<true>(X, ldx, n, m, (kt + 1) * TBK, row0, tid, off, ra);
        tn_load<true>(X, ldx, n, m, (kt + 1) * TBK, col0, tid, off, rb);
      } else {
        tn_load<false>(X, ldx, n, m, (kt + 1) * TBK, row0, tid, off, ra);
        tn_load<false>(X, ldx, n, m, (kt + 1) * TBK, col0, tid, off, rb);
      }
    }
    // operand pairs: d2 index of (k, column 32 h + 2 li) inside the wave's 64 columns
    const d2* Ap = reinterpret_cast<const d2*>(lds[cur][0]) + lk * (TT / 2) + wm * 32 + li;
    const d2* Bp = reinterpret_cast<const d2*>(lds[cur][1]) + lk * (TT / 2) + wn * 32 + li;
#pragma unroll
    for (int ks = 0; ks < TBK; ks += 4) {
      const d2 a01 = Ap[ks * (TT / 2)], a23 = Ap[ks * (TT / 2) + 16];
      const d2 b01 = Bp[ks * (TT / 2)], b23 = Bp[ks * (TT / 2) + 16];
      const double a[4] = {a01.x, a01.y, a23.x, a23.y};
      const double bb[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
      if (ks == 8 && kt + 1 < nk) {  // the next tile's loads were issued 48 MFMAs ago
        tn_store(lds[cur ^ 1][0], tid, ra);
        tn_store(lds[cur ^ 1][1], tid, rb);
      }
    }
    __syncthreads();
  }
  // epilogue: accumulator (i, j), register r of lane (li, lk) is tile element
  //   row = 64 wm + 32 (i >> 1) + 2 (lk + 4 r) + (i & 1),  column = 64 wn + 32 (j >> 1) + 2 li + (j & 1)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t gc = col0 + wn * 64 + 32 * (j >> 1) + 2 * li + (j & 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t gr = row0 + wm * 64 + 32 * (i >> 1) + 2 * (lk + 4 * r) + (i & 1);
        if (FULL || (gr < m && gc < m)) C[gr * ldc + gc] = accumulate ? C[gr * ldc + gc] + acc[i][j][r] : acc[i][j][r];
      }
    }
}

__global__ void __launch_bounds__(256, 2) syrk_tn_kernel(const double* __restrict__ X, int64_t ldx,
                                                         int64_t n, int64_t m, double* __restrict__ C,
                                                         int64_t ldc, int tiles, int accumulate) {
  __shared__ __attribute__((aligned(16))) double lds[2][2][TBK * TT];
  // linear block id -> lower-triangular tile (ti >= tj)
  const int64_t b = blockIdx.x;
  int64_t ti = (int64_t)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > b) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
  const int64_t tj = b - ti * (ti + 1) / 2;
  if (ti >= tiles) return;
  const int64_t row0 = ti * TT, col0 = tj * TT;
  const bool aligned = ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && (ldx % 2 == 0);
  if (aligned && row0 + TT <= m && col0 + TT <= m && 16 * ldx * 8 < ((int64_t)1 << 32))
    syrk_tn_tile<true>(X, ldx, n, m, C, ldc, row0, col0, lds, accumulate);
  else
    syrk_tn_tile<false>(X, ldx, n, m, C, ldc, row0, col0, lds, accumulate);
}

// The same Gram tiles with the contraction cut into `splits` row ranges (round 6): block = (split, tile); split s sums rows
// [s rows_per, (s + 1) rows_per) into its own m x ldc image C + s c_stride, gram_sum_parts_kernel adds the images in a fixed order.
// A Gram tile's k loop is as long as the factor is tall (252 000 ... 900 000 rows), so a launch whose tile count is a few
// rounds of the chip's 512 slots pays a whole tile time for its last, partly filled round (configs[4]: 1653 tiles = 3.2 rounds
// -> 4; configs[2]: 666 tiles = 1.3 -> 2); cut in S the same work is S times as many units of 1 / S the length.
__global__ void __launch_bounds__(256, 2) syrk_tn_split_kernel(const double* __restrict__ X, int64_t ldx, int64_t n, int64_t m,
                                                               double* __restrict__ C, int64_t ldc, int tiles, int64_t ntile,
                                                               int64_t rows_per, int64_t c_stride) {
  __shared__ __attribute__((aligned(16))) double lds[2][2][TBK * TT];
  const int64_t split = (int64_t)blockIdx.x / ntile, b = (int64_t)blockIdx.x - split * ntile;
  int64_t ti = (int64_t)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > b) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
  const int64_t tj = b - ti * (ti + 1) / 2;
  if (ti >= tiles) return;
  const int64_t r0 = split * rows_per;
  int64_t nn = n - r0;
  if (nn > rows_per) nn = rows_per;
  if (nn < 0) nn = 0;  // (an empty range still writes its zero tile)
  X += r0 * ldx;
  C += split * c_stride;
  const int64_t row0 = ti * TT, col0 = tj * TT;
  const bool aligned = ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && (ldx % 2 == 0);
  if (aligned && row0 + TT <= m && col0 + TT <= m && 16 * ldx * 8 < ((int64_t)1 << 32))
    syrk_tn_tile<true>(X, ldx, nn, m, C, ldc, row0, col0, lds, 0);
  else
    syrk_tn_tile<false>(X, ldx, nn, m, C, ldc, row0, col0, lds, 0);
}

__global__ void __launch_bounds__(256) gram_sum_parts_kernel(double* __restrict__ out, const double* __restrict__ parts,
                                                             int splits, int64_t stride, int64_t count, int64_t ldc) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= count) return;
  const int64_t r = e / ldc, c = e - r * ldc;
  if (c / TT > r / TT) return;  // tiles above the diagonal: not written by the Gram kernel, left alone like it leaves them
  double s = parts[e];
  for (int k = 1; k < splits; ++k) s += parts[k * stride + e];
  out[e] = s;
}

// C (m x m, lower tiles) = X^T X over the n rows of X; option nys.syrk_split (default 1): cut along the rows when the tile count
// is only a few rounds of the chip
static int gram_tn(gdml_ctx* ctx, const double* X, int64_t ldx, int64_t n, int64_t m, double* Cout, int64_t ldc) {
  const int tiles = (int)((m + TT - 1) / TT);
  const int64_t T = (int64_t)tiles * (tiles + 1) / 2;
  int S = 1;
  if (ctx_opt_i(ctx, "nys.syrk_split", 1) != 0 && T < 8192) {
    S = (int)((8192 + T - 1) / T);
    if (S > 8) S = 8;
    while (S > 1 && n / S < 8192) --S;                             // every range long enough to be worth a tile's prologue
    while (S > 1 && (int64_t)S * m * ldc * 8 > ((int64_t)2 << 30)) --S;  // at most 2 GiB of partial images (measured: no gain beyond)
  }
  void* parts = nullptr;
  // (no room for the images: the one-pass form -- in a sharded build a rank that failed here would leave its peers in the
  //  all-reduce that follows)
  if (S > 1 && ctx_alloc(ctx, &parts, (int64_t)S * m * ldc * 8) != GDML_OK) {
    parts = nullptr;
    S = 1;
  }
  if (S <= 1) {
    hipLaunchKernelGGL(syrk_tn_kernel, dim3((unsigned)T), dim3(256), 0, ctx->stream, X, ldx, n, m, Cout, ldc, tiles, 0);
    HIP_CHECK(ctx, hipGetLastError());
    return GDML_OK;
  }
  const int64_t rows_per = ((n + S - 1) / S + TBK - 1) / TBK * TBK;
  hipLaunchKernelGGL(syrk_tn_split_kernel, dim3((unsigned)(T * S)), dim3(256), 0, ctx->stream, X, ldx, n, m, (double*)parts, ldc,
                     tiles, T, rows_per, m * ldc);
  hipLaunchKernelGGL(gram_sum_parts_kernel, dim3((unsigned)ceil_div(m * ldc, 256)), dim3(256), 0, ctx->stream, Cout,
                     (const double*)parts, S, m * ldc, m * ldc, ldc);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the images go back before the next allocation needs them
  int rc = ctx_free(ctx, parts);
  if (e != hipSuccess) return gdml_fail(ctx, GDML_ERR_HIP, "gram_tn: %s", hipGetErrorString(e));
  return rc;
}

// out[r] = sum_c X[r][c]^2   (leverage scores, iterative.py:107-109)
__global__ void __launch_bounds__(256) row_sqnorm_kernel(const double* __restrict__ X, int64_t ld,
                                                         int64_t n, int64_t m, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  double s = 0.0;
  for (int64_t c = lane; c < m; c += 64) {
    const double v = X[r * ld + c];
    s += v * v;
  }
  s = wave_sum(s);
  if (lane == 0) out[r] = s;
}

// t_part[rc][c] = sum_{r in chunk rc} X[r][c] v[r].  HBM bound (X is read once per call): a thread owns two adjacent
// columns (16-byte loads; rows are padded to a multiple of 16 doubles, so the pair of an odd last column stays inside the
// row) and keeps eight rows in flight.
// the factor is streamed (25 GB per pass against 256 MB of Infinity Cache): NT = non-temporal loads
template <bool NT>
__device__ __forceinline__ d2 ld_stream(const d2* p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__global__ void __launch_bounds__(256) gemv_t_part_kernel(const double* __restrict__ X, int64_t ld,
                                                          int64_t n, int64_t m,
                                                          const double* __restrict__ v, int rows_per,
                                                          double* __restrict__ part) {
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per;
  const int64_t r1 = (r0 + rows_per < n) ? r0 + rows_per : n;
  if (c >= m) return;
  const double* col = X + c;
  d2 s = {0.0, 0.0};
  int64_t r = r0;
  for (; r + 8 <= r1; r += 8) {
    d2 x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = ld_stream<NT>(reinterpret_cast<const d2*>(col + (r + u) * ld));
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double vr = v[r + u];
      s.x += x[u].x * vr;
      s.y += x[u].y * vr;
    }
  }
  for (; r < r1; ++r) {
    const d2 x = *reinterpret_cast<const d2*>(col + r * ld);
    const double vr = v[r];
    s.x += x.x * vr;
    s.y += x.y * vr;
  }
  part[(int64_t)blockIdx.y * m + c] = s.x;
  if (c + 1 < m) part[(int64_t)blockIdx.y * m + c + 1] = s.y;
}
__global__ void __launch_bounds__(256) reduce_parts_kernel(const double* __restrict__ part, int64_t m,
                                                           int nparts, double* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= m) return;
  double s = 0.0;
  for (int p = 0; p < nparts; ++p) s += part[(int64_t)p * m + c];
  out[c] = s;
}
// out[r] = (sum_c X[r][c] t[c] - v[r]) * inv_lam  (PRECON; otherwise the plain product sum_c X[r][c] t[c]).  One wavefront
// per row, 16-byte loads, four of them in flight per lane.
template <bool NT, bool PRECON = true>
__global__ void __launch_bounds__(256) gemv_n_precon_kernel(const double* __restrict__ X, int64_t ld,
                                                            int64_t n, int64_t m,
                                                            const double* __restrict__ t,
                                                            const double* __restrict__ v, double inv_lam,
                                                            double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const double* row = X + r * ld;
  const int64_t m2 = m & ~(int64_t)1;  // even part: pairs; an odd last column is added by lane 0
  double s0 = 0.0, s1 = 0.0;
  int64_t c = 2 * lane;
  for (; c + 3 * 128 < m2; c += 4 * 128) {
    d2 x[4], tt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      x[u] = ld_stream<NT>(reinterpret_cast<const d2*>(row + c + 128 * u));
      tt[u] = *reinterpret_cast<const d2*>(t + c + 128 * u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s0 += x[u].x * tt[u].x;
      s1 += x[u].y * tt[u].y;
    }
  }
  for (; c < m2; c += 128) {
    const d2 x = *reinterpret_cast<const d2*>(row + c);
    const d2 tt = *reinterpret_cast<const d2*>(t + c);
    s0 += x.x * tt.x;
    s1 += x.y * tt.y;
  }
  if (lane == 0 && m2 < m) s0 += row[m2] * t[m2];
  const double s = wave_sum(s0 + s1);
  if (lane == 0) out[r] = PRECON ? (s - v[r]) * inv_lam : s;
}

// ---- matrix-free form of the preconditioner (pcg.precon_form): K_mn v is the kernel mat-vec followed by a gather of the
// inducing entries, K_nm t a scatter of t into an n-vector followed by the mat-vec (K is symmetric)
__global__ void __launch_bounds__(256) gather_idx_kernel(const double* __restrict__ s, const int64_t* __restrict__ idx,
                                                         int64_t m, double* __restrict__ t) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q < m) t[q] = s[idx[q]];
}
__global__ void __launch_bounds__(256) scatter_idx_kernel(const double* __restrict__ t, const int64_t* __restrict__ idx,
                                                          int64_t m, double* __restrict__ e) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q < m) e[idx[q]] = t[q];
}
__global__ void __launch_bounds__(256) precon_finish_kernel(const double* __restrict__ w, const double* __restrict__ v,
                                                            double inv_lam, int64_t n, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (w[i] - v[i]) * inv_lam;
}

// ---- fp32-stored form of the factor (pcg.precon_form = 3).  The two GEMVs of an application are HBM bound at the rate
// the memory delivers (16 n m bytes per application); the factor's information content is not: what the preconditioner needs
// from X is its column SPACE to an angle theta with theta^2 sigma_max << lam (fp32: 6e-8), and the spectrum
// 1 - lam / (sigma_i + lam) on it -- which no rounded X can carry (1 - 1e-12), and which the stored fp64 factor itself only
// holds to ~1e-10 (profiles/r05_pcg_bisect.txt).  So X is rounded to fp32 (half the bytes) and the spectrum is put back
// by an m x m matrix:   P v = (X32 T0 X32^T v - v) / lam,   T0 = L_G^-T M0 L_G^-1,
// G = X32^T X32 = L_G L_G^T (fp64 Gram of the rounded factor: Q32 = X32 L_G^-T is orthonormal to rounding), and M0 the
// exact factor's operator in ITS orthonormal basis: X = Q L0^T with L0 L0^T = G0 = X^T X = I - lam L^-1 L^-T (the Gram of the
// exact factor; L L^T = K_nm^T K_nm + lam I, iterative.py:293-306), so X X^T = Q (L0^T L0) Q^T and M0 = L0^T L0.
// In exact arithmetic and without rounding (X32 = X, L_G = L0: T0 = I) this IS (X X^T v - v)/lam; with it, it is that
// operator carried over to range(X32).  Round 5 used G0 = L0 L0^T in the place of M0 = L0^T L0: the same eigenvalues on
// eigenvectors rotated by O(1 - s_min^2) inside range(X) -- invisible at lam = 1e-10 on well-supported inducing columns
// (s^2 ~ 1 - 1e-10), 3 % of the operator on the round-6 fixture n10_p2_pbc (lam = 1e-4, s_min^2 = 0.06), where the test
// that compares the forms found it.  M0 = G0 + (C^T C - C C^T) with C = I - L0: the difference is second order in C, so
// the entries ~1 of G0 (the spectrum 1 - lam / (sigma + lam) the form exists to carry) are never re-rounded.
typedef float f4 __attribute__((ext_vector_type(4)));

// X32 <- float(X); pad columns [m, ld) of X32 <- 0.  X itself stays as it is (leverage scores, fall-back to the fp64 form)
__global__ void __launch_bounds__(256) round_f32_kernel(const double* __restrict__ X, int64_t ld, int64_t n, int64_t m,
                                                        float* __restrict__ X32) {
  const int64_t r = blockIdx.x;
  const double* row = X + r * ld;
  float* row32 = X32 + r * ld;
  for (int64_t c = threadIdx.x; c < ld; c += 256) row32[c] = c < m ? (float)row[c] : 0.0f;
}
// Compensated accumulation of the chunk Gram matrices (lower tiles): G += Gp with the running compensation in Gc.  The
// products of two fp32 values are exact in fp64, a chunk of 2048 rows is one short MFMA chain per entry, and the chunks are
// summed here with Kahan's scheme: the diagonal of G (entries ~1, the only ones whose rounding matters on the scale
// lam / sigma_max ~ 1e-13 the operator lives on) comes out to ~2e-16 instead of eps sqrt(n / 4) ~ 3e-14.
__global__ void __launch_bounds__(256) kahan_acc_lower_kernel(double* __restrict__ G, double* __restrict__ Gc,
                                                              const double* __restrict__ Gp, int64_t ld, int64_t m, int first) {
  const int64_t r = blockIdx.x;
  const int64_t cend = ((r / TT) + 1) * TT < m ? ((r / TT) + 1) * TT : m;  // the lower TILES hold valid data
  for (int64_t c = threadIdx.x; c < cend; c += 256) {
    const int64_t e = r * ld + c;
    if (first) {
      G[e] = Gp[e];
      Gc[e] = 0.0;
    } else {
      const double y = Gp[e] - Gc[e];
      const double t = G[e] + y;
      Gc[e] = (t - G[e]) - y;
      G[e] = t;
    }
  }
}
// A <- A - I
__global__ void __launch_bounds__(256) sub_eye_kernel(double* __restrict__ A, int64_t ld, int64_t m) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < m) A[t * ld + t] -= 1.0;
}
// T0 <- G0 - H - H^T + S   (in place on the buffer that holds G0)
__global__ void __launch_bounds__(256) t0_combine_kernel(double* __restrict__ T0, const double* __restrict__ H,
                                                         const double* __restrict__ S, int64_t ld, int64_t m) {
  const int64_t r = blockIdx.x;
  for (int64_t c = threadIdx.x; c < m; c += 256) T0[r * ld + c] = ((T0[r * ld + c] - H[r * ld + c]) - H[c * ld + r]) + S[r * ld + c];
}
// D (rows x ld doubles) <- double(float(X rows)), pad columns 0: what widen_f32_kernel gives for a rounded copy, without the copy
__global__ void __launch_bounds__(256) widen_round_kernel(const double* __restrict__ X, int64_t ld, int64_t m, double* __restrict__ D) {
  const int64_t r = blockIdx.x;
  for (int64_t c = threadIdx.x; c < ld; c += 256) D[r * ld + c] = c < m ? (double)(float)X[r * ld + c] : 0.0;
}
// In-place rounding of the factor (option pcg.f32_inplace, round 6): row r of X (ld doubles) becomes ld floats at the START of
// the same row -- X32 with a row pitch of 2 ld floats, no second buffer.  Chunks of 256 columns front to back: the floats of
// chunk k overlay doubles [128 k, 128 k + 128), all of which were read in this or an earlier chunk; floats [m, ld) <- 0.
__global__ void __launch_bounds__(256) round_f32_inplace_kernel(double* __restrict__ X, int64_t ld, int64_t m) {
  const int64_t r = blockIdx.x;
  const double* row = X + r * ld;
  float* row32 = reinterpret_cast<float*>(X + r * ld);
  for (int64_t c0 = 0; c0 < ld; c0 += 256) {
    const int64_t c = c0 + threadIdx.x;
    const float v = (c < m) ? (float)row[c < ld ? c : ld - 1] : 0.0f;
    __syncthreads();
    if (c < ld) row32[c] = v;
    __syncthreads();
  }
}
// D (rows x ld doubles) <- double(X32 rows): the Gram matrix of the rounded factor is accumulated chunk by chunk
__global__ void __launch_bounds__(256) widen_f32_kernel(const float* __restrict__ X32, int64_t ld, double* __restrict__ D) {
  const int64_t r = blockIdx.x;
  const float* src = X32 + r * ld;
  double* dst = D + r * ld;
  for (int64_t c = threadIdx.x; c < ld; c += 256) dst[c] = (double)src[c];
}
__global__ void __launch_bounds__(256) cg_scale_kernel(double* __restrict__ A, int64_t count, double f) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count) A[i] *= f;
}
// A (lower triangle valid) mirrored into the upper triangle; then A <- I - A
__global__ void __launch_bounds__(256) sym_fill_kernel(double* __restrict__ A, int64_t ld, int64_t m) {
  const int64_t r = blockIdx.x;
  for (int64_t c = threadIdx.x; c < r; c += 256) A[c * ld + r] = A[r * ld + c];
}
__global__ void __launch_bounds__(256) eye_minus_kernel(double* __restrict__ A, int64_t ld, int64_t m) {
  const int64_t r = blockIdx.x;
  for (int64_t c = threadIdx.x; c < m; c += 256) A[r * ld + c] = (c == r ? 1.0 : 0.0) - A[r * ld + c];
}
// C <- I - L (lower triangle incl. the diagonal of L), 0 above the diagonal and in the pad columns [m, ld)
__global__ void __launch_bounds__(256) eye_minus_lower_kernel(const double* __restrict__ L, double* __restrict__ C, int64_t ld,
                                                              int64_t m) {
  const int64_t r = blockIdx.x;
  for (int64_t c = threadIdx.x; c < ld; c += 256) C[r * ld + c] = c > r || c >= m ? 0.0 : (c == r ? 1.0 : 0.0) - L[r * ld + c];
}
// t_part[rc][c] = sum_{r in chunk rc} X32[r][c] v[r]: a thread owns four adjacent columns (16-byte loads); products and
// sums in fp64 (the fp32 values are exact doubles).  The sums are COMPENSATED (Kahan): what the operator needs from
// X32^T v on the dominant subspace is 1 - lam / (sigma + lam) ~ 1 - 1e-12, and a plain chain of rows_per additions loses
// eps sqrt(rows_per) of that (profiles/r05_f32_diag.txt: 1.4e-2 of the top direction's action against 1.8e-3 for a blocked
// CPU sum, the difference between a 300-iteration plateau and none); the kernel is HBM bound, the four extra additions
// per element are free.
struct KahanSum {
  double s = 0.0, c = 0.0;
  __device__ __forceinline__ void add_prod(double x, double v) {
    const double y = __builtin_fma(x, v, -c);
    const double t = s + y;
    c = (t - s) - y;
    s = t;
  }
  __device__ __forceinline__ void add(double x) {
    const double y = x - c;
    const double t = s + y;
    c = (t - s) - y;
    s = t;
  }
};
template <bool NT>
__global__ void __launch_bounds__(256) gemv_t_part_f32_kernel(const float* __restrict__ X, int64_t ld, int64_t n, int64_t m,
                                                              const double* __restrict__ v, int rows_per,
                                                              double* __restrict__ part) {
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per;
  const int64_t r1 = (r0 + rows_per < n) ? r0 + rows_per : n;
  if (c >= m) return;
  const float* col = X + c;
  KahanSum s0, s1, s2, s3;
  int64_t r = r0;
  for (; r + 8 <= r1; r += 8) {
    f4 x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f4* p = reinterpret_cast<const f4*>(col + (r + u) * ld);
      x[u] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double vr = v[r + u];
      s0.add_prod((double)x[u].x, vr);
      s1.add_prod((double)x[u].y, vr);
      s2.add_prod((double)x[u].z, vr);
      s3.add_prod((double)x[u].w, vr);
    }
  }
  for (; r < r1; ++r) {
    const f4 x = *reinterpret_cast<const f4*>(col + r * ld);
    const double vr = v[r];
    s0.add_prod((double)x.x, vr);
    s1.add_prod((double)x.y, vr);
    s2.add_prod((double)x.z, vr);
    s3.add_prod((double)x.w, vr);
  }
  double* o = part + (int64_t)blockIdx.y * m + c;
  o[0] = s0.s;
  if (c + 1 < m) o[1] = s1.s;
  if (c + 2 < m) o[2] = s2.s;
  if (c + 3 < m) o[3] = s3.s;
}
__global__ void __launch_bounds__(256) reduce_parts_kahan_kernel(const double* __restrict__ part, int64_t m, int nparts,
                                                                 double* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= m) return;
  KahanSum s;
  for (int p = 0; p < nparts; ++p) s.add(part[(int64_t)p * m + c]);
  out[c] = s.s;
}
// out[r] = (sum_c X32[r][c] t[c] - v[r]) * inv_lam.  A wavefront owns RW consecutive rows: a row of X32 is half as long
// as the fp64 vector t it is multiplied with, so with one row per wavefront the t reads (from L2) would be twice the X32
// reads (from HBM); t is loaded once per RW rows.  t is padded with zeros up to a multiple of 4, X32's pad columns are zero.
template <bool NT, int RW>
__global__ void __launch_bounds__(256) gemv_n_precon_f32_kernel(const float* __restrict__ X, int64_t ld, int64_t n, int64_t m,
                                                                const double* __restrict__ t, const double* __restrict__ v,
                                                                double inv_lam, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (r0 >= n) return;
  const float* row[RW];
#pragma unroll
  for (int k = 0; k < RW; ++k) row[k] = X + ((r0 + k < n) ? r0 + k : n - 1) * ld;
  const int64_t m4 = (m + 3) & ~(int64_t)3;
  double s[RW];
#pragma unroll
  for (int k = 0; k < RW; ++k) s[k] = 0.0;
  int64_t c = 4 * lane;
  for (; c + 256 < m4; c += 512) {
    f4 x[2][RW];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < RW; ++k) {
        const f4* p = reinterpret_cast<const f4*>(row[k] + c + 256 * u);
        x[u][k] = NT ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const d2 ta = *reinterpret_cast<const d2*>(t + c + 256 * u);
      const d2 tb = *reinterpret_cast<const d2*>(t + c + 256 * u + 2);
#pragma unroll
      for (int k = 0; k < RW; ++k)
        s[k] += ((double)x[u][k].x * ta.x + (double)x[u][k].y * ta.y) + ((double)x[u][k].z * tb.x + (double)x[u][k].w * tb.y);
    }
  }
  for (; c < m4; c += 256) {
    const d2 ta = *reinterpret_cast<const d2*>(t + c);
    const d2 tb = *reinterpret_cast<const d2*>(t + c + 2);
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      const f4 x = *reinterpret_cast<const f4*>(row[k] + c);
      s[k] += ((double)x.x * ta.x + (double)x.y * ta.y) + ((double)x.z * tb.x + (double)x.w * tb.y);
    }
  }
#pragma unroll
  for (int k = 0; k < RW; ++k) {
    const double sk = wave_sum(s[k]);
    if (lane == 0 && r0 + k < n) out[r0 + k] = (sk - v[r0 + k]) * inv_lam;
  }
}

// ---- vector kernels of the PCG loop.  The CG scalars (rho, p.Ap, ||r||^2) never leave the device: a dot product is
// 256 per-workgroup partials, and every consumer sums them itself in one fixed order (same value in every workgroup and on
// every rank), so an iteration is a chain of launches without a host round trip.
#define PCG_PARTS 256
__global__ void __launch_bounds__(256) dot_part_kernel(const double* __restrict__ a,
                                                       const double* __restrict__ b, int64_t n,
                                                       double* __restrict__ part) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    s += a[i] * b[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// sum of PCG_PARTS partials, identical in every thread of every 256-thread workgroup that calls it
template <bool COHERENT = false>
__device__ __forceinline__ double sum_parts(const double* part, double* red /* LDS, 4 doubles */) {
  // COHERENT: the partials were written by other workgroups of the SAME launch -> read them past the per-CU L1
  const double v = COHERENT ? __hip_atomic_load(part + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : part[threadIdx.x];
  const double s = wave_sum(v);
  __syncthreads();  // red may still be read from a previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
// p = z + (rho / rho_prev) p        (first iteration: p = z)
__global__ void __launch_bounds__(256) pcg_p_update_kernel(double* __restrict__ p, const double* __restrict__ z,
                                                           const double* __restrict__ part_rho,
                                                           const double* __restrict__ part_rho_prev, int first,
                                                           int64_t n) {
  __shared__ double red[4];
  double beta = 0.0;
  if (!first) {
    const double rho = sum_parts(part_rho, red);
    const double rho_prev = sum_parts(part_rho_prev, red);
    beta = rho / rho_prev;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    p[i] = first ? z[i] : z[i] + beta * p[i];
}
// alpha = rho / (p.Ap), q = -(A p):  x_out = x_in + alpha p,  r += alpha q,  and ||r||^2 of the updated residual: the last
// workgroup to finish sums the partials and publishes the value to the host-mapped slot `rr_host` (and to rr_dev).
__global__ void __launch_bounds__(256) pcg_xr_update_kernel(const double* x_in, double* x_out,  // the same vector at depth 0
                                                            double* __restrict__ r, const double* __restrict__ p,
                                                            const double* __restrict__ q,
                                                            const double* __restrict__ part_rho,
                                                            const double* __restrict__ part_pq,
                                                            double* __restrict__ part_rr, unsigned* __restrict__ counter,
                                                            double* __restrict__ rr_dev, volatile double* rr_host,
                                                            int64_t n) {
  __shared__ double red[4];
  __shared__ unsigned ticket;
  const double rho = sum_parts(part_rho, red);
  const double pq = sum_parts(part_pq, red);
  const double alpha = rho / (-pq);
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    x_out[i] = x_in[i] + alpha * p[i];
    const double rv = r[i] + alpha * q[i];
    r[i] = rv;
    s += rv * rv;
  }
  s = wave_sum(s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    part_rr[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    __threadfence();
    ticket = atomicAdd(counter, 1u);
  }
  __syncthreads();
  if (ticket == gridDim.x - 1) {  // every partial is visible: one fixed-order sum
    __threadfence();
    const double rr = sum_parts<true>(part_rr, red);
    if (threadIdx.x == 0) {
      *rr_dev = rr;
      *rr_host = rr;
      __threadfence_system();
      *counter = 0u;
    }
  }
}
// ||r||^2 of a residual that did not come out of pcg_xr_update_kernel (the start vector): partials -> slot
__global__ void __launch_bounds__(256) pcg_publish_kernel(const double* __restrict__ part, double* __restrict__ rr_dev,
                                                          volatile double* rr_host) {
  __shared__ double red[4];
  const double rr = sum_parts(part, red);
  if (threadIdx.x == 0) {
    *rr_dev = rr;
    *rr_host = rr;
    __threadfence_system();
  }
}
__global__ void __launch_bounds__(256) vec_axpy_kernel(double* __restrict__ y, const double* __restrict__ x,
                                                       double a, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] += a * x[i];
}

// X[:, 0:m] <- X L^-T  for the n rows of X (L: m x m lower), blocked 512 / 64 like the Cholesky.
// Round 6: LEFT-looking (option nys.trsm_left, default 1): a 512-column strip receives everything the solved strips to its left
// owe it in ONE product of depth k0 and is then solved, instead of being read, updated by 512 and written back once per earlier
// strip -- m / 1024 times less traffic on X and long k loops (the K = 512 update ran at 0.55 of the MFMA peak inside the
// configs[3] build, profiles/r06_cfg34_kernel_stats.txt).  Same flops, same operations in another order.
static int tall_trsm(gdml_ctx* ctx, const double* L, double* X, int64_t n, int64_t m, int64_t ld) {
  const int64_t NB = 512;
  hipStream_t st = ctx->stream;
  const bool left = ctx_opt_i(ctx, "nys.trsm_left", 1) != 0;
  for (int64_t k0 = 0; k0 < m; k0 += NB) {
    const int64_t nb = (m - k0 < NB) ? m - k0 : NB;
    if (left && k0 > 0)  // X[:, k0:k0+nb] -= X[:, 0:k0] L[k0:k0+nb, 0:k0]^T
      GDML_TRY(launch_gemm_nt_sub(ctx, st, X, ld, L + k0 * ld, ld, X + k0, ld, n, nb, k0, 0));
    const bool aligned = ((reinterpret_cast<uintptr_t>(L) | reinterpret_cast<uintptr_t>(X)) & 31) == 0 && (ld % 4 == 0);
    if (nb % 64 == 0 && aligned && ctx_opt_i(ctx, "chol.panel_kernel", 1)) {
      // whole NB-wide strip in one row-local launch (the kernel of the Cholesky panels) instead of 8 x (trsm64 + K = 64 GEMM)
      GDML_TRY(launch_panel_trsm(ctx, st, L + k0 * ld + k0, X + k0, ld, (int)nb, n));
    } else
    for (int64_t jj = 0; jj < nb; jj += 64) {
      const int64_t c0 = k0 + jj;
      const int w = (int)((nb - jj < 64) ? nb - jj : 64);
      GDML_TRY(launch_trsm64(ctx, st, L + c0 * ld + c0, X + c0, ld, w, n));
      const int64_t ncols = k0 + nb - (c0 + w);
      if (ncols > 0)
        GDML_TRY(launch_gemm_nt_sub(ctx, st, X + c0, ld, L + (c0 + w) * ld + c0, ld, X + c0 + w, ld, n,
                                    ncols, w, 0));
    }
    const int64_t t0 = k0 + nb;
    if (!left && t0 < m)
      GDML_TRY(launch_gemm_nt_sub(ctx, st, X + k0, ld, L + t0 * ld + k0, ld, X + t0, ld, n, m - t0, nb, 0));
  }
  return GDML_OK;
}

// Cholesky with escalating diagonal jitter (iterative.py:414-471).  A: m x m (lower referenced).
// Returns GDML_OK with *ok = 1 on success, *ok = 0 if every attempt failed.
// force_fail (test hook nys.force_fail): treat the first force_fail attempts as failed, whatever the factorisation says.
static int cho_factor_stable_dev(gdml_ctx* ctx, double* A, int64_t m, int64_t ld, double* backup,
                                 bool pre_reg, int eps_mag_max, int* ok, int force_fail = 0, int* n_jitter = nullptr) {
  const double eps = 2.220446049250313e-16;
  int eps_mag = (int)floor(log10(eps));  // -16
  if (pre_reg) {
    hipLaunchKernelGGL(add_diag_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, ctx->stream, A, ld, m, eps);
    eps_mag += 1;
  }
  *ok = 0;
  int attempt = 0;
  for (int mag = eps_mag; mag <= eps_mag_max; ++mag, ++attempt) {
    HIP_CHECK(ctx, hipMemcpy2DAsync(backup, m * 8, A, ld * 8, m * 8, m, hipMemcpyDeviceToDevice,
                                    ctx->stream));
    int info = 0;
    GDML_TRY(chol_factor_device(ctx, A, m, ld, &info));
    if (info == 0 && attempt >= force_fail) {
      *ok = 1;
      if (n_jitter) *n_jitter = attempt;
      return GDML_OK;
    }
    HIP_CHECK(ctx, hipMemcpy2DAsync(A, ld * 8, backup, m * 8, m * 8, m, hipMemcpyDeviceToDevice,
                                    ctx->stream));
    hipLaunchKernelGGL(add_diag_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, ctx->stream, A, ld, m,
                       pow(10.0, (double)mag));
  }
  return GDML_OK;
}

// Row-shard geometry of the resident Nystroem matrix: this rank's n_loc rows are the entries [row0, row0 + n_loc) of the
// replicated device vectors, which are padded to n_pad = world * chunk doubles (VecLayout, common.h: the reference order when
// only forces are trained; rank-major with each rank's energy rows behind its force rows under energy constraints).
typedef VecLayout ShardGeo;
static ShardGeo shard_geo(const gdml_ctx* ctx) {
  if (ctx->K_sharded) {
    ShardGeo g = vec_layout(ctx, ctx->K_use_E);
    g.n = ctx->K_rows_global;
    return g;
  }
  ShardGeo g;
  g.M = ctx->ts.M; g.N3 = 3 * (int64_t)ctx->ts.N; g.n_ff = g.M * g.N3; g.per = g.M;
  g.n = g.n_loc = g.chunk = g.n_pad = ctx->K_rows;
  g.row0 = 0;
  return g;
}

// dst rows (work block) <- -K[idx[q], :] for the rows this rank owns, zero otherwise (iterative.py:250)
__global__ void __launch_bounds__(256) gather_neg_rows_sharded_kernel(const double* __restrict__ X,
                                                                      double* __restrict__ S, int64_t ld,
                                                                      const int64_t* __restrict__ idx,
                                                                      int64_t m, VecLayout L) {
  const int64_t q = blockIdx.x;
  const int64_t r = L.pos(idx[q]) - L.row0;  // idx: reference order; the rank's rows: one run of the vector layout
  double* dst = S + q * ld;
  if (r >= 0 && r < L.n_loc) {
    const double* src = X + r * ld;
    for (int64_t c = threadIdx.x; c < m; c += 256) dst[c] = -src[c];
  } else {
    for (int64_t c = threadIdx.x; c < m; c += 256) dst[c] = 0.0;
  }
}

// Form of the preconditioner application (option pcg.precon_form):
//   0  the stored n x m fp64 factor, streamed twice per application -- the reference's operator (iterative.py:120-140)
//   1  matrix-free (two kernel mat-vecs + the m x m matrix Z).  Experimental: the same operator to ~1e-8 of its norm, which
//      is NOT enough -- (X X^T - I) v cancels to lam / (sigma + lam) ~ 1e-12 on the dominant subspace, X = K_nm Z is only
//      orthonormal when formed row by row (cond(K_nm) ~ 1e10: |K_nm| |Z| eps ~ 1e-5), and PCG at lam = 1e-10 does not
//      converge with it (profiles/r05_pcg_bisect.txt).  Usable for lam >~ 1e-6; never chosen automatically.
//   3  the factor stored in fp32 + the m x m Gram correction T0 (above): half the bytes per application
//   2  automatic (default): 3 once the factor reaches 1 GiB per rank, else 0 -- every reference-parity fixture of the test
//      suite stays on the reference's form.  The extra build work of form 3 (one more Gram pass over the factor, 2 n m^2
//      flops, ~8 m^3 for T0) is paid back twice: half the bytes per application, and -- the larger effect -- FEWER
//      iterations: the stored factor carries the spectrum 1 - lam / (sigma + lam) of the dominant directions only to ~1e-10
//      (its Gram matrix is off the exact one by that much), which parks PCG on a plateau for hundreds of iterations; T0
//      carries it to ~1e-16 (configs[2]: 861 -> 306 iterations, configs[4]: 2261 -> 841, profiles/r05_precon_forms.txt).
// Every input of the decision is the same on every rank.
static int choose_precon_form(const gdml_ctx* ctx, const ShardGeo& sg, int64_t m, int64_t ld) {
  const int opt = ctx_opt_i(ctx, "pcg.precon_form", 2);
  if (opt == 0 || opt == 1 || opt == 3) return opt;
  return 8.0 * (double)sg.chunk * (double)m >= (double)((int64_t)1 << 30) ? 3 : 0;
}

// fp32 copy and Gram correction of form 3, the matrix-free form's Z: released when the form is rejected and when the
// matrix buffer they belong to changes size (assemble.hip) -- not on every assembly: PCG restarts re-factor at one size
void precon_release_f32(gdml_ctx* ctx) {
  if (ctx->precon_X32 && !ctx->precon_X32_inplace) ctx_free(ctx, ctx->precon_X32);
  ctx->precon_X32_inplace = false;
  if (ctx->precon_T0) ctx_free(ctx, ctx->precon_T0);
  ctx->precon_X32 = nullptr; ctx->precon_X32_bytes = 0;
  ctx->precon_T0 = nullptr; ctx->precon_T0_bytes = 0;
}
void precon_release_aux(gdml_ctx* ctx) {
  precon_release_f32(ctx);
  if (ctx->precon_Z) ctx_free(ctx, ctx->precon_Z);
  ctx->precon_Z = nullptr; ctx->precon_Z_bytes = 0;
}

static int ensure_buf(gdml_ctx* ctx, void** p, int64_t* have, int64_t want) {
  if (*have >= want) return GDML_OK;
  if (*p) GDML_TRY(ctx_free(ctx, *p));
  *p = nullptr;
  *have = 0;
  GDML_TRY(ctx_alloc(ctx, p, want));
  *have = want;
  return GDML_OK;
}

// fp32 form: X (complete fp64 factor, rows of this rank) and S = L (lower, L L^T = K_nm^T K_nm + lam I) -> X32, T0.
// X itself is not modified.  *usable = 0 when the Gram matrix of the rounded factor cannot be factored or is too ill
// conditioned for X32 L_G^-T to be orthonormal to ~1e-9 (squared singular values sigma_i / (sigma_i + lam) of X below the
// threshold: inducing columns the data does not support, a numerically rank-deficient K_mm).  The caller then keeps the
// reference's fp64 form.
static int build_f32_form(gdml_ctx* ctx, double lam, double* X, const double* S, int64_t n_loc, int64_t m, int64_t ld,
                          int* usable, bool inplace) {
  *usable = 0;
  ctx->opts["pcg.f32_last_min_pivot"] = 0.0;
  hipStream_t st = ctx->stream;
  void* tmp = nullptr;
  {  // no room for the fp32 copy next to the fp64 factor: keep the reference's form (X is untouched at this point)
    int rc_a = inplace ? GDML_OK  // (the rounded factor will overlay the fp64 one: gdml_nystroem_factor converts it at the end)
                       : ensure_buf(ctx, (void**)&ctx->precon_X32, &ctx->precon_X32_bytes, (n_loc > 0 ? n_loc : 1) * ld * 4);
    if (rc_a == GDML_OK) rc_a = ensure_buf(ctx, (void**)&ctx->precon_T0, &ctx->precon_T0_bytes, m * ld * 8);
    if (rc_a == GDML_OK) rc_a = ctx_alloc(ctx, &tmp, 4 * m * ld * 8);
    // sharded: every rank must apply the same form (row blocks of ONE operator) -- free HBM differs between the ranks, so
    // the decision is collective: the form is kept only if the buffers fit everywhere.  EVERY local outcome goes through
    // the collective first (a rank that returned early on a non-OOM error left the others hanging in it); the error is
    // reported afterwards.
    double fits = rc_a == GDML_OK ? 1.0 : 0.0;
    if (comm_active(ctx)) {
      double* d_flag;
      GDML_TRY(ctx_slot(ctx, 10, 64, &d_flag));
      HIP_CHECK(ctx, hipMemcpyAsync(d_flag, &fits, 8, hipMemcpyHostToDevice, st));
      GDML_TRY(comm_allreduce_sum(ctx, d_flag, 1));
      HIP_CHECK(ctx, hipMemcpyAsync(&fits, d_flag, 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(ctx, hipStreamSynchronize(st));
      fits = fits >= (double)ctx->world - 0.5 ? 1.0 : 0.0;
    }
    if (fits < 0.5) {
      if (tmp) ctx_free(ctx, tmp);
      precon_release_f32(ctx);  // a partly allocated form (X32 fits, T0 or the work space does not) holds no memory
      return (rc_a != GDML_OK && rc_a != GDML_ERR_OOM) ? rc_a : GDML_OK;
    }
  }
  double* Rb = (double*)tmp;   // L^-T, later E_z = L_G^-T - I
  double* G = Rb + m * ld;     // Gram of the rounded factor, then its Cholesky factor, then S = W1 E_z^T
  double* Gc = G + m * ld;     // Kahan compensation of the Gram sum, then H = -W1
  double* Gp = Gc + m * ld;    // Gram matrix of one chunk of rows
  double* G0 = ctx->precon_T0; // lives in the T0 buffer until T0 itself is formed in place
  const int tiles = (int)((m + TT - 1) / TT);
  const dim3 tri((unsigned)(tiles * (tiles + 1) / 2));
  auto identity = [&](double* A) -> int {
    HIP_CHECK(ctx, hipMemsetAsync(A, 0, m * ld * 8, st));
    hipLaunchKernelGGL(add_diag_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, st, A, ld, m, 1.0);
    return GDML_OK;
  };
  auto body = [&]() -> int {
    // G0 = I - lam L^-1 L^-T: R = L^-T (the solve applied to the identity), lam R^T R on the MFMA Gram kernel
    GDML_TRY(identity(Rb));
    GDML_TRY(tall_trsm(ctx, S, Rb, m, m, ld));
    hipLaunchKernelGGL(cg_scale_kernel, dim3(ceil_div(m * ld, 256)), dim3(256), 0, st, Rb, m * ld, sqrt(lam));
    hipLaunchKernelGGL(syrk_tn_kernel, tri, dim3(256), 0, st, Rb, ld, m, m, G0, ld, tiles, 0);
    hipLaunchKernelGGL(sym_fill_kernel, dim3((unsigned)m), dim3(256), 0, st, G0, ld, m);
    hipLaunchKernelGGL(eye_minus_kernel, dim3((unsigned)m), dim3(256), 0, st, G0, ld, m);
    // rounded factor and its Gram matrix (summed over the row shards).  The fp32 values are widened chunk by chunk into a
    // work buffer (X stays untouched), every chunk's Gram matrix is one short MFMA chain per entry, and the chunks are
    // summed with compensation (kahan_acc_lower_kernel): the diagonal of G is what the spectrum correction hangs on
    if (n_loc > 0 && !inplace)
      hipLaunchKernelGGL(round_f32_kernel, dim3((unsigned)n_loc), dim3(256), 0, st, X, ld, n_loc, m, ctx->precon_X32);
    {
      int64_t chunk = (int64_t)ctx_opt(ctx, "pcg.f32_gram_rows", 2048);
      chunk = chunk < 256 ? 256 : chunk / 16 * 16;
      if (chunk > n_loc) chunk = n_loc > 0 ? n_loc : 1;
      double* D;
      GDML_TRY(ctx_slot(ctx, 11, chunk * ld * 8, &D));
      HIP_CHECK(ctx, hipMemsetAsync(G, 0, 2 * m * ld * 8, st));  // G and Gc (upper tiles are never written again)
      for (int64_t r0 = 0; r0 < n_loc; r0 += chunk) {
        const int64_t rows = (n_loc - r0 < chunk) ? n_loc - r0 : chunk;
        if (inplace)
          hipLaunchKernelGGL(widen_round_kernel, dim3((unsigned)rows), dim3(256), 0, st, X + r0 * ld, ld, m, D);
        else
          hipLaunchKernelGGL(widen_f32_kernel, dim3((unsigned)rows), dim3(256), 0, st, ctx->precon_X32 + r0 * ld, ld, D);
        hipLaunchKernelGGL(syrk_tn_kernel, tri, dim3(256), 0, st, D, ld, rows, m, Gp, ld, tiles, 0);
        hipLaunchKernelGGL(kahan_acc_lower_kernel, dim3((unsigned)m), dim3(256), 0, st, G, Gc, Gp, ld, m, r0 == 0 ? 1 : 0);
      }
    }
    GDML_TRY(comm_allreduce_sum(ctx, G, m * ld));
    int inf = 0;
    int rcg = chol_factor_device(ctx, G, m, ld, &inf);
    if (rcg == GDML_ERR_NOT_PD || inf != 0) return GDML_OK;  // *usable stays 0: X is intact, the caller keeps the fp64 form
    GDML_TRY(rcg);
    {  // conditioning of the rounded factor's Gram matrix, from its Cholesky pivots: X32 L_G^-T is orthonormal to
       // eps cond(G); beyond ~1e7 the reference's form is kept (pcg.f32_min_pivot: threshold on the smallest squared pivot)
      std::vector<double> diag((size_t)m);
      HIP_CHECK(ctx, hipMemcpy2DAsync(diag.data(), 8, G, (ld + 1) * 8, 8, m, hipMemcpyDeviceToHost, st));
      HIP_CHECK(ctx, hipStreamSynchronize(st));
      double dmin = 1e300;
      for (double d : diag) dmin = d < dmin ? d : dmin;
      ctx->opts["pcg.f32_last_min_pivot"] = dmin * dmin;  // diagnostic, read back with gdml_get_option
      if (!(dmin * dmin >= ctx_opt(ctx, "pcg.f32_min_pivot", 1e-7))) return GDML_OK;
    }
    // G0 <- M0 = L0^T L0 = G0 + (C^T C - C C^T), C = I - L0, L0 L0^T = G0 (see the derivation above build_f32_form's kernels)
    {
      HIP_CHECK(ctx, hipMemcpyAsync(Gp, G0, m * ld * 8, hipMemcpyDeviceToDevice, st));
      int inf0 = 0;
      const int rc0 = chol_factor_device(ctx, Gp, m, ld, &inf0);
      if (rc0 == GDML_ERR_NOT_PD || inf0 != 0) return GDML_OK;  // (cannot happen for a Gram matrix that passed the pivot test above)
      GDML_TRY(rc0);
      hipLaunchKernelGGL(eye_minus_lower_kernel, dim3((unsigned)m), dim3(256), 0, st, Gp, Rb, ld, m);
      hipLaunchKernelGGL(syrk_tn_kernel, tri, dim3(256), 0, st, Rb, ld, m, m, Gc, ld, tiles, 0);  // C^T C (lower tiles)
      hipLaunchKernelGGL(sym_fill_kernel, dim3((unsigned)m), dim3(256), 0, st, Gc, ld, m);
      GDML_TRY(launch_gemm_nt_sub(ctx, st, Rb, ld, Rb, ld, Gc, ld, m, m, m, 0));                   // - C C^T
      hipLaunchKernelGGL(vec_axpy_kernel, dim3(ceil_div(m * ld, 256)), dim3(256), 0, st, G0, Gc, 1.0, m * ld);
    }
    // T0 = Zt M0 Zt^T with Zt = L_G^-T = I + E_z.  Formed as G0 + W1 + W1^T + W1 E_z^T, W1 = E_z G0: every product has a
    // small factor (|E_z| ~ 1 - s_min^2), so the MFMA sums carry errors far below eps, and the entries ~1 of T0 come from
    // G0 by three additions -- as a plain triple product the diagonal would lose eps sqrt(m) ~ 5e-15, the scale of the
    // spectrum the matrix exists to carry.
    GDML_TRY(identity(Rb));
    GDML_TRY(tall_trsm(ctx, G, Rb, m, m, ld));
    hipLaunchKernelGGL(sub_eye_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, st, Rb, ld, m);  // Rb = E_z
    double* H = Gc;
    double* Sx = G;
    HIP_CHECK(ctx, hipMemsetAsync(H, 0, m * ld * 8, st));
    GDML_TRY(launch_gemm_nt_sub(ctx, st, Rb, ld, G0, ld, H, ld, m, m, m, 0));   // H = -(E_z G0^T) = -W1
    HIP_CHECK(ctx, hipMemsetAsync(Sx, 0, m * ld * 8, st));
    GDML_TRY(launch_gemm_nt_sub(ctx, st, H, ld, Rb, ld, Sx, ld, m, m, m, 0));   // S = -(H E_z^T) = W1 E_z^T
    hipLaunchKernelGGL(t0_combine_kernel, dim3((unsigned)m), dim3(256), 0, st, ctx->precon_T0, H, Sx, ld, m);
    HIP_CHECK(ctx, hipGetLastError());
    *usable = 1;
    return GDML_OK;
  };
  int rc = body();
  int rc2 = ctx_free(ctx, tmp);
  if (rc != GDML_OK || !*usable) precon_release_f32(ctx);  // rejected form: the fp64 factor is what stays resident
  return rc != GDML_OK ? rc : rc2;
}

// leverage scores = squared row norms of the complete factor X (iterative.py:107-109), replicated on every rank
static int lev_scores_to_host(gdml_ctx* ctx, const ShardGeo& sg, double* lev_scores_out) {
  const int64_t n_loc = sg.n_loc, m = ctx->precon_m, ld = ctx->K_ld;
  double* X = ctx->precon;
  if (ctx->precon_stage < 2) {  // the matrix-free form left X = K_nm L_mm^-T: the second solve (iterative.py:335-345) now
    GDML_TRY(tall_trsm(ctx, ctx->K + n_loc * ld, X, n_loc, m, ld));
    ctx->precon_stage = 2;
  }
  double* d_lev;
  GDML_TRY(ctx_slot(ctx, 2, sg.n_pad * 8, &d_lev));
  if (n_loc > 0)
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3(ceil_div(n_loc, 4)), dim3(256), 0, ctx->stream, X, ld, n_loc, m,
                       d_lev + sg.row0);
  GDML_TRY(comm_allgather_inplace(ctx, d_lev, sg.chunk));
  return vec_download(ctx, sg, d_lev, lev_scores_out);  // reference order
}

extern "C" int gdml_nystroem_lev_scores(gdml_ctx* ctx, double* lev_scores_out) {
  if (!ctx || !lev_scores_out) return GDML_ERR_INVALID;
  if (!ctx->precon || !ctx->K)
    return gdml_fail(ctx, GDML_ERR_STATE, "gdml_nystroem_lev_scores: no Nystroem factor resident (an assembly since "
                                          "gdml_nystroem_factor overwrites it)");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const ShardGeo sg = shard_geo(ctx);
  if (ctx->precon_X32_inplace) {  // the fp64 factor was rounded in place: the scores were taken before that
    if ((int64_t)ctx->precon_lev_cache.size() != sg.n) return gdml_fail(ctx, GDML_ERR_STATE, "gdml_nystroem_lev_scores: no cached scores");
    memcpy(lev_scores_out, ctx->precon_lev_cache.data(), (size_t)sg.n * 8);
    return GDML_OK;
  }
  int rc = lev_scores_to_host(ctx, sg, lev_scores_out);
  if (rc == GDML_ERR_HIP) comm_abort(ctx);
  return rc;
}

extern "C" int gdml_nystroem_factor(gdml_ctx* ctx, double lam, const int64_t* idx, int64_t m,
                                    double* lev_scores_out, double* LinvKmn_host_out, int* info) {
  if (!ctx || !idx || m < 1) return GDML_ERR_INVALID;
  if (!ctx->K || ctx->K_cols != m || ctx->K_extra < m)
    return gdml_fail(ctx, GDML_ERR_STATE,
                     "gdml_nystroem_factor: needs the (n+m) x m matrix of gdml_assemble_K(INDEX, extra=m)");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const ShardGeo sg = shard_geo(ctx);
  const int64_t n_loc = sg.n_loc, ld = ctx->K_ld;
  if (info) *info = 0;
  double* X = ctx->K;               // this rank's rows of K_nm
  double* S = ctx->K + n_loc * ld;  // m x m work block (replicated)
  int form = choose_precon_form(ctx, sg, m, ld);
  // fp32 form: the rounded factor overlays the fp64 one (no 1.5 x footprint; leverage scores are cached first) unless the
  // caller wants the fp64 factor kept (pcg.f32_inplace = 0)
  const bool f32_inplace = ctx_opt_i(ctx, "pcg.f32_inplace", 1) != 0;
  if (ctx->precon_X32_inplace) {  // the overlay of the previous factor went with the assembly that rebuilt this matrix
    ctx->precon_X32 = nullptr;
    ctx->precon_X32_inplace = false;
  }
  ctx->precon_lev_cache.clear();
  // matrix-free form: Z = L_mm^-T L^-T, built by applying to the m x m identity every triangular solve X receives
  double* Z = nullptr;
  if (form == 1) {
    GDML_TRY(ensure_buf(ctx, (void**)&ctx->precon_Z, &ctx->precon_Z_bytes, m * ld * 8));
    Z = ctx->precon_Z;
  }
  GDML_TRY(ensure_buf(ctx, (void**)&ctx->precon_idx, &ctx->precon_idx_bytes, m * 8));
  int64_t* d_idx = ctx->precon_idx;
  void* tmp = nullptr;
  GDML_TRY(ctx_alloc(ctx, &tmp, m * m * 8));
  double* backup = (double*)tmp;
  // the second solve of X is only needed by the stored form, the leverage scores and the host copy
  const bool want_X = form != 1 || lev_scores_out || LinvKmn_host_out;
  int stage = 1;
  int rc = GDML_OK;
  auto body = [&]() -> int {
    HIP_CHECK(ctx, hipMemcpyAsync(d_idx, idx, m * 8, hipMemcpyHostToDevice, ctx->stream));
    phase_begin(ctx);
    if (Z) {
      HIP_CHECK(ctx, hipMemsetAsync(Z, 0, m * ld * 8, ctx->stream));
      hipLaunchKernelGGL(add_diag_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, ctx->stream, Z, ld, m, 1.0);
    }
    // K_mm = -K[idx, :]: every rank contributes the rows it owns, the sum replicates the block
    hipLaunchKernelGGL(gather_neg_rows_sharded_kernel, dim3((unsigned)m), dim3(256), 0, ctx->stream, X, S,
                       ld, d_idx, m, sg);
    GDML_TRY(comm_allreduce_sum(ctx, S, m * ld));
    int ok = 0;
    int n_jit = 0;
    GDML_TRY(cho_factor_stable_dev(ctx, S, m, ld, backup, true, 1, &ok, ctx_opt_i(ctx, "nys.force_fail", 0), &n_jit));  // iterative.py:263
    if (info) *info |= n_jit << 8;
    if (!ok)
      return gdml_fail(ctx, GDML_ERR_NOT_PD,
                       "Failed to factorize despite strong regularization (max: 10)! You could try a larger sigma.");
    GDML_TRY(tall_trsm(ctx, S, X, n_loc, m, ld));  // K_nm <- K_nm L_mm^-T  (iterative.py:276-286)
    if (Z) GDML_TRY(tall_trsm(ctx, S, Z, m, m, ld));
    // inner = K_nm^T K_nm + lam I  (iterative.py:293-294): local SYRK, summed over the shards
    GDML_TRY(gram_tn(ctx, X, ld, n_loc, m, S, ld));
    GDML_TRY(comm_allreduce_sum(ctx, S, m * ld));
    hipLaunchKernelGGL(add_diag_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, ctx->stream, S, ld, m, lam);
    GDML_TRY(cho_factor_stable_dev(ctx, S, m, ld, backup, false, -14, &ok));  // iterative.py:304-306
    if (ok && ctx_opt_i(ctx, "nys.force_qr", 0)) ok = 0;  // test hook: take the alternative branch
    if (ok) {
      if (want_X) {
        GDML_TRY(tall_trsm(ctx, S, X, n_loc, m, ld));  // iterative.py:335-345
        stage = 2;
      }
      if (Z) GDML_TRY(tall_trsm(ctx, S, Z, m, m, ld));
      if (form == 3) {
        int usable = 0;
        GDML_TRY(build_f32_form(ctx, lam, X, S, n_loc, m, ld, &usable, f32_inplace));
        if (!usable) form = 0;
      }
    } else {
      if (form == 3) form = 0;  // no Cholesky factor of the inner matrix to take the exact Gram from: the reference's form
      // "QR fact. (alt.)" (iterative.py:313-324): R of the stacked matrix [K_nm; sqrt(lam) I], i.e. the Cholesky
      // factor of K_nm^T K_nm + lam I obtained WITHOUT trusting the Gram matrix.  Here: shifted CholeskyQR3 -- three
      // rounds of (Gram on fp64 MFMA, small Cholesky, tall triangular solve); the first Gram is shifted so that
      // its factorisation cannot fail, the two repetitions remove the shift's error (R = R3 R2 R1 is never
      // formed: the triangular solves apply R1^-1, R2^-1, R3^-1 to K_nm in turn).  Backward stable like
      // Householder QR for cond([K_nm; sqrt(lam) I]) up to ~1/eps, and GEMM-shaped instead of panel-bound.
      if (info) *info |= 1;
      const double eps = 2.220446049250313e-16;
      double* B = S;  // the stacked sqrt(lam) I block lives right below X: one Gram launch covers both
      HIP_CHECK(ctx, hipMemsetAsync(B, 0, m * ld * 8, ctx->stream));
      hipLaunchKernelGGL(add_diag_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, ctx->stream, B, ld, m, sqrt(lam));
      void* gtmp = nullptr;
      GDML_TRY(ctx_alloc(ctx, &gtmp, m * ld * 8));
      double* G = (double*)gtmp;
      int rcq = GDML_OK;
      for (int pass = 0; pass < 3 && rcq == GDML_OK; ++pass) {
        // the replicated block B enters the Gram sum once: on rank 0 (or when nothing is sharded)
        const int64_t rows = n_loc + ((!ctx->K_sharded || ctx->rank == 0) ? m : 0);
        const int tiles = (int)((m + TT - 1) / TT);
        hipLaunchKernelGGL(syrk_tn_kernel, dim3((unsigned)(tiles * (tiles + 1) / 2)), dim3(256), 0, ctx->stream, X, ld,
                           rows, m, G, ld, tiles, 0);
        rcq = comm_allreduce_sum(ctx, G, m * ld);
        if (rcq != GDML_OK) break;
        if (pass == 0) {
          std::vector<double> diag((size_t)m);
          hipError_t e = hipMemcpy2DAsync(diag.data(), 8, G, (ld + 1) * 8, 8, m, hipMemcpyDeviceToHost, ctx->stream);
          if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
          if (e != hipSuccess) {
            rcq = gdml_fail(ctx, GDML_ERR_HIP, "Gram diagonal: %s", hipGetErrorString(e));
            break;
          }
          double tr = 0.0;
          for (double d : diag) tr += d;  // ||A||_2^2 <= ||A||_F^2 = trace(A^T A)
          const double rows_tot = (double)sg.n + (double)m;
          const double shift = 11.0 * ((double)m * rows_tot + (double)m * ((double)m + 1.0)) * eps * tr;
          hipLaunchKernelGGL(add_diag_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, ctx->stream, G, ld, m, shift);
        }
        int inf = 0;
        rcq = chol_factor_device(ctx, G, m, ld, &inf);
        if (rcq == GDML_OK && inf != 0)
          rcq = gdml_fail(ctx, GDML_ERR_NOT_PD, "Nystroem factor: K_nm^T K_nm + lam I is singular to working precision "
                                                "(shifted CholeskyQR pass %d failed at pivot %d)", pass + 1, inf);
        if (rcq == GDML_OK) rcq = tall_trsm(ctx, G, X, n_loc + m, m, ld);  // X and B (every rank its own copy of B)
        if (rcq == GDML_OK && Z) rcq = tall_trsm(ctx, G, Z, m, m, ld);
      }
      stage = 2;  // the three factors are gone after this branch: X is always completed here
      int rcf = ctx_free(ctx, gtmp);
      GDML_TRY(rcq);
      GDML_TRY(rcf);
    }
    GDML_TRY(phase_end(ctx, "precon"));
    return GDML_OK;
  };
  rc = body();
  if (rc == GDML_ERR_HIP) comm_abort(ctx);  // a LOCAL failure (errors computed from replicated data hit every rank alike)
  if (rc == GDML_OK) {
    ctx->precon = X;
    ctx->precon_m = m;
    ctx->precon_n = sg.n;
    ctx->precon_form = form;
    ctx->precon_stage = stage;
    ctx->precon_sig = ctx->K_sig;
    ctx->precon_use_E = ctx->K_use_E;
    if (info && form == 1) *info |= 2;
    if (info && form == 3) *info |= 4;
    if (lev_scores_out) {
      rc = lev_scores_to_host(ctx, sg, lev_scores_out);
      if (rc == GDML_ERR_HIP) comm_abort(ctx);
    }
    if (rc == GDML_OK && LinvKmn_host_out) {
      // host gets this rank's columns of L^-1 K_mn as an (m x n_loc) array: transpose of X
      std::vector<double> h((size_t)n_loc * m);
      hipError_t e = hipMemcpy2DAsync(h.data(), m * 8, X, ld * 8, m * 8, n_loc, hipMemcpyDeviceToHost,
                                      ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess)
        rc = gdml_fail(ctx, GDML_ERR_HIP, "factor copy: %s", hipGetErrorString(e));
      else
        for (int64_t r = 0; r < n_loc; ++r)
          for (int64_t c = 0; c < m; ++c) LinvKmn_host_out[c * n_loc + r] = h[(size_t)r * m + c];
    }
    ctx->precon_lev_cache.clear();
    ctx->precon_X32_ld = ld;
    if (rc == GDML_OK && form == 3 && f32_inplace) {
      // everything that needs the fp64 factor has happened; the leverage scores (restart policy, iterative.py:107-109) are
      // taken now and served from the cache, then X is rounded where it stands
      ctx->precon_lev_cache.resize((size_t)sg.n);
      rc = lev_scores_to_host(ctx, sg, ctx->precon_lev_cache.data());
      if (rc == GDML_ERR_HIP) comm_abort(ctx);
      if (rc == GDML_OK) {
        if (n_loc > 0) hipLaunchKernelGGL(round_f32_inplace_kernel, dim3((unsigned)n_loc), dim3(256), 0, ctx->stream, X, ld, m);
        ctx->precon_X32 = reinterpret_cast<float*>(X);
        ctx->precon_X32_bytes = 0;
        ctx->precon_X32_inplace = true;
        ctx->precon_X32_ld = 2 * ld;
        if (hipGetLastError() != hipSuccess) rc = gdml_fail(ctx, GDML_ERR_HIP, "in-place rounding of the factor failed to launch");
      }
    }
  }
  int rc2 = ctx_free(ctx, tmp);
  return rc != GDML_OK ? rc : rc2;
}

// Matrix-free form: d_out = (K_nm Z Z^T K_mn d_v - d_v)/lam.  The n x m factor X = K_nm Z is never read: K_mn v is the
// kernel mat-vec followed by a gather of the m inducing entries, K_nm t a scatter into an n-vector followed by the
// mat-vec -- 2 x 10 M^2 P D flops and 16 m^2 bytes per application instead of 16 n m bytes (at configs[2]: 2 x 1.0 ms +
// 0.3 ms against 7.8 ms).  Sharded: the mat-vecs are query-sharded and all-gather their result; Z is replicated.
static int precon_apply_mf(gdml_ctx* ctx, double lam, const double* d_v, double* d_out) {
  const int64_t m = ctx->precon_m, n = ctx->precon_n, ld = ctx->K_ld;
  if (!ctx->precon_Z || !ctx->precon_idx) return gdml_fail(ctx, GDML_ERR_STATE, "matrix-free preconditioner not resident");
  if (ctx->world > 1 && ctx->precon_use_E)
    return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "matrix-free preconditioner: not with sharded energy constraints");
  Model& md = ctx->model;
  if (!md.xp || md.M != ctx->ts.M || md.N != ctx->ts.N || md.P != ctx->ts.P || md.sig != ctx->precon_sig)
    GDML_TRY(operator_model_from_trainset(ctx, ctx->precon_sig));
  int64_t n_pad = n;
  if (ctx->world > 1) {
    int64_t p0, p1, per;
    shard_points(ctx, ctx->ts.M, &p0, &p1, &per);
    n_pad = per * 3 * ctx->ts.N * ctx->world;
    if (n_pad < n) n_pad = n;
  }
  n_pad = (n_pad + 31) / 32 * 32;
  const int64_t m_pad = (m + 31) / 32 * 32;
  const int rows_per = 512;
  const int nparts = (int)((m + rows_per - 1) / rows_per);
  double* buf;
  GDML_TRY(ctx_slot(ctx, 10, (2 * n_pad + 2 * m_pad + (int64_t)nparts * m_pad) * 8, &buf));
  double* s = buf;            // K v, later K e
  double* e = s + n_pad;      // scattered m-vector
  double* t1 = e + n_pad;
  double* t2 = t1 + m_pad;
  double* part = t2 + m_pad;
  hipStream_t st = ctx->stream;
  const double* Z = ctx->precon_Z;
  GDML_TRY(matvec_device(ctx, 0.0, ctx->precon_use_E, d_v, n, s));
  hipLaunchKernelGGL(gather_idx_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, st, s, ctx->precon_idx, m, t1);
  hipLaunchKernelGGL(gemv_t_part_kernel<false>, dim3(ceil_div(m, 512), nparts), dim3(256), 0, st, Z, ld, m, m, t1, rows_per,
                     part);
  hipLaunchKernelGGL(reduce_parts_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, st, part, m, nparts, t2);
  hipLaunchKernelGGL((gemv_n_precon_kernel<false, false>), dim3(ceil_div(m, 4)), dim3(256), 0, st, Z, ld, m, m, t2,
                     (const double*)nullptr, 1.0, t1);
  HIP_CHECK(ctx, hipMemsetAsync(e, 0, n_pad * 8, st));
  hipLaunchKernelGGL(scatter_idx_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, st, t1, ctx->precon_idx, m, e);
  GDML_TRY(matvec_device(ctx, 0.0, ctx->precon_use_E, e, n, s));
  hipLaunchKernelGGL(precon_finish_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, s, d_v, 1.0 / lam, n, d_out);
  ctx->launch_counter += 7;
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}

// d_out = (X X^T d_v - d_v)/lam on replicated (padded) device vectors; X row-sharded
static int precon_apply_device(gdml_ctx* ctx, double lam, const double* d_v, double* d_out) {
  if (!ctx->precon) return gdml_fail(ctx, GDML_ERR_STATE, "no preconditioner resident");
  if (ctx->precon_form == 1) return precon_apply_mf(ctx, lam, d_v, d_out);
  const ShardGeo sg = shard_geo(ctx);
  const int64_t n_loc = sg.n_loc, m = ctx->precon_m, ld = ctx->K_ld;
  if (ctx->precon_form == 3) {
    // (X32 T0 X32^T v - v)/lam: the two passes read n_loc x m x 4 bytes each; T0 (m x m, fp64, replicated) in between
    if (!ctx->precon_X32 || !ctx->precon_T0) return gdml_fail(ctx, GDML_ERR_STATE, "fp32 preconditioner not resident");
    const bool nt = ctx_opt_i(ctx, "pcg.gemv_plain", 0) == 0;
    int rows_per = ctx_opt_i(ctx, "pcg.f32_rows_per", 1024);
    if (rows_per < 64) rows_per = 64;
    int rw = ctx_opt_i(ctx, "pcg.f32_rw", 4);
    int nparts = (int)((n_loc + rows_per - 1) / rows_per);
    if (nparts < 1) nparts = 1;
    double* buf;
    GDML_TRY(ctx_slot(ctx, 3, ((int64_t)nparts * m + 2 * ld + 4) * 8, &buf));
    double* part = buf;
    double* t = buf + (((int64_t)nparts * m + 1) & ~(int64_t)1);
    double* u = t + ld;
    const int slot = ktime_begin(ctx);
    if (n_loc > 0) {
      if (nt)
        hipLaunchKernelGGL(gemv_t_part_f32_kernel<true>, dim3(ceil_div(m, 1024), nparts), dim3(256), 0, ctx->stream,
                           ctx->precon_X32, ctx->precon_X32_ld, n_loc, m, d_v + sg.row0, rows_per, part);
      else
        hipLaunchKernelGGL(gemv_t_part_f32_kernel<false>, dim3(ceil_div(m, 1024), nparts), dim3(256), 0, ctx->stream,
                           ctx->precon_X32, ctx->precon_X32_ld, n_loc, m, d_v + sg.row0, rows_per, part);
      hipLaunchKernelGGL(reduce_parts_kahan_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, ctx->stream, part, m, nparts, t);
    } else {
      HIP_CHECK(ctx, hipMemsetAsync(t, 0, m * 8, ctx->stream));
    }
    GDML_TRY(comm_allreduce_sum(ctx, t, m));
    HIP_CHECK(ctx, hipMemsetAsync(u, 0, ld * 8, ctx->stream));  // pad entries [m, m4) stay zero
    hipLaunchKernelGGL((gemv_n_precon_kernel<false, false>), dim3(ceil_div(m, 4)), dim3(256), 0, ctx->stream, ctx->precon_T0,
                       ld, m, m, t, (const double*)nullptr, 1.0, u);
    if (n_loc > 0) {
      const float* X32 = ctx->precon_X32;
      const double *vv = d_v + sg.row0;
      double* oo = d_out + sg.row0;
      const double il = 1.0 / lam;
#define GDML_GEMV_N_F32(NT_, RW_)                                                                                          \
  hipLaunchKernelGGL((gemv_n_precon_f32_kernel<NT_, RW_>), dim3(ceil_div(n_loc, 4 * RW_)), dim3(256), 0, ctx->stream, X32, \
                     ctx->precon_X32_ld, n_loc, m, u, vv, il, oo)
      if (rw >= 4) { if (nt) GDML_GEMV_N_F32(true, 4); else GDML_GEMV_N_F32(false, 4); }
      else if (rw >= 2) { if (nt) GDML_GEMV_N_F32(true, 2); else GDML_GEMV_N_F32(false, 2); }
      else { if (nt) GDML_GEMV_N_F32(true, 1); else GDML_GEMV_N_F32(false, 1); }
#undef GDML_GEMV_N_F32
    }
    ktime_end(ctx, slot, "precon_gemv", 2.0 * 4.0 * (double)n_loc * (double)m + 8.0 * (double)m * (double)m);
    ctx->launch_counter += 5;
    HIP_CHECK(ctx, hipGetLastError());
    return comm_allgather_inplace(ctx, d_out, sg.chunk);
  }
  // non-temporal loads of the factor (it is streamed: 25 GB per pass at configs[2]): PCG iteration 9.66 -> 9.01 ms
  // (profiles/r04_gemv_nt_ab.txt); pcg.gemv_plain = 1: plain loads (A/B)
  const bool nt = ctx_opt_i(ctx, "pcg.gemv_plain", 0) == 0;
  int rows_per = 2048;
  int nparts = (int)((n_loc + rows_per - 1) / rows_per);
  if (nparts < 1) nparts = 1;
  double* buf;
  GDML_TRY(ctx_slot(ctx, 3, ((int64_t)nparts * m + m + 2) * 8, &buf));
  double* part = buf;
  double* t = buf + (((int64_t)nparts * m + 1) & ~(int64_t)1);  // 16-byte aligned: read in pairs
  const int kslot = ktime_begin(ctx);
  if (n_loc > 0) {
    if (nt)
      hipLaunchKernelGGL(gemv_t_part_kernel<true>, dim3(ceil_div(m, 512), nparts), dim3(256), 0, ctx->stream,
                         ctx->precon, ld, n_loc, m, d_v + sg.row0, rows_per, part);
    else
      hipLaunchKernelGGL(gemv_t_part_kernel<false>, dim3(ceil_div(m, 512), nparts), dim3(256), 0, ctx->stream,
                         ctx->precon, ld, n_loc, m, d_v + sg.row0, rows_per, part);
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, ctx->stream, part, m,
                       nparts, t);
  } else {
    HIP_CHECK(ctx, hipMemsetAsync(t, 0, m * 8, ctx->stream));
  }
  GDML_TRY(comm_allreduce_sum(ctx, t, m));
  if (n_loc > 0) {
    if (nt)
      hipLaunchKernelGGL(gemv_n_precon_kernel<true>, dim3(ceil_div(n_loc, 4)), dim3(256), 0, ctx->stream, ctx->precon, ld,
                         n_loc, m, t, d_v + sg.row0, 1.0 / lam, d_out + sg.row0);
    else
      hipLaunchKernelGGL(gemv_n_precon_kernel<false>, dim3(ceil_div(n_loc, 4)), dim3(256), 0, ctx->stream, ctx->precon, ld,
                         n_loc, m, t, d_v + sg.row0, 1.0 / lam, d_out + sg.row0);
  }
  ktime_end(ctx, kslot, "precon_gemv", 2.0 * 8.0 * (double)n_loc * (double)m);
  ctx->launch_counter += 3;
  HIP_CHECK(ctx, hipGetLastError());
  return comm_allgather_inplace(ctx, d_out, sg.chunk);
}

extern "C" int gdml_precon_apply(gdml_ctx* ctx, double lam, const double* v, int64_t n, double* out) {
  if (!ctx || !v || !out) return GDML_ERR_INVALID;
  if (!ctx->precon || n != ctx->precon_n)
    return gdml_fail(ctx, GDML_ERR_STATE, "gdml_precon_apply: no matching preconditioner resident");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const ShardGeo sg = shard_geo(ctx);
  void* buf = nullptr;
  GDML_TRY(ctx_alloc(ctx, &buf, 2 * sg.n_pad * 8));
  double* dv = (double*)buf;
  double* dout = dv + sg.n_pad;
  int rc = GDML_OK;
  hipError_t e = hipMemsetAsync(dv, 0, 2 * sg.n_pad * 8, ctx->stream);
  if (e != hipSuccess) rc = gdml_fail(ctx, GDML_ERR_HIP, "memset: %s", hipGetErrorString(e));
  if (rc == GDML_OK) rc = vec_upload(ctx, sg, v, dv);
  if (rc == GDML_OK) rc = precon_apply_device(ctx, lam, dv, dout);
  if (rc == GDML_OK) rc = vec_download(ctx, sg, dout, out);
  int rc2 = ctx_free(ctx, buf);
  return rc != GDML_OK ? rc : rc2;
}

// Preconditioned CG for A x = y, A v = -(K v - lam v)  (iterative.py:740-752; scipy cg: stop when ||r|| < rtol*||y||,
// atol = 0).  All vectors AND the CG scalars stay on the device (replicated on every rank when sharded: the mat-vec and the
// preconditioner all-gather their row shards, dot products are evaluated redundantly and identically).
//
// The host never waits for the iteration it has just queued.  Every iteration ends by publishing ||r||^2 into a ring of
// host-mapped slots; the host reads the residual of iteration s only when it is about to queue iteration s + DEPTH
// (option pcg.depth, default 2), so DEPTH iterations are always in flight and the convergence test / callback of scipy's
// loop are evaluated with that lag -- but with exactly scipy's semantics: the iterates x_s ... x_{s+DEPTH} live in a ring of
// DEPTH + 1 device vectors, and when the test (or the callback) for iteration s says stop, x_s is what is returned; the
// at most DEPTH iterations queued beyond it are discarded.  Decisions derive from replicated scalars, so every rank of a
// sharded run queues the same sequence of collectives.
struct PcgRun {
  int64_t n = 0;
  int depth = 0, ring = 0;
  double* xring = nullptr;  // (depth + 1) x n_pad iterates
  int64_t n_pad = 0;
  int64_t cb_iter = -1;     // iteration the callback is reporting (gdml_pcg_x), -1 outside a callback
  VecLayout lay;            // order of the device vectors (reference order unless sharded with energy constraints)
};
static thread_local PcgRun* g_pcg_run = nullptr;
static thread_local gdml_ctx* g_pcg_ctx = nullptr;

extern "C" int gdml_pcg_x(gdml_ctx* ctx, double* x_host_out) {
  if (!ctx || !x_host_out) return GDML_ERR_INVALID;
  PcgRun* run = g_pcg_run;
  if (!run || g_pcg_ctx != ctx || run->cb_iter < 0)
    return gdml_fail(ctx, GDML_ERR_STATE, "gdml_pcg_x: only valid inside the callback of a running gdml_pcg");
  // x_s is complete (its residual has been read) and its slot is not rewritten before the callback returns: copy it on the
  // second stream, past the iterations queued on the compute stream
  const double* xs = run->xring + (run->cb_iter % (run->depth + 1)) * run->n_pad;
  return vec_download(ctx, run->lay, xs, x_host_out, ctx->stream2);
}

extern "C" int gdml_pcg(gdml_ctx* ctx, double lam, int use_E_cstr, const double* y, const double* x0,
                        int64_t n, double rtol, int64_t maxiter, int use_precon, gdml_pcg_cb cb,
                        int64_t cb_every, void* user, double* x_out, int64_t* iters_out,
                        double* resid_out, int* info_out) {
  if (!ctx || !y || !x_out) return GDML_ERR_INVALID;
  if (use_precon && (!ctx->precon || ctx->precon_n != n))
    return gdml_fail(ctx, GDML_ERR_STATE, "gdml_pcg: no matching preconditioner resident");
  if (!ctx->model.xp || !ctx->ts.x)
    return gdml_fail(ctx, GDML_ERR_STATE,
                     "gdml_pcg: training set and operator model must be resident "
                     "(gdml_train_upload + gdml_predict_upload_model)");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // device vectors in the layout of VecLayout (common.h).  Sharded with energy constraints the entries are rank-major with
  // zero padding inside: the vector kernels then run over all n_pad entries (nv) instead of the first n
  const VecLayout lay = vec_layout(ctx, use_E_cstr);
  if (lay.n != n) return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_pcg: n mismatch");
  if (use_precon && ctx->world > 1 && (ctx->precon_use_E != 0) != (use_E_cstr != 0))
    return gdml_fail(ctx, GDML_ERR_STATE, "gdml_pcg: the resident preconditioner was built with another use_E_cstr");
  const int64_t nv = lay.two_seg ? lay.n_pad : n;
  int64_t n_pad = (lay.n_pad + 31) / 32 * 32;
  int depth = ctx_opt_i(ctx, "pcg.depth", 2);
  if (depth < 0) depth = 0;
  if (depth > 14) depth = 14;
  // the host-staged collectives synchronise the stream inside every mat-vec anyway: nothing to pipeline
  if (ctx->host_allreduce) depth = 0;
  const int XR = depth + 1, RING = depth + 2;
  // device: [x ring | r | z | p | q | b | partials rho[2], pq, rr | rr_dev | counter]
  const int64_t n_vec = XR + 5;
  const int64_t tail = 4 * PCG_PARTS + 64;
  void* buf = nullptr;
  GDML_TRY(ctx_alloc(ctx, &buf, (n_vec * n_pad + tail) * 8));
  double* xring = (double*)buf;
  double* r = xring + XR * n_pad;
  double* z = r + n_pad;
  double* p = z + n_pad;
  double* q = p + n_pad;
  double* b = q + n_pad;
  double* part_rho[2] = {b + n_pad, b + n_pad + PCG_PARTS};
  double* part_pq = part_rho[1] + PCG_PARTS;
  double* part_rr = part_pq + PCG_PARTS;
  double* rr_dev = part_rr + PCG_PARTS;
  unsigned* counter = (unsigned*)(rr_dev + 8);
  // host-mapped residual ring + one event per slot
  double* h_rr = nullptr;
  std::vector<hipEvent_t> ev((size_t)RING, nullptr);
  PcgRun run;
  run.n = n; run.depth = depth; run.ring = RING; run.xring = xring; run.n_pad = n_pad; run.lay = lay;
  int rc = GDML_OK, info = 1;
  int64_t it = 0;
  double rn = 0.0;
  const int64_t x_final_slot_none = -1;
  int64_t x_final = x_final_slot_none;  // iterate index to return
  auto xs = [&](int64_t s) { return xring + (s % XR) * n_pad; };
  auto body = [&]() -> int {
    HIP_CHECK(ctx, hipHostMalloc((void**)&h_rr, RING * 8, hipHostMallocMapped));
    double* h_rr_dev = nullptr;
    HIP_CHECK(ctx, hipHostGetDevicePointer((void**)&h_rr_dev, h_rr, 0));
    for (auto& e : ev) HIP_CHECK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipStream_t st = ctx->stream;
    HIP_CHECK(ctx, hipMemsetAsync(buf, 0, (n_vec * n_pad + tail) * 8, st));
    GDML_TRY(vec_upload(ctx, lay, y, b));
    // ||b||: the one synchronous read of the solve
    hipLaunchKernelGGL(dot_part_kernel, dim3(PCG_PARTS), dim3(256), 0, st, b, b, nv, part_rr);
    hipLaunchKernelGGL(pcg_publish_kernel, dim3(1), dim3(256), 0, st, part_rr, rr_dev, h_rr_dev);
    HIP_CHECK(ctx, hipStreamSynchronize(st));
    const double bnrm = sqrt(h_rr[0]);
    if (bnrm == 0.0) {
      info = 0;
      x_final = 0;  // x ring slot 0 is all zeros
      return GDML_OK;
    }
    const double atol = rtol * bnrm;
    double* x = xs(0);
    if (x0) {
      GDML_TRY(vec_upload(ctx, lay, x0, x));
      GDML_TRY(matvec_device(ctx, lam, use_E_cstr, x, n, q));  // q = K x - lam x = -A x
      HIP_CHECK(ctx, hipMemcpyAsync(r, b, nv * 8, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(vec_axpy_kernel, dim3(ceil_div(nv, 256)), dim3(256), 0, st, r, q, 1.0, nv);  // r = b - A x
    } else {
      HIP_CHECK(ctx, hipMemcpyAsync(r, b, nv * 8, hipMemcpyDeviceToDevice, st));
    }
    hipLaunchKernelGGL(dot_part_kernel, dim3(PCG_PARTS), dim3(256), 0, st, r, r, nv, part_rr);
    hipLaunchKernelGGL(pcg_publish_kernel, dim3(1), dim3(256), 0, st, part_rr, rr_dev, h_rr_dev + 0);
    HIP_CHECK(ctx, hipEventRecord(ev[0], st));

    int64_t enq = 0, seen = 0;  // iterations queued; residuals ||r_0|| .. ||r_{seen-1}|| processed
    // processes ||r_s||: callback of iteration s (s >= 1), then scipy's test at the top of iteration s.  Returns 1 = stop.
    auto process = [&](int64_t s, int* stop) -> int {
      HIP_CHECK(ctx, hipEventSynchronize(ev[s % RING]));
      rn = sqrt(h_rr[s % RING]);
      *stop = 0;
      if (s >= 1 && cb && cb_every > 0 && (s % cb_every) == 0) {
        run.cb_iter = s;
        const int stop_cb = cb(s, rn, user);
        run.cb_iter = -1;
        if (stop_cb != 0) {
          info = 2;
          it = s;
          x_final = s;
          *stop = 1;
          return GDML_OK;
        }
      }
      if (rn < atol) {
        info = 0;
        it = s;
        x_final = s;
        *stop = 1;
      }
      return GDML_OK;
    };
    for (;;) {
      int stop = 0;
      while (seen <= enq - depth || (enq >= maxiter && seen <= enq)) {
        GDML_TRY(process(seen, &stop));
        if (stop) return GDML_OK;
        ++seen;
      }
      if (enq >= maxiter) {  // every residual up to ||r_maxiter|| has been tested
        info = 1;
        it = maxiter;
        x_final = maxiter;
        return GDML_OK;
      }
      // ---- queue iteration enq: x_{enq} -> x_{enq+1}
      const int cur = (int)(enq & 1);
      const double* zz = r;
      if (use_precon) {
        GDML_TRY(precon_apply_device(ctx, lam, r, z));
        zz = z;
      }
      hipLaunchKernelGGL(dot_part_kernel, dim3(PCG_PARTS), dim3(256), 0, st, r, zz, nv, part_rho[cur]);
      hipLaunchKernelGGL(pcg_p_update_kernel, dim3(PCG_PARTS), dim3(256), 0, st, p, zz, part_rho[cur], part_rho[cur ^ 1],
                         enq == 0 ? 1 : 0, nv);
      GDML_TRY(matvec_device(ctx, lam, use_E_cstr, p, n, q));  // q = -(A p)
      hipLaunchKernelGGL(dot_part_kernel, dim3(PCG_PARTS), dim3(256), 0, st, p, q, nv, part_pq);
      hipLaunchKernelGGL(pcg_xr_update_kernel, dim3(PCG_PARTS), dim3(256), 0, st, xs(enq), xs(enq + 1), r, p, q,
                         part_rho[cur], part_pq, part_rr, counter, rr_dev, h_rr_dev + ((enq + 1) % RING), nv);
      ctx->launch_counter += 4;
      HIP_CHECK(ctx, hipGetLastError());
      HIP_CHECK(ctx, hipEventRecord(ev[(enq + 1) % RING], st));
      ++enq;
    }
  };
  phase_begin(ctx);
  g_pcg_run = &run;
  g_pcg_ctx = ctx;
  rc = body();
  g_pcg_run = nullptr;
  g_pcg_ctx = nullptr;
  if (rc == GDML_ERR_HIP) comm_abort(ctx);  // a LOCAL failure (errors computed from replicated data hit every rank alike)
  if (rc == GDML_OK) rc = phase_end(ctx, "pcg");
  if (rc == GDML_OK) {
    // the iterations queued beyond x_final are discarded; they still have to drain before the buffers go away
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = gdml_fail(ctx, GDML_ERR_HIP, "sync: %s", hipGetErrorString(e));
    if (rc == GDML_OK) rc = vec_download(ctx, lay, xs(x_final < 0 ? 0 : x_final), x_out);
  } else {
    (void)hipStreamSynchronize(ctx->stream);
  }
  for (auto e : ev)
    if (e) (void)hipEventDestroy(e);
  if (h_rr) (void)hipHostFree(h_rr);
  if (iters_out) *iters_out = it;
  if (resid_out) *resid_out = rn;
  if (info_out) *info_out = info;
  int rc2 = ctx_free(ctx, buf);
  return rc != GDML_OK ? rc : rc2;
}
