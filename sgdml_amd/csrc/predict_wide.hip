// Batched prediction contraction for large molecules (D = N(N-1)/2 > 256, i.e. N > 23) on the MFMA pipe.
//
// Same math as predict_mfma_kernel (sgdml/predict.py:84-245; torch batching torchtools.py:877-1046), but the
// 16 x D accumulators and the query operand of that kernel no longer fit a wavefront's registers (D = 861 for 42
// atoms, 1770 for 60), so the two contractions run as tiled GEMMs with the (table row, query) scalars in HBM:
//   1. S = X_p X_q^T,  T = JA_p X_q^T                 (MP x B each, K = D)      gemm_nt_sub (chol.hip)
//   2. per pair: |d|^2 = |x_q|^2 + |X_r|^2 - 2 S (clamped), a = T - X_r.JA_r, the Matern scalars; S <- w1, T <- b2;
//      column sums of w1 and of the energy terms                                matern_pairs_kernel
//   3. F_x = x_q * sum_r w1 - W1^T X_p - B2^T JA_p     (B x D, K = MP)           gemm_tn_split_kernel + reduce_tn_kernel (here)
//   4. F = J_x^T F_x, E                                                          predict_epilogue_kernel (JS = 1)
// Algorithmic flops 8 MP B D on v_mfma_f64_16x16x4_f64 against ~10 MP B D on the VALU for the wave kernel; the pair
// scalars cost 4 x 8 MP B bytes of HBM traffic per call (written and read once each), processed in query chunks so
// that the work buffers stay below ~2 GB.  The |d|^2 expansion cancels for coincident points; every Matern quantity
// multiplying O(1) data is second order in the distance there (same argument and same 1e-11 parity tolerance as
// the D <= 256 MFMA kernel).
#include <math.h>

#include "common.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

namespace {

#define WT 128
#define WBK 16
#define WP 144

__device__ __forceinline__ void w_load(const double* __restrict__ X, int64_t ld, int64_t nk, int64_t ncol, int64_t k0,
                                       int64_t c0, int tid, d2 (&r)[4]) {
  // 16 k-rows x 128 columns = 1024 chunks of 2 doubles: chunk c -> k-row c/64, column pair c%64
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int cidx = tid + 256 * s;
    const int kr = cidx >> 6, cc = (cidx & 63) * 2;
    const int64_t gk = k0 + kr, gc = c0 + cc;
    d2 v = {0.0, 0.0};
    if (gk < nk) {
      if (gc < ncol) v.x = X[gk * ld + gc];
      if (gc + 1 < ncol) v.y = X[gk * ld + gc + 1];
    }
    r[s] = v;
  }
}
__device__ __forceinline__ void w_store(double* __restrict__ S, int tid, const d2 (&r)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int cidx = tid + 256 * s;
    const int kr = cidx >> 6, cc = (cidx & 63) * 2;
    S[kr * WP + cc] = r[s].x;
    S[kr * WP + cc + 1] = r[s].y;
  }
}

// P_z[m x n] = A^T B over this split's share of the nk rows;  A: nk x m (lda), B: nk x n (ldb), row-major.  128 x 128 tiles,
// 16 rows of A / B per k-tile, fp64 MFMA.  Both operand tiles are 16 x 128 blocks with contiguous rows: interior tiles move
// them in 16-byte pieces (check-free global_load_dwordx4 from a wave-uniform base, ds_write_b128 into an unpadded [k][128]
// image, one ds_read_b128 per PAIR of operand blocks -- the column-pair relabelling of syrk_tn_kernel, cg.hip).
//
// SPLIT-K: the back contraction of the prediction has a short output (queries x D: 16 x 7 tiles at configs[3]) and a very
// long contraction (the M P table rows: 54 000), so one workgroup per tile leaves most of the chip idle (112 workgroups
// for 512 slots: 17.8 of the 41 ms of a configs[3] PCG iteration, profiles/r04_cg_shape_kernels.txt).  Each of gridDim.z
// workgroups per tile takes a contiguous share of the k-tiles and writes its partial tile; reduce_tn_kernel subtracts the
// partials from C in a fixed order (deterministic: no atomics).
__device__ __forceinline__ void w_load_u(const double* __restrict__ X, int64_t ld, int64_t k0, int64_t c0, const unsigned (&off)[4],
                                         d2 (&r)[4]) {
  typedef const __attribute__((address_space(1))) char* gcptr;
  typedef const __attribute__((address_space(1))) d2* gd2ptr;
  const uint64_t p = reinterpret_cast<uint64_t>(X + k0 * ld + c0);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
  gcptr b = (gcptr)(((uint64_t)hi << 32) | lo);
#pragma unroll
  for (int s = 0; s < 4; ++s) r[s] = *(gd2ptr)(b + off[s]);
}
__device__ __forceinline__ void w_store16(double* __restrict__ S, int tid, const d2 (&r)[4]) {
  d2* S2 = reinterpret_cast<d2*>(S);
#pragma unroll
  for (int s = 0; s < 4; ++s) S2[tid + 256 * s] = r[s];
}

template <bool FULL>
__device__ __forceinline__ void gemm_tn_tile(const double* __restrict__ A, int64_t lda, const double* __restrict__ B, int64_t ldb,
                                             double* __restrict__ Pz, int64_t m, int64_t n, int64_t nk, int64_t kt0, int64_t kt1,
                                             int64_t row0, int64_t col0, double (*lds)[2][WBK * WT]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lk = lane >> 4;
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  unsigned offA[4], offB[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int cidx = tid + 256 * s;
    offA[s] = (unsigned)(((int64_t)(cidx >> 6) * lda + (cidx & 63) * 2) * 8);
    offB[s] = (unsigned)(((int64_t)(cidx >> 6) * ldb + (cidx & 63) * 2) * 8);
  }
  const int64_t kt_full = FULL ? nk / WBK : 0;  // k-tiles that can be loaded without checks
  d2 ra[4], rb[4];
  auto load = [&](int64_t kt) {
    if (kt < kt_full) {
      w_load_u(A, lda, kt * WBK, row0, offA, ra);
      w_load_u(B, ldb, kt * WBK, col0, offB, rb);
    } else {
      w_load(A, lda, nk, m, kt * WBK, row0, tid, ra);
      w_load(B, ldb, nk, n, kt * WBK, col0, tid, rb);
    }
  };
  if (kt0 < kt1) {
    load(kt0);
    w_store16(lds[0][0], tid, ra);
    w_store16(lds[0][1], tid, rb);
  }
  __syncthreads();
  for (int64_t kt = kt0; kt < kt1; ++kt) {
    const int cur = (int)((kt - kt0) & 1);
    if (kt + 1 < kt1) load(kt + 1);
    const d2* Ap = reinterpret_cast<const d2*>(lds[cur][0]) + lk * (WT / 2) + wm * 32 + li;
    const d2* Bp = reinterpret_cast<const d2*>(lds[cur][1]) + lk * (WT / 2) + wn * 32 + li;
#pragma unroll
    for (int ks = 0; ks < WBK; ks += 4) {
      const d2 a01 = Ap[ks * (WT / 2)], a23 = Ap[ks * (WT / 2) + 16];
      const d2 b01 = Bp[ks * (WT / 2)], b23 = Bp[ks * (WT / 2) + 16];
      const double a[4] = {a01.x, a01.y, a23.x, a23.y};
      const double bb[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
      if (ks == 8 && kt + 1 < kt1) {
        w_store16(lds[cur ^ 1][0], tid, ra);
        w_store16(lds[cur ^ 1][1], tid, rb);
      }
    }
    __syncthreads();
  }
  // accumulator (i, j), register r of lane (li, lk): row = 64 wm + 32 (i >> 1) + 2 (lk + 4 r) + (i & 1), column likewise with li
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t gc = col0 + wn * 64 + 32 * (j >> 1) + 2 * li + (j & 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t gr = row0 + wm * 64 + 32 * (i >> 1) + 2 * (lk + 4 * r) + (i & 1);
        if (gr < m && gc < n) Pz[gr * n + gc] = acc[i][j][r];
      }
    }
}

// a_cols, b_cols: columns of A / B that may be READ without checks (>= m, n: memory that exists behind the operands' used
// columns; what is read there only reaches output entries that are not stored)
__global__ void __launch_bounds__(256, 2) gemm_tn_split_kernel(const double* __restrict__ A, int64_t lda, int64_t a_cols,
                                                               const double* __restrict__ B, int64_t ldb, int64_t b_cols,
                                                               double* __restrict__ P, int64_t m, int64_t n, int64_t nk) {
  __shared__ __attribute__((aligned(16))) double lds[2][2][WBK * WT];
  const int64_t row0 = (int64_t)blockIdx.y * WT, col0 = (int64_t)blockIdx.x * WT;
  const int64_t nt = (nk + WBK - 1) / WBK;
  const int64_t kt0 = nt * blockIdx.z / gridDim.z, kt1 = nt * (blockIdx.z + 1) / gridDim.z;
  double* Pz = P + (int64_t)blockIdx.z * m * n;
  const bool aligned = (((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0) && (lda % 2 == 0) &&
                       (ldb % 2 == 0) && 16 * lda * 8 < ((int64_t)1 << 32) && 16 * ldb * 8 < ((int64_t)1 << 32);
  if (aligned && row0 + WT <= a_cols && col0 + WT <= b_cols)
    gemm_tn_tile<true>(A, lda, B, ldb, Pz, m, n, nk, kt0, kt1, row0, col0, lds);
  else
    gemm_tn_tile<false>(A, lda, B, ldb, Pz, m, n, nk, kt0, kt1, row0, col0, lds);
}

// C[e] -= sum_z P[z][e], z in ascending order
__global__ void __launch_bounds__(256) reduce_tn_kernel(const double* __restrict__ P, int64_t mn, int nz, double* __restrict__ C) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= mn) return;
  double s = 0.0;
  for (int z = 0; z < nz; ++z) s += P[(int64_t)z * mn + e];
  C[e] -= s;
}

// rows padded with zeros to a pitch that is a multiple of 16 doubles: the NT GEMM then takes its interior fast path
// (16-byte aligned rows, K a multiple of the k-tile) for any D
__global__ void __launch_bounds__(256) pad_rows_kernel(const double* __restrict__ X, int64_t rows, int D, int Dp,
                                                       double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * Dp) return;
  const int64_t r = e / Dp;
  const int k = (int)(e - r * Dp);
  out[e] = k < D ? X[r * D + k] : 0.0;
}

// nX[r] = |X_r|^2, cX[r] = X_r . JA_r  (one wavefront per table row)
__global__ void __launch_bounds__(256) wide_row_stats_kernel(const double* __restrict__ xp, const double* __restrict__ jap,
                                                             int64_t MP, int D, double* __restrict__ nX,
                                                             double* __restrict__ cX) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= MP) return;
  double s = 0.0, c = 0.0;
  for (int k = lane; k < D; k += 64) {
    const double x = xp[r * D + k];
    s += x * x;
    c += x * jap[r * D + k];
  }
  s = wave_sum(s);
  c = wave_sum(c);
  if (lane == 0) {
    nX[r] = s;
    cX[r] = c;
  }
}

// |x_q|^2 per query
__global__ void __launch_bounds__(256) query_norm_kernel(const double* __restrict__ xq, int64_t B, int D,
                                                         double* __restrict__ nx) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= B) return;
  double s = 0.0;
  for (int k = lane; k < D; k += 64) {
    const double v = xq[q * D + k];
    s += v * v;
  }
  s = wave_sum(s);
  if (lane == 0) nx[q] = s;
}

// S, T hold -X_r.x_q and -JA_r.x_q (gemm_nt_sub subtracts); in place: S <- w1, T <- b2.  One thread per query
// column, a strip of `rows_per` table rows per workgroup row; partial column sums per strip (deterministic).
__global__ void __launch_bounds__(256) matern_pairs_kernel(double* __restrict__ S, double* __restrict__ T, int64_t ld,
                                                           int64_t MP, int64_t Bc, const double* __restrict__ nx,
                                                           const double* __restrict__ nX, const double* __restrict__ cX,
                                                           const double* __restrict__ aE, double sig, int rows_per,
                                                           double* __restrict__ part_w, double* __restrict__ part_e) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per;
  const int64_t r1 = (r0 + rows_per < MP) ? r0 + rows_per : MP;
  if (q >= Bc) return;
  const double inv_sig = 1.0 / sig, sqrt5 = 2.23606797749978969641;
  const double fact = 5.0 / (3.0 * sig * sig * sig), dscale = 5.0 / sig, inv_3sig = 1.0 / (3.0 * sig);
  const double nq = nx[q];
  double sw = 0.0, se = 0.0;
  for (int64_t r = r0; r < r1; ++r) {
    const double p1 = -S[r * ld + q], p2 = -T[r * ld + q];
    double s2 = nq + nX[r] - 2.0 * p1;
    s2 = s2 > 0.0 ? s2 : 0.0;
    const double sa = p2 - cX[r];
    const double nrm = sqrt5 * sqrt(s2);
    const double ex = exp(-nrm * inv_sig);
    const double b = fact * ex;
    const double b2 = b * (nrm + sig);
    double w1 = dscale * sa * b;
    double e = sa * b2;
    if (aE) {
      const double ae = aE[r];
      w1 += ae * b2;
      e += ae * (1.0 + (nrm * inv_sig) * (1.0 + nrm * inv_3sig)) * ex;
    }
    S[r * ld + q] = w1;
    T[r * ld + q] = b2;
    sw += w1;
    se += e;
  }
  part_w[(int64_t)blockIdx.y * Bc + q] = sw;
  part_e[(int64_t)blockIdx.y * Bc + q] = se;
}

// F_x[q][k] = x_q[k] * sum_r w1[r][q];  E[q] = sum of the strip partials
__global__ void __launch_bounds__(256) fx_init_kernel(const double* __restrict__ xq, int64_t Bc, int D,
                                                      const double* __restrict__ part_w,
                                                      const double* __restrict__ part_e, int nparts,
                                                      double* __restrict__ Fx, double* __restrict__ E) {
  __shared__ double cs;
  const int64_t q = blockIdx.x;
  if (threadIdx.x < 64) {
    double s = 0.0, e = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 64) {
      s += part_w[(int64_t)p * Bc + q];
      e += part_e[(int64_t)p * Bc + q];
    }
    s = wave_sum(s);
    e = wave_sum(e);
    if (threadIdx.x == 0) {
      cs = s;
      E[q] = e;
    }
  }
  __syncthreads();
  const double c = cs;
  for (int k = threadIdx.x; k < D; k += 256) Fx[q * D + k] = xq[q * D + k] * c;
}

}  // namespace

// F_x (B x D) and the unscaled energies (B) of B queries against the resident model, D > 256.
// part_F, part_E: the (JS = 1) buffers predict_epilogue_kernel reads.
int predict_wide_device(gdml_ctx* ctx, const double* d_xq, int64_t B, double* part_F, double* part_E) {
  Model& md = ctx->model;
  const int D = md.D;
  const int64_t MP = md.M * md.P;
  hipStream_t st = ctx->stream;
  // query chunk: two MP x Bc pair-scalar matrices below ~2 GB
  // Round 6 (option predict.wide_pad): table rows padded to a multiple of 64 (the stack [X; JA] is then whole 128-row tiles) and
  // query chunks to a multiple of 128 with zero rows -- no edge tiles in the two contractions (their guarded path is 2-3 x
  // slower per tile and used to be the launch's tail: configs[4] 6000 x 3000 had 70 of 1128)
  const bool pad = ctx_opt_i(ctx, "predict.wide_pad", 1) != 0;
  const int64_t MPp = pad ? (MP + 63) / 64 * 64 : MP;
  int64_t Bc = (int64_t)(1.0e9 / (8.0 * (double)MPp));
  Bc = Bc / 128 * 128;
  if (Bc < 256) Bc = 256;
  if (Bc > B) Bc = pad ? (B + 127) / 128 * 128 : (B + 1) / 2 * 2;
  const int rows_per = 512;
  const int nparts = (int)((MP + rows_per - 1) / rows_per);
  const int Dp = (D + 15) / 16 * 16;
  double* w;
  GDML_TRY(ctx_slot(ctx, 7, (2 * MPp * Bc + 2 * MP + Bc + 2 * (int64_t)nparts * Bc + (2 * MPp + Bc) * (int64_t)Dp) * 8, &w));
  double* S = w;
  double* T = S + MPp * Bc;
  double* nX = T + MPp * Bc;
  double* cX = nX + MP;
  double* nx = cX + MP;
  double* pw = nx + Bc;
  double* pe = pw + (int64_t)nparts * Bc;
  double* Xpad = pe + (int64_t)nparts * Bc;
  double* Jpad = Xpad + MPp * Dp;
  double* Qpad = Jpad + MPp * Dp;
  hipLaunchKernelGGL(wide_row_stats_kernel, dim3(ceil_div(MP, 4)), dim3(256), 0, st, md.xp, md.jap, MP, D, nX, cX);
  hipLaunchKernelGGL(pad_rows_kernel, dim3(ceil_div(MP * Dp, 256)), dim3(256), 0, st, md.xp, MP, D, Dp, Xpad);
  hipLaunchKernelGGL(pad_rows_kernel, dim3(ceil_div(MP * Dp, 256)), dim3(256), 0, st, md.jap, MP, D, Dp, Jpad);
  if (MPp > MP) {
    HIP_CHECK(ctx, hipMemsetAsync(Xpad + MP * Dp, 0, (size_t)((MPp - MP) * Dp * 8), st));
    HIP_CHECK(ctx, hipMemsetAsync(Jpad + MP * Dp, 0, (size_t)((MPp - MP) * Dp * 8), st));
  }
  for (int64_t q0 = 0; q0 < B; q0 += Bc) {
    const int64_t bc = (B - q0 < Bc) ? B - q0 : Bc;
    const double* xq = d_xq + q0 * D;
    const int64_t bcp = pad ? (bc + 127) / 128 * 128 : bc;  // <= Bc
    // (padded: the first contraction OVERWRITES its whole 2 MPp x bcp block -- no clearing, and the product does not read S back;
    //  configs[3] cleared and re-read 1.8 GB per mat-vec)
    if (!pad) HIP_CHECK(ctx, hipMemsetAsync(S, 0, 2 * MPp * Bc * 8, st));
    hipLaunchKernelGGL(query_norm_kernel, dim3(ceil_div(bc, 4)), dim3(256), 0, st, xq, bc, D, nx);
    hipLaunchKernelGGL(pad_rows_kernel, dim3(ceil_div(bc * Dp, 256)), dim3(256), 0, st, xq, bc, D, Dp, Qpad);
    if (bcp > bc) HIP_CHECK(ctx, hipMemsetAsync(Qpad + bc * Dp, 0, (size_t)((bcp - bc) * Dp * 8), st));
    // S = -X_p X_q^T, T = -JA_p X_q^T  (zero padding contributes nothing)
    // one launch for both: [Xpad; Jpad] and [S; T] are contiguous stacks of 2 MP rows
    if (pad) GDML_TRY(launch_gemm_nt_neg(ctx, st, Xpad, Dp, Qpad, Dp, S, Bc, 2 * MPp, bcp, Dp));
    else GDML_TRY(launch_gemm_nt_sub_fill(ctx, st, Xpad, Dp, Qpad, Dp, S, Bc, 2 * MPp, bcp, Dp));
    hipLaunchKernelGGL(matern_pairs_kernel, dim3(ceil_div(bc, 256), nparts), dim3(256), 0, st, S, T, Bc, MP, bc, nx, nX, cX,
                       md.has_aE ? md.aE : nullptr, md.sig, rows_per, pw, pe);
    double* Fx = part_F + q0 * D;
    hipLaunchKernelGGL(fx_init_kernel, dim3((unsigned)bc), dim3(256), 0, st, xq, bc, D, pw, pe, nparts, Fx, part_E + q0);
    // F_x -= W1^T X_p + B2^T JA_p as ONE contraction over the 2 MP stacked rows ([S; T] and [Xpad; Jpad] are contiguous),
    // split over the table rows so that the launch fills the chip
    {
      const int64_t tiles = (int64_t)ceil_div(D, WT) * ceil_div(bc, WT), nt = (2 * MPp + WBK - 1) / WBK;
      int nz = (int)((3 * 512 + tiles - 1) / tiles);
      if (nz > 32) nz = 32;
      if ((int64_t)nz * 64 > nt) nz = (int)(nt / 64);  // at least 64 k-tiles per split
      if (nz < 1) nz = 1;
      if (ctx_opt_i(ctx, "predict.tn_fill", 1) != 0) {
        // (round 6, late) among the admissible split counts from there up, the one whose units fill whole rounds of the chip's
        // 512 slots best: configs[3] has 112 tiles -- 14 splits are 1568 units = 3.06 rounds, i.e. four rounds for three
        // rounds' worth of work (0.77); 32 splits are exactly seven.  A split more costs bc x D doubles of partial sums.
        int best = nz;
        double best_score = -1.0;
        for (int z = nz; z <= 32 && (int64_t)z * 64 <= nt && (int64_t)z * bc * D * 8 <= ((int64_t)1 << 30); ++z) {
          const int64_t units = tiles * z, rounds = (units + 511) / 512;
          const double score = (double)units / (double)(rounds * 512) - 0.002 * (z - nz);
          if (score > best_score) { best_score = score; best = z; }
        }
        nz = best;
      }
      double* Pp;
      GDML_TRY(ctx_slot(ctx, 9, (int64_t)nz * bc * D * 8, &Pp));
      dim3 grid((unsigned)ceil_div(D, WT), (unsigned)ceil_div(bc, WT), (unsigned)nz);
      // unchecked loads wherever memory exists: S / T have Bc columns (a multiple of 128 when padded: zeros past bc), and a
      // table row's columns past Dp are the next row's (the last row's: the query block behind the tables) -- they only feed
      // output columns >= D, which are not stored.  Before: 14 % (configs[3]) of the tiles took the checked path
      const int64_t a_cols = pad ? bcp : bc, b_cols = pad ? ((int64_t)D + WT - 1) / WT * WT : (int64_t)Dp;
      hipLaunchKernelGGL(gemm_tn_split_kernel, grid, dim3(256), 0, st, S, Bc, a_cols, Xpad, (int64_t)Dp, b_cols, Pp, bc, (int64_t)D, 2 * MPp);
      hipLaunchKernelGGL(reduce_tn_kernel, dim3(ceil_div(bc * D, 256)), dim3(256), 0, st, Pp, bc * D, nz, Fx);
    }
    ctx->launch_counter += 7;
    HIP_CHECK(ctx, hipGetLastError());
  }
  return GDML_OK;
}
