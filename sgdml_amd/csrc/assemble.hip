// Kernel-matrix assembly on gfx950.
//
// Replaces GDMLTrain._assemble_kernel_mat / _assemble_kernel_mat_wkr (sgdml/train.py:97-302,
// :1260-1535) and GDMLTorchAssemble (sgdml/torchtools.py:110-392).
//
// Math, un-negated, per block (i,j), d_p = x_i - P_p x_j, n_p = sqrt5 |d_p|,
// b_p = 5 exp(-n_p/sig)/(3 sig^4), c_p = (sig^2 + sig n_p) b_p:
//     K_ij = sum_p [ 5 b_p (J_i^T d_p)(J_j^pT d_p)^T  -  c_p J_i^T J_j^p ]
// The reference forms the dense D x 3N matrices and multiplies (2 D (3N)^2 flops per block).
// Here the 6-nonzeros-per-row structure of the Jacobians is used instead:
//   v_p = J_i^T d_p, u_p = J_j^pT d_p                  (3N-vectors, N-1 terms per entry)
//   (J_i^T J_j^p)[(a,.),(b,.)] = G_i(a,a') (x) G_j(b,pi_p a)      if a' = pi_p^-1(b) != a
//                              = sum_m G_i(a,m) (x) G_j(b,pi_p m) if pi_p(a) = b  ("diagonal")
// with G_x(a,m) = d(desc{a,m})/d r_a = (r_m - r_a)/d^3 = +-g_x[pair(a,m)], so every output element
// costs 2 FMAs per permutation and the kernel is bound by the HBM write of K (8 (3N)^2 bytes per
// block).
//
// The kernels live in their own files: assemble_strip.hip (P = 1, 11 <= N <= 21, all columns: the benchmark path),
// assemble_wave.hip (P = 1, N <= 21: column subsets, row-cyclic shares), assemble_pts.hip (8 <= N <= 24 with a permutation
// group, dense columns), assemble_perm.hip (everything else: any group, any N).  This file resolves the
// column selection (train.py:1335-1407), owns the matrix buffer, dispatches, and carries the energy-constraint columns.
#include "common.h"

__device__ __forceinline__ double block_sum(double v, double* red, int tid, int nwaves) {
  v = wave_sum(v);
  __syncthreads();  // protects red[] reuse
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < nwaves; ++w) s += red[w];
  return s;
}

// Energy-constraint columns (train.py:250-300): for E column of point jj and every i,
//   K[3N i + c, col] = -sum_p w_p (d_p^T J_i^p)[c],  d_p = x_jj - P_p x_i,
//   K[3N M + i, col] = -sum_p (1 + n/sig (1 + n/(3 sig))) exp(-n/sig)
struct EColArgs {
  const double* x;
  const double* g;
  const int32_t* tp;
  const int32_t* perm;
  const int32_t* pinv;
  int64_t M;
  int N, D, P;
  double sig;
  const int32_t* jj_list;   // n_e points whose E column is requested
  const int32_t* out_cols;  // output column for each
  double* K;
  int64_t ld;
  int64_t i0;               // first row point of this launch
  // row mode (distributed Cholesky with energy constraints, assemble_erows_cyclic_launch): the same values written as ROW
  // out_cols[.] of A = -K + lam I -- K is symmetric (train.py:235-248 and :281-296 are one formula with the roles of the two
  // points exchanged), the factorisation reads rows, and a rank of the block-row-cyclic layout stores only its own
  int row_mode;
  double lam;
  // column mode with row-sharded storage (sharded Nystroem matrix): force rows of point i at 3N (i - i_beg), its energy row
  // at e_row0 + i (unsharded: i_beg = 0, e_row0 = 3N M)
  int64_t i_beg, e_row0;
};

// where the value of (row point i, component t) / the energy-energy value of row point i goes
__device__ __forceinline__ void ecol_store(const EColArgs& A, int64_t i, int t, int64_t col, double v) {
  if (A.row_mode) A.K[col * A.ld + i * (3 * A.N) + t] = -v;
  else A.K[((i - A.i_beg) * (3 * A.N) + t) * A.ld + col] = v;
}
__device__ __forceinline__ void ecol_store_ee(const EColArgs& A, int64_t i, int64_t jj, int64_t col, double v) {
  if (A.row_mode) A.K[col * A.ld + A.M * (3 * A.N) + i] = (i == jj) ? A.lam - v : -v;
  else A.K[(A.e_row0 + i) * A.ld + col] = v;
}

__global__ void __launch_bounds__(256) ecol_kernel(EColArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int N = A.N, D = A.D, N3 = 3 * N;
  double* xq = smem;        // D  (x_jj)
  double* dv = xq + D;      // D
  double* red = dv + D;     // 32
  const int tid = threadIdx.x, T = blockDim.x, nwaves = T >> 6;
  const int64_t jj = A.jj_list[blockIdx.x];
  const int64_t col = A.out_cols[blockIdx.x];
  const int64_t i = (int64_t)blockIdx.y + A.i0;  // grid.y is tiled by the host (65535 limit)
  for (int k = tid; k < D; k += T) xq[k] = A.x[jj * D + k];
  const double* xi = A.x + i * D;        // row point: descriptor and compressed Jacobian through L1/L2 (any N)
  const double* gi = A.g + i * 3 * (int64_t)D;
  const double sig = A.sig, inv_sig = 1.0 / sig;
  const double sqrt5 = 2.23606797749978969641;
  const double e_fact = 5.0 / (3.0 * sig * sig * sig);
  double out0 = 0.0, out1 = 0.0;  // up to 2 outputs per thread (3N <= 512)
  double ee = 0.0;
  for (int p = 0; p < A.P; ++p) {
    const int32_t* tp = A.tp + (size_t)p * D;
    const int32_t* perm = A.perm + (size_t)p * N;
    const int32_t* pinv = A.pinv + (size_t)p * N;
    __syncthreads();
    double part = 0.0;
    for (int k = tid; k < D; k += T) {
      double dk = xq[k] - xi[tp[k]];
      dv[k] = dk;
      part += dk * dk;
    }
    const double nrm2 = block_sum(part, red, tid, nwaves);
    const double nrm = sqrt5 * sqrt(nrm2);
    const double ex = exp(-nrm * inv_sig);
    const double w = e_fact * (nrm + sig) * ex;
    for (int r = 0; r < 2; ++r) {
      int t = tid + r * T;
      if (t < N3) {
        int bb = t / 3, be = t - 3 * bb;
        int ap = pinv[bb];
        double s = 0.0;
        for (int m = 0; m < N; ++m) {
          if (m == ap) continue;
          int q = perm[m];
          double gv = gi[pair_idx(bb, q) * 3 + be];
          s += dv[pair_idx(ap, m)] * (bb < q ? gv : -gv);
        }
        if (r == 0)
          out0 -= w * s;
        else
          out1 -= w * s;
      }
    }
    ee -= (1.0 + (nrm * inv_sig) * (1.0 + nrm / (3.0 * sig))) * ex;
  }
  if (tid < N3) ecol_store(A, i, tid, col, out0);
  if (tid + T < N3) ecol_store(A, i, tid + T, col, out1);
  if (tid == 0) ecol_store_ee(A, i, jj, col, ee);
}

// The same columns for molecules beyond the kernel above (3N > 512, or descriptors that do not fit its LDS tables): nothing
// staged but the P weights; every output element is summed by one thread, outermost loop over the outputs.
__global__ void __launch_bounds__(256) ecol_big_kernel(EColArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int N = A.N, D = A.D, N3 = 3 * N;
  double* wS = smem;       // P
  double* red = wS + A.P;  // 32
  const int tid = threadIdx.x, T = blockDim.x, nwaves = T >> 6;
  const int64_t jj = A.jj_list[blockIdx.x];
  const int64_t col = A.out_cols[blockIdx.x];
  const int64_t i = (int64_t)blockIdx.y + A.i0;
  const double* xq = A.x + jj * D;
  const double* xi = A.x + i * D;
  const double* gi = A.g + i * 3 * (int64_t)D;
  const double sig = A.sig, inv_sig = 1.0 / sig;
  const double sqrt5 = 2.23606797749978969641;
  const double e_fact = 5.0 / (3.0 * sig * sig * sig);
  double ee = 0.0;
  for (int p = 0; p < A.P; ++p) {
    const int32_t* tp = A.tp + (size_t)p * D;
    double part = 0.0;
    for (int k = tid; k < D; k += T) {
      const double dk = xq[k] - xi[tp[k]];
      part += dk * dk;
    }
    const double nrm2 = block_sum(part, red, tid, nwaves);
    const double nrm = sqrt5 * sqrt(nrm2);
    const double ex = exp(-nrm * inv_sig);
    if (tid == 0) wS[p] = e_fact * (nrm + sig) * ex;
    ee -= (1.0 + (nrm * inv_sig) * (1.0 + nrm / (3.0 * sig))) * ex;
  }
  __syncthreads();
  for (int t = tid; t < N3; t += T) {
    const int bb = t / 3, be = t - 3 * bb;
    double out = 0.0;
    for (int p = 0; p < A.P; ++p) {
      const int32_t* tp = A.tp + (size_t)p * D;
      const int32_t* perm = A.perm + (size_t)p * N;
      const int ap = A.pinv[(size_t)p * N + bb];
      double s = 0.0;
      for (int m = 0; m < N; ++m) {
        if (m == ap) continue;
        const int q = perm[m];
        const double gv = gi[pair_idx(bb, q) * 3 + be];
        const int kk = pair_idx(ap, m);
        const double dk = xq[kk] - xi[tp[kk]];
        s += dk * (bb < q ? gv : -gv);
      }
      out -= wS[p] * s;
    }
    ecol_store(A, i, t, col, out);
  }
  if (tid == 0) ecol_store_ee(A, i, jj, col, ee);
}

// Launch of the energy-constraint kernels: n_e requested points (d_ep) with their output columns -- or, in row mode, output
// rows -- (d_ec), all M partner points.
static int ecol_launch(gdml_ctx* ctx, double sig, const int32_t* d_ep, const int32_t* d_ec, int64_t n_e, double* K, int64_t ld,
                       int row_mode, double lam, int64_t i_beg = 0, int64_t i_end = -1) {
  TrainSet& ts = ctx->ts;
  const int64_t M = ts.M;
  const int N = ts.N, N3 = 3 * N;
  EColArgs E;
  E.x = ts.x; E.g = ts.g; E.tp = ts.tp; E.perm = ts.perm; E.pinv = ts.pinv;
  E.M = M; E.N = N; E.D = ts.D; E.P = ts.P; E.sig = sig;
  E.jj_list = d_ep; E.out_cols = d_ec; E.K = K; E.ld = ld;
  E.row_mode = row_mode; E.lam = lam;
  if (i_end < 0) i_end = M;
  E.i_beg = row_mode ? 0 : i_beg;
  E.e_row0 = (i_beg == 0 && i_end == M) ? M * N3 : (i_end - i_beg) * N3 - i_beg;
  // two outputs per thread and the descriptor tables in LDS, or (large molecules) the table-free kernel
  const bool small = N3 <= 512 && (size_t)(2 * ts.D + 32) * 8 <= (size_t)160 * 1024;
  const size_t lds = small ? (size_t)(2 * ts.D + 32) * 8 : (size_t)(ts.P + 32) * 8;
  if (small) hipFuncSetAttribute((const void*)ecol_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  else hipFuncSetAttribute((const void*)ecol_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int64_t i0 = i_beg; i0 < i_end; i0 += 65535) {  // grid.y limit
    E.i0 = i0;
    const int64_t ny = (i_end - i0 < 65535) ? i_end - i0 : 65535;
    const dim3 grid((unsigned)n_e, (unsigned)ny);
    if (small) hipLaunchKernelGGL(ecol_kernel, grid, dim3(256), lds, ctx->stream, E);
    else hipLaunchKernelGGL(ecol_big_kernel, grid, dim3(256), lds, ctx->stream, E);
    ctx->launch_counter++;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return gdml_fail(ctx, GDML_ERR_HIP, "ecol launch: %s", hipGetErrorString(e));
  return GDML_OK;
}

// Energy-constraint ROWS of A = -K + lam I owned by this rank in the block-row-cyclic layout (global rows 3N M + e, e < M):
// force columns in full, energy columns with the regularised diagonal.  The reference assembles them as rows AND columns
// (train.py:235-248, :250-300); the distributed Cholesky reads the lower triangle, so the rows are all it needs.
int assemble_erows_cyclic_launch(gdml_ctx* ctx, double sig, double lam, double* K, int64_t ld, int cyc_W, int cyc_rank,
                                 int cyc_nb) {
  TrainSet& ts = ctx->ts;
  const int64_t n_ff = ts.M * 3 * (int64_t)ts.N;
  const int W = cyc_W > 0 ? cyc_W : 1;
  std::vector<int32_t> e_pts, e_rows;
  for (int64_t e = 0; e < ts.M; ++e) {
    const int64_t g = n_ff + e, b = g / cyc_nb;
    if ((int)(b % W) != cyc_rank) continue;
    e_pts.push_back((int32_t)e);
    e_rows.push_back((int32_t)((b / W) * cyc_nb + g % cyc_nb));
  }
  if (e_pts.empty()) return GDML_OK;
  int32_t *d_ep = nullptr, *d_er = nullptr;
  GDML_TRY(ctx_alloc(ctx, (void**)&d_ep, e_pts.size() * 4));
  int rc = ctx_alloc(ctx, (void**)&d_er, e_rows.size() * 4);
  if (rc == GDML_OK) {
    hipError_t h = hipMemcpyAsync(d_ep, e_pts.data(), e_pts.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    if (h == hipSuccess) h = hipMemcpyAsync(d_er, e_rows.data(), e_rows.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    if (h == hipSuccess) h = hipStreamSynchronize(ctx->stream);  // the host vectors leave scope
    if (h != hipSuccess) rc = gdml_fail(ctx, GDML_ERR_HIP, "assemble_erows_cyclic: %s", hipGetErrorString(h));
  }
  if (rc == GDML_OK) rc = ecol_launch(ctx, sig, d_ep, d_er, (int64_t)e_pts.size(), K, ld, 1, lam);
  if (rc == GDML_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = gdml_fail(ctx, GDML_ERR_HIP, "assemble_erows_cyclic: sync");
  if (d_ep) ctx_free(ctx, d_ep);
  if (d_er) ctx_free(ctx, d_er);
  return rc;
}

// Rows of A = -K + lam I owned by this rank in the block-row-cyclic layout of the distributed Cholesky (any P,
// any N): full column range, lower blocks.  K: the rank's local matrix.
int assemble_cyclic_launch(gdml_ctx* ctx, double sig, double lam, double* K, int64_t ld, int cyc_W, int cyc_rank,
                           int cyc_nb) {
  TrainSet& ts = ctx->ts;
  if (assemble_wave_applicable(ctx))
    return assemble_wave_launch(ctx, sig, 0, nullptr, nullptr, 0, ts.M, K, ld, 0, ts.M, 1, lam, cyc_W, cyc_rank, cyc_nb);
  return assemble_perm_launch(ctx, sig, 0, nullptr, nullptr, 0, ts.M, 0, K, ld, 0, ts.M, 1, lam, cyc_W > 1 ? cyc_W : 0,
                              cyc_rank, cyc_nb);
}

// as_A: assemble for the analytic solve (gdml_assemble_A): where the register-resident kernel applies the
// matrix is produced directly as A = -K + lam I, blocks on/below the block diagonal only.
static int assemble_impl(gdml_ctx* ctx, double sig, int use_E_cstr, int col_kind, int64_t col_a, int64_t col_b,
                         const int64_t* idx, int64_t n_idx, int64_t alloc_extra_rows, double* K_host_out,
                         int64_t ldk, int as_A, double lam) {
  if (!ctx) return GDML_ERR_INVALID;
  TrainSet& ts = ctx->ts;
  if (!ts.x) return gdml_fail(ctx, GDML_ERR_STATE, "gdml_assemble_K: call gdml_train_upload first");
  if (!(sig > 0) || alloc_extra_rows < 0)
    return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_assemble_K: bad sig / alloc_extra_rows");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int64_t M = ts.M;
  const int N = ts.N, N3 = 3 * N;
  const int64_t n_ff = M * N3;
  const int64_t n_rows = n_ff + (use_E_cstr ? M : 0);

  // ---- resolve the column selection into: dense point range, or (jlist, colmap), + E columns
  std::vector<int32_t> jlist, colmap, e_pts, e_cols;
  int64_t n_cols = 0, j0 = 0, n_j = 0;
  bool dense = true;
  if (col_kind == GDML_COLS_ALL) {
    j0 = 0;
    n_j = M;
    n_cols = n_rows;
    if (use_E_cstr)
      for (int64_t e = 0; e < M; ++e) {
        e_pts.push_back((int32_t)e);
        e_cols.push_back((int32_t)(n_ff + e));
      }
  } else if (col_kind == GDML_COLS_POINTS) {
    // reference slice path (train.py:1357-1374): points [col_a, col_b) of the 2M-long list
    int64_t lim = M + (use_E_cstr ? M : 0);
    if (col_a < 0 || col_b < col_a || col_b > lim)
      return gdml_fail(ctx, GDML_ERR_INVALID, "point range [%lld,%lld) out of [0,%lld]",
                       (long long)col_a, (long long)col_b, (long long)lim);
    j0 = col_a;
    int64_t jf_end = col_b < M ? col_b : M;
    n_j = jf_end > col_a ? jf_end - col_a : 0;
    n_cols = n_j * N3;
    for (int64_t e = (col_a > M ? col_a : M); e < col_b; ++e) {
      e_pts.push_back((int32_t)(e - M));
      e_cols.push_back((int32_t)n_cols++);
    }
  } else if (col_kind == GDML_COLS_INDEX) {
    if (!idx || n_idx < 1) return gdml_fail(ctx, GDML_ERR_INVALID, "empty column index list");
    if (n_idx > n_rows)
      return gdml_fail(ctx, GDML_ERR_INVALID, "Columns indexed beyond range.");  // train.py:1351
    dense = false;
    int64_t prev = -1;
    for (int64_t q = 0; q < n_idx; ++q) {
      int64_t cidx = idx[q];
      if (cidx <= prev || cidx >= n_rows)
        return gdml_fail(ctx, GDML_ERR_INVALID,
                         "column indices must be sorted, unique and < %lld (train.py:1341-1345)",
                         (long long)n_rows);
      prev = cidx;
      if (cidx < n_ff) {
        int32_t pt = (int32_t)(cidx / N3);
        if (jlist.empty() || jlist.back() != pt) {
          jlist.push_back(pt);
          colmap.insert(colmap.end(), N3, -1);
        }
        colmap[(jlist.size() - 1) * N3 + (cidx % N3)] = (int32_t)q;
      } else {
        e_pts.push_back((int32_t)(cidx - n_ff));
        e_cols.push_back((int32_t)q);
      }
    }
    n_j = (int64_t)jlist.size();
    n_cols = n_idx;
  } else {
    return gdml_fail(ctx, GDML_ERR_INVALID, "unknown col_kind %d", col_kind);
  }
  if (!e_pts.empty() && !use_E_cstr)
    return gdml_fail(ctx, GDML_ERR_INVALID, "energy columns requested without use_E_cstr");
  if (K_host_out && ldk < n_cols)
    return gdml_fail(ctx, GDML_ERR_INVALID, "ldk (%lld) < n_cols (%lld)", (long long)ldk,
                     (long long)n_cols);

  // ---- row sharding: with a communicator (gdml_comm_init, world > 1) the Nystroem matrix
  // (index-list columns + extra rows) holds only this rank's training points
  int64_t i_beg = 0, i_end = M;
  const bool sharded = ctx->world > 1 && col_kind == GDML_COLS_INDEX && alloc_extra_rows > 0;
  if (sharded) {
    // energy constraints (round 6): a rank holds the force rows of its points followed by THEIR energy rows
    shard_points(ctx, M, &i_beg, &i_end, nullptr);
  }
  const int64_t n_rows_store = sharded ? (i_end - i_beg) * (N3 + (use_E_cstr ? 1 : 0)) : n_rows;

  // ---- (re)allocate the device matrix
  const int64_t tot_rows = n_rows_store + alloc_extra_rows;
  const int64_t ld = (n_cols + 15) / 16 * 16;  // rows start on 128-byte boundaries
  // the buffer is reused when the size is unchanged (hyper-parameter sweeps, benchmark loops)
  if (ctx->K && ctx->K_bytes != tot_rows * ld * 8) {
    GDML_TRY(ctx_free(ctx, ctx->K));
    ctx->K = nullptr;
    ctx->precon = nullptr;
    precon_release_aux(ctx);  // the auxiliary buffers of a preconditioner that lived in the old matrix go with it
  }
  if (!ctx->K) {
    GDML_TRY(ctx_alloc(ctx, (void**)&ctx->K, tot_rows * ld * 8));
    ctx->K_bytes = tot_rows * ld * 8;
  }
  ctx->precon = nullptr;  // a resident preconditioner lives in this buffer and is now overwritten
  ctx->K_rows = n_rows_store;
  ctx->K_rows_global = n_rows;
  ctx->K_sharded = sharded;
  ctx->K_cols = n_cols;
  ctx->K_extra = alloc_extra_rows;
  ctx->K_ld = ld;
  ctx->K_factored = false;
  ctx->K_destroyed = false;
  ctx->K_rhs_row = false;
  ctx->K_sig = sig;
  ctx->K_use_E = use_E_cstr;
  // fused form for the analytic solve: the strip / register-resident kernels (P = 1, N <= 21) or, for permutation groups and
  // larger molecules, the general kernels in their lower / negated mode (= the row-cyclic mode with one rank)
  const bool lower_A = as_A && col_kind == GDML_COLS_ALL && !use_E_cstr && !sharded && ctx_opt_i(ctx, "asm.lower", 1) != 0;
  ctx->K_is_A = lower_A;
  if (lower_A) ctx->K_lam = lam;

  int32_t *d_jlist = nullptr, *d_colmap = nullptr, *d_ep = nullptr, *d_ec = nullptr;
  if (!dense && n_j > 0) {
    GDML_TRY(ctx_alloc(ctx, (void**)&d_jlist, n_j * 4));
    GDML_TRY(ctx_alloc(ctx, (void**)&d_colmap, n_j * N3 * 4));
    HIP_CHECK(ctx, hipMemcpyAsync(d_jlist, jlist.data(), n_j * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(ctx, hipMemcpyAsync(d_colmap, colmap.data(), n_j * N3 * 4, hipMemcpyHostToDevice,
                                  ctx->stream));
  }
  if (!e_pts.empty()) {
    GDML_TRY(ctx_alloc(ctx, (void**)&d_ep, e_pts.size() * 4));
    GDML_TRY(ctx_alloc(ctx, (void**)&d_ec, e_cols.size() * 4));
    HIP_CHECK(ctx, hipMemcpyAsync(d_ep, e_pts.data(), e_pts.size() * 4, hipMemcpyHostToDevice,
                                  ctx->stream));
    HIP_CHECK(ctx, hipMemcpyAsync(d_ec, e_cols.data(), e_cols.size() * 4, hipMemcpyHostToDevice,
                                  ctx->stream));
  }

  phase_begin(ctx);
  int rc = GDML_OK;
  if (n_j > 0) {
    if (i_end <= i_beg)
      rc = GDML_OK;
    else if (lower_A && !assemble_wave_applicable(ctx))
      rc = assemble_cyclic_launch(ctx, sig, lam, ctx->K, ld, 1, 0, 512);
    else if (dense && !use_E_cstr && i_beg == 0 && i_end == M && assemble_strip_applicable(ctx))
      rc = assemble_strip_launch(ctx, sig, ctx->K, ld, lower_A ? 1 : 0, lam);
    else if (assemble_wave_applicable(ctx))
      rc = assemble_wave_launch(ctx, sig, use_E_cstr, d_jlist, d_colmap, j0, n_j, ctx->K, ld, i_beg, i_end,
                                lower_A ? 1 : 0, lam);
    else
      rc = assemble_perm_launch(ctx, sig, use_E_cstr, d_jlist, d_colmap, j0, n_j, 0, ctx->K, ld, i_beg, i_end, 0, 0.0, 0, 0, 0,
                                dense ? nullptr : colmap.data());
  }
  if (rc == GDML_OK && !e_pts.empty()) {
    if (i_end > i_beg) rc = ecol_launch(ctx, sig, d_ep, d_ec, (int64_t)e_pts.size(), ctx->K, ld, 0, 0.0, i_beg, i_end);
  }
  if (rc == GDML_OK) rc = phase_end(ctx, "assemble");
  if (d_jlist) ctx_free(ctx, d_jlist);
  if (d_colmap) ctx_free(ctx, d_colmap);
  if (d_ep) ctx_free(ctx, d_ep);
  if (d_ec) ctx_free(ctx, d_ec);
  GDML_TRY(rc);

  if (K_host_out) {
    HIP_CHECK(ctx, hipMemcpy2DAsync(K_host_out, ldk * 8, ctx->K, ld * 8, n_cols * 8, n_rows_store,
                                    hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return GDML_OK;
}

extern "C" int gdml_assemble_K(gdml_ctx* ctx, double sig, int use_E_cstr, int col_kind,
                               int64_t col_a, int64_t col_b, const int64_t* idx, int64_t n_idx,
                               int64_t alloc_extra_rows, double* K_host_out, int64_t ldk) {
  return assemble_impl(ctx, sig, use_E_cstr, col_kind, col_a, col_b, idx, n_idx, alloc_extra_rows, K_host_out, ldk,
                       0, 0.0);
}

extern "C" int gdml_assemble_A(gdml_ctx* ctx, double sig, double lam, int use_E_cstr, int64_t alloc_extra_rows) {
  return assemble_impl(ctx, sig, use_E_cstr, GDML_COLS_ALL, 0, 0, nullptr, 0, alloc_extra_rows, nullptr, 0, 1, lam);
}
