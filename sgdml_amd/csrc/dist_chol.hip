// Distributed analytic solve: the kernel matrix block-partitioned over the GPUs of a node (one process per GPU).
//
// The reference has no distributed code (its only multi-GPU path is nn.DataParallel over query batches,
// sgdml/train.py:1463-1469); this is the analytic branch of Analytic.solve (sgdml/solvers/analytic.py:65-99) for
// systems that do not fit -- or should not wait for -- one GPU: BASELINE.json configs[3] (n = 252 000: 508 GB) and
// configs[4].
//
// Layout: block-ROW cyclic.  The matrix is row-major and the factorisation works on rows (the panel solve is row
// local, the trailing update of a row needs the panel rows of the columns left of its diagonal), so global row
// block b (NB = 512 rows, full width) lives on rank b % W as local row block b / W.  Every rank assembles only its
// own rows (A = -K + lam I, lower blocks: assemble_wave_kernel<N, CYC> for P = 1, N <= 21, the LDS kernel in its
// row-cyclic mode for everything else), so the matrix never exists in one place:
// n^2 * 8 / W bytes per GPU.  Cyclic ownership balances the lower-triangular work (row b has b blocks).
//
// Per panel k (right-looking):
//   1. owner(k) factors the 512 x 512 diagonal block                                   (local)
//   2. the factored block goes to every rank                      2 MB    all-reduce of a zero-padded buffer
//   3. every rank solves ITS rows of the panel: X <- X L_kk^-T                          (panel_trsm_kernel, local)
//   4. the panel is gathered: every rank needs all rows of it      (n - k0) * 512 * 8 B   all-gather + unpack
//   5. every rank updates its rows of the trailing matrix          fp64 MFMA GEMM with the block-cyclic lower
//                                                                  tile predicate (gemm_nt_sub, CyclicLower)
// The right-hand side is carried by EVERY rank as one extra local row (replicated, 1 row), so the forward
// substitution happens inside steps 3/5 like on one GPU (gdml_chol_set_rhs).  Backward substitution: the owner of
// block k solves L_kk^T x_k = z_k - sum_{i>k} L[i,k]^T x_i; the sum is spread over the ranks that own the rows i, each
// keeps an accumulator and one 512-double all-reduce per step collects it.
// Collective volume per rank: sum_k (n - k0) 512 * 8 B = 4 n^2 B (16 GB at n = 63 000) gathered over the whole
// factorisation, against n^3 / (3 W) flops: at 8 GPUs ~0.1 s of xGMI time for ~0.2 s of MFMA time; steps 1-3 of panel
// k+1 only need the first 512 columns of update k, so the gather can be overlapped with the rest of the update (not
// done yet: everything runs on the compute stream in order).
#include "common.h"

namespace {

struct Cyc {
  int W, rank;
  int64_t n, nb, nblk;
  __host__ __device__ int64_t rows_of(int64_t b) const { return (n - b * nb < nb) ? n - b * nb : nb; }
  __host__ __device__ int64_t nloc_blocks(int r) const { return nblk > r ? (nblk - r + W - 1) / W : 0; }
  __host__ __device__ int64_t local_rows(int r) const {  // blocks are full except possibly the globally last one
    const int64_t lb = nloc_blocks(r);
    if (lb == 0) return 0;
    const int64_t last_glob = (lb - 1) * W + r;
    return (lb - 1) * nb + rows_of(last_glob);
  }
  // number of rank r's blocks with global index <= k  (= first local block below panel k)
  __host__ __device__ int64_t lb0(int r, int64_t k) const { return k >= r ? (k - r) / W + 1 : 0; }
};

// Lbuf (nb x nb, ld = nb, zero above the diagonal and outside w x w) <- factored diagonal block; [nb*nb] <- info
__global__ void __launch_bounds__(256) pack_diag_kernel(const double* __restrict__ D, int64_t ld, int w, int nb,
                                                        double* __restrict__ Lbuf, const int* __restrict__ info) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < (int64_t)nb * nb) {
    const int r = (int)(e / nb), c = (int)(e % nb);
    Lbuf[e] = (r < w && c <= r) ? D[(int64_t)r * ld + c] : 0.0;
  }
  if (e == 0) Lbuf[(int64_t)nb * nb] = (double)(*info);
}

// rows of the solved panel (m x w at stride ld) -> contiguous chunk (m x nb)
__global__ void __launch_bounds__(256) pack_rows_kernel(const double* __restrict__ X, int64_t ld, int64_t m, int w, int nb,
                                                        double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= m * nb) return;
  const int64_t r = e / nb;
  const int c = (int)(e % nb);
  out[e] = c < w ? X[r * ld + c] : 0.0;
}

// gathered rank-major chunks -> panel in global row order: P[g - t0] = G[owner(g)][local index of g below the panel]
__global__ void __launch_bounds__(256) unpack_panel_kernel(const double* __restrict__ G, int64_t chunk, Cyc c, int64_t k,
                                                           int64_t t0, double* __restrict__ P) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t rows = c.n - t0;
  if (e >= rows * c.nb) return;
  const int64_t g = t0 + e / c.nb;
  const int col = (int)(e % c.nb);
  const int64_t b = g / c.nb;
  const int r = (int)(b % c.W);
  const int64_t l = (b / c.W - c.lb0(r, k)) * c.nb + g % c.nb;
  P[e] = G[(int64_t)r * chunk + l * c.nb + col];
}

// acc[c] += sum_r L[r][c] x[r]  for c < ncols  (rows of one block times its solved x)
__global__ void __launch_bounds__(256) gemv_t_acc_kernel(const double* __restrict__ L, int64_t ld, int w, int64_t ncols,
                                                         const double* __restrict__ x, double* __restrict__ acc) {
  __shared__ double xs[512];
  for (int r = threadIdx.x; r < w; r += 256) xs[r] = x[r];
  __syncthreads();
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= ncols) return;
  double s = 0.0;
  for (int r = 0; r < w; ++r) s += L[(int64_t)r * ld + c] * xs[r];
  acc[c] += s;
}

__global__ void __launch_bounds__(256) sub_kernel(double* __restrict__ out, const double* __restrict__ a,
                                                  const double* __restrict__ b, int n) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < n) out[t] = a[t] - b[t];
}

}  // namespace

extern "C" int gdml_dist_chol_solve(gdml_ctx* ctx, double sig, double lam, const double* y, int64_t n_in,
                                    double* alphas_out, int* info_out) {
  if (!ctx || !y || !alphas_out) return GDML_ERR_INVALID;
  if (info_out) *info_out = 0;
  TrainSet& ts = ctx->ts;
  if (!ts.x) return gdml_fail(ctx, GDML_ERR_STATE, "gdml_dist_chol_solve: call gdml_train_upload first");
  if (ctx->virtual_rank) return gdml_fail(ctx, GDML_ERR_STATE, "gdml_dist_chol_solve needs a real communicator");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int64_t N3 = 3 * (int64_t)ts.N, n = ts.M * N3;
  if (n_in != n) return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_dist_chol_solve: n mismatch");
  Cyc c;
  c.W = ctx->world > 0 ? ctx->world : 1;
  c.rank = ctx->rank;
  c.n = n;
  c.nb = (int64_t)ctx_opt(ctx, "dist.nb", 512);
  if (c.nb % 128 != 0 || c.nb < 128 || c.nb > 512) return gdml_fail(ctx, GDML_ERR_INVALID, "dist.nb must be 128, 256, 384 or 512");
  c.nblk = (n + c.nb - 1) / c.nb;
  const int64_t nb = c.nb, Lr = c.local_rows(c.rank), ld = (n + 15) / 16 * 16;
  int64_t max_lr = 0;
  for (int r = 0; r < c.W; ++r) max_lr = max_lr > c.local_rows(r) ? max_lr : c.local_rows(r);
  hipStream_t st = ctx->stream;

  // ---- local matrix: my row blocks + one replicated right-hand-side row, in the context's matrix buffer
  const int64_t need = (Lr + 1) * ld * 8;
  if (ctx->K && ctx->K_bytes != need) {
    GDML_TRY(ctx_free(ctx, ctx->K));
    ctx->K = nullptr;
  }
  if (!ctx->K) {
    GDML_TRY(ctx_alloc(ctx, (void**)&ctx->K, need));
    ctx->K_bytes = need;
  }
  ctx->precon = nullptr;
  ctx->K_rows = Lr; ctx->K_cols = n; ctx->K_extra = 1; ctx->K_ld = ld;
  ctx->K_factored = false; ctx->K_is_A = false; ctx->K_rhs_row = false; ctx->K_sharded = true;
  ctx->K_destroyed = true;  // nothing but this function understands the layout
  double* A = ctx->K;

  void* tmp = nullptr;
  const int64_t chunk = (max_lr + 1) * nb;                    // one rank's rows of a panel (+ rhs row), padded
  const int64_t tmp_doubles = (nb * nb + 16) + c.W * chunk + n * nb + 3 * n + 2 * nb + 64 + c.W;
  GDML_TRY(ctx_alloc(ctx, &tmp, tmp_doubles * 8));
  double* Lbuf = (double*)tmp;                 // nb x nb (+ info slot)
  double* G = Lbuf + nb * nb + 16;             // W chunks
  double* P = G + c.W * chunk;                 // panel in global row order
  double* d_acc = P + n * nb;                  // backward substitution: accumulator, solution, z
  double* d_x = d_acc + n;
  double* d_z = d_x + n;
  double* d_blk = d_z + n;                     // 2 nb scratch
  int rc = GDML_OK, info = 0;

  auto body = [&]() -> int {
    // ---- assembly of my rows (lower blocks of A = -K + lam I) and the right-hand side
    phase_begin(ctx);
    GDML_TRY(assemble_cyclic_launch(ctx, sig, lam, A, ld, c.W, c.rank, (int)nb));
    GDML_TRY(phase_end(ctx, "assemble"));
    HIP_CHECK(ctx, hipMemcpyAsync(A + Lr * ld, y, n * 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(ctx, hipStreamSynchronize(st));  // y is the caller's pageable array
    HIP_CHECK(ctx, hipMemsetAsync(ctx->d_info, 0, sizeof(int), st));

    // ---- factorisation
    phase_begin(ctx);
    for (int64_t k = 0; k < c.nblk; ++k) {
      const int64_t k0 = k * nb, w = c.rows_of(k), t0 = k0 + w;
      const int owner = (int)(k % c.W);
      if (owner == c.rank) {
        const int64_t lr0 = (k / c.W) * nb;
        // diagonal block at local rows lr0.., columns k0..: address it like a global square matrix of order k0 + w
        double* Av = A + (lr0 - k0) * ld;
        GDML_TRY(panel_factor_steps(ctx, st, Av, k0 + w, ld, k0, w));
        hipLaunchKernelGGL(pack_diag_kernel, dim3(ceil_div(nb * nb, 256)), dim3(256), 0, st, A + lr0 * ld + k0, ld, (int)w,
                           (int)nb, Lbuf, ctx->d_info);
      } else {
        HIP_CHECK(ctx, hipMemsetAsync(Lbuf, 0, (nb * nb + 1) * 8, st));
      }
      GDML_TRY(comm_allreduce_sum(ctx, Lbuf, nb * nb + 1));
      // my rows below the panel (blocks with global index > k) + the right-hand-side row
      const int64_t lb0 = c.lb0(c.rank, k);
      const int64_t r_below = lb0 * nb < Lr ? lb0 * nb : Lr;  // (the globally last block may be short: then nothing but
      const int64_t m_blk = Lr - r_below;                     //  the right-hand-side row follows it)
      const int64_t m_all = m_blk + 1;
      double* X = A + r_below * ld + k0;
      if (w % 64 == 0) {
        GDML_TRY(launch_panel_trsm(ctx, st, Lbuf, X, ld, (int)w, m_all, nb));
      } else {  // ragged last block: 64-wide steps
        for (int64_t jj = 0; jj < w; jj += 64) {
          const int ww = (int)((w - jj < 64) ? w - jj : 64);
          GDML_TRY(launch_trsm64(ctx, st, Lbuf + jj * nb + jj, X + jj, ld, ww, m_all, nb));
          const int64_t rest = w - jj - ww;
          if (rest > 0)
            GDML_TRY(launch_gemm_nt_sub(ctx, st, X + jj, ld, Lbuf + (jj + ww) * nb + jj, nb, X + jj + ww, ld, m_all, rest, ww, 0));
        }
      }
      if (t0 >= n) break;
      // gather the panel rows of all ranks (block rows only; the rhs row is nobody's column)
      int64_t m_pad = 0;
      for (int r = 0; r < c.W; ++r) {
        const int64_t mr = c.local_rows(r) - c.lb0(r, k) * nb;
        if (mr > m_pad) m_pad = mr;
      }
      const int64_t ck = m_pad * nb;
      if (m_blk > 0)
        hipLaunchKernelGGL(pack_rows_kernel, dim3(ceil_div(m_blk * nb, 256)), dim3(256), 0, st, X, ld, m_blk, (int)w, (int)nb,
                           G + (int64_t)c.rank * ck);
      GDML_TRY(comm_allgather_inplace(ctx, G, ck));
      hipLaunchKernelGGL(unpack_panel_kernel, dim3(ceil_div((n - t0) * nb, 256)), dim3(256), 0, st, G, ck, c, k, t0, P);
      // trailing update of my rows: C[my rows, t0:n] -= X_mine P^T, lower tiles of the cyclic layout only
      CyclicLower cl;
      cl.W = c.W; cl.rank = c.rank; cl.lb0 = lb0; cl.nb = nb; cl.col0 = t0; cl.block_rows = m_blk;
      GDML_TRY(launch_gemm_nt_sub_cyclic(ctx, st, X, ld, P, nb, A + r_below * ld + t0, ld, m_all, n - t0, w, cl));
      HIP_CHECK(ctx, hipGetLastError());
    }
    GDML_TRY(phase_end(ctx, "factor"));
    // first failing pivot over all ranks (0 = none): every rank reports its own in slot `rank` of a summed vector
    {
      int my_info = 0;
      HIP_CHECK(ctx, hipMemcpyAsync(&my_info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_CHECK(ctx, hipStreamSynchronize(st));
      std::vector<double> v((size_t)c.W, 0.0);
      v[(size_t)c.rank] = (double)my_info;
      HIP_CHECK(ctx, hipMemcpyAsync(d_blk, v.data(), c.W * 8, hipMemcpyHostToDevice, st));
      HIP_CHECK(ctx, hipStreamSynchronize(st));
      GDML_TRY(comm_allreduce_sum(ctx, d_blk, c.W));
      HIP_CHECK(ctx, hipMemcpyAsync(v.data(), d_blk, c.W * 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(ctx, hipStreamSynchronize(st));
      for (double f : v)
        if (f > 0.0 && (info == 0 || (int)f < info)) info = (int)f;
    }
    if (info != 0) return GDML_OK;

    // ---- backward substitution  L^T x = z   (z = my copy of the carried right-hand-side row)
    phase_begin(ctx);
    HIP_CHECK(ctx, hipMemcpyAsync(d_z, A + Lr * ld, n * 8, hipMemcpyDeviceToDevice, st));
    HIP_CHECK(ctx, hipMemsetAsync(d_acc, 0, 2 * n * 8, st));  // acc and x
    for (int64_t k = c.nblk - 1; k >= 0; --k) {
      const int64_t k0 = k * nb, w = c.rows_of(k);
      double* sblk = d_blk;  // sum over ranks of the accumulators of this block
      HIP_CHECK(ctx, hipMemcpyAsync(sblk, d_acc + k0, w * 8, hipMemcpyDeviceToDevice, st));
      GDML_TRY(comm_allreduce_sum(ctx, sblk, w));
      if ((int)(k % c.W) == c.rank) {
        const int64_t lr0 = (k / c.W) * nb;
        double* rhs = d_blk + nb;
        hipLaunchKernelGGL(sub_kernel, dim3(ceil_div(w, 256)), dim3(256), 0, st, rhs, d_z + k0, sblk, (int)w);
        GDML_TRY(chol_bwd_device(ctx, A + lr0 * ld + k0, w, ld, rhs, d_x + k0));
        if (k0 > 0)
          hipLaunchKernelGGL(gemv_t_acc_kernel, dim3(ceil_div(k0, 256)), dim3(256), 0, st, A + lr0 * ld, ld, (int)w, k0,
                             d_x + k0, d_acc);
      }
    }
    GDML_TRY(comm_allreduce_sum(ctx, d_x, n));  // every block of x was written by exactly one rank
    GDML_TRY(phase_end(ctx, "solve"));
    HIP_CHECK(ctx, hipMemcpyAsync(alphas_out, d_x, n * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(ctx, hipStreamSynchronize(st));
    for (int64_t i = 0; i < n; ++i) alphas_out[i] = -alphas_out[i];  // analytic.py:99
    return GDML_OK;
  };
  rc = body();
  int rc2 = ctx_free(ctx, tmp);
  if (rc != GDML_OK) return rc;
  if (rc2 != GDML_OK) return rc2;
  if (info_out) *info_out = info;
  if (info != 0)
    return gdml_fail(ctx, GDML_ERR_NOT_PD, "%d-th leading minor of the array is not positive definite", info);
  return GDML_OK;
}
