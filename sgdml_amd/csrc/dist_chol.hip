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
// Per panel k (right-looking; with option dist.lookahead = 1 one panel of look-ahead over three streams, by default the
// same steps in order on one stream):
//   critical stream   1. every rank solves ITS rows of the panel: X <- X L_kk^-T          (panel_trsm_kernel, local)
//                     2. owner(k+1) sends its solved rows of block k+1 to every rank   2 MB    ncclBroadcast
//                     3. every rank updates the NEXT panel's columns of its rows          (K = 512 GEMM, 512 columns)
//                     4. owner(k+1) factors the 512 x 512 diagonal block k+1 and sends it   2 MB    ncclBroadcast
//   collective stream 5. the panel is gathered: every rank needs all rows of it   (n - k0) * 512 * 8 B   all-gather + unpack
//   bulk stream       6. every rank updates the columns right of block k+1 of its rows    fp64 MFMA GEMM with the
//                                                                  block-cyclic lower tile predicate (CyclicLower)
// Steps 1-4 of panel k+1 run while step 6 of panel k is still busy, step 5 of panel k runs under step 6 of panel k-1
// (two panel buffers).  One rank, n = 63 000 (profiles/r03_dist_chol_lookahead_1rank.txt): 1.53 s against 1.39 s for the
// single-GPU schedule with its fused diagonal-block role -- the panel chain is hidden.  All collectives are issued by one
// host thread in the same order on every rank; with ONE communicator RCCL runs them in that order, so the block
// broadcast of step 4 queues behind the gather of step 5 (shorter than step 6 from W = 2 on).  The host-staged backend
// (ranks sharing a GPU: tests) runs every collective synchronously.  The RCCL branch has not run on more than one GPU
// yet (tests/test_hip_scale.py::test_distributed_cholesky_rccl_one_gpu_per_rank is skipped on one-GPU boxes).
// Energy constraints (train.py:235-300): n = 3N M + M; the energy rows are the last M global rows (assemble_erows_cyclic_launch).
// The right-hand side is carried by EVERY rank as one extra local row (replicated, 1 row), so the forward
// substitution happens inside steps 1/3/6 like on one GPU (gdml_chol_set_rhs).  Backward substitution: the owner of
// block k solves L_kk^T x_k = z_k - sum_{i>k} L[i,k]^T x_i; the sum is spread over the ranks that own the rows i, each
// keeps an accumulator and one 512-double all-reduce per step collects it.
// Collective volume per rank: sum_k (n - k0) 512 * 8 B = 4 n^2 B (16 GB at n = 63 000) gathered over the whole
// factorisation, against n^3 / (3 W) flops: at 8 GPUs ~0.1 s of xGMI time for ~0.2 s of MFMA time.
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
  // Energy constraints (train.py:235-300, round 6): y carries the M energy labels behind the forces, the system has M more
  // rows / columns.  The force rows keep their place in the block-row-cyclic layout (it is defined by the global row index),
  // the energy rows 3N M + e follow in the last row blocks and are assembled by assemble_erows_cyclic_launch on their owners.
  const int64_t N3 = 3 * (int64_t)ts.N, n_ff = ts.M * N3;
  const bool use_E = n_in == n_ff + ts.M;
  const int64_t n = n_ff + (use_E ? ts.M : 0);
  if (n_in != n)
    return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_dist_chol_solve: y must hold 3N M values, or 3N M + M with energy constraints");
  if ((ctx->world <= 1) && ctx_opt_i(ctx, "dist.force_panels", 0) == 0) {
    // One rank: the block-row-cyclic layout IS the plain layout and no collective moves data, so the single-GPU schedule
    // applies as it stands -- K = 1024 trailing updates with the diagonal block factored inside them (chol.hip) instead of
    // this file's K = 512 panels (1.39-1.53 s against 1.27 s at n = 63 000, profiles/r03_dist_chol_lookahead_1rank.txt).
    // Option dist.force_panels = 1 keeps the panel schedule (tests and probes of the distributed code on one rank).
    GDML_TRY(gdml_assemble_A(ctx, sig, lam, use_E ? 1 : 0, 1));
    GDML_TRY(gdml_chol_set_rhs(ctx, y, n));
    int inf = 0;
    const int rc1 = gdml_chol_factor(ctx, lam, &inf);
    if (info_out) *info_out = inf;
    if (rc1 != GDML_OK) return rc1;
    return gdml_chol_solve(ctx, nullptr, n, 0, alphas_out);
  }
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
  const int64_t tmp_doubles = 2 * (nb * nb + 16) + 2 * c.W * chunk + 2 * n * nb + 3 * n + 2 * nb + 64 + c.W;
  GDML_TRY(ctx_alloc(ctx, &tmp, tmp_doubles * 8));
  double* Lbuf = (double*)tmp;                 // nb x nb (+ info slot): factored diagonal block of the current panel
  double* Pn = Lbuf + nb * nb + 16;            // nb x nb: solved panel rows of the NEXT block (look-ahead)
  double* Gb[2] = {Pn + nb * nb + 16, Pn + nb * nb + 16 + c.W * chunk};  // W chunks, two panels in flight
  double* Pb[2] = {Gb[1] + c.W * chunk, Gb[1] + c.W * chunk + n * nb};   // panel in global row order, two in flight
  double* d_acc = Pb[1] + n * nb;              // backward substitution: accumulator, solution, z
  double* d_x = d_acc + n;
  double* d_z = d_x + n;
  double* d_blk = d_z + n;                     // 2 nb scratch
  int rc = GDML_OK, info = 0;

  auto body = [&]() -> int {
    // ---- assembly of my rows (lower blocks of A = -K + lam I) and the right-hand side
    phase_begin(ctx);
    GDML_TRY(assemble_cyclic_launch(ctx, sig, lam, A, ld, c.W, c.rank, (int)nb));
    if (use_E) GDML_TRY(assemble_erows_cyclic_launch(ctx, sig, lam, A, ld, c.W, c.rank, (int)nb));
    GDML_TRY(phase_end(ctx, "assemble"));
    HIP_CHECK(ctx, hipMemcpyAsync(A + Lr * ld, y, n * 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(ctx, hipStreamSynchronize(st));  // y is the caller's pageable array
    HIP_CHECK(ctx, hipMemsetAsync(ctx->d_info, 0, sizeof(int), st));

    // ---- factorisation.  Option dist.lookahead = 1: one panel of look-ahead over three streams:
    //   sa (critical path)  panel solve k -> broadcast of the solved rows of block k+1 -> update of the NEXT panel's columns
    //                       -> owner(k+1) factors diagonal block k+1 -> broadcast of it   (broadcasts: second communicator)
    //   sn (collective)     pack + all-gather + unpack of the whole panel k            (under the bulk update of panel k-1)
    //   sb (bulk)           update of the columns right of block k+1 by panel k         (under the critical path of k+1)
    // dist.lookahead = 0: the same steps in the same order on ONE stream and one communicator -- the schedule that needs
    // nothing from RCCL beyond in-order execution.  The three-stream schedule issues collectives of two communicators from
    // two streams (always from this one host thread, in the same order on every rank); it has run with one rank and with
    // host-staged collectives (2-3 processes sharing the GPU), not yet on several physical GPUs.
    phase_begin(ctx);
    HIP_CHECK(ctx, hipStreamSynchronize(st));  // assembly, right-hand side, info slot: visible to the other streams
    // default (round 6): look-ahead from two ranks on -- the per-panel model in DESIGN.md section 5: without it the 4 n^2 / W ... 4 n^2
    // bytes of panel all-gather per rank are serialised with the update on one stream and exposed in full
    const bool la = ctx_opt_i(ctx, "dist.lookahead", c.W > 1 ? 1 : 0) != 0;
    hipStream_t sa = la ? ctx->stream2 : st, sb = st, sn = st;  // stream2 is the high-priority one
    hipStream_t sn_own = nullptr;
    hipEvent_t ev_x[2] = {nullptr, nullptr}, ev_p[2] = {nullptr, nullptr}, ev_b[2] = {nullptr, nullptr};
    auto setup = [&]() -> int {  // a failure here must reach comm_abort below: the peers are about to enter a broadcast
      if (la) {
        GDML_TRY(comm_ensure_second(ctx));  // block broadcasts on their own communicator (same option on every rank)
        HIP_CHECK(ctx, hipStreamCreateWithFlags(&sn_own, hipStreamNonBlocking));
        sn = sn_own;
      }
      for (int e = 0; e < 2; ++e) {
        HIP_CHECK(ctx, hipEventCreateWithFlags(&ev_x[e], hipEventDisableTiming));
        HIP_CHECK(ctx, hipEventCreateWithFlags(&ev_p[e], hipEventDisableTiming));
        HIP_CHECK(ctx, hipEventCreateWithFlags(&ev_b[e], hipEventDisableTiming));
      }
      return GDML_OK;
    };
    bool have_b[2] = {false, false};
    auto factor_and_bcast = [&](int64_t k) -> int {  // diagonal block k: owner factors, everybody gets it (Lbuf)
      const int64_t k0 = k * nb, w = c.rows_of(k);
      const int owner = (int)(k % c.W);
      if (owner == c.rank) {
        const int64_t lr0 = (k / c.W) * nb;
        // diagonal block at local rows lr0.., columns k0..: address it like a global square matrix of order k0 + w
        double* Av = A + (lr0 - k0) * ld;
        GDML_TRY(panel_factor_steps(ctx, sa, Av, k0 + w, ld, k0, w));
        hipLaunchKernelGGL(pack_diag_kernel, dim3(ceil_div(nb * nb, 256)), dim3(256), 0, sa, A + lr0 * ld + k0, ld, (int)w,
                           (int)nb, Lbuf, ctx->d_info);
      }
      return comm_broadcast_on(ctx, Lbuf, nb * nb + 1, owner, sa, la);
    };
    auto loop = [&]() -> int {
      GDML_TRY(factor_and_bcast(0));
      for (int64_t k = 0; k < c.nblk; ++k) {
        const int64_t k0 = k * nb, w = c.rows_of(k), t0 = k0 + w;
        const int par = (int)(k & 1);
        // my rows below the panel (blocks with global index > k) + the right-hand-side row
        const int64_t lb0 = c.lb0(c.rank, k);
        const int64_t r_below = lb0 * nb < Lr ? lb0 * nb : Lr;  // (the globally last block may be short: then nothing but
        const int64_t m_blk = Lr - r_below;                     //  the right-hand-side row follows it)
        const int64_t m_all = m_blk + 1;
        double* X = A + r_below * ld + k0;
        // the columns of this panel received their last update from the bulk stream (panel k-1's update right of block k
        // covers them only for k-1's look-ahead block = these columns: done on sa) -- nothing to wait for here
        if (w % 64 == 0) {
          GDML_TRY(launch_panel_trsm(ctx, sa, Lbuf, X, ld, (int)w, m_all, nb));
        } else {  // ragged last block: 64-wide steps
          for (int64_t jj = 0; jj < w; jj += 64) {
            const int ww = (int)((w - jj < 64) ? w - jj : 64);
            GDML_TRY(launch_trsm64(ctx, sa, Lbuf + jj * nb + jj, X + jj, ld, ww, m_all, nb));
            const int64_t rest = w - jj - ww;
            if (rest > 0)
              GDML_TRY(launch_gemm_nt_sub(ctx, sa, X + jj, ld, Lbuf + (jj + ww) * nb + jj, nb, X + jj + ww, ld, m_all, rest, ww, 0));
          }
        }
        if (t0 >= n) break;
        HIP_CHECK(ctx, hipEventRecord(ev_x[par], sa));  // X of panel k is final
        const int64_t w1 = c.rows_of(k + 1);
        const int owner1 = (int)((k + 1) % c.W);
        // ---- sa: solved rows of block k+1 to everybody (they are the first rows below the panel on their owner)
        if (owner1 == c.rank)
          hipLaunchKernelGGL(pack_rows_kernel, dim3(ceil_div(w1 * nb, 256)), dim3(256), 0, sa, X, ld, w1, (int)w, (int)nb, Pn);
        GDML_TRY(comm_broadcast_on(ctx, Pn, nb * nb, owner1, sa, la));
        // ---- sn: the whole panel (block rows only; the rhs row is nobody's column)
        int64_t m_pad = 0;
        for (int r = 0; r < c.W; ++r) {
          const int64_t mr = c.local_rows(r) - c.lb0(r, k) * nb;
          if (mr > m_pad) m_pad = mr;
        }
        const int64_t ck = m_pad * nb;
        const bool bulk = n - t0 - w1 > 0;  // columns right of block k+1 exist
        if (bulk) {
          HIP_CHECK(ctx, hipStreamWaitEvent(sn, ev_x[par], 0));
          if (have_b[par]) HIP_CHECK(ctx, hipStreamWaitEvent(sn, ev_b[par], 0));  // buffers of panel k-2 are free
          if (m_blk > 0)
            hipLaunchKernelGGL(pack_rows_kernel, dim3(ceil_div(m_blk * nb, 256)), dim3(256), 0, sn, X, ld, m_blk, (int)w, (int)nb,
                               Gb[par] + (int64_t)c.rank * ck);
          GDML_TRY(comm_allgather_inplace_on(ctx, Gb[par], ck, sn));
          hipLaunchKernelGGL(unpack_panel_kernel, dim3(ceil_div((n - t0) * nb, 256)), dim3(256), 0, sn, Gb[par], ck, c, k, t0, Pb[par]);
          HIP_CHECK(ctx, hipEventRecord(ev_p[par], sn));
        }
        // ---- sa: update of the next panel's columns [t0, t0 + w1) of my rows; they took panel k-1's bulk update
        if (have_b[par ^ 1]) HIP_CHECK(ctx, hipStreamWaitEvent(sa, ev_b[par ^ 1], 0));
        CyclicLower cl;
        cl.W = c.W; cl.rank = c.rank; cl.lb0 = lb0; cl.nb = nb; cl.col0 = t0; cl.block_rows = m_blk;
        GDML_TRY(launch_gemm_nt_sub_cyclic(ctx, sa, X, ld, Pn, nb, A + r_below * ld + t0, ld, m_all, w1, w, cl));
        // ---- sa: diagonal block k+1 is complete on its owner
        GDML_TRY(factor_and_bcast(k + 1));
        // ---- sb: everything right of block k+1
        if (bulk) {
          HIP_CHECK(ctx, hipStreamWaitEvent(sb, ev_p[par], 0));
          cl.col0 = t0 + w1;
          GDML_TRY(launch_gemm_nt_sub_cyclic(ctx, sb, X, ld, Pb[par] + w1 * nb, nb, A + r_below * ld + t0 + w1, ld, m_all,
                                             n - t0 - w1, w, cl));
          HIP_CHECK(ctx, hipEventRecord(ev_b[par], sb));
          have_b[par] = true;
        }
        HIP_CHECK(ctx, hipGetLastError());
      }
      return GDML_OK;
    };
    int rc_loop = setup();
    if (rc_loop == GDML_OK) rc_loop = loop();
    (void)hipStreamSynchronize(sn);
    (void)hipStreamSynchronize(sb);
    (void)hipStreamSynchronize(sa);
    for (int e = 0; e < 2; ++e) {
      if (ev_x[e]) (void)hipEventDestroy(ev_x[e]);
      if (ev_p[e]) (void)hipEventDestroy(ev_p[e]);
      if (ev_b[e]) (void)hipEventDestroy(ev_b[e]);
    }
    if (sn_own) (void)hipStreamDestroy(sn_own);
    if (rc_loop != GDML_OK) comm_abort(ctx);  // the peers are (or will be) inside a collective this rank never joins
    GDML_TRY(rc_loop);
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
  // a LOCAL failure outside the factorisation loop (a launch error or an allocation that only this rank could not get, in the
  // assembly or the substitution): the peers are or will be inside a collective this rank never joins
  if (rc == GDML_ERR_HIP || rc == GDML_ERR_OOM) comm_abort(ctx);
  int rc2 = ctx_free(ctx, tmp);
  if (rc != GDML_OK) return rc;
  if (rc2 != GDML_OK) return rc2;
  if (info_out) *info_out = info;
  if (info != 0)
    return gdml_fail(ctx, GDML_ERR_NOT_PD, "%d-th leading minor of the array is not positive definite", info);
  return GDML_OK;
}
