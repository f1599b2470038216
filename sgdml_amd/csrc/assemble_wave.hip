// Register-resident kernel-matrix assembly for molecules with 3N <= 64 (one output column per lane)
// and a single identity permutation (plain GDML, the benchmark configuration).
//
// One wavefront owns one row point i and walks over consecutive column points j.  Lane l < 3N is
// the output column c = (b, beta) of block (i, j).  Everything the lane needs is private:
//     Ri[al][m] = G_i(b,m)[al]   (3N doubles, resident for the whole walk)
//     Xi[m]     = x_i[pair(b,m)] (N doubles, resident)
//     Rj[m]     = G_j(b,m)[beta] , Xj[m] = x_j[pair(b,m)]      (2N loads per j, prefetched)
// with G_x(b,m) = (r_m - r_b)/d^3 read from dense, m-major tables built once per upload
// (coalesced 8*3N-byte rows).  Per block:
//     Dv[m] = Xi[m] - Xj[m];  |d|^2 = (1/6) sum over lanes of sum_m Dv^2   (each atom has 3 lanes)
//     u[c]  = sum_m Dv[m] Rj[m];   v[(b,al)] = sum_m Dv[m] Ri[al][m];   dg[al] = sum_m Ri[al][m] Rj[m]
//     K[(a,al), c] = 5 b_p v[(a,al)] u[c] + c_p Ri[al][a] Rj[a]        (a != b)
//                  = 5 b_p v[(b,al)] u[c] - c_p dg[al]                 (a == b)
// (same formula as assemble.hip with G_i(a,b) = -G_i(b,a)).  Only v crosses lanes (64-entry LDS
// vector) plus one wave reduction for the norm; no workgroup barriers.  ~700 instructions per
// 63x63 block against 31.7 KB written: the kernel is bound by the HBM write stream, and because a
// wave writes the 504-byte row segments of consecutive j back to back the partially covered cache
// lines merge in the XCD's L2.
#include "common.h"

// dense tables: XF[i][m][b] = x_i[pair(b,m)] (0 if m == b), GD[i][m][b][al] = G_i(b,m)[al]
__global__ void __launch_bounds__(256) dense_tables_kernel(const double* __restrict__ x,
                                                           const double* __restrict__ g, int64_t M,
                                                           int N, int D, double* __restrict__ XF,
                                                           double* __restrict__ GD) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t NN = (int64_t)N * N;
  if (t >= M * NN) return;
  const int64_t i = t / NN;
  const int mb = (int)(t - i * NN);
  const int m = mb / N, b = mb - m * N;
  double xv = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
  if (m != b) {
    const int k = pair_idx(b, m);
    xv = x[i * D + k];
    // compressed g[k] = (r_hi - r_lo)/d^3 ; G(b,m) = (r_m - r_b)/d^3 = +g if b < m else -g
    const double s = (b < m) ? 1.0 : -1.0;
    g0 = s * g[(i * D + k) * 3 + 0];
    g1 = s * g[(i * D + k) * 3 + 1];
    g2 = s * g[(i * D + k) * 3 + 2];
  }
  XF[t] = xv;
  GD[t * 3 + 0] = g0;
  GD[t * 3 + 1] = g1;
  GD[t * 3 + 2] = g2;
}

struct WaveArgs {
  const double* XF;
  const double* GD;
  int64_t M;
  int N;
  double sig;
  int use_E;
  int64_t e_row0;  // energy-constraint row of point i is row e_row0 + i of K: 3N M, or (sharded rows) 3N (i_end - i_beg) - i_beg
  const int32_t* jlist;
  const int32_t* colmap;
  int64_t j0, n_j;
  int64_t i_beg;  // first row point handled by this launch (rows are written relative to it)
  int j_chunk;
  double* K;
  int64_t ld;
  int lower;      // 1: only blocks j <= i (dense column range), values stored as -K, + lam on the diagonal:
  double lam;     //    the matrix A = -K + lam I in the form the Cholesky factorisation reads (analytic.py:65,82)
  // distributed Cholesky (CYC instantiation): K is this rank's share of a block-row-cyclic layout -- global row
  // block b (cyc_nb rows) lives on rank b % cyc_W as local row block b / cyc_W; rows of other ranks are not stored
  int cyc_W, cyc_rank, cyc_nb;
};

template <int N, bool CYC = false>
__global__ void __launch_bounds__(64) assemble_wave_kernel(WaveArgs A) {
  constexpr int N3 = 3 * N;
  __shared__ double vsh[64];
  const int lane = threadIdx.x;
  const bool act = lane < N3;
  const int c = act ? lane : 0;
  const int b = c / 3, beta = c - 3 * b;
  const int64_t i = A.i_beg + blockIdx.x;
  const int64_t jb_beg = (int64_t)blockIdx.y * A.j_chunk;
  int64_t jb_end = (jb_beg + A.j_chunk < A.n_j) ? jb_beg + A.j_chunk : A.n_j;
  if (A.lower && jb_end > i - A.j0 + 1) jb_end = i - A.j0 + 1;  // blocks on/below the block diagonal only
  if (jb_beg >= jb_end) return;
  const double sgn = A.lower ? -1.0 : 1.0;
  // CYC: the 3N rows of point i fall into at most two row blocks; local row of block-relative row r is
  // base0 + r (r < rb) or base1 + (r - rb); base < 0 = not this rank's block
  int64_t base0 = 0, base1 = 0;
  int rb = N3;
  if (CYC) {
    const int64_t g0 = i * N3, b0 = g0 / A.cyc_nb;
    const int64_t rest = (b0 + 1) * A.cyc_nb - g0;
    rb = rest < N3 ? (int)rest : N3;
    base0 = (b0 % A.cyc_W == A.cyc_rank) ? (b0 / A.cyc_W) * A.cyc_nb + g0 % A.cyc_nb : -1;
    base1 = ((b0 + 1) % A.cyc_W == A.cyc_rank) ? ((b0 + 1) / A.cyc_W) * A.cyc_nb : -1;
    if (base0 < 0 && (base1 < 0 || rb == N3)) return;
  }

  const double sig = A.sig, inv_sig = 1.0 / sig;
  const double sqrt5 = 2.23606797749978969641;
  const double base_div = 5.0 / (3.0 * sig * sig * sig * sig);
  const double e_fact = 5.0 / (3.0 * sig * sig * sig);

  // resident row-point data
  double Ri[3][N], Xi[N];
  {
    const double* gd = A.GD + i * (int64_t)N * N3;
    const double* xf = A.XF + i * (int64_t)N * N;
#pragma unroll
    for (int m = 0; m < N; ++m) {
      Xi[m] = xf[m * N + b];
#pragma unroll
      for (int al = 0; al < 3; ++al) Ri[al][m] = gd[m * N3 + 3 * b + al];
    }
  }
  // prefetch of the first column point
  double Rj[N], Xj[N];
  auto fetch = [&](int64_t jb) {
    const int64_t j = A.jlist ? A.jlist[jb] : A.j0 + jb;
    const double* gd = A.GD + j * (int64_t)N * N3 + c;
    const double* xf = A.XF + j * (int64_t)N * N + b;
#pragma unroll
    for (int m = 0; m < N; ++m) {
      Rj[m] = gd[m * N3];
      Xj[m] = xf[m * N];
    }
  };
  fetch(jb_beg);

  for (int64_t jb = jb_beg; jb < jb_end; ++jb) {
    // consume the prefetched column point
    double Dv[N], rj[N];
    double ss = 0.0, u = 0.0, v0 = 0.0, v1 = 0.0, v2 = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma unroll
    for (int m = 0; m < N; ++m) {
      rj[m] = Rj[m];
      Dv[m] = Xi[m] - Xj[m];
    }
    if (jb + 1 < jb_end) fetch(jb + 1);
#pragma unroll
    for (int m = 0; m < N; ++m) {
      const double d = Dv[m];
      ss += d * d;
      u += d * rj[m];
      v0 += d * Ri[0][m];
      v1 += d * Ri[1][m];
      v2 += d * Ri[2][m];
      g0 += Ri[0][m] * rj[m];
      g1 += Ri[1][m] * rj[m];
      g2 += Ri[2][m] * rj[m];
    }
    // |d|^2: every atom is counted by its 3 lanes and every pair twice
    const double nrm2 = wave_sum(act ? ss : 0.0) * (1.0 / 6.0);
    const double nrm = sqrt5 * sqrt(nrm2);
    const double ex = exp(-nrm * inv_sig);
    const double bp = sgn * ex * base_div;  // sign of the stored matrix folded into both coefficients
    const double cp = (sig * sig + sig * nrm) * bp;
    const double uc = 5.0 * bp * u;
    // exchange v through LDS (single wavefront: LDS operations are in order)
    __builtin_amdgcn_wave_barrier();
    vsh[lane] = (beta == 0) ? v0 : (beta == 1 ? v1 : v2);
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();

    const int64_t outcol = A.colmap ? (int64_t)A.colmap[jb * N3 + c] : jb * N3 + c;
    if (act && outcol >= 0) {
      double* dst = A.K + ((i - A.i_beg) * N3) * A.ld + outcol;
#pragma unroll
      for (int a = 0; a < N; ++a) {
        const double w = cp * rj[a];
        double o0 = vsh[3 * a + 0] * uc + Ri[0][a] * w;
        double o1 = vsh[3 * a + 1] * uc + Ri[1][a] * w;
        double o2 = vsh[3 * a + 2] * uc + Ri[2][a] * w;
        if (a == b) {
          o0 = vsh[3 * a + 0] * uc - cp * g0;
          o1 = vsh[3 * a + 1] * uc - cp * g1;
          o2 = vsh[3 * a + 2] * uc - cp * g2;
          if (A.lower && A.j0 + jb == i) {  // diagonal block: + lam on the matrix diagonal (row 3a+al == column c)
            o0 += (beta == 0) ? A.lam : 0.0;
            o1 += (beta == 1) ? A.lam : 0.0;
            o2 += (beta == 2) ? A.lam : 0.0;
          }
        }
        if (CYC) {
          const double o[3] = {o0, o1, o2};
#pragma unroll
          for (int al = 0; al < 3; ++al) {
            const int r = 3 * a + al;
            const int64_t lr = r < rb ? (base0 < 0 ? -1 : base0 + r) : (base1 < 0 ? -1 : base1 + (r - rb));
            if (lr >= 0) A.K[lr * A.ld + outcol] = o[al];
          }
        } else {
          dst[0] = o0;
          dst[A.ld] = o1;
          dst[2 * A.ld] = o2;
          dst += 3 * A.ld;
        }
      }
      if (A.use_E) A.K[(A.e_row0 + i) * A.ld + outcol] = -e_fact * (nrm + sig) * ex * u;  // train.py:235-248 (never with lower)
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Build (or rebuild) the dense tables for the resident training set.
int build_dense_tables(gdml_ctx* ctx) {
  TrainSet& ts = ctx->ts;
  const int64_t NN = (int64_t)ts.N * ts.N;
  if (ts.XF) return GDML_OK;
  GDML_TRY(ctx_alloc(ctx, (void**)&ts.XF, ts.M * NN * 8));
  GDML_TRY(ctx_alloc(ctx, (void**)&ts.GD, ts.M * NN * 24));
  hipLaunchKernelGGL(dense_tables_kernel, dim3(ceil_div(ts.M * NN, 256)), dim3(256), 0, ctx->stream, ts.x,
                     ts.g, ts.M, ts.N, ts.D, ts.XF, ts.GD);
  ctx->launch_counter++;
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}

bool assemble_wave_applicable(const gdml_ctx* ctx) {
  const TrainSet& ts = ctx->ts;
  if (!ctx_opt_i(ctx, "asm.wave", 1)) return false;
  if (ts.P != 1 || ts.N > 21 || ts.N < 2) return false;
  for (int a = 0; a < ts.N; ++a)
    if (ts.h_perm[a] != a) return false;
  return true;
}

int assemble_wave_launch(gdml_ctx* ctx, double sig, int use_E, const int32_t* d_jlist,
                         const int32_t* d_colmap, int64_t j0, int64_t n_j, double* K, int64_t ld,
                         int64_t i_beg, int64_t i_end, int lower, double lam, int cyc_W, int cyc_rank, int cyc_nb) {
  TrainSet& ts = ctx->ts;
  GDML_TRY(build_dense_tables(ctx));
  WaveArgs A;
  A.XF = ts.XF; A.GD = ts.GD; A.M = ts.M; A.N = ts.N; A.sig = sig; A.use_E = use_E;
  A.e_row0 = (i_beg == 0 && i_end == ts.M) ? ts.M * 3 * (int64_t)ts.N : (i_end - i_beg) * 3 * (int64_t)ts.N - i_beg;
  A.jlist = d_jlist; A.colmap = d_colmap; A.j0 = j0; A.n_j = n_j; A.K = K; A.ld = ld;
  A.i_beg = i_beg;
  A.lower = lower;
  A.lam = lam;
  A.cyc_W = cyc_W; A.cyc_rank = cyc_rank; A.cyc_nb = cyc_nb;
  if (cyc_W > 0 && !lower) return gdml_fail(ctx, GDML_ERR_INVALID, "assemble_wave: the row-cyclic layout is only built in the lower form");
  if (lower && (d_jlist || d_colmap || use_E || j0 != 0 || i_beg != 0))
    return gdml_fail(ctx, GDML_ERR_INVALID, "assemble_wave: lower form needs the dense full column range");
  const int64_t n_i = i_end - i_beg;
  if (n_i <= 0) return GDML_OK;
  // column points walked by one wavefront: 64 for full rows; 32 in the lower form (measured 3.90 vs 4.06-4.22 ms at the
  // benchmark size, tools/asm_lower_probe.py: shorter walks balance the triangular rows better)
  int j_chunk = ctx_opt_i(ctx, "asm.j_chunk", lower ? 32 : 64);
  while (j_chunk > 8 && n_i * ((n_j + j_chunk - 1) / j_chunk) < 8192) j_chunk >>= 1;
  A.j_chunk = j_chunk;
  dim3 grid((unsigned)n_i, (unsigned)((n_j + j_chunk - 1) / j_chunk));
  const int slot = ktime_begin(ctx);
  switch (ts.N) {
#define WC(v)                                                                                          \
  case v:                                                                                              \
    if (cyc_W > 0) hipLaunchKernelGGL((assemble_wave_kernel<v, true>), grid, dim3(64), 0, ctx->stream, A); \
    else hipLaunchKernelGGL((assemble_wave_kernel<v, false>), grid, dim3(64), 0, ctx->stream, A);          \
    break;
    WC(2) WC(3) WC(4) WC(5) WC(6) WC(7) WC(8) WC(9) WC(10) WC(11) WC(12) WC(13) WC(14) WC(15) WC(16)
    WC(17) WC(18) WC(19) WC(20) WC(21)
#undef WC
    default: return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "assemble_wave: N out of range");
  }
  // algorithmic bytes: every requested element written once (lower form: the n_i (n_i + 1) / 2 blocks with j <= i)
  const double blocks = lower ? 0.5 * (double)n_i * (double)(n_i + 1) : (double)n_i * (double)n_j;
  ktime_end(ctx, slot, "assemble", 8.0 * blocks * 9.0 * ts.N * ts.N);
  ctx->launch_counter++;
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}
