// Kernel-matrix assembly in 64-COLUMN STRIPS (P = 1, identity permutation, 11 <= N <= 21): full-cache-line stores.
//
// assemble_wave_kernel maps lane = column of ONE 3N-wide block: its row segments are 8*3N = 504 bytes at a 504-byte
// pitch, so every store touches two partially covered 128-byte lines and the pattern tops out at ~4.2-4.6 TB/s
// (tools/store_bw.hip).  Here a wavefront owns the 64 consecutive GLOBAL columns [64 s, 64 s + 64) -- a 512-byte,
// 128-byte-aligned segment of every matrix row; as a pure store pattern that runs at the full-row write rate
// (6.0-6.1 TB/s, profiles/r02_store_pattern_probe.txt).  A workgroup is 4 wavefronts = 4 adjacent strips and walks over
// row points i.  A strip straddles NBS = 2 (N = 21) or 3 column blocks, so a lane's column belongs to block j0 + jsel.
//   resident per lane (registers):   Rj[m] = G_j(b,m)[beta]   of the lane's column (j, b, beta)
//   resident per workgroup (LDS):    x_j in pair-matrix form and as descriptor row for the <= NJ column points it touches
//   staged per row point (LDS, double buffered, shared by the 4 wavefronts; the image [G_i | x_i pair matrix | x_i row]
//   is one contiguous, padded row of ts.TS, fetched one row point ahead with 16-byte loads)
// Per (strip, i):
//   norms |x_i - x_jq|^2 from the descriptor rows (lane-strided, one wave reduction per column point q < NBS)
//   row role (lane = row (a, al)):      v_q[(a,al)] = sum_m (x_i[pair(a,m)] - x_jq[pair(a,m)]) G_i(a,m)[al]  -> LDS vector
//   column role (lane = column):        u = sum_m (x_i[pair(b,m)] - x_j[pair(b,m)]) Rj[m],  g[al] = sum_m G_i(b,m)[al] Rj[m]
//   K[(a,al), col] = 5 b_q v_q[(a,al)] u + c_q G_i(b,a)[al] Rj[a]  - (a == b) c_q g[al]        (G_i(b,b) = 0)
// Same arithmetic as assemble_wave.hip (train.py:199-227 with the 6-nonzeros-per-row Jacobian structure), same
// lower / negated form for the analytic solve (gdml_assemble_A).
// Ordering inside an iteration: loads of point i+1 are issued first, consumed (written to the other LDS buffer) after the
// scalar phase and BEFORE the 63 stores of point i, so the only vmcnt wait of the loop covers the stores of point i-1,
// which had the whole scalar phase to drain.
#include "common.h"

struct StripArgs {
  const double* XF;  // (M,N,N)   x[pair(b,m)], m-major
  const double* GD;  // (M,N,N,3) G(b,m), m-major
  const double* x;   // (M,D) descriptor rows
  const double* TS;  // (M,TOTP) packed per-point image
  int64_t M, n;      // training points, matrix order 3N M
  double sig;
  int i_chunk;       // row points walked by one workgroup
  int lower;         // 1: store -K + lam I, only columns of blocks j <= i
  double lam;
  double* K;
  int64_t ld;
};

typedef double dbl2 __attribute__((ext_vector_type(2)));
constexpr int STRIP_W = 4;  // wavefronts (adjacent strips) per workgroup

template <int N>
struct StripDims {
  static constexpr int N3 = 3 * N, D = N * (N - 1) / 2, NN = N * N;
  static constexpr int NBS = 62 / N3 + 2;                   // column blocks one 64-column strip can touch
  static constexpr int NJ = (64 * STRIP_W - 2) / N3 + 2;    // column blocks one workgroup can touch
  static constexpr int KD = (D + 63) / 64;
  static constexpr int O_XF = N * N3, O_X = N * N3 + NN, TOT = N * N3 + NN + D;
  static constexpr int TOTP = (TOT + 511) / 512 * 512;      // image pitch: 64 STRIP_W threads x 16-byte pieces
  static constexpr int NV = TOTP / (128 * STRIP_W);
};

__global__ void __launch_bounds__(256) strip_pack_kernel(const double* __restrict__ GD, const double* __restrict__ XF,
                                                         const double* __restrict__ x, int64_t M, int o_xf, int o_x,
                                                         int tot, int totp, double* __restrict__ TS) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= M * totp) return;
  const int64_t i = t / totp;
  const int e = (int)(t - i * totp);
  double v = 0.0;
  if (e < o_xf) v = GD[i * o_xf + e];
  else if (e < o_x) v = XF[i * (int64_t)(o_x - o_xf) + (e - o_xf)];
  else if (e < tot) v = x[i * (int64_t)(tot - o_x) + (e - o_x)];
  TS[t] = v;
}

template <int N>
__global__ void __launch_bounds__(64 * STRIP_W, 2) assemble_strip_kernel(StripArgs A) {
  using Dm = StripDims<N>;
  constexpr int N3 = Dm::N3, D = Dm::D, NN = Dm::NN, NBS = Dm::NBS, NJ = Dm::NJ, KD = Dm::KD;
  constexpr int O_XF = Dm::O_XF, O_X = Dm::O_X, TOTP = Dm::TOTP, NV = Dm::NV;
  __shared__ dbl2 S2[2][TOTP / 2];
  __shared__ double XFj[NJ][NN];
  __shared__ double xjs[NJ][D];
  __shared__ double vsh[STRIP_W][NBS][65];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t s = (int64_t)blockIdx.x * STRIP_W + wave;
  const int64_t gc = 64 * s + lane;
  const bool valid = gc < A.n;
  const int64_t jw0 = ((int64_t)blockIdx.x * 64 * STRIP_W) / N3;  // first column point of the workgroup
  int64_t j0 = (64 * s) / N3;
  if (j0 > A.M - 1) j0 = A.M - 1;
  const int64_t jl = valid ? gc / N3 : j0;
  const int cl = valid ? (int)(gc - jl * N3) : 0;
  const int b = cl / 3, beta = cl - 3 * b;
  const int jsel = (int)(jl - j0);
  const int a_r = lane < N3 ? lane / 3 : 0;  // row-role atom (lane = 3 a + al)
  const int q0 = (int)(j0 - jw0);            // wave-uniform index of the strip's first column point in XFj / xjs

  // row chunks are counted from the workgroup's diagonal point in the lower form: every live workgroup walks a full
  // chunk, and the dispatch order (chunk-major) leaves the sparsely populated chunk rows for the end of the launch
  const int64_t i_lo = (A.lower ? jw0 : 0) + (int64_t)blockIdx.y * A.i_chunk;
  const int64_t i_hi = (i_lo + A.i_chunk < A.M) ? i_lo + A.i_chunk : A.M;
  if (i_lo >= i_hi) return;

  const double sig = A.sig, inv_sig = 1.0 / sig;
  const double sqrt5 = 2.23606797749978969641;
  const double base_div = (A.lower ? -1.0 : 1.0) * 5.0 / (3.0 * sig * sig * sig * sig);

  // ---- resident column-point data: registers (own column) and LDS (the workgroup's column points)
  double Rj[N];
  {
    const double* gd = A.GD + jl * (int64_t)N * N3 + cl;
#pragma unroll
    for (int m = 0; m < N; ++m) Rj[m] = gd[m * N3];
    for (int e = tid; e < NJ * NN; e += 64 * STRIP_W) {
      const int q = e / NN;
      const int64_t jq = (jw0 + q < A.M) ? jw0 + q : A.M - 1;
      XFj[q][e - q * NN] = A.XF[jq * (int64_t)NN + (e - q * NN)];
    }
    for (int e = tid; e < NJ * D; e += 64 * STRIP_W) {
      const int q = e / D;
      const int64_t jq = (jw0 + q < A.M) ? jw0 + q : A.M - 1;
      xjs[q][e - q * D] = A.x[jq * (int64_t)D + (e - q * D)];
    }
  }
  dbl2 pf[NV];
  {
    const dbl2* src = reinterpret_cast<const dbl2*>(A.TS + i_lo * (int64_t)TOTP);
#pragma unroll
    for (int t = 0; t < NV; ++t) S2[0][tid + 64 * STRIP_W * t] = src[tid + 64 * STRIP_W * t];
  }
  __syncthreads();
  int cur = 0;
  double* const vq0 = &vsh[wave][0][0];
  const double* const Xjl = &XFj[(q0 + jsel < NJ) ? q0 + jsel : NJ - 1][b];  // x_j[pair(b, m)] of the lane's column: Xjl[m N]
  double* const colbase = A.K + 64 * s;  // wave-uniform

  for (int64_t i = i_lo; i < i_hi; ++i, cur ^= 1) {
    const double* S = reinterpret_cast<const double*>(&S2[cur][0]);
    {  // unconditional (the last iteration re-reads its own point): one vmcnt wait per iteration, no divergent paths
      const int64_t in = (i + 1 < i_hi) ? i + 1 : i;
      const dbl2* src = reinterpret_cast<const dbl2*>(A.TS + in * (int64_t)TOTP);
#pragma unroll
      for (int t = 0; t < NV; ++t) pf[t] = src[tid + 64 * STRIP_W * t];
    }
    // ---- Matern scalars of the NBS column points
    double bp[NBS], cp[NBS];
#pragma unroll
    for (int q = 0; q < NBS; ++q) {
      const int qq = (q0 + q < NJ) ? q0 + q : NJ - 1;
      double ss = 0.0;
#pragma unroll
      for (int t = 0; t < KD; ++t) {
        const int k = lane + 64 * t;
        const double d = (k < D) ? S[O_X + k] - xjs[qq][k] : 0.0;
        ss += d * d;
      }
      const double nrm = sqrt5 * sqrt(wave_sum(ss));
      const double ex = exp(-nrm * inv_sig);
      bp[q] = ex * base_div;
      cp[q] = (sig * sig + sig * nrm) * bp[q];
    }
    // ---- row role: v_q[(a, al)] for this lane's row
    {
      double v[NBS];
#pragma unroll
      for (int q = 0; q < NBS; ++q) v[q] = 0.0;
      if (lane < N3) {
#pragma unroll
        for (int m = 0; m < N; ++m) {
          const double gi = S[m * N3 + lane];
          const double xi = S[O_XF + m * N + a_r];
#pragma unroll
          for (int q = 0; q < NBS; ++q) {
            const int qq = (q0 + q < NJ) ? q0 + q : NJ - 1;
            v[q] += (xi - XFj[qq][m * N + a_r]) * gi;
          }
          if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < NBS; ++q) vq0[q * 65 + lane] = v[q];
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- column role
    double u = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma unroll
    for (int m = 0; m < N; ++m) {
      const double rj = Rj[m];
      u += (S[O_XF + m * N + b] - Xjl[m * N]) * rj;
      g0 += S[m * N3 + 3 * b + 0] * rj;
      g1 += S[m * N3 + 3 * b + 1] * rj;
      g2 += S[m * N3 + 3 * b + 2] * rj;
      if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    // pin the reductions here: LLVM otherwise sinks them into the store branch and keeps every loaded LDS value live
    asm volatile("" : "+v"(u), "+v"(g0), "+v"(g1), "+v"(g2));
    double bpl = bp[0], cpl = cp[0];
#pragma unroll
    for (int q = 1; q < NBS; ++q) {
      bpl = (jsel == q) ? bp[q] : bpl;
      cpl = (jsel == q) ? cp[q] : cpl;
    }
    const double uc = 5.0 * bpl * u;
    const bool diag_blk = A.lower && jl == i;
    const double c0 = -cpl * g0 + ((diag_blk && beta == 0) ? A.lam : 0.0);  // a == b correction (+ lam on the diagonal)
    const double c1 = -cpl * g1 + ((diag_blk && beta == 1) ? A.lam : 0.0);
    const double c2 = -cpl * g2 + ((diag_blk && beta == 2) ? A.lam : 0.0);
    // ---- next point's image into the other buffer (read again only after the barrier below)
#pragma unroll
    for (int t = 0; t < NV; ++t) S2[cur ^ 1][tid + 64 * STRIP_W * t] = pf[t];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): vq0 of this wavefront is written (LDS operations are in order)
    __builtin_amdgcn_wave_barrier();

    if (valid && (!A.lower || jl <= i)) {
      const double* vq = vq0 + jsel * 65;
      double* dst = colbase + (i * N3) * A.ld;
#pragma unroll
      for (int a = 0; a < N; ++a) {
        const double w = cpl * Rj[a];
        const bool dg = a == b;
        const double o0 = vq[3 * a + 0] * uc + (S[a * N3 + 3 * b + 0] * w + (dg ? c0 : 0.0));
        const double o1 = vq[3 * a + 1] * uc + (S[a * N3 + 3 * b + 1] * w + (dg ? c1 : 0.0));
        const double o2 = vq[3 * a + 2] * uc + (S[a * N3 + 3 * b + 2] * w + (dg ? c2 : 0.0));
        dst[lane] = o0;
        dst[A.ld + lane] = o1;
        dst[2 * A.ld + lane] = o2;
        dst += 3 * A.ld;
        if ((a & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // bounds the hoisting of LDS reads (register pressure)
      }
    }
    __syncthreads();
  }
}

int build_dense_tables(gdml_ctx* ctx);

bool assemble_strip_applicable(const gdml_ctx* ctx) {
  const TrainSet& ts = ctx->ts;
  return assemble_wave_applicable(ctx) && ts.N >= 11 && ts.N <= 21 && ctx_opt_i(ctx, "asm.strip", 1) != 0;
}

template <int N>
static int strip_launch_n(gdml_ctx* ctx, StripArgs& A) {
  using Dm = StripDims<N>;
  TrainSet& ts = ctx->ts;
  if (!ts.TS) {
    GDML_TRY(ctx_alloc(ctx, (void**)&ts.TS, ts.M * (int64_t)Dm::TOTP * 8));
    hipLaunchKernelGGL(strip_pack_kernel, dim3((unsigned)ceil_div(ts.M * (int64_t)Dm::TOTP, 256)), dim3(256), 0, ctx->stream,
                       ts.GD, ts.XF, ts.x, ts.M, Dm::O_XF, Dm::O_X, Dm::TOT, Dm::TOTP, ts.TS);
    ctx->launch_counter++;
  }
  A.TS = ts.TS;
  const int64_t n_wg = (A.n + 64 * STRIP_W - 1) / (64 * STRIP_W);
  A.i_chunk = ctx_opt_i(ctx, "asm.i_chunk", 32);
  while (A.i_chunk > 4 && n_wg * ((ts.M + A.i_chunk - 1) / A.i_chunk) < 4096) A.i_chunk >>= 1;
  dim3 grid((unsigned)n_wg, (unsigned)((ts.M + A.i_chunk - 1) / A.i_chunk));
  const int slot = ktime_begin(ctx);
  hipLaunchKernelGGL(assemble_strip_kernel<N>, grid, dim3(64 * STRIP_W), 0, ctx->stream, A);
  const double M = (double)ts.M;
  const double blocks = A.lower ? 0.5 * M * (M + 1.0) : M * M;
  ktime_end(ctx, slot, "assemble", 8.0 * blocks * 9.0 * ts.N * ts.N);
  ctx->launch_counter++;
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}

// All columns of all row points: the full un-negated K (lower = 0) or A = -K + lam I, lower blocks (lower = 1).
int assemble_strip_launch(gdml_ctx* ctx, double sig, double* K, int64_t ld, int lower, double lam) {
  TrainSet& ts = ctx->ts;
  GDML_TRY(build_dense_tables(ctx));
  StripArgs A;
  A.XF = ts.XF; A.GD = ts.GD; A.x = ts.x; A.TS = nullptr; A.M = ts.M; A.n = ts.M * 3 * (int64_t)ts.N; A.sig = sig;
  A.lower = lower; A.lam = lam; A.K = K; A.ld = ld; A.i_chunk = 32;
  switch (ts.N) {
#define SC(v) case v: return strip_launch_n<v>(ctx, A);
    SC(11) SC(12) SC(13) SC(14) SC(15) SC(16) SC(17) SC(18) SC(19) SC(20) SC(21)
#undef SC
    default: return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "assemble_strip: N out of range");
  }
}
