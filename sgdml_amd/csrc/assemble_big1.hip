// Kernel-matrix assembly for LARGE molecules WITHOUT a permutation group (P = 1, 22 <= N <= 256), dense column ranges.
//
// Reference: sgdml/train.py:97-232 (_assemble_kernel_mat_wkr) with the identity permutation, i.e. for every pair of points
//   K[3N i : 3N (i + 1), 3N j : 3N (j + 1)] = J_i^T [ 5 b d (d^T J_j) - (sig^2 + sig n) b J_j ],   d = x_i - x_j,  n = sqrt(5) |d|,
//   b = 5 exp(-n / sig) / (3 sig^4)                                                            (SURVEY.md appendix A).
// J is the sparse descriptor Jacobian (desc.py:422-471: row k = pair (a_k > b_k) has +g_k in the columns of atom b_k and -g_k
// in those of atom a_k), so with u = J_i^T d, w = J_j^T d (3N vectors) the block is
//   K(a alpha, b beta) = 5 b u_a[alpha] w_b[beta] - (sig^2 + sig n) b T_ab[alpha][beta],
//   T_ab = - g_i(ab) g_j(ab)^T  for a != b  (one descriptor entry couples two atoms),   T_aa = sum_m g_i(am) g_j(am)^T.
// A rank-one term plus one 3 x 3 outer product per atom pair: ~30 N^2 flops for 72 N^2 bytes written -- HBM-write bound.
//
// Round 6.  The general kernel (assemble_perm.hip) carries the machinery of permutation groups (images of rows under every
// permutation, per-permutation V / O phases, transposed stores) also at P = 1 and reaches 0.24-0.28 of HBM at N = 100; this
// kernel is the P = 1 case written down directly:
//   one workgroup (256 threads) per row point i and up to BIG1_J = 4 ADJACENT column points (what a workgroup writes into one matrix
//   row is then 4 x 3N x 8 contiguous bytes: the HBM write rate depends on that run length);
//   phase 1  Q adjacent lanes per atom (the largest power of two <= 256 / N) walk its N - 1 partners (tables x, g of the two points
//            straight from L2: 64 bytes per visit, four visits in flight) and accumulate u_a, w_a, T_aa and |d|^2; partial sums meet
//            by lane shuffles, one row of 16 doubles per atom in LDS;
//   phase 2  a wavefront owns a row atom; its 64 lanes are 64 consecutive COLUMNS of the block (column atom = lane / 3), each lane
//            forms its column's three entries (one per row of the atom) from u_a, w_b (LDS) and the pair's Jacobian entries:
//            every store instruction writes 512 contiguous bytes of one matrix row.
// lower != 0: A = -K + lam I for the analytic solve, blocks j <= i only (gdml_assemble_A).
#include "common.h"

namespace {

struct Big1Args {
  const double* x;  // (M, D)
  const double* g;  // (M, D, 3)
  int64_t M;
  int N, D;
  double sig, lam;
  int64_t j0, n_j, col0, i_beg, i_end;
  int lower;
  double* K;
  int64_t ld;
  int Q;  // lanes per atom in phase 1: the largest power of two <= 256 / N
  int dbg;  // timing-only ablation (asm.big1 = 2: no phase 1, 3: no phase 2, 4: phase 2 without its gathers)
};

// pair index of the descriptor entry that couples atoms a != b (np.tril_indices(N, -1) order, desc.py:264)
__device__ __forceinline__ int pair_index(int a, int b) { return a > b ? a * (a - 1) / 2 + b : b * (b - 1) / 2 + a; }

#define BIG1_J 4  // column points per workgroup: 4 x 3N x 8 bytes of every matrix row are written back to back (see below)

__global__ void __launch_bounds__(256) assemble_big1_kernel(Big1Args A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int N = A.N, D = A.D, N3 = 3 * N;
  const int tid = threadIdx.x;
  // this workgroup: row point i, column points j_first .. j_first + nj - 1 (nj <= BIG1_J)
  int64_t i, j_first;
  int nj;
  if (A.lower) {  // row i has ceil((i + 1) / BIG1_J) groups of column points j <= i; S(i) = groups of the rows before i
    const int64_t b = blockIdx.x;
    auto S = [](int64_t r) -> int64_t {  // sum_{r' < r} ceil((r' + 1) / BIG1_J)
      const int64_t f = r / BIG1_J, rem = r - f * BIG1_J;
      return BIG1_J * f * (f + 1) / 2 + rem * (f + 1);
    };
    i = (int64_t)sqrt(2.0 * BIG1_J * (double)b);
    while (i > 0 && S(i) > b) --i;
    while (S(i + 1) <= b) ++i;
    j_first = (b - S(i)) * BIG1_J;
    nj = (int)((i + 1 - j_first < BIG1_J) ? i + 1 - j_first : BIG1_J);
  } else {
    const int64_t gpr = (A.n_j + BIG1_J - 1) / BIG1_J;  // groups per row point
    i = A.i_beg + (int64_t)blockIdx.x / gpr;
    const int64_t v = ((int64_t)blockIdx.x % gpr) * BIG1_J;
    j_first = A.j0 + v;
    nj = (int)((A.n_j - v < BIG1_J) ? A.n_j - v : BIG1_J);
  }
  const double* __restrict__ xi = A.x + i * D;
  const double* __restrict__ gi = A.g + i * (int64_t)D * 3;

  // ---- phase 1, per column point: per-atom sums.  Q (a power of two) adjacent lanes per atom, partner m = q, q + Q, ...; four
  // partners per trip with clamped indices and zero weights instead of branches, the Q partial sums meet by lane shuffles
  const int Q = A.Q;
  double* const U = smem;  // [BIG1_J][N][16]: u (3), w (3), T_aa (9), |d|^2
  if (tid < N * Q && A.dbg != 2) {
    const int a = tid / Q, q = tid - a * Q;
    for (int jj = 0; jj < nj; ++jj) {
      const double* __restrict__ xj = A.x + (j_first + jj) * D;
      const double* __restrict__ gj = A.g + (j_first + jj) * (int64_t)D * 3;
      double acc[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] = 0.0;
      for (int m0 = q; m0 < N; m0 += 4 * Q) {
        int kk[4];
        double wt[4], sg[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + r * Q;
          const bool ok = m < N && m != a;
          const int mc = ok ? m : (a == 0 ? 1 : 0);  // any valid partner: its weight is zero
          kk[r] = pair_index(a, mc);
          wt[r] = ok ? 1.0 : 0.0;
          sg[r] = (a < mc) ? 1.0 : -1.0;  // atom a is the pair's second atom (+g) when a < m
        }
        double dx[4], ga[4][3], gb[4][3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dx[r] = xi[kk[r]] - xj[kk[r]];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            ga[r][c] = gi[3 * kk[r] + c];
            gb[r][c] = gj[3 * kk[r] + c];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double d = dx[r] * wt[r], sd = sg[r] * d;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            acc[c] += ga[r][c] * sd;
            acc[3 + c] += gb[r][c] * sd;
            const double gw = ga[r][c] * wt[r];
            acc[6 + 3 * c] += gw * gb[r][0];
            acc[7 + 3 * c] += gw * gb[r][1];
            acc[8 + 3 * c] += gw * gb[r][2];
          }
          acc[15] += d * d;
        }
      }
      for (int o = 1; o < Q; o <<= 1)
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] += __shfl_xor(acc[c], o, 64);
      if (q == 0)
#pragma unroll
        for (int c = 0; c < 16; ++c) U[(jj * N + a) * 16 + c] = acc[c];
    }
  }
  __syncthreads();
  // scalars of the nj blocks (every wavefront computes them: |d|^2 = half the sum over atoms, each entry was visited from both sides)
  const double sig = A.sig;
  const bool neg = A.lower != 0;  // A = -K + lam I
  double c5v[BIG1_J], csv[BIG1_J];
#pragma unroll
  for (int jj = 0; jj < BIG1_J; ++jj) {
    double nn = 0.0;
    if (jj < nj)
      for (int a = tid & 63; a < N; a += 64) nn += U[(jj * N + a) * 16 + 15];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nn += __shfl_xor(nn, o, 64);
    const double nrm = sqrt(2.5 * nn);  // sqrt(5 * nn / 2)
    const double bb = 5.0 * exp(-nrm / sig) / (3.0 * sig * sig * sig * sig);
    c5v[jj] = neg ? -5.0 * bb : 5.0 * bb;
    csv[jj] = neg ? -(sig * sig + sig * nrm) * bb : (sig * sig + sig * nrm) * bb;
  }

  // ---- phase 2.  A wavefront owns row atoms a = w, w + 4, ...; its 64 lanes are 64 CONSECUTIVE columns (3 b + beta), so every
  // store instruction writes 512 contiguous bytes of one matrix row, and the nj column points of the workgroup are adjacent:
  // nj x 3N x 8 bytes of a row back to back from ONE wavefront.  Measured on the way (profiles/r06_assemble_big1.txt, N = 100, 60 GB
  // lower form, general kernel 30.3 ms): a lane owning a 3 x 3 block and storing nine 8-byte pieces at a 24-byte stride 28.0 ms;
  // coalesced 512-byte pieces of ONE column point per workgroup 29.7 ms, with its Jacobian gathers removed 27.5 ms -- i.e. the store
  // pattern itself (tools/store_bw.hip's 'seg' patterns: 2.4-2.9 TB/s for one segment per row against 4.5 for eight); FOUR adjacent
  // column points 22.8 ms; the four column points dealt to the four wavefronts instead (each 3N x 8 bytes of the same rows) 28.0 ms;
  // one matrix row at a time (gathers and LDS reads repeated per row) 32.8 ms; the column points one after the other inside a row atom
  // (per-point quantities wave uniform, but a partly filled last chunk per point and 512-byte pieces that start anywhere) 26.0 ms.
  const int64_t row0 = A.lower ? i * N3 : (i - A.i_beg) * N3;
  const int64_t colb = A.lower ? j_first * N3 : A.col0 + (j_first - A.j0) * N3;
  double* __restrict__ Kb = A.K + row0 * A.ld + colb;
  const int lane = tid & 63, wv = tid >> 6;
  const int ncols = nj * N3;
  for (int a = wv; a < N && A.dbg != 3; a += 4) {
    {
      double* orow = Kb + (int64_t)(3 * a) * A.ld;
      for (int c0 = 0; c0 < ncols; c0 += 128) {  // two column chunks per trip: their loads are in flight together
        double ga[2][3], gb[2], wb[2], uu[2][3], tt[2][3], ccs[2];
        int colv[2];
        bool dg[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int col = c0 + 64 * r + lane;
          colv[r] = col;
          const int cc = col < ncols ? col : ncols - 1;
          const int jj = cc / N3, cb = cc - jj * N3;
          const int b = cb / 3, be = cb - 3 * b;
          const double* __restrict__ gj = A.g + (j_first + jj) * (int64_t)D * 3;
          const int k = (b == a || A.dbg == 4) ? 0 : pair_index(a, b);  // (dbg 4, timing only: no gathers)
          ga[r][0] = gi[3 * k]; ga[r][1] = gi[3 * k + 1]; ga[r][2] = gi[3 * k + 2];
          gb[r] = gj[3 * k + be];
          const double* Uj = U + (size_t)jj * N * 16;
          double c5 = c5v[0], cs = csv[0];
#pragma unroll
          for (int x = 1; x < BIG1_J; ++x)
            if (jj == x) { c5 = c5v[x]; cs = csv[x]; }
          wb[r] = Uj[b * 16 + 3 + be] * c5;
          uu[r][0] = Uj[a * 16]; uu[r][1] = Uj[a * 16 + 1]; uu[r][2] = Uj[a * 16 + 2];
          tt[r][0] = Uj[a * 16 + 6 + be]; tt[r][1] = Uj[a * 16 + 9 + be]; tt[r][2] = Uj[a * 16 + 12 + be];
          ccs[r] = cs;
          dg[r] = (b == a);
          if (dg[r] && neg && i == j_first + jj) tt[r][be] -= A.lam / cs;  // diagonal of the matrix: r = u w - cs t  ->  + lam
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          if (colv[r] >= ncols) continue;
          double t0, t1, t2;
          if (dg[r]) {
            t0 = tt[r][0]; t1 = tt[r][1]; t2 = tt[r][2];
          } else {
            t0 = -ga[r][0] * gb[r]; t1 = -ga[r][1] * gb[r]; t2 = -ga[r][2] * gb[r];
          }
          const double r0 = uu[r][0] * wb[r] - ccs[r] * t0, r1 = uu[r][1] * wb[r] - ccs[r] * t1, r2 = uu[r][2] * wb[r] - ccs[r] * t2;
          orow[colv[r]] = r0;
          orow[A.ld + colv[r]] = r1;
          orow[2 * A.ld + colv[r]] = r2;
        }
      }
    }
  }
}

}  // namespace

bool assemble_big1_applicable(const gdml_ctx* ctx) {
  const TrainSet& ts = ctx->ts;
  return ctx_opt_i(ctx, "asm.big1", 1) != 0 && ts.P == 1 && ts.N >= 22 && ts.N <= 256;
}

// Column points [j0, j0 + n_j) written at col0 + 3N v, row points [i_beg, i_end); lower: A = -K + lam I, blocks j <= i of
// the whole matrix (j0 = i_beg = 0, n_j = i_end = M).
int assemble_big1_launch(gdml_ctx* ctx, double sig, int64_t j0, int64_t n_j, int64_t col0, double* K, int64_t ld, int64_t i_beg,
                         int64_t i_end, int lower, double lam) {
  TrainSet& ts = ctx->ts;
  if (n_j <= 0 || i_end <= i_beg) return GDML_OK;
  Big1Args A;
  A.x = ts.x; A.g = ts.g; A.M = ts.M; A.N = ts.N; A.D = ts.D; A.sig = sig; A.lam = lam;
  A.j0 = j0; A.n_j = n_j; A.col0 = col0; A.i_beg = i_beg; A.i_end = i_end; A.lower = lower; A.K = K; A.ld = ld;
  const int64_t n_i = i_end - i_beg;
  int64_t blocks = 0;
  if (lower) {
    for (int64_t r = 0; r < n_i; ++r) blocks += (r + 1 + BIG1_J - 1) / BIG1_J;
  } else {
    blocks = n_i * ((n_j + BIG1_J - 1) / BIG1_J);
  }
  if (blocks > 0x7fffffffLL) return GDML_ERR_UNSUPPORTED;
  int Q = 1;
  while (2 * Q * ts.N <= 256 && Q < 32) Q *= 2;
  A.Q = Q;
  A.dbg = ctx_opt_i(ctx, "asm.big1", 1);
  const size_t lds = (size_t)BIG1_J * ts.N * 16 * 8;  // 51 KB at 100 atoms, 131 KB at 256
  if (lds > (size_t)48 * 1024)
    (void)hipFuncSetAttribute((const void*)assemble_big1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int slot = ktime_begin(ctx);
  hipLaunchKernelGGL(assemble_big1_kernel, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, A);
  ktime_end(ctx, slot, "assemble", 8.0 * (lower ? 0.5 * (double)n_i * (double)(n_i + 1) : (double)n_i * (double)n_j) * 9.0 * ts.N * ts.N);
  ctx->launch_counter++;
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}
