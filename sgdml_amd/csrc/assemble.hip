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
// Work decomposition: one workgroup = one column point j x a chunk of row points i.
// x_j, g_j stay in LDS; per i the workgroup stages x_i, g_i in LDS, computes d/u/v/diag
// cooperatively, then every thread owns one output column c and 3*AC consecutive rows, so that a
// wavefront writes contiguous 3N-double row segments of K.
#include "common.h"

struct AsmArgs {
  const double* x;
  const double* g;
  const int32_t* tp;
  const int32_t* perm;
  const int32_t* pinv;
  int64_t M;
  int N, D, P;
  double sig;
  int use_E;
  const int32_t* jlist;   // n_j column points (null: j = j0 + blockIdx.x)
  const int32_t* colmap;  // (n_j, 3N) output column per block column, -1 = skip (null: dense)
  int64_t j0;
  int64_t col0;  // dense mode: output column of point j0
  int64_t i_beg, i_end;  // row points handled by this launch (rows written relative to i_beg)
  int64_t n_j;   // number of column points
  int i_chunk;   // column points walked by one workgroup
  const double* GD;  // dense m-major G table (assemble_wave.hip) or null: G_j is then staged in LDS
  const double* XF;  // dense m-major x table (used together with GD)
  int dbg;  // GDML_ASM_DEBUG ablation bits: 1 skip stores, 2 skip phase A2, 4 skip phase B
  double* K;
  int64_t ld;
  // distributed Cholesky: K = this rank's share of the block-row-cyclic layout (global row block b of cyc_nb rows
  // lives on rank b % cyc_W as local block b / cyc_W); values stored as -K, + cyc_lam on the matrix diagonal; only
  // column points j <= the last row point of the workgroup are walked (lower blocks)
  int cyc_W, cyc_rank, cyc_nb;
  double cyc_lam;
};

__device__ __forceinline__ double block_sum(double v, double* red, int tid, int nwaves) {
  v = wave_sum(v);
  __syncthreads();  // protects red[] reuse
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < nwaves; ++w) s += red[w];
  return s;
}

// Layout of the cooperative phase (all in LDS, dense and index-free so that every inner loop is a
// plain length-N dot product):
//   GJp[b][m][.] = G_j(b, pi m)          (N x N x 3, zero where m = pi^-1 b)   per (j, p)
//   Gi [a][m][.] = G_i(a, m)             (N x N x 3, zero diagonal)            per row point
//   DvF[a][m]    = d_p[pair(a,m)]        (N x N symmetric, zero diagonal)      per (row point, p)
// with G_x(a,m) = (r_m - r_a)/d^3 = +-g_x[pair(a,m)].  Then
//   v[a,.]  = sum_m DvF[a][m] Gi[a][m][.]        u[b,.] = sum_m DvF[pi^-1 b][m] GJp[b][m][.]
//   dg[a]   = sum_m Gi[a][m] (x) GJp[pi a][m]    |d_p|^2 = 1/2 sum DvF^2
// and an output element is  5 b_p v[a,al] u[b,be] - c_p Gi[a][pi^-1 b][al] GJp[b][a][be]
// (dg[a] instead of the product when pi a = b).
//
// Loop nest: a workgroup keeps IB row points i resident and walks over CONSECUTIVE column points j.
// Adjacent 3N-wide row segments of K are therefore written by the same CU a few microseconds
// apart, so the partially covered cache lines at the segment borders merge in that XCD's L2
// (measured: 2.4 TB/s for isolated 504-byte segments vs 4.2 TB/s back-to-back, 6 TB/s full rows).
// The (j,p) tables of the next iteration are prefetched into registers during the current one.
template <int AC, int IB, int MINW>
__global__ void __launch_bounds__(512, MINW) assemble_kernel(AsmArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int N = A.N, D = A.D, P = A.P, N3 = 3 * N, NN = N * N;
  // gjg (large molecules): only the row points' dense G_i lives in LDS; x and G_j come from the
  // global dense tables (XF, GD) and no index tables are needed
  const bool gjg = A.GD != nullptr;
  double* xjp = smem;                                  // D        (absent if gjg)
  double* GJp = xjp + (gjg ? 0 : D);                   // 3 NN     (absent if gjg)
  double* xi = GJp + (gjg ? 0 : 3 * NN);               // IB x D   (absent if gjg)
  double* Gi = xi + (gjg ? 0 : IB * D);                // IB x 3 NN
  double* DvF = Gi + IB * 3 * NN;   // IB x NN
  double* u = DvF + IB * NN;        // IB x 3N
  double* vv = u + IB * N3;         // IB x 3N
  double* dg = vv + IB * N3;        // IB x 9N
  double* red = dg + IB * 9 * N;    // IB x 16
  int* perm_s = reinterpret_cast<int*>(red + IB * 16);  // N
  int* pinv_s = perm_s + N;                             // N
  int* pidx = pinv_s + N;                               // NN  pair index of (a,m)        (absent if gjg)
  int* stab = pidx + NN;   // 3 NN: offset k*3+al in g (bit 31: negate), -1: zero          (absent if gjg)
  int* mtab = stab + 3 * NN;                            // NN: m of the flattened (a,m)   (absent if gjg)

  const int tid = threadIdx.x, T = blockDim.x, nwaves = T >> 6;
  const int64_t i0 = A.i_beg + (int64_t)blockIdx.x * IB;
  const int nb = (A.i_end - i0 < IB) ? (int)(A.i_end - i0) : IB;
  const int64_t jb_beg = (int64_t)blockIdx.y * A.i_chunk;
  int64_t jb_end = (jb_beg + A.i_chunk < A.n_j) ? jb_beg + A.i_chunk : A.n_j;
  if (A.cyc_W > 0) {  // lower blocks only (dense columns: j = jb), and only row points with a row on this rank
    const int64_t i_last = i0 + nb - 1;
    if (jb_end > i_last + 1) jb_end = i_last + 1;
    if (jb_beg >= jb_end) return;
    const int64_t b_first = (i0 * (3 * N)) / A.cyc_nb, b_last = ((i_last + 1) * (3 * N) - 1) / A.cyc_nb;
    bool mine = false;
    for (int64_t bb = b_first; bb <= b_last; ++bb) mine = mine || (bb % A.cyc_W == A.cyc_rank);
    if (!mine) return;
  }

  const int n_chunks = (N + AC - 1) / AC;
  const int item = tid;
  const bool active = item < n_chunks * N3;
  const int chunk = active ? item / N3 : 0;
  const int c = active ? item - chunk * N3 : 0;
  const int b = c / 3, beta = c - 3 * b;

  if (!gjg) {
    for (int e = tid; e < NN; e += T) {
      const int a = e / N, m = e - a * N;
      pidx[e] = (a == m) ? 0 : pair_idx(a, m);
      mtab[e] = m;
    }
    for (int q = tid; q < 3 * NN; q += T) {
      const int am = q / 3, al = q - 3 * am;
      const int a = am / N, m = am - a * N;
      int v = -1;
      if (a != m) v = (pair_idx(a, m) * 3 + al) | ((a < m) ? 0 : (int)0x80000000u);
      stab[q] = v;
    }
  }
  __syncthreads();
  // resident row points: x_i and the dense G_i
  for (int ib = 0; ib < IB; ++ib) {
    const int64_t i = (ib < nb) ? i0 + ib : i0 + nb - 1;
    if (gjg) {
      const double* gd = A.GD + i * (int64_t)NN * 3;
      for (int q = tid; q < 3 * NN; q += T) {
        const int am = q / 3, al = q - 3 * am;
        const int a = am / N, m = am - a * N;
        Gi[ib * 3 * NN + q] = gd[(m * N + a) * 3 + al];  // G_i(a,m)[al] = GD[i][m][a][al]
      }
    } else {
      for (int k = tid; k < D; k += T) xi[ib * D + k] = A.x[i * D + k];
      for (int q = tid; q < 3 * NN; q += T) {
        const int t = stab[q];
        double v = 0.0;
        if (t != -1) {
          v = A.g[i * 3 * D + (t & 0x7fffffff)];
          if (t < 0) v = -v;
        }
        Gi[ib * 3 * NN + q] = v;
      }
    }
  }

  const double sig = A.sig;
  const double inv_sig = 1.0 / sig;
  const double sqrt5 = 2.23606797749978969641;
  const double base_div = 5.0 / (3.0 * sig * sig * sig * sig);
  const double e_fact = 5.0 / (3.0 * sig * sig * sig);

  // (j,p) tables, flattened step t = (jb - jb_beg) * P + p: permuted x_j (D) + dense permuted G_j (3 NN)
  const int per_pt = gjg ? 0 : D + 3 * NN;
  constexpr int PFI = 4;  // register prefetch slots per thread
  const bool use_pf = per_pt <= PFI * T;
  double pf[PFI];
  int pf_perm = 0;
  auto jp_value = [&](int r, int64_t j, int p) -> double {
    if (r < D) return A.x[j * D + A.tp[(size_t)p * D + r]];
    const int q = r - D;
    const int bm = q / 3, be = q - 3 * bm;
    const int m = mtab[bm];
    const int t = stab[(bm - m + A.perm[(size_t)p * N + m]) * 3 + be];
    if (t == -1) return 0.0;
    const double gv = A.g[j * 3 * D + (t & 0x7fffffff)];
    return (t < 0) ? -gv : gv;
  };
  auto jp_store = [&](int r, double v) {
    if (r < D)
      xjp[r] = v;
    else
      GJp[r - D] = v;
  };
  auto jp_fetch = [&](int64_t step) {
    const int64_t jb = jb_beg + step / P;
    const int p = (int)(step - (step / P) * P);
    const int64_t j = A.jlist ? A.jlist[jb] : A.j0 + jb;
#pragma unroll
    for (int s = 0; s < PFI; ++s) {
      const int r = tid + s * T;
      pf[s] = (r < per_pt) ? jp_value(r, j, p) : 0.0;
    }
    if (tid < N)
      pf_perm = A.perm[(size_t)p * N + tid];
    else if (tid < 2 * N)
      pf_perm = A.pinv[(size_t)p * N + tid - N];
  };

  const int64_t n_steps = (jb_end - jb_beg) * P;
  if (use_pf && n_steps > 0) jp_fetch(0);

  double acc[IB][AC][3];
  double erow[IB];
  for (int64_t step = 0; step < n_steps; ++step) {
    const int64_t jb = jb_beg + step / P;
    const int p = (int)(step - (step / P) * P);
    const double* GDj = gjg ? A.GD + (A.jlist ? (int64_t)A.jlist[jb] : A.j0 + jb) * (int64_t)N * N3 : nullptr;
    __syncthreads();  // previous step's readers of the (j,p) tables and DvF/u/v/dg are done
    if (use_pf) {
#pragma unroll
      for (int s = 0; s < PFI; ++s) {
        const int r = tid + s * T;
        if (r < per_pt) jp_store(r, pf[s]);
      }
      if (tid < N)
        perm_s[tid] = pf_perm;
      else if (tid < 2 * N)
        pinv_s[tid - N] = pf_perm;
      if (step + 1 < n_steps) jp_fetch(step + 1);
    } else {
      const int64_t j = A.jlist ? A.jlist[jb] : A.j0 + jb;
      for (int r = tid; r < per_pt; r += T) jp_store(r, jp_value(r, j, p));
      for (int e = tid; e < N; e += T) {
        perm_s[e] = A.perm[(size_t)p * N + e];
        pinv_s[e] = A.pinv[(size_t)p * N + e];
      }
    }
    if (p == 0) {
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) {
        erow[ib] = 0.0;
#pragma unroll
        for (int aa = 0; aa < AC; ++aa) acc[ib][aa][0] = acc[ib][aa][1] = acc[ib][aa][2] = 0.0;
      }
    }
    __syncthreads();  // tables visible
    // ---- A1: dense difference matrices and squared norms for the IB points
    double part[IB];
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) part[ib] = 0.0;
    if (gjg) {
      const double* XFj = A.XF + (A.jlist ? (int64_t)A.jlist[jb] : A.j0 + jb) * (int64_t)NN;
      for (int e = tid; e < NN; e += T) {
        const int a = e / N, m = e - a * N;
        const double xjk = XFj[perm_s[m] * N + perm_s[a]];  // x_j[pair(pi a, pi m)]
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) {
          const int64_t i = (ib < nb) ? i0 + ib : i0 + nb - 1;
          const double dk = (a == m) ? 0.0 : A.XF[i * (int64_t)NN + m * N + a] - xjk;
          DvF[ib * NN + e] = dk;
          part[ib] += dk * dk;
        }
      }
    } else {
      for (int e = tid; e < NN; e += T) {
        const int k = pidx[e];
        const bool diag = mtab[e] * (N + 1) == e;
        const double xjk = xjp[k];
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) {
          const double dk = diag ? 0.0 : xi[ib * D + k] - xjk;
          DvF[ib * NN + e] = dk;
          part[ib] += dk * dk;
        }
      }
    }
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) part[ib] = wave_sum(part[ib]);
    if ((tid & 63) == 0) {
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) red[ib * 16 + (tid >> 6)] = part[ib];
    }
    __syncthreads();  // DvF[] and red[] visible
    // ---- A2: u_p (3N), v_p (3N), diagonal blocks (9N) for each of the IB points
    for (int t2 = tid; t2 < ((A.dbg & 2) ? 0 : IB * 15 * N); t2 += T) {
      const int ib = t2 / (15 * N), t = t2 - ib * 15 * N;
      const double* Dv = DvF + ib * NN;
      const double* Gib = Gi + ib * 3 * NN;
      if (t < N3) {  // u[b,be] = sum_m DvF[pinv b][m] GJp[b][m][be]
        const int bb = t / 3, be = t - 3 * bb;
        const double* dr = Dv + pinv_s[bb] * N;
        double s = 0.0;
        if (gjg) {
          const double* gr = GDj + 3 * bb + be;  // G_j(b, pi m)[be] = GD[j][pi m][b][be]
#pragma unroll 4
          for (int m = 0; m < N; ++m) s += dr[m] * gr[perm_s[m] * N3];
        } else {
          const double* gr = GJp + bb * N3 + be;
#pragma unroll 7
          for (int m = 0; m < N; ++m) s += dr[m] * gr[3 * m];
        }
        u[ib * N3 + t] = s;
      } else if (t < 2 * N3) {  // v[a,al] = sum_m DvF[a][m] Gi[a][m][al]
        const int tt = t - N3;
        const int a = tt / 3, al = tt - 3 * a;
        const double* dr = Dv + a * N;
        const double* gr = Gib + a * N3 + al;
        double s = 0.0;
#pragma unroll 7
        for (int m = 0; m < N; ++m) s += dr[m] * gr[3 * m];
        vv[ib * N3 + tt] = s;
      } else {  // dg[a][al][be] = sum_m Gi[a][m][al] GJp[perm a][m][be]
        const int tt = t - 2 * N3;
        const int a = tt / 9, r = tt - 9 * a;
        const int al = r / 3, be = r - 3 * al;
        const double* g1 = Gib + a * N3 + al;
        double s = 0.0;
        if (gjg) {
          const double* g2 = GDj + 3 * perm_s[a] + be;  // G_j(pi a, pi m)[be]
#pragma unroll 4
          for (int m = 0; m < N; ++m) s += g1[3 * m] * g2[perm_s[m] * N3];
        } else {
          const double* g2 = GJp + perm_s[a] * N3 + be;
#pragma unroll 7
          for (int m = 0; m < N; ++m) s += g1[3 * m] * g2[3 * m];
        }
        dg[ib * 9 * N + tt] = s;
      }
    }
    __syncthreads();
    // ---- B: accumulate this thread's outputs
    if (active && !(A.dbg & 4)) {
      const int ap = pinv_s[b];
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) {
        double nrm2 = 0.0;
        for (int w = 0; w < nwaves; ++w) nrm2 += red[ib * 16 + w];
        const double nrm = sqrt5 * sqrt(0.5 * nrm2);
        const double ex = exp(-nrm * inv_sig);
        const double bp = ex * base_div;
        const double cp = (sig * sig + sig * nrm) * bp;
        const double uc_raw = u[ib * N3 + c];
        const double uc = 5.0 * bp * uc_raw;
        const double* Gib = Gi + ib * 3 * NN;
        const double* vb = vv + ib * N3;
        const double* dgb = dg + ib * 9 * N;
#pragma unroll
        for (int aa = 0; aa < AC; ++aa) {
          const int a = chunk * AC + aa;
          if (a < N) {
            double t0, t1, t2;
            if (a != ap) {
              const double w = -cp * (gjg ? GDj[perm_s[a] * N3 + 3 * b + beta] : GJp[(b * N + a) * 3 + beta]);
              const double* gia = Gib + (a * N + ap) * 3;
              t0 = gia[0] * w;
              t1 = gia[1] * w;
              t2 = gia[2] * w;
            } else {
              t0 = -cp * dgb[a * 9 + 0 + beta];
              t1 = -cp * dgb[a * 9 + 3 + beta];
              t2 = -cp * dgb[a * 9 + 6 + beta];
            }
            acc[ib][aa][0] += vb[3 * a + 0] * uc + t0;
            acc[ib][aa][1] += vb[3 * a + 1] * uc + t1;
            acc[ib][aa][2] += vb[3 * a + 2] * uc + t2;
          }
        }
        if (A.use_E && chunk == 0) erow[ib] -= e_fact * (nrm + sig) * ex * uc_raw;  // train.py:235-248
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- after the last permutation: write the IB blocks (rows 3N i + 3a + alpha, column outcol)
    if (p == P - 1 && active) {
      const int64_t outcol = A.colmap ? (int64_t)A.colmap[jb * N3 + c] : A.col0 + jb * N3 + c;
      if (outcol >= 0 && (!(A.dbg & 1) || acc[0][0][0] == 1.2345e-300)) {
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) {
          if (ib < nb) {
            const int64_t i = i0 + ib;
#pragma unroll
            for (int aa = 0; aa < AC; ++aa) {
              const int a = chunk * AC + aa;
              if (a < N) {
                if (A.cyc_W > 0) {
#pragma unroll
                  for (int al = 0; al < 3; ++al) {
                    const int64_t grow = i * N3 + 3 * a + al, gb = grow / A.cyc_nb;
                    if (gb % A.cyc_W == A.cyc_rank)
                      A.K[((gb / A.cyc_W) * A.cyc_nb + grow % A.cyc_nb) * A.ld + outcol] =
                          -acc[ib][aa][al] + (grow == outcol ? A.cyc_lam : 0.0);
                  }
                } else {
                  double* dst = A.K + ((int64_t)(i - A.i_beg) * N3 + 3 * a) * A.ld + outcol;
                  dst[0] = acc[ib][aa][0];
                  dst[A.ld] = acc[ib][aa][1];
                  dst[2 * A.ld] = acc[ib][aa][2];
                }
              }
            }
            if (A.use_E && chunk == 0) A.K[(A.M * N3 + i) * A.ld + outcol] = erow[ib];
          }
        }
      }
    }
  }
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
};

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
  if (tid < N3) A.K[(i * N3 + tid) * A.ld + col] = out0;
  if (tid + T < N3) A.K[(i * N3 + tid + T) * A.ld + col] = out1;
  if (tid == 0) A.K[(A.M * N3 + i) * A.ld + col] = ee;
}

template <int AC, int IB, int MINW>
static void launch_asm(gdml_ctx* ctx, const AsmArgs& A, dim3 grid, int T, size_t lds) {
  hipFuncSetAttribute((const void*)assemble_kernel<AC, IB, MINW>,
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((assemble_kernel<AC, IB, MINW>), grid, dim3(T), lds, ctx->stream, A);
}

static size_t asm_lds_bytes(int N, int D, int IB, bool gjg = false) {
  const size_t NN = (size_t)N * N;
  size_t dbl = (gjg ? 0 : D + 3 * NN) + (size_t)IB * ((gjg ? 0 : D) + 3 * NN + NN + 15 * N + 16);
  size_t ints = 2 * N + (gjg ? 0 : NN + 3 * NN + NN);
  return dbl * 8 + ints * 4 + 16;
}

static int assemble_dispatch(gdml_ctx* ctx, AsmArgs& A, int64_t n_j) {
  const int N = A.N, D = A.D;
  static const int acs[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
  // threads per workgroup: measured optimum by molecule size (profiles/r02_assemble_lds_sweep.txt)
  const int t_target = ctx_opt_i(ctx, "asm.threads", N >= 22 ? 512 : (N >= 12 && N <= 16 ? 128 : 256));
  int AC = 32;
  for (int v : acs) {
    int items = ((N + v - 1) / v) * 3 * N;
    if (items <= t_target) {
      AC = v;
      break;
    }
  }
  int items = ((N + AC - 1) / AC) * 3 * N;
  if (items > 512)
    return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "assembly kernel supports up to %d atoms", 64);
  int T = ((items + 63) / 64) * 64;
  // batch of row points per iteration (option asm.ib caps it): 4 if it fits in LDS, else 2, else 1
  int IB = 1;
  if (AC <= 6) {
    const int want = ctx_opt_i(ctx, "asm.ib", 2);
    for (int cand : {4, 2}) {
      if (cand <= want && asm_lds_bytes(N, D, cand) <= 150 * 1024) {
        IB = cand;
        break;
      }
    }
  }
  const int minw = ctx_opt_i(ctx, "asm.minw", 2);
  size_t lds = asm_lds_bytes(N, D, IB);
  A.GD = nullptr;
  A.XF = nullptr;
  if (lds > 160 * 1024 || ctx_opt_i(ctx, "asm.gj_global", 0)) {
    // large molecule: keep only the row point's dense table in LDS, read G_j from the global table
    extern int build_dense_tables(gdml_ctx * ctx);
    GDML_TRY(build_dense_tables(ctx));
    A.GD = ctx->ts.GD;
    A.XF = ctx->ts.XF;
    lds = asm_lds_bytes(N, D, IB, true);
  }
  if (lds > 160 * 1024)
    return gdml_fail(ctx, GDML_ERR_UNSUPPORTED,
                     "assembly kernel needs %zu bytes of LDS for N=%d (limit 160 KiB, N <= 66)", lds, N);
  // column points per workgroup: long enough to amortise the resident row points, short enough
  // that the grid has >= ~8 workgroups per CU
  const int64_t n_ib = (A.i_end - A.i_beg + IB - 1) / IB;
  int j_chunk = ctx_opt_i(ctx, "asm.j_chunk", 64);
  while (j_chunk > 4 && n_ib * ((n_j + j_chunk - 1) / j_chunk) < 4096) j_chunk >>= 1;
  A.i_chunk = j_chunk;
  A.n_j = n_j;
  dim3 grid((unsigned)n_ib, (unsigned)((n_j + j_chunk - 1) / j_chunk));
  const int slot = ktime_begin(ctx);
#define LAUNCH(ac, ib)                                                  \
  do {                                                                  \
    if (minw >= 4) launch_asm<ac, ib, 4>(ctx, A, grid, T, lds);         \
    else launch_asm<ac, ib, 2>(ctx, A, grid, T, lds);                   \
  } while (0)
#define CASE(v)                          \
  case v:                                \
    if (IB == 4) LAUNCH(v, 4);           \
    else if (IB == 2) LAUNCH(v, 2);      \
    else LAUNCH(v, 1);                   \
    break;
#define CASE1(v)                                 \
  case v:                                        \
    launch_asm<v, 1, 2>(ctx, A, grid, T, lds);   \
    break;
  switch (AC) {
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(6) CASE1(8) CASE1(12) CASE1(16) CASE1(24) CASE1(32)
  }
#undef CASE
#undef CASE1
#undef LAUNCH
  // algorithmic bytes: every requested element of K written once
  ktime_end(ctx, slot, "assemble", 8.0 * (double)(A.i_end - A.i_beg) * 3.0 * N * (double)n_j * 3.0 * N);
  ctx->launch_counter++;
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}

// Rows of A = -K + lam I owned by this rank in the block-row-cyclic layout of the distributed Cholesky (any P,
// N <= 64): full column range, lower blocks.  K: the rank's local matrix.
int assemble_cyclic_launch(gdml_ctx* ctx, double sig, double lam, double* K, int64_t ld, int cyc_W, int cyc_rank,
                           int cyc_nb) {
  TrainSet& ts = ctx->ts;
  if (assemble_wave_applicable(ctx))
    return assemble_wave_launch(ctx, sig, 0, nullptr, nullptr, 0, ts.M, K, ld, 0, ts.M, 1, lam, cyc_W, cyc_rank, cyc_nb);
  if (ctx_opt_i(ctx, "asm.perm", 1))
    return assemble_perm_launch(ctx, sig, 0, nullptr, nullptr, 0, ts.M, 0, K, ld, 0, ts.M, 1, lam, cyc_W > 1 ? cyc_W : 0,
                                cyc_rank, cyc_nb);
  AsmArgs A;
  A.x = ts.x; A.g = ts.g; A.tp = ts.tp; A.perm = ts.perm; A.pinv = ts.pinv;
  A.M = ts.M; A.N = ts.N; A.D = ts.D; A.P = ts.P; A.sig = sig; A.use_E = 0;
  A.jlist = nullptr; A.colmap = nullptr; A.j0 = 0; A.col0 = 0; A.i_chunk = 8;
  A.dbg = 0;
  A.K = K; A.ld = ld;
  A.i_beg = 0; A.i_end = ts.M;
  A.cyc_W = cyc_W; A.cyc_rank = cyc_rank; A.cyc_nb = cyc_nb; A.cyc_lam = lam;
  return assemble_dispatch(ctx, A, ts.M);
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
    if (use_E_cstr)
      return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "sharded assembly does not support energy constraints");
    shard_points(ctx, M, &i_beg, &i_end, nullptr);
  }
  const int64_t n_rows_store = sharded ? (i_end - i_beg) * N3 : n_rows;

  // ---- (re)allocate the device matrix
  const int64_t tot_rows = n_rows_store + alloc_extra_rows;
  const int64_t ld = (n_cols + 15) / 16 * 16;  // rows start on 128-byte boundaries
  // the buffer is reused when the size is unchanged (hyper-parameter sweeps, benchmark loops)
  if (ctx->K && ctx->K_bytes != tot_rows * ld * 8) {
    GDML_TRY(ctx_free(ctx, ctx->K));
    ctx->K = nullptr;
    ctx->precon = nullptr;
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
  // fused form for the analytic solve: the register-resident kernel (P = 1, N <= 21) or, for permutation groups and
  // larger molecules, the LDS kernel in its lower / negated mode (= its row-cyclic mode with one rank)
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
    AsmArgs A;
    A.x = ts.x; A.g = ts.g; A.tp = ts.tp; A.perm = ts.perm; A.pinv = ts.pinv;
    A.M = M; A.N = N; A.D = ts.D; A.P = ts.P; A.sig = sig; A.use_E = use_E_cstr;
    A.jlist = d_jlist; A.colmap = d_colmap; A.j0 = j0; A.col0 = 0; A.i_chunk = 8;
    A.dbg = ctx_opt_i(ctx, "asm.debug", 0);
    A.K = ctx->K; A.ld = ld;
    A.cyc_W = 0; A.cyc_rank = 0; A.cyc_nb = 0; A.cyc_lam = 0.0;
    A.i_beg = i_beg; A.i_end = i_end;
    if (i_end <= i_beg)
      rc = GDML_OK;
    else if (lower_A && !assemble_wave_applicable(ctx))
      rc = assemble_cyclic_launch(ctx, sig, lam, ctx->K, ld, 1, 0, 512);
    else if (dense && !use_E_cstr && i_beg == 0 && i_end == M && assemble_strip_applicable(ctx))
      rc = assemble_strip_launch(ctx, sig, ctx->K, ld, lower_A ? 1 : 0, lam);
    else if (assemble_wave_applicable(ctx))
      rc = assemble_wave_launch(ctx, sig, use_E_cstr, d_jlist, d_colmap, j0, n_j, ctx->K, ld, i_beg, i_end,
                                lower_A ? 1 : 0, lam);
    else if (ctx_opt_i(ctx, "asm.perm", 1))
      rc = assemble_perm_launch(ctx, sig, use_E_cstr, d_jlist, d_colmap, j0, n_j, 0, ctx->K, ld, i_beg, i_end, 0, 0.0, 0, 0, 0);
    else
      rc = assemble_dispatch(ctx, A, n_j);
  }
  if (rc == GDML_OK && !e_pts.empty()) {
    if (N3 > 512) rc = gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "E-constraint columns need 3N <= 512");
    else {
      EColArgs E;
      E.x = ts.x; E.g = ts.g; E.tp = ts.tp; E.perm = ts.perm; E.pinv = ts.pinv;
      E.M = M; E.N = N; E.D = ts.D; E.P = ts.P; E.sig = sig;
      E.jj_list = d_ep; E.out_cols = d_ec; E.K = ctx->K; E.ld = ld;
      size_t lds = (size_t)(2 * ts.D + 32) * 8;
      hipFuncSetAttribute((const void*)ecol_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds);
      for (int64_t i0 = 0; i0 < M; i0 += 65535) {  // grid.y limit
        E.i0 = i0;
        const int64_t ny = (M - i0 < 65535) ? M - i0 : 65535;
        hipLaunchKernelGGL(ecol_kernel, dim3((unsigned)e_pts.size(), (unsigned)ny), dim3(256), lds, ctx->stream, E);
        ctx->launch_counter++;
      }
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) rc = gdml_fail(ctx, GDML_ERR_HIP, "ecol launch: %s", hipGetErrorString(e));
    }
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
