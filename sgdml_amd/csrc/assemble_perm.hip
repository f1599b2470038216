// General kernel-matrix assembly: any permutation group, any molecule size (replaces the LDS kernel of assemble.hip
// wherever the register-resident P = 1 kernels do not apply).
//
// Reference: sgdml/train.py:97-302 (_assemble_kernel_mat_wkr), torchtools.py:110-392.  Math as in assemble.hip:
//   K_ij = sum_p [ 5 b_p v_p u_p^T - c_p J_i^T J_j^p ],   d_p = x_i - P_p x_j
//   v_p[a]  = sum_m d_p[pair(a,m)] G_i(a,m)                      (row atom a)
//   u_p[b]  = sum_m' (x_i[pair(pi^-1 b, pi^-1 m')] - x_j[pair(b,m')]) G_j(b,m')   (column atom b)
//   (J_i^T J_j^p)[(a,.),(b,.)] = G_i(a, pi^-1 b) (x) G_j(b, pi a)            pi^-1 b != a
//                              = sum_m' G_i(a, pi^-1 m') (x) G_j(b, m')      pi a = b      ("dg")
// with the dense m-major tables of assemble_wave.hip: XF[x][m][b] = x[pair(b,m)], GD[x][m][b] = G_x(b,m) = (r_m-r_b)/d^3.
//
// Mapping (the old kernel spent its time in barrier-separated cooperative phases per (j, p): 2.4 % of the fp64 rate):
//   * a workgroup owns a STRIP of 64 consecutive column atoms = 192 consecutive output columns (1536 bytes, 128-byte
//     aligned) -- it straddles NQ column points -- and walks over row points i;
//   * every wavefront has lane = column atom (j, b) with its 3 columns, so that G_i(a, pi^-1 b) and v_p[a] are read
//     once for 3 x 3 outputs (the LDS read rate, not the fp64 rate, bounds a one-column-per-lane mapping);
//   * the W wavefronts split the ROW atoms (a = k W + w): 9 NA accumulators per lane, summed over p in registers;
//   * per row point and group of PG permutations two phases, one barrier each:
//       V  (tasks spread over the wavefronts)  V12(p, pass): lane = (column point q, row atom a): v_p, partial |d_p|^2
//                                              V3(p):        lane = column atom: u_p (3), dg_p (3 x 3)      -> LDS
//       O  every wavefront, its own row atoms:  acc[a] += beta v_p[a] (x) u_p + (-c_p) G_i(a,a') (x) G_j(b, pi a) + [a = a'] (-c_p) dg_p
//   * finished rows are transposed through a per-wavefront LDS row so that every store instruction writes 64 consecutive
//     doubles (full cache lines; the pattern of assemble_strip.hip).
// Row-point image (GD_i, XF_i) and the strip's G_j table live in LDS when they fit (IMG / GJS), otherwise they are read
// through L1/L2 from the dense tables (any N up to GDML_MAX_ATOMS; more than W NA row atoms take several rounds).
#include "common.h"

struct PermArgs {
  const double* XF;   // (M,N,N)
  const double* GD;   // (M,N,N,3)
  const int32_t* perm;  // (P,N)  pi_p
  const int32_t* pinv;  // (P,N)  pi_p^-1
  int64_t M;
  int N, P;
  double sig;
  int use_E;             // also write the energy-constraint row K[3N M + i, .]  (train.py:235-248)
  const int32_t* jlist;  // virtual column point -> training point (null: j0 + v)
  const int32_t* colmap; // (n_j, 3N) output column or -1 (null: col0 + 3N v + c)
  int64_t j0, n_j, col0, n_cols;
  int64_t i_beg, i_end;  // row points of this launch; rows are written relative to i_beg
  int i_chunk;
  int lower;             // store -K + lam I, only blocks j <= i (dense full column range)
  double lam;
  int cyc_W, cyc_rank, cyc_nb;  // block-row-cyclic local layout of the distributed Cholesky (implies lower)
  int fast_store;        // dense columns, plain row layout: transposed full-line stores
  int NQ, PG, npass;
  int o_SG, o_GjS, o_vs, o_part, o_scal, o_ud, o_tr, o_perm;  // LDS offsets in doubles
  double* K;
  int64_t ld;
};

constexpr int PERM_W = 8;  // wavefronts per workgroup

// NA: row atoms per wavefront and round (3: N <= 24, two workgroups per CU; 6: N <= 48 in one round)
template <int NA, bool IMG, bool GJS>
__global__ void __launch_bounds__(64 * PERM_W, 2) assemble_perm_kernel(PermArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int W = PERM_W;
  constexpr int T = 64 * W;
  const int N = A.N, N3 = 3 * N, NN = N * N, P = A.P, NQ = A.NQ, PG = A.PG, npass = A.npass;
  // image of the row point (IMG): [G | X] = [m][b][al] | [m][b], two buffers of 4 N^2 doubles
  double* const IM0 = smem + A.o_SG;
  double* const GjS = smem + A.o_GjS;    // [m][lane][be] G_j(b,m) of the lane's column atom (GJS)
  double* const vs = smem + A.o_vs;      // [pl][q][a][al]
  double* const part = smem + A.o_part;  // [pl][q][a]
  double* const scal = smem + A.o_scal;  // [w][pl][q][3]   beta, -c, E-row coefficient
  double* const ud = smem + A.o_ud;      // [pl][12][lane]  u (3), dg (3 x 3)
  double* const tr = smem + A.o_tr;      // [w][192]        transposed output row (aliases the V-phase results)
  int* const permS = reinterpret_cast<int*>(smem + A.o_perm);
  int* const pinvS = permS + P * N;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t s = blockIdx.x;
  const int64_t n_ca = A.n_j * N;  // column atoms of this launch
  const int64_t gav = 64 * s + lane;
  const bool cvalid = gav < n_ca;
  const int jv0 = (int)((64 * s) / N);  // first (virtual) column point of the strip
  const int jv = cvalid ? (int)(gav / N) : (int)A.n_j - 1;
  const int b = cvalid ? (int)(gav - (int64_t)jv * N) : 0;
  const int q = jv - jv0;
  const int jpt = A.jlist ? A.jlist[jv] : (int)A.j0 + jv;
  const bool lower = A.lower != 0;

  const int64_t i_lo = (lower ? (int64_t)jv0 : A.i_beg) + (int64_t)blockIdx.y * A.i_chunk;
  const int64_t i_top = lower ? A.M : A.i_end;
  const int64_t i_hi = (i_lo + A.i_chunk < i_top) ? i_lo + A.i_chunk : i_top;
  if (i_lo >= i_hi) return;

  const double sig = A.sig, inv_sig = 1.0 / sig;
  const double sqrt5 = 2.23606797749978969641;
  const double base_div = 5.0 / (3.0 * sig * sig * sig * sig);
  const double e_fact = 5.0 / (3.0 * sig * sig * sig);

  // ---- resident tables: permutations, the strip's column-atom data
  for (int e = tid; e < P * N; e += T) {
    permS[e] = A.perm[e];
    pinvS[e] = A.pinv[e];
  }
  if (GJS) {
    const double* gd = A.GD + ((int64_t)jpt * NN + b) * 3;
    for (int m = w; m < N; m += W) {
      GjS[(m * 64 + lane) * 3 + 0] = gd[m * N3 + 0];
      GjS[(m * 64 + lane) * 3 + 1] = gd[m * N3 + 1];
      GjS[(m * 64 + lane) * 3 + 2] = gd[m * N3 + 2];
    }
  }
  if (IMG) {
    const double* gi = A.GD + i_lo * (int64_t)NN * 3;
    const double* xi = A.XF + i_lo * (int64_t)NN;
    for (int e = tid; e < 3 * NN; e += T) IM0[e] = gi[e];
    for (int e = tid; e < NN; e += T) IM0[3 * NN + e] = xi[e];
  }
  int cur = 0;

  // output columns of the lane (general store path) and of the transposed rows (fast path)
  int outcol[3];
#pragma unroll
  for (int be = 0; be < 3; ++be) {
    int oc = -1;
    if (cvalid) oc = A.colmap ? A.colmap[(int64_t)jv * N3 + 3 * b + be] : (int)A.col0 + jv * N3 + 3 * b + be;
    outcol[be] = oc;
  }
  int tcol[3], tpt[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int64_t c = 192 * s + 64 * t + lane;  // relative to col0
    tcol[t] = (c < n_ca * 3) ? (int)(A.col0 + c) : -1;
    tpt[t] = (int)(c / N3);
  }

  constexpr int NPF = IMG ? (4 * (W * NA) * (W * NA) + T - 1) / T : 1;  // 4 N^2 doubles of the next image over T threads
  const int n_rounds = (N + W * NA - 1) / (W * NA);
  const int n_groups = (P + PG - 1) / PG;

  for (int64_t i = i_lo; i < i_hi; ++i) {
    const int64_t row0 = i * N3;                 // first row of the point in the full matrix
    // transposed-row stores: which of the lane's three columns are written for this row point, and where the matrix
    // diagonal crosses them (lower form: + lam)
    bool tok[3];
    int dcol[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      tok[t] = tcol[t] >= 0 && (!lower || tpt[t] <= i);
      dcol[t] = lower ? (int)((int64_t)tcol[t] - row0) : -1;
    }
    const double lamv = lower ? A.lam : 0.0;
    const int64_t lrow0 = row0 - A.i_beg * N3;   // ... in the stored matrix (plain layout)
    int coff0 = 0;
    int64_t clrow0 = 0, clrow1 = 0;
    bool cmine0 = true, cmine1 = true;
    if (A.cyc_W > 0) {  // block-row-cyclic layout: one division per row point, none per row
      const int64_t cb0 = row0 / A.cyc_nb;
      coff0 = (int)(row0 - cb0 * A.cyc_nb);
      cmine0 = (cb0 % A.cyc_W) == A.cyc_rank;
      cmine1 = ((cb0 + 1) % A.cyc_W) == A.cyc_rank;
      clrow0 = (cb0 / A.cyc_W) * A.cyc_nb;
      clrow1 = ((cb0 + 1) / A.cyc_W) * A.cyc_nb;
      const bool spans = coff0 + N3 > A.cyc_nb;
      if (!(cmine0 || (spans && cmine1))) {  // no row of this point on this rank
        if (IMG && i + 1 < i_hi) {  // keep the image in step
          __syncthreads();
          const double* gi = A.GD + (i + 1) * (int64_t)NN * 3;
          const double* xi = A.XF + (i + 1) * (int64_t)NN;
          double* im = IM0 + cur * 4 * NN;
          for (int e = tid; e < 3 * NN; e += T) im[e] = gi[e];
          for (int e = tid; e < NN; e += T) im[3 * NN + e] = xi[e];
        }
        continue;
      }
    }
    __syncthreads();  // image of point i (and the resident tables) visible; the previous point's LDS readers are done
    const double* const SG = IM0 + cur * 4 * NN;
    const double* const SX = SG + 3 * NN;
    double* const IMn = IM0 + (cur ^ 1) * 4 * NN;  // next point's image, filled during the first V phase
    const double* const GDi = A.GD + i * (int64_t)NN * 3;
    const double* const XFi = A.XF + i * (int64_t)NN;

    for (int r = 0; r < n_rounds; ++r) {
      double acc[NA][3][3];
      double erow[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < NA; ++k)
#pragma unroll
        for (int al = 0; al < 3; ++al) acc[k][al][0] = acc[k][al][1] = acc[k][al][2] = 0.0;

      for (int g0 = 0; g0 < P; g0 += PG) {
        const int npg = (P - g0 < PG) ? P - g0 : PG;
        if (r == 0 || n_groups > 1) {
          // ================= phase V
          // the next row point's image travels global -> registers -> other LDS buffer: underneath the first V phase when
          // the registers are there (NA = 3), in batches ahead of it otherwise
          constexpr bool PF_REG = IMG && NA == 3;
          double pf[PF_REG ? NPF : 1];
          const bool do_pf = IMG && g0 == 0;
          if (do_pf) {
            const int64_t in = (i + 1 < i_hi) ? i + 1 : i;
            const double* gi = A.GD + in * (int64_t)NN * 3;
            const double* xi = A.XF + in * (int64_t)NN;
            if (PF_REG) {
#pragma unroll
              for (int t = 0; t < NPF; ++t) {
                const int e = tid + T * t;
                pf[t] = (e < 3 * NN) ? gi[e] : ((e < 4 * NN) ? xi[e - 3 * NN] : 0.0);
              }
            } else {
#pragma unroll 6
              for (int e = tid; e < 4 * NN; e += T) IMn[e] = (e < 3 * NN) ? gi[e] : xi[e - 3 * NN];
            }
          }
          const int ntask = npg * (npass + 1);
          for (int t = w; t < ntask; t += W) {
            const int pl = t / (npass + 1), kind = t - pl * (npass + 1);
            const int p = g0 + pl;
            const int* pm_tab = permS + p * N;
            const int* pi_tab = pinvS + p * N;
            if (kind < npass) {
              // ---- V12: lane = (column point qq, row atom a)
              const int idx = kind * 64 + lane;
              const int qq = idx / N;
              const int a = idx - qq * N;
              const int jvq = jv0 + qq;
              if (qq < NQ && jvq < A.n_j) {
                const int64_t jq = A.jlist ? (int64_t)A.jlist[jvq] : A.j0 + jvq;
                const double* xfj = A.XF + jq * NN + pm_tab[a];
                double v0 = 0.0, v1 = 0.0, v2 = 0.0, nn = 0.0;
#pragma unroll 4
                for (int m = 0; m < N; ++m) {
                  const int pm = pm_tab[m];
                  const double xi = IMG ? SX[m * N + a] : XFi[m * N + a];
                  const double xj = xfj[pm * N];
                  const double d = xi - xj;
                  const double* g = IMG ? SG + (m * N + a) * 3 : GDi + (m * N + a) * 3;
                  nn += d * d;
                  v0 += d * g[0];
                  v1 += d * g[1];
                  v2 += d * g[2];
                }
                double* dst = vs + ((pl * NQ + qq) * N + a) * 3;
                dst[0] = v0;
                dst[1] = v1;
                dst[2] = v2;
                part[(pl * NQ + qq) * N + a] = nn;
              }
            } else {
              // ---- V3: lane = column atom (j, b)
              const int ap = pi_tab[b];
              const double* xfj = A.XF + (int64_t)jpt * NN + b;
              const double* gdj = A.GD + ((int64_t)jpt * NN + b) * 3;
              double u0 = 0.0, u1 = 0.0, u2 = 0.0;
              double d00 = 0.0, d01 = 0.0, d02 = 0.0, d10 = 0.0, d11 = 0.0, d12 = 0.0, d20 = 0.0, d21 = 0.0, d22 = 0.0;
#pragma unroll 2
              for (int mp = 0; mp < N; ++mp) {
                const int mi = pi_tab[mp];
                const double xi = IMG ? SX[mi * N + ap] : XFi[mi * N + ap];
                const double xj = xfj[mp * N];
                const double d = xi - xj;
                const double* rj = GJS ? GjS + (mp * 64 + lane) * 3 : gdj + mp * N3;
                const double* gi = IMG ? SG + (mi * N + ap) * 3 : GDi + (mi * N + ap) * 3;
                const double r0 = rj[0], r1 = rj[1], r2 = rj[2];
                const double g0v = gi[0], g1v = gi[1], g2v = gi[2];
                u0 += d * r0; u1 += d * r1; u2 += d * r2;
                d00 += g0v * r0; d01 += g0v * r1; d02 += g0v * r2;
                d10 += g1v * r0; d11 += g1v * r1; d12 += g1v * r2;
                d20 += g2v * r0; d21 += g2v * r1; d22 += g2v * r2;
              }
              double* dst = ud + pl * 12 * 64 + lane;
              dst[0 * 64] = u0; dst[1 * 64] = u1; dst[2 * 64] = u2;
              dst[3 * 64] = d00; dst[4 * 64] = d01; dst[5 * 64] = d02;
              dst[6 * 64] = d10; dst[7 * 64] = d11; dst[8 * 64] = d12;
              dst[9 * 64] = d20; dst[10 * 64] = d21; dst[11 * 64] = d22;
            }
          }
          if (PF_REG && do_pf) {
#pragma unroll
            for (int t = 0; t < NPF; ++t) {
              const int e = tid + T * t;
              if (e < 4 * NN) IMn[e] = pf[t];
            }
          }
          __syncthreads();
        }
        // ================= Matern scalars of the (pl, q) pairs: every wavefront for itself
        double* const sc_w = scal + (size_t)w * PG * NQ * 3;
        for (int t = lane; t < npg * NQ; t += 64) {
          const int pl = t / NQ, qq = t - pl * NQ;
          double nrm2 = 0.0;
          if (jv0 + qq < A.n_j) {
            const double* pp = part + (pl * NQ + qq) * N;
            for (int a = 0; a < N; ++a) nrm2 += pp[a];
          }
          const double nrm = sqrt5 * sqrt(0.5 * nrm2);
          const double ex = exp(-nrm * inv_sig);
          const double bp = ex * base_div;
          sc_w[t * 3 + 0] = 5.0 * bp;
          sc_w[t * 3 + 1] = -(sig * sig + sig * nrm) * bp;
          sc_w[t * 3 + 2] = -e_fact * (nrm + sig) * ex;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // ================= phase O: this wavefront's row atoms, all permutations of the group
        for (int pl = 0; pl < npg; ++pl) {
          const int p = g0 + pl;
          const int* pm_tab = permS + p * N;
          const int ap = pinvS[p * N + b];
          const double* sc = sc_w + (pl * NQ + q) * 3;
          const double beta = sc[0], cn = sc[1];
          const double* udp = ud + pl * 12 * 64 + lane;
          const double ur0 = udp[0], ur1 = udp[64], ur2 = udp[128];
          const double U0 = beta * ur0, U1 = beta * ur1, U2 = beta * ur2;
          {  // [pi a = b] blocks: -c_p dg_p goes to the one row atom a = pi^-1 b, if it is one of this wavefront's
            double DG[3][3];
#pragma unroll
            for (int al = 0; al < 3; ++al)
#pragma unroll
              for (int be = 0; be < 3; ++be) DG[al][be] = cn * udp[(3 + al * 3 + be) * 64];
#pragma unroll
            for (int k = 0; k < NA; ++k) {
              if (ap == (r * NA + k) * W + w) {
#pragma unroll
                for (int al = 0; al < 3; ++al)
#pragma unroll
                  for (int be = 0; be < 3; ++be) acc[k][al][be] += DG[al][be];
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (A.use_E && w == 0 && r == 0) {
            const double ce = sc[2];
            erow[0] += ce * ur0; erow[1] += ce * ur1; erow[2] += ce * ur2;
          }
          const double* vq = vs + (pl * NQ + q) * N * 3;
          const double* gjg = A.GD + ((int64_t)jpt * NN + b) * 3;
#pragma unroll
          for (int k = 0; k < NA; ++k) {
            const int a = (r * NA + k) * W + w;
            if (a < N) {
              const int pa = pm_tab[a];
              const double* vv = vq + a * 3;
              const double* gi = IMG ? SG + (ap * N + a) * 3 : GDi + (ap * N + a) * 3;
              const double* gj = GJS ? GjS + (pa * 64 + lane) * 3 : gjg + pa * N3;
              const double v0 = vv[0], v1 = vv[1], v2 = vv[2];
              const double g0v = gi[0], g1v = gi[1], g2v = gi[2];
              const double w0 = cn * gj[0], w1 = cn * gj[1], w2 = cn * gj[2];
              acc[k][0][0] += v0 * U0 + g0v * w0;
              acc[k][0][1] += v0 * U1 + g0v * w1;
              acc[k][0][2] += v0 * U2 + g0v * w2;
              acc[k][1][0] += v1 * U0 + g1v * w0;
              acc[k][1][1] += v1 * U1 + g1v * w1;
              acc[k][1][2] += v1 * U2 + g1v * w2;
              acc[k][2][0] += v2 * U0 + g2v * w0;
              acc[k][2][1] += v2 * U1 + g2v * w1;
              acc[k][2][2] += v2 * U2 + g2v * w2;
            }
            __builtin_amdgcn_sched_barrier(0);  // bounds the hoisting of the next atom's LDS reads (register pressure)
          }
        }
        __syncthreads();  // V-phase results of this group are free
      }

      // ---- write the rows of this round
      double* const trw = tr + w * 192;
#pragma unroll
      for (int k = 0; k < NA; ++k) {
        const int a = (r * NA + k) * W + w;
        if (a < N) {
#pragma unroll
          for (int al = 0; al < 3; ++al) {
            const int rr = 3 * a + al;                 // row inside the point
            const int64_t grow = row0 + rr;            // row of the full matrix
            int64_t lrow = lrow0 + rr;
            bool row_ok = true;
            if (A.cyc_W > 0) {  // the point's rows lie in row block cb0 (from offset coff0) and, past its end, in cb0 + 1
              const bool second = coff0 + rr >= A.cyc_nb;
              row_ok = second ? cmine1 : cmine0;
              lrow = second ? clrow1 + (coff0 + rr - A.cyc_nb) : clrow0 + coff0 + rr;
            }
            double o0 = acc[k][al][0], o1 = acc[k][al][1], o2 = acc[k][al][2];
            if (lower) { o0 = -o0; o1 = -o1; o2 = -o2; }
            if (A.fast_store) {
              trw[3 * lane + 0] = o0;
              trw[3 * lane + 1] = o1;
              trw[3 * lane + 2] = o2;
              __builtin_amdgcn_s_waitcnt(0xc07f);
              __builtin_amdgcn_wave_barrier();
              double* dst = A.K + lrow * A.ld;
#pragma unroll
              for (int t = 0; t < 3; ++t) {
                const double val = trw[64 * t + lane] + ((dcol[t] == rr) ? lamv : 0.0);
                if (tok[t]) dst[(unsigned)tcol[t]] = val;
              }
              __builtin_amdgcn_wave_barrier();
            } else if (row_ok && (!lower || jv <= i)) {
              double* dst = A.K + lrow * A.ld;
              if (outcol[0] >= 0) dst[outcol[0]] = o0 + ((lower && (int64_t)outcol[0] == grow) ? A.lam : 0.0);
              if (outcol[1] >= 0) dst[outcol[1]] = o1 + ((lower && (int64_t)outcol[1] == grow) ? A.lam : 0.0);
              if (outcol[2] >= 0) dst[outcol[2]] = o2 + ((lower && (int64_t)outcol[2] == grow) ? A.lam : 0.0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (A.use_E && w == 0 && r == 0) {
        double* dst = A.K + (A.M * N3 + i) * A.ld;
        if (outcol[0] >= 0) dst[outcol[0]] = erow[0];
        if (outcol[1] >= 0) dst[outcol[1]] = erow[1];
        if (outcol[2] >= 0) dst[outcol[2]] = erow[2];
      }
    }
    if (IMG) cur ^= 1;
  }
}

int build_dense_tables(gdml_ctx* ctx);

// LDS layout for one choice of (W, IMG, GJS, PG); returns the byte size
static size_t perm_layout(int N, int P, int NA, bool img, bool gjs, int PG, PermArgs* A) {
  const int W = PERM_W;
  const int NN = N * N;
  const int NQ = (62 + N) / N + 1;
  int o = 0;
  A->o_SG = o; o += img ? 8 * NN : 0;  // two buffers of [G | X]
  A->o_GjS = o; o += gjs ? N * 64 * 3 : 0;
  const int aux0 = o;
  A->o_vs = o; o += PG * NQ * N * 3;
  A->o_part = o; o += PG * NQ * N;
  A->o_ud = o; o += PG * 12 * 64;
  if (N <= W * NA) {  // one round per row point: the transposed rows alias vs | part | ud (free after the last O phase)
    A->o_tr = aux0;
    if (o - aux0 < W * 192) o = aux0 + W * 192;
  } else {            // several rounds reuse the V-phase results: own buffer
    A->o_tr = o; o += W * 192;
  }
  A->o_scal = o; o += W * PG * NQ * 3;
  o = (o + 1) & ~1;
  A->o_perm = o; o += (2 * P * N + 1) / 2;
  A->NQ = NQ;
  A->PG = PG;
  A->npass = (NQ * N + 63) / 64;
  return (size_t)o * 8;
}

template <int NA, bool IMG, bool GJS>
static void perm_launch_t(gdml_ctx* ctx, const PermArgs& A, dim3 grid, size_t lds) {
  hipFuncSetAttribute((const void*)assemble_perm_kernel<NA, IMG, GJS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((assemble_perm_kernel<NA, IMG, GJS>), grid, dim3(64 * PERM_W), lds, ctx->stream, A);
}

// Launch over the column points [0, n_j) of (jlist | j0 + v) and the row points [i_beg, i_end).
int assemble_perm_launch(gdml_ctx* ctx, double sig, int use_E, const int32_t* d_jlist, const int32_t* d_colmap, int64_t j0,
                         int64_t n_j, int64_t col0, double* K, int64_t ld, int64_t i_beg, int64_t i_end, int lower, double lam,
                         int cyc_W, int cyc_rank, int cyc_nb) {
  TrainSet& ts = ctx->ts;
  if (n_j <= 0 || i_end <= i_beg) return GDML_OK;
  GDML_TRY(build_dense_tables(ctx));
  const int N = ts.N, P = ts.P;
  if ((lower || cyc_W > 0) && (d_jlist || d_colmap || use_E || j0 != 0 || i_beg != 0 || n_j != ts.M || i_end != ts.M))
    return gdml_fail(ctx, GDML_ERR_INVALID, "assemble_perm: the lower form needs the dense full column range");
  if (cyc_W > 0 && 3 * N > cyc_nb)
    return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "assemble_perm: row-cyclic layout needs 3N <= %d", cyc_nb);
  PermArgs A;
  memset(&A, 0, sizeof(A));
  A.XF = ts.XF; A.GD = ts.GD; A.perm = ts.perm; A.pinv = ts.pinv;
  A.M = ts.M; A.N = N; A.P = P; A.sig = sig; A.use_E = use_E;
  A.jlist = d_jlist; A.colmap = d_colmap; A.j0 = j0; A.n_j = n_j; A.col0 = col0; A.n_cols = n_j * 3 * N;
  A.i_beg = i_beg; A.i_end = i_end; A.lower = (lower || cyc_W > 0) ? 1 : 0; A.lam = lam;
  A.cyc_W = cyc_W; A.cyc_rank = cyc_rank; A.cyc_nb = cyc_nb;
  A.K = K; A.ld = ld;
  A.fast_store = (!d_colmap && cyc_W == 0 && (col0 % 16) == 0 && ctx_opt_i(ctx, "asm.perm_fast_store", 1)) ? 1 : 0;

  // ---- shape: wavefronts per workgroup, what lives in LDS, permutations per group
  const int W = PERM_W;
  const int NA = (N <= 3 * W && ctx_opt_i(ctx, "asm.perm_na", 3) == 3) ? 3 : 6;
  const bool img_ok = N <= W * NA;
  int img = 0, gjs = 0, PG = 0;
  // options: asm.perm_img / asm.perm_gjs (0 / 1 force, -1 automatic), asm.perm_pg (permutations per group, 0 automatic)
  const int opt_gjs = ctx_opt_i(ctx, "asm.perm_gjs", -1), opt_img = ctx_opt_i(ctx, "asm.perm_img", -1);
  const int opt_pg = ctx_opt_i(ctx, "asm.perm_pg", 0);
  int pg_max = P < 8 ? P : 8, pg_want = P < 4 ? P : 4;
  if (opt_pg > 0) pg_max = pg_want = (opt_pg < P ? opt_pg : P);
  struct Cand { int img, gjs; };
  std::vector<Cand> cands;
  if (img_ok && opt_img != 0) {
    if (opt_gjs != 0) cands.push_back({1, 1});
    if (opt_gjs != 1) cands.push_back({1, 0});
  }
  if (opt_img != 1 || !img_ok) cands.push_back({0, 0});
  // one workgroup of 8 wavefronts per CU (register-bound): the whole LDS is the budget; a candidate is taken when it holds at
  // least pg_want permutations per group, on the last pass with whatever fits
  const size_t budgets[3] = {(size_t)160 * 1024, (size_t)160 * 1024, (size_t)160 * 1024};
  bool found = false;
  for (int bi = 0; bi < 3 && !found; ++bi)
    for (const Cand& c : cands) {
      int pg = pg_max;
      PermArgs tmp;
      while (pg >= 1 && perm_layout(N, P, NA, c.img, c.gjs, pg, &tmp) > budgets[bi]) --pg;
      if (pg >= (bi < 2 ? pg_want : 1)) {
        img = c.img; gjs = c.gjs; PG = pg; found = true;
        break;
      }
    }
  if (!found) return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "assemble_perm: no LDS layout for N=%d P=%d", N, P);
  PG = (P + (P + PG - 1) / PG - 1) / ((P + PG - 1) / PG);  // equal groups
  const size_t lds = perm_layout(N, P, NA, img, gjs, PG, &A);

  // ---- grid: strips x chunks of row points
  const int64_t n_strips = (n_j * N + 63) / 64;
  const int64_t n_i = i_end - i_beg;
  int i_chunk = ctx_opt_i(ctx, "asm.perm_i_chunk", 16);
  while (i_chunk > 2 && n_strips * ((n_i + i_chunk - 1) / i_chunk) < 2048) i_chunk >>= 1;
  A.i_chunk = i_chunk;
  dim3 grid((unsigned)n_strips, (unsigned)((n_i + i_chunk - 1) / i_chunk));
  const int slot = ktime_begin(ctx);
  if (NA == 3) {
    if (img && gjs) perm_launch_t<3, true, true>(ctx, A, grid, lds);
    else if (img) perm_launch_t<3, true, false>(ctx, A, grid, lds);
    else perm_launch_t<3, false, false>(ctx, A, grid, lds);
  } else {
    if (img && gjs) perm_launch_t<6, true, true>(ctx, A, grid, lds);
    else if (img) perm_launch_t<6, true, false>(ctx, A, grid, lds);
    else perm_launch_t<6, false, false>(ctx, A, grid, lds);
  }
  const double blocks = A.lower ? 0.5 * (double)n_i * (double)(n_i + 1) : (double)n_i * (double)n_j;
  ktime_end(ctx, slot, "assemble", 8.0 * blocks * 9.0 * N * N);
  ctx->launch_counter++;
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}
