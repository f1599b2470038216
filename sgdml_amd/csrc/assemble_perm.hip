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
//   * the 8 wavefronts split the ROW atoms (a = k 8 + w): 9 NA accumulators per lane, summed over p in registers;
//   * per row point and group of PG permutations two phases, one barrier each:
//       V  (tasks spread over the wavefronts)  V12(p, pass): lane = (column point q, row atom a): v_p, |d_p|^2 and from
//                                                            it the Matern scalars of (p, q)
//                                              V3(p):        lane = column atom: u_p (3), dg_p (3 x 3)      -> LDS
//       O  every wavefront, its own row atoms:  acc[a] += beta v_p[a] (x) u_p + (-c_p) G_i(a,a') (x) G_j(b, pi a)
//                                               and the one row atom a = pi^-1 b gets (-c_p) dg_p
//   * finished rows are transposed through LDS (three rows at a time) so that every store instruction writes 64
//     consecutive doubles (full cache lines; the pattern of assemble_strip.hip).
// Everything an inner loop reads comes from LDS when it fits -- row-point image (GD_i, XF_i: IMG), the strip's G_j
// table (GJS), its x_j tables (JX) -- otherwise through L1/L2 from the dense tables (any N up to GDML_MAX_ATOMS;
// more than 8 NA row atoms take several rounds).  Permutation rows are held one entry per lane and read with
// v_readlane (wave-uniform index), so no inner loop has an index load in its dependency chain.
#include "common.h"
#include <type_traits>

struct PermArgs {
  const double* XF;   // (M,N,N)
  const double* GD;   // (M,N,N,3)
  const int32_t* perm;  // (P,N)  pi_p
  const int32_t* pinv;  // (P,N)  pi_p^-1
  int64_t M;
  int N, P;
  double sig;
  int use_E;             // also write the energy-constraint row K[3N M + i, .]  (train.py:235-248)
  int64_t e_row0;        // that row is row e_row0 + i of K: 3N M, or (sharded rows) 3N (i_end - i_beg) - i_beg
  const int32_t* jlist;  // virtual column point -> training point (null: j0 + v)
  const int32_t* colmap; // (n_j, 3N) output column or -1 (null: col0 + 3N v + c)
  // compact index-list mode: strips are 64 REQUESTED column atoms (entry = virtual point * N + atom, -1 = idle lane) of at
  // most NQ consecutive virtual points starting at strip_jv0[s]; null: strips of 64 consecutive column atoms
  const int32_t* calist;
  const int32_t* strip_jv0;
  int64_t j0, n_j, col0;
  int64_t i_beg, i_end;  // row points of this launch; rows are written relative to i_beg
  int i_chunk;
  int lower;             // store -K + lam I, only blocks j <= i (dense full column range)
  double lam;
  int cyc_W, cyc_rank, cyc_nb;  // block-row-cyclic local layout of the distributed Cholesky (implies lower)
  int fast_store;        // dense columns, plain row layout: transposed full-line stores
  int dbg;               // timing-only ablation mask (asm.perm_debug): 1 no stores, 2 no V tasks, 4 no O phase, 8 no image prefetch
  int NQ, PG, npass, n_img;
  int o_IM, o_GjS, o_XjS, o_Xq, o_vs, o_part, o_scal, o_ud, o_tr, o_pinv;  // LDS offsets in doubles
  double* K;
  int64_t ld;
};


// entry m of a permutation row; m is wave-uniform.  BIG = 0 (N <= 64) / 1 (N <= 128): the row is held one entry per lane
// (row0: 0..63, row1: 64..127) and read with v_readlane -- no index load in the dependency chain of an inner loop.  Branch-free:
// a branch here keeps the compiler from unrolling / pipelining the loops around it.  BIG = 2 (any N): a broadcast read of the
// row's LDS copy (molecules beyond 128 atoms; the reference has no size limit, train.py:97-302).
template <int BIG>
__device__ __forceinline__ int perm_at(int row0, int row1, const int* rowS, int m) {
  if (BIG == 2) return rowS[m];
  if (BIG == 0) return __builtin_amdgcn_readlane(row0, m);
  const int lo = __builtin_amdgcn_readlane(row0, m & 63), hi = __builtin_amdgcn_readlane(row1, m & 63);
  return (m < 64) ? lo : hi;
}

// Global load that stays where it is written.  A plain load carried across loop iterations in a register is rewritten by
// LLVM into a load at the point of use (InstCombine folds a PHI of loads into a load of a PHI of addresses), which turns
// every software-prefetched operand back into load -> s_waitcnt vmcnt(0) -> use; a relaxed atomic load at wavefront scope is
// the same global_load instruction (no cache-bypass bits) but is neither folded nor reordered.
__device__ __forceinline__ double ldg_pin(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

template <bool PIN>
__device__ __forceinline__ double ldg(const double* p) {
  return PIN ? ldg_pin(p) : *p;
}

// inclusive scan inside segments of consecutive lanes; pos = position of the lane in its segment.  Only additions of
// the segment's own terms (a scan over the whole wavefront and a difference would cancel: |d|^2 = 0 must stay 0)
__device__ __forceinline__ double seg_incl_scan(double v, int pos) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double t = __shfl_up(v, off, 64);
    if (pos >= off) v += t;
  }
  return v;
}

// W: wavefronts per workgroup; NA: row atoms per wavefront and round (W NA >= N: one round); BIG: 0 N <= 64, 1 N <= 128,
// 2 any N (permutation entries from LDS instead of lane-held rows).
// Shapes built: (W, NA) = (4, 6): N <= 24, two independent workgroups per CU when the LDS allows; (8, 3): N <= 24, one
// workgroup; (8, 6): N <= 48 in one round, larger molecules in several.
template <int W, int NA, bool IMG, bool GJS, bool JX, int BIG>
__global__ void __launch_bounds__(64 * W, 2) assemble_perm_kernel(PermArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int T = 64 * W;
#ifndef PERM_ABL
#define PERM_ABL 0  // timing-only ablation of the V phase (tools/asm_perm_vabl.sh): 1 no V3 tasks, 2 no V12 tasks, 4 V12 without
#endif              // global loads, 8 V12 without LDS reads, 16 V12 without epilogue, 32 V3 without global loads, 64 V3 without LDS reads
  // iterations per stage of the V-phase contractions (three stages of operands are in flight): fewer when more of the
  // operands come from global memory and have to wait in registers
  constexpr int VU3 = (IMG && !GJS) ? 2 : 3;
  // Row-point image in LDS: the few global operands left (x_j, G_j) run through the three-buffer pipeline below.  Without
  // it (large molecules) every operand is global and the pipeline loses (three 8-byte pinned loads per G vector instead of
  // 16 + 8, twice the registers: N = 100 6.35 vs 5.34 ms): one stage at a time, all its loads up front.
  constexpr bool PIPE = IMG;
  const int N = A.N, N3 = 3 * N, NN = N * N, P = A.P, NQ = A.NQ, PG = A.PG, npass = A.npass;
  const int SEG = BIG ? (N + 63) >> 6 : 1;  // 64-atom segments of a point (V12 passes of large molecules)
  const int ppp = BIG ? 1 : 64 / N;          // whole column points per V12 pass (N <= 64)
  // image of the row point (IMG): [G | X] = [m][b][al] | [m][b]; one or two buffers of 4 N^2 doubles
  double* const IM0 = smem + A.o_IM;
  double* const GjS = smem + A.o_GjS;    // [m][lane][be] G_j(b,m) of the lane's column atom (GJS)
  double* const XjS = smem + A.o_XjS;    // [m][lane]     x_j[pair(b,m)]                       (JX)
  double* const Xq = smem + A.o_Xq;      // [q][m][b]     dense x of the strip's column points  (JX)
  double* const vs = smem + A.o_vs;      // [pl][q][a][al]
  double* const part = smem + A.o_part;  // [pl][q][seg]  |d|^2 pieces (N > 64)
  double* const scal = smem + A.o_scal;  // [pl][q][3]    beta, -c, E-row coefficient   (N > 64: one copy per wavefront)
  double* const ud = smem + A.o_ud;      // [pl][12][lane]  u (3), dg (3 x 3)
  double* const tr = smem + A.o_tr;      // [w][3][192]   transposed output rows (aliases the V-phase results)
  int* const pinvS = reinterpret_cast<int*>(smem + A.o_pinv);  // [P][N]
  int* const permS = pinvS + P * N;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t s = blockIdx.x;
  const int64_t n_ca = A.n_j * N;  // column atoms of this launch
  int64_t gav = 64 * s + lane;
  bool cvalid = gav < n_ca;
  int jv0 = (int)((64 * s) / N);  // first (virtual) column point of the strip
  if (A.calist) {  // compact list: this lane's requested column atom, or an idle lane
    const int ent = A.calist[64 * s + lane];
    jv0 = A.strip_jv0[s];
    cvalid = ent >= 0;
    gav = cvalid ? ent : (int64_t)jv0 * N;
  }
  const int jv = cvalid ? (int)(gav / N) : (A.calist ? jv0 : (int)A.n_j - 1);
  const int b = cvalid ? (int)(gav - (int64_t)jv * N) : 0;
  const int q = jv - jv0;
  const int jpt = A.jlist ? A.jlist[jv] : (int)A.j0 + jv;
  const bool lower = A.lower != 0;
  const bool two_img = A.n_img == 2;

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
    pinvS[e] = A.pinv[e];
    permS[e] = A.perm[e];
  }
  if (GJS) {
    const double* gd = A.GD + ((int64_t)jpt * NN + b) * 3;
    for (int m = w; m < N; m += W) {
      GjS[(m * 64 + lane) * 3 + 0] = gd[m * N3 + 0];
      GjS[(m * 64 + lane) * 3 + 1] = gd[m * N3 + 1];
      GjS[(m * 64 + lane) * 3 + 2] = gd[m * N3 + 2];
    }
  }
  if (JX) {
    const double* xf = A.XF + (int64_t)jpt * NN + b;
    for (int m = w; m < N; m += W) XjS[m * 64 + lane] = xf[m * N];
    for (int e = tid; e < NQ * NN; e += T) {
      const int qq = e / NN;
      int64_t jvq = jv0 + qq;
      if (jvq > A.n_j - 1) jvq = A.n_j - 1;
      const int64_t jq = A.jlist ? (int64_t)A.jlist[jvq] : A.j0 + jvq;
      Xq[e] = A.XF[jq * NN + (e - qq * NN)];
    }
  }
  if (IMG) {
    const double* gi = A.GD + i_lo * (int64_t)NN * 3;
    const double* xi = A.XF + i_lo * (int64_t)NN;
    for (int e = tid; e < 3 * NN; e += T) IM0[e] = gi[e];
    for (int e = tid; e < NN; e += T) IM0[3 * NN + e] = xi[e];
  }
  int cur = 0;

  // V12 passes this strip needs: dense strips touch 2 .. NQ column points (a strip of 64 atoms starts anywhere in a point)
  int npass_e = npass;
  if (!A.calist) {
    int64_t last = (64 * s + 63) / N;
    if (last > A.n_j - 1) last = A.n_j - 1;
    int nq_used = (int)(last - jv0) + 1;
    if (nq_used > NQ) nq_used = NQ;
    npass_e = BIG ? nq_used * SEG : (nq_used + ppp - 1) / ppp;
  }
  constexpr bool PF_REG = IMG && NA == 3;  // next image through registers underneath the first V phase (when they are there)
  constexpr int NPF = PF_REG ? (4 * (W * NA) * (W * NA) + T - 1) / T : 1;
  const int n_rounds = (N + W * NA - 1) / (W * NA);
  const int n_groups = (P + PG - 1) / PG;

  for (int64_t i = i_lo; i < i_hi; ++i) {
    const int64_t row0 = i * N3;  // first row of the point in the full matrix
    const double lamv = lower ? A.lam : 0.0;
    const int64_t lrow0 = row0 - A.i_beg * N3;  // ... in the stored matrix (plain layout)
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
        if (IMG && i + 1 < i_hi) {           // keep the image in step
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
    double* const IMn = IM0 + (two_img ? (cur ^ 1) : 0) * 4 * NN;  // where the next point's image goes
    const double* const GDi = A.GD + i * (int64_t)NN * 3;
    const double* const XFi = A.XF + i * (int64_t)NN;
    const int64_t i_next = (i + 1 < i_hi) ? i + 1 : i;

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
          double pf[NPF];
          const bool do_pf = IMG && two_img && g0 == 0 && !(A.dbg & 8);
          if (do_pf) {
            const double* gi = A.GD + i_next * (int64_t)NN * 3;
            const double* xi = A.XF + i_next * (int64_t)NN;
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
          // ---- V12 for a batch of PB permutations of the group (pl0 ...): lane = (column point qq, row atom a), whole
          // points per pass (N <= 64) or 64-atom segments.  x_i and G_i(a, m) do not depend on the permutation: they
          // are read once per m for the whole batch (the V phase is bound by LDS throughput, not by arithmetic:
          // profiles/r04_assemble_perm_ablation.txt); per permutation only x_j[pair(pi a, pi m)] differs.
          auto v12_batch = [&](auto PBc, int kind, int pl0) {
            constexpr int PB = decltype(PBc)::value;
            constexpr int VU12 = (PB >= 4) ? 2 : 3;  // iterations per stage
            int qq, a, seg = 0;
            if (!BIG) {
              const int ql = lane / N;
              qq = kind * ppp + ql;
              a = lane - ql * N;
              if (ql >= ppp) qq = NQ;  // idle lanes
            } else {
              qq = kind / SEG;
              seg = kind - qq * SEG;
              a = 64 * seg + lane;
            }
            const int jvq = jv0 + qq;
            const bool ok = qq < NQ && a < N && jvq < A.n_j;
            const int ac = ok ? a : 0, qc = ok ? qq : 0;
            const int jvc = (jvq < A.n_j) ? jvq : (int)A.n_j - 1;
            const int64_t jq = A.jlist ? (int64_t)A.jlist[jvc] : A.j0 + jvc;
            const double* const xjg = A.XF + jq * NN;  // x_j (global; JX: the LDS copy)
            const double* const xjl = Xq + qc * NN;
            int pr0[PB], pr1[PB], pa[PB];
            const int* prS[PB];
#pragma unroll
            for (int bb = 0; bb < PB; ++bb) {
              const int plb = (pl0 + bb < npg) ? pl0 + bb : npg - 1;  // short last batch: repeats its last permutation
              const int p = g0 + plb;
              pr0[bb] = (lane < N) ? permS[p * N + lane] : 0;
              pr1[bb] = (BIG && lane + 64 < N) ? permS[p * N + 64 + lane] : 0;
              prS[bb] = permS + p * N;
              pa[bb] = permS[p * N + ac];
            }
            double v[PB][3], nn[PB];
#pragma unroll
            for (int bb = 0; bb < PB; ++bb) v[bb][0] = v[bb][1] = v[bb][2] = nn[bb] = 0.0;
            // The contraction over m runs in stages of VU12 iterations through three operand buffers: the GLOBAL operands
            // of stage s + 2 are requested (ldg_pin) before the multiply-adds of stage s, so a global round trip is
            // covered by two stages of work; LDS operands are read inside the stage.  (History: the compiler refuses to
            // unroll a loop with a run-time trip count around v_readlane, and it moves plain prefetch loads back to
            // their use.)  Past-the-end iterations re-read entry N - 1 and are multiplied by zero.
            struct V12Buf {
              double xj[JX ? 1 : PB][JX ? 1 : VU12];
              double xi[IMG ? 1 : VU12];
              double g[IMG ? 1 : VU12][3];
            };
            auto ld12 = [&](int m0, V12Buf& B) {
              if ((JX && IMG) || (PERM_ABL & 4)) return;
#pragma unroll
              for (int u = 0; u < VU12; ++u) {
                const int m = (m0 + u < N) ? m0 + u : N - 1;
                if (!JX) {
#pragma unroll
                  for (int bb = 0; bb < PB; ++bb) B.xj[bb][u] = ldg<PIPE>(xjg + pa[bb] + perm_at<BIG>(pr0[bb], pr1[bb], prS[bb], m) * N);
                }
                if (!IMG) {
                  B.xi[u] = XFi[m * N + ac];
                  const double* g = GDi + (m * N + ac) * 3;
                  B.g[u][0] = g[0];
                  B.g[u][1] = g[1];
                  B.g[u][2] = g[2];
                }
              }
            };
            auto comp12 = [&](int m0, const V12Buf& B) {
              double xi_[VU12], g_[VU12][3], xj_[PB][VU12];
#pragma unroll
              for (int u = 0; u < VU12; ++u) {
                const int m = (m0 + u < N) ? m0 + u : N - 1;
                xi_[u] = (PERM_ABL & 8) ? 1.0 + m : (IMG ? SX[m * N + ac] : B.xi[u]);
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) g_[u][c3] = (PERM_ABL & 8) ? 0.5 * m : (IMG ? SG[(m * N + ac) * 3 + c3] : B.g[u][c3]);
#pragma unroll
                for (int bb = 0; bb < PB; ++bb)
                  xj_[bb][u] = (PERM_ABL & 4) ? 0.25 * perm_at<BIG>(pr0[bb], pr1[bb], prS[bb], m)
                                              : (JX ? xjl[pa[bb] + perm_at<BIG>(pr0[bb], pr1[bb], prS[bb], m) * N] : B.xj[bb][u]);
              }
#pragma unroll
              for (int u = 0; u < VU12; ++u) {
                // past-the-end iterations: multiplied by zero, not selected -- a select on the wave-uniform condition
                // becomes a branch around the operand loads, and with branches in the body LLVM sinks every stage's
                // multiply-adds below the last stage's loads
                const double lv = (m0 + u < N) ? 1.0 : 0.0;
#pragma unroll
                for (int bb = 0; bb < PB; ++bb) {
                  const double d = (xi_[u] - xj_[bb][u]) * lv;
                  nn[bb] += d * d;
                  v[bb][0] += d * g_[u][0];
                  v[bb][1] += d * g_[u][1];
                  v[bb][2] += d * g_[u][2];
                }
              }
            };
            if (!PIPE) {
              V12Buf B0;
              for (int m0 = 0; m0 < N; m0 += VU12) {
                ld12(m0, B0);
                comp12(m0, B0);
                __builtin_amdgcn_sched_barrier(0);
              }
            } else {
              V12Buf B0, B1, B2;
              ld12(0, B0);
              ld12(VU12, B1);
              for (int m0 = 0; m0 < N; m0 += 3 * VU12) {
                ld12(m0 + 2 * VU12, B2);
                comp12(m0, B0);
                __builtin_amdgcn_sched_barrier(0);
                ld12(m0 + 3 * VU12, B0);
                comp12(m0 + VU12, B1);
                __builtin_amdgcn_sched_barrier(0);
                ld12(m0 + 4 * VU12, B1);
                comp12(m0 + 2 * VU12, B2);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
#pragma unroll
            for (int bb = 0; bb < PB; ++bb) {
              const int pl = pl0 + bb;
              if (pl >= npg) break;  // wave-uniform
              if ((PERM_ABL & 16) && v[bb][0] + nn[bb] != 1.2345e-300) break;
              if (ok) {
                double* dst = vs + ((pl * NQ + qq) * N + a) * 3;
                dst[0] = v[bb][0];
                dst[1] = v[bb][1];
                dst[2] = v[bb][2];
              }
              // |d_p|^2 of every point of the pass (segmented sum over its N lanes), then the Matern scalars
              if (!BIG) {
                const double nrm2 = seg_incl_scan(ok ? nn[bb] : 0.0, a);  // complete in the lane of the point's last atom
                if (ok && a == N - 1) {
                  const double nrm = sqrt5 * sqrt(0.5 * nrm2);
                  const double ex = exp(-nrm * inv_sig);
                  const double bp = ex * base_div;
                  double* sc = scal + (pl * NQ + qq) * 3;
                  sc[0] = 5.0 * bp;
                  sc[1] = -(sig * sig + sig * nrm) * bp;
                  sc[2] = -e_fact * (nrm + sig) * ex;
                }
              } else {
                const double tot = wave_sum(ok ? nn[bb] : 0.0);
                if (lane == 0 && qq < NQ) part[(pl * NQ + qq) * SEG + seg] = tot;
              }
            }
          };
          // Tasks of the group, longest first: npass_e x nbatch batched V12 tasks, then one V3 task per permutation
          // (batches of 2 / 4 permutations per task measured: N = 42, P = 27 32.7 / 38.5 ms against 31.4 with one -- fewer,
          // longer tasks leave wavefronts idle and put four norm epilogues in a row on the critical path)
          const int PBG = 1;
          const int nbatch = (npg + PBG - 1) / PBG;
          const int nv12 = npass_e * nbatch;
          const int ntask = (A.dbg & 2) ? 0 : nv12 + npg;
          for (int t = w; t < ntask; t += W) {
            if ((PERM_ABL & 1) && t >= nv12) continue;
            if ((PERM_ABL & 2) && t < nv12) continue;
            if (t < nv12) {
              const int kind = t / nbatch, pl0 = (t - kind * nbatch) * PBG;
              if (PBG == 4) v12_batch(std::integral_constant<int, 4>{}, kind, pl0);
              else if (PBG == 2) v12_batch(std::integral_constant<int, 2>{}, kind, pl0);
              else v12_batch(std::integral_constant<int, 1>{}, kind, pl0);
            } else {
              const int pl = t - nv12;
              const int p = g0 + pl;
              // ---- V3: lane = column atom (j, b)
              const int pi0 = (lane < N) ? pinvS[p * N + lane] : 0;
              const int pi1 = (BIG && lane + 64 < N) ? pinvS[p * N + 64 + lane] : 0;
              const int* const piS = pinvS + p * N;
              const int ap = pinvS[p * N + b];
              const double* xfj = A.XF + (int64_t)jpt * NN + b;
              const double* gdj = A.GD + ((int64_t)jpt * NN + b) * 3;
              double u0 = 0.0, u1 = 0.0, u2 = 0.0;
              double d00 = 0.0, d01 = 0.0, d02 = 0.0, d10 = 0.0, d11 = 0.0, d12 = 0.0, d20 = 0.0, d21 = 0.0, d22 = 0.0;
              struct V3Buf {
                double xj[JX ? 1 : VU3];
                double r[GJS ? 1 : VU3][3];
                double xi[IMG ? 1 : VU3];
                double gi[IMG ? 1 : VU3][3];
              };
              auto ld3 = [&](int m0, V3Buf& B) {
                if ((JX && GJS && IMG) || (PERM_ABL & 32)) return;
#pragma unroll
                for (int u = 0; u < VU3; ++u) {
                  const int mp = (m0 + u < N) ? m0 + u : N - 1;
                  if (!JX) B.xj[u] = ldg<PIPE>(xfj + mp * N);
                  if (!GJS) {
                    B.r[u][0] = ldg<PIPE>(gdj + mp * N3);
                    B.r[u][1] = ldg<PIPE>(gdj + mp * N3 + 1);
                    B.r[u][2] = ldg<PIPE>(gdj + mp * N3 + 2);
                  }
                  if (!IMG) {
                    const int mi = perm_at<BIG>(pi0, pi1, piS, mp);
                    B.xi[u] = XFi[mi * N + ap];
                    const double* gi = GDi + (mi * N + ap) * 3;
                    B.gi[u][0] = gi[0];
                    B.gi[u][1] = gi[1];
                    B.gi[u][2] = gi[2];
                  }
                }
              };
              auto comp3 = [&](int m0, const V3Buf& B) {
                double xj_[VU3], xi_[VU3], r_[VU3][3], gi_[VU3][3];
#pragma unroll
                for (int u = 0; u < VU3; ++u) {
                  const int mp = (m0 + u < N) ? m0 + u : N - 1;
                  const int mi = perm_at<BIG>(pi0, pi1, piS, mp);
                  xj_[u] = (PERM_ABL & 32) ? 0.25 * mi : (JX ? XjS[mp * 64 + lane] : B.xj[u]);
                  xi_[u] = (PERM_ABL & 64) ? 1.0 + mi : (IMG ? SX[mi * N + ap] : B.xi[u]);
#pragma unroll
                  for (int c3 = 0; c3 < 3; ++c3) {
                    r_[u][c3] = (PERM_ABL & 32) ? 0.125 * mi : (GJS ? GjS[(mp * 64 + lane) * 3 + c3] : B.r[u][c3]);
                    gi_[u][c3] = (PERM_ABL & 64) ? 0.5 * mi : (IMG ? SG[(mi * N + ap) * 3 + c3] : B.gi[u][c3]);
                  }
                }
#pragma unroll
                for (int u = 0; u < VU3; ++u) {
                  const double lv = (m0 + u < N) ? 1.0 : 0.0;  // every product below has a factor r
                  const double d = xi_[u] - xj_[u];
                  const double r0 = r_[u][0] * lv, r1 = r_[u][1] * lv, r2 = r_[u][2] * lv;
                  const double g0v = gi_[u][0], g1v = gi_[u][1], g2v = gi_[u][2];
                  u0 += d * r0; u1 += d * r1; u2 += d * r2;
                  d00 += g0v * r0; d01 += g0v * r1; d02 += g0v * r2;
                  d10 += g1v * r0; d11 += g1v * r1; d12 += g1v * r2;
                  d20 += g2v * r0; d21 += g2v * r1; d22 += g2v * r2;
                }
              };
              if (!PIPE) {
                V3Buf B0;
                for (int m0 = 0; m0 < N; m0 += VU3) {
                  ld3(m0, B0);
                  comp3(m0, B0);
                  __builtin_amdgcn_sched_barrier(0);
                }
              } else {
                V3Buf B0, B1, B2;
                ld3(0, B0);
                ld3(VU3, B1);
                for (int m0 = 0; m0 < N; m0 += 3 * VU3) {
                  ld3(m0 + 2 * VU3, B2);
                  comp3(m0, B0);
                  __builtin_amdgcn_sched_barrier(0);
                  ld3(m0 + 3 * VU3, B0);
                  comp3(m0 + VU3, B1);
                  __builtin_amdgcn_sched_barrier(0);
                  ld3(m0 + 4 * VU3, B1);
                  comp3(m0 + 2 * VU3, B2);
                  __builtin_amdgcn_sched_barrier(0);
                }
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
        // ================= large molecules: the Matern scalars from the segment sums, every wavefront for itself
        const double* sc_base = scal;
        if (BIG) {
          double* const sc_w = scal + (size_t)w * PG * NQ * 3;
          for (int t = lane; t < npg * NQ; t += 64) {
            double nrm2 = 0.0;
            for (int sg = 0; sg < SEG; ++sg) nrm2 += part[t * SEG + sg];
            const double nrm = sqrt5 * sqrt(0.5 * nrm2);
            const double ex = exp(-nrm * inv_sig);
            const double bp = ex * base_div;
            sc_w[t * 3 + 0] = 5.0 * bp;
            sc_w[t * 3 + 1] = -(sig * sig + sig * nrm) * bp;
            sc_w[t * 3 + 2] = -e_fact * (nrm + sig) * ex;
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);
          __builtin_amdgcn_wave_barrier();
          sc_base = sc_w;
        }
        // ================= phase O: this wavefront's row atoms, all permutations of the group
        for (int pl = 0; pl < ((A.dbg & 4) ? 0 : npg); ++pl) {
          const int p = g0 + pl;
          const int pr0 = (lane < N) ? permS[p * N + lane] : 0;
          const int pr1 = (BIG && lane + 64 < N) ? permS[p * N + 64 + lane] : 0;
          const int* const prS = permS + p * N;
          const int ap = pinvS[p * N + b];
          const double* sc = sc_base + (pl * NQ + q) * 3;
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
          if (A.use_E && w == 0 && r == 0) {
            const double ce = sc[2];
            erow[0] += ce * ur0; erow[1] += ce * ur1; erow[2] += ce * ur2;
          }
          const double* vq = vs + (pl * NQ + q) * N * 3;
          const double* gjg = A.GD + ((int64_t)jpt * NN + b) * 3;
          const double* gib = IMG ? SG + ap * N3 : GDi + ap * N3;  // G_i(., pi^-1 b): [a][al]
#pragma unroll
          for (int h = 0; h < NA / 3; ++h) {
            // operands of three row atoms in one batch of loads, then their 81 multiply-adds
            double vv[3][3], gi[3][3], gj[3][3];
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
              const int a0 = (r * NA + 3 * h + kk) * W + w;
              const int a = (a0 < N) ? a0 : N - 1;
              const int pa = perm_at<BIG>(pr0, pr1, prS, a);
              const double* gjp = GJS ? GjS + (pa * 64 + lane) * 3 : gjg + pa * N3;
#pragma unroll
              for (int c3 = 0; c3 < 3; ++c3) {
                vv[kk][c3] = vq[a * 3 + c3];
                gi[kk][c3] = gib[a * 3 + c3];
                gj[kk][c3] = gjp[c3];
              }
            }
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
              const int k = 3 * h + kk;
              const int a0 = (r * NA + k) * W + w;
              if (a0 < N) {
                const double w0 = cn * gj[kk][0], w1 = cn * gj[kk][1], w2 = cn * gj[kk][2];
#pragma unroll
                for (int al = 0; al < 3; ++al) {
                  acc[k][al][0] = fma(gi[kk][al], w0, fma(vv[kk][al], U0, acc[k][al][0]));
                  acc[k][al][1] = fma(gi[kk][al], w1, fma(vv[kk][al], U1, acc[k][al][1]));
                  acc[k][al][2] = fma(gi[kk][al], w2, fma(vv[kk][al], U2, acc[k][al][2]));
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __syncthreads();  // V-phase results of this group are free
      }

      // ---- write the rows of this round: three rows (one row atom) per LDS round trip.
      // The per-lane constants of the stores (columns, diagonal crossing, LDS addresses) are recomputed here from an
      // opaque copy of the lane index: kept live across the V and O phases they end up in scratch, and a scratch reload
      // between two global stores has to wait (vmcnt counts both) for the stores before it -- the store stream ran
      // one store at a time.
      int lane_s = lane;
      asm volatile("" : "+v"(lane_s));
      double* const trw = tr + w * 3 * 192;
      int outcol[3] = {-1, -1, -1};  // output columns of the lane (general store path)
      int tcol[3] = {-1, -1, -1};    // ... of the transposed rows (fast path), relative to column 0
      bool tok[3] = {false, false, false};
      int dcol[3] = {-1, -1, -1};    // where the matrix diagonal crosses them (lower form: + lam)
      if (A.fast_store) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int64_t c = 192 * s + 64 * t + lane_s;  // relative to col0
          tcol[t] = (c < n_ca * 3) ? (int)(A.col0 + c) : -1;
          const int tpt = (int)((unsigned)c / (unsigned)N3);
          tok[t] = tcol[t] >= 0 && (!lower || tpt <= i);
          dcol[t] = lower ? (int)((int64_t)tcol[t] - row0) : -1;
        }
      }
      if (!A.fast_store || A.use_E) {
        const int bs = (lane_s == lane) ? b : 0;  // (always b: ties the loads below to this point of the program)
#pragma unroll
        for (int be = 0; be < 3; ++be)
          if (cvalid) outcol[be] = A.colmap ? A.colmap[(int64_t)jv * N3 + 3 * bs + be] : (int)A.col0 + jv * N3 + 3 * bs + be;
      }
#pragma unroll
      for (int k = 0; k < NA; ++k) {
        const int a = (r * NA + k) * W + w;
        if (a < N && (!(A.dbg & 1) || acc[k][0][0] == 1.2345e-300)) {
          if (A.fast_store) {
#pragma unroll
            for (int al = 0; al < 3; ++al) {
#pragma unroll
              for (int be = 0; be < 3; ++be) trw[al * 192 + 3 * lane_s + be] = lower ? -acc[k][al][be] : acc[k][al][be];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            double val[3][3];
#pragma unroll
            for (int al = 0; al < 3; ++al)
#pragma unroll
              for (int t = 0; t < 3; ++t) val[al][t] = trw[al * 192 + 64 * t + lane_s];
#pragma unroll
            for (int al = 0; al < 3; ++al) {
              const int rr = 3 * a + al;
              double* dst = A.K + (lrow0 + rr) * A.ld;
#pragma unroll
              for (int t = 0; t < 3; ++t) {
                const double o = val[al][t] + ((dcol[t] == rr) ? lamv : 0.0);
                if (tok[t]) dst[(unsigned)tcol[t]] = o;
              }
            }
            __builtin_amdgcn_wave_barrier();
          } else {
#pragma unroll
            for (int al = 0; al < 3; ++al) {
              const int rr = 3 * a + al;        // row inside the point
              const int64_t grow = row0 + rr;   // row of the full matrix
              int64_t lrow = lrow0 + rr;
              bool row_ok = true;
              if (A.cyc_W > 0) {  // the point's rows lie in row block cb0 (from offset coff0) and, past its end, in cb0 + 1
                const bool second = coff0 + rr >= A.cyc_nb;
                row_ok = second ? cmine1 : cmine0;
                lrow = second ? clrow1 + (coff0 + rr - A.cyc_nb) : clrow0 + coff0 + rr;
              }
              if (row_ok && (!lower || jv <= i)) {
                double* dst = A.K + lrow * A.ld;
#pragma unroll
                for (int be = 0; be < 3; ++be) {
                  const double o = (lower ? -acc[k][al][be] : acc[k][al][be]) + ((lower && (int64_t)outcol[be] == grow) ? A.lam : 0.0);
                  if (outcol[be] >= 0) dst[outcol[be]] = o;
                }
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (A.use_E && w == 0 && r == 0) {
        double* dst = A.K + (A.e_row0 + i) * A.ld;
        if (outcol[0] >= 0) dst[outcol[0]] = erow[0];
        if (outcol[1] >= 0) dst[outcol[1]] = erow[1];
        if (outcol[2] >= 0) dst[outcol[2]] = erow[2];
      }
    }
    if (IMG) {
      if (two_img) cur ^= 1;
      else if (i + 1 < i_hi) {  // single buffer: every reader of the image passed the last barrier
        const double* gi = A.GD + i_next * (int64_t)NN * 3;
        const double* xi = A.XF + i_next * (int64_t)NN;
#pragma unroll 6
        for (int e = tid; e < 4 * NN; e += T) IM0[e] = (e < 3 * NN) ? gi[e] : xi[e - 3 * NN];
      }
    }
  }
}

int build_dense_tables(gdml_ctx* ctx);

// LDS layout for one choice of what is resident; returns the byte size
static size_t perm_layout(int N, int P, int W, int NA, int n_img, bool gjs, bool jx, int PG, PermArgs* A, int nq = 0) {
  const int NN = N * N;
  const int NQ = nq > 0 ? nq : (62 + N) / N + 1;
  const int SEG = (N + 63) / 64;
  int o = 0;
  A->o_IM = o; o += n_img * 4 * NN;
  A->o_GjS = o; o += gjs ? N * 64 * 3 : 0;
  A->o_XjS = o; o += jx ? N * 64 : 0;
  A->o_Xq = o; o += jx ? NQ * NN : 0;
  const int aux0 = o;
  A->o_vs = o; o += PG * NQ * N * 3;
  A->o_ud = o; o += PG * 12 * 64;
  if (N <= W * NA) {  // one round per row point: the transposed rows alias vs | ud (free after the last O phase)
    A->o_tr = aux0;
    if (o - aux0 < W * 3 * 192) o = aux0 + W * 3 * 192;
  } else {            // several rounds reuse the V-phase results: own buffer
    A->o_tr = o; o += W * 3 * 192;
  }
  A->o_part = o; o += PG * NQ * SEG;
  A->o_scal = o; o += (N > 64 ? W : 1) * PG * NQ * 3;
  o = (o + 1) & ~1;
  A->o_pinv = o; o += (2 * P * N + 1) / 2;
  A->NQ = NQ;
  A->PG = PG;
  A->n_img = n_img;
  A->npass = (N <= 64) ? (NQ + (64 / N) - 1) / (64 / N) : NQ * SEG;
  return (size_t)o * 8;
}

template <int W, int NA, bool IMG, bool GJS, bool JX, int BIG>
static void perm_launch_t(gdml_ctx* ctx, const PermArgs& A, dim3 grid, size_t lds) {
  (void)hipFuncSetAttribute((const void*)assemble_perm_kernel<W, NA, IMG, GJS, JX, BIG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  hipLaunchKernelGGL((assemble_perm_kernel<W, NA, IMG, GJS, JX, BIG>), grid, dim3(64 * W), lds, ctx->stream, A);
}

// Launch over the column points [0, n_j) of (jlist | j0 + v) and the row points [i_beg, i_end).
int assemble_perm_launch(gdml_ctx* ctx, double sig, int use_E, const int32_t* d_jlist, const int32_t* d_colmap, int64_t j0,
                         int64_t n_j, int64_t col0, double* K, int64_t ld, int64_t i_beg, int64_t i_end, int lower, double lam,
                         int cyc_W, int cyc_rank, int cyc_nb, const int32_t* h_colmap) {
  TrainSet& ts = ctx->ts;
  if (n_j <= 0 || i_end <= i_beg) return GDML_OK;
  // small molecules, whole column points, plain row layout: the producer / consumer kernel of assemble_pts.hip
  if (!d_jlist && !d_colmap && col0 == 0 && cyc_W == 0 && assemble_pts_applicable(ctx)) {
    const int rc = assemble_pts_launch(ctx, sig, use_E, j0, n_j, K, ld, i_beg, i_end, lower, lam);
    if (rc != GDML_ERR_UNSUPPORTED) return rc;  // no LDS layout for this (N, P): the general kernel below
  }
  GDML_TRY(build_dense_tables(ctx));
  const int N = ts.N, P = ts.P;
  if ((lower || cyc_W > 0) && (d_jlist || d_colmap || use_E || j0 != 0 || i_beg != 0 || n_j != ts.M || i_end != ts.M))
    return gdml_fail(ctx, GDML_ERR_INVALID, "assemble_perm: the lower form needs the dense full column range");
  if (cyc_W > 0 && 3 * N > cyc_nb)
    return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "assemble_perm: row-cyclic layout needs 3N <= %d", cyc_nb);
  // no permutation group, more than 21 atoms, dense column range, plain layout: the direct P = 1 kernel of assemble_big1.hip
  if (!d_jlist && !d_colmap && !use_E && cyc_W == 0 && assemble_big1_applicable(ctx))
    return assemble_big1_launch(ctx, sig, j0, n_j, col0, K, ld, i_beg, i_end, lower ? 1 : 0, lam);
  // 25 ... 42 atoms, plain layout: the MFMA / fixed-atom-split kernel of assemble_perm2.hip -- dense column ranges, and
  // (round 6) index lists that request WHOLE column points in list order (every column of each listed point, output column
  // 3N v + c: what the iterative solver's K_nm is, iterative.py:229-247 -- configs[3] spent 0.69 s per build on the general
  // kernel for it)
  bool whole_points = d_jlist != nullptr && d_colmap != nullptr && h_colmap != nullptr && col0 == 0 && !lower;
  if (whole_points)
    for (int64_t e = 0; e < n_j * 3 * N && whole_points; ++e) whole_points = h_colmap[e] == (int32_t)e;
  if (((!d_jlist && !d_colmap) || whole_points) && !use_E && cyc_W == 0 && assemble_perm2_applicable(ctx)) {
    const int rc = assemble_perm2_launch(ctx, sig, j0, n_j, col0, K, ld, i_beg, i_end, lower ? 1 : 0, lam, whole_points ? d_jlist : nullptr);
    if (rc != GDML_ERR_UNSUPPORTED) return rc;
  }
  PermArgs A;
  memset(&A, 0, sizeof(A));
  A.XF = ts.XF; A.GD = ts.GD; A.perm = ts.perm; A.pinv = ts.pinv;
  A.M = ts.M; A.N = N; A.P = P; A.sig = sig; A.use_E = use_E;
  A.e_row0 = (i_beg == 0 && i_end == ts.M) ? ts.M * 3 * (int64_t)N : (i_end - i_beg) * 3 * (int64_t)N - i_beg;
  A.jlist = d_jlist; A.colmap = d_colmap; A.j0 = j0; A.n_j = n_j; A.col0 = col0;
  A.i_beg = i_beg; A.i_end = i_end; A.lower = (lower || cyc_W > 0) ? 1 : 0; A.lam = lam;
  A.cyc_W = cyc_W; A.cyc_rank = cyc_rank; A.cyc_nb = cyc_nb;
  A.K = K; A.ld = ld;
  A.dbg = ctx_opt_i(ctx, "asm.perm_debug", 0);
  A.fast_store = (!d_colmap && cyc_W == 0 && (col0 % 16) == 0 && ctx_opt_i(ctx, "asm.perm_fast_store", 1)) ? 1 : 0;

  // ---- shape: wavefronts and row atoms per wavefront, what lives in LDS, permutations per group.  Registers allow two
  // wavefronts per SIMD: either one workgroup of 8 wavefronts per CU with the whole LDS, or (N <= 24) two independent
  // workgroups of 4 that do not share barriers, with 80 KB each.  Residency levels, in the order they pay (the O phase
  // reads G_i and G_j three times per row atom and permutation, the V phase everything once per permutation):
  //   0: nothing   1: row-point image   2: + the strip's G_j table   3: + the x_j tables
  // asm.perm_w (0 automatic) picks the workgroup shape, asm.perm_level (-1 automatic) caps the level, asm.perm_pg fixes
  // the group size, asm.perm_nimg the image buffers.
  const int opt_w = ctx_opt_i(ctx, "asm.perm_w", 0), opt_na = ctx_opt_i(ctx, "asm.perm_na", 0);
  int W = 8, NA = 6;
  if (N <= 24) {  // measured (profiles/r03_assemble_perm_shapes.txt): 8 x 3 beats 4 x 6 (19.2 vs 21.9 ms at N = 21, P = 4, M = 1000)
    if (opt_w == 4) { W = 4; NA = 6; }
    else NA = (opt_na == 6) ? 6 : 3;
  }
  const bool img_ok = N <= W * NA;
  const size_t budget = (W == 4 && ctx_opt_i(ctx, "asm.perm_lds_kb", 80) <= 80) ? (size_t)80 * 1024 : (size_t)160 * 1024;
  const int opt_level = ctx_opt_i(ctx, "asm.perm_level", -1), opt_pg = ctx_opt_i(ctx, "asm.perm_pg", 0);
  const int opt_nimg = ctx_opt_i(ctx, "asm.perm_nimg", 0);
  int pg_max = P < 8 ? P : 8;
  if (opt_pg > 0) pg_max = opt_pg < P ? opt_pg : P;
  int level = -1, PG = 0, n_img = 0;
  PermArgs tmp;
  for (int lv = (img_ok ? 3 : 0); lv >= 0 && level < 0; --lv) {
    if (opt_level >= 0 && lv > opt_level) continue;
    // group size that must fit for this level to be taken: min(P, 4), else min(P, 2); level 0 takes whatever fits
    const int wants[3] = {pg_max < 4 ? pg_max : 4, pg_max < 2 ? pg_max : 2, 1};
    for (int wi = 0; wi < 3; ++wi) {
      const int want = (opt_pg > 0) ? pg_max : wants[wi];
      if (wi == 2 && lv > 0 && pg_max > 1 && opt_pg <= 0) break;  // rather a lower level than one-permutation groups
      if (perm_layout(N, P, W, NA, lv >= 1 ? 1 : 0, lv >= 2, lv >= 3, want, &tmp) <= budget) {
        level = lv;
        PG = want;
        break;
      }
    }
  }
  if (level < 0) return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "assemble_perm: no LDS layout for N=%d P=%d", N, P);
  n_img = level >= 1 ? 1 : 0;
  if (n_img == 1 && opt_nimg != 1 && perm_layout(N, P, W, NA, 2, level >= 2, level >= 3, PG, &tmp) <= budget) n_img = 2;
  while (PG < pg_max && perm_layout(N, P, W, NA, n_img, level >= 2, level >= 3, PG + 1, &tmp) <= budget) ++PG;
  PG = (P + (P + PG - 1) / PG - 1) / ((P + PG - 1) / PG);  // equal groups

  // ---- index-list columns: strips of REQUESTED column atoms.  The columns the iterative solver asks for are individual
  // matrix columns (iterative.py:372-379, :401-411), a few per training point, and a strip of 64 consecutive column atoms
  // computes all 3N columns of every point it touches -- a whole kernel matrix for one K_nm.  Here the atoms that own a
  // requested column are packed 64 to a strip (at most nq consecutive points per strip: the V phase works per column point);
  // taken when it at least halves the number of strips.
  std::vector<int32_t> h_ca, h_jv0;
  int nq_list = 0;
  const int nq_dense = (62 + N) / N + 1;
  if (h_colmap && d_colmap && !lower && cyc_W == 0 && ctx_opt_i(ctx, "asm.perm_compact", 1)) {
    for (int nq : {16, 12, 8, 6, 4}) {
      if (nq <= nq_dense) break;
      if (perm_layout(N, P, W, NA, n_img, level >= 2, level >= 3, PG, &tmp, nq) <= budget) {
        nq_list = nq;
        break;
      }
    }
    if (nq_list) {
      int used = 0, npts = 0;
      auto close = [&]() {
        while (used < 64) {
          h_ca.push_back(-1);
          ++used;
        }
        used = 0;
        npts = 0;
      };
      for (int64_t v = 0; v < n_j; ++v) {
        const int32_t* cm = h_colmap + v * 3 * N;
        bool first_of_point = true;
        for (int b = 0; b < N; ++b) {
          if (cm[3 * b] < 0 && cm[3 * b + 1] < 0 && cm[3 * b + 2] < 0) continue;
          if (used == 64 || (first_of_point && used > 0 && npts == nq_list)) close();
          if (used == 0) h_jv0.push_back((int32_t)v);
          if (first_of_point || used == 0) ++npts;
          first_of_point = false;
          h_ca.push_back((int32_t)(v * N + b));
          ++used;
        }
      }
      if (used > 0) close();
      const int64_t strips_dense = (n_j * N + 63) / 64;
      if ((int64_t)h_jv0.size() * 2 > strips_dense) {  // does not pay: the dense strips
        h_ca.clear();
        h_jv0.clear();
        nq_list = 0;
      }
    }
  }
  const size_t lds = perm_layout(N, P, W, NA, n_img, level >= 2, level >= 3, PG, &A, nq_list);
  int32_t *d_ca = nullptr, *d_jv0 = nullptr;
  if (nq_list) {
    // one allocation for both lists (freed below, after the launch)
    GDML_TRY(ctx_alloc(ctx, (void**)&d_ca, (int64_t)(h_ca.size() + h_jv0.size()) * 4));
    d_jv0 = d_ca + h_ca.size();
    hipError_t e = hipMemcpyAsync(d_ca, h_ca.data(), h_ca.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_jv0, h_jv0.data(), h_jv0.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the host vectors go out of scope with this call
    if (e != hipSuccess) {
      (void)ctx_free(ctx, d_ca);
      return gdml_fail(ctx, GDML_ERR_HIP, "assemble_perm: column-atom list upload: %s", hipGetErrorString(e));
    }
    A.calist = d_ca;
    A.strip_jv0 = d_jv0;
  }

  // ---- grid: strips x chunks of row points
  const int64_t n_strips = nq_list ? (int64_t)h_jv0.size() : (n_j * N + 63) / 64;
  const int64_t n_i = i_end - i_beg;
  int i_chunk = ctx_opt_i(ctx, "asm.perm_i_chunk", 16);
  if (i_chunk < 1) i_chunk = 1;
  while (i_chunk > 2 && n_strips * ((n_i + i_chunk - 1) / i_chunk) < 2048) i_chunk >>= 1;
  A.i_chunk = i_chunk;
  dim3 grid((unsigned)n_strips, (unsigned)((n_i + i_chunk - 1) / i_chunk));
  const int slot = ktime_begin(ctx);
#define PERM_GO(w, na)                                                                       \
  do {                                                                                       \
    if (level == 3) perm_launch_t<w, na, true, true, true, 0>(ctx, A, grid, lds);        \
    else if (level == 2) perm_launch_t<w, na, true, true, false, 0>(ctx, A, grid, lds);  \
    else if (level == 1) perm_launch_t<w, na, true, false, false, 0>(ctx, A, grid, lds); \
    else perm_launch_t<w, na, false, false, false, 0>(ctx, A, grid, lds);                \
  } while (0)
  // beyond 64 atoms the LDS-row variant (measured faster than two lane-held rows + select: N = 100 5.20 -> 4.63 ms, N = 128
  // 4.31 -> 3.88, N = 70 P = 4 10.59 -> 10.16: profiles/r04_large_molecules.txt); asm.perm_lds_rows = 0 keeps the lane-held
  // rows for 64 < N <= 128 as the A/B
  if (N > 128 || (N > 64 && ctx_opt_i(ctx, "asm.perm_lds_rows", 1))) perm_launch_t<8, 6, false, false, false, 2>(ctx, A, grid, lds);
  else if (N > 64) perm_launch_t<8, 6, false, false, false, 1>(ctx, A, grid, lds);
  else if (W == 4) PERM_GO(4, 6);
  else if (NA == 3) PERM_GO(8, 3);
  else PERM_GO(8, 6);
#undef PERM_GO
  const double blocks = A.lower ? 0.5 * (double)n_i * (double)(n_i + 1) : (double)n_i * (double)n_j;
  ktime_end(ctx, slot, "assemble", 8.0 * blocks * 9.0 * N * N);
  ctx->launch_counter++;
  HIP_CHECK(ctx, hipGetLastError());
  if (d_ca) GDML_TRY(ctx_free(ctx, d_ca));
  return GDML_OK;
}
