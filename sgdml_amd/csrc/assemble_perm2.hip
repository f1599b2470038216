// Kernel-matrix assembly for molecules with a permutation group, 25 ... 42 atoms, dense column ranges (round 5).
//
// Reference: sgdml/train.py:97-302 (_assemble_kernel_mat_wkr), torchtools.py:110-392.  Math as in assemble_perm.hip:
//   K_ij = sum_p [ beta_p v_p u_p^T + cn_p S_p ],   d_p = x_i - P_p x_j,   beta_p = 5 b_p,  cn_p = -(sig^2 + sig |d_p|') b_p
//   for a column atom b and its partner m' (b != m'),  a' = pi^-1 b,  mi = pi^-1 m',  d = x_i(a', mi) - x_j(b, m'):
//     |d_p|^2 += d^2,   u_p[b] += d G_j(b, m'),   v_p[a'] += d G_i(a', mi),   dg_p[b] += G_i(a', mi) (x) G_j(b, m')
//   S_p[(a,.),(b,.)] = G_i(a, pi^-1 b) (x) G_j(b, pi a)  for pi a != b (the tables are 0 on their diagonals),  dg_p[b] for a = pi^-1 b.
// assemble_perm_kernel runs this at 0.045 of HBM / 0.115 of the fp64 VALU model at N = 42, P = 27 (this kernel: 0.090 / 0.23, 2.0 x): 256-VGPR wavefronts, one
// workgroup per CU, every operand of the 9 N^2 P outer-product and the 9 N^2 P single-term multiply-adds read from LDS.  Here:
//   * atoms are renumbered, the ones NO permutation moves (F) first.  Descriptor entries between two such atoms contribute the
//     same to every permutation: they are summed once per block (base pass: u0, v0, dg0, nn0), a permutation adds only the
//     entries that touch a moved atom (configs[3]: 9 of 42 atoms move -> 0.38 of the entries);
//   * ONE fused pass per permutation over the pairs (b, m') gives |d|^2, u, v and dg; lane = (permutation of the group of 8,
//     slot), slot = (row b, chunk of the m' range): the loops are uniform, nothing is reduced across lanes except the chunks;
//   * the outer products sum_p beta_p v_p u_p^T run on v_mfma_f64_16x16x4.  Tile rows / columns are relabelled so that a lane
//     owns whole 3 x 3 atom blocks: a tile GROUP (s, t) = 16 row atoms x 16 column atoms, its 9 tiles = the (al, be)
//     components; lane l: column atom 16 t + (l & 15), row atoms 16 s + (l >> 4) + 4 r (C layout: row (l >> 4) + 4 r).  The
//     single terms are added to the same registers: per (row atom, column atom, permutation) two 24-byte LDS reads for 9
//     multiply-adds -- and only in tile groups that contain a moved atom; tile groups of fixed atoms get them once with sum_p cn_p;
//   * one workgroup = 9 wavefronts = the 3 x 3 tile groups of a (48 x 48)-atom block; it owns a column point j (its table
//     stays in LDS) and walks over row points i; finished rows are stored straight from the registers, 24 bytes per lane and
//     row (DIRECT; the variant that stages them through LDS for full-row stores is kept as the A/B: 6 % slower, seven more barriers).
// Everything else (index lists, energy-constraint rows, block-cyclic layouts, other sizes) stays on assemble_perm_kernel.
// Arithmetic pinned on the CPU: tools/perm2_emulate.py (run by the CPU test suite).
#include "common.h"
#include <algorithm>
#include <type_traits>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int P2_NW = 9;
constexpr int P2_T = 64 * P2_NW;
constexpr int P2_MAXN = 42;  // LDS: two N x N x 32-byte tables + 8 permutations of V-phase results
// LDS layout in doubles, fixed for the largest molecule (compile-time bases: no address registers held across the phases)
constexpr int L_TI = 0, L_TJ = L_TI + 4 * P2_MAXN * P2_MAXN, L_U = L_TJ + 4 * P2_MAXN * P2_MAXN, L_V = L_U + 3 * P2_MAXN * 8,
              L_DG = L_V + 3 * P2_MAXN * 8, L_B0 = L_DG + 9 * P2_MAXN * 8, L_NN = L_B0 + 16 * P2_MAXN, L_PERM = L_NN + 8 * P2_NW;

// v from the lane with index (lane ^ 1), (lane ^ 2), (lane ^ 4), (lane ^ 8): DPP moves inside a row of 16 lanes, no LDS round trip.
// (^ 4 as the mirror image inside 8 lanes: only for values that are already equal within each quad)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
#define P2_XOR1 0xB1       /* quad_perm [1,0,3,2] */
#define P2_XOR2 0x4E       /* quad_perm [2,3,0,1] */
#define P2_HALF_MIRROR 0x141
#define P2_ROR8 0x128      /* row_ror:8 = lane ^ 8 */

// exp out of line: inlined, its polynomial coefficients are hoisted out of the block loop into registers the kernel does not have
// (they came back from scratch one by one, a serialised round trip each, every group of every block)
__device__ __attribute__((noinline)) double p2_exp(double x) { return exp(x); }

// Barrier between phases that only exchange LDS data: wait for this wavefront's LDS operations, not for its global stores
// (__syncthreads() carries a vmcnt(0): in the row passes every barrier waited for the rows just stored to reach memory)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Perm2Args {
  const double* TP;     // (M,N,N,4) packed per-point tables in internal numbering: [a][m] = (x(a,m), G(a,m))
  const uint8_t* blob;  // plan: perm | pinv (internal numbering, bytes), src offsets, sigma, tasks
  int o_src, o_sigma, o_tasks;
  int n_tasks;
  int64_t M;
  int N, P, nF;
  double sig;
  int64_t j0, n_j, col0, i_beg, i_end;
  const int32_t* jlist;  // column POINTS of an index list that requests whole points (K_nm of the iterative solver), or null: j0 + v
  int i_chunk;
  int lower;
  double lam;
  double* K;
  int64_t ld;
  int l_task;  // LDS offset of the task descriptors in doubles (behind the byte permutation tables)
  int l_cn;    // ... of cn_p of all permutations (post mode)
  int l_sigma, l_ed, l_pt;  // ... of the row map, of the per-group table of moved-atom diagonal terms, of its pair tables (bytes)
  int nFb;                  // fixed rows whose fixed-partner part comes from the base pass (the others ride in full-range tasks)
  int es;                   // es mode (needs post): single terms of moved x moved blocks summed over the permutations by four lanes per block, once per (i, j)
  int ed, npairs, o_pt;     // ed mode (needs post): the diagonal terms of moved atoms summed per (row atom, column atom) pair by idle lanes
  int post, nE;  // single / diagonal terms of fixed atoms once per block (nE = moved atoms <= 16)
  unsigned long long* trace;  // asm.perm2_debug & 1024: shader-clock stamps of one workgroup's phases (tools/perm2_check.py trace)
  int dbg;  // timing-only ablation: 1 no stores, 2 no V tasks, 4 no single terms, 8 no MFMA, 16 no base pass, 32 no exp, 64 no diagonal terms,
            // 128 no staging of the rows, 256 no image prefetch
};

// LIST: column points come from A.jlist (a separate instantiation: the dense kernel's register allocation is untouched --
// tests/test_isa_schedule.py pins its scratch budget)
template <bool TRACE, bool DIRECT, bool LIST = false>
__global__ void __launch_bounds__(P2_T) assemble_perm2_kernel(Perm2Args A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int N = A.N, N3 = 3 * N, NN = N * N, P = A.P, nF = A.nF;
  double* const TI = smem + L_TI;  // [a][m] = (x_i(a,m), G_i(a,m))  internal numbering
  double* const TJ = smem + L_TJ;
  double* const U = smem + L_U;    // [3 b + be][8]
  double* const V = smem + L_V;    // [3 a + al][8]
  double* const DG = smem + L_DG;  // [9 b + k][8]
  double* const ST = U;            // finished rows (aliases U | V | DG): [36][3N]
  double* const B0 = smem + L_B0;  // [b][16]: u0 (3), v0 (3), dg0 (9), row part of nn0
  double* const NNS = smem + L_NN; // [wavefront][8]
  uint8_t* const permS = reinterpret_cast<uint8_t*>(smem + L_PERM);
  uint8_t* const pinvS = permS + P * N;
  double* const CN = smem + A.l_cn;
  double* const ED = smem + A.l_ed;  // [pair][9]
  uint8_t* const pmapS = reinterpret_cast<uint8_t*>(smem + A.l_pt);  // [nE][nE] pair index or 255, then [pair][2] = (row atom, column atom)
  uint8_t* const plistS = pmapS + A.nE * A.nE;
  uint32_t* const taskS = reinterpret_cast<uint32_t*>(smem + A.l_task);
  const int32_t* const sigma_g = reinterpret_cast<const int32_t*>(A.blob + A.o_sigma);
  int* const sigma = reinterpret_cast<int*>(smem + A.l_sigma);  // LDS copy: a global load between two row stores would wait for the stores

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t jv = blockIdx.x;
  const int64_t jpt = LIST ? (int64_t)A.jlist[jv] : A.j0 + jv;
  const bool lower = A.lower != 0;
  const int64_t i_lo = (lower ? jv : A.i_beg) + (int64_t)blockIdx.y * A.i_chunk;
  const int64_t i_top = lower ? A.M : A.i_end;
  const int64_t i_hi = (i_lo + A.i_chunk < i_top) ? i_lo + A.i_chunk : i_top;
  if (i_lo >= i_hi) return;

  // phase stamps of one workgroup (wavefront 0, blocks 2 and 3 of its walk): id, clock
  const bool tracing = TRACE && A.trace != nullptr && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && tid == 0;
  int trace_n = 0;
  auto stamp = [&](int id, int64_t blk) {
    if (TRACE && tracing && blk - i_lo >= 2 && blk - i_lo < 4 && trace_n < 250) {
      A.trace[2 * trace_n] = (unsigned long long)id;
      A.trace[2 * trace_n + 1] = __builtin_amdgcn_s_memtime();
      ++trace_n;
    }
  };
  const double sig = A.sig, inv_sig = 1.0 / sig;
  const double sqrt5 = 2.23606797749978969641;
  const double base_div = 5.0 / (3.0 * sig * sig * sig * sig);

  // ---- resident tables
  for (int e = tid; e < 2 * P * N; e += P2_T) permS[e] = A.blob[e];
  for (int e = tid; e < N; e += P2_T) sigma[e] = sigma_g[e];
  if (A.ed)
    for (int e = tid; e < A.nE * A.nE + 2 * A.npairs; e += P2_T) pmapS[e] = A.blob[A.o_pt + e];
  {
    const uint32_t* tg = reinterpret_cast<const uint32_t*>(A.blob + A.o_tasks);
    for (int e = tid; e < 4 * A.n_tasks + 4; e += P2_T) taskS[e] = tg[e];  // + 16 bytes: first task of every wavefront
  }
  for (int e = tid; e < 15 * P2_MAXN * 8; e += P2_T) U[e] = 0.0;  // U | V | DG: lanes of a short last group read finite values
  // a point's packed table (2 N^2 units of 16 bytes, already in internal order) straight into LDS: global_load_lds_dwordx4,
  // lane-linear destination, no registers, no wait here -- the next __syncthreads() drains it
  auto dma_table = [&](double* T, int64_t pt) {
    const char* srcb = reinterpret_cast<const char*>(A.TP + pt * NN * 4);
    char* dstb = reinterpret_cast<char*>(T);
    for (int u0 = w * 64; u0 < 2 * NN; u0 += P2_T) {
      const int u = u0 + lane;
      if (u < 2 * NN)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcb + (int64_t)u * 16),
                                         (__attribute__((address_space(3))) void*)(dstb + u0 * 16), 16, 0, 0);
    }
  };
  dma_table(TJ, jpt);
  dma_table(TI, i_lo);

  // this wavefront's V-phase tasks (stored wavefront by wavefront): [t_first, t_last), read once (scalar registers)
  const int t_first = __builtin_amdgcn_readfirstlane((int)A.blob[A.o_tasks + 16 * A.n_tasks + w]);
  const int t_last = __builtin_amdgcn_readfirstlane((int)A.blob[A.o_tasks + 16 * A.n_tasks + w + 1]);
  // ---- the tile group (gs, gt) of this wavefront in the O phase.  What a lane derives from its index (column atom, row atoms)
  // is recomputed inside each phase from an opaque copy of the index: kept alive across the phases these values (and the
  // lane masks of their validity flags) push loop invariants of the inner loops out to scratch.
  const int gs = w / 3, gt = w - 3 * gs;
  const bool group_live = 16 * gs < N && 16 * gt < N;
  const bool pureF = 16 * gs + 15 < nF && 16 * gt + 15 < nF;  // fixed atoms only: single terms once, with sum_p cn_p
// a fresh, opaque copy of the lane / thread index: what a phase derives from it (LDS addresses above all) is then not hoisted out
// of the group / block loops into registers that live across every other phase (they ended up in scratch, and a scratch reload
// between two global stores waits for the stores)
#define P2_FRESH(name, expr) \
  int name = (expr);         \
  asm volatile("" : "+v"(name))
#define P2_LANE_CONSTS()                                                     \
  int lane_o = lane;                                                         \
  asm volatile("" : "+v"(lane_o));                                           \
  const int g = lane_o >> 4, n = lane_o & 15;                                \
  const int cb = 16 * gt + n;                                                \
  const bool cb_ok = cb < N;                                                 \
  const int cbc = cb_ok ? cb : N - 1;                                        \
  int arc[4];                                                                \
  _Pragma("unroll") for (int r = 0; r < 4; ++r) {                            \
    const int a_ = 16 * gs + g + 4 * r;                                      \
    arc[r] = (a_ < N) ? a_ : N - 1;                                          \
  }

  // ---- single terms cn_p G_i(a, pi^-1 b) (x) G_j(b, pi a) of this lane's four atom blocks for n_its permutations from gfirst
  // (cnvec: lane pl holds cn of permutation gfirst + pl).  The byte lookups of permutation pl + 1 are requested before the
  // multiply-adds of permutation pl.  Post mode: only blocks of two MOVED atoms (weight 0 elsewhere).
  const bool ee_wave = group_live && 16 * gs + 15 >= nF && 16 * gt + 15 >= nF;  // tile group with moved rows and moved columns
  auto run_singles = [&](d4 (&acc)[3][3], const int (&arc)[4], int cbc, int gfirst, int n_its, int npg_g, double cnvec) {
    if (A.dbg & 4) n_its = 0;
    double wr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wr[r] = (!A.post || (arc[r] >= nF && cbc >= nF)) ? 1.0 : 0.0;
    const uint8_t* pr = permS + gfirst * N;
    const uint8_t* pin = pinvS + gfirst * N;
    int ap_n = pin[cbc], pa_n0 = pr[arc[0]], pa_n1 = pr[arc[1]], pa_n2 = pr[arc[2]], pa_n3 = pr[arc[3]];
#pragma unroll 1
    for (int pl = 0; pl < n_its; ++pl) {
      const int ap = ap_n;
      const int pa[4] = {pa_n0, pa_n1, pa_n2, pa_n3};
      const int adv = (pl + 1 < npg_g) ? N : 0;
      pr += adv;
      pin += adv;
      ap_n = pin[cbc];
      pa_n0 = pr[arc[0]]; pa_n1 = pr[arc[1]]; pa_n2 = pr[arc[2]]; pa_n3 = pr[arc[3]];
      double cn;
      {
        const long long bb = __builtin_bit_cast(long long, cnvec);
        const int lo = __builtin_amdgcn_readlane((int)bb, pl), hi = __builtin_amdgcn_readlane((int)(bb >> 32), pl);
        cn = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
      }
      // the operands of all four row atoms in one batch of reads (one LDS round trip per permutation)
      double a0[4], q0[4];
      d2 a12[4], q12[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double* gi = TI + (arc[r] * N + ap) * 4;
        const double* gj = TJ + (cbc * N + pa[r]) * 4;
        a0[r] = gi[1];
        a12[r] = *reinterpret_cast<const d2*>(gi + 2);
        q0[r] = gj[1];
        q12[r] = *reinterpret_cast<const d2*>(gj + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double cnr = cn * wr[r];
        const double w0 = cnr * q0[r], w1 = cnr * q12[r].x, w2 = cnr * q12[r].y;
        acc[0][0][r] += a0[r] * w0; acc[0][1][r] += a0[r] * w1; acc[0][2][r] += a0[r] * w2;
        acc[1][0][r] += a12[r].x * w0; acc[1][1][r] += a12[r].x * w1; acc[1][2][r] += a12[r].x * w2;
        acc[2][0][r] += a12[r].y * w0; acc[2][1][r] += a12[r].y * w1; acc[2][2][r] += a12[r].y * w2;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- ed mode: the diagonal terms of moved atoms of one group, summed per (row atom, column atom) pair into ED by idle lanes
  // (end of the O phase), are added by the lane that holds the pair's block -- one batch of reads per group instead of nine
  // reads and 36 multiply-adds per permutation on the critical wavefront.  n_its = 0 or 1 (no branch around accumulator code).
  auto add_ed = [&](d4 (&acc)[3][3], const int (&arc)[4], int cbc, int n_its) {
    for (int it = 0; it < n_its; ++it) {
      double mw[4];
      int pidx[4];
      const int cbe = (cbc >= nF) ? cbc - nF : 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ae = (arc[r] >= nF) ? arc[r] - nF : 0;
        pidx[r] = pmapS[ae * A.nE + cbe];
      }
      double x[4][9];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 9; ++k) x[r][k] = ED[(pidx[r] < A.npairs ? pidx[r] : 0) * 9 + k];
#pragma unroll
      for (int r = 0; r < 4; ++r) mw[r] = ((arc[r] >= nF) & (cbc >= nF) & (pidx[r] < A.npairs)) ? 1.0 : 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k / 3][k % 3][r] += mw[r] * x[r][k];
    }
  };

  for (int64_t i = i_lo; i < i_hi; ++i) {
    __syncthreads();  // image of point i (and the resident tables) visible; the previous block's staging is read
    stamp(0, i);
    // ================= base pass: pairs of fixed atoms, once per block.  lane = (row of the task, m' mod 8)
    double nn0 = 0.0;
    if (nF > 0 && !(A.dbg & 16)) {
      P2_FRESH(lane_b, threadIdx.x & 63);
      const int r8 = lane_b >> 3, c = lane_b & 7;
      for (int t0 = w; 8 * t0 < nF; t0 += P2_NW) {
        const int b = 8 * t0 + r8;
        const bool ok = b < nF;
        const int bc = ok ? b : 0;
        const double* ti = TI + bc * N * 4;
        const double* tj = TJ + bc * N * 4;
        double s[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) s[k] = 0.0;
        for (int m0 = 0; m0 < nF; m0 += 24) {  // three entries (six table reads) per trip: an LDS round trip per entry otherwise
          d4 av[3], qv[3];
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int m = m0 + 8 * t + c;
            const int mc = (m < nF) ? m : nF - 1;
            av[t] = *reinterpret_cast<const d4*>(ti + 4 * mc);
            qv[t] = *reinterpret_cast<const d4*>(tj + 4 * mc);
          }
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int m = m0 + 8 * t + c;
            const double lv = (ok && m < nF) ? 1.0 : 0.0;
            const d4 a = av[t], q = qv[t];
            const double d = (a.x - q.x) * lv;
            const double g0 = a.y * lv, g1 = a.z * lv, g2 = a.w * lv;
            s[15] += d * d;
            s[0] += d * q.y; s[1] += d * q.z; s[2] += d * q.w;
            s[3] += d * a.y; s[4] += d * a.z; s[5] += d * a.w;
            s[6] += g0 * q.y; s[7] += g0 * q.z; s[8] += g0 * q.w;
            s[9] += g1 * q.y; s[10] += g1 * q.z; s[11] += g1 * q.w;
            s[12] += g2 * q.y; s[13] += g2 * q.z; s[14] += g2 * q.w;
          }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) s[k] += dpp_f64<P2_XOR1>(s[k]);
#pragma unroll
        for (int k = 0; k < 16; ++k) s[k] += dpp_f64<P2_XOR2>(s[k]);
#pragma unroll
        for (int k = 0; k < 16; ++k) s[k] += dpp_f64<P2_HALF_MIRROR>(s[k]);
        if (ok && c == 0) {
#pragma unroll
          for (int k = 0; k < 16; ++k) B0[b * 16 + k] = s[k];
        }
      }
      __syncthreads();
      nn0 = wave_sum(lane_b < A.nFb ? B0[lane_b * 16 + 15] : 0.0);  // (fixed rows that ride in a full-range task count their pairs there)
    }
    stamp(1, i);

    d4 acc[3][3];
#pragma unroll
    for (int al = 0; al < 3; ++al)
#pragma unroll
      for (int be = 0; be < 3; ++be) acc[al][be] = d4{0.0, 0.0, 0.0, 0.0};
    double ctot = 0.0, cn_prev = 0.0;

    for (int g0 = 0; g0 < P; g0 += 8) {
      const int npg = (P - g0 < 8) ? P - g0 : 8;
      {  // post mode: the single terms of the PREVIOUS group on the wavefront with moved rows and moved columns (zero
         // iterations on the others), while the rest of the workgroup is in this group's V phase
        P2_LANE_CONSTS();
        const bool go = A.post && ee_wave && g0 > 0;
        const bool go_s = go && !A.es;  // es mode: the moved x moved single terms come out of the once-per-block pass too
        run_singles(acc, arc, cbc, go_s ? g0 - 8 : 0, go_s ? 8 : 0, 8, cn_prev);
        add_ed(acc, arc, cbc, (go && A.ed) ? 1 : 0);
      }
      // ================= phase V: lane = (slot, permutation of the group)
      {
        P2_FRESH(lane_v, threadIdx.x & 63);
        const int pl = lane_v & 7, slot = lane_v >> 3;
        const int p = g0 + (pl < npg ? pl : npg - 1);
        const uint8_t* const pin = pinvS + p * N;
        double nnl = 0.0;
        const int t_end = (A.dbg & 2) ? 0 : t_last;
        for (int t = t_first; t < t_end; ++t) {
          const uint32_t d0 = __builtin_amdgcn_readfirstlane(taskS[4 * t]), d1 = __builtin_amdgcn_readfirstlane(taskS[4 * t + 1]),
                         d2w = __builtin_amdgcn_readfirstlane(taskS[4 * t + 2]);  // wave-uniform: scalar registers
          const int mb = d2w & 0xff, me = (d2w >> 8) & 0xff, lg = (d2w >> 16) & 0xff;
          const bool base = (d2w >> 24) != 0;
          const int ri = slot >> lg, c = slot & ((1 << lg) - 1);
          const unsigned long long rows = (unsigned long long)d0 | ((unsigned long long)d1 << 32);
          const int rowb = (int)((rows >> (8 * ri)) & 0xff);
          const bool act = rowb != 255 && pl < npg;
          const int b = (rowb != 255) ? rowb : 0;
          const int per = (me - mb + (1 << lg) - 1) >> lg;
          const int m0 = mb + c * per;
          const int m1 = (m0 + per < me) ? m0 + per : me;
          const int ap = pin[b];
          const double* ti = TI + ap * N * 4;
          const double* tj = TJ + b * N * 4;
          double nn = 0.0, u0 = 0.0, u1 = 0.0, u2 = 0.0, v0 = 0.0, v1 = 0.0, v2 = 0.0;
          double d00 = 0.0, d01 = 0.0, d02 = 0.0, d10 = 0.0, d11 = 0.0, d12 = 0.0, d20 = 0.0, d21 = 0.0, d22 = 0.0;
          // two entries per trip: their four table reads are issued together, and the byte lookups of the NEXT two entries
          // are requested before the multiply-adds (volatile: a plain load carried around the loop is moved back to its use)
          typedef const volatile __attribute__((address_space(3))) uint8_t* lds_vbyte;
          const lds_vbyte pinv_v = (lds_vbyte)pin;
          auto mclamp = [&](int m) { return (m < N) ? m : N - 1; };
          // NB entries per trip: their 2 NB table reads are issued together, and the byte lookups of the NEXT NB entries are
          // requested before the multiply-adds.  Rows without diagonal-term products have the registers for three entries
          // per trip (an LDS round trip costs several hundred cycles here: fewer, fuller trips).
          auto vloop = [&](auto DGc) {
            constexpr bool WITH_DG = decltype(DGc)::value;
            constexpr int NB = WITH_DG ? 2 : 3;
            int mi[NB];
#pragma unroll
            for (int t = 0; t < NB; ++t) mi[t] = pinv_v[mclamp(m0 + t)];
#pragma unroll 1
            for (int k = 0; k < per; k += NB) {
              d4 av[NB], qv[NB];
#pragma unroll
              for (int t = 0; t < NB; ++t) {
                av[t] = *reinterpret_cast<const d4*>(ti + 4 * mi[t]);
                qv[t] = *reinterpret_cast<const d4*>(tj + 4 * mclamp(m0 + k + t));
              }
#pragma unroll
              for (int t = 0; t < NB; ++t) mi[t] = pinv_v[mclamp(m0 + k + NB + t)];
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int t = 0; t < NB; ++t) {
                const d4 a = av[t], q = qv[t];
                const double lv = (act && m0 + k + t < m1) ? 1.0 : 0.0;  // multiplied, not branched on: the loop stays uniform
                const double d = (a.x - q.x) * lv;
                nn += d * d;
                u0 += d * q.y; u1 += d * q.z; u2 += d * q.w;
                v0 += d * a.y; v1 += d * a.z; v2 += d * a.w;
                if (WITH_DG) {
                  const double g0v = a.y * lv, g1v = a.z * lv, g2v = a.w * lv;
                  d00 += g0v * q.y; d01 += g0v * q.z; d02 += g0v * q.w;
                  d10 += g1v * q.y; d11 += g1v * q.z; d12 += g1v * q.w;
                  d20 += g2v * q.y; d21 += g2v * q.z; d22 += g2v * q.w;
                }
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          };
          // post mode: the diagonal terms of fixed rows come out of the once-per-block pass
          if (base && A.post) vloop(std::false_type{});
          else vloop(std::true_type{});
          nnl += nn;
          if (lg > 0) {  // the chunks of a row sit in neighbouring slots: lane ^ 8 inside a row of 16 lanes
            u0 += dpp_f64<P2_ROR8>(u0); u1 += dpp_f64<P2_ROR8>(u1); u2 += dpp_f64<P2_ROR8>(u2);
            v0 += dpp_f64<P2_ROR8>(v0); v1 += dpp_f64<P2_ROR8>(v1); v2 += dpp_f64<P2_ROR8>(v2);
            d00 += dpp_f64<P2_ROR8>(d00); d01 += dpp_f64<P2_ROR8>(d01); d02 += dpp_f64<P2_ROR8>(d02);
            d10 += dpp_f64<P2_ROR8>(d10); d11 += dpp_f64<P2_ROR8>(d11); d12 += dpp_f64<P2_ROR8>(d12);
            d20 += dpp_f64<P2_ROR8>(d20); d21 += dpp_f64<P2_ROR8>(d21); d22 += dpp_f64<P2_ROR8>(d22);
          }
          for (int s = 1; s < lg; ++s) {
            const int off = 8 << s;
            u0 += __shfl_xor(u0, off, 64); u1 += __shfl_xor(u1, off, 64); u2 += __shfl_xor(u2, off, 64);
            v0 += __shfl_xor(v0, off, 64); v1 += __shfl_xor(v1, off, 64); v2 += __shfl_xor(v2, off, 64);
            d00 += __shfl_xor(d00, off, 64); d01 += __shfl_xor(d01, off, 64); d02 += __shfl_xor(d02, off, 64);
            d10 += __shfl_xor(d10, off, 64); d11 += __shfl_xor(d11, off, 64); d12 += __shfl_xor(d12, off, 64);
            d20 += __shfl_xor(d20, off, 64); d21 += __shfl_xor(d21, off, 64); d22 += __shfl_xor(d22, off, 64);
          }
          if (act && c == 0) {
            if (base) {  // fixed row: the permutation-independent part
              const double* b0 = B0 + b * 16;
              u0 += b0[0]; u1 += b0[1]; u2 += b0[2];
              v0 += b0[3]; v1 += b0[4]; v2 += b0[5];
              d00 += b0[6]; d01 += b0[7]; d02 += b0[8];
              d10 += b0[9]; d11 += b0[10]; d12 += b0[11];
              d20 += b0[12]; d21 += b0[13]; d22 += b0[14];
            }
            double* du = U + (3 * b) * 8 + pl;
            du[0] = u0; du[8] = u1; du[16] = u2;
            double* dv = V + (3 * ap) * 8 + pl;
            dv[0] = v0; dv[8] = v1; dv[16] = v2;
            double* dd = DG + (9 * b) * 8 + pl;
            dd[0] = d00; dd[8] = d01; dd[16] = d02;
            dd[24] = d10; dd[32] = d11; dd[40] = d12;
            dd[48] = d20; dd[56] = d21; dd[64] = d22;
          }
        }
        nnl += __shfl_xor(nnl, 8, 64);
        nnl += __shfl_xor(nnl, 16, 64);
        nnl += __shfl_xor(nnl, 32, 64);
        if (lane_v < 8) NNS[w * 8 + lane_v] = nnl;
      }
      stamp(2, i);
      if (TRACE && A.trace != nullptr && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && (tid & 63) == 0 && i - i_lo == 2)
        A.trace[600 + w * 4 + (g0 >> 3)] = __builtin_amdgcn_s_memtime();  // when every wavefront reaches the barrier after V
      __syncthreads();
      stamp(3, i);
      // ================= Matern scalars of the group (every wavefront for itself; lanes 0 .. 7 hold them)
      double beta_v = 0.0, cn_v = 0.0;
      {
        double nn = nn0;
        {
          double pn[P2_NW];
          P2_FRESH(lane_s, threadIdx.x & 63);
#pragma unroll
          for (int ww = 0; ww < P2_NW; ++ww) pn[ww] = NNS[ww * 8 + (lane_s & 7)];
#pragma unroll
          for (int ww = 0; ww < P2_NW; ++ww) nn += pn[ww];
        }
        const double nrm = sqrt5 * sqrt(0.5 * nn);
        const double ex = (A.dbg & 32) ? 1.0 : p2_exp(-nrm * inv_sig);
        const double bp = ex * base_div;
        if (lane < npg) {
          beta_v = 5.0 * bp;
          cn_v = -(sig * sig + sig * nrm) * bp;
        }
      }
      if (w == 0 && lane < npg) CN[g0 + lane] = cn_v;
      stamp(4, i);
      // ================= phase O.  (No branch around a piece of code that updates the accumulators: at every such join the
      // compiler keeps two copies of the 72 accumulator registers alive and spills; the optional parts are loops whose trip
      // count is zero instead.)
      {
        P2_LANE_CONSTS();
        const int arow = 16 * gs + n;  // row atom of the A operand
        const bool arow_ok = arow < N;
        const int arowc = arow_ok ? arow : N - 1;
        auto cn_of = [&](int pl) -> double {  // lane pl's value, wave-uniform: through scalar registers
          const long long bb = __builtin_bit_cast(long long, cn_v);
          const int lo = __builtin_amdgcn_readlane((int)bb, pl), hi = __builtin_amdgcn_readlane((int)(bb >> 32), pl);
          return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
        };
        for (int pl = 0; pl < npg; ++pl) ctot += cn_of(pl);
        // ---- single terms of this group (post mode: none here -- the one wavefront that has any runs them while the others
        // are in the NEXT group's V phase, see the top of the group loop)
        run_singles(acc, arc, cbc, g0, A.post ? 0 : ((pureF || !group_live) ? 0 : npg), npg, cn_v);
        stamp(5, i);
        // ---- the row atom pi^-1 b of this lane's column atom gets cn dg_p[b], if it is one of the lane's four.  The target
        // register of every permutation of the group (or none) is packed into one word first; the loop then has no LDS lookup
        // in front of its weights, its reads do not depend on anything, two permutations are in flight.  Branch-free inside
        // (the weight is cn in the matching register, 0 in the others); wavefronts without any such lane in the whole group
        // (off-diagonal tile groups of fixed atoms) run zero iterations.
        {
          unsigned tgt = 0;  // 4 bits per permutation: 1 << r of the matching register
          int apv[8];
#pragma unroll
          for (int pl = 0; pl < 8; ++pl) apv[pl] = pinvS[(g0 + (pl < npg ? pl : npg - 1)) * N + cbc];  // eight lookups in flight together
#pragma unroll
          for (int pl = 0; pl < 8; ++pl) {  // (bitwise, not short-circuit: no branch around a lookup)
            const int rr = apv[pl] - 16 * gs - g;
            const bool m = (pl < npg) & cb_ok & ((A.post == 0) | (cb >= nF)) & (rr >= 0) & (rr < 16) & ((rr & 3) == 0);
            tgt |= m ? (1u << (4 * pl + ((rr >> 2) & 3))) : 0u;
          }
          const int n_dg = (__builtin_amdgcn_ballot_w64(tgt != 0) != 0 && !(A.dbg & 64) && !A.ed) ? npg : 0;
          const double* dgb = DG + (9 * cbc) * 8;
#pragma unroll 2
          for (int pl = 0; pl < n_dg; ++pl) {
            double x[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) x[k] = dgb[8 * k + pl];
            const double cn = cn_of(pl);
            const unsigned t4 = tgt >> (4 * pl);
            double mw[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) mw[r] = ((t4 >> r) & 1u) ? cn : 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[k / 3][k % 3][r] += mw[r] * x[k];
          }
        }
        stamp(6, i);
        // ---- the outer products last: the MFMAs are issued and the wavefront goes on to the barrier and the next V phase
        // (which does not touch the accumulators) while they run
        const int nks = (A.dbg & 8) ? 0 : ((npg > 4) ? 2 : 1);
        for (int ks = 0; ks < nks; ++ks) {
          const int plk = 4 * ks + g;
          const double bk = __shfl(beta_v, plk, 64);
          double av[3], bv[3];
#pragma unroll
          for (int al = 0; al < 3; ++al) {
            const double x = V[(3 * arowc + al) * 8 + plk];
            av[al] = arow_ok ? bk * x : 0.0;
          }
#pragma unroll
          for (int be = 0; be < 3; ++be) {
            const double x = U[(3 * cbc + be) * 8 + plk];
            bv[be] = cb_ok ? x : 0.0;
          }
#pragma unroll
          for (int al = 0; al < 3; ++al)
#pragma unroll
            for (int be = 0; be < 3; ++be) acc[al][be] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[al], bv[be], acc[al][be], 0, 0, 0);
        }
      }
      if (A.ed && !(A.dbg & 64) && 64 * w < 9 * A.npairs) {  // (no accumulator code in here; wavefronts beyond the last pair skip it)
        P2_FRESH(tid_e, threadIdx.x);
        const int pair = tid_e / 9, k = tid_e - 9 * pair;
        const bool okp = pair < A.npairs;
        const int pc = okp ? pair : 0;
        const int pa_ = plistS[2 * pc], pb_ = plistS[2 * pc + 1];  // row atom, column atom (internal numbers)
        int by[8];
        d2 dv[4];
#pragma unroll
        for (int pl = 0; pl < 8; ++pl) by[pl] = pinvS[(g0 + (pl < npg ? pl : npg - 1)) * N + pb_];
#pragma unroll
        for (int h = 0; h < 4; ++h) dv[h] = *reinterpret_cast<const d2*>(DG + (9 * pb_ + k) * 8 + 2 * h);
        double sum = 0.0;
#pragma unroll
        for (int pl = 0; pl < 8; ++pl) {
          const long long bb = __builtin_bit_cast(long long, cn_v);
          const int lo = __builtin_amdgcn_readlane((int)bb, pl), hi = __builtin_amdgcn_readlane((int)(bb >> 32), pl);
          const double cn = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);  // 0 beyond the group
          const double x = (pl & 1) ? dv[pl >> 1].y : dv[pl >> 1].x;
          sum += (by[pl] == pa_) ? cn * x : 0.0;
        }
        if (okp) ED[pair * 9 + k] = sum;
      }
      cn_prev = cn_v;
      stamp(7, i);
      __syncthreads();  // the V-phase results of this group are free
      stamp(8, i);
    }
    {  // post mode: single terms of the last group
      P2_LANE_CONSTS();
      const int g_last = ((P - 1) / 8) * 8;
      run_singles(acc, arc, cbc, g_last, (A.post && ee_wave && !A.es) ? P - g_last : 0, P - g_last, cn_prev);
      add_ed(acc, arc, cbc, (A.post && A.ed && ee_wave) ? 1 : 0);
    }
    // ================= once per block (post mode): everything a fixed atom is involved in, summed over the permutations first.
    //   A1(a, e) = sum_p cn_p G_i(a, pi_p^-1 e),  B1(a, e) = sum_p cn_p G_j(a, pi_p e)      a fixed, e moved
    //   single terms:   fixed x fixed  ctot G_i(a,b) (x) G_j(b,a);   fixed x moved  A1(a,b) (x) G_j(b,a);   moved x fixed  G_i(a,b) (x) B1(b,a)
    //   diagonal term of a fixed atom:  ctot dg0[a] + sum_e A1(a, e) (x) G_j(a, e)
    // 16 lanes per fixed atom (one per moved atom), 36 atoms per round; tables in the (now free) V-phase buffers, entries
    // laid out like the table entries (pad, 3 values) so that the consumers read either with the same instructions.
    double* const A1S = U;
    double* const B1S = A1S + nF * A.nE * 4;
    double* const FDS = B1S + nF * A.nE * 4;
    if (A.post) {
      P2_FRESH(tid_p, threadIdx.x);
      const int ei = tid_p & 15, rs = tid_p >> 4, nE = A.nE;
      // W[x][y] = sum of cn_p over the permutations with pi_p x = y (x, y moved): A1(a, e) = sum_e' G_i(a, e') W[e'][e],
      // B1(a, e) = sum_e' G_j(a, e') W[e][e'] -- nE uniform steps without byte lookups instead of P gathers per lane
      double* const WS = FDS + ((9 * nF + 1) & ~1);
      for (int t = tid_p; t < nE * nE; t += P2_T) {
        const int x = nF + t / nE, y = nF + t - (t / nE) * nE;
        double wv = 0.0;
        for (int p0 = 0; p0 < P; p0 += 9) {  // nine lookups and nine cn_p in flight together
          int by[9];
          double cv[9];
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            const int p = (p0 + k < P) ? p0 + k : P - 1;
            by[k] = permS[p * N + x];
            cv[k] = CN[p];
          }
#pragma unroll
          for (int k = 0; k < 9; ++k) wv += ((p0 + k < P) & (by[k] == y)) ? cv[k] : 0.0;
        }
        WS[t] = wv;
      }
      // es mode: the single terms of the nE x nE blocks of two moved atoms, four lanes per block (permutations q, q + 4, ...),
      // two permutations per batch of reads; summed inside the quad and left in ES for the lane that holds the block
      double* const ES = WS + ((nE * nE + 1) & ~1);
      if (A.es) {
        for (int t0 = 0; t0 < 4 * nE * nE; t0 += P2_T) {
          const int t = t0 + ((tid_p >= 192) ? tid_p - 192 : tid_p + P2_T - 192);  // from wavefront 3 on: 0 and 1 build W meanwhile
          const bool okt = t < 4 * nE * nE;
          const int pair = okt ? (t >> 2) : 0, q = t & 3;
          const int a = nF + pair / nE, b = nF + pair - (pair / nE) * nE;
          const double* tia = TI + a * N * 4;
          const double* tjb = TJ + b * N * 4;
          double e[9];
#pragma unroll
          for (int k = 0; k < 9; ++k) e[k] = 0.0;
          for (int p0 = q; p0 < P; p0 += 8) {
            double cn[2], a0[2], q0[2];
            d2 a12[2], q12[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int p = (p0 + 4 * h < P) ? p0 + 4 * h : P - 1;
              cn[h] = (p0 + 4 * h < P) ? CN[p] : 0.0;
              const double* gi = tia + 4 * (int)pinvS[p * N + b];
              const double* gj = tjb + 4 * (int)permS[p * N + a];
              a0[h] = gi[1];
              a12[h] = *reinterpret_cast<const d2*>(gi + 2);
              q0[h] = gj[1];
              q12[h] = *reinterpret_cast<const d2*>(gj + 2);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const double w0 = cn[h] * q0[h], w1 = cn[h] * q12[h].x, w2 = cn[h] * q12[h].y;
              e[0] += a0[h] * w0; e[1] += a0[h] * w1; e[2] += a0[h] * w2;
              e[3] += a12[h].x * w0; e[4] += a12[h].x * w1; e[5] += a12[h].x * w2;
              e[6] += a12[h].y * w0; e[7] += a12[h].y * w1; e[8] += a12[h].y * w2;
            }
          }
#pragma unroll
          for (int k = 0; k < 9; ++k) e[k] += dpp_f64<P2_XOR1>(e[k]);
#pragma unroll
          for (int k = 0; k < 9; ++k) e[k] += dpp_f64<P2_XOR2>(e[k]);
          if (okt && q == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) ES[pair * 9 + k] = e[k];
          }
        }
      }
      __syncthreads();
      for (int ab = 0; ab < nF; ab += P2_T / 16) {
        const int a = ab + rs;
        const bool ok = a < nF && ei < nE;
        const int ac = (a < nF) ? a : 0;
        const int eic = (ei < nE) ? ei : 0;
        const int e = nF + eic;
        const double* tia = TI + (ac * N + nF) * 4;
        const double* tja = TJ + (ac * N + nF) * 4;
        double s10 = 0.0, s11 = 0.0, s12 = 0.0, s20 = 0.0, s21 = 0.0, s22 = 0.0;
#pragma unroll 3
        for (int e2 = 0; e2 < nE; ++e2) {
          const double w1 = WS[e2 * nE + eic], w2 = WS[eic * nE + e2];
          const double* gi = tia + 4 * e2;
          const double* gj = tja + 4 * e2;
          const double a0 = gi[1];
          const d2 a12 = *reinterpret_cast<const d2*>(gi + 2);
          const double q0 = gj[1];
          const d2 q12 = *reinterpret_cast<const d2*>(gj + 2);
          s10 += w1 * a0; s11 += w1 * a12.x; s12 += w1 * a12.y;
          s20 += w2 * q0; s21 += w2 * q12.x; s22 += w2 * q12.y;
        }
        const double* tja0 = TJ + ac * N * 4;
        const double* gje = tja0 + 4 * e;
        const double okv = ok ? 1.0 : 0.0;
        const double e0 = gje[1] * okv;
        const d2 e12 = *reinterpret_cast<const d2*>(gje + 2);
        const double e1 = e12.x * okv, e2 = e12.y * okv;
        double fd[9] = {s10 * e0, s10 * e1, s10 * e2, s11 * e0, s11 * e1, s11 * e2, s12 * e0, s12 * e1, s12 * e2};
#pragma unroll
        for (int k = 0; k < 9; ++k) fd[k] += dpp_f64<P2_XOR1>(fd[k]);
#pragma unroll
        for (int k = 0; k < 9; ++k) fd[k] += dpp_f64<P2_XOR2>(fd[k]);
#pragma unroll
        for (int k = 0; k < 9; ++k) fd[k] += dpp_f64<P2_HALF_MIRROR>(fd[k]);
#pragma unroll
        for (int k = 0; k < 9; ++k) fd[k] += dpp_f64<P2_ROR8>(fd[k]);
        if (ok) {
          double* da = A1S + (a * nE + ei) * 4;
          da[1] = s10; da[2] = s11; da[3] = s12;
          double* db = B1S + (a * nE + ei) * 4;
          db[1] = s20; db[2] = s21; db[3] = s22;
        }
        if (a < nF && ei == 0) {
          double g0v[9];
#pragma unroll
          for (int k = 0; k < 9; ++k) g0v[k] = B0[a * 16 + 6 + k];  // (all nine in flight: a read-add-write per k is nine round trips)
#pragma unroll
          for (int k = 0; k < 9; ++k) FDS[a * 9 + k] = fd[k] + ctot * g0v[k];
        }
      }
      __syncthreads();
    }
    stamp(9, i);
    P2_LANE_CONSTS();
    {  // the once-per-block single terms of this lane's four atom blocks (no post mode: tile groups of fixed atoms only, ctot)
      const bool fb = cbc < nF;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = arc[r];
        const bool fa = a < nF;
        const double* gi = TI + (a * N + cbc) * 4;
        const double* gj = TJ + (cbc * N + a) * 4;
        double scale = pureF ? ctot : 0.0;
        if (A.post) {
          gi = (fa && !fb) ? A1S + (a * A.nE + (cbc - nF)) * 4 : gi;
          gj = (!fa && fb) ? B1S + (cbc * A.nE + (a - nF)) * 4 : gj;
          scale = (fa && fb) ? ctot : ((!fa && !fb) ? 0.0 : 1.0);
        }
        if (A.dbg & 4) scale = 0.0;
        const double a0 = gi[1];
        const d2 a12 = *reinterpret_cast<const d2*>(gi + 2);
        const double q0 = gj[1];
        const d2 q12 = *reinterpret_cast<const d2*>(gj + 2);
        const double w0 = scale * q0, w1 = scale * q12.x, w2 = scale * q12.y;
        acc[0][0][r] += a0 * w0; acc[0][1][r] += a0 * w1; acc[0][2][r] += a0 * w2;
        acc[1][0][r] += a12.x * w0; acc[1][1][r] += a12.x * w1; acc[1][2][r] += a12.x * w2;
        acc[2][0][r] += a12.y * w0; acc[2][1][r] += a12.y * w1; acc[2][2][r] += a12.y * w2;
      }
      {  // es mode: the summed single terms of this lane's moved x moved blocks (zero iterations elsewhere)
        const int n_es = (A.post && A.es && ee_wave) ? 1 : 0;
        const double* const ESr = FDS + ((9 * nF + 1) & ~1) + ((A.nE * A.nE + 1) & ~1);
        for (int it = 0; it < n_es; ++it) {
          double x[4][9], mwe[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool m = (arc[r] >= nF) & (cbc >= nF);
            const int pe = m ? (arc[r] - nF) * A.nE + (cbc - nF) : 0;
            mwe[r] = m ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) x[r][k] = ESr[pe * 9 + k];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k / 3][k % 3][r] += mwe[r] * x[r][k];
        }
      }
      // diagonal terms of fixed atoms (post mode): the lane that holds the block (b, b)
      double mw[4];
      bool anyd = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool m = A.post && cb_ok && fb && arc[r] == cbc && !(A.dbg & 64);
        mw[r] = m ? 1.0 : 0.0;
        anyd |= m;
      }
      const int n_fd = (__builtin_amdgcn_ballot_w64(anyd) != 0) ? 1 : 0;
      for (int it = 0; it < n_fd; ++it) {
        const double* fdp = FDS + (fb ? cbc : 0) * 9;
        double x[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) x[k] = fdp[k];
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[k / 3][k % 3][r] += mw[r] * x[k];
      }
    }
    stamp(10, i);
    if (A.post || DIRECT) __syncthreads();  // the tables and the image are read: staging / the next image may overwrite them
    stamp(11, i);

    const int sig_b = sigma[cbc];
    const int64_t i_next = (i + 1 < i_hi) ? i + 1 : i;
    if (DIRECT) {
      // ================= rows out, straight from the registers: a lane holds, for each of its four row atoms and three
      // components, three consecutive doubles of one row (columns 3 sigma(b) ...).  36 eight-byte stores per lane that the L2
      // merges into whole lines -- against four LDS staging passes with seven barriers for full-row stores: this kernel is nowhere
      // near the store bandwidth, the passes were 15 % of a block.
      if (!(A.dbg & 256)) dma_table(TI, i_next);
      int sa[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) sa[r] = sigma[arc[r]];
      if (!(A.dbg & 1)) {
        const int64_t cbase = A.col0 + jv * N3 + 3 * sig_b;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (group_live && cb_ok && 16 * gs + g + 4 * r < N) {
#pragma unroll
            for (int al = 0; al < 3; ++al) {
              const int64_t grow = i * N3 + 3 * sa[r] + al;
              double* dst = A.K + (grow - A.i_beg * N3) * A.ld + cbase;
              const int64_t dcol = grow - cbase;  // 0 .. 2 where the matrix diagonal crosses these three columns
#pragma unroll
              for (int be = 0; be < 3; ++be) {
                const double o = lower ? -acc[al][be][r] : acc[al][be][r];
                dst[be] = o + ((lower && dcol == be) ? A.lam : 0.0);
              }
            }
          }
        }
      }
      stamp(13, i);
      continue;
    }

    // ================= rows out: four passes (register r of every lane = 12 row atoms = 36 rows) through LDS
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r > 0) lds_barrier();  // the previous pass is read out of the staging buffer (its stores may still be on their way)
      if (group_live && cb_ok && 16 * gs + g + 4 * r < N && !(A.dbg & 128)) {
        double* dst = ST + ((gs * 4 + g) * 3) * N3 + 3 * sig_b;
#pragma unroll
        for (int al = 0; al < 3; ++al)
#pragma unroll
          for (int be = 0; be < 3; ++be) dst[al * N3 + be] = lower ? -acc[al][be][r] : acc[al][be][r];
      }
      lds_barrier();
      stamp(12, i);
      if (!(A.dbg & 1)) {
        // this wavefront's four rows of the pass: row map, then all eight values, then the stores -- one LDS round trip each
        // (3N <= 126: two stores per lane and row)
        int arow4[4], orow4[4], otmp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int rsl = 4 * w + k;  // 0 .. 35
          const int ra = rsl / 3;
          arow4[k] = 16 * (ra >> 2) + (ra & 3) + 4 * r;
          otmp[k] = sigma[arow4[k] < N ? arow4[k] : N - 1];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) orow4[k] = __builtin_amdgcn_readfirstlane(otmp[k]);  // wave-uniform: scalar row addresses
        P2_FRESH(lane_r, threadIdx.x & 63);
        const int c1 = lane_r + 64;
        const bool ok1 = c1 < N3;
        double o0[4], o1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int rsl = 4 * w + k;
          o0[k] = ST[rsl * N3 + lane_r];
          o1[k] = ST[rsl * N3 + (ok1 ? c1 : lane_r)];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int rsl = 4 * w + k;
          const int al = rsl - 3 * (rsl / 3);
          if (arow4[k] < N) {
            const int64_t grow = i * N3 + 3 * orow4[k] + al;
            double* dstg = A.K + (grow - A.i_beg * N3) * A.ld + A.col0 + jv * N3;
            const int64_t dcol = grow - (A.col0 + jv * N3);  // where the matrix diagonal crosses this row segment
            dstg[lane_r] = o0[k] + ((lower && lane_r == dcol) ? A.lam : 0.0);
            if (ok1) dstg[c1] = o1[k] + ((lower && c1 == dcol) ? A.lam : 0.0);
          }
        }
      }
      // every wavefront is past its last read of the current image since the barrier of pass 0; requested here, behind the
      // stores of two passes (in front of them its reads delayed the stores), it has passes 2 and 3 to land
      if (r == 1 && !(A.dbg & 256)) dma_table(TI, i_next);
      stamp(13, i);
    }
  }
}

int build_dense_tables(gdml_ctx* ctx);

// TP[pt][e] = (x, G) of the dense tables' entry src[e]: the per-point tables in internal numbering, one 32-byte entry per ordered pair
__global__ void __launch_bounds__(256) perm2_pack_kernel(const double* __restrict__ XF, const double* __restrict__ GD,
                                                         const int32_t* __restrict__ src, int64_t M, int NN, double* __restrict__ TP) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= M * NN) return;
  const int64_t pt = t / NN;
  const int e = (int)(t - pt * NN);
  const int64_t so = pt * NN + src[e];
  d4 v;
  v.x = XF[so];
  v.y = GD[so * 3];
  v.z = GD[so * 3 + 1];
  v.w = GD[so * 3 + 2];
  *reinterpret_cast<d4*>(TP + t * 4) = v;
}

bool assemble_perm2_applicable(const gdml_ctx* ctx) {
  const TrainSet& ts = ctx->ts;
  if (!ctx_opt_i(ctx, "asm.perm2", 1)) return false;
  // Measured (profiles/r05_assemble_perm2.txt): the per-block cost of this kernel is nearly independent of N and P (17 barriers
  // and a dozen short dependent LDS chains per block and group with one workgroup of 9 wavefronts per CU), so it only wins where
  // assemble_perm_kernel's N^2 P work is largest: N = 42, P = 27 1.8x; N = 36, P = 27 1.17x; N = 42, P = 6 1.15x; N = 30, P = 6 0.72x.
  const int min_n = ctx_opt_i(ctx, "asm.perm2_min_n", 36), min_p = ctx_opt_i(ctx, "asm.perm2_min_p", 6);
  if (ts.P < min_p || ts.N < min_n || ts.N > P2_MAXN) return false;
  return ts.P >= 16 || ts.N >= min_n + 4 || min_n < 36;  // small groups only pay on the largest molecules
}

// Plan of the group: internal numbering (fixed atoms first), permutations in it, V-phase tasks.  Built once per training set.
static int perm2_plan(gdml_ctx* ctx) {
  TrainSet& ts = ctx->ts;
  // the plan depends on three options; they are read at the point of use like every other one: a changed value rebuilds it
  const int key = 1 + ctx_opt_i(ctx, "asm.perm2_split", 1) + 2 * ctx_opt_i(ctx, "asm.perm2_post", 1) + 4 * ctx_opt_i(ctx, "asm.perm2_es", 1) +
                  8 * ctx_opt_i(ctx, "asm.perm2_chunk", 12);
  if (ts.p2 && ts.p2_TP && ts.p2_key == key) return GDML_OK;
  // (a plan is valid only with BOTH buffers and the packed tables in place: p2_key is set last, a failure on the way
  //  leaves no half-built plan behind for the next assembly to launch on)
  auto drop = [&]() {
    if (ts.p2) ctx_free(ctx, ts.p2);
    if (ts.p2_TP) ctx_free(ctx, ts.p2_TP);
    ts.p2 = nullptr;
    ts.p2_TP = nullptr;
    ts.p2_key = 0;
  };
  drop();
  const int N = ts.N, P = ts.P;
  std::vector<int> moved(N, 0);
  for (int p = 0; p < P; ++p)
    for (int a = 0; a < N; ++a)
      if (ts.h_perm[(size_t)p * N + a] != a) moved[a] = 1;
  std::vector<int32_t> sigma;
  for (int a = 0; a < N; ++a)
    if (!moved[a]) sigma.push_back(a);
  int nF = (int)sigma.size();
  if (nF < 2 || !ctx_opt_i(ctx, "asm.perm2_split", 1)) {  // no split: natural order
    nF = 0;
    sigma.clear();
    for (int a = 0; a < N; ++a) sigma.push_back(a);
  } else {
    for (int a = 0; a < N; ++a)
      if (moved[a]) sigma.push_back(a);
  }
  std::vector<int32_t> inv(N);
  for (int a = 0; a < N; ++a) inv[sigma[a]] = a;
  std::vector<uint8_t> blob((size_t)2 * P * N);
  for (int p = 0; p < P; ++p)
    for (int a = 0; a < N; ++a) {
      const int pa = inv[ts.h_perm[(size_t)p * N + sigma[a]]];
      blob[(size_t)p * N + a] = (uint8_t)pa;
      blob[(size_t)(P + p) * N + pa] = (uint8_t)a;
    }
  // tasks: rows of E (all partners), then rows of F (partners in E); chunks so that a lane walks <= ~12 entries
  struct Task { uint8_t rows[8]; uint8_t mb, me, lg, base; uint32_t pad; };
  static_assert(sizeof(Task) == 16, "task descriptor");
  std::vector<Task> tasks;
  auto add_rows = [&](int b0, int b1, int mb, int me, int base) {
    if (b1 <= b0 || me <= mb) return;
    int lg = 0;
    while (lg < 3 && ((me - mb + (1 << lg) - 1) >> lg) > ctx_opt_i(ctx, "asm.perm2_chunk", 12)) ++lg;
    const int rows_per = 8 >> lg;
    for (int b = b0; b < b1; b += rows_per) {
      Task t;
      memset(&t, 0, sizeof(t));
      for (int k = 0; k < 8; ++k) t.rows[k] = (k < rows_per && b + k < b1) ? (uint8_t)(b + k) : (uint8_t)255;
      t.mb = (uint8_t)mb; t.me = (uint8_t)me; t.lg = (uint8_t)lg; t.base = (uint8_t)base;
      tasks.push_back(t);
    }
  };
  // the fixed rows beyond a multiple of 8 would be a nearly empty task of their own: if the last full-range task has spare row
  // slots they ride there (computed like a moved row: every partner, per permutation -- the same sums)
  int n_extra = 0;
  {
    int lgE = 0;
    while (lgE < 3 && ((N + (1 << lgE) - 1) >> lgE) > ctx_opt_i(ctx, "asm.perm2_chunk", 12)) ++lgE;
    const int rows_per = 8 >> lgE, nE_ = N - nF;
    const int spare = (rows_per - nE_ % rows_per) % rows_per;
    if (nF > 0 && nF % 8 != 0 && nF % 8 <= spare) n_extra = nF % 8;
  }
  add_rows(nF - n_extra, N, 0, N, 0);
  add_rows(0, nF - n_extra, nF, N, 1);
  // tasks to wavefronts, longest first onto the least loaded one (cost ~ trips of two entries + the chunk reduction); in post
  // mode the wavefronts whose tile group has moved rows and moved columns start with the single terms of 8 permutations
  const bool post_plan = nF >= 2 && N - nF <= 16 && ctx_opt_i(ctx, "asm.perm2_post", 1);
  double load[P2_NW];
  for (int w = 0; w < P2_NW; ++w) {
    const int gs = w / 3, gt = w % 3;
    const bool live = 16 * gs < N && 16 * gt < N;
    // (only without es mode that wavefront still runs per-permutation single terms during the V phase)
    load[w] = (post_plan && !ctx_opt_i(ctx, "asm.perm2_es", 1) && live && 16 * gs + 15 >= nF && 16 * gt + 15 >= nF) ? 11.0 : 0.0;
  }
  auto cost = [&](const Task& t) { return 0.5 * (((t.me - t.mb + (1 << t.lg) - 1) >> t.lg) + 1) + 1.0 + 0.7 * t.lg; };
  std::stable_sort(tasks.begin(), tasks.end(), [&](const Task& a, const Task& b) { return cost(a) > cost(b); });
  std::vector<std::vector<Task>> per_wave(P2_NW);
  for (const Task& t : tasks) {
    int best = 0;
    for (int w = 1; w < P2_NW; ++w)
      if (load[w] < load[best]) best = w;
    per_wave[best].push_back(t);
    load[best] += cost(t);
  }
  uint8_t wt_off[16] = {0};
  tasks.clear();
  for (int w = 0; w < P2_NW; ++w) {
    wt_off[w] = (uint8_t)tasks.size();
    for (const Task& t : per_wave[w]) tasks.push_back(t);
  }
  for (int w = P2_NW; w < 16; ++w) wt_off[w] = (uint8_t)tasks.size();
  auto align_to = [&](size_t al) { while (blob.size() % al) blob.push_back(0); };
  align_to(16);
  ts.p2_o[0] = (int)blob.size();  // src
  {
    std::vector<int32_t> srcv((size_t)N * N);  // internal entry (a, m)  <-  dense entry (sigma m, sigma a) = G(sigma a, sigma m)
    for (int a = 0; a < N; ++a)
      for (int m = 0; m < N; ++m) srcv[(size_t)a * N + m] = sigma[m] * N + sigma[a];
    const uint8_t* b = reinterpret_cast<const uint8_t*>(srcv.data());
    blob.insert(blob.end(), b, b + srcv.size() * 4);
  }
  align_to(16);
  ts.p2_o[1] = (int)blob.size();  // sigma
  {
    const uint8_t* b = reinterpret_cast<const uint8_t*>(sigma.data());
    blob.insert(blob.end(), b, b + sigma.size() * 4);
  }
  align_to(16);
  ts.p2_o[2] = (int)blob.size();  // tasks
  {
    const uint8_t* b = reinterpret_cast<const uint8_t*>(tasks.data());
    blob.insert(blob.end(), b, b + tasks.size() * sizeof(Task));
    blob.insert(blob.end(), wt_off, wt_off + 16);
  }
  // pairs (row atom a, column atom b) of moved atoms with pi_p a = b for some p (a == b included): pair map + pair list
  ts.p2_o[3] = (int)blob.size();
  ts.p2_npairs = 0;
  if (nF > 0) {
    const int nE = N - nF;
    std::vector<uint8_t> pmap((size_t)nE * nE, 255), plist;
    const uint8_t* permI = blob.data();  // [p][a] internal numbering
    for (int a = nF; a < N; ++a)
      for (int b = nF; b < N; ++b) {
        bool hit = false;
        for (int p = 0; p < P && !hit; ++p) hit = permI[(size_t)p * N + a] == b;
        if (hit && plist.size() / 2 < 255) {
          pmap[(size_t)(a - nF) * nE + (b - nF)] = (uint8_t)(plist.size() / 2);
          plist.push_back((uint8_t)a);
          plist.push_back((uint8_t)b);
        } else if (hit) {
          plist.clear();  // too many pairs for a byte index: no ed mode
          a = N;
          break;
        }
      }
    ts.p2_npairs = (int)(plist.size() / 2);
    blob.insert(blob.end(), pmap.begin(), pmap.end());
    blob.insert(blob.end(), plist.begin(), plist.end());
  }
  ts.p2_nF = nF;
  ts.p2_nFb = nF - n_extra;
  ts.p2_ntasks = (int)tasks.size();
  const int64_t NN = (int64_t)N * N;
  int rc = ctx_alloc(ctx, (void**)&ts.p2, (int64_t)blob.size());
  if (rc == GDML_OK) rc = ctx_alloc(ctx, (void**)&ts.p2_TP, ts.M * NN * 32);  // the packed per-point tables (32 N^2 bytes per point)
  if (rc == GDML_OK && (hipMemcpyAsync(ts.p2, blob.data(), blob.size(), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                        hipStreamSynchronize(ctx->stream) != hipSuccess))  // (the host vector goes out of scope)
    rc = gdml_fail(ctx, GDML_ERR_HIP, "perm2_plan: upload of the plan failed");
  if (rc == GDML_OK) {
    hipLaunchKernelGGL(perm2_pack_kernel, dim3(ceil_div(ts.M * NN, 256)), dim3(256), 0, ctx->stream, ts.XF, ts.GD,
                       reinterpret_cast<const int32_t*>(ts.p2 + ts.p2_o[0]), ts.M, (int)NN, ts.p2_TP);
    ctx->launch_counter++;
    if (hipGetLastError() != hipSuccess) rc = gdml_fail(ctx, GDML_ERR_HIP, "perm2_plan: table packing failed to launch");
  }
  if (rc != GDML_OK) {
    drop();
    // no memory for the extra tables: not an error of the assembly -- the general kernel needs none of them
    return rc == GDML_ERR_OOM ? GDML_ERR_UNSUPPORTED : rc;
  }
  ts.p2_key = key;
  return GDML_OK;
}

// Column points [j0, j0 + n_j) written at col0 + 3N v, row points [i_beg, i_end); lower: A = -K + lam I, blocks j <= i.
int assemble_perm2_launch(gdml_ctx* ctx, double sig, int64_t j0, int64_t n_j, int64_t col0, double* K, int64_t ld, int64_t i_beg,
                          int64_t i_end, int lower, double lam, const int32_t* d_jlist) {
  TrainSet& ts = ctx->ts;
  if (n_j <= 0 || i_end <= i_beg) return GDML_OK;
  GDML_TRY(build_dense_tables(ctx));
  GDML_TRY(perm2_plan(ctx));
  const int N = ts.N, P = ts.P;
  Perm2Args A;
  memset(&A, 0, sizeof(A));
  A.TP = ts.p2_TP; A.blob = ts.p2;
  A.o_src = ts.p2_o[0]; A.o_sigma = ts.p2_o[1]; A.o_tasks = ts.p2_o[2];
  A.n_tasks = ts.p2_ntasks;
  A.M = ts.M; A.N = N; A.P = P; A.nF = ts.p2_nF; A.nFb = ts.p2_nFb; A.sig = sig;
  A.j0 = j0; A.n_j = n_j; A.col0 = col0; A.i_beg = i_beg; A.i_end = i_end;
  A.jlist = d_jlist;
  A.lower = lower; A.lam = lam; A.K = K; A.ld = ld;
  A.dbg = ctx_opt_i(ctx, "asm.perm2_debug", 0);
  // small LDS items into the two free regions: the unused rows of B0 (16 doubles per moved atom) and the tail behind the byte
  // permutation tables
  A.nE = N - ts.p2_nF;
  A.post = (ts.p2_nF >= 2 && A.nE <= 16 && 8 * ts.p2_nF * A.nE + 9 * ts.p2_nF + 1 + A.nE * A.nE <= 15 * P2_MAXN * 8 && ctx_opt_i(ctx, "asm.perm2_post", 1)) ? 1 : 0;
  A.npairs = ts.p2_npairs;
  A.o_pt = ts.p2_o[3];
  A.es = (A.post && 8 * ts.p2_nF * A.nE + 9 * ts.p2_nF + 2 + (9 + 1) * A.nE * A.nE + 2 <= 15 * P2_MAXN * 8 && ctx_opt_i(ctx, "asm.perm2_es", 1)) ? 1 : 0;
  A.ed = (A.post && ts.p2_npairs > 0 && 9 * ts.p2_npairs <= P2_T && ctx_opt_i(ctx, "asm.perm2_ed", 1)) ? 1 : 0;
  int o = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    int r1 = L_B0 + 16 * (ts.p2_nF > 0 ? ts.p2_nF : 0), r1_end = L_NN;  // free rows of B0
    int r2 = L_PERM + (2 * P * N + 7) / 8;                                // tail
    auto place = [&](int n) {  // first fit: B0 rows, then the tail
      if (r1 + n <= r1_end) { const int at = r1; r1 += n; return at; }
      const int at = r2; r2 += n; return at;
    };
    A.l_ed = A.ed ? place(9 * A.npairs) : 0;
    A.l_cn = place(P);
    A.l_task = place(2 * ts.p2_ntasks + 2);
    A.l_sigma = place((N + 1) / 2);
    A.l_pt = A.ed ? place((A.nE * A.nE + 2 * A.npairs + 7) / 8) : 0;
    o = r2;
    if ((size_t)o * 8 <= (size_t)160 * 1024 || !A.ed) break;
    A.ed = 0;  // does not fit with the pair tables: the per-permutation diagonal terms
  }
  const size_t lds = (size_t)o * 8;
  if (lds > (size_t)160 * 1024) return GDML_ERR_UNSUPPORTED;  // (many permutations: the byte tables) -- the general kernel
  const int64_t n_i = i_end - i_beg;
  int i_chunk = ctx_opt_i(ctx, "asm.perm2_i_chunk", 16);
  if (i_chunk < 1) i_chunk = 1;
  while (i_chunk > 2 && n_j * ((n_i + i_chunk - 1) / i_chunk) < 1024) i_chunk >>= 1;
  A.i_chunk = i_chunk;
  dim3 grid((unsigned)n_j, (unsigned)((n_i + i_chunk - 1) / i_chunk));
  const bool direct = ctx_opt_i(ctx, "asm.perm2_direct", 1) != 0;
  (void)hipFuncSetAttribute((const void*)assemble_perm2_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute((const void*)assemble_perm2_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute((const void*)assemble_perm2_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute((const void*)assemble_perm2_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  unsigned long long* d_trace = nullptr;
  if (A.dbg & 1024) {
    GDML_TRY(ctx_alloc(ctx, (void**)&d_trace, 1024 * 8));
    HIP_CHECK(ctx, hipMemsetAsync(d_trace, 0, 1024 * 8, ctx->stream));
    A.trace = d_trace;
  }
  const int slot = ktime_begin(ctx);
  if (A.jlist != nullptr) {  // whole-point index list: its own instantiation (staged rows)
    (void)hipFuncSetAttribute((const void*)assemble_perm2_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((assemble_perm2_kernel<false, false, true>), grid, dim3(P2_T), lds, ctx->stream, A);
  } else if (d_trace && direct) hipLaunchKernelGGL((assemble_perm2_kernel<true, true>), grid, dim3(P2_T), lds, ctx->stream, A);
  else if (d_trace) hipLaunchKernelGGL((assemble_perm2_kernel<true, false>), grid, dim3(P2_T), lds, ctx->stream, A);
  else if (direct) hipLaunchKernelGGL((assemble_perm2_kernel<false, true>), grid, dim3(P2_T), lds, ctx->stream, A);
  else hipLaunchKernelGGL((assemble_perm2_kernel<false, false>), grid, dim3(P2_T), lds, ctx->stream, A);
  if (d_trace) {  // phase stamps of one workgroup: id, shader clock (100 MHz), difference to the previous stamp
    std::vector<unsigned long long> h(1024);
    HIP_CHECK(ctx, hipMemcpyAsync(h.data(), d_trace, 1024 * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 250 && h[2 * k + 1]; ++k)
      fprintf(stderr, "perm2 trace %3d id %2llu t %llu dt %lld\n", k, h[2 * k], h[2 * k + 1], k ? (long long)(h[2 * k + 1] - h[2 * k - 1]) : 0LL);
    for (int g = 0; g < 4; ++g) {  // arrival of the nine wavefronts at the barrier behind the V phase, relative to the first one
      unsigned long long t0 = ~0ull;
      for (int w = 0; w < P2_NW; ++w)
        if (h[600 + w * 4 + g] && h[600 + w * 4 + g] < t0) t0 = h[600 + w * 4 + g];
      fprintf(stderr, "perm2 varrive group %d:", g);
      for (int w = 0; w < P2_NW; ++w) fprintf(stderr, " w%d %lld", w, h[600 + w * 4 + g] ? (long long)(h[600 + w * 4 + g] - t0) : -1LL);
      fprintf(stderr, "\n");
    }
    GDML_TRY(ctx_free(ctx, d_trace));
  }
  const double blocks = lower ? 0.5 * (double)n_i * (double)(n_i + 1) : (double)n_i * (double)n_j;
  ktime_end(ctx, slot, "assemble", 8.0 * blocks * 9.0 * N * N);
  ctx->launch_counter++;
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}
