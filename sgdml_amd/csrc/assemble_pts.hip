// Kernel-matrix assembly for SMALL molecules with a permutation group (8 <= N <= 21 in practice: three row-point images,
// the strip's tables and two V-result areas have to fit the 160 KB of LDS; larger N and P = 1 go to assemble_perm.hip):
// strips of WHOLE column points, producer / consumer wavefronts, everything in LDS.
// N = 21, P = 4, M = 1000: 10.0 ms = 0.40 of HBM (assemble_perm.hip 19.5, the round-2 LDS kernel 61.3).
//
// Reference: sgdml/train.py:97-302 (_assemble_kernel_mat_wkr).  Math and tables as in assemble_perm.hip:
//   K_ij = sum_p [ 5 b_p v_p u_p^T - c_p J_i^T J_j^p ],   d_p = x_i - P_p x_j
//   v_p[a]  = sum_m (x_i[pair(a,m)] - x_j[pair(pi a, pi m)]) G_i(a,m)                      (row atom a)
//   u_p[b]  = sum_m' (x_i[pair(pi^-1 b, pi^-1 m')] - x_j[pair(b,m')]) G_j(b,m')             (column atom b)
//   (J_i^T J_j^p)[(a,.),(b,.)] = G_i(a, pi^-1 b) (x) G_j(b, pi a)   and for pi a = b:  dg_p[b] = sum_m' G_i(pi^-1 b, pi^-1 m') (x) G_j(b,m')
//
// What assemble_perm.hip (one workgroup per CU, 8 wavefronts at 255 VGPRs, barrier-separated V / O / store phases) left on
// the table (profiles/r03_assemble_perm_shapes.txt): the V phase was bound by LDS reads issued from two wavefronts per SIMD,
// and nothing overlapped.  Here:
//   * a strip is PPS = floor(64 / N) whole column points: lane = (column point q, column atom b), no partially covered
//     point, so the strip's x_j table serves the row role (v_p) and the column role (u_p) alike and one pass covers v_p;
//   * the workgroup is split by ROLE.  Consumer wavefront w owns the row atoms 3 w .. 3 w + 2 (27 accumulators per lane),
//     runs the O phase of step t and writes its 9 rows; the producer wavefronts compute v_p, |d_p|^2 -> Matern scalars,
//     u_p, dg_p of step t + 1 into the other half of a double-buffered LDS area meanwhile.  A step is (row point i, group
//     of <= 4 permutations); one barrier per step;
//   * 12 wavefronts per CU at <= 168 VGPRs: three per SIMD instead of two, and roles with short live ranges;
//   * three rotating LDS images of row points: the image of point i + 2 is requested at the start of unit i with LDS-DMA
//     (global_load_lds: no registers, no wait) into the buffer point i - 1 freed, and only has to have landed by the end
//     of the unit (the register hand-over of the first version ended up in scratch with a wait at the top of the unit:
//     a cache-hot image was worth 4.7 of 19 ms).
// Stores: a row of a strip is 3 N PPS consecutive doubles (1512 bytes at N = 21); each row is transposed through a
// 1.5 KB per-wavefront LDS buffer so that a store instruction writes 64 consecutive doubles.
#include <type_traits>

#include "common.h"

struct PtsArgs {
  const double* XF;     // (M,N,N)    XF[x][m][b] = x[pair(b,m)]
  const double* GD;     // (M,N,N,3)  GD[x][m][b] = G_x(b,m)
  const int32_t* perm;  // (P,N) pi_p
  const int32_t* pinv;  // (P,N) pi_p^-1
  int64_t M;
  int N, P;
  double sig, lam;
  int use_E;            // also write the energy-constraint row K[3N M + i, .]
  int lower;            // store -K + lam I, only blocks j <= i
  int64_t j0, n_j;      // column points [j0, j0 + n_j); output column of (j0 + v, b, be) is 3N v + 3 b + be
  int64_t i_beg, i_end; // row points; rows are written relative to i_beg
  int i_chunk;
  int PPS, NO, NV;      // column points per strip, consumer / producer wavefronts
  int n_g, pg_eff;      // permutation groups per row point, permutations per group
  int o_img, o_gjs, o_xjs, o_vb, o_tr, o_perm;  // LDS offsets in doubles
  int nt_store;         // non-temporal stores of K (asm.pts_nt)
  int xcd_map, n_strips; // XCD-contiguous strip order (asm.pts_xcd)
  int dbg;              // timing-only ablation (asm.pts_debug): 1 no stores, 2 no producer tasks, 4 no O phase
  double* K;
  int64_t ld;
};

constexpr int PTS_MAXQ = 8;  // column points per strip (N >= 8)

// LDS area of one step's V results for groups of PG permutations (doubles)
template <int PG>
struct PtsVB {
  static constexpr int VS = PG * 3 * 64, UD = PG * 12 * 64, SC = PG * PTS_MAXQ * 4, SIZE = VS + UD + SC;
};

__device__ __forceinline__ double pts_seg_scan(double v, int pos) {  // inclusive scan inside segments of consecutive lanes
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    const double t = __shfl_up(v, off, 64);
    if (pos >= off) v += t;
  }
  return v;
}

// PG: permutations per step (1, 2, 4).  A group that has fewer (the last one, or P = 3) repeats its first permutation in the
// unused slots with zero Matern scalars: no branch depends on the group size.
// LOWER: the negated lower form of the analytic solver.  SPECIAL: energy-constraint rows, ablation masks, plain stores -- the
// production instantiations carry none of these as run-time flags (one such flag inside the k loop cost the GEMM 3.6 %).
template <int PG, int NA, bool LOWER, bool SPECIAL>
__global__ void __launch_bounds__(768) assemble_pts_kernel(PtsArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using VBL = PtsVB<PG>;
  const int N = A.N, N3 = 3 * N, NN = N * N, P = A.P, PPS = A.PPS, NO = A.NO, NV = A.NV;
  const int T = 64 * (NO + NV);
  double* const IMG = smem + A.o_img;  // [3][ G: [m][b][c] (3 NN) | X: [m][b] (NN) ]: row point ti lives in buffer ti % 3
  double* const GjS = smem + A.o_gjs;  // [m][c][lane]  G_j(b,m)[c] of the lane's column atom
  double* const XjS = smem + A.o_xjs;  // [m][lane]     x_j[pair(b,m)]
  double* const VB = smem + A.o_vb;    // [2][ vs: [pl][c][lane=(q,a)] | ud: [pl][12][lane] | scal: [pl][q][4] ]
  double* const TR = smem + A.o_tr;    // [consumer][192] one transposed output row
  int* const permS = reinterpret_cast<int*>(smem + A.o_perm);  // [P][N]
  int* const pinvS = permS + P * N;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = w >= NO;
  const int vw = w - NO;  // producer index
  // blockIdx.x -> strip: workgroup x runs on XCD x % 8 (gridDim.x is a multiple of 8).  Full form: XCD q takes the q-th eighth
  // of the strips, so that the workgroups resident on one XCD own ADJACENT strips and walk the same rows together -- the
  // cache lines that straddle two strips (rows are 3 N PPS doubles, not a multiple of 16) meet in one L2 and leave it whole
  // (10.6 vs 10.9 ms).  Lower form: a strip's work falls linearly with its index, contiguous eighths would give XCD 0 twice
  // the average (9.7 ms instead of 6.4): plain order there (groups of 8 adjacent strips dealt round-robin measured worse
  // than either).
  const int spx = (int)gridDim.x >> 3;
  const int strip = A.xcd_map ? ((int)blockIdx.x & 7) * spx + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  if (strip >= A.n_strips) return;
  const int jv0 = strip * PPS;  // first (virtual) column point of the strip
  const int ql = lane / N;
  const bool lane_in = ql < PPS;  // lanes past the strip's last point idle (their loads are clamped to lane 0's data)
  const int q = lane_in ? ql : 0;
  const int b = lane_in ? lane - ql * N : 0;
  const int lrow = lane < N ? lane : 0;  // lane that holds entry `lane` of a permutation row
  const int64_t jvq = (jv0 + q < A.n_j) ? jv0 + q : A.n_j - 1;
  const int64_t jpt = A.j0 + jvq;
  constexpr bool lower = LOWER;
  const int dbg = SPECIAL ? A.dbg : 0;
  const bool use_E = SPECIAL && A.use_E != 0;
  const bool nt_store = SPECIAL ? A.nt_store != 0 : true;

  const int64_t i_lo = (lower ? A.j0 + jv0 : A.i_beg) + (int64_t)blockIdx.y * A.i_chunk;
  const int64_t i_top = lower ? A.M : A.i_end;
  const int64_t i_hi = (i_lo + A.i_chunk < i_top) ? i_lo + A.i_chunk : i_top;
  if (i_lo >= i_hi) return;
  const int n_i = (int)(i_hi - i_lo);
  const int n_g = A.n_g, pg_eff = A.pg_eff;  // groups per row point, permutations per group (the last may have fewer)

  const double sig = A.sig, inv_sig = 1.0 / sig;
  const double sqrt5 = 2.23606797749978969641;
  const double base_div = 5.0 / (3.0 * sig * sig * sig * sig);
  const double e_fact = 5.0 / (3.0 * sig * sig * sig);

  // ---- resident tables and the first two images
  for (int e = tid; e < P * N; e += T) {
    permS[e] = A.perm[e];
    pinvS[e] = A.pinv[e];
  }
  {
    const double* gd = A.GD + ((int64_t)jpt * NN + b) * 3;
    const double* xf = A.XF + (int64_t)jpt * NN + b;
    for (int m = w; m < N; m += NO + NV) {
      GjS[(m * 3 + 0) * 64 + lane] = gd[m * N3 + 0];
      GjS[(m * 3 + 1) * 64 + lane] = gd[m * N3 + 1];
      GjS[(m * 3 + 2) * 64 + lane] = gd[m * N3 + 2];
      XjS[m * 64 + lane] = xf[m * N];
    }
  }
  for (int k = 0; k < 2; ++k) {
    const int64_t ii = (i_lo + k < i_hi) ? i_lo + k : i_lo;
    const double* gi = A.GD + ii * (int64_t)NN * 3;
    const double* xi = A.XF + ii * (int64_t)NN;
    double* im = IMG + k * 4 * NN;
    for (int e = tid; e < 3 * NN; e += T) im[e] = gi[e];
    for (int e = tid; e < NN; e += T) im[3 * NN + e] = xi[e];
  }
  __syncthreads();

  // ---- producer side: the V results of step (row point in image buffer b3, group g) into half `par` of the V area
  auto produce = [&](int b3, int g, int par) {
    const int g0 = g * pg_eff;
    const int npg = (P - g0 < pg_eff) ? P - g0 : pg_eff;
    const double* const SG = IMG + b3 * 4 * NN;
    const double* const SX = SG + 3 * NN;
    double* const vs = VB + par * VBL::SIZE;
    double* const ud = vs + VBL::VS;
    double* const scal = ud + VBL::UD;
    if (dbg & 2) return;
    for (int tk = vw; tk < 1 + PG; tk += NV) {
      if (tk == 0) {
        // ---- row role: lane = (q, a = b): v_p, |d_p|^2 for every permutation of the group
        int prow[PG];
        const double* xjp[PG];
#pragma unroll
        for (int pl = 0; pl < PG; ++pl) {
          const int p = g0 + (pl < npg ? pl : 0);
          prow[pl] = permS[p * N + lrow];
          xjp[pl] = XjS + q * N + permS[p * N + b];
        }
        double v[PG][3], nn[PG];
#pragma unroll
        for (int pl = 0; pl < PG; ++pl) v[pl][0] = v[pl][1] = v[pl][2] = nn[pl] = 0.0;
        const double* gp = SG + b * 3;
        const double* xp = SX + b;
        // KB iterations per trip, all their loads first (one LDS latency), then the arithmetic; the scheduling barriers keep
        // the compiler from interleaving them one gather at a time (it serialises load -> wait -> 4 FMAs otherwise)
        auto row_block = [&](int m, auto kb_tag) {
          constexpr int KB = decltype(kb_tag)::value;
          double gv[KB][3], xi[KB], xj[KB][PG];
#pragma unroll
          for (int k = 0; k < KB; ++k) {
            gv[k][0] = gp[(m + k) * N3]; gv[k][1] = gp[(m + k) * N3 + 1]; gv[k][2] = gp[(m + k) * N3 + 2];
            xi[k] = xp[(m + k) * N];
#pragma unroll
            for (int pl = 0; pl < PG; ++pl) xj[k][pl] = xjp[pl][__builtin_amdgcn_readlane(prow[pl], m + k) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < KB; ++k)
#pragma unroll
            for (int pl = 0; pl < PG; ++pl) {
              const double d = xi[k] - xj[k][pl];
              nn[pl] = fma(d, d, nn[pl]);
              v[pl][0] = fma(d, gv[k][0], v[pl][0]);
              v[pl][1] = fma(d, gv[k][1], v[pl][1]);
              v[pl][2] = fma(d, gv[k][2], v[pl][2]);
            }
          __builtin_amdgcn_sched_barrier(0);
        };
        {
          int m = 0;
          for (; m + 2 <= N; m += 2) row_block(m, std::integral_constant<int, 2>());
          for (; m < N; ++m) row_block(m, std::integral_constant<int, 1>());
        }
#pragma unroll
        for (int pl = 0; pl < PG; ++pl) {
          vs[(pl * 3 + 0) * 64 + lane] = v[pl][0];
          vs[(pl * 3 + 1) * 64 + lane] = v[pl][1];
          vs[(pl * 3 + 2) * 64 + lane] = v[pl][2];
          // |d_p|^2 of the lane's point: segmented sum over its N lanes (additions of the segment's own terms only, so
          // that coincident geometries give exactly 0), complete in the lane of the point's last atom
          const double nrm2 = pts_seg_scan(lane_in ? nn[pl] : 0.0, b);
          if (lane_in && b == N - 1) scal[(pl * PTS_MAXQ + q) * 4 + 3] = nrm2;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // the Matern scalars of all (permutation, point) pairs in one pass: lane = pl * 8 + q
        if (lane < PG * PTS_MAXQ) {
          double* sc = scal + lane * 4;
          const bool live = (lane >> 3) < npg && (lane & 7) < PPS;
          const double nrm = sqrt5 * sqrt(0.5 * (live ? sc[3] : 0.0));
          const double ex = exp(-nrm * inv_sig);
          const double bp = ex * base_div;
          sc[0] = live ? 5.0 * bp : 0.0;
          sc[1] = live ? -(sig * sig + sig * nrm) * bp : 0.0;
          sc[2] = live ? -e_fact * (nrm + sig) * ex : 0.0;
        }
      } else {
        // ---- column role: lane = column atom (q, b): u_p (3), dg_p (3 x 3) of one permutation
        const int pl = tk - 1;
        const int p = g0 + (pl < npg ? pl : 0);
        const int ap = pinvS[p * N + b];
        const int prow = pinvS[p * N + lrow];
        const double* xjl = XjS + lane;
        const double* gjl = GjS + lane;
        const double* sxa = SX + ap;
        const double* sga = SG + ap * 3;
        double u0 = 0.0, u1 = 0.0, u2 = 0.0;
        double d00 = 0.0, d01 = 0.0, d02 = 0.0, d10 = 0.0, d11 = 0.0, d12 = 0.0, d20 = 0.0, d21 = 0.0, d22 = 0.0;
        auto col_block = [&](int mp, auto kb_tag) {
          constexpr int KB = decltype(kb_tag)::value;
          double xi[KB], gv[KB][3], rv[KB][3], xj[KB];
#pragma unroll
          for (int k = 0; k < KB; ++k) {
            const int mi = __builtin_amdgcn_readlane(prow, mp + k);
            xi[k] = sxa[mi * N];
            gv[k][0] = sga[mi * N3]; gv[k][1] = sga[mi * N3 + 1]; gv[k][2] = sga[mi * N3 + 2];
            rv[k][0] = gjl[((mp + k) * 3 + 0) * 64]; rv[k][1] = gjl[((mp + k) * 3 + 1) * 64]; rv[k][2] = gjl[((mp + k) * 3 + 2) * 64];
            xj[k] = xjl[(mp + k) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < KB; ++k) {
            const double d = xi[k] - xj[k];
            const double r0 = rv[k][0], r1 = rv[k][1], r2 = rv[k][2];
            u0 = fma(d, r0, u0); u1 = fma(d, r1, u1); u2 = fma(d, r2, u2);
            d00 = fma(gv[k][0], r0, d00); d01 = fma(gv[k][0], r1, d01); d02 = fma(gv[k][0], r2, d02);
            d10 = fma(gv[k][1], r0, d10); d11 = fma(gv[k][1], r1, d11); d12 = fma(gv[k][1], r2, d12);
            d20 = fma(gv[k][2], r0, d20); d21 = fma(gv[k][2], r1, d21); d22 = fma(gv[k][2], r2, d22);
          }
          __builtin_amdgcn_sched_barrier(0);
        };
        {
          int mp = 0;
          for (; mp + 3 <= N; mp += 3) col_block(mp, std::integral_constant<int, 3>());
          for (; mp < N; ++mp) col_block(mp, std::integral_constant<int, 1>());
        }
        double* dst = ud + pl * 12 * 64 + lane;
        dst[0 * 64] = u0; dst[1 * 64] = u1; dst[2 * 64] = u2;
        dst[3 * 64] = d00; dst[4 * 64] = d01; dst[5 * 64] = d02;
        dst[6 * 64] = d10; dst[7 * 64] = d11; dst[8 * 64] = d12;
        dst[9 * 64] = d20; dst[10 * 64] = d21; dst[11 * 64] = d22;
      }
    }
  };

  if (producer) produce(0, 0, 0);
  __syncthreads();

  // The two roles run the same sequence of barriers (one per step, a second one behind every image hand-over) in separate
  // loops, so that neither carries the other's registers.
  if (producer) {
    // image of row point i_lo + tp into buffer b3: 8 N^2 dwords, 64 per instruction, this wavefront's share
    auto request = [&](int tp, int b3) {
      const int64_t ip = (dbg & 8) ? i_lo : i_lo + tp;  // ablation 8: a cache-hot image
      const uint32_t* gsrc = reinterpret_cast<const uint32_t*>(A.GD + ip * (int64_t)NN * 3);
      const uint32_t* xsrc = reinterpret_cast<const uint32_t*>(A.XF + ip * (int64_t)NN);
      char* dst = reinterpret_cast<char*>(IMG + b3 * 4 * NN);
      for (int e0 = vw * 64; e0 < 8 * NN; e0 += NV * 64) {
        const int e = e0 + lane;
        if (e < 8 * NN) {
          const uint32_t* src = (e < 6 * NN) ? gsrc + e : xsrc + (e - 6 * NN);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(dst + e0 * 4), 4, 0, 0);
        }
      }
    };
    int par = 0, b_cur = 0;  // b_cur = ti % 3
    for (int ti = 0; ti < n_i; ++ti) {
      const int b_nxt = b_cur == 2 ? 0 : b_cur + 1, b_nn = b_nxt == 2 ? 0 : b_nxt + 1;
      if (ti + 2 < n_i) request(ti + 2, b_nn);
      for (int g = 0; g < n_g; ++g) {
        par ^= 1;
        if (g + 1 < n_g) produce(b_cur, g + 1, par);
        else if (ti + 1 < n_i) produce(b_nxt, 0, par);
        if (g == n_g - 1) __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the requested image has landed
        __syncthreads();  // this step consumed, the next one produced
      }
      b_cur = b_nxt;
    }
    return;
  }

  // ---- consumer: column bookkeeping of the transposed rows
  const int n_cols = PPS * N3;  // columns of a full strip
  int tcol[3], tpt[3];
#pragma unroll
  for (int t3 = 0; t3 < 3; ++t3) {
    const int c = 64 * t3 + lane;  // column inside the strip
    const int pt = c / N3;
    const bool ok = c < n_cols && jv0 + pt < A.n_j;
    tcol[t3] = ok ? jv0 * N3 + c : -1;
    tpt[t3] = pt;
  }
  double* const trw = TR + w * 192;
  const int a_base = NA * w;  // first row atom of this consumer
  int arow[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) arow[k] = (a_base + k < N) ? a_base + k : N - 1;
  double acc[NA][3][3];
  double erow[3] = {0.0, 0.0, 0.0};
  const double* const gjl = GjS + lane;

  int par = 0, b_cur = 0;
  for (int ti = 0; ti < n_i; ++ti) {
    const int64_t i = i_lo + ti;
    const double* const SG = IMG + b_cur * 4 * NN;
    b_cur = b_cur == 2 ? 0 : b_cur + 1;
#pragma unroll
    for (int k = 0; k < NA; ++k)
#pragma unroll
      for (int al = 0; al < 3; ++al) acc[k][al][0] = acc[k][al][1] = acc[k][al][2] = 0.0;
    erow[0] = erow[1] = erow[2] = 0.0;
    for (int g = 0; g < n_g; ++g) {
      // ================= phase O of step (i, g): this wavefront's three row atoms, the permutations of the group
      const int g0 = g * pg_eff;
      const int npg = (P - g0 < pg_eff) ? P - g0 : pg_eff;
      const double* const vs = VB + par * VBL::SIZE;
      const double* const ud = vs + VBL::VS;
      const double* const scal = ud + VBL::UD;
      par ^= 1;
      // the permutation-table entries of the whole group in one batch of loads (one LDS latency instead of one per
      // permutation); unused slots repeat the group's first permutation, their Matern scalars are zero
      int ap_g[PG], prow_g[PG];
#pragma unroll
      for (int pl = 0; pl < PG; ++pl) {
        const int p = g0 + (pl < npg ? pl : 0);
        ap_g[pl] = pinvS[p * N + b];
        prow_g[pl] = permS[p * N + lrow];
      }
#pragma unroll
      for (int pl = 0; pl < PG; ++pl) {
        if ((dbg & 4) || pl >= npg) break;
        const int ap = ap_g[pl];
        const int prow = prow_g[pl];
        const double* sc = scal + (pl * PTS_MAXQ + q) * 4;
        const double beta = sc[0], cn = sc[1];
        const double* udp = ud + pl * 12 * 64 + lane;
        const double ur0 = udp[0], ur1 = udp[64], ur2 = udp[128];
        const double U0 = beta * ur0, U1 = beta * ur1, U2 = beta * ur2;
        {  // [pi a = b]: -c_p dg_p goes to the one row atom a = pi^-1 b, if it is one of this wavefront's
          const int kk = ap - a_base;
          if (__any(kk >= 0 && kk < NA)) {
            double DG[3][3];
#pragma unroll
            for (int al = 0; al < 3; ++al)
#pragma unroll
              for (int be = 0; be < 3; ++be) DG[al][be] = cn * udp[(3 + al * 3 + be) * 64];
#pragma unroll
            for (int k = 0; k < NA; ++k) {
              if (kk == k) {
#pragma unroll
                for (int al = 0; al < 3; ++al)
#pragma unroll
                  for (int be = 0; be < 3; ++be) acc[k][al][be] += DG[al][be];
              }
            }
          }
        }
        if (use_E && w == 0) {
          const double ce = sc[2];
          erow[0] = fma(ce, ur0, erow[0]); erow[1] = fma(ce, ur1, erow[1]); erow[2] = fma(ce, ur2, erow[2]);
        }
        const double* gib = SG + ap * N3;             // G_i(., pi^-1 b): [a][al]
        const double* vq = vs + pl * 3 * 64 + q * N;  // v_p of the lane's column point: [c][a]
        double vv[NA][3], gi[NA][3], gj[NA][3];
#pragma unroll
        for (int k = 0; k < NA; ++k) {
          const int a = arow[k];
          const int pa = __builtin_amdgcn_readlane(prow, a);
#pragma unroll
          for (int c3 = 0; c3 < 3; ++c3) {
            vv[k][c3] = vq[c3 * 64 + a];
            gi[k][c3] = gib[a * 3 + c3];
            gj[k][c3] = gjl[(pa * 3 + c3) * 64];
          }
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
          const double w0 = cn * gj[k][0], w1 = cn * gj[k][1], w2 = cn * gj[k][2];
#pragma unroll
          for (int al = 0; al < 3; ++al) {
            acc[k][al][0] = fma(gi[k][al], w0, fma(vv[k][al], U0, acc[k][al][0]));
            acc[k][al][1] = fma(gi[k][al], w1, fma(vv[k][al], U1, acc[k][al][1]));
            acc[k][al][2] = fma(gi[k][al], w2, fma(vv[k][al], U2, acc[k][al][2]));
          }
        }
      }
      __syncthreads();  // this step consumed, the next one produced
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      // ---- the rows of this wavefront, underneath the producers' next step: one row per LDS round trip, stores of 64
      // consecutive doubles
      const int64_t row0 = i * N3;
      const int64_t lrow0 = row0 - A.i_beg * N3;
      bool tok[3];
      int dcol[3];
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3) {
        tok[t3] = tcol[t3] >= 0 && (!lower || A.j0 + jv0 + tpt[t3] <= i);
        dcol[t3] = lower ? (int)((int64_t)tcol[t3] + A.j0 * N3 - row0) : -1;
      }
      const double lamv = lower ? A.lam : 0.0;
#pragma unroll
      for (int k = 0; k < NA; ++k) {
        const int a = a_base + k;
        if (a < N && (!(dbg & 1) || acc[k][0][0] == 1.2345e-300)) {
#pragma unroll
          for (int al = 0; al < 3; ++al) {
            const int rr = 3 * a + al;
#pragma unroll
            for (int be = 0; be < 3; ++be) trw[3 * lane + be] = lower ? -acc[k][al][be] : acc[k][al][be];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            double val[3];
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) val[t3] = trw[64 * t3 + lane];
            double* dst = A.K + (lrow0 + rr) * A.ld;
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
              const double o = val[t3] + ((dcol[t3] == rr) ? lamv : 0.0);
              if (tok[t3]) {
                if (nt_store) __builtin_nontemporal_store(o, dst + (unsigned)tcol[t3]);  // K is streamed out: keep the L2 for the images
                else dst[(unsigned)tcol[t3]] = o;
              }
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
      if (use_E && w == 0 && lane_in && jv0 + q < A.n_j) {
        double* dst = A.K + (A.M * N3 + i) * A.ld + (int64_t)(jv0 + q) * N3 + 3 * b;
        dst[0] = erow[0]; dst[1] = erow[1]; dst[2] = erow[2];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

int build_dense_tables(gdml_ctx* ctx);

bool assemble_pts_applicable(const gdml_ctx* ctx) {
  const TrainSet& ts = ctx->ts;
  // measured (profiles/r03_assemble_pts.txt): 1.3-1.5x over assemble_perm.hip for permutation groups; for P = 1 (N = 22-24,
  // the register-resident kernels cover N <= 21) the general kernel is as fast, so it keeps that case (asm.pts = 2 forces)
  const int opt = ctx_opt_i(ctx, "asm.pts", 1);
  return ts.N >= 8 && ts.N <= 24 && (opt == 2 || (opt == 1 && ts.P > 1));
}

template <int PG, int NA, bool LOWER, bool SPECIAL>
static void pts_launch_t(gdml_ctx* ctx, PtsArgs& A, dim3 grid, size_t* lds_out = nullptr) {
  const int N = A.N, NN = N * N;
  int o = 0;
  A.o_img = o; o += 3 * 4 * NN;
  A.o_gjs = o; o += N * 3 * 64;
  A.o_xjs = o; o += N * 64;
  A.o_vb = o; o += 2 * PtsVB<PG>::SIZE;
  A.o_tr = o; o += A.NO * 192;
  A.o_perm = o; o += (2 * A.P * N + 1) / 2;
  const size_t lds = (size_t)o * 8;
  if (lds_out) { *lds_out = lds; return; }
  (void)hipFuncSetAttribute((const void*)assemble_pts_kernel<PG, NA, LOWER, SPECIAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((assemble_pts_kernel<PG, NA, LOWER, SPECIAL>), grid, dim3(64 * (A.NO + A.NV)), lds, ctx->stream, A);
}

// Column points [j0, j0 + n_j) (output columns from 0), row points [i_beg, i_end) (rows relative to i_beg).
int assemble_pts_launch(gdml_ctx* ctx, double sig, int use_E, int64_t j0, int64_t n_j, double* K, int64_t ld, int64_t i_beg,
                        int64_t i_end, int lower, double lam) {
  TrainSet& ts = ctx->ts;
  if (n_j <= 0 || i_end <= i_beg) return GDML_OK;
  GDML_TRY(build_dense_tables(ctx));
  const int N = ts.N, P = ts.P;
  if (lower && (use_E || j0 != 0 || i_beg != 0 || n_j != ts.M || i_end != ts.M))
    return gdml_fail(ctx, GDML_ERR_INVALID, "assemble_pts: the lower form needs the dense full column range");
  if ((int64_t)P * N > 4096) return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "assemble_pts: permutation tables of %d x %d entries", P, N);
  PtsArgs A;
  memset(&A, 0, sizeof(A));
  A.XF = ts.XF; A.GD = ts.GD; A.perm = ts.perm; A.pinv = ts.pinv;
  A.M = ts.M; A.N = N; A.P = P; A.sig = sig; A.lam = lam; A.use_E = use_E; A.lower = lower ? 1 : 0;
  A.j0 = j0; A.n_j = n_j; A.i_beg = i_beg; A.i_end = i_end;
  A.K = K; A.ld = ld;
  A.dbg = ctx_opt_i(ctx, "asm.pts_debug", 0);
  A.nt_store = ctx_opt_i(ctx, "asm.pts_nt", 1);
  A.PPS = 64 / N;
  A.n_g = (P + 3) / 4;
  A.pg_eff = (P + A.n_g - 1) / A.n_g;  // equal groups of at most four
  const int PG = A.pg_eff >= 3 ? 4 : A.pg_eff;
  // row atoms per consumer wavefront: three (1 and 2 -- more consumers, fewer producers -- measured slower or equal at
  // N = 9 .. 21, the producers being the critical path: profiles/r03_assemble_pts.txt)
  const bool special = use_E || A.dbg != 0 || !A.nt_store;
  auto dispatch = [&](dim3 g, size_t* lds_out) {
#define PTS_GO(pg)                                                                         \
  do {                                                                                     \
    if (special) { if (A.lower) pts_launch_t<pg, 3, true, true>(ctx, A, g, lds_out); else pts_launch_t<pg, 3, false, true>(ctx, A, g, lds_out); } \
    else { if (A.lower) pts_launch_t<pg, 3, true, false>(ctx, A, g, lds_out); else pts_launch_t<pg, 3, false, false>(ctx, A, g, lds_out); }      \
  } while (0)
    if (PG == 1) PTS_GO(1); else if (PG == 2) PTS_GO(2); else PTS_GO(4);
#undef PTS_GO
  };
  // 12 wavefronts: ceil(N / 3) consumers, then one producer per task of a step (row role + one column role per permutation
  // of the group) as far as they fit
  A.NO = (N + 2) / 3;
  A.NV = 12 - A.NO < 1 + PG ? 12 - A.NO : 1 + PG;
  const int nv_opt = ctx_opt_i(ctx, "asm.pts_nv", 0);
  if (nv_opt >= 1 && nv_opt <= 12 - A.NO) A.NV = nv_opt;
  size_t lds = 0;
  if (A.NV >= 1) dispatch(dim3(1), &lds);
  if (A.NV < 1 || lds > 160 * 1024) return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "assemble_pts: no LDS layout for N=%d P=%d", N, P);
  const int64_t n_strips = (n_j + A.PPS - 1) / A.PPS;
  const int64_t n_i = i_end - i_beg;
  int i_chunk = ctx_opt_i(ctx, "asm.pts_i_chunk", 64);
  if (i_chunk < 1) i_chunk = 1;
  while (i_chunk > 4 && n_strips * ((n_i + i_chunk - 1) / i_chunk) < 1024) i_chunk >>= 1;
  A.i_chunk = i_chunk;
  A.xcd_map = ctx_opt_i(ctx, "asm.pts_xcd", 1) && !A.lower;
  A.n_strips = (int)n_strips;
  dim3 grid((unsigned)((n_strips + 7) / 8 * 8), (unsigned)((n_i + i_chunk - 1) / i_chunk));
  const int slot = ktime_begin(ctx);
  dispatch(grid, nullptr);
  const double blocks = A.lower ? 0.5 * (double)n_i * (double)(n_i + 1) : (double)n_i * (double)n_j;
  ktime_end(ctx, slot, "assemble", 8.0 * blocks * 9.0 * N * N);
  ctx->launch_counter++;
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}
