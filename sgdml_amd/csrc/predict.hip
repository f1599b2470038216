// Energy / force prediction contraction on gfx950.
//
// Replaces GDMLPredict.predict / _predict_wkr (sgdml/predict.py:84-245, :1146-1294),
// GDMLPredict.set_alphas (:551-601), the permuted-table construction (:426-447) and
// GDMLTorchPredict._forward (sgdml/torchtools.py:877-1046).
//
// Per query x (descriptor) and permuted training row (j,p):  d = x - X_jp, n = sqrt5 |d|,
//   b = 5/(3 sig^3) exp(-n/sig), a = d . (J alpha)_jp
//   F_x += (5/sig) a b d - b (n + sig) (J alpha)_jp ;  E' += a b (n + sig)      (predict.py:199-217)
//   with alphas_E: F_x += aE_j b (n+sig) d ; E' += aE_j (1 + n/sig (1 + n/(3 sig))) exp(-n/sig)
// and finally F = J_x^T F_x (desc.py:388-408).
//
// Three kernels, chosen by batch size (predict_device): predict_mfma_kernel<NT> for B >= 256 and D <= 256
// (both contractions on v_mfma_f64_16x16x4_f64, see its header below), predict_bulk_kernel<NCH> (its VALU
// predecessor, GDML_PREDICT_NO_MFMA=1) and predict_kernel<KPL,QB> for small batches / large D:
// Layout of predict_kernel: the descriptor index k runs across the 64 lanes of a wavefront (KPL entries per
// lane), so every table row is one coalesced read; each wavefront owns QB queries (held in
// registers) and a contiguous split of the table rows, reduces |d|^2 and a with cross-lane
// shuffles, and keeps its partial F_x in registers.  Partials of the splits are summed in a fixed
// order by the epilogue kernel (deterministic), which also applies J_x^T.
#include <stddef.h>

#include "common.h"
#include <chrono>
#include <thread>

int upload_perms(gdml_ctx* ctx, const int64_t* tril_perms, int P, int N, std::vector<int32_t>& h_tp,
                 std::vector<int32_t>& h_perm, std::vector<int32_t>& h_pinv);

// out[(m*P+p)*D + k] = in[m*D + tp[p*D+k]]
__global__ void __launch_bounds__(256) permute_rows_kernel(const double* __restrict__ in,
                                                           const int32_t* __restrict__ tp, int64_t M,
                                                           int D, int P, double* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = M * P * D;
  if (t >= total) return;
  int k = (int)(t % D);
  int64_t mp = t / D;
  int p = (int)(mp % P);
  int64_t m = mp / P;
  out[t] = in[m * D + tp[(size_t)p * D + k]];
}

__global__ void __launch_bounds__(256) repeat_kernel(const double* __restrict__ in, int64_t M, int P,
                                                     double* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < M * P) out[t] = in[t / P];
}

// (J v)[m,k] = g[m,k,:] . (v[m,j_k,:] - v[m,i_k,:])     (desc.py:368-385)
__global__ void __launch_bounds__(256) jdotv_kernel(const double* __restrict__ g,
                                                    const double* __restrict__ v, int64_t M, int N,
                                                    int D, double* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * D) return;
  int64_t m = t / D;
  int k = (int)(t - m * D);
  int i = (int)((1.0 + sqrt(1.0 + 8.0 * (double)k)) * 0.5);
  while (i * (i - 1) / 2 > k) --i;
  while ((i + 1) * i / 2 <= k) ++i;
  int j = k - i * (i - 1) / 2;
  const double* vi = v + (m * N + i) * 3;
  const double* vj = v + (m * N + j) * 3;
  const double* gg = g + t * 3;
  out[t] = gg[0] * (vj[0] - vi[0]) + gg[1] * (vj[1] - vi[1]) + gg[2] * (vj[2] - vi[2]);
}

typedef double d2x __attribute__((ext_vector_type(2)));

struct PredArgs {
  const double* xq;   // (B,D) query descriptors
  const double* xp;   // (MP,D)
  const double* jap;  // (MP,D)
  const double* aE;   // (MP) or null
  int64_t B, MP;
  int D;
  double sig;
  int JS;             // row splits
  int64_t rows_per_split;
  double* part_F;     // (JS,B,D)
  double* part_E;     // (JS,B)
};

template <int KPL, int QB>
__global__ void __launch_bounds__(256) predict_kernel(PredArgs A) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_qt = (A.B + QB - 1) / QB;
  if (gw >= n_qt * A.JS) return;
  const int64_t qt = gw / A.JS;
  const int split = (int)(gw - qt * A.JS);
  const int D = A.D;
  const double sig = A.sig, inv_sig = 1.0 / sig;
  const double sqrt5 = 2.23606797749978969641;
  const double fact = 5.0 / (3.0 * sig * sig * sig);
  const double dscale = 5.0 / sig;
  const double inv_3sig = 1.0 / (3.0 * sig);

  double x[QB][KPL], Fx[QB][KPL], E[QB];
#pragma unroll
  for (int q = 0; q < QB; ++q) {
    const int64_t qi = qt * QB + q;
    E[q] = 0.0;
#pragma unroll
    for (int t = 0; t < KPL; ++t) {
      const int k = lane + 64 * t;
      x[q][t] = (qi < A.B && k < D) ? A.xq[qi * D + k] : 0.0;
      Fx[q][t] = 0.0;
    }
  }
  const int64_t r0 = (int64_t)split * A.rows_per_split;
  const int64_t r1 = (r0 + A.rows_per_split < A.MP) ? r0 + A.rows_per_split : A.MP;
  const bool has_aE = A.aE != nullptr;

  // Table rows two at a time with the next row requested before the current one is used: as a plain row loop the loads
  // of row r + 1 were issued after the reductions of row r -- one memory round trip per row and wavefront (16 rows =
  // ~12 of the kernel's 15.7 us for a single query against 1000 training points).  The loads are relaxed wavefront-scope
  // atomic loads (plain global_load_dwordx2): LLVM otherwise folds the loop-carried values back into a load at the
  // head of the loop body (foldPHIArgLoadIntoPHI), right in front of their use.
  auto load_row = [&](int64_t r, double (&X)[KPL], double (&JA)[KPL], double& ae) {
#pragma unroll
    for (int t = 0; t < KPL; ++t) {
      const int k = lane + 64 * t;
      const int kc = k < D ? k : D - 1;  // padding lanes read a valid entry and drop it (d = 0 - 0 below)
      const double xv = __hip_atomic_load(A.xp + r * D + kc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      const double jv = __hip_atomic_load(A.jap + r * D + kc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      X[t] = (k < D) ? xv : 0.0;
      JA[t] = (k < D) ? jv : 0.0;
    }
    ae = has_aE ? __hip_atomic_load(A.aE + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0.0;
  };
  auto row = [&](const double (&X)[KPL], const double (&JA)[KPL], double ae) {
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      double d[KPL];
      double s2 = 0.0, sa = 0.0;
#pragma unroll
      for (int t = 0; t < KPL; ++t) {
        d[t] = x[q][t] - X[t];  // padding lanes: 0 - 0
        s2 += d[t] * d[t];
        sa += d[t] * JA[t];
      }
      s2 = wave_sum(s2);
      sa = wave_sum(sa);
      const double nrm = sqrt5 * sqrt(s2);
      const double ex = exp(-nrm * inv_sig);
      const double b = fact * ex;
      const double b2 = b * (nrm + sig);
      double w1 = dscale * sa * b;
      E[q] += sa * b2;
      if (has_aE) {
        w1 += ae * b2;
        E[q] += ae * (1.0 + (nrm * inv_sig) * (1.0 + nrm * inv_3sig)) * ex;
      }
#pragma unroll
      for (int t = 0; t < KPL; ++t) Fx[q][t] += w1 * d[t] - b2 * JA[t];
    }
  };
  if (r0 < r1) {
    double X0[KPL], JA0[KPL], X1[KPL], JA1[KPL], ae0, ae1;
    int64_t r = r0;
    load_row(r, X0, JA0, ae0);
    for (; r + 1 < r1; r += 2) {
      load_row(r + 1, X1, JA1, ae1);
      row(X0, JA0, ae0);
      load_row(r + 2 < r1 ? r + 2 : r1 - 1, X0, JA0, ae0);
      row(X1, JA1, ae1);
    }
    if (r < r1) row(X0, JA0, ae0);  // odd count: X0 holds row r1 - 1
  }
#pragma unroll
  for (int q = 0; q < QB; ++q) {
    const int64_t qi = qt * QB + q;
    if (qi < A.B) {
#pragma unroll
      for (int t = 0; t < KPL; ++t) {
        const int k = lane + 64 * t;
        if (k < D) A.part_F[((int64_t)split * A.B + qi) * D + k] = Fx[q][t];
      }
      if (lane == 0) A.part_E[(int64_t)split * A.B + qi] = E[q];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Large molecules (D > 1024, up to N = 128: D = 8128), batches below the GEMM pipeline of predict_wide.hip:
// one workgroup of 512 threads per (query, row split); thread t owns the descriptor entries k = t + 512 s
// (x, F_x and the current table row in registers), the two dot products of a row are block reductions.
// Same arithmetic as predict_kernel (predict.py:168-245).
// ------------------------------------------------------------------------------------------
template <int KT>
__global__ void __launch_bounds__(512) predict_big_kernel(PredArgs A) {
  __shared__ double red[2][2][8];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t qi = blockIdx.x;
  const int split = blockIdx.y;
  const int D = A.D;
  const double sig = A.sig, inv_sig = 1.0 / sig;
  const double sqrt5 = 2.23606797749978969641;
  const double fact = 5.0 / (3.0 * sig * sig * sig);
  const double dscale = 5.0 / sig;
  const double inv_3sig = 1.0 / (3.0 * sig);
  double x[KT], Fx[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const int k = tid + 512 * t;
    x[t] = (k < D) ? A.xq[qi * D + k] : 0.0;
    Fx[t] = 0.0;
  }
  double E = 0.0;
  const int64_t r0 = (int64_t)split * A.rows_per_split;
  const int64_t r1 = (r0 + A.rows_per_split < A.MP) ? r0 + A.rows_per_split : A.MP;
  const bool has_aE = A.aE != nullptr;
  int buf = 0;
  for (int64_t r = r0; r < r1; ++r, buf ^= 1) {
    double d[KT], JA[KT];
    double s2 = 0.0, sa = 0.0;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int k = tid + 512 * t;
      const double X = (k < D) ? A.xp[r * D + k] : 0.0;
      JA[t] = (k < D) ? A.jap[r * D + k] : 0.0;
      d[t] = x[t] - X;
      s2 += d[t] * d[t];
      sa += d[t] * JA[t];
    }
    s2 = wave_sum(s2);
    sa = wave_sum(sa);
    if (lane == 0) {
      red[buf][0][wv] = s2;
      red[buf][1][wv] = sa;
    }
    __syncthreads();  // one barrier per row: the two buffers alternate
    s2 = 0.0;
    sa = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s2 += red[buf][0][u];
      sa += red[buf][1][u];
    }
    const double nrm = sqrt5 * sqrt(s2);
    const double ex = exp(-nrm * inv_sig);
    const double b = fact * ex;
    const double b2 = b * (nrm + sig);
    double w1 = dscale * sa * b;
    E += sa * b2;
    if (has_aE) {
      const double ae = A.aE[r];
      w1 += ae * b2;
      E += ae * (1.0 + (nrm * inv_sig) * (1.0 + nrm * inv_3sig)) * ex;
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) Fx[t] += w1 * d[t] - b2 * JA[t];
  }
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const int k = tid + 512 * t;
    if (k < D) A.part_F[((int64_t)split * A.B + qi) * D + k] = Fx[t];
  }
  if (tid == 0) A.part_E[(int64_t)split * A.B + qi] = E;
}

// ------------------------------------------------------------------------------------------
// Bulk variant (B >= 32, D <= 256): no cross-lane reductions at all.
// A workgroup owns 32 queries and a split of the table rows, processed in tiles of 32 rows.
//   phase 1: every thread owns 2 x 2 (query,row) pairs and runs over k with the query / table
//            tiles staged in LDS as [k][q] / [k][r] (16-byte reads, broadcast across lanes);
//            |d|^2 and a stay in registers, the Matern scalars are evaluated once per pair and
//            written to LDS as w1[r][q], b2[r][q];
//   phase 2: every thread owns one query and every 8th k (x and F_x in registers) and accumulates
//            F_x[q][k] += w1 (x - X_r[k]) - b2 JA_r[k] over the 32 rows.
// Per (query,row,k): 6 VALU instructions and < 2 LDS reads.
// ------------------------------------------------------------------------------------------
#define PQ 32
#define PR 32
#define PKC 32
#define PPITCH 34

template <int NCH>  // number of 32-wide k chunks, D <= 32 NCH
__global__ void __launch_bounds__(256, 2) predict_bulk_kernel(PredArgs A) {
  __shared__ __attribute__((aligned(16))) double xs[PKC * PPITCH];
  __shared__ __attribute__((aligned(16))) double Xs[PKC * PPITCH];
  __shared__ __attribute__((aligned(16))) double Js[PKC * PPITCH];
  __shared__ __attribute__((aligned(16))) double w1s[PR * PPITCH];
  __shared__ __attribute__((aligned(16))) double b2s[PR * PPITCH];
  const int tid = threadIdx.x;
  const int D = A.D;
  const int64_t q0 = (int64_t)blockIdx.x * PQ;
  const int split = blockIdx.y;
  const int64_t r_beg = (int64_t)split * A.rows_per_split;
  const int64_t r_end = (r_beg + A.rows_per_split < A.MP) ? r_beg + A.rows_per_split : A.MP;
  const double sig = A.sig, inv_sig = 1.0 / sig;
  const double sqrt5 = 2.23606797749978969641;
  const double fact = 5.0 / (3.0 * sig * sig * sig);
  const double dscale = 5.0 / sig;
  const double inv_3sig = 1.0 / (3.0 * sig);
  const bool has_aE = A.aE != nullptr;

  // phase-1 ownership: queries qa, qa+1 ; rows ra, ra+1 of the tile
  const int qa = (tid & 15) * 2, ra = (tid >> 4) * 2;
  // phase-2 ownership: query q2, k = 32 c + kg + 8 i
  const int q2 = tid & 31, kg = tid >> 5;
  double xr[NCH * 4], Fx[NCH * 4];
#pragma unroll
  for (int a = 0; a < NCH * 4; ++a) {
    const int k = 32 * (a >> 2) + kg + 8 * (a & 3);
    const int64_t qi = q0 + q2;
    xr[a] = (qi < A.B && k < D) ? A.xq[qi * D + k] : 0.0;
    Fx[a] = 0.0;
  }
  double E0 = 0.0, E1 = 0.0;  // partial energies of queries qa, qa+1

  // staging helpers: element e in [0,1024): row/query = e >> 5, kk = e & 31
  auto stage_rows = [&](int64_t r0, int c) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int e = tid + 256 * s;
      const int rr = e >> 5, kk = e & 31;
      const int64_t r = r0 + rr;
      const int k = 32 * c + kk;
      const bool ok = r < r_end && k < D;
      Xs[kk * PPITCH + rr] = ok ? A.xp[r * D + k] : 0.0;
      Js[kk * PPITCH + rr] = ok ? A.jap[r * D + k] : 0.0;
    }
  };

  for (int64_t r0 = r_beg; r0 < r_end; r0 += PR) {
    // ---------------- phase 1
    double s2[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, sa[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    for (int c = 0; c < NCH; ++c) {
      __syncthreads();  // previous readers of xs/Xs/Js done
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int e = tid + 256 * s;
        const int qq = e >> 5, kk = e & 31;
        const int64_t qi = q0 + qq;
        const int k = 32 * c + kk;
        xs[kk * PPITCH + qq] = (qi < A.B && k < D) ? A.xq[qi * D + k] : 0.0;
      }
      stage_rows(r0, c);
      __syncthreads();
#pragma unroll 8
      for (int kk = 0; kk < PKC; ++kk) {
        const d2x xv = *reinterpret_cast<const d2x*>(&xs[kk * PPITCH + qa]);
        const d2x Xv = *reinterpret_cast<const d2x*>(&Xs[kk * PPITCH + ra]);
        const d2x Jv = *reinterpret_cast<const d2x*>(&Js[kk * PPITCH + ra]);
        const double d00 = xv.x - Xv.x, d01 = xv.x - Xv.y, d10 = xv.y - Xv.x, d11 = xv.y - Xv.y;
        s2[0][0] += d00 * d00; sa[0][0] += d00 * Jv.x;
        s2[0][1] += d01 * d01; sa[0][1] += d01 * Jv.y;
        s2[1][0] += d10 * d10; sa[1][0] += d10 * Jv.x;
        s2[1][1] += d11 * d11; sa[1][1] += d11 * Jv.y;
      }
    }
    // Matern scalars once per pair
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t r = r0 + ra + j;
        double w1 = 0.0, b2 = 0.0, e = 0.0;
        if (r < r_end) {
          const double nrm = sqrt5 * sqrt(s2[i][j]);
          const double ex = exp(-nrm * inv_sig);
          const double b = fact * ex;
          b2 = b * (nrm + sig);
          w1 = dscale * sa[i][j] * b;
          e = sa[i][j] * b2;
          if (has_aE) {
            const double ae = A.aE[r];
            w1 += ae * b2;
            e += ae * (1.0 + (nrm * inv_sig) * (1.0 + nrm * inv_3sig)) * ex;
          }
        }
        w1s[(ra + j) * PPITCH + qa + i] = w1;
        b2s[(ra + j) * PPITCH + qa + i] = b2;
        if (i == 0) E0 += e; else E1 += e;
      }
    // ---------------- phase 2
    for (int c = 0; c < NCH; ++c) {
      __syncthreads();  // w1s/b2s visible (first chunk); previous readers of Xs/Js done
      if (NCH > 1 || true) stage_rows(r0, c);
      __syncthreads();
#pragma unroll 4
      for (int rr = 0; rr < PR; rr += 2) {
        const double w1a = w1s[rr * PPITCH + q2], w1b = w1s[(rr + 1) * PPITCH + q2];
        const double b2a = b2s[rr * PPITCH + q2], b2b = b2s[(rr + 1) * PPITCH + q2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int kk = kg + 8 * i;
          const d2x Xv = *reinterpret_cast<const d2x*>(&Xs[kk * PPITCH + rr]);
          const d2x Jv = *reinterpret_cast<const d2x*>(&Js[kk * PPITCH + rr]);
          const double x = xr[c * 4 + i];
          double f = Fx[c * 4 + i];
          f += w1a * (x - Xv.x);
          f -= b2a * Jv.x;
          f += w1b * (x - Xv.y);
          f -= b2b * Jv.y;
          Fx[c * 4 + i] = f;
        }
      }
    }
  }
  // ---- outputs: partial F_x (q2, k) and partial E (reduced over the 16 threads sharing qa)
  {
    const int64_t qi = q0 + q2;
    if (qi < A.B) {
#pragma unroll
      for (int a = 0; a < NCH * 4; ++a) {
        const int k = 32 * (a >> 2) + kg + 8 * (a & 3);
        if (k < D) A.part_F[((int64_t)split * A.B + qi) * D + k] = Fx[a];
      }
    }
  }
  __syncthreads();
  w1s[(tid >> 4) * PPITCH + qa] = E0;
  w1s[(tid >> 4) * PPITCH + qa + 1] = E1;
  __syncthreads();
  if (tid < PQ && q0 + tid < A.B) {
    double e = 0.0;
    for (int g = 0; g < 16; ++g) e += w1s[g * PPITCH + tid];
    A.part_E[(int64_t)split * A.B + q0 + tid] = e;
  }
}

// ------------------------------------------------------------------------------------------
// MFMA variant of the bulk kernel (B >= 256, D <= 256): the contractions over k and the accumulation
// over the table rows are GEMM-shaped, so they go to v_mfma_f64_16x16x4_f64 (same peak as the fp64
// VALU on MI355X, but one instruction carries 2048 flops and needs no per-FMA operand traffic).
// Per workgroup: 64 queries (16 per wavefront) against one split of the table rows, 16 rows at a time:
//   phase 1 (transposed):  P1^T = X x^T,  P2^T = JA x^T   (K = D; X/JA tile from LDS, x in registers)
//            |d|^2 = |x|^2 + |X_r|^2 - 2 P1 (clamped at 0),  a = P2 - X_r.JA_r   (row terms precomputed)
//            -- every Matern quantity that multiplies O(1) data is second order in n at n = 0, so the
//            cancellation in |d|^2 for coincident points (training-set mode) is harmless;
//   scalars: w1, b2 for the lane's 4 (row, query) pairs.  The C/D layout of the transposed product
//            (row = l>>4 + 4 i, query = l&15) IS the A-operand layout of phase 2, so the weights go
//            from one MFMA to the next without leaving the lane;
//   phase 2: F_x -= [w1 | b2] [X ; JA]   (K = 32 per row tile), the 16 x D accumulators of the wave
//            stay in registers for the whole split;  F_x += x * sum_r w1 at the end.
// The X/JA row tiles (16 x D each) are double-buffered in LDS: the next tile is fetched into registers
// while the current one is being used, one barrier per tile.  One workgroup per CU (the accumulators,
// the query operand and the prefetch registers need more than 256 VGPRs).
// ------------------------------------------------------------------------------------------
typedef double d4p __attribute__((ext_vector_type(4)));

// nX[r] = |X_r|^2, cX[r] = X_r . JA_r  (one wavefront per table row)
__global__ void __launch_bounds__(256) row_stats_kernel(const double* __restrict__ xp,
                                                        const double* __restrict__ jap, int64_t MP, int D,
                                                        double* __restrict__ nX, double* __restrict__ cX) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= MP) return;
  double s = 0.0, c = 0.0;
  for (int k = lane; k < D; k += 64) {
    const double x = xp[r * D + k];
    s += x * x;
    c += x * jap[r * D + k];
  }
  s = wave_sum(s);
  c = wave_sum(c);
  if (lane == 0) {
    nX[r] = s;
    cX[r] = c;
  }
}

#ifndef PMF_ABL
#define PMF_ABL 0
#endif
// The MFMAs of this kernel are written as inline asm so that the register classes are fixed: the
// accumulators and the query operand live in AccVGPRs for the whole kernel, LDS operands and the Matern
// scalars in VGPRs.  (Left to itself the allocator keeps the accumulators in VGPRs and copies each one
// to AccVGPRs and back around every chain of MFMAs, which stalls on the result of the last MFMA.)
// Hazards the compiler cannot see through the asm: operands written by a VALU instruction or an
// AccVGPR copy just before the MFMA (s_nop 1 = the 2 wait states the compiler itself inserts for VALU write -> MFMA read; it mostly issues while the previous MFMA
// still occupies the pipe, so it is free) and MFMA results read by non-MFMA code (mfma_results_ready()).
__device__ __forceinline__ void mfma_av(d4p& c, double a_vgpr, double b_agpr) {
  asm("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c) : "v"(a_vgpr), "a"(b_agpr));
}
__device__ __forceinline__ void mfma_vv(d4p& c, double a_vgpr, double b_vgpr) {
  asm("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a_vgpr), "v"(b_vgpr));
}
// MFMA result -> non-MFMA read needs software wait states (16-pass DGEMM): 3 x 16 is ample
__device__ __forceinline__ void mfma_results_ready(d4p& c0, d4p& c1) {
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" : "+a"(c0), "+a"(c1));
}
__device__ __forceinline__ void mfma_results_ready_v(d4p& c0, d4p& c1) {
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(c0), "+v"(c1));
}
#define MQ 64  // queries per workgroup
#define MR 16  // table rows per tile

template <int NT>  // NT = number of 16-column tiles covering D
__global__ void __launch_bounds__(256, 1) predict_mfma_kernel(PredArgs A, const double* __restrict__ nX,
                                                             const double* __restrict__ cX) {
  extern __shared__ __attribute__((aligned(16))) double lds[];  // 2 x (X tile | JA tile), each MR x P
  constexpr int P = 16 * NT + 2;  // pitch: conflict-free for the phase-1 operand pattern
  constexpr int KS = 4 * NT;      // k-steps of 4
  constexpr int TILE = MR * P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int D = A.D;
  const int64_t q0w = (int64_t)blockIdx.x * MQ + 16 * wave;
  const int64_t myq = q0w + li;
  const int split = blockIdx.y;
  const int64_t r_beg = (int64_t)split * A.rows_per_split;
  const int64_t r_end = (r_beg + A.rows_per_split < A.MP) ? r_beg + A.rows_per_split : A.MP;
  const double sig = A.sig, inv_sig = 1.0 / sig;
  const double sqrt5 = 2.23606797749978969641;
  const double fact = 5.0 / (3.0 * sig * sig * sig);
  const double dscale = 5.0 / sig;
  const double inv_3sig = 1.0 / (3.0 * sig);
  const bool has_aE = A.aE != nullptr;

  for (int i = tid; i < 4 * TILE; i += 256) lds[i] = 0.0;  // pad columns must be finite (x is 0 there)

  // query operand: lane (li, lk) holds x[q = li][4 ks + lk]  (B operand of phase 1)
  double xA[KS];
  double nx = 0.0;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int k = 4 * ks + lk;
    const double v = (myq < A.B && k < D) ? A.xq[myq * D + k] : 0.0;
    xA[ks] = v;
    nx += v * v;
  }
  nx += __shfl_xor(nx, 16, 64);
  nx += __shfl_xor(nx, 32, 64);

  d4p acc2[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc2[t] = (d4p){0.0, 0.0, 0.0, 0.0};
  double w1s = 0.0, es = 0.0;

  // staging: a row tile is one contiguous, 16-byte aligned block of MR * D doubles (MR and the split
  // starts are even); thread handles the double2 elements tid + 256 s
  typedef double d2p __attribute__((ext_vector_type(2)));
  constexpr int NS2 = NT / 2;  // 8 D / 256 <= NT / 2 double2 per thread
  const int nelem2 = MR * D / 2;
  d2p px[NS2], pj[NS2];
  const d2p* gx = nullptr;
  const d2p* gj = nullptr;
  int64_t lim2 = 0;
  auto issue_begin = [&](int64_t r0, bool valid) {  // valid == false: harmless reload of an existing tile
    lim2 = valid ? (r_end - r0) * D : 0;             // elements of this tile that exist
    gx = reinterpret_cast<const d2p*>(A.xp + r0 * D);
    gj = reinterpret_cast<const d2p*>(A.jap + r0 * D);
  };
  auto issue_step = [&](int s) {
    const int idx = tid + 256 * s;
    const bool ok0 = idx < nelem2 && 2 * idx < lim2, ok1 = idx < nelem2 && 2 * idx + 1 < lim2;
    const int idc = ok0 ? idx : 0;  // unconditional load from a valid address, then select; when the
    const d2p vx = gx[idc], vj = gj[idc];  // table ends on an odd element the pair reads the 8-byte pad
    px[s] = (d2p){ok0 ? vx[0] : 0.0, ok1 ? vx[1] : 0.0};
    pj[s] = (d2p){ok0 ? vj[0] : 0.0, ok1 ? vj[1] : 0.0};
  };
  int c_row = 0, c_k = 0;
  const int drow = 512 / D, dk = 512 - drow * D;
  auto commit_begin = [&]() {
    int e0 = 2 * tid;  // element index inside the tile
    asm volatile("" : "+v"(e0));  // keep the address arithmetic inside the loop (cheaper than spilling it)
    c_row = e0 / D;
    c_k = e0 - c_row * D;
  };
  auto commit_step = [&](int s, double* Xd) {
    if (tid + 256 * s < nelem2) {
      const int o0 = c_row * P + c_k;
      const int o1 = (c_k + 1 < D) ? o0 + 1 : o0 + 1 + (P - D);  // second element may start the next row
      Xd[o0] = px[s][0];
      Xd[o1] = px[s][1];
      Xd[TILE + o0] = pj[s][0];
      Xd[TILE + o1] = pj[s][1];
    }
    c_row += drow;
    c_k += dk;
    if (c_k >= D) {
      c_k -= D;
      c_row += 1;
    }
  };
  __syncthreads();
  issue_begin(r_beg, true);
#pragma unroll
  for (int st = 0; st < NS2; ++st) issue_step(st);
  commit_begin();
#pragma unroll
  for (int st = 0; st < NS2; ++st) commit_step(st, lds);
  __syncthreads();

  int cur = 0;
  for (int64_t r0 = r_beg; r0 < r_end; r0 += MR) {
    const bool has_next = r0 + MR < r_end;
    issue_begin(has_next ? r0 + MR : r0, has_next);  // steps are interleaved with the phase-1 MFMAs
    // row terms of the lane's 4 pairs (row = r0 + lk + 4 rr)
    double nXr[4], cXr[4], aer[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int64_t r = r0 + lk + 4 * rr;
      const bool rok = r < r_end;
      nXr[rr] = rok ? nX[r] : 0.0;
      cXr[rr] = rok ? cX[r] : 0.0;
      aer[rr] = (rok && has_aE) ? A.aE[r] : 0.0;
    }
    const double* Xs = lds + cur * 2 * TILE;
    const double* Js = Xs + TILE;
    // ---------------- phase 1: P^T[row][query]
    d4p p1 = (d4p){0.0, 0.0, 0.0, 0.0}, p2 = (d4p){0.0, 0.0, 0.0, 0.0};
    {
      const double* xa = Xs + li * P + lk;
      const double* ja = Js + li * P + lk;
      constexpr int KSE = (PMF_ABL == 3 ? 4 : KS);  // ablation 3: 4 k-steps only
      constexpr int NG = KSE / 4;                   // groups of 4 k-steps, operands read one group ahead
      double oa[2][4], oj[2][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        oa[0][u] = xa[4 * u];
        oj[0][u] = ja[4 * u];
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            oa[(g + 1) & 1][u] = xa[16 * (g + 1) + 4 * u];
            oj[(g + 1) & 1][u] = ja[16 * (g + 1) + 4 * u];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          mfma_av(p1, oa[g & 1][u], xA[4 * g + u]);
          mfma_av(p2, oj[g & 1][u], xA[4 * g + u]);
        }
#if PMF_ABL != 4
        if ((g & 1) == 0 && g / 2 < NS2) {  // next tile's global loads ride in the MFMA shadow
          issue_step(g / 2);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x026, 3, 0);
          }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    mfma_results_ready(p1, p2);
    // ---------------- Matern scalars for (row = r0 + lk + 4 rr, query = li)
    double w1[4], b2[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const bool rok = r0 + lk + 4 * rr < r_end;
      double s2 = nx + nXr[rr] - 2.0 * p1[rr];
      s2 = s2 > 0.0 ? s2 : 0.0;
      const double sa = p2[rr] - cXr[rr];
#if PMF_ABL == 1  // ablation: no sqrt / exp
      const double nrm = sqrt5 * s2;
      const double ex = 1.0 - nrm * inv_sig;
#else
      const double nrm = sqrt5 * sqrt(s2);
      const double ex = exp(-nrm * inv_sig);
#endif
      const double b = fact * ex;
      double b2v = b * (nrm + sig);
      double w1v = dscale * sa * b;
      double e = sa * b2v;
      if (has_aE) {
        w1v += aer[rr] * b2v;
        e += aer[rr] * (1.0 + (nrm * inv_sig) * (1.0 + nrm * inv_3sig)) * ex;
      }
      w1[rr] = rok ? w1v : 0.0;
      b2[rr] = rok ? b2v : 0.0;
      w1s += w1[rr];
      es += rok ? e : 0.0;
    }
    // ---------------- phase 2: acc2[query][k] += [w1 | b2] [X ; JA]
    {
      constexpr int NTE = (PMF_ABL == 2 ? 1 : NT);  // ablation 2: one column tile only
      const double* xb = Xs + lk * P + li;
      const double* jb = Js + lk * P + li;
      double ox[2][4], oz[2][4];  // operands of one column tile, read one tile ahead
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ox[0][u] = xb[4 * u * P];
        oz[0][u] = jb[4 * u * P];
      }
#pragma unroll
      for (int t = 0; t < NTE; ++t) {
        if (t + 1 < NTE) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            ox[(t + 1) & 1][u] = xb[4 * u * P + 16 * (t + 1)];
            oz[(t + 1) & 1][u] = jb[4 * u * P + 16 * (t + 1)];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          mfma_vv(acc2[t], w1[u], ox[t & 1][u]);
          mfma_vv(acc2[t], b2[u], oz[t & 1][u]);
        }
#if PMF_ABL != 4
        if ((t & 1) == 0 && t / 2 < NS2) {  // LDS writes of the next tile ride in the MFMA shadow
          if (t == 0) commit_begin();
          commit_step(t / 2, lds + (cur ^ 1) * 2 * TILE);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x206, 6, 0);
          }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  mfma_results_ready_v(acc2[NT - 1], acc2[NT - 2]);
  // ---- per-query totals live in lanes by li; the accumulators hold query lk + 4 rr
  w1s += __shfl_xor(w1s, 16, 64);
  w1s += __shfl_xor(w1s, 32, 64);
  es += __shfl_xor(es, 16, 64);
  es += __shfl_xor(es, 32, 64);
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int qsrc = lk + 4 * rr;
    const double wt = __shfl(w1s, qsrc, 64);
    const int64_t qi = q0w + qsrc;
    if (qi < A.B) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int k = 16 * t + li;
        if (k < D) A.part_F[((int64_t)split * A.B + qi) * D + k] = A.xq[qi * D + k] * wt - acc2[t][rr];
      }
    }
  }
  if (lk == 0 && myq < A.B) A.part_E[(int64_t)split * A.B + myq] = es;
}

// One workgroup per query: sum the split partials in order, then F = J_x^T F_x.
// LDSFX = false (D beyond the LDS row, JS = 1): F_x is read where the GEMM pipeline left it.
template <bool LDSFX>
__global__ void __launch_bounds__(256) predict_epilogue_kernel(const double* __restrict__ part_F,
                                                               const double* __restrict__ part_E,
                                                               const double* __restrict__ gq,
                                                               int64_t B, int N, int D, int JS,
                                                               double* __restrict__ E_out,
                                                               double* __restrict__ F_out) {
  extern __shared__ __attribute__((aligned(16))) double fxs[];
  const int64_t q = blockIdx.x;
  const int tid = threadIdx.x, T = blockDim.x;
  const double* fx = LDSFX ? fxs : part_F + q * D;
  // Loads in batches of 8 with clamped indices, sums in the original order: as plain loops every partial (JS of them: 62 for
  // a single query against 1000 training points) and every G entry of the back-projection (N - 1 per output) was one load
  // + s_waitcnt vmcnt(0) -- ~100 dependent memory round trips in the single-geometry latency path.
  if (LDSFX) {
    for (int k = tid; k < D; k += T) {
      double s = 0.0;
      for (int sp0 = 0; sp0 < JS; sp0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int sp = sp0 + u < JS ? sp0 + u : JS - 1;
          v[u] = part_F[((int64_t)sp * B + q) * D + k];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (sp0 + u < JS) s += v[u];
      }
      fxs[k] = s;
    }
  }
  if (tid == 0 && E_out) {
    double s = 0.0;
    for (int sp0 = 0; sp0 < JS; sp0 += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part_E[(int64_t)(sp0 + u < JS ? sp0 + u : JS - 1) * B + q];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (sp0 + u < JS) s += v[u];
    }
    E_out[q] = s;
  }
  __syncthreads();
  const double* g = gq + q * 3 * D;
  for (int t = tid; t < 3 * N; t += T) {
    const int a = t / 3, al = t - 3 * a;
    const int other = a == 0 ? 1 : 0;  // any partner != a, for the clamped slots
    double s = 0.0;
    for (int m0 = 0; m0 < N && N >= 2; m0 += 8) {  // (a single atom has no descriptor entries: nothing to read)
      double gv[8], fv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int mc = m0 + u < N ? m0 + u : N - 1;
        const int k = pair_idx(a, mc == a ? other : mc);
        gv[u] = g[k * 3 + al];
        fv[u] = fx[k];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = m0 + u;
        if (m < N && m != a) {
          const double v = gv[u] * fv[u];
          s += (a < m) ? v : -v;
        }
      }
    }
    F_out[q * 3 * N + t] = s;
  }
}

template <int KPL, int QB>
static void launch_pred(gdml_ctx* ctx, const PredArgs& A) {
  int64_t n_qt = (A.B + QB - 1) / QB;
  int64_t waves = n_qt * A.JS;
  hipLaunchKernelGGL((predict_kernel<KPL, QB>), dim3((unsigned)ceil_div(waves, 4)), dim3(256), 0,
                     ctx->stream, A);
}

template <int KPL>
static void dispatch_qb(gdml_ctx* ctx, const PredArgs& A, int QB) {
  constexpr int MAXQB = KPL <= 4 ? 8 : (KPL == 8 ? 4 : (KPL == 16 ? 2 : 1));
  if (QB > MAXQB) QB = MAXQB;
  if constexpr (MAXQB >= 8) if (QB == 8) return launch_pred<KPL, 8>(ctx, A);
  if constexpr (MAXQB >= 4) if (QB >= 4) return launch_pred<KPL, 4>(ctx, A);
  if constexpr (MAXQB >= 2) if (QB >= 2) return launch_pred<KPL, 2>(ctx, A);
  return launch_pred<KPL, 1>(ctx, A);
}

static int max_qb_for(int KPL) { return KPL <= 4 ? 8 : (KPL == 8 ? 4 : (KPL == 16 ? 2 : 1)); }

int predict_device(gdml_ctx* ctx, const double* d_xq, const double* d_gq, int64_t B, double* d_E,
                   double* d_F) {
  Model& md = ctx->model;
  if (!md.xp) return gdml_fail(ctx, GDML_ERR_STATE, "predict: no model resident");
  if (B == 0) return GDML_OK;
  const int D = md.D, N = md.N;
  const int64_t MP = md.M * md.P;
  int KPL = 1;
  while (KPL * 64 < D) KPL <<= 1;
  const bool big = D > 1024;  // beyond the register-resident wave kernel
  // below ~1e9 (row, query, descriptor) triples the seven launches of the pipeline cost more than the wave kernel
  // (tools/predict_wide_probe.py); option predict.mfma_wide = 2 forces it (tests)
  const int wide_opt = ctx_opt_i(ctx, "predict.mfma_wide", 1);
  // D > 8192 (N > 128): beyond the register rows of predict_big_kernel<16> -- the GEMM pipeline has no size limit and
  // takes every batch there (a single query too: correct, if not what the pipeline is tuned for)
  const bool beyond = D > 8192;
  const bool wide = beyond || ((B >= 256) && (D > 256) && !ctx_opt_i(ctx, "predict.wave_only", 0) &&
                               (wide_opt == 2 || (wide_opt == 1 && (double)MP * (double)B * (double)D >= 1.0e9)));
  if (wide) {  // large molecules: the contractions as tiled fp64-MFMA GEMMs (predict_wide.hip)
    double* part;
    GDML_TRY(ctx_slot(ctx, 0, (B * (int64_t)D + B) * 8, &part));
    const int slot = ktime_begin(ctx);
    GDML_TRY(predict_wide_device(ctx, d_xq, B, part, part + B * (int64_t)D));
    ktime_end(ctx, slot, "predict", 10.0 * (double)D * (double)B * (double)MP);
    if (beyond)  // F_x row longer than the LDS row of the epilogue: read in place (JS = 1)
      hipLaunchKernelGGL(predict_epilogue_kernel<false>, dim3((unsigned)B), dim3(256), 0, ctx->stream, part,
                         part + B * (int64_t)D, d_gq, B, N, D, 1, d_E, d_F);
    else
      hipLaunchKernelGGL(predict_epilogue_kernel<true>, dim3((unsigned)B), dim3(256), (size_t)D * 8, ctx->stream, part,
                         part + B * (int64_t)D, d_gq, B, N, D, 1, d_E, d_F);
    ctx->launch_counter++;
    HIP_CHECK(ctx, hipGetLastError());
    return GDML_OK;
  }
  const bool bulk = !big && (B >= 256) && (D <= 256) && !ctx_opt_i(ctx, "predict.wave_only", 0);
  const bool mfma = bulk && ctx_opt_i(ctx, "predict.mfma", 1);
  int QB = max_qb_for(KPL);
  while (QB > 1 && B < QB) QB >>= 1;  // do not waste query slots
  {
    // small batches: fewer queries per wavefront until the grid (query tiles x row splits) fills the chip
    const int fill = ctx_opt_i(ctx, "predict.fill", 1024);
    const int64_t js_cap = MP / 16 > 1 ? MP / 16 : 1;
    while (QB > 1 && ((B + QB - 1) / QB) * js_cap < fill) QB >>= 1;
  }
  int64_t n_qt = mfma ? (B + MQ - 1) / MQ : bulk ? (B + PQ - 1) / PQ : (B + QB - 1) / QB;
  int64_t JS = ((mfma ? 512 : bulk ? 2048 : 4096) + n_qt - 1) / n_qt;
  int64_t max_js = bulk ? (MP / 64 > 1 ? MP / 64 : 1) : (MP / 16 > 1 ? MP / 16 : 1);
  if (big) {  // one workgroup per (query, split)
    n_qt = B;
    JS = (1024 + B - 1) / B;
    max_js = MP / 4 > 1 ? MP / 4 : 1;
  }
  if (JS > max_js) JS = max_js;
  if (JS < 1) JS = 1;
  if (mfma) {
    // one workgroup per CU: pick the split count that minimises (rounds of workgroups) x (row tiles per
    // workgroup + fixed per-workgroup cost of ~3 tiles)
    const int64_t ncu = ctx->num_cus;
    int64_t best = 1, best_cost = INT64_MAX;
    for (int64_t js = 1; js <= 64 && js <= max_js; ++js) {
      const int64_t rows = (((MP + js - 1) / js) + MR - 1) / MR * MR;
      const int64_t js_eff = (MP + rows - 1) / rows;
      const int64_t rounds = (n_qt * js_eff + ncu - 1) / ncu;
      const int64_t cost = rounds * (rows / MR + 3);
      if (cost < best_cost) {
        best_cost = cost;
        best = js;
      }
    }
    JS = best;
  }
  int64_t rps = (MP + JS - 1) / JS;
  if (bulk) rps = (rps + PR - 1) / PR * PR;
  JS = (MP + rps - 1) / rps;

  int64_t need = (JS * B * (int64_t)D + JS * B) * 8;
  double* part;
  GDML_TRY(ctx_slot(ctx, 0, need, &part));

  PredArgs A;
  A.xq = d_xq; A.xp = md.xp; A.jap = md.jap; A.aE = md.has_aE ? md.aE : nullptr;
  A.B = B; A.MP = MP; A.D = D; A.sig = md.sig; A.JS = (int)JS; A.rows_per_split = rps;
  A.part_F = part; A.part_E = part + JS * B * (int64_t)D;
  const int slot = ktime_begin(ctx);
  if (mfma) {
    double* stats;
    GDML_TRY(ctx_slot(ctx, 4, 2 * MP * 8, &stats));
    hipLaunchKernelGGL(row_stats_kernel, dim3(ceil_div(MP, 4)), dim3(256), 0, ctx->stream, md.xp, md.jap, MP,
                       D, stats, stats + MP);
    dim3 grid((unsigned)n_qt, (unsigned)JS);
    const int nt = 2 * ((D + 31) / 32);  // even number of 16-column tiles
    const size_t lds_bytes = (size_t)4 * MR * (16 * nt + 2) * 8;
    switch (nt) {
#define MC(v)                                                                                         \
  case v:                                                                                             \
    hipFuncSetAttribute((const void*)predict_mfma_kernel<v>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                        (int)lds_bytes);                                                              \
    hipLaunchKernelGGL(predict_mfma_kernel<v>, grid, dim3(256), lds_bytes, ctx->stream, A, stats,      \
                       stats + MP);                                                                   \
    break;
      MC(2) MC(4) MC(6) MC(8) MC(10) MC(12) MC(14) MC(16)
#undef MC
      default: return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "predict: bad tile count");
    }
    ctx->launch_counter++;
  } else if (bulk) {
    dim3 grid((unsigned)n_qt, (unsigned)JS);
    const int nch = (D + 31) / 32;
    switch (nch) {
#define BC(v) case v: hipLaunchKernelGGL(predict_bulk_kernel<v>, grid, dim3(256), 0, ctx->stream, A); break;
      BC(1) BC(2) BC(3) BC(4) BC(5) BC(6) BC(7) default: hipLaunchKernelGGL(predict_bulk_kernel<8>, grid, dim3(256), 0, ctx->stream, A); break;
#undef BC
    }
  } else if (big) {
    dim3 grid((unsigned)B, (unsigned)JS);
    if (D <= 4096) hipLaunchKernelGGL(predict_big_kernel<8>, grid, dim3(512), 0, ctx->stream, A);
    else hipLaunchKernelGGL(predict_big_kernel<16>, grid, dim3(512), 0, ctx->stream, A);
  } else {
    switch (KPL) {
      case 1: dispatch_qb<1>(ctx, A, QB); break;
      case 2: dispatch_qb<2>(ctx, A, QB); break;
      case 4: dispatch_qb<4>(ctx, A, QB); break;
      case 8: dispatch_qb<8>(ctx, A, QB); break;
      case 16: dispatch_qb<16>(ctx, A, QB); break;
      default: dispatch_qb<32>(ctx, A, QB); break;
    }
  }
  // algorithmic work: ~10 D flops per (query, table row)
  ktime_end(ctx, slot, "predict", 10.0 * (double)D * (double)B * (double)MP);
  ctx->launch_counter++;
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) {
    hipLaunchKernelGGL(predict_epilogue_kernel<true>, dim3((unsigned)B), dim3(256), (size_t)D * 8,
                       ctx->stream, A.part_F, A.part_E, d_gq, B, N, D, (int)JS, d_E, d_F);
    ctx->launch_counter++;
    e = hipGetLastError();
  }
  if (e != hipSuccess) return gdml_fail(ctx, GDML_ERR_HIP, "predict launch: %s", hipGetErrorString(e));
  return GDML_OK;
}

static int model_free(gdml_ctx* ctx) {
  Model& md = ctx->model;
  GDML_TRY(ctx_free(ctx, md.xp));
  GDML_TRY(ctx_free(ctx, md.jap));
  GDML_TRY(ctx_free(ctx, md.ja));
  GDML_TRY(ctx_free(ctx, md.aE));
  GDML_TRY(ctx_free(ctx, md.tp));
  md = Model();
  return GDML_OK;
}

extern "C" int gdml_predict_upload_model(gdml_ctx* ctx, const double* R_desc,
                                         const double* R_d_desc_alpha, int64_t M, int N,
                                         const int64_t* tril_perms, int P, double sig,
                                         const double* alphas_E) {
  if (!ctx) return GDML_ERR_INVALID;
  if (!R_desc || !R_d_desc_alpha || !tril_perms || M < 1 || N < 2 || N > GDML_MAX_ATOMS || P < 1 ||
      !(sig > 0))
    return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_predict_upload_model: bad arguments");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  std::vector<int32_t> h_tp, h_perm, h_pinv;
  GDML_TRY(upload_perms(ctx, tril_perms, P, N, h_tp, h_perm, h_pinv));
  GDML_TRY(model_free(ctx));
  Model& md = ctx->model;
  const int D = N * (N - 1) / 2;
  md.M = M; md.N = N; md.D = D; md.P = P; md.sig = sig;
  GDML_TRY(ctx_alloc(ctx, (void**)&md.xp, M * P * (int64_t)D * 8 + 16));  // +pad: paired loads of an odd tail
  GDML_TRY(ctx_alloc(ctx, (void**)&md.jap, M * P * (int64_t)D * 8 + 16));
  GDML_TRY(ctx_alloc(ctx, (void**)&md.ja, M * (int64_t)D * 8));
  GDML_TRY(ctx_alloc(ctx, (void**)&md.aE, M * P * 8));
  GDML_TRY(ctx_alloc(ctx, (void**)&md.tp, (int64_t)P * D * 4));
  double* tmp;
  GDML_TRY(ctx_scratch(ctx, M * (int64_t)D * 8 + M * 8, &tmp));
  HIP_CHECK(ctx, hipMemcpyAsync(md.tp, h_tp.data(), (size_t)P * D * 4, hipMemcpyHostToDevice,
                                ctx->stream));
  HIP_CHECK(ctx, hipMemcpyAsync(tmp, R_desc, M * D * 8, hipMemcpyHostToDevice, ctx->stream));
  int64_t tot = M * P * (int64_t)D;
  hipLaunchKernelGGL(permute_rows_kernel, dim3(ceil_div(tot, 256)), dim3(256), 0, ctx->stream, tmp,
                     md.tp, M, D, P, md.xp);
  HIP_CHECK(ctx, hipMemcpyAsync(md.ja, R_d_desc_alpha, M * D * 8, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(permute_rows_kernel, dim3(ceil_div(tot, 256)), dim3(256), 0, ctx->stream, md.ja,
                     md.tp, M, D, P, md.jap);
  md.has_aE = alphas_E != nullptr;
  if (alphas_E) {
    double* tE = tmp + M * (int64_t)D;
    HIP_CHECK(ctx, hipMemcpyAsync(tE, alphas_E, M * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(repeat_kernel, dim3(ceil_div(M * P, 256)), dim3(256), 0, ctx->stream, tE, M,
                       P, md.aE);
  }
  HIP_CHECK(ctx, hipGetLastError());
  HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return GDML_OK;
}

// alphas (device) -> model tables.  d_alphas_F is (M,3N), d_alphas_E (M) or null.
int set_alphas_device(gdml_ctx* ctx, const double* d_alphas_F, const double* d_alphas_E) {
  Model& md = ctx->model;
  TrainSet& ts = ctx->ts;
  if (!md.xp) return gdml_fail(ctx, GDML_ERR_STATE, "set_alphas: no model resident");
  if (!ts.g || ts.M != md.M || ts.N != md.N)
    return gdml_fail(ctx, GDML_ERR_STATE,
                     "set_alphas: training Jacobians (gdml_train_upload) missing or mismatched");
  const int64_t M = md.M;
  const int D = md.D, P = md.P;
  hipLaunchKernelGGL(jdotv_kernel, dim3(ceil_div(M * D, 256)), dim3(256), 0, ctx->stream, ts.g,
                     d_alphas_F, M, md.N, D, md.ja);
  int64_t tot = M * P * (int64_t)D;
  hipLaunchKernelGGL(permute_rows_kernel, dim3(ceil_div(tot, 256)), dim3(256), 0, ctx->stream, md.ja,
                     md.tp, M, D, P, md.jap);
  ctx->launch_counter += 2;
  md.has_aE = d_alphas_E != nullptr;
  if (d_alphas_E) {
    hipLaunchKernelGGL(repeat_kernel, dim3(ceil_div(M * P, 256)), dim3(256), 0, ctx->stream,
                       d_alphas_E, M, P, md.aE);
    ctx->launch_counter++;
  }
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}

extern "C" int gdml_set_alphas(gdml_ctx* ctx, const double* alphas_F, const double* alphas_E) {
  if (!ctx || !alphas_F) return GDML_ERR_INVALID;
  Model& md = ctx->model;
  if (!md.xp) return gdml_fail(ctx, GDML_ERR_STATE, "gdml_set_alphas: upload a model first");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int64_t n = md.M * 3 * md.N;
  double* buf;
  GDML_TRY(ctx_scratch(ctx, (n + md.M) * 8, &buf));
  HIP_CHECK(ctx, hipMemcpyAsync(buf, alphas_F, n * 8, hipMemcpyHostToDevice, ctx->stream));
  if (alphas_E)
    HIP_CHECK(ctx, hipMemcpyAsync(buf + n, alphas_E, md.M * 8, hipMemcpyHostToDevice, ctx->stream));
  GDML_TRY(set_alphas_device(ctx, buf, alphas_E ? buf + n : nullptr));
  HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return GDML_OK;
}

// ------------------------------------------------------------------------------------------
// Single launch for small HOST batches (the ASE-calculator / MD step, sgdml/intf/ase_calc.py:93-106): descriptors,
// contraction and back-projection in one kernel, geometries in through the kernel arguments, E / F out through
// host-mapped memory, completion signalled by a sequence number the host polls -- no copy command, no second and third
// launch, no stream synchronisation.  (Three launches + three copies cost 55.7 us host-to-host for ~25 us of kernels:
// profiles/r04_latency_path.txt.)
//   grid = (row shares, queries).  Every wavefront: x_q from R (kernel arguments) into registers -> its share of the table
//     rows (the row loop of predict_kernel) -> F_x, E partials of the workgroup's four wavefronts summed through LDS -> one
//     partial per workgroup;
//   the LAST workgroup of a query to finish (ticket): sums that query's partials in a fixed order, computes the Jacobian
//     entries (r_i - r_j) / d^3 from R again, F = J^T F_x, writes E, F to the mapped block; the last QUERY to finish then
//     writes the sequence number.
// B <= 8 queries, D <= 256 (N <= 23), 3 N B <= 384 coordinates (3 KB of kernel arguments: six queries at N = 21).
// ------------------------------------------------------------------------------------------
#define FUSED_MAX_COORDS 384
#define FUSED_MAX_Q 8
struct FusedArgs {
  const double* xp;
  const double* jap;
  const double* aE;
  int64_t MP;
  int D, N, B, want_E;
  double sig;
  int rows_per_wave;
  double* part;                // (B, n_wg, D + 1) workgroup partials: F_x, then E
  unsigned* counter;           // [q] workgroups of query q that have finished; [FUSED_MAX_Q] queries done
  double* out;                 // host-mapped: E (B), F (B, 3N)
  unsigned long long* flag;    // host-mapped
  unsigned long long seq;
  Lattice L;
  double R[FUSED_MAX_COORDS];  // (B, N, 3)
};
typedef const __attribute__((address_space(4))) double* kernarg_dp;

// r_i - r_j of query q (minimum image with a lattice: desc.py:44-77), from the kernel-argument copy of R
__device__ __forceinline__ void fused_pair_diff(kernarg_dp R, const Lattice& L, int N, int q, int i, int j, double (&d)[3]) {
  const int bi = (q * N + i) * 3, bj = (q * N + j) * 3;
  double d0 = R[bi] - R[bj], d1 = R[bi + 1] - R[bj + 1], d2 = R[bi + 2] - R[bj + 2];
  if (L.use) {
    const double c0 = rint(L.inv[0] * d0 + L.inv[1] * d1 + L.inv[2] * d2);
    const double c1 = rint(L.inv[3] * d0 + L.inv[4] * d1 + L.inv[5] * d2);
    const double c2 = rint(L.inv[6] * d0 + L.inv[7] * d1 + L.inv[8] * d2);
    d0 -= L.lat[0] * c0 + L.lat[1] * c1 + L.lat[2] * c2;
    d1 -= L.lat[3] * c0 + L.lat[4] * c1 + L.lat[5] * c2;
    d2 -= L.lat[6] * c0 + L.lat[7] * c1 + L.lat[8] * c2;
  }
  d[0] = d0; d[1] = d1; d[2] = d2;
}
__device__ __forceinline__ void fused_pair_of(int k, int& i, int& j) {  // k = i (i - 1) / 2 + j, i > j
  i = (int)((1.0 + sqrt(1.0 + 8.0 * (double)k)) * 0.5);
  while (i * (i - 1) / 2 > k) --i;
  while ((i + 1) * i / 2 <= k) ++i;
  j = k - i * (i - 1) / 2;
}

template <int KPL>
__global__ void __launch_bounds__(256) predict_fused_kernel(FusedArgs A) {
  __shared__ double s_fx[4][KPL * 64];  // per-wavefront F_x; reused by the query's last workgroup: [0] = its summed F_x
  __shared__ double s_E[4];
  __shared__ double s_g[KPL * 64 * 3];
  __shared__ unsigned s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = A.D, N = A.N, B = A.B;
  const int q = (int)blockIdx.y;  // a workgroup serves ONE query: grid = (row shares, queries)
  kernarg_dp Rk = (kernarg_dp)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() +
                               offsetof(FusedArgs, R));
  const double sig = A.sig, inv_sig = 1.0 / sig;
  const double sqrt5 = 2.23606797749978969641;
  const double fact = 5.0 / (3.0 * sig * sig * sig);
  const double dscale = 5.0 / sig;
  const double inv_3sig = 1.0 / (3.0 * sig);

  // ---- query descriptor (same arithmetic as desc_kernel)
  double x[KPL], Fx[KPL], E = 0.0;
#pragma unroll
  for (int t = 0; t < KPL; ++t) {
    const int k = lane + 64 * t;
    double v = 0.0;
    if (k < D) {
      int i, j;
      fused_pair_of(k, i, j);
      double d[3];
      fused_pair_diff(Rk, A.L, N, q, i, j, d);
      v = 1.0 / sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    }
    x[t] = v;
    Fx[t] = 0.0;
  }

  // ---- this wavefront's table rows (the row loop of predict_kernel)
  const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
  const int64_t r0 = gw * A.rows_per_wave;
  const int64_t r1 = (r0 + A.rows_per_wave < A.MP) ? r0 + A.rows_per_wave : A.MP;
  const bool has_aE = A.aE != nullptr;
  auto load_row = [&](int64_t r, double (&X)[KPL], double (&JA)[KPL], double& ae) {
#pragma unroll
    for (int t = 0; t < KPL; ++t) {
      const int k = lane + 64 * t;
      const int kc = k < D ? k : D - 1;
      const double xv = __hip_atomic_load(A.xp + r * D + kc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      const double jv = __hip_atomic_load(A.jap + r * D + kc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      X[t] = (k < D) ? xv : 0.0;
      JA[t] = (k < D) ? jv : 0.0;
    }
    ae = has_aE ? __hip_atomic_load(A.aE + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0.0;
  };
  auto row = [&](const double (&X)[KPL], const double (&JA)[KPL], double ae) {
    double d[KPL];
    double s2 = 0.0, sa = 0.0;
#pragma unroll
    for (int t = 0; t < KPL; ++t) {
      d[t] = x[t] - X[t];
      s2 += d[t] * d[t];
      sa += d[t] * JA[t];
    }
    s2 = wave_sum(s2);
    sa = wave_sum(sa);
    const double nrm = sqrt5 * sqrt(s2);
    const double ex = exp(-nrm * inv_sig);
    const double b = fact * ex;
    const double b2 = b * (nrm + sig);
    double w1 = dscale * sa * b;
    E += sa * b2;
    if (has_aE) {
      w1 += ae * b2;
      E += ae * (1.0 + (nrm * inv_sig) * (1.0 + nrm * inv_3sig)) * ex;
    }
#pragma unroll
    for (int t = 0; t < KPL; ++t) Fx[t] += w1 * d[t] - b2 * JA[t];
  };
  if (r0 < r1) {
    double X0[KPL], JA0[KPL], X1[KPL], JA1[KPL], ae0, ae1;
    int64_t r = r0;
    load_row(r, X0, JA0, ae0);
    for (; r + 1 < r1; r += 2) {
      load_row(r + 1, X1, JA1, ae1);
      row(X0, JA0, ae0);
      load_row(r + 2 < r1 ? r + 2 : r1 - 1, X0, JA0, ae0);
      row(X1, JA1, ae1);
    }
    if (r < r1) row(X0, JA0, ae0);
  }

  // ---- the workgroup's four wavefronts -> one partial (padding lanes hold zeros)
#pragma unroll
  for (int t = 0; t < KPL; ++t) s_fx[wave][lane + 64 * t] = Fx[t];
  if (lane == 0) s_E[wave] = E;
  __syncthreads();
  const int n_wg = (int)gridDim.x;
  double* my_part = A.part + ((int64_t)q * n_wg + blockIdx.x) * (D + 1);
  for (int k = tid; k < D; k += 256) my_part[k] = (s_fx[0][k] + s_fx[1][k]) + (s_fx[2][k] + s_fx[3][k]);
  if (tid == 0) my_part[D] = (s_E[0] + s_E[1]) + (s_E[2] + s_E[3]);
  __threadfence();
  __syncthreads();
  if (tid == 0) s_ticket = atomicAdd(A.counter + q, 1u);
  __syncthreads();
  if (s_ticket != gridDim.x - 1) return;

  // ---- the query's last workgroup: every partial of this query is visible
  __threadfence();
  double* fxs = &s_fx[0][0];
  const double* qpart = A.part + (int64_t)q * n_wg * (D + 1);
  for (int k = tid; k < D + 1; k += 256) {
    double s = 0.0;
    for (int w0 = 0; w0 < n_wg; w0 += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int w = w0 + u < n_wg ? w0 + u : n_wg - 1;
        v[u] = __hip_atomic_load(qpart + (int64_t)w * (D + 1) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (w0 + u < n_wg) s += v[u];
    }
    if (k < D) fxs[k] = s;
    else if (A.want_E) A.out[q] = s;
  }
  for (int k = tid; k < D; k += 256) {  // Jacobian entries (r_i - r_j) / d^3  (desc.py:193-205)
    int i, j;
    fused_pair_of(k, i, j);
    double d[3];
    fused_pair_diff(Rk, A.L, N, q, i, j, d);
    const double dist = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const double inv3 = 1.0 / (dist * dist * dist);
    s_g[k * 3 + 0] = d[0] * inv3;
    s_g[k * 3 + 1] = d[1] * inv3;
    s_g[k * 3 + 2] = d[2] * inv3;
  }
  __syncthreads();
  for (int t = tid; t < 3 * N; t += 256) {  // F = J_x^T F_x  (desc.py:405-408), same order as predict_epilogue_kernel
    const int a = t / 3, al = t - 3 * a;
    double s = 0.0;
    for (int m = 0; m < N; ++m) {
      if (m == a) continue;
      const int k = pair_idx(a, m);
      const double v = s_g[k * 3 + al] * fxs[k];
      s += (a < m) ? v : -v;
    }
    A.out[B + (int64_t)q * 3 * N + t] = s;
  }
  __threadfence_system();  // this query's E, F are in host memory before it is counted as done
  __syncthreads();
  if (tid == 0) {
    A.counter[q] = 0u;
    const unsigned done = atomicAdd(A.counter + FUSED_MAX_Q, 1u);
    if (done == (unsigned)B - 1) {  // the last query to finish publishes the sequence number
      __threadfence_system();
      A.counter[FUSED_MAX_Q] = 0u;
      __hip_atomic_store(A.flag, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Host side of the single-launch path.  Returns GDML_OK with *done = 0 when the shape is not served.
static int predict_fused(gdml_ctx* ctx, const double* R, int64_t B, const double* lat, const double* lat_inv, double* E_out,
                         double* F_out, int* done) {
  *done = 0;
  Model& md = ctx->model;
  const int N = md.N, D = md.D;
  if (B < 1 || B > FUSED_MAX_Q || D > 256 || D < 1 || B * 3 * N > FUSED_MAX_COORDS || ctx->profiling ||
      !ctx_opt_i(ctx, "predict.fused", 1))
    return GDML_OK;
  if (!ctx->h_map) {
    if (hipHostMalloc((void**)&ctx->h_map, 8192, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
      (void)hipGetLastError();
      ctx->h_map = nullptr;
      return GDML_OK;
    }
    memset(ctx->h_map, 0, 8192);
    if (hipHostGetDevicePointer((void**)&ctx->h_map_dev, ctx->h_map, 0) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipHostFree(ctx->h_map);
      ctx->h_map = nullptr;
      return GDML_OK;
    }
  }
  if (!ctx->d_fused_counter) {
    GDML_TRY(ctx_alloc(ctx, (void**)&ctx->d_fused_counter, 64));
    HIP_CHECK(ctx, hipMemsetAsync(ctx->d_fused_counter, 0, 64, ctx->stream));
  }
  const int64_t MP = md.M * md.P;
  int rows = ctx_opt_i(ctx, "predict.fused_rows", 16);
  if (rows < 2) rows = 2;
  // at most 4 x 256 wavefronts (a few per CU: the launch is latency bound, not throughput bound)
  while ((MP + rows - 1) / rows > 1024) rows *= 2;
  const int64_t n_waves = (MP + rows - 1) / rows;
  const int n_wg = (int)((n_waves + 3) / 4);
  double* part;
  GDML_TRY(ctx_slot(ctx, 0, (int64_t)n_wg * B * (D + 1) * 8, &part));
  FusedArgs A;
  A.xp = md.xp; A.jap = md.jap; A.aE = md.has_aE ? md.aE : nullptr;
  A.MP = MP; A.D = D; A.N = N; A.B = (int)B; A.want_E = E_out != nullptr;
  A.sig = md.sig; A.rows_per_wave = rows; A.part = part; A.counter = ctx->d_fused_counter;
  A.out = ctx->h_map_dev + 8;
  A.flag = (unsigned long long*)ctx->h_map_dev;
  A.seq = ++ctx->fused_seq;
  memset(&A.L, 0, sizeof(A.L));
  if (lat && lat_inv) {
    memcpy(A.L.lat, lat, sizeof(A.L.lat));
    memcpy(A.L.inv, lat_inv, sizeof(A.L.inv));
    A.L.use = 1;
  }
  memcpy(A.R, R, (size_t)B * 3 * N * 8);
  int KPL = 1;
  while (KPL * 64 < D) KPL <<= 1;
  const dim3 grid((unsigned)n_wg, (unsigned)B);
  switch (KPL) {
    case 1: hipLaunchKernelGGL(predict_fused_kernel<1>, grid, dim3(256), 0, ctx->stream, A); break;
    case 2: hipLaunchKernelGGL(predict_fused_kernel<2>, grid, dim3(256), 0, ctx->stream, A); break;
    default: hipLaunchKernelGGL(predict_fused_kernel<4>, grid, dim3(256), 0, ctx->stream, A); break;
  }
  ctx->launch_counter++;
  HIP_CHECK(ctx, hipGetLastError());
  // completion: the kernel's last store is the sequence number (system scope, after E and F)
  volatile unsigned long long* flag = (volatile unsigned long long*)ctx->h_map;
  bool seen = false;
  if (ctx_opt_i(ctx, "predict.fused_spin", 1)) {
    // bounded by WALL CLOCK (2 ms: the kernel takes ~20 us; a 2^26-iteration bound was seconds of busy waiting on recent
    // x86), checked every 256 polls; beyond it the stream synchronisation below takes over
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
    for (unsigned spin = 0;; ++spin) {
      if (*flag == A.seq) {
        seen = true;
        break;
      }
      if ((spin & 255u) == 255u && std::chrono::steady_clock::now() > t_end) break;
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#else
      std::this_thread::yield();
#endif
    }
  }
  if (!seen) {
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (*flag != A.seq) return gdml_fail(ctx, GDML_ERR_HIP, "single-launch prediction: the kernel finished without publishing");
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  const double* h_out = ctx->h_map + 8;
  if (E_out) memcpy(E_out, h_out, (size_t)B * 8);
  memcpy(F_out, h_out + B, (size_t)B * 3 * N * 8);
  *done = 1;
  return GDML_OK;
}

static int predict_common(gdml_ctx* ctx, const double* R, bool R_on_device, int64_t B,
                          const double* lat, const double* lat_inv, double* E_out, double* F_out,
                          bool out_on_device) {
  Model& md = ctx->model;
  if (!md.xp) return gdml_fail(ctx, GDML_ERR_STATE, "gdml_predict: upload a model first");
  if (!F_out) return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_predict: F_out is NULL");
  if ((lat == nullptr) != (lat_inv == nullptr))
    return gdml_fail(ctx, GDML_ERR_INVALID, "lattice and inverse must both be given or both NULL");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int N = md.N, D = md.D;
  if (R != nullptr && !R_on_device && !out_on_device && B >= 1 && B <= FUSED_MAX_Q) {
    int done = 0;
    GDML_TRY(predict_fused(ctx, R, B, lat, lat_inv, E_out, F_out, &done));
    if (done) return GDML_OK;
  }
  const double *d_xq, *d_gq;
  double* buf = nullptr;
  double *d_E = E_out, *d_F = F_out;
  bool use_pin = false;
  constexpr int64_t PIN_BYTES = 1 << 20;
  if (R == nullptr) {  // training-set mode (predict.py:1221-1233)
    TrainSet& ts = ctx->ts;
    if (!ts.x || ts.N != N)
      return gdml_fail(ctx, GDML_ERR_STATE,
                       "gdml_predict(R=NULL) needs the training descriptors (gdml_train_upload)");
    B = ts.M;
    d_xq = ts.x;
    d_gq = ts.g;
    if (!out_on_device) {
      GDML_TRY(ctx_scratch(ctx, (B + B * 3 * N) * 8, &buf));
      d_E = buf;
      d_F = buf + B;
    }
  } else {
    if (B < 0) return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_predict: B < 0");
    if (B == 0) return GDML_OK;
    int64_t nR = B * N * 3, nx = B * (int64_t)D, ng = nx * 3, nout = out_on_device ? 0 : B + nR;
    GDML_TRY(ctx_scratch(ctx, (nR + nx + ng + nout) * 8, &buf));
    double* d_R = buf;
    double* dx = buf + nR;
    double* dg = dx + nx;
    // small host batches (the single-geometry / MD use case) are staged through pinned memory: one truly
    // asynchronous copy in, one out, instead of three pageable copies
    use_pin = !R_on_device && !out_on_device && (nR + B + nR) * 8 <= PIN_BYTES;
    if (use_pin && !ctx->h_pin) {
      if (hipHostMalloc((void**)&ctx->h_pin, PIN_BYTES, hipHostMallocDefault) == hipSuccess)
        ctx->h_pin_bytes = PIN_BYTES;
      else
        use_pin = false;
    }
    if (R_on_device) {
      HIP_CHECK(ctx, hipMemcpyAsync(d_R, R, nR * 8, hipMemcpyDeviceToDevice, ctx->stream));
    } else if (use_pin) {
      memcpy(ctx->h_pin, R, nR * 8);
      HIP_CHECK(ctx, hipMemcpyAsync(d_R, ctx->h_pin, nR * 8, hipMemcpyHostToDevice, ctx->stream));
    } else {
      HIP_CHECK(ctx, hipMemcpyAsync(d_R, R, nR * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    if (!out_on_device) {
      d_E = dg + ng;
      d_F = d_E + B;
    }
    phase_begin(ctx);
    GDML_TRY(desc_device(ctx, R_on_device ? R : d_R, B, N, lat, lat_inv, dx, dg));
    d_xq = dx;
    d_gq = dg;
  }
  if (R == nullptr) phase_begin(ctx);
  GDML_TRY(predict_device(ctx, d_xq, d_gq, B, (E_out || !out_on_device) ? d_E : nullptr, d_F));
  GDML_TRY(phase_end(ctx, "predict"));
  if (!out_on_device) {
    if (use_pin) {  // E and F are adjacent in the work buffer: one copy
      double* h_out = ctx->h_pin + B * 3 * N;
      HIP_CHECK(ctx, hipMemcpyAsync(h_out, d_E, (B + B * 3 * N) * 8, hipMemcpyDeviceToHost, ctx->stream));
      HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      if (E_out) memcpy(E_out, h_out, B * 8);
      memcpy(F_out, h_out + B, B * 3 * N * 8);
    } else {
      if (E_out) HIP_CHECK(ctx, hipMemcpyAsync(E_out, d_E, B * 8, hipMemcpyDeviceToHost, ctx->stream));
      HIP_CHECK(ctx, hipMemcpyAsync(F_out, d_F, B * 3 * N * 8, hipMemcpyDeviceToHost, ctx->stream));
      HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  return GDML_OK;
}

extern "C" int gdml_predict(gdml_ctx* ctx, const double* R, int64_t B, const double* lat,
                            const double* lat_inv, double* E_out, double* F_out) {
  if (!ctx) return GDML_ERR_INVALID;
  return predict_common(ctx, R, false, B, lat, lat_inv, E_out, F_out, false);
}

extern "C" int gdml_predict_dev(gdml_ctx* ctx, const double* R_dev, int64_t B, const double* lat,
                                const double* lat_inv, double* E_dev, double* F_dev) {
  if (!ctx) return GDML_ERR_INVALID;
  if (!R_dev) return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_predict_dev: R_dev is NULL");
  return predict_common(ctx, R_dev, true, B, lat, lat_inv, E_dev, F_dev, true);
}

// ------------------------------------------------------------------------------------------
// Test / validation error sums on the device (the reference's cli.test loop, sgdml/cli.py:1564-1605,
// _online_err :1170): predictions stay in HBM, only eight sums come back.
//   sums[0..1] = sum |dE|, sum dE^2                (0 if E_ref is NULL)
//   sums[2..3] = sum |dF|, sum dF^2                over all 3N B force components
//   sums[4..5] = sum |d|F||, sum (d|F|)^2          per-atom force magnitudes
//   sums[6..7] = sum a, sum a^2, a = arccos(clip(f_pred.f_ref/(|f_pred||f_ref|)))/pi per atom
// One thread per (geometry, atom); per-block partials are reduced in a fixed order.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) err_part_kernel(const double* __restrict__ E, const double* __restrict__ F,
                                                       const double* __restrict__ E_ref,
                                                       const double* __restrict__ F_ref, int64_t B, int N,
                                                       double std, double c, double* __restrict__ part) {
  __shared__ double red[8][4];
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (t < B * N) {
    const int64_t q = t / N;
    const int a = (int)(t - q * N);
    const double* fp = F + (q * N + a) * 3;
    const double* fr = F_ref + (q * N + a) * 3;
    double p0 = fp[0] * std, p1 = fp[1] * std, p2 = fp[2] * std;
    double d0 = fr[0] - p0, d1 = fr[1] - p1, d2 = fr[2] - p2;
    v[2] = fabs(d0) + fabs(d1) + fabs(d2);
    v[3] = d0 * d0 + d1 * d1 + d2 * d2;
    const double mp = sqrt(p0 * p0 + p1 * p1 + p2 * p2), mr = sqrt(fr[0] * fr[0] + fr[1] * fr[1] + fr[2] * fr[2]);
    const double dm = mp - mr;
    v[4] = fabs(dm);
    v[5] = dm * dm;
    double cs = (p0 * fr[0] + p1 * fr[1] + p2 * fr[2]) / (mp * mr);
    cs = fmin(1.0, fmax(-1.0, cs));
    const double ang = acos(cs) * 0.31830988618379067154;  // / pi
    v[6] = ang;  // >= 0
    v[7] = ang * ang;
    if (a == 0 && E_ref != nullptr) {
      const double de = E_ref[q] - (E[q] * std + c);
      v[0] = fabs(de);
      v[1] = de * de;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double s = wave_sum(v[k]);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int k = threadIdx.x;
    part[(int64_t)blockIdx.x * 8 + k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
  }
}
__global__ void __launch_bounds__(64) err_reduce_kernel(const double* __restrict__ part, int64_t nblocks,
                                                        double* __restrict__ out) {
  const int k = threadIdx.x;
  if (k >= 8) return;
  double s = 0.0;
  for (int64_t b = 0; b < nblocks; ++b) s += part[b * 8 + k];
  out[k] = s;
}

extern "C" int gdml_predict_errors(gdml_ctx* ctx, const double* R, int64_t B, const double* lat,
                                   const double* lat_inv, double std, double c, const double* E_ref,
                                   const double* F_ref, double* sums8_out) {
  if (!ctx) return GDML_ERR_INVALID;
  if (!R || !F_ref || !sums8_out || B < 1)
    return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_predict_errors: bad arguments");
  Model& md = ctx->model;
  if (!md.xp) return gdml_fail(ctx, GDML_ERR_STATE, "gdml_predict_errors: upload a model first");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int N = md.N;
  const int64_t nF = B * 3 * N;
  const int64_t nblocks = (B * N + 255) / 256;
  void* buf = nullptr;
  GDML_TRY(ctx_alloc(ctx, &buf, (3 * nF + 2 * B + nblocks * 8 + 8) * 8));
  double* dR = (double*)buf;
  double* dFref = dR + nF;
  double* dF = dFref + nF;
  double* dEref = dF + nF;
  double* dE = dEref + B;
  double* dpart = dE + B;
  double* dout = dpart + nblocks * 8;
  int rc = GDML_OK;
  hipError_t e = hipMemcpyAsync(dR, R, nF * 8, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dFref, F_ref, nF * 8, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && E_ref) e = hipMemcpyAsync(dEref, E_ref, B * 8, hipMemcpyHostToDevice, ctx->stream);
  if (e != hipSuccess) rc = gdml_fail(ctx, GDML_ERR_HIP, "H2D: %s", hipGetErrorString(e));
  if (rc == GDML_OK) rc = predict_common(ctx, dR, true, B, lat, lat_inv, dE, dF, true);
  if (rc == GDML_OK) {
    hipLaunchKernelGGL(err_part_kernel, dim3((unsigned)nblocks), dim3(256), 0, ctx->stream, dE, dF,
                       E_ref ? dEref : nullptr, dFref, B, N, std, c, dpart);
    hipLaunchKernelGGL(err_reduce_kernel, dim3(1), dim3(64), 0, ctx->stream, dpart, nblocks, dout);
    e = hipMemcpyAsync(sums8_out, dout, 64, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = gdml_fail(ctx, GDML_ERR_HIP, "errors: %s", hipGetErrorString(e));
  }
  int rc2 = ctx_free(ctx, buf);
  return rc != GDML_OK ? rc : rc2;
}

// out = K v - lam v  via set_alphas + training-set prediction (iterative.py:183-204).
// d_v, d_out device vectors of length n = 3NM (+M).  Uses ctx->model built on the training set.
__global__ void __launch_bounds__(256) matvec_finish_kernel(const double* __restrict__ F,
                                                            const double* __restrict__ E,
                                                            const double* __restrict__ v, int64_t nF,
                                                            int64_t nE, double lam,
                                                            double* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nF)
    out[t] = F[t] - lam * v[t];
  else if (t < nF + nE)
    out[t] = -E[t - nF] - lam * v[t];
}

// coefficients in the reference order from a vector in the sharded layout (VecLayout::pos)
__global__ void __launch_bounds__(256) vec_to_ref_kernel(const double* __restrict__ v, VecLayout L, double* __restrict__ ref) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g < L.n) ref[g] = v[L.pos(g)];
}

int matvec_device(gdml_ctx* ctx, double lam, int use_E_cstr, const double* d_v, int64_t n,
                  double* d_out) {
  Model& md = ctx->model;
  TrainSet& ts = ctx->ts;
  if (!md.xp || !ts.x || md.M != ts.M)
    return gdml_fail(ctx, GDML_ERR_STATE, "kernel_matvec: training set / operator model not resident");
  const int64_t M = ts.M, N3 = 3 * (int64_t)ts.N, nF = M * N3, nE = use_E_cstr ? M : 0;
  if (n != nF + nE) return gdml_fail(ctx, GDML_ERR_INVALID, "kernel_matvec: n mismatch");
  // query shard of this rank (all training points when there is no communicator); d_v / d_out are
  // replicated vectors in the layout of VecLayout (common.h): padded to world * chunk doubles when sharded, and with energy
  // constraints rank-major -- the coefficients are brought back into the reference order for set_alphas, the rank's outputs
  // (forces of its query points, then their energies) are exactly its contiguous chunk
  const VecLayout L = vec_layout(ctx, use_E_cstr);
  if (L.two_seg) {
    double* vref;
    GDML_TRY(ctx_slot(ctx, 12, n * 8, &vref));
    hipLaunchKernelGGL(vec_to_ref_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, d_v, L, vref);
    GDML_TRY(set_alphas_device(ctx, vref, vref + nF));
  } else {
    GDML_TRY(set_alphas_device(ctx, d_v, use_E_cstr ? d_v + nF : nullptr));
  }
  int64_t p0 = 0, p1 = M, per = M;
  if (ctx->world > 1) shard_points(ctx, M, &p0, &p1, &per);
  const int64_t B = p1 - p0;
  double* dF;
  GDML_TRY(ctx_slot(ctx, 1, (B * N3 + B + 8) * 8, &dF));
  double* dE = dF + B * N3;
  if (B > 0) {
    GDML_TRY(predict_device(ctx, ts.x + p0 * ts.D, ts.g + p0 * ts.D * 3, B, use_E_cstr ? dE : nullptr, dF));
    if (ctx->world > 1) {
      hipLaunchKernelGGL(matvec_finish_kernel, dim3(ceil_div(L.n_loc, 256)), dim3(256), 0, ctx->stream, dF,
                         dE, d_v + L.row0, B * N3, use_E_cstr ? B : (int64_t)0, lam, d_out + L.row0);
    } else {
      hipLaunchKernelGGL(matvec_finish_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, dF, dE,
                         d_v, nF, nE, lam, d_out);
    }
    ctx->launch_counter++;
    HIP_CHECK(ctx, hipGetLastError());
  }
  // with a communicator the gather is issued for every world size (a one-rank communicator runs the same call)
  if (ctx->world > 1 || comm_active(ctx)) GDML_TRY(comm_allgather_inplace(ctx, d_out, L.chunk));
  return GDML_OK;
}

// Build the prediction model used as the matrix-free kernel operator directly from the resident
// training set (what Iterative._init_kernel_operator does with a dummy model,
// sgdml/solvers/iterative.py:150-167): permuted descriptor table, zero coefficients.
int operator_model_from_trainset(gdml_ctx* ctx, double sig) {
  TrainSet& ts = ctx->ts;
  if (!ts.x) return gdml_fail(ctx, GDML_ERR_STATE, "kernel operator: no training set resident");
  Model& md = ctx->model;
  const int64_t M = ts.M;
  const int D = ts.D, P = ts.P;
  if (!(md.xp && md.M == M && md.N == ts.N && md.P == P)) {
    GDML_TRY(model_free(ctx));
    md.M = M; md.N = ts.N; md.D = D; md.P = P;
    GDML_TRY(ctx_alloc(ctx, (void**)&md.xp, M * P * (int64_t)D * 8 + 16));  // +pad: paired loads of an odd tail
    GDML_TRY(ctx_alloc(ctx, (void**)&md.jap, M * P * (int64_t)D * 8 + 16));
    GDML_TRY(ctx_alloc(ctx, (void**)&md.ja, M * (int64_t)D * 8));
    GDML_TRY(ctx_alloc(ctx, (void**)&md.aE, M * P * 8));
    GDML_TRY(ctx_alloc(ctx, (void**)&md.tp, (int64_t)P * D * 4));
  }
  md.sig = sig;
  md.has_aE = false;
  HIP_CHECK(ctx, hipMemcpyAsync(md.tp, ts.tp, (size_t)P * D * 4, hipMemcpyDeviceToDevice, ctx->stream));
  int64_t tot = M * P * (int64_t)D;
  hipLaunchKernelGGL(permute_rows_kernel, dim3(ceil_div(tot, 256)), dim3(256), 0, ctx->stream, ts.x,
                     md.tp, M, D, P, md.xp);
  HIP_CHECK(ctx, hipMemsetAsync(md.jap, 0, tot * 8, ctx->stream));
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}

extern "C" int gdml_kernel_matvec(gdml_ctx* ctx, double lam, int use_E_cstr, const double* v,
                                  int64_t n, double* out) {
  if (!ctx || !v || !out) return GDML_ERR_INVALID;
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!ctx->ts.x) return gdml_fail(ctx, GDML_ERR_STATE, "kernel_matvec: training set / operator model not resident");
  const VecLayout L = vec_layout(ctx, use_E_cstr);
  if (n != L.n) return gdml_fail(ctx, GDML_ERR_INVALID, "kernel_matvec: n mismatch");
  const int64_t n_pad = L.n_pad;
  void* buf = nullptr;
  GDML_TRY(ctx_alloc(ctx, &buf, 2 * n_pad * 8));
  double* dv = (double*)buf;
  double* dout = dv + n_pad;
  int rc = GDML_OK;
  hipError_t e = hipMemsetAsync(buf, 0, 2 * n_pad * 8, ctx->stream);
  if (e != hipSuccess) rc = gdml_fail(ctx, GDML_ERR_HIP, "memset: %s", hipGetErrorString(e));
  if (rc == GDML_OK) rc = vec_upload(ctx, L, v, dv);
  if (rc == GDML_OK) {
    phase_begin(ctx);
    rc = matvec_device(ctx, lam, use_E_cstr, dv, n, dout);
    if (rc == GDML_OK) rc = phase_end(ctx, "matvec");
  }
  if (rc == GDML_OK) rc = vec_download(ctx, L, dout, out);
  int rc2 = ctx_free(ctx, buf);
  return rc != GDML_OK ? rc : rc2;
}
