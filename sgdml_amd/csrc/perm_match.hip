// Pairwise atom matching of the symmetry search (reference: sgdml/utils/perm.py:53-92 _bipartite_match_wkr, :95-255 bipartite_match).
//
// For every pair i < j of M geometries the reference solves an N x N linear assignment problem on the host
// (scipy.optimize.linear_sum_assignment in a process pool): cost[a][b] = -sum_k |V_i[a][k]| |V_j[b][k]| with V the eigenvectors of
// the interatomic distance matrix by decreasing eigenvalue, atoms of different species pushed above every same-species entry;
// the assignment is kept if permuting geometry i brings its distance matrix strictly closer (Frobenius norm, beyond
// numpy.isclose) to geometry j's.  M is up to 1000 (train.py:565-573): 499 500 assignments -- 17 s of host time at N = 21 next to a
// 1.3 s training run.  Here ONE wavefront per pair:
//   1. the cost matrix into the workgroup's own slice of a device buffer (lane = column b; the slice is re-read with
//      device-scope loads: it is rewritten for every pair the workgroup takes, the vector L1 is not coherent with that);
//   2. the shortest-augmenting-path algorithm scipy implements (Crouse 2016; rectangular_lsap.cpp): per row a Dijkstra scan whose
//      relaxation step runs over the open columns 64 at a time, the minimum by a wavefront reduction, duals u / v, the
//      shortest-path costs and the matching in LDS.  With distinct path costs the result does not depend on scan order; on an
//      exact tie a column that is still free wins (scipy's rule), then the lower index;
//   3. both Frobenius norms in ONE order of summation, so that an identity assignment reproduces `before` bit for bit.
// The eigenvectors come from sym_eig_kernel below (batched Jacobi) unless the caller brings them; the spanning tree and the group
// closure are host work (utils/perm.py).
#include "common.h"

#include <cmath>
#include <vector>

namespace {

struct PermMatchArgs {
  const double* absv;   // (M, N, N)  [m][a][k] = |V_m[a][k]|
  const double* absvT;  // (M, N, N)  [m][k][b] = |V_m[b][k]|
  const double* adj;    // (M, N, N)
  const int32_t* species;  // (N)
  int64_t M;
  int N;
  int64_t pair_begin, pair_end;  // pairs in row-major order of the strict upper triangle: (0,1), (0,2), ..., (M-2,M-1)
  double* slices;   // gridDim.x slices of N x N doubles
  double* cost_out;  // (M, M): entry (i, j), i < j
  int32_t* found_ij;  // (capacity, 2)
  int32_t* found_perm;  // (capacity, N)
  unsigned* counter;
  unsigned capacity;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double ld_dev(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(64) perm_match_kernel(PermMatchArgs A) {
  extern __shared__ __attribute__((aligned(8))) unsigned char smem_raw[];
  const int N = A.N, NN = N * N, lane = threadIdx.x;
  double* const u = reinterpret_cast<double*>(smem_raw);
  double* const v = u + N;
  double* const sh = v + N;  // shortest path cost to column j
  int* const path = reinterpret_cast<int*>(sh + N);
  int* const row4col = path + N;
  int* const col4row = row4col + N;
  int* const SR = col4row + N;
  int* const SC = SR + N;
  double* const C = A.slices + (int64_t)blockIdx.x * NN;
  const double INF = __builtin_huge_val();

  for (int64_t pr = A.pair_begin + blockIdx.x; pr < A.pair_end; pr += gridDim.x) {
    // pair index -> (i, j): row i holds M - 1 - i pairs; rows before i hold i (2M - i - 1) / 2
    int64_t i;
    {
      const double Md = (double)A.M;
      double t = (2.0 * Md - 1.0 - sqrt((2.0 * Md - 1.0) * (2.0 * Md - 1.0) - 8.0 * (double)pr)) * 0.5;
      i = (int64_t)t;
      if (i < 0) i = 0;
      while (i > 0 && i * (2 * A.M - i - 1) / 2 > pr) --i;
      while ((i + 1) * (2 * A.M - i - 2) / 2 <= pr) ++i;
    }
    const int64_t j = pr - i * (2 * A.M - i - 1) / 2 + i + 1;
    const double* Vi = A.absv + i * NN;
    const double* VjT = A.absvT + j * NN;
    const double* Ai = A.adj + i * NN;
    const double* Aj = A.adj + j * NN;

    // ---- 1. cost matrix (lane = entry (a, b): N^2 / 64 rounds of an N-term sum)
    double cmax = 0.0;
    for (int e = lane; e < NN; e += 64) {
      const int a = e / N, b = e - a * N;
      double s = 0.0;
      for (int k = 0; k < N; ++k) s += Vi[a * N + k] * VjT[k * N + b];
      C[e] = -s;
      cmax = fmax(cmax, fabs(s));
    }
    cmax = wave_max(cmax);
    __threadfence();
    for (int e = lane; e < NN; e += 64) {
      const int a = e / N, b = e - a * N;
      if (A.species[a] != A.species[b]) C[e] = ld_dev(C + e) + cmax;
    }
    for (int e = lane; e < N; e += 64) {
      u[e] = 0.0;
      v[e] = 0.0;
      row4col[e] = -1;
      col4row[e] = -1;
    }
    __threadfence();
    __syncthreads();

    // ---- 2. assignment: one augmenting path per row
    for (int cur = 0; cur < N; ++cur) {
      for (int e = lane; e < N; e += 64) {
        sh[e] = INF;
        SR[e] = 0;
        SC[e] = 0;
      }
      __syncthreads();
      double minVal = 0.0;
      int row = cur, sink = -1;
      while (sink < 0) {
        if (lane == 0) SR[row] = 1;
        const double ui = u[row];
        double best = INF;
        int bestj = 0x7fffffff, bestfree = 0;
        for (int b = lane; b < N; b += 64) {
          if (SC[b]) continue;
          const double r = minVal + ld_dev(C + row * N + b) - ui - v[b];
          double s = sh[b];
          if (r < s) {
            path[b] = row;
            sh[b] = r;
            s = r;
          }
          const int fr = row4col[b] < 0;
          if (s < best || (s == best && fr && !bestfree)) {
            best = s;
            bestj = b;
            bestfree = fr;
          }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const double ob = __shfl_xor(best, o, 64);
          const int oj = __shfl_xor(bestj, o, 64), of = __shfl_xor(bestfree, o, 64);
          const bool take = ob < best || (ob == best && (of > bestfree || (of == bestfree && oj < bestj)));
          if (take) {
            best = ob;
            bestj = oj;
            bestfree = of;
          }
        }
        if (!(best < INF)) break;  // infeasible: cannot happen with finite costs
        minVal = best;
        const int jsel = bestj;
        if (lane == 0) SC[jsel] = 1;
        const int owner = row4col[jsel];
        if (owner < 0) sink = jsel;
        else row = owner;
        __syncthreads();
      }
      if (sink < 0) break;
      // duals (rectangular_lsap.cpp: "update dual variables")
      for (int e = lane; e < N; e += 64) {
        if (e == cur) u[e] += minVal;
        else if (SR[e]) u[e] += minVal - sh[col4row[e]];
      }
      for (int e = lane; e < N; e += 64)
        if (SC[e]) v[e] -= minVal - sh[e];
      __syncthreads();
      if (lane == 0) {  // augment along the path
        int jj = sink;
        for (;;) {
          const int ii = path[jj];
          row4col[jj] = ii;
          const int t = col4row[ii];
          col4row[ii] = jj;
          jj = t;
          if (ii == cur) break;
        }
      }
      __syncthreads();
    }

    {  // (an unfinished matching -- non-finite input -- is reported as the identity: never kept)
      int bad = 0;
      for (int e = lane; e < N; e += 64) bad |= col4row[e] < 0;
      bad = __any(bad);
      __syncthreads();
      if (bad)
        for (int e = lane; e < N; e += 64) col4row[e] = e;
      __syncthreads();
    }
    // ---- 3. distance of the two distance matrices before / after permuting geometry i
    double sb = 0.0, sa = 0.0;
    for (int e = lane; e < NN; e += 64) {
      const int a = e / N, b = e - a * N;
      const double t = Aj[e];
      const double d0 = Ai[e] - t;
      const double d1 = Ai[col4row[a] * N + col4row[b]] - t;
      sb += d0 * d0;
      sa += d1 * d1;
    }
    const double before = sqrt(wave_sum(sb)), after = sqrt(wave_sum(sa));
    const bool keep_before = after >= before;
    const bool close = fabs(before - after) <= 1e-8 + 1e-5 * fabs(after);  // numpy.isclose(before, after)
    unsigned slot = 0;
    if (lane == 0) {
      A.cost_out[i * A.M + j] = keep_before ? before : after;
      if (!keep_before && !close) slot = atomicAdd(A.counter, 1u) + 1u;
    }
    slot = __shfl(slot, 0, 64);
    if (slot != 0 && slot <= A.capacity) {
      if (lane == 0) {
        A.found_ij[2 * (int64_t)(slot - 1)] = (int32_t)i;
        A.found_ij[2 * (int64_t)(slot - 1) + 1] = (int32_t)j;
      }
      for (int e = lane; e < N; e += 64) A.found_perm[(int64_t)(slot - 1) * N + e] = col4row[e];
    }
    __syncthreads();
  }
}


// ------------------------------------------------------------------------------------------
// Eigenvectors of the M symmetric N x N distance matrices (the reference: numpy.linalg.eig per geometry, perm.py:183-187; the
// host form here: one batched LAPACK eigh -- 4.5 s for 1000 geometries of 100 atoms, more than the matching itself).  One
// workgroup per matrix, cyclic two-sided Jacobi with the round-robin ordering: a step rotates N/2 disjoint index pairs at once
// (angles from the current matrix, then all row updates, then all column updates -- disjoint rotations commute), N - 1 steps
// visit every pair once.  The working matrix and the accumulated rotations live in LDS when both fit (N <= 100), the rotations
// in the output buffer otherwise (N <= 140; above that both in device memory).  Converged when the off-diagonal mass is below
// 1e-30 of the diagonal's -- eigenvectors to a few ulp for separated eigenvalues.  Output: |V| with columns by decreasing
// eigenvalue, the form the matching consumes (only |V| enters, perm.py:66-70, so the sign convention is immaterial).
// ------------------------------------------------------------------------------------------
struct SymEigArgs {
  const double* adj;  // (M, N, N)
  double* absv;       // (M, N, N) out: [a][k]
  double* absvT;      // (M, N, N) out: [k][a]
  double* work;       // gridDim.x x 2 x N x NP doubles (only the parts that do not fit in LDS are used)
  int64_t M;
  int N, NP;          // NP: row pitch of the working matrices (N + 1 when N is even: column accesses spread over the banks)
  int a_in_lds, v_in_lds;
};

__global__ void __launch_bounds__(256) sym_eig_kernel(SymEigArgs S) {
  extern __shared__ __attribute__((aligned(8))) unsigned char eig_raw[];
  const int N = S.N, NP = S.NP, tid = threadIdx.x;
  const int ne = (N + 1) & ~1, npairs = ne / 2, m = ne - 1;
  double* lds = reinterpret_cast<double*>(eig_raw);
  double* cs = lds;              // [npairs][2]
  int* pq = reinterpret_cast<int*>(cs + 2 * npairs);  // [npairs][2]
  double* red = reinterpret_cast<double*>(pq + 2 * npairs + (npairs & 1) * 2);  // [8]
  double* next = red + 8;
  double* gw = S.work + (int64_t)blockIdx.x * 2 * N * NP;
  double* A = S.a_in_lds ? next : gw;
  if (S.a_in_lds) next += N * NP;
  double* V = S.v_in_lds ? next : gw + N * NP;

  for (int64_t mat = blockIdx.x; mat < S.M; mat += gridDim.x) {
    const double* G = S.adj + mat * N * N;
    for (int e = tid; e < N * N; e += 256) {
      const int r = e / N, c = e - r * N;
      A[r * NP + c] = 0.5 * (G[e] + G[c * N + r]);
      V[r * NP + c] = (r == c) ? 1.0 : 0.0;
    }
    __threadfence_block();
    __syncthreads();
    for (int sweep = 0; sweep < 30; ++sweep) {
      // off-diagonal mass against the diagonal's
      double off = 0.0, dia = 0.0;
      for (int e = tid; e < N * N; e += 256) {
        const int r = e / N, c = e - r * N;
        const double a = A[r * NP + c];
        if (r == c) dia += a * a; else off += a * a;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o, 64); dia += __shfl_xor(dia, o, 64); }
      if ((tid & 63) == 0) { red[2 * (tid >> 6)] = off; red[2 * (tid >> 6) + 1] = dia; }
      __syncthreads();
      off = red[0] + red[2] + red[4] + red[6];
      dia = red[1] + red[3] + red[5] + red[7];
      __syncthreads();
      if (off <= 1e-30 * (dia + off)) break;
      for (int step = 0; step < m; ++step) {
        if (tid < npairs) {
          int p, q;
          if (tid == 0) { p = step % m; q = m; }
          else { p = (step + tid) % m; q = (step + m - tid) % m; }
          if (p > q) { const int t = p; p = q; q = t; }
          double c = 1.0, s = 0.0;
          if (q < N) {
            const double apq = A[p * NP + q];
            if (apq != 0.0) {
              const double tau = (A[q * NP + q] - A[p * NP + p]) / (2.0 * apq);
              const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
              c = 1.0 / sqrt(1.0 + t * t);
              s = t * c;
            }
          } else {
            q = -1;  // the padding index of an odd N: no rotation
          }
          cs[2 * tid] = c; cs[2 * tid + 1] = s;
          pq[2 * tid] = p; pq[2 * tid + 1] = q;
        }
        __syncthreads();
        // rows p, q of A:  A <- J^T A
        for (int e = tid; e < npairs * N; e += 256) {
          const int k = e / N, j = e - k * N;
          const int p = pq[2 * k], q = pq[2 * k + 1];
          if (q < 0) continue;
          const double c = cs[2 * k], s = cs[2 * k + 1];
          const double ap = A[p * NP + j], aq = A[q * NP + j];
          A[p * NP + j] = c * ap - s * aq;
          A[q * NP + j] = s * ap + c * aq;
        }
        __threadfence_block();
        __syncthreads();
        // columns p, q of A and of V:  A <- A J,  V <- V J
        for (int e = tid; e < npairs * N; e += 256) {
          const int k = e / N, i = e - k * N;
          const int p = pq[2 * k], q = pq[2 * k + 1];
          if (q < 0) continue;
          const double c = cs[2 * k], s = cs[2 * k + 1];
          const double ap = A[i * NP + p], aq = A[i * NP + q];
          A[i * NP + p] = c * ap - s * aq;
          A[i * NP + q] = s * ap + c * aq;
          const double vp = V[i * NP + p], vq = V[i * NP + q];
          V[i * NP + p] = c * vp - s * vq;
          V[i * NP + q] = s * vp + c * vq;
        }
        __threadfence_block();
        __syncthreads();
        // (the rotated pair's off-diagonal entry is zero up to rounding: make it exactly symmetric-zero)
        if (tid < npairs && pq[2 * tid + 1] >= 0) {
          A[pq[2 * tid] * NP + pq[2 * tid + 1]] = 0.0;
          A[pq[2 * tid + 1] * NP + pq[2 * tid]] = 0.0;
        }
        __syncthreads();
      }
    }
    // columns by decreasing eigenvalue (ties: lower index first), absolute values out
    double* O = S.absv + mat * N * N;
    double* OT = S.absvT + mat * N * N;
    for (int i = tid; i < N; i += 256) {
      const double li = A[i * NP + i];
      int rank = 0;
      for (int j = 0; j < N; ++j) {
        const double lj = A[j * NP + j];
        rank += (lj > li) || (lj == li && j < i);
      }
      for (int a = 0; a < N; ++a) {
        const double av = fabs(V[a * NP + i]);
        O[a * N + rank] = av;
        OT[rank * N + a] = av;
      }
    }
    __threadfence_block();
    __syncthreads();
  }
}

}  // namespace

// |V| (columns by decreasing eigenvalue) of M symmetric N x N matrices already on the device; d_v [a][k], d_vT [k][a]
static int launch_sym_eig(gdml_ctx* ctx, const double* d_adj, double* d_v, double* d_vT, int64_t M, int N) {
  SymEigArgs e;
  e.adj = d_adj; e.absv = d_v; e.absvT = d_vT; e.M = M; e.N = N;
  e.NP = (N % 2 == 0) ? N + 1 : N;
  const int ne = (N + 1) & ~1, npairs = ne / 2;
  const size_t fixed = (size_t)(2 * npairs) * 8 + (size_t)(2 * npairs + (npairs & 1) * 2) * 4 + 8 * 8;
  const size_t mat_b = (size_t)N * e.NP * 8, lds_max = 160 * 1024;
  e.a_in_lds = fixed + mat_b <= lds_max;
  e.v_in_lds = e.a_in_lds && fixed + 2 * mat_b <= lds_max;
  const size_t eig_lds = fixed + (e.a_in_lds ? mat_b : 0) + (e.v_in_lds ? mat_b : 0);
  int64_t eg = (int64_t)ctx->num_cus * (eig_lds > 80 * 1024 ? 1 : eig_lds > 40 * 1024 ? 2 : 4);
  if (eg > M) eg = M;
  double* d_w = nullptr;
  GDML_TRY(ctx_alloc(ctx, (void**)&d_w, eg * 2 * (int64_t)N * e.NP * 8));
  e.work = d_w;
  hipError_t se = hipSuccess;
  if (eig_lds > 48 * 1024)
    se = hipFuncSetAttribute(reinterpret_cast<const void*>(sym_eig_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)eig_lds);
  if (se == hipSuccess) {
    const int es = ktime_begin(ctx);
    hipLaunchKernelGGL(sym_eig_kernel, dim3((unsigned)eg), dim3(256), eig_lds, ctx->stream, e);
    ktime_end(ctx, es, "sym_eig", (double)M);
    ctx->launch_counter++;
    se = hipGetLastError();
    if (se == hipSuccess) se = hipStreamSynchronize(ctx->stream);  // (the work buffer is released below)
  }
  ctx_free(ctx, d_w);
  if (se != hipSuccess) return gdml_fail(ctx, GDML_ERR_HIP, "eigenvector kernel of the symmetry search: %s", hipGetErrorString(se));
  return GDML_OK;
}

// |eigenvectors| of M symmetric N x N matrices, columns by decreasing eigenvalue (what gdml_perm_match computes when it is not
// handed any): for tests and for callers that want them on the host.
extern "C" int gdml_sym_eig_absv(gdml_ctx* ctx, const double* adj, int64_t M, int N, double* absv_out) {
  if (!ctx) return GDML_ERR_INVALID;
  if (!adj || !absv_out || M < 0 || N < 1 || N > GDML_MAX_ATOMS)
    return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_sym_eig_absv: bad arguments (M=%lld N=%d)", (long long)M, N);
  if (M == 0) return GDML_OK;
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int64_t bytes = M * (int64_t)N * N * 8;
  double *d_a = nullptr, *d_v = nullptr, *d_t = nullptr;
  int rc = ctx_alloc(ctx, (void**)&d_a, bytes);
  if (rc == GDML_OK) rc = ctx_alloc(ctx, (void**)&d_v, bytes);
  if (rc == GDML_OK) rc = ctx_alloc(ctx, (void**)&d_t, bytes);
  if (rc == GDML_OK && hipMemcpyAsync(d_a, adj, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
    rc = gdml_fail(ctx, GDML_ERR_HIP, "gdml_sym_eig_absv: upload failed");
  if (rc == GDML_OK) rc = launch_sym_eig(ctx, d_a, d_v, d_t, M, N);
  if (rc == GDML_OK && (hipMemcpyAsync(absv_out, d_v, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                        hipStreamSynchronize(ctx->stream) != hipSuccess))
    rc = gdml_fail(ctx, GDML_ERR_HIP, "gdml_sym_eig_absv: download failed");
  ctx_free(ctx, d_a); ctx_free(ctx, d_v); ctx_free(ctx, d_t);
  return rc;
}

// absv: (M,N,N) |eigenvectors| of the distance matrices, columns by decreasing eigenvalue, or NULL: computed on the device
// (sym_eig_kernel); adj: (M,N,N) distance matrices;
// species: (N).  cost_out (M,M): entries (i,j), i < j (the rest is zero); found_ij (capacity,2) / found_perm
// (capacity,N): the kept assignments in no particular order, *n_found of them (if *n_found > capacity the call has to be repeated
// with more room: the first `capacity` are valid).
extern "C" int gdml_perm_match(gdml_ctx* ctx, const double* absv, const double* adj, const int32_t* species, int64_t M, int N,
                               double* cost_out, int32_t* found_ij, int32_t* found_perm, int64_t capacity, int64_t* n_found) {
  if (!ctx) return GDML_ERR_INVALID;
  if (!adj || !species || !cost_out || !found_ij || !found_perm || !n_found || M < 0 || N < 1 || N > GDML_MAX_ATOMS ||
      capacity < 0 || M > 2000000)
    return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_perm_match: bad arguments (M=%lld N=%d)", (long long)M, N);
  *n_found = 0;
  if (M < 2) return GDML_OK;
  const size_t lds = (size_t)N * (3 * 8 + 5 * 4);
  if (lds > 64 * 1024)
    return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "gdml_perm_match: N = %d atoms exceed the matching kernel's LDS tables", N);
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int64_t NN = (int64_t)N * N, pairs = M * (M - 1) / 2;
  int64_t grid = (int64_t)ctx->num_cus * 32;  // a wavefront per pair, 32 per CU in flight
  if (grid > pairs) grid = pairs;
  while (grid > 256 && grid * NN * 8 > ((int64_t)1 << 31)) grid /= 2;  // at most 2 GiB of cost slices
  if (capacity > pairs) capacity = pairs;
  // host -> device: the distance matrices; |V| and its per-geometry transpose when the caller brings the eigenvectors
  std::vector<double> hT;
  if (absv) {
    hT.resize((size_t)(M * NN));
    for (int64_t m = 0; m < M; ++m)
      for (int a = 0; a < N; ++a)
        for (int k = 0; k < N; ++k) hT[(size_t)(m * NN + (int64_t)k * N + a)] = absv[m * NN + (int64_t)a * N + k];
  }
  double *d_v = nullptr, *d_vT = nullptr, *d_adj = nullptr, *d_sl = nullptr, *d_cost = nullptr;
  int32_t *d_sp = nullptr, *d_ij = nullptr, *d_pm = nullptr;
  unsigned* d_cnt = nullptr;
  int rc = GDML_OK;
  auto release = [&]() {
    ctx_free(ctx, d_v); ctx_free(ctx, d_vT); ctx_free(ctx, d_adj); ctx_free(ctx, d_sl); ctx_free(ctx, d_cost);
    ctx_free(ctx, d_sp); ctx_free(ctx, d_ij); ctx_free(ctx, d_pm); ctx_free(ctx, d_cnt);
  };
#define PM_TRY(expr)              \
  do {                            \
    rc = (expr);                  \
    if (rc != GDML_OK) {          \
      release();                  \
      return rc;                  \
    }                             \
  } while (0)
#define PM_HIP(call)                                                                                     \
  do {                                                                                                   \
    hipError_t e__ = (call);                                                                             \
    if (e__ != hipSuccess) {                                                                             \
      release();                                                                                         \
      return gdml_fail(ctx, GDML_ERR_HIP, "gdml_perm_match: %s: %s", #call, hipGetErrorString(e__));     \
    }                                                                                                    \
  } while (0)
  PM_TRY(ctx_alloc(ctx, (void**)&d_v, M * NN * 8));
  PM_TRY(ctx_alloc(ctx, (void**)&d_vT, M * NN * 8));
  PM_TRY(ctx_alloc(ctx, (void**)&d_adj, M * NN * 8));
  PM_TRY(ctx_alloc(ctx, (void**)&d_sl, grid * NN * 8));
  PM_TRY(ctx_alloc(ctx, (void**)&d_cost, M * M * 8));
  PM_TRY(ctx_alloc(ctx, (void**)&d_sp, (int64_t)N * 4));
  PM_TRY(ctx_alloc(ctx, (void**)&d_ij, (capacity > 0 ? capacity : 1) * 2 * 4));
  PM_TRY(ctx_alloc(ctx, (void**)&d_pm, (capacity > 0 ? capacity : 1) * (int64_t)N * 4));
  PM_TRY(ctx_alloc(ctx, (void**)&d_cnt, 64));
  hipStream_t st = ctx->stream;
  PM_HIP(hipMemcpyAsync(d_adj, adj, (size_t)(M * NN * 8), hipMemcpyHostToDevice, st));
  phase_begin(ctx);
  if (absv) {
    PM_HIP(hipMemcpyAsync(d_v, absv, (size_t)(M * NN * 8), hipMemcpyHostToDevice, st));
    PM_HIP(hipMemcpyAsync(d_vT, hT.data(), (size_t)(M * NN * 8), hipMemcpyHostToDevice, st));
  } else {
    rc = launch_sym_eig(ctx, d_adj, d_v, d_vT, M, N);
    if (rc != GDML_OK) {
      release();
      return rc;
    }
  }
  PM_HIP(hipMemcpyAsync(d_sp, species, (size_t)N * 4, hipMemcpyHostToDevice, st));
  PM_HIP(hipMemsetAsync(d_cnt, 0, 64, st));
  PM_HIP(hipMemsetAsync(d_cost, 0, (size_t)(M * M * 8), st));
  PermMatchArgs a;
  a.absv = d_v; a.absvT = d_vT; a.adj = d_adj; a.species = d_sp; a.M = M; a.N = N;
  a.pair_begin = 0; a.pair_end = pairs;
  a.slices = d_sl; a.cost_out = d_cost; a.found_ij = d_ij; a.found_perm = d_pm; a.counter = d_cnt;
  a.capacity = (unsigned)capacity;
  if (lds > 48 * 1024)
    PM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(perm_match_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int slot = ktime_begin(ctx);
  hipLaunchKernelGGL(perm_match_kernel, dim3((unsigned)grid), dim3(64), lds, st, a);
  ktime_end(ctx, slot, "perm_match", (double)pairs);
  ctx->launch_counter++;
  PM_HIP(hipGetLastError());
  PM_TRY(phase_end(ctx, "perm_match"));
  unsigned cnt = 0;
  PM_HIP(hipMemcpyAsync(&cnt, d_cnt, 4, hipMemcpyDeviceToHost, st));
  PM_HIP(hipMemcpyAsync(cost_out, d_cost, (size_t)(M * M * 8), hipMemcpyDeviceToHost, st));
  PM_HIP(hipStreamSynchronize(st));
  *n_found = (int64_t)cnt;
  const int64_t take = (int64_t)cnt < capacity ? (int64_t)cnt : capacity;
  if (take > 0) {
    PM_HIP(hipMemcpyAsync(found_ij, d_ij, (size_t)(take * 2 * 4), hipMemcpyDeviceToHost, st));
    PM_HIP(hipMemcpyAsync(found_perm, d_pm, (size_t)(take * N * 4), hipMemcpyDeviceToHost, st));
    PM_HIP(hipStreamSynchronize(st));
  }
  release();
#undef PM_TRY
#undef PM_HIP
  return GDML_OK;
}
