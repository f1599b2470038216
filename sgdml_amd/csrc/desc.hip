// Descriptor kernel (replaces sgdml/utils/desc.py:_from_r / Desc.from_R) and the training-set
// upload (replaces the H2D copies at sgdml/train.py:1447-1448).
#include "common.h"

// One thread per (geometry m, pair k).  Pair order = np.tril_indices(N,-1): k -> (i_k > j_k).
// x[m,k] = 1/|r_i - r_j| ; g[m,k,:] = (r_i - r_j)/d^3   (desc.py:163, :193-205)
__global__ void __launch_bounds__(256) desc_kernel(const double* __restrict__ R, int64_t M, int N,
                                                   int D, Lattice L, double* __restrict__ x,
                                                   double* __restrict__ g) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * (int64_t)D) return;
  int64_t m = t / D;
  int k = (int)(t - m * D);
  // invert k = i(i-1)/2 + j, i > j
  int i = (int)((1.0 + sqrt(1.0 + 8.0 * (double)k)) * 0.5);
  while (i * (i - 1) / 2 > k) --i;
  while ((i + 1) * i / 2 <= k) ++i;
  int j = k - i * (i - 1) / 2;
  const double* ri = R + (m * N + i) * 3;
  const double* rj = R + (m * N + j) * 3;
  double d0 = ri[0] - rj[0], d1 = ri[1] - rj[1], d2 = ri[2] - rj[2];
  if (L.use) {
    // minimum image: diff -= lat @ rint(lat_inv @ diff)   (desc.py:44-77; np.around = half-even)
    double c0 = rint(L.inv[0] * d0 + L.inv[1] * d1 + L.inv[2] * d2);
    double c1 = rint(L.inv[3] * d0 + L.inv[4] * d1 + L.inv[5] * d2);
    double c2 = rint(L.inv[6] * d0 + L.inv[7] * d1 + L.inv[8] * d2);
    d0 -= L.lat[0] * c0 + L.lat[1] * c1 + L.lat[2] * c2;
    d1 -= L.lat[3] * c0 + L.lat[4] * c1 + L.lat[5] * c2;
    d2 -= L.lat[6] * c0 + L.lat[7] * c1 + L.lat[8] * c2;
  }
  double dist = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
  double inv3 = 1.0 / (dist * dist * dist);
  x[t] = 1.0 / dist;
  g[t * 3 + 0] = d0 * inv3;
  g[t * 3 + 1] = d1 * inv3;
  g[t * 3 + 2] = d2 * inv3;
}

int desc_device(gdml_ctx* ctx, const double* d_R, int64_t M, int N, const double* lat,
                const double* lat_inv, double* d_x, double* d_g) {
  Lattice L;
  memset(&L, 0, sizeof(L));
  if (lat && lat_inv) {
    memcpy(L.lat, lat, sizeof(L.lat));
    memcpy(L.inv, lat_inv, sizeof(L.inv));
    L.use = 1;
  }
  int D = N * (N - 1) / 2;
  int64_t total = M * (int64_t)D;
  if (total == 0) return GDML_OK;
  hipLaunchKernelGGL(desc_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, ctx->stream, d_R, M, N,
                     D, L, d_x, d_g);
  ctx->launch_counter++;
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}

extern "C" int gdml_desc_from_R(gdml_ctx* ctx, const double* R, int64_t M, int N, const double* lat,
                                const double* lat_inv, double* R_desc_out, double* R_d_desc_out) {
  if (!ctx) return GDML_ERR_INVALID;
  if (!R || !R_desc_out || !R_d_desc_out || M < 0 || N < 2 || N > GDML_MAX_ATOMS)
    return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_desc_from_R: bad arguments (M=%lld N=%d)",
                     (long long)M, N);
  if ((lat == nullptr) != (lat_inv == nullptr))
    return gdml_fail(ctx, GDML_ERR_INVALID, "lattice and inverse must both be given or both NULL");
  if (M == 0) return GDML_OK;
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int64_t D = (int64_t)N * (N - 1) / 2;
  int64_t bR = M * N * 3 * 8, bx = M * D * 8, bg = M * D * 24;
  double* buf;
  GDML_TRY(ctx_scratch(ctx, bR + bx + bg, &buf));
  double* d_R = buf;
  double* d_x = buf + M * N * 3;
  double* d_g = d_x + M * D;
  HIP_CHECK(ctx, hipMemcpyAsync(d_R, R, bR, hipMemcpyHostToDevice, ctx->stream));
  phase_begin(ctx);
  GDML_TRY(desc_device(ctx, d_R, M, N, lat, lat_inv, d_x, d_g));
  GDML_TRY(phase_end(ctx, "desc"));
  HIP_CHECK(ctx, hipMemcpyAsync(R_desc_out, d_x, bx, hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(ctx, hipMemcpyAsync(R_d_desc_out, d_g, bg, hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return GDML_OK;
}

// Recover the atom permutation that induces a descriptor permutation (inverse of Desc.perm,
// desc.py:509-539): tp[k] = pair{pi(i_k), pi(j_k)}.  Returns false if tp is not of that form.
static bool atom_perm_from_tril_perm(const int32_t* tp, int N, int32_t* pi) {
  int D = N * (N - 1) / 2;
  auto pidx = [](int a, int b) {
    int hi = a > b ? a : b, lo = a > b ? b : a;
    return hi * (hi - 1) / 2 + lo;
  };
  std::vector<int> pi_i(D), pj_j(D);
  {
    int k = 0;
    for (int i = 1; i < N; ++i)
      for (int j = 0; j < i; ++j, ++k) {
        pi_i[k] = i;
        pj_j[k] = j;
      }
  }
  for (int k = 0; k < D; ++k)
    if (tp[k] < 0 || tp[k] >= D) return false;
  if (N == 2) {
    pi[0] = 0;
    pi[1] = 1;
  } else {
    for (int a = 0; a < N; ++a) {
      int m1 = (a + 1) % N, m2 = (a + 2) % N;
      int k1 = tp[pidx(a, m1)], k2 = tp[pidx(a, m2)];
      int s1[2] = {pi_i[k1], pj_j[k1]}, s2[2] = {pi_i[k2], pj_j[k2]};
      int common = -1, cnt = 0;
      for (int u = 0; u < 2; ++u)
        for (int v = 0; v < 2; ++v)
          if (s1[u] == s2[v]) {
            common = s1[u];
            ++cnt;
          }
      if (cnt != 1) return false;
      pi[a] = common;
    }
  }
  std::vector<char> seen(N, 0);
  for (int a = 0; a < N; ++a) {
    if (pi[a] < 0 || pi[a] >= N || seen[pi[a]]) return false;
    seen[pi[a]] = 1;
  }
  for (int k = 0; k < D; ++k)
    if (tp[k] != pidx(pi[pi_i[k]], pi[pj_j[k]])) return false;
  return true;
}

int upload_perms(gdml_ctx* ctx, const int64_t* tril_perms, int P, int N, std::vector<int32_t>& h_tp,
                 std::vector<int32_t>& h_perm, std::vector<int32_t>& h_pinv) {
  int D = N * (N - 1) / 2;
  h_tp.resize((size_t)P * D);
  h_perm.resize((size_t)P * N);
  h_pinv.resize((size_t)P * N);
  for (int p = 0; p < P; ++p) {
    for (int k = 0; k < D; ++k) {
      int64_t v = tril_perms[(size_t)p * D + k];
      if (v < 0 || v >= D)
        return gdml_fail(ctx, GDML_ERR_INVALID, "tril_perms[%d][%d]=%lld out of range", p, k,
                         (long long)v);
      h_tp[(size_t)p * D + k] = (int32_t)v;
    }
    if (!atom_perm_from_tril_perm(&h_tp[(size_t)p * D], N, &h_perm[(size_t)p * N]))
      return gdml_fail(ctx, GDML_ERR_INVALID,
                       "tril_perms row %d is not induced by an atom permutation (Desc.perm)", p);
    for (int a = 0; a < N; ++a) h_pinv[(size_t)p * N + h_perm[(size_t)p * N + a]] = a;
  }
  return GDML_OK;
}

extern "C" int gdml_train_upload(gdml_ctx* ctx, const double* R_desc, const double* R_d_desc,
                                 int64_t M, int N, const int64_t* tril_perms, int P) {
  if (!ctx) return GDML_ERR_INVALID;
  if (!R_desc || !R_d_desc || !tril_perms || M < 1 || N < 2 || N > GDML_MAX_ATOMS || P < 1)
    return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_train_upload: bad arguments (M=%lld N=%d P=%d)",
                     (long long)M, N, P);
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  TrainSet& ts = ctx->ts;
  int D = N * (N - 1) / 2;
  GDML_TRY(upload_perms(ctx, tril_perms, P, N, ts.h_tp, ts.h_perm, ts.h_pinv));
  GDML_TRY(ctx_free(ctx, ts.x));
  GDML_TRY(ctx_free(ctx, ts.g));
  GDML_TRY(ctx_free(ctx, ts.tp));
  GDML_TRY(ctx_free(ctx, ts.perm));
  GDML_TRY(ctx_free(ctx, ts.pinv));
  GDML_TRY(ctx_free(ctx, ts.XF));
  GDML_TRY(ctx_free(ctx, ts.GD));
  GDML_TRY(ctx_free(ctx, ts.TS));
  GDML_TRY(ctx_free(ctx, ts.p2));
  GDML_TRY(ctx_free(ctx, ts.p2_TP));
  ts.p2 = nullptr;
  ts.p2_TP = nullptr;
  ts.XF = ts.GD = ts.TS = nullptr;
  ts.x = ts.g = nullptr;
  ts.tp = ts.perm = ts.pinv = nullptr;
  ts.M = M; ts.N = N; ts.D = D; ts.P = P;
  GDML_TRY(ctx_alloc(ctx, (void**)&ts.x, M * D * 8));
  GDML_TRY(ctx_alloc(ctx, (void**)&ts.g, M * D * 24));
  GDML_TRY(ctx_alloc(ctx, (void**)&ts.tp, (int64_t)P * D * 4));
  GDML_TRY(ctx_alloc(ctx, (void**)&ts.perm, (int64_t)P * N * 4));
  GDML_TRY(ctx_alloc(ctx, (void**)&ts.pinv, (int64_t)P * N * 4));
  HIP_CHECK(ctx, hipMemcpyAsync(ts.x, R_desc, M * D * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_CHECK(ctx, hipMemcpyAsync(ts.g, R_d_desc, M * D * 24, hipMemcpyHostToDevice, ctx->stream));
  HIP_CHECK(ctx, hipMemcpyAsync(ts.tp, ts.h_tp.data(), (size_t)P * D * 4, hipMemcpyHostToDevice,
                                ctx->stream));
  HIP_CHECK(ctx, hipMemcpyAsync(ts.perm, ts.h_perm.data(), (size_t)P * N * 4,
                                hipMemcpyHostToDevice, ctx->stream));
  HIP_CHECK(ctx, hipMemcpyAsync(ts.pinv, ts.h_pinv.data(), (size_t)P * N * 4,
                                hipMemcpyHostToDevice, ctx->stream));
  HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return GDML_OK;
}
