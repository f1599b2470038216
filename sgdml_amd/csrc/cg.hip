// Nystroem-preconditioned CG on the device (replaces sgdml/solvers/iterative.py) -- see below.
#include "common.h"

extern "C" int gdml_nystroem_factor(gdml_ctx* ctx, double lam, const int64_t* idx, int64_t m,
                                    double* lev_scores_out, double* LinvKmn_host_out, int* info) {
  if (!ctx) return GDML_ERR_INVALID;
  return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "gdml_nystroem_factor: not built yet");
}
extern "C" int gdml_precon_apply(gdml_ctx* ctx, double lam, const double* v, int64_t n, double* out) {
  if (!ctx) return GDML_ERR_INVALID;
  return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "gdml_precon_apply: not built yet");
}
extern "C" int gdml_pcg(gdml_ctx* ctx, double lam, int use_E_cstr, const double* y, const double* x0,
                        int64_t n, double rtol, int64_t maxiter, int use_precon, gdml_pcg_cb cb,
                        int64_t cb_every, void* user, double* x_out, int64_t* iters_out,
                        double* resid_out, int* info_out) {
  if (!ctx) return GDML_ERR_INVALID;
  return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "gdml_pcg: not built yet");
}
extern "C" int gdml_comm_unique_id(void* id128_out) { return GDML_ERR_UNSUPPORTED; }
extern "C" int gdml_comm_init(gdml_ctx* ctx, const void* id128, int rank, int world) {
  if (!ctx) return GDML_ERR_INVALID;
  return gdml_fail(ctx, GDML_ERR_UNSUPPORTED, "gdml_comm_init: not built yet");
}
extern "C" int gdml_comm_info(gdml_ctx* ctx, int* rank_out, int* world_out) {
  if (!ctx) return GDML_ERR_INVALID;
  if (rank_out) *rank_out = ctx->rank;
  if (world_out) *world_out = ctx->world;
  return GDML_OK;
}
