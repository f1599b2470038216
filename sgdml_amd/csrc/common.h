// Internal definitions shared by the HIP translation units of libgdml_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/gdml_hip.h"

// Sanity bound on the molecule size, not a kernel limit: 32-bit index arithmetic over the N x N x 3 dense tables and the
// D = N (N - 1) / 2 descriptor rows holds far beyond it; memory (M N^2 32 bytes of dense tables) ends earlier.
#define GDML_MAX_ATOMS 4096

// periodic cell handed to the descriptor code by value (desc.hip, the single-launch prediction path of predict.hip)
struct Lattice {
  double lat[9];
  double inv[9];
  int use;
};

struct PhaseStat {
  double ms = 0.0;
  int64_t launches = 0;
};

// Per-kernel accumulators filled when profiling is enabled (gdml_profile): HIP events are
// recorded around every launch of the named kernel on the compute stream.
struct KernelStat {
  double ms = 0.0;       // sum of launch durations
  int64_t launches = 0;
  double work = 0.0;     // algorithmic work (flops or bytes) summed over launches
};
struct PendingTiming {
  std::string name;
  hipEvent_t e0, e1;
  double work;
};

// Device-resident training set (gdml_train_upload).
struct TrainSet {
  int64_t M = 0;
  int N = 0, D = 0, P = 0;
  double* x = nullptr;       // (M,D)    descriptors
  double* g = nullptr;       // (M,D,3)  compressed Jacobians
  int32_t* tp = nullptr;     // (P,D)    descriptor permutations
  int32_t* perm = nullptr;   // (P,N)    atom permutations  pi_p
  int32_t* pinv = nullptr;   // (P,N)    inverse atom permutations
  double* XF = nullptr;      // (M,N,N)   dense x[pair(b,m)] table, m-major (assemble_wave.hip)
  double* GD = nullptr;      // (M,N,N,3) dense G(b,m) table, m-major
  double* TS = nullptr;      // (M,pitch) packed per-point image [GD | XF | x | 0-pad] staged by assemble_strip.hip
  uint8_t* p2 = nullptr;     // plan of assemble_perm2.hip (internal atom numbering, byte permutations, V-phase tasks)
  double* p2_TP = nullptr;   // (M,N,N,4) packed per-point tables in that numbering
  int p2_o[4] = {0, 0, 0, 0}, p2_nF = 0, p2_nFb = 0, p2_ntasks = 0, p2_npairs = 0, p2_key = 0;
  std::vector<int32_t> h_tp, h_perm, h_pinv;
};

// Device-resident model for prediction (gdml_predict_upload_model).
struct Model {
  int64_t M = 0;
  int N = 0, D = 0, P = 0;
  double sig = 0;
  double* xp = nullptr;    // (M*P,D) permuted training descriptors (predict.py:426-431)
  double* jap = nullptr;   // (M*P,D) permuted J_m alpha_m          (predict.py:433-441)
  double* ja = nullptr;    // (M,D)   unpermuted J alpha (scratch for set_alphas)
  double* aE = nullptr;    // (M*P)   alphas_E repeated per perm (predict.py:443-447) or null
  int32_t* tp = nullptr;   // (P,D)
  bool has_aE = false;
};

struct gdml_ctx {
  int device = 0;
  int num_cus = 256;  // compute units of the device (grid sizing)
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;
  hipStream_t kt_stream = nullptr;                       // stream the per-kernel timers record on (default: stream)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_la[6] = {};  // hand-offs between the two streams (look-ahead, overlapped panel solves)
  std::string err;
  std::map<std::string, double> opts;  // tuning / ablation options (gdml_set_option); absent key = built-in default
  int64_t held = 0;
  std::map<void*, int64_t> allocs;
  std::map<std::string, PhaseStat> phases;
  std::string phase_pending;         // phase whose end event has been recorded but not read yet
  int64_t phase_pending_launches = 0;
  double* h_pin = nullptr;           // pinned host staging for small transfers (single-geometry latency path)
  int64_t h_pin_bytes = 0;
  // single-launch prediction of small host batches (predict_fused_kernel): host-mapped result block the kernel writes
  // directly ([0] = completion sequence number, [8 ...] = E, F), its device alias, the launch counter of the last-block-done
  // reduction and the sequence number of the last launch
  double* h_map = nullptr;
  double* h_map_dev = nullptr;
  unsigned* d_fused_counter = nullptr;
  unsigned long long fused_seq = 0;
  int64_t launch_counter = 0;
  unsigned* gemm_queue = nullptr;    // ring of tile-counter sets (8 x 64 bytes each) of the persistent GEMM launches
  int gemm_queue_sets = 256, gemm_queue_next = 0;
  int gemm_trace_seen = 0;           // fused GEMM launches seen since option gemm.trace was set (chol.hip)
  bool profiling = false;
  std::map<std::string, KernelStat> kstats;
  std::vector<PendingTiming> pending;
  std::vector<hipEvent_t> event_pool;

  TrainSet ts;
  Model model;

  // kernel matrix / Cholesky factor
  double* K = nullptr;
  int64_t K_rows = 0, K_cols = 0, K_extra = 0, K_ld = 0, K_bytes = 0;
  bool K_factored = false;
  bool K_destroyed = false; // a failed Cholesky or an LU consumed the buffer: assemble again before factoring
  bool K_is_A = false;      // the buffer already holds A = -K + lam I (lower blocks; gdml_assemble_A): factor skips the sign flip
  bool K_rhs_row = false;   // row K_rows of the buffer carries a right-hand side (gdml_chol_set_rhs)
  double* d_rhs = nullptr;  // device copy of that right-hand side (iterative refinement)
  double K_lam = 0, K_sig = 0;
  int K_use_E = 0;

  // Nystroem preconditioner L^-1 K_mn (m x n)
  double* precon = nullptr;
  int64_t precon_m = 0, precon_n = 0;
  // form of the resident preconditioner (option pcg.precon_form): 0 = the stored n x m factor X = K_nm Z is streamed twice
  // per application; 1 = matrix-free, P v = (K_nm Z Z^T K_mn v - v)/lam through two kernel mat-vecs and the m x m matrix
  // Z = L_mm^-T L^-T (cg.hip).  precon_stage: triangular solves applied to X so far (1: only L_mm^-T -- the matrix-free
  // form does not need the second one until somebody asks for leverage scores).
  int precon_form = 0, precon_stage = 2, precon_use_E = 0;
  double precon_sig = 0;
  double* precon_Z = nullptr;     // m x K_ld, upper triangular
  int64_t precon_Z_bytes = 0;
  int64_t* precon_idx = nullptr;  // device copy of the inducing column indices (m)
  int64_t precon_idx_bytes = 0;
  // form 3: the factor rounded to fp32 (n_loc x K_ld floats) + the m x m correction T0 that makes
  // (X32 T0 X32^T - I)/lam the exact Woodbury inverse on range(X32) (cg.hip)
  float* precon_X32 = nullptr;
  int64_t precon_X32_bytes = 0;
  bool precon_X32_inplace = false;        // X32 overlays the fp64 factor in the matrix buffer (row pitch 2 K_ld floats): not ours to free
  int64_t precon_X32_ld = 0;              // row pitch of X32 in floats
  std::vector<double> precon_lev_cache;   // leverage scores taken before the in-place rounding
  double* precon_T0 = nullptr;
  int64_t precon_T0_bytes = 0;

  // scratch
  double* scratch = nullptr;
  int64_t scratch_bytes = 0;
  double* slot[13] = {};  // cached work buffers (ctx_slot)
  int64_t slot_bytes[13] = {};
  int* d_info = nullptr;

  // comm
  void* comm = nullptr;  // ncclComm_t
  void* comm2 = nullptr; // second communicator over the same ranks (ncclCommSplit): latency-critical small broadcasts of the
                         // distributed Cholesky, so that they do not queue behind the panel all-gather on `comm`
  int rank = 0, world = 1;
  // gdml_comm_suspend: the communicator parked here while the context acts as a single GPU (redundant solves)
  struct {
    bool on = false;
    void *comm = nullptr, *comm2 = nullptr;
    int rank = 0, world = 1;
    bool virtual_rank = false;
    gdml_host_allreduce ar = nullptr;
    gdml_host_allgather ag = nullptr;
  } parked;
  bool virtual_rank = false;   // shard arithmetic only, collectives skipped (tests)
  gdml_host_allreduce host_allreduce = nullptr;  // host-staged collectives (gdml_comm_init_host)
  bool comm_aborted = false;  // comm_abort() ran: every later collective fails instead of acting like a single rank
  gdml_host_allgather host_allgather = nullptr;
  void* host_coll_user = nullptr;
  double* h_coll = nullptr;    // pinned staging buffer of the host-staged collectives
  int64_t h_coll_bytes = 0;
  int64_t coll_calls = 0;      // collectives issued (any backend) since gdml_comm_init*
  double coll_bytes = 0.0;     // payload bytes this rank handed to them
  bool K_sharded = false;      // resident K holds only this rank's rows (Nystroem path)
  int64_t K_rows_global = 0;
};

int gdml_fail(gdml_ctx* ctx, int code, const char* fmt, ...);
// option lookup (ctx.hip): value of `key`, or dflt when it was never set
double ctx_opt(const gdml_ctx* ctx, const char* key, double dflt);
static inline int ctx_opt_i(const gdml_ctx* ctx, const char* key, int dflt) { return (int)ctx_opt(ctx, key, (double)dflt); }

#define HIP_CHECK(ctx, call)                                                              \
  do {                                                                                    \
    hipError_t e__ = (call);                                                              \
    if (e__ != hipSuccess)                                                                \
      return gdml_fail(ctx, e__ == hipErrorOutOfMemory ? GDML_ERR_OOM : GDML_ERR_HIP,     \
                       "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,  \
                       __LINE__);                                                         \
  } while (0)

#define GDML_TRY(expr)          \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != GDML_OK) return rc__; \
  } while (0)

// context-tracked allocation helpers (ctx.hip)
int ctx_alloc(gdml_ctx* ctx, void** p, int64_t bytes);
int ctx_free(gdml_ctx* ctx, void* p);
int ctx_scratch(gdml_ctx* ctx, int64_t bytes, double** out);
void phase_begin(gdml_ctx* ctx);
// kernel timing (no-ops unless ctx->profiling)
int phase_resolve(gdml_ctx* ctx);  // reads a pending phase timer (waits for its end event)
void precon_release_f32(gdml_ctx* ctx);  // cg.hip
void precon_release_aux(gdml_ctx* ctx);
int ktime_begin(gdml_ctx* ctx);  // returns slot or -1
void ktime_end(gdml_ctx* ctx, int slot, const char* name, double work);
int ktime_collect(gdml_ctx* ctx);
int phase_end(gdml_ctx* ctx, const char* name);

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- device helpers -----------------------------------------------------------------
// index of the descriptor entry for the unordered atom pair {a,b}, a != b
__device__ __forceinline__ int pair_idx(int a, int b) {
  int hi = a > b ? a : b;
  int lo = a > b ? b : a;
  return (hi * (hi - 1)) / 2 + lo;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// internal cross-TU entry points
int assemble_launch(gdml_ctx* ctx, double sig, int use_E_cstr, const int32_t* d_jlist,
                    int64_t n_j, const int32_t* d_colmap, int64_t col0_all);
int desc_device(gdml_ctx* ctx, const double* d_R, int64_t M, int N, const double* lat,
                const double* lat_inv, double* d_x, double* d_g);
int predict_device(gdml_ctx* ctx, const double* d_xq, const double* d_gq, int64_t B, double* d_E,
                   double* d_F);
int predict_wide_device(gdml_ctx* ctx, const double* d_xq, int64_t B, double* part_F, double* part_E);
int set_alphas_device(gdml_ctx* ctx, const double* d_alphas_F, const double* d_alphas_E);
int matvec_device(gdml_ctx* ctx, double lam, int use_E_cstr, const double* d_v, int64_t n,
                  double* d_out);
int chol_factor_device(gdml_ctx* ctx, double* A, int64_t n, int64_t ld, int* info, int64_t n_rows = 0);
int lu_factor_device(gdml_ctx* ctx, double* A, int64_t n, int64_t ld, int64_t* d_piv, int* info_out);
int chol_solve_device(gdml_ctx* ctx, const double* L, int64_t n, int64_t ld, double* d_b,
                      double* d_z, double* d_x);
int operator_model_from_trainset(gdml_ctx* ctx, double sig);
int launch_gemm_nt_sub(gdml_ctx* ctx, hipStream_t st, const double* A, int64_t lda, const double* B,
                       int64_t ldb, double* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int lower);
int launch_gemm_nt_neg(gdml_ctx* ctx, hipStream_t st, const double* A, int64_t lda, const double* B, int64_t ldb,
                       double* C, int64_t ldc, int64_t M, int64_t N, int64_t K);  // C = -A B^T (C is not read)
int launch_gemm_nt_sub_fill(gdml_ctx* ctx, hipStream_t st, const double* A, int64_t lda, const double* B, int64_t ldb,
                            double* C, int64_t ldc, int64_t M, int64_t N, int64_t K);  // tile shape by chip fill
// lower structure of a block-row-cyclic local matrix (see GemmArgs in chol.hip)
struct CyclicLower {
  int W = 0, rank = 0;
  int64_t lb0 = 0, nb = 0, col0 = 0, block_rows = 0;
};
int launch_gemm_nt_sub_cyclic(gdml_ctx* ctx, hipStream_t st, const double* A, int64_t lda, const double* B, int64_t ldb,
                              double* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const CyclicLower& cyc);
int launch_trsm64(gdml_ctx* ctx, hipStream_t st, const double* Ld, double* X, int64_t ld, int w,
                  int64_t m, int64_t ldl = 0);
int launch_panel_trsm(gdml_ctx* ctx, hipStream_t st, const double* L, double* X, int64_t ld, int nb, int64_t m,
                      int64_t ldl = 0);
int panel_factor_steps(gdml_ctx* ctx, hipStream_t st, double* A, int64_t n, int64_t ld, int64_t k0, int64_t nb);
int chol_bwd_device(gdml_ctx* ctx, const double* L, int64_t n, int64_t ld, double* d_z, double* d_x);
int ctx_slot(gdml_ctx* ctx, int slot, int64_t bytes, double** out);
void shard_points(const gdml_ctx* ctx, int64_t M, int64_t* p0, int64_t* p1, int64_t* pts_per);
// Layout of the replicated device vectors of the sharded solvers (Nystroem factor rows, PCG vectors, mat-vec in / out).
//   no communicator (world <= 1): the reference order, n entries, no padding;
//   sharded, forces only: the reference order padded to world * chunk, chunk = 3N per (per = points per rank): a rank's
//     rows [row0, row0 + n_loc) are one contiguous run and one all-gather of `chunk` entries replicates a result;
//   sharded WITH energy constraints (round 6; train.py:235-300 appends the M energy entries behind the 3N M force entries):
//     rank-major -- rank r's chunk of (3N + 1) per entries = [force entries of its points | their energy entries | padding] --
//     so that a rank's n_loc local rows are STILL one contiguous run and everything between the boundaries (factor rows,
//     GEMVs, all-gathers, CG vector updates over n_pad entries with zero padding) runs unchanged.  The order exists only
//     inside the library: vectors are permuted where they enter / leave (vec_upload / vec_download, pos()).
struct VecLayout {
  int64_t n = 0, n_ff = 0, M = 0, N3 = 0, per = 0;
  int64_t chunk = 0, n_pad = 0, row0 = 0, n_loc = 0;
  int two_seg = 0;
  __host__ __device__ int64_t pos(int64_t g) const {  // reference index -> position in the device vector
    if (!two_seg) return g;
    if (g < n_ff) {
      const int64_t pt = g / N3, r = pt / per;
      return r * chunk + (pt - r * per) * N3 + (g - pt * N3);
    }
    const int64_t e = g - n_ff, r = e / per;
    const int64_t cnt = (M - r * per < per) ? M - r * per : per;
    return r * chunk + cnt * N3 + (e - r * per);
  }
};
VecLayout vec_layout(const gdml_ctx* ctx, int use_E_cstr);
// host vector (reference order, L.n entries) -> zero-padded device vector of L.n_pad entries, and back (synchronous)
int vec_upload(gdml_ctx* ctx, const VecLayout& L, const double* host_ref, double* dev);
int vec_download(gdml_ctx* ctx, const VecLayout& L, const double* dev, double* host_ref, hipStream_t st = nullptr);
int comm_allgather_inplace(gdml_ctx* ctx, double* buf, int64_t chunk);
int comm_allreduce_sum(gdml_ctx* ctx, double* buf, int64_t count);
int comm_allgather_inplace_on(gdml_ctx* ctx, double* buf, int64_t chunk, hipStream_t st);
int comm_broadcast_on(gdml_ctx* ctx, double* buf, int64_t count, int root, hipStream_t st, bool second_comm = false);
void comm_destroy(gdml_ctx* ctx);
int comm_ensure_second(gdml_ctx* ctx);
// A rank that fails locally between collectives (a launch error in a panel loop) must not leave its peers blocked in
// the collective it will never join: RCCL communicators are aborted (ncclCommAbort: the peers' pending and later
// operations end with an error), and this context refuses further collectives.
void comm_abort(gdml_ctx* ctx);
static inline bool comm_active(const gdml_ctx* ctx) { return (ctx->comm || ctx->host_allreduce) && !ctx->virtual_rank; }
bool assemble_wave_applicable(const gdml_ctx* ctx);
bool assemble_strip_applicable(const gdml_ctx* ctx);
int assemble_strip_launch(gdml_ctx* ctx, double sig, double* K, int64_t ld, int lower, double lam);
int assemble_erows_cyclic_launch(gdml_ctx* ctx, double sig, double lam, double* K, int64_t ld, int cyc_W, int cyc_rank,
                                 int cyc_nb);
int assemble_cyclic_launch(gdml_ctx* ctx, double sig, double lam, double* K, int64_t ld, int cyc_W, int cyc_rank,
                           int cyc_nb);
bool assemble_perm2_applicable(const gdml_ctx* ctx);
bool assemble_big1_applicable(const gdml_ctx* ctx);  // assemble_big1.hip: P = 1, 22 <= N <= 256, dense column ranges
int assemble_big1_launch(gdml_ctx* ctx, double sig, int64_t j0, int64_t n_j, int64_t col0, double* K, int64_t ld, int64_t i_beg,
                         int64_t i_end, int lower, double lam);
int assemble_perm2_launch(gdml_ctx* ctx, double sig, int64_t j0, int64_t n_j, int64_t col0, double* K, int64_t ld, int64_t i_beg,
                          int64_t i_end, int lower, double lam, const int32_t* d_jlist = nullptr);
int assemble_perm_launch(gdml_ctx* ctx, double sig, int use_E, const int32_t* d_jlist, const int32_t* d_colmap, int64_t j0,
                         int64_t n_j, int64_t col0, double* K, int64_t ld, int64_t i_beg, int64_t i_end, int lower, double lam,
                         int cyc_W, int cyc_rank, int cyc_nb, const int32_t* h_colmap = nullptr);
bool assemble_pts_applicable(const gdml_ctx* ctx);
int assemble_pts_launch(gdml_ctx* ctx, double sig, int use_E, int64_t j0, int64_t n_j, double* K, int64_t ld, int64_t i_beg,
                        int64_t i_end, int lower, double lam);
int assemble_wave_launch(gdml_ctx* ctx, double sig, int use_E, const int32_t* d_jlist,
                         const int32_t* d_colmap, int64_t j0, int64_t n_j, double* K, int64_t ld,
                         int64_t i_beg, int64_t i_end, int lower = 0, double lam = 0.0, int cyc_W = 0,
                         int cyc_rank = 0, int cyc_nb = 0);
