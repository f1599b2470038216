/*
 * gdml_hip.h -- C ABI of libgdml_hip.so: the MI355X (gfx950) implementation of sGDML's
 * kernel linear-algebra hot path.
 *
 * This is the drop-in boundary.  Every entry point replaces one
 * NumPy / SciPy / PyTorch call site of the reference (cited as file:line relative to the
 * reference repository root); the Python classes in sgdml_amd/ bind them with ctypes and
 * keep the reference's public API (GDMLTrain / GDMLPredict / Analytic / Iterative).
 *
 * Conventions
 *   - plain C: opaque context pointer, host pointers to float64 / int64 row-major (C-order)
 *     arrays owned by the caller unless a parameter is documented as a DEVICE pointer;
 *   - every function returns 0 on success or a negative gdml_status; it never throws and
 *     never calls exit(); gdml_last_error() returns a human-readable description;
 *   - one context per GPU; a context is not thread-safe, different contexts are independent;
 *   - all arithmetic is IEEE float64 (the reference computes in float64 everywhere:
 *     sgdml/torchtools.py:49, sgdml/train.py:1484, sgdml/predict.py:80);
 *   - N = atoms, D = N(N-1)/2 descriptor entries in np.tril_indices(N,-1) order
 *     (sgdml/utils/desc.py:264), M = training points, P = permutations, dim_i = 3N.
 */
#ifndef GDML_HIP_H
#define GDML_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gdml_ctx gdml_ctx;

typedef enum {
  GDML_OK = 0,
  GDML_ERR_INVALID = -1,     /* bad argument (maps to ValueError / AssertionError)        */
  GDML_ERR_HIP = -2,         /* HIP runtime error                                          */
  GDML_ERR_OOM = -3,         /* device allocation failed (reference: OOM re-batching,
                                sgdml/torchtools.py:349-387; here the host decides)        */
  GDML_ERR_STATE = -4,       /* call order violated (e.g. solve before assemble)          */
  GDML_ERR_NOT_PD = -5,      /* Cholesky met a non-positive pivot (LinAlgError analogue,
                                sgdml/solvers/analytic.py:101)                            */
  GDML_ERR_UNSUPPORTED = -6, /* size beyond what this build's kernels support             */
  GDML_ERR_COMM = -7         /* RCCL failure                                               */
} gdml_status;

/* ---- library / context --------------------------------------------------------------- */

/* ABI version of this header (bumped on any signature change). */
int gdml_abi_version(void);

/* Number of visible HIP devices (reference: torch.cuda.device_count(), train.py:1464). */
int gdml_device_count(int* n_out);

/* PCI bus id ("0000:c1:00.0") of visible device `device` into out[len >= 16]: the physical identity of a GPU, so that
 * ranks whose launcher gave each of them a private one-device view can still tell whether they share a GPU (RCCL
 * refuses duplicate devices).  No reference counterpart (the reference is single-process). */
int gdml_device_pci_bus_id(int device, char* out, int len);

/* Create / destroy the per-GPU context: owns one compute stream, one copy stream, all
 * device buffers and (after gdml_comm_init) the RCCL communicator. */
int gdml_ctx_create(int device, gdml_ctx** ctx_out);
int gdml_ctx_destroy(gdml_ctx* ctx);

/* Last error text of a context (ctx may be NULL: text of the last failed gdml_ctx_create). */
const char* gdml_last_error(const gdml_ctx* ctx);

/* Block until all work queued on the context's streams has finished. */
int gdml_sync(gdml_ctx* ctx);

/* Device memory currently held by the context, and free/total HBM of its device (bytes). */
int gdml_mem_info(gdml_ctx* ctx, int64_t* held, int64_t* free_b, int64_t* total_b);

/* Process-level device arena: reserve ONE block of `bytes` on the context's device and keep it until the process ends
 * (or until gdml_mem_reserve(ctx, 0, ...), which fails with GDML_ERR_STATE while any buffer is carved from it).  The large
 * buffers (>= 1 GiB) of every context on that device -- the kernel matrix, the Nystroem matrix, the fp32 copy of the
 * preconditioner factor -- are carved from it first-fit instead of hipMalloc / hipFree, which cost seconds per call beyond
 * ~128 GB on this driver; the block survives gdml_ctx_destroy.  A request that fits no gap goes to hipMalloc.  gdml_mem_info
 * counts the idle part of the arena as free.  reserved_out: bytes now held.
 * (The reference sizes its work to host RAM, sgdml/train.py:949-964; there is nothing to mirror here.) */
int gdml_mem_reserve(gdml_ctx* ctx, int64_t bytes, int64_t* reserved_out);

/* Elapsed milliseconds (HIP events on the compute stream) of the most recent call of the
 * named phase: "desc", "assemble", "factor", "solve", "predict", "matvec", "precon".
 * Also returns how many kernel launches the phase issued. */
int gdml_phase_ms(gdml_ctx* ctx, const char* phase, double* ms_out, int64_t* launches_out);

/* Per-kernel timing for roofline reports: after gdml_profile(ctx, 1) every launch of the hot
 * kernels is bracketed by HIP events on the compute stream; gdml_kernel_stat returns the summed
 * launch duration, the launch count and the summed ALGORITHMIC work (flops for "gemm_nt_sub",
 * bytes for "assemble" and "predict") since profiling was enabled.  Off by default. */
int gdml_profile(gdml_ctx* ctx, int enable);
int gdml_kernel_stat(gdml_ctx* ctx, const char* kernel, double* ms_out, int64_t* launches_out,
                     double* work_out);

/* Tuning / ablation options of a context (all have built-in defaults; nothing here changes results beyond
 * rounding).  Read at the point of use, so a value set between two calls applies to the next call.  The
 * library reads no environment variables except GDML_OPTIONS="key=value,..." (applied at gdml_ctx_create).
 *   asm.wave (1)          register-resident assembly kernel for P = 1, N <= 21 (0: the general kernel); asm.j_chunk its
 *                         column points per workgroup
 *   asm.lower (1)         analytic path: assemble only blocks on/below the diagonal, as -K + lam I
 *   asm.strip (1)         64-column-strip assembly kernel (full-line stores) for P = 1, 11 <= N <= 21, all columns
 *   asm.i_chunk (32)      row points walked by one wavefront of the strip kernel
 *   asm.pts (1; 2 = also for P = 1) small molecules (8 <= N <= 24) with a permutation group: whole-point strips, producer / consumer
 *                         wavefronts (csrc/assemble_pts.hip); asm.pts_nv (0 = automatic) producer wavefronts, asm.pts_nt (1) non-temporal stores of K, asm.pts_xcd (1) adjacent strips on one XCD, asm.pts_i_chunk (64)
 *                         row points per workgroup, asm.pts_debug (0) timing-only ablation mask (results are wrong when set)
 *   general assembly kernel (any permutation group, any N: column-atom strips, csrc/assemble_perm.hip):
 *   asm.perm_level (-1 = automatic: 0 nothing, 1 row-point image, 2 + G_j strip, 3 + x_j tables resident in LDS),
 *   asm.perm_debug (0)    timing-only ablation mask of that kernel (results are wrong when set)
 *   asm.perm_w (0 = automatic: 4 wavefronts per workgroup for N <= 24, else 8), asm.perm_lds_kb (80: per workgroup of 4),
 *   asm.perm_nimg (0 = automatic), asm.perm_pg (0 = automatic), asm.perm_na (0 = automatic), asm.perm_fast_store (1),
 *   asm.perm_i_chunk (16) its shape: image buffers, permutations per group, row atoms per wavefront, transposed
 *                         full-line stores, row points per workgroup
 *   asm.perm_compact (1)  index-list columns: strips of the REQUESTED column atoms instead of all atoms of every point touched
 *   asm.perm_lds_rows (1) permutation entries from the LDS copy for 64 < N <= 128 (always beyond 128 atoms); 0 = two lane-held rows (A/B)
 *   asm.big1 (1)          no permutation group and 22 <= N <= 256, dense column range: the direct P = 1 kernel (assemble_big1.hip,
 *                         round 6) instead of the general permutation kernel; 0 = the general kernel (A/B)
 *   asm.perm2 (1)         molecules with a permutation group of at least asm.perm2_min_p (6) elements, asm.perm2_min_n (36) <= N <= 42
 *                         (groups below 16 elements: from 4 atoms more), dense column ranges -- where it beats the general kernel
 *                         (1.8x at N = 42, P = 27): outer products on
 *                         the fp64 MFMA pipe, contributions of atom pairs no permutation moves summed once per block
 *                         (csrc/assemble_perm2.hip); asm.perm2_split (1; 0 = every pair per permutation), asm.perm2_post (1: single and diagonal
 *                         terms that involve an atom no permutation moves summed over the permutations once per block; 0 = per permutation), asm.perm2_ed (1: diagonal terms of
 *                         moved atoms summed per (row atom, column atom) pair by idle lanes; needs asm.perm2_post), asm.perm2_es (1: single terms of
 *                         moved x moved blocks summed over the permutations once per block by four lanes per block; needs asm.perm2_post), asm.perm2_direct (1: finished rows stored straight
 *                         from the registers, 24 bytes per lane and row; 0 = through LDS in four passes of full-row stores), asm.perm2_chunk (12) pair
 *                         entries per lane and task, asm.perm2_i_chunk (16) row points per workgroup, asm.perm2_debug (0)
 *                         timing-only ablation mask (results are wrong when set)
 *   gemm.debug (0)        ablation mask of the GEMM kernel (separate instantiation; 0 = production kernel)
 *   gemm.persist (0)      1: fused trailing updates of at least four rounds run as 2 resident workgroups per CU that pull tiles
 *                         from per-XCD counters (round 6; measured 1.5 % slower than one workgroup per tile: off)
 *   gemm.n64 (0)          1: trailing updates on 128 x 64 tiles with three workgroups per CU (A/B: profiles/r06_gemm_n64_ab.txt)
 *   gemm.fill_tiles (0)   prediction contractions (D > 256): 128 x 64 tiles when they fill the chip better than 128 x 128 ones
 *                         (measured slower on the launch it was meant for: profiles/r06_matvec_probe.txt)
 *   predict.wide_pad (1)  the same contractions on tables / queries padded to whole tiles (no edge tiles); 0 = round 5's shapes
 *   predict.tn_fill (1)   split count of the prediction back contraction (D > 256) chosen so that its units fill whole rounds of the
 *                         chip (0 = round 5's rule: at least three rounds' worth)
 *   nys.syrk_split (1)    Gram matrix K_nm^T K_nm of the Nystroem build cut along the rows into up to 8 partial sums when its tile
 *                         count is only a few rounds of the chip (a tile's k loop is as long as the factor is tall); 0 = one pass
 *   nys.trsm_left (1)     tall triangular solves of the Nystroem build left-looking (one deep product per 512-column strip); 0 = right-looking
 *   gemm.trace (0)        k > 0: the k-th fused launch runs the traced instantiation and leaves gemm_trace.bin (tools/gemm_trace.py)
 *   gemm.nt_c (0)         non-temporal loads / stores of the C tile (after rocBLAS's Tensile kernel for this shape:
 *                         profiles/r03_vendor_kernels.txt; no gain measured)
 *   gemm.lds16 (3)        fused GEMM launches: 3 = the production loop (16-byte LDS layout, operand pairs of the next half k-tile
 *                         requested 16 MFMAs ahead, last k-tile peeled); 2 = the same with the last k-tile inside the loop (A/B).
 *                         The 8-byte layout, the late-commit and the load-subtract-store variants of rounds 2-3 are gone from
 *                         the fused kernel (their A/Bs: profiles/r03_gemm_*.txt)
 *   chol.nb (512: a multiple of 64 up to 512, anything else falls back to 512), chol.fused_diag (1), chol.fused_min_rows (12288), chol.panel_kernel (1), chol.panel_fused (1),
 *   chol.small_update (1) rank-64 updates inside the panel chain through rank64_update_kernel (0: the GEMM tile kernel; A/B)
 *   chol.lookahead (1)    factorisation schedule (fused_diag = 0: the round-1 second-stream look-ahead schedule)
 *   chol.outer (1024)     panel pairs: K = 2 nb trailing update in two launches (= chol.nb: single panels only)
 *   chol.outer_min_rows (16384)  trailing rows below which new panels are single again
 *   chol.block (0)        W > 0 (multiple of 1024, >= 2048, n >= 3 W): two-level factorisation -- column blocks of W columns, each
 *                         factored with all rows below it carried along, then one lower update of depth W (0.90 of the peak instead
 *                         of 0.86) for everything to its right; the blocks' own factorisation eats the gain: 0.6-1.3 % slower
 *                         (profiles/r06_chol_block.txt).  chol.block_f (2): weight of rows x columns in a tall block's thresholds
 *   chol.merge_gemm1 (1)  the pair's own columns as first super-tile column of the trailing-update launch
 *   chol.tail_lookahead (1)  tail: step chain of the next panel on the high-priority stream
 *   trsm.debug (0)        timing-only ablation mask of the row-local panel solve (results are wrong when set)
 *   trsv.persist (1)      backward substitution as one persistent launch
 *   predict.wave_only (0), predict.mfma (1), predict.mfma_wide (1), predict.fill   prediction kernel choice
 *   predict.fused (1)     gdml_predict with 1-8 host geometries (D <= 256, at most 384 coordinates): ONE launch -- geometries in through the kernel
 *                         arguments, descriptors + contraction + last-workgroup reduction + back-projection, E / F out through
 *                         host-mapped memory (0: descriptor kernel, contraction, epilogue and three copies); predict.fused_rows
 *                         (16) table rows per wavefront, predict.fused_spin (1) completion by polling the kernel's sequence
 *                         number instead of synchronising the stream.  This path records no "predict" phase time
 *   lu.nb (64)            panel width of the LU fallback
 *   comm.force_collectives (0)  issue collectives even for world == 1 without a communicator (tests)
 *   dist.nb (512)         row-block size of the distributed Cholesky (multiple of 128)
 *   dist.lookahead (1 from two ranks on, else 0)  distributed Cholesky: 1 = one panel of look-ahead over three streams (block
 *                         broadcasts on a second communicator); 0 = every step in order on the compute stream
 *   dist.force_panels (0) 1: a one-rank gdml_dist_chol_solve keeps the K = 512 panel schedule of the distributed code instead of
 *                         the single-GPU factorisation it otherwise degenerates to (tests / probes)
 *   nys.force_qr (0)      take the alternative (QR-equivalent) branch of the second Nystroem factorisation (tests)
 *   nys.force_fail (0)    treat the first k attempts of the jitter-stabilised Cholesky of K_mm as failed (tests)
 *   pcg.depth (2)         PCG iterations queued ahead of the host's convergence test / callback (0 = synchronous)
 *   pcg.gemv_plain (0)    preconditioner GEMVs with plain instead of non-temporal loads of the streamed factor (A/B)
 *   pcg.precon_form (2)   application of the Nystroem preconditioner: 0 = stored n x m fp64 factor streamed twice per application
 *                         (the reference's form, iterative.py:120-140); 3 = the factor stored in fp32 plus an m x m Gram
 *                         correction (half the bytes per application; the exact Woodbury inverse on the rounded factor's column
 *                         space: csrc/cg.hip); 2 = automatic: 3 once the factor reaches 1 GiB per rank, else 0; 1 = matrix-free
 *                         (two kernel mat-vecs + an m x m matrix; experimental: does not converge at lam = 1e-10, never automatic)
 *   pcg.f32_rows_per (1024), pcg.f32_rw (4)  shape of the fp32 form's two GEMVs: rows per partial sum of X32^T v, rows per
 *                         wavefront of X32 t (A/B)
 *   pcg.f32_gram_rows (2048)  fp32 form: rows per chunk of the compensated Gram sum of the rounded factor
 *   pcg.f32_inplace (1)   fp32 form: the rounded factor overlays the fp64 factor in the matrix buffer (no second n x m buffer; the
 *                         leverage scores are cached first); 0 = a separate fp32 copy beside the fp64 factor (round 5)
 *   pcg.f32_min_pivot (1e-7)  fp32 form: smallest squared Cholesky pivot of the rounded factor's Gram matrix below which the
 *                         reference's fp64 form is kept (gdml_get_option("pcg.f32_last_min_pivot") reads the last value seen)
 * Unknown keys return GDML_ERR_INVALID. */
int gdml_set_option(gdml_ctx* ctx, const char* key, double value);
int gdml_get_option(gdml_ctx* ctx, const char* key, double* value_out, int* is_set_out);

/* ---- descriptors  (replaces Desc.from_R, sgdml/utils/desc.py:288-365, :208-239) --------
 * R (M,3N) -> R_desc (M,D) = 1/|r_i - r_j|, R_d_desc (M,D,3) = (r_i - r_j)/d^3.
 * lat / lat_inv: 3x3 row-major lattice (columns = vectors) and its inverse, or both NULL;
 * with a lattice the pair differences are wrapped to the minimum image first
 * (desc.py:44-77, round-half-even like np.around). */
int gdml_desc_from_R(gdml_ctx* ctx, const double* R, int64_t M, int N, const double* lat,
                     const double* lat_inv, double* R_desc_out, double* R_d_desc_out);

/* ---- pairwise atom matching of the symmetry search  (replaces the pool of sgdml/utils/perm.py:200-221 and its worker _bipartite_match_wkr, :53-92;
 * SURVEY.md 8(f)4).  One wavefront per pair i < j of the M geometries: cost[a][b] = -sum_k absv_i[a][k] absv_j[b][k], atoms of
 * different species pushed above every same-species entry, the linear assignment by the shortest-augmenting-path algorithm
 * scipy.optimize.linear_sum_assignment implements, kept when it brings geometry i's distance matrix closer to j's
 * (perm.py:76-89: after < before and not numpy.isclose).
 *   absv (M,N,N): |eigenvectors| of the distance matrices, columns by decreasing eigenvalue, or NULL (computed on the device);
 *   adj (M,N,N): distance matrices;
 *   species (N) int32;  cost_out (M,M): entry (i,j), i < j = the pair's remaining distance (the rest 0);
 *   found_ij (capacity,2), found_perm (capacity,N) int32: the kept assignments, in no particular order; *n_found their number --
 *   if it exceeds `capacity` only the first `capacity` were stored (call again with more room; M (M-1)/2 always suffices). */
/* |eigenvectors| (M,N,N) of M symmetric N x N matrices, columns by decreasing eigenvalue (replaces the per-geometry
 * numpy.linalg.eig of perm.py:183-187): one workgroup per matrix, cyclic two-sided Jacobi in LDS.  gdml_perm_match runs the same
 * kernel when `absv` is NULL. */
int gdml_sym_eig_absv(gdml_ctx* ctx, const double* adj, int64_t M, int N, double* absv_out);
int gdml_perm_match(gdml_ctx* ctx, const double* absv, const double* adj, const int32_t* species, int64_t M, int N,
                    double* cost_out, int32_t* found_ij, int32_t* found_perm, int64_t capacity, int64_t* n_found);

/* ---- training set residency (replaces the H2D copies at sgdml/train.py:1447-1448) -------
 * tril_perms is the (P,D) descriptor-permutation table, i.e. the un-linearised form of
 * tril_perms_lin (train.py:897-904): tril_perms[p][k] = tril_perms_lin[k*P+p] - p*D.  Each
 * row must be induced by an atom permutation (Desc.perm, desc.py:509-539); the library
 * recovers the atom permutations and rejects anything else with GDML_ERR_INVALID. */
int gdml_train_upload(gdml_ctx* ctx, const double* R_desc, const double* R_d_desc, int64_t M,
                      int N, const int64_t* tril_perms, int P);

/* ---- kernel matrix assembly (replaces GDMLTrain._assemble_kernel_mat, train.py:1260-1535,
 *      worker train.py:97-302, torch backend torchtools.py:110-392) -----------------------
 * Builds the UN-negated kernel matrix with rows = 3N*M (+M if use_E_cstr) and the selected
 * columns, element order exactly as train.py:1535, into a device buffer owned by the
 * context (it stays there for gdml_chol_factor / gdml_nystroem_*), and optionally copies
 * it to K_host_out (row stride ldk doubles, at least n_cols) when that is not NULL.
 *   col_kind GDML_COLS_ALL    : all columns (col_a, col_b, idx ignored)
 *   col_kind GDML_COLS_POINTS : columns of training points [col_a, col_b)  (whole 3N blocks;
 *                               the reference's slice path train.py:1357-1374)
 *   col_kind GDML_COLS_INDEX  : n_idx sorted unique global column indices idx[] (may include
 *                               energy-constraint columns >= 3N*M; train.py:1376-1407)
 * alloc_extra_rows: extra (uninitialised) rows appended to the device matrix and to the
 * host copy's shape contract (train.py:1418,1484). */
enum { GDML_COLS_ALL = 0, GDML_COLS_POINTS = 1, GDML_COLS_INDEX = 2 };
int gdml_assemble_K(gdml_ctx* ctx, double sig, int use_E_cstr, int col_kind, int64_t col_a,
                    int64_t col_b, const int64_t* idx, int64_t n_idx, int64_t alloc_extra_rows,
                    double* K_host_out, int64_t ldk);

/* Analytic path (Analytic.solve, analytic.py:65-94): the system matrix A = -K + lam I in the form
 * gdml_chol_factor consumes, assembled in one pass -- sign flip (analytic.py:65) and regularisation
 * (analytic.py:82) fused into the stores, and only the blocks on/below the block diagonal written (the
 * factorisation never reads the strict upper triangle): 4 n^2 bytes of HBM traffic instead of 8 n^2 written +
 * 8 n^2 re-read and re-written.  All columns, matrix stays on the device; alloc_extra_rows as above (1 for
 * gdml_chol_set_rhs).  Where the fused form is not available (permutations, energy constraints, N > 21) this is
 * gdml_assemble_K(GDML_COLS_ALL) and gdml_chol_factor applies sign and shift itself.  gdml_chol_factor must be
 * called with the same lam. */
int gdml_assemble_A(gdml_ctx* ctx, double sig, double lam, int use_E_cstr, int64_t alloc_extra_rows);

/* Shape of the device-resident matrix produced by the last gdml_assemble_K. */
int gdml_K_shape(gdml_ctx* ctx, int64_t* n_rows, int64_t* n_cols, int64_t* extra_rows);

/* ---- analytic solve (replaces Analytic.solve, sgdml/solvers/analytic.py:65-99) ----------
 * Requires a square device-resident K from gdml_assemble_K(GDML_COLS_ALL).
 * gdml_chol_factor: A = -K + lam*I, in-place blocked Cholesky A = L L^T on fp64 MFMA.
 *   *info = 0 ok; > 0: leading minor of that order is not positive definite (LAPACK dpotrf
 *   convention, what scipy.linalg.cho_factor raises LinAlgError for, analytic.py:94-101);
 *   the function then returns GDML_ERR_NOT_PD and K is destroyed (re-assemble to retry).
 * gdml_chol_solve: alphas = -(A^-1 y)   (analytic.py:97-99).
 * n_refine > 0 adds that many steps of iterative refinement with the matrix-free kernel
 * operator (needs the training set of gdml_train_upload; 0 = exactly LAPACK semantics).
 * gdml_chol_set_rhs (optional, between gdml_assemble_K(..., alloc_extra_rows >= 1) and
 *   gdml_chol_factor): hands the right-hand side over before the factorisation.  y is carried as an extra
 *   row of the matrix, so the panel solves and trailing updates of the factorisation perform the forward
 *   substitution L z = y on the way (cho_solve's first triangular solve, analytic.py:97, for free);
 *   gdml_chol_solve(y = NULL) then only runs the backward substitution. */
int gdml_chol_set_rhs(gdml_ctx* ctx, const double* y, int64_t n);
/* LU branch of Analytic.solve (analytic.py:101-114: scipy.linalg.solve = LAPACK dgesv, taken when cho_factor
 * raised LinAlgError): needs the FULL un-negated K of gdml_assemble_K(GDML_COLS_ALL) (assemble again after a
 * failed gdml_chol_factor, which destroys the matrix), forms A = -K + lam I on both triangles, factors
 * P A = L U with partial pivoting on the device (blocked, trailing updates on fp64 MFMA) and returns
 * alphas = -(A^-1 y).  *info > 0: U(info,info) is exactly zero (dgetrf convention; scipy raises "Matrix is
 * singular") and the function returns GDML_ERR_NOT_PD.  The matrix is consumed.
 * The least-squares branch (analytic.py:138) only runs for a non-square K, which Analytic.solve never builds. */
int gdml_lu_solve(gdml_ctx* ctx, double lam, const double* y, int64_t n, double* alphas_out, int* info);
int gdml_chol_factor(gdml_ctx* ctx, double lam, int* info);
int gdml_chol_solve(gdml_ctx* ctx, const double* y, int64_t n, int n_refine, double* alphas_out);

/* ---- prediction (replaces GDMLPredict.__init__/set_alphas/predict,
 *      sgdml/predict.py:426-447, :551-601, :1146-1294; worker :84-245; torch :877-1046) ----
 * gdml_predict_upload_model: training descriptors R_desc (M,D) [note: the model file stores
 *   its transpose, train.py:807], R_d_desc_alpha (M,D), tril_perms (P,D), sig, optional
 *   alphas_E (M) (NULL if the model has no energy constraints).  Builds the permuted tables
 *   of predict.py:426-441 on the device.
 * gdml_set_alphas: training-mode re-targeting (predict.py:551-601): needs the Jacobians of
 *   gdml_train_upload; computes J_m alpha_m on the device (desc.py:368-385).
 * gdml_predict: R (B,3N) host geometries, or R == NULL for the training-set mode
 *   (predict.py:1221-1233) which uses the descriptors of gdml_train_upload.  Outputs are
 *   UNSCALED (host applies std and c exactly as predict.py:1286-1288): E_out (B) may be NULL
 *   (return_E=False), F_out (B,3N). */
int gdml_predict_upload_model(gdml_ctx* ctx, const double* R_desc, const double* R_d_desc_alpha,
                              int64_t M, int N, const int64_t* tril_perms, int P, double sig,
                              const double* alphas_E);
int gdml_set_alphas(gdml_ctx* ctx, const double* alphas_F, const double* alphas_E);
int gdml_predict(gdml_ctx* ctx, const double* R, int64_t B, const double* lat,
                 const double* lat_inv, double* E_out, double* F_out);

/* Device-resident variant used by benchmarks and MD loops: R_dev / E_dev / F_dev are DEVICE
 * pointers (hipMalloc'd by the caller or gdml_dev_alloc); nothing crosses PCIe. */
int gdml_predict_dev(gdml_ctx* ctx, const double* R_dev, int64_t B, const double* lat,
                     const double* lat_inv, double* E_dev, double* F_dev);

/* Test / validation error sums evaluated on the device (replaces the body of the reference's
 * cli.test loop, sgdml/cli.py:1564-1605 with _online_err :1170): predicts B host geometries R,
 * scales with std and c like predict.py:1286-1288, compares with the labels and returns
 *   sums8_out = { sum|dE|, sum dE^2, sum|dF|, sum dF^2, sum|d|F||, sum(d|F|)^2, sum a, sum a^2 },
 * a = arccos(clip(cos(f_pred, f_ref)))/pi per atom.  E_ref may be NULL (sums 0, 1 are then 0).
 * The caller divides by the sample sizes exactly as _online_err does. */
int gdml_predict_errors(gdml_ctx* ctx, const double* R, int64_t B, const double* lat,
                        const double* lat_inv, double std, double c, const double* E_ref,
                        const double* F_ref, double* sums8_out);

/* Kernel mat-vec  out = K v - lam v  through the prediction contraction (replaces
 * Iterative._K_vec, sgdml/solvers/iterative.py:183-204).  n = 3NM (+M with E constraints). */
int gdml_kernel_matvec(gdml_ctx* ctx, double lam, int use_E_cstr, const double* v, int64_t n,
                       double* out);

/* ---- iterative solver (replaces sgdml/solvers/iterative.py) ------------------------------
 * gdml_nystroem_factor (iterative.py:208-351 + :414-471): requires the (n+m) x m matrix of
 *   gdml_assemble_K(GDML_COLS_INDEX, idx, m, alloc_extra_rows = m).  Computes on the device
 *   L^-1 K_mn (m x n) with the jitter-escalation semantics of _cho_factor_stable, keeps it
 *   resident as the preconditioner, returns the leverage scores (column squared norms,
 *   iterative.py:107-109) in lev_scores_out (n) and optionally the factor in
 *   LinvKmn_host_out (m x n row-major, may be NULL).  *info: bits 8.. = number of jitter escalations the Cholesky of
 *   K_mm needed (iterative.py:442-463); bit 0 = the second Cholesky failed and
 *   the alternative branch ran (the reference's QR of [K_nm; sqrt(lam) I], iterative.py:313-324; here a
 *   shifted CholeskyQR3 on fp64 MFMA with the same R^T R up to rounding).
 *   lev_scores_out may be NULL: the scores are then computed on demand by gdml_nystroem_lev_scores (valid until the next
 *   assembly overwrites the matrix).  *info bit 1 / bit 2: the preconditioner will be applied matrix-free / from the fp32
 *   copy of the factor (option pcg.precon_form); the matrix-free form skips the second tall triangular solve of the factor
 *   until somebody asks for the scores.
 * gdml_precon_apply: out = (L^T L v - v)/lam  (iterative.py:120-140).  Matrix-free form: with X = L^-1 K_mn = Z^T K_mn,
 *   Z = L_mm^-T L^-T (m x m), L^T L v = K_nm Z Z^T K_mn v: K_mn v is gdml_kernel_matvec followed by a gather of the m inducing
 *   entries, K_nm t a scatter followed by the mat-vec; the n x m factor is not read.  fp32 form: (X32 T0 X32^T v - v)/lam.
 * gdml_pcg: preconditioned CG for (-K + lam I) x = y with scipy.sparse.linalg.cg semantics
 *   (iterative.py:740-752: rtol*||y||, atol = 0, x0 optional).  Vectors and CG scalars stay on the device; the host
 *   reads the residual of an iteration only when it queues the iteration `pcg.depth` (default 2) steps later, so the GPU
 *   never waits for the host, and returns exactly the iterate scipy would (the iterates in flight live in a ring).
 *   cb(iter, resid, user) is called every cb_every iterations (0 = never) with resid = ||y - A x_iter|| (the recurrence
 *   residual after the update, what the reference's callback reads from scipy's frame, iterative.py:626-632); a non-zero
 *   return stops the solve with x_iter (used for CGRestartException, iterative.py:729).  Inside the callback -- and only
 *   there -- gdml_pcg_x copies x_iter to the host (checkpoints, iterative.py:675-735; restarts): the iterate does not
 *   cross PCIe unless somebody asks for it.
 *   info_out: 0 converged, 1 maxiter reached, 2 stopped by callback. */
typedef int (*gdml_pcg_cb)(int64_t iter, double resid, void* user);
int gdml_nystroem_factor(gdml_ctx* ctx, double lam, const int64_t* idx, int64_t m,
                         double* lev_scores_out, double* LinvKmn_host_out, int* info);
int gdml_nystroem_lev_scores(gdml_ctx* ctx, double* lev_scores_out);
int gdml_precon_apply(gdml_ctx* ctx, double lam, const double* v, int64_t n, double* out);
int gdml_pcg(gdml_ctx* ctx, double lam, int use_E_cstr, const double* y, const double* x0,
             int64_t n, double rtol, int64_t maxiter, int use_precon, gdml_pcg_cb cb,
             int64_t cb_every, void* user, double* x_out, int64_t* iters_out, double* resid_out,
             int* info_out);
int gdml_pcg_x(gdml_ctx* ctx, double* x_host_out);

/* ---- multi-GPU (new: the reference has no collective) ----------------------
 * One process per GPU.  Rank 0 calls gdml_comm_unique_id (128 bytes) and ships it to the
 * other ranks by any host channel (sgdml_amd/dist.py uses torch.distributed's store); every rank then
 * calls gdml_comm_init (id128 = NULL: "virtual rank" without a communicator, shard arithmetic only --
 * tests).  After that
 *   - the iterative path is sharded over the training points (contiguous row shards):
 *     gdml_assemble_K(GDML_COLS_INDEX, ..., alloc_extra_rows > 0) assembles the rank's rows of K_nm,
 *     gdml_nystroem_factor / gdml_precon_apply / gdml_kernel_matvec / gdml_pcg exchange m- and n-vectors with
 *     RCCL all-reduce / all-gather over xGMI; with use_E_cstr a rank holds the force rows of its points followed by their
 *     energy rows (vectors cross the ABI in the reference order -- forces, then energies -- on every rank; the rank-major
 *     order of the device vectors stays inside the library);
 *   - the analytic path is gdml_dist_chol_solve (block-row-cyclic matrix, below).
 * The collectives are issued for every communicator size, including one rank.  gdml_chol_* / gdml_lu_solve stay
 * single-GPU entry points. */
int gdml_comm_unique_id(void* id128_out);
int gdml_comm_init(gdml_ctx* ctx, const void* id128, int rank, int world);
int gdml_comm_info(gdml_ctx* ctx, int* rank_out, int* world_out);
/* gdml_comm_suspend(ctx, 1) parks the communicator: until gdml_comm_suspend(ctx, 0) the context behaves like a single GPU
 * without one (rank 0 of 1: unsharded assembly, local Cholesky / LU / PCG, no collective).  For work every rank performs
 * redundantly on its own GPU because the sharded solvers do not carry it: the LU branch of a matrix that is not positive
 * definite (analytic.py:101-114).  (Energy constraints, train.py:235-300, are carried by both sharded solvers since round 6.) */
int gdml_comm_suspend(gdml_ctx* ctx, int suspend);

/* Distributed analytic solve (new; Analytic.solve, analytic.py:65-99, for systems beyond one GPU -- BASELINE.json
 * configs[3], [4]): the system matrix A = -K + lam I is assembled block-ROW-cyclic over the ranks of the communicator
 * (512-row blocks, every rank only its own rows: n^2 * 8 / world bytes per GPU), factored by a right-looking blocked
 * Cholesky (diagonal block broadcast, row-local panel solve, panel all-gather over RCCL, local fp64-MFMA trailing
 * update) with the right-hand side carried as a replicated extra row, and solved back; every rank receives
 * alphas = -(A^-1 y).  n = 3N M, or 3N M + M for a system with energy constraints (train.py:235-300: y then carries the M
 * energy labels behind the forces, the energy rows are assembled on the ranks that own them).  Needs gdml_train_upload (any P)
 * and a communicator (gdml_comm_init / gdml_comm_init_host; without one it runs on a single GPU).  *info as gdml_chol_factor. */
int gdml_dist_chol_solve(gdml_ctx* ctx, double sig, double lam, const double* y, int64_t n, double* alphas_out,
                         int* info);

/* Host-staged collectives: the same sharded algorithms with the two collectives delegated to the caller
 * (e.g. torch.distributed's gloo backend).  The library copies the device buffer to pinned host memory,
 * calls the callback, which must complete the collective IN PLACE on that host buffer, and copies it back.
 *   allreduce(buf, count, user): buf[0:count] <- sum over ranks
 *   allgather(buf, chunk, user): buf holds world*chunk doubles, rank r's contribution at [r*chunk, (r+1)*chunk)
 * A non-zero return aborts the calling operation with GDML_ERR_COMM.  Used where RCCL cannot run: several
 * ranks sharing one GPU (tests of the sharded code path on a one-GPU box), or a node without xGMI peers. */
typedef int (*gdml_host_allreduce)(double* buf, int64_t count, void* user);
typedef int (*gdml_host_allgather)(double* buf, int64_t chunk, void* user);
int gdml_comm_init_host(gdml_ctx* ctx, int rank, int world, gdml_host_allreduce allreduce,
                        gdml_host_allgather allgather, void* user);

/* Collectives issued and payload bytes handed to them by this rank since the communicator was set up. */
int gdml_comm_stats(gdml_ctx* ctx, int64_t* calls_out, double* bytes_out);

/* ---- raw device buffers for callers that keep data resident -------------------------- */
int gdml_dev_alloc(gdml_ctx* ctx, int64_t bytes, void** dev_out);
int gdml_dev_free(gdml_ctx* ctx, void* dev);
int gdml_memcpy_h2d(gdml_ctx* ctx, void* dev, const void* host, int64_t bytes);
int gdml_memcpy_d2h(gdml_ctx* ctx, void* host, const void* dev, int64_t bytes);

/* Device pointer + leading dimension of the resident kernel matrix / factor (for tests). */
int gdml_K_dev(gdml_ctx* ctx, double** K_dev_out, int64_t* ld_out);

#ifdef __cplusplus
}
#endif
#endif /* GDML_HIP_H */
