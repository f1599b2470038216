// Analytic solve on gfx950: blocked fp64 Cholesky on the MFMA pipe + triangular solves.
//
// Replaces scipy.linalg.cho_factor / cho_solve as called by Analytic.solve
// (sgdml/solvers/analytic.py:94-99; LAPACK dpotrf/dpotrs underneath) and the Cholesky / TRSM /
// SYRK steps of Iterative._nystroem_cholesky_factor (sgdml/solvers/iterative.py:263-345).
//
// Storage: row-major, LOWER triangle referenced (A = L L^T, L overwrites the lower triangle; the
// strict upper triangle is scratch and may be overwritten with garbage).
//
// Right-looking blocked algorithm, panel width NB (512), ONE stream.  Per panel k:
//   gemm_nt_sub            columns of panel k+1:  C[t0:n, t0:t1] -= P P^T                     (K = NB)
//   gemm_nt_sub_diag       the rest of the trailing matrix (SYRK, lower tiles); workgroup 0 of this launch factors
//                          the NB x NB diagonal block of panel k+1 meanwhile (diag_block_role: potrf64_wg, substitution,
//                          MFMA rank-64 updates) -- the dependent step chain is hidden inside the big launch
//   panel_trsm             row-local solve of the rows below that block, one launch (strip in registers, MFMA updates)
// Near the end (SYRK shorter than the single-workgroup block factorisation) and for the first / ragged panels the
// block is factored by the multi-workgroup 64-wide step chain (potrf_trsm64 + K = 64 GEMM) instead.
// A right-hand side stored as an extra row is carried through (forward substitution for free); the backward
// substitution is one persistent launch with a per-block fallback.  n^3/3 of the flops are in gemm_nt_sub.
// The round-1 schedule (step chain on a second stream with one panel of look-ahead) is still selectable with option
// chol.fused_diag = 0; profiles/r02_panel_fusion_ab.txt has the comparison.  The CU-masked / split-stream / chunked
// variants measured in rounds 1-2 (profiles/r01_chol_timeline_split.txt, r02_sched_probe.txt) are gone.
#include "common.h"
#include <type_traits>
#include <utility>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------
// C[M x N] -= A[M x K] * B[N x K]^T     (row-major, leading dimensions lda/ldb/ldc)
// lower != 0: C is square-symmetric-updated, only tiles with tile_row >= tile_col are computed.
// Workgroup: 256 threads = 4 waves (2 x 2), tile 128 x 128, each wave 64 x 64 = 4 x 4 MFMA tiles.
// LDS: [row][BK+1] doubles per operand and stage: pitch 17 doubles = 34 dwords makes the MFMA operand
// pattern lane -> (row = l&15, k = l>>4) conflict-free for ds_read_b64 AND for the ds_read2_b64 the
// compiler merges them into (16-lane groups, 32 banks: 34 i mod 32 = 2 i), and the 8-byte row writes too.
// ------------------------------------------------------------------------------------------
#define GT 128
#define GBK 16
#define GPITCH 17
#define G6N 64  // tile columns of the 128 x 64 kernel (gemm.n64)
#ifndef GEMM_COMMIT_KS
#define GEMM_COMMIT_KS 12  // k-step after which the prefetched tile is written to LDS (0,4,8,12): 12 = as late as possible,
                           // the global loads of the next tile get the whole tile to arrive (4 -> 12: -1.1 % factorisation time)
#endif

struct GemmArgs {
  const double* A;
  const double* B;
  double* C;
  int64_t M, N, K, lda, ldb, ldc;
  int lower;
  int tiles_m, tiles_n;
  int64_t n_super;  // number of 8x8 super tiles enumerated
  int64_t s_begin;  // first super tile of this launch (a launch may cover a sub-range)
  int super_n;      // super-tile columns
  int aligned;      // A, B 16-byte aligned with even leading dimensions (vector loads legal)
  // block-row-cyclic lower structure (distributed Cholesky): C rows are LOCAL rows, local row block lb (cyc_nb rows)
  // is global block (cyc_lb0 + lb) * cyc_W + cyc_rank; a tile is computed iff its first column (cyc_col0 + col0,
  // global) is not right of its last global row; local rows >= cyc_block_rows (carried right-hand side) take all
  int cyc_W, cyc_rank;
  int64_t cyc_lb0, cyc_nb, cyc_col0, cyc_block_rows;
  // fused launch (gemm_nt_sub_diag_kernel): workgroup 0 factors the next panel's diagonal block meanwhile
  double* diagA;    // top-left element of that nb x nb block (leading dimension ldc), or null
  int diag_nbw;     // nb / 64
  int64_t diag_off; // global index of its first row (LAPACK info)
  int* diag_info;
  int dbg;          // option gemm.debug: ablation bits: 1 no epilogue, 2 no tile loads, 4 no LDS reads, 8 no barrier
  int nt_c;         // option gemm.nt_c: non-temporal loads / stores of the C tile (it is streamed: keep the L2 for the panels)
  // merged trailing update (lower): the first super-tile COLUMN (the next outer panel's own columns) is enumerated
  // first; the tiles of its leading ready_rows x ready_rows tile block (the diagonal block workgroup 0 is about to
  // factor) count themselves into *ready when their C tile is stored, workgroup 0 waits for ready_target
  int col0_first;
  int* ready;
  int ready_target, ready_rows;
  // second, plain (non-symmetric) problem carried by the first tiles2 workgroups of the launch:  C2 -= A2 B2^T,
  // M2 x N2, depth K2, same leading dimensions (the K = NB update of panel b's columns by panel a); tiles above the
  // diagonal of its leading N2 x N2 block are skipped; with `ready` set ITS leading tiles are the counted ones
  const double* A2;
  const double* B2;
  double* C2;
  int64_t M2, N2, K2;
  int tiles2;
  // persistent launch (gemm_nt_sub_persist_kernel): 8 tile counters, one per XCD, 64 bytes apart; total work items
  unsigned* queue;
  int64_t n_items;
  // option gemm.trace (tools/gemm_trace.py): per-tile shader-clock stamps of the traced instantiation, 8 words per workgroup
  unsigned long long* trace;
};

template <bool PIPE, bool CACC, int CKS = GEMM_COMMIT_KS>
__global__ void __launch_bounds__(256, 2) gemm_nt_sub_diag_kernel(GemmArgs g);
__global__ void __launch_bounds__(256, 2) gemm_nt_sub_diag_trace_kernel(GemmArgs g);
template <int CKS, int FLAGS = 2>
__global__ void __launch_bounds__(256, 2) gemm_nt_sub_persist_kernel(GemmArgs g);
__global__ void __launch_bounds__(256, 3) gemm_nt_sub_n64_kernel(GemmArgs g);

template <bool FULL>
__device__ __forceinline__ void gemm_load_tile(const double* __restrict__ G, int64_t ld,
                                               int64_t row0, int64_t nrows, int64_t k0, int64_t K,
                                               int tid, d2 (&r)[4]) {
  // 128 rows x 16 k = 1024 chunks of 2 doubles; thread t -> chunks t, t+256, ...: row = c/8, kc = c%8
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    int cidx = tid + 256 * s;
    int row = cidx >> 3, kc = (cidx & 7) * 2;
    int64_t gr = row0 + row, gk = k0 + kc;
    if (FULL) {
      r[s] = *reinterpret_cast<const d2*>(G + gr * ld + gk);
    } else {
      // branch-free guarded load: clamp the address, then zero what is out of range
      int64_t cr = gr < nrows ? gr : nrows - 1;
      int64_t ck0 = gk < K ? gk : K - 1, ck1 = gk + 1 < K ? gk + 1 : K - 1;
      double v0 = G[cr * ld + ck0], v1 = G[cr * ld + ck1];
      d2 v;
      v.x = (gr < nrows && gk < K) ? v0 : 0.0;
      v.y = (gr < nrows && gk + 1 < K) ? v1 : 0.0;
      r[s] = v;
    }
  }
}

// 16-byte layout (option gemm.lds16): element (row, kq = k / 2) is the pair (k, k + 1) of a row, stored at d2 index
// kq * 128 + (row ^ kq).  One ds_write_b128 per chunk instead of two ds_write_b64, one ds_read_b128 per operand block and
// PAIR of k-steps instead of two 8-byte reads; the XOR keeps both conflict free: a group of 8 store lanes holds one row and
// kq = 0..7 (8 different bank quads), a 16-lane read group holds 16 different (row ^ kq) & 15 (searched by script, see
// DESIGN 3.2).  The MFMA's k index is a label: lane group g = lane >> 4 feeds k = 4 g + s into k-step s (both operands), so
// that a lane's four k of a tile are contiguous.
// Full tiles, 16-byte aligned: wave-uniform base (tile corner + k offset: SGPRs) + the thread's constant 32-bit byte offsets
// (row * ld + kc of its four chunks; a tile spans < 2^31 bytes): saddr-form loads, no 64-bit address arithmetic per k-tile.
// Full tiles, 16-byte aligned: wave-uniform base (tile corner + k offset) + the thread's constant 32-bit byte offsets (row * ld + kc
// of its four chunks; a tile spans < 2^31 bytes).  Round 6 measured the same loads as BUFFER loads (tile corner in a resource, lane
// offset as voffset, k offset as soffset: no 64-bit lane arithmetic at all): 2.0-2.2 % SLOWER factorisation on two boxes
// (profiles/r06_gemm_variants.txt); and issuing them 16 MFMAs later (after the first MFMA group): +5.8 % -- the 48 MFMAs
// between issue and LDS commit are needed.
__device__ __forceinline__ void gemm_load_tile_g(const double* __restrict__ base, const unsigned (&off)[4], d2 (&r)[4]) {
  typedef const __attribute__((address_space(1))) char* gcptr;
  typedef const __attribute__((address_space(1))) d2* gd2ptr;
  const uint64_t p = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
  gcptr b = (gcptr)(((uint64_t)hi << 32) | lo);
#pragma unroll
  for (int s = 0; s < 4; ++s) r[s] = *(gd2ptr)(b + off[s]);
}

__device__ __forceinline__ void gemm_store_tile16(double* __restrict__ S, int tid, const d2 (&r)[4]) {
  d2* S2 = reinterpret_cast<d2*>(S);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int cidx = tid + 256 * s;
    const int row = cidx >> 3, kq = cidx & 7;
    S2[kq * 128 + (row ^ kq)] = r[s];
  }
}

__device__ __forceinline__ void gemm_store_tile(double* __restrict__ S, int tid, const d2 (&r)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    int cidx = tid + 256 * s;
    int row = cidx >> 3, kc = (cidx & 7) * 2;
    S[row * GPITCH + kc] = r[s].x;
    S[row * GPITCH + kc + 1] = r[s].y;
  }
}

// ABL = true only in the ablation instantiation (option gemm.debug != 0): the production kernel carries
// none of the ablation branches.
// FLAGS: 1 = traced instantiation; 2 = persistent caller: lane-derived values are recomputed per tile from an opaque copy of the
// thread index (otherwise they are hoisted out of the tile loop and spilled across the k loop)
template <bool FULL, bool ABL, bool PIPE = false, bool CACC = true, int CKS = GEMM_COMMIT_KS, int FLAGS = 0>
__device__ __forceinline__ void gemm_tile_body(const GemmArgs& g, double (*lds)[2][GT * GPITCH],
                                               int64_t row0, int64_t col0, int64_t trace_slot = 0, unsigned* s_next = nullptr) {
  const int dbg = ABL ? g.dbg : 0;
  // traced instantiation only: stamps of the tile's phases and the time parked before / in the k-tile barrier
  unsigned long long tr_t0 = 0, tr_t1 = 0, tr_t3 = 0, tr_vm = 0, tr_bar = 0, tr_mx = 0;
  constexpr bool TR = (FLAGS & 1) != 0;
  if constexpr (TR) tr_t0 = __builtin_amdgcn_s_memtime();
  int tid_ = threadIdx.x;
  if constexpr ((FLAGS & 2) != 0) asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lk = lane >> 4;

  // C -= A B^T as acc = C; acc += (-A) B^T; C = acc for interior tiles: the 64 C loads per lane are issued with the first
  // operand tiles (their latency is paid once, together with the prologue's), the epilogue is stores only -- instead of
  // four load->store round trips after the last MFMA.  f64 MFMA C/D layout: col = lane & 15, row = (lane >> 4) + 4 r.
  constexpr bool OW = (FLAGS & 4) != 0;  // overwrite: C = -A B^T (C is not read: the caller need not clear it)
  constexpr bool CIN = FULL && !ABL && CACC;
  constexpr double SGN = PIPE ? -1.0 : 1.0;  // 16-byte layout: the accumulator holds -C + A B^T
  double* Cw = g.C + (row0 + wm * 64 + lk) * g.ldc + col0 + wn * 64 + li;
  // wave-uniform tile corner + 32-bit lane offset: the 64 row addresses stay in SGPRs (saddr form), no address VGPRs
  const int wu = __builtin_amdgcn_readfirstlane(wave);
  double* const Ct = g.C + (row0 + (wu >> 1) * 64) * g.ldc + col0 + (wu & 1) * 64;
  const unsigned coff = (unsigned)lk * (unsigned)g.ldc + (unsigned)li;
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (CIN && !OW) {
        if (g.nt_c) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = SGN * __builtin_nontemporal_load((Ct + (int64_t)(i * 16 + 4 * r) * g.ldc) + coff + j * 16);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = SGN * (Ct + (int64_t)(i * 16 + 4 * r) * g.ldc)[coff + j * 16];
        }
      } else {
        acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
      }
    }

  const int64_t nk = (g.K + GBK - 1) / GBK;
  // (a staggered k start per tile, after Tensile's StaggerU, was tried and cost 1.5 %: profiles/r03_gemm_ntc_stagger_ab.txt;
  //  its 64-bit wrap-around compare in the loop head is gone with it)
  auto kofs = [&](int64_t kt) -> int64_t { return kt * GBK; };
  d2 ra[4], rb[4];
  gemm_load_tile<FULL>(g.A, g.lda, row0, g.M, kofs(0), g.K, tid, ra);
  gemm_load_tile<FULL>(g.B, g.ldb, col0, g.N, kofs(0), g.K, tid, rb);
  if constexpr (PIPE) {
    gemm_store_tile16(lds[0][0], tid, ra);
    gemm_store_tile16(lds[0][1], tid, rb);
  } else {
    gemm_store_tile(lds[0][0], tid, ra);
    gemm_store_tile(lds[0][1], tid, rb);
  }
  __syncthreads();
  if constexpr (TR) tr_t1 = __builtin_amdgcn_s_memtime();

  if constexpr (PIPE) {
    // ---- 16-byte LDS layout (gemm_store_tile16): per k-tile two batches of 8 ds_read_b128 + 32 MFMAs.  The sign lives in
    // the accumulator here (acc = -C, acc += A B^T, C = -acc): no negation of A operands in the loop.
    const int c16 = lane & 15, gq = lane >> 4;
    // full tiles: saddr-form loads (wave-uniform tile corner + the thread's four constant 32-bit byte offsets)
    unsigned offA[4], offB[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int cidx = tid + 256 * s4;
      offA[s4] = (unsigned)(((int64_t)(cidx >> 3) * g.lda + (cidx & 7) * 2) * 8);
      offB[s4] = (unsigned)(((int64_t)(cidx >> 3) * g.ldb + (cidx & 7) * 2) * 8);
    }
    const double* const Abase = g.A + row0 * g.lda;
    const double* const Bbase = g.B + col0 * g.ldb;
    auto load_ab = [&](int64_t kt_, d2 (&ra_)[4], d2 (&rb_)[4]) {
      if constexpr (FULL) {
        const int64_t ko = kofs(kt_);
        gemm_load_tile_g(Abase + ko, offA, ra_);
        gemm_load_tile_g(Bbase + ko, offB, rb_);
      } else {
        gemm_load_tile<FULL>(g.A, g.lda, row0, g.M, kofs(kt_), g.K, tid, ra_);
        gemm_load_tile<FULL>(g.B, g.ldb, col0, g.N, kofs(kt_), g.K, tid, rb_);
      }
    };
    if constexpr (CKS == 4 || CKS == 5) {
      // read-ahead form (gemm.lds16 = 2): the operand pairs of the NEXT half are requested while 16 MFMAs of the current
      // one are still to issue, across the tile boundary (the barrier sits before the tile's last 16 MFMAs)
      auto read_half = [&](d2 (&a_)[4], d2 (&b_)[4], int buf, int h) {
        const int kq = 2 * gq + h;
        const int cx = c16 ^ kq;
        const d2* Ap = reinterpret_cast<const d2*>(lds[buf][0]) + kq * 128 + wm * 64 + cx;
        const d2* Bp = reinterpret_cast<const d2*>(lds[buf][1]) + kq * 128 + wn * 64 + cx;
#pragma unroll
        for (int i = 0; i < 4; ++i) a_[i] = Ap[16 * i];
#pragma unroll
        for (int j = 0; j < 4; ++j) b_[j] = Bp[16 * j];
      };
      auto mfma16 = [&](const d2 (&a_)[4], const d2 (&b_)[4], int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(t ? a_[i].y : a_[i].x, t ? b_[j].y : b_[j].x, acc[i][j], 0, 0, 0);
      };
      d2 a0[4], b0[4], a1[4], b1[4];
      read_half(a0, b0, 0, 0);
      const int nk32 = (int)nk;
      if constexpr (CKS == 4) {
        for (int kt = 0; kt < nk32; ++kt) {
          const int cur = kt & 1;
          if (kt + 1 < nk32) load_ab(kt + 1, ra, rb);
          mfma16(a0, b0, 0);
          read_half(a1, b1, cur, 1);
          mfma16(a0, b0, 1);
          mfma16(a1, b1, 0);
          if (kt + 1 < nk32) {
            gemm_store_tile16(lds[cur ^ 1][0], tid, ra);
            gemm_store_tile16(lds[cur ^ 1][1], tid, rb);
          }
          __syncthreads();
          if (kt + 1 < nk32) read_half(a0, b0, cur ^ 1, 0);
          mfma16(a1, b1, 1);
        }
      } else {
        // CKS = 5 (production): the same schedule with the last k-tile peeled -- no conditionals inside the loop, -0.3 % of the
        // factorisation (1322 -> 1318 and 1289 -> 1286 ms on two boxes, profiles/r04_gemm_peel_stagger_ab.txt).  The schedule
        // is pinned by a sched_barrier at EVERY phase boundary: with one missing (between the first MFMA group and the
        // read-ahead) a later, unrelated edit of this file made the scheduler sink the reads behind two groups and the kernel
        // lost 3 % (1322 vs 1282 ms on one box) without any change to this loop's source.  Committing the prefetched tile to
        // LDS 16 MFMAs earlier (so that its ds_writes have drained at the barrier) was measured with it: +0.3 %, not kept.
        auto step = [&](int kt, auto has_next_t) {
          constexpr bool HN = decltype(has_next_t)::value;
          const int cur = kt & 1;
          // (sched_barrier: without the loop's conditionals the machine scheduler hoists the commit and the barrier to the
          //  20th MFMA and sinks the read-ahead behind the last one)
          if constexpr (HN) load_ab(kt + 1, ra, rb);
          __builtin_amdgcn_sched_barrier(0);
          mfma16(a0, b0, 0);
          __builtin_amdgcn_sched_barrier(0);  // (between the groups too: in one region the reads were sunk behind BOTH groups and
          read_half(a1, b1, cur, 1);          //  the third group waited for them -- +3 % factorisation time on a fast box)
          __builtin_amdgcn_sched_barrier(0);
          mfma16(a0, b0, 1);
          __builtin_amdgcn_sched_barrier(0);
          mfma16(a1, b1, 0);
          __builtin_amdgcn_sched_barrier(0);
          unsigned long long ta_ = 0, tm_ = 0;
          if constexpr (TR && HN) {
            ta_ = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            tm_ = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
          if constexpr (HN) {
            gemm_store_tile16(lds[cur ^ 1][0], tid, ra);
            gemm_store_tile16(lds[cur ^ 1][1], tid, rb);
          }
          if constexpr (HN) {
            __syncthreads();
            if constexpr (TR) {
              const unsigned long long tb_ = __builtin_amdgcn_s_memtime();
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              const unsigned long long dv = tm_ - ta_, db = tb_ - tm_;
              tr_vm += dv; tr_bar += db;
              const unsigned long long mv = tr_mx >> 32, mb = tr_mx & 0xffffffffull;
              tr_mx = ((dv > mv ? dv : mv) << 32) | (db > mb ? db : mb);
            }
            read_half(a0, b0, cur ^ 1, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          mfma16(a1, b1, 1);
          __builtin_amdgcn_sched_barrier(0);
        };
        for (int kt = 0; kt + 1 < nk32; ++kt) step(kt, std::true_type{});
        // persistent caller: the index of the workgroup's NEXT tile is requested here -- the counter's round trip (device
        // scope: microseconds) passes under the last 64 MFMAs and the stores.  Requested at the top of a tile it sat in
        // front of the tile's own loads (memory operations of a wavefront return in order: every tile started with the
        // round trip exposed, 11 k cycles between two tiles of a slot -- no better than a workgroup launch).
        unsigned nxt_ = 0;
        if constexpr ((FLAGS & 2) != 0 && FULL) {
          if (tid == 0) nxt_ = atomicAdd(g.queue, 1u);
        }
        step(nk32 - 1, std::false_type{});
        if constexpr ((FLAGS & 2) != 0 && FULL) {
          if (tid == 0) {
            s_next[0] = nxt_;
            s_next[1] = 1u;  // "fetched"
          }
        }
      }
    } else
    for (int64_t kt = 0; kt < nk; ++kt) {
      const int cur = (int)(kt & 1);
      if (kt + 1 < nk && !(dbg & 2)) {
        gemm_load_tile<FULL>(g.A, g.lda, row0, g.M, kofs(kt + 1), g.K, tid, ra);
        gemm_load_tile<FULL>(g.B, g.ldb, col0, g.N, kofs(kt + 1), g.K, tid, rb);
      }
      const d2* Ad = reinterpret_cast<const d2*>(lds[cur][0]);
      const d2* Bd = reinterpret_cast<const d2*>(lds[cur][1]);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kq = 2 * gq + h;
        const int cx = c16 ^ kq;
        const d2* Ap = Ad + kq * 128 + wm * 64 + cx;
        const d2* Bp = Bd + kq * 128 + wn * 64 + cx;
        d2 a[4], bb[4];
        if (dbg & 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) a[i] = bb[i] = (d2){(double)(lane + i + h), (double)(lane - i)};
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) a[i] = Ap[16 * i];
#pragma unroll
          for (int j = 0; j < 4; ++j) bb[j] = Bp[16 * j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, bb[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, bb[j].y, acc[i][j], 0, 0, 0);
        if (h == (CKS >= 8 ? 1 : 0) && kt + 1 < nk && !(dbg & 2)) {
          gemm_store_tile16(lds[cur ^ 1][0], tid, ra);
          gemm_store_tile16(lds[cur ^ 1][1], tid, rb);
        }
      }
      if (!(dbg & 8)) __syncthreads();
    }
  } else {
    for (int64_t kt = 0; kt < nk; ++kt) {
      const int cur = (int)(kt & 1);
      if (kt + 1 < nk && !(dbg & 2)) {
        gemm_load_tile<FULL>(g.A, g.lda, row0, g.M, kofs(kt + 1), g.K, tid, ra);
        gemm_load_tile<FULL>(g.B, g.ldb, col0, g.N, kofs(kt + 1), g.K, tid, rb);
      }
      const double* As = lds[cur][0] + (wm * 64 + li) * GPITCH + lk;
      const double* Bs = lds[cur][1] + (wn * 64 + li) * GPITCH + lk;
  #pragma unroll
      for (int ks = 0; ks < GBK; ks += 4) {
        double a[4], bb[4];
        if (dbg & 4) {
  #pragma unroll
          for (int i = 0; i < 4; ++i) a[i] = bb[i] = (double)(lane + i + ks);
        } else {
  #pragma unroll
          for (int i = 0; i < 4; ++i) a[i] = -As[i * 16 * GPITCH + ks];
  #pragma unroll
          for (int j = 0; j < 4; ++j) bb[j] = Bs[j * 16 * GPITCH + ks];
        }
  #pragma unroll
        for (int i = 0; i < 4; ++i)
  #pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
        if (ks == CKS && kt + 1 < nk && !(dbg & 2)) {
          // the next tile's global loads were issued ~32 MFMAs ago: write them to the other LDS
          // buffer now so that the stores drain under the remaining MFMAs of this tile
          gemm_store_tile(lds[cur ^ 1][0], tid, ra);
          gemm_store_tile(lds[cur ^ 1][1], tid, rb);
        }
      }
      if (!(dbg & 8)) __syncthreads();
    }
  }
  if (dbg & 1) {
    if (acc[0][0][0] == 1.2345e-300) g.C[0] = 0.0;  // keep the accumulators alive
    return;
  }

  // ---- epilogue: C -= acc.  f64 MFMA C/D layout: col = lane & 15, row = (lane >> 4) + 4 r.
  // All loads of a 64x16 column strip are issued before the first use (no serialised round trips).
  if constexpr (TR) tr_t3 = __builtin_amdgcn_s_memtime();
  auto trace_out = [&]() {
    if constexpr (TR) {
      if (g.trace != nullptr && (threadIdx.x & 63) == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t4 = __builtin_amdgcn_s_memtime();
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        unsigned long long* o = g.trace + (trace_slot * 4 + (threadIdx.x >> 6)) * 8;
        o[0] = tr_t0; o[1] = tr_t1; o[2] = tr_t3; o[3] = t4; o[4] = tr_vm; o[5] = tr_bar; o[6] = tr_mx;
        o[7] = ((unsigned long long)xcc << 32) | hw;
      }
    }
  };
  if (CIN) {
    // recompute the store addresses from laundered copies: the compiler otherwise keeps the prologue's 32 load addresses
    // alive across the main loop (spilled to scratch)
    double* Ct2g = Ct;
    unsigned coff2 = coff;
    asm volatile("" : "+v"(Ct2g), "+v"(coff2));
    // (the laundered pointer is generic: back to the global address space, or the 64 stores are FLAT stores)
    __attribute__((address_space(1))) double* Ct2 = (__attribute__((address_space(1))) double*)Ct2g;
    if (g.nt_c) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(SGN * acc[i][j][r], (Ct2 + (int64_t)(i * 16 + 4 * r) * g.ldc) + coff2 + j * 16);
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) (Ct2 + (int64_t)(i * 16 + 4 * r) * g.ldc)[coff2 + j * 16] = SGN * acc[i][j][r];
    trace_out();
    return;
  }
  if constexpr (OW) {  // edge tile of an overwriting product: guarded stores, nothing read
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t gc = col0 + wn * 64 + j * 16 + li;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t gr = row0 + wm * 64 + i * 16 + lk + 4 * r;
          if (gc < g.N && gr < g.M) g.C[gr * g.ldc + gc] = SGN * acc[i][j][r];
        }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double cv[4][4];
    if (FULL) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[i][r] = Cw[(i * 16 + 4 * r) * g.ldc + j * 16];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cw[(i * 16 + 4 * r) * g.ldc + j * 16] = cv[i][r] + SGN * acc[i][j][r];
    } else {
      const int64_t gc = col0 + wn * 64 + j * 16 + li;
      const bool cok = gc < g.N;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t gr = row0 + wm * 64 + i * 16 + lk + 4 * r;
          const bool ok = cok && gr < g.M;
          const int64_t cr = gr < g.M ? gr : g.M - 1, cc = cok ? gc : g.N - 1;
          cv[i][r] = g.C[cr * g.ldc + cc];
          (void)ok;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t gr = row0 + wm * 64 + i * 16 + lk + 4 * r;
          if (cok && gr < g.M) g.C[gr * g.ldc + gc] = cv[i][r] + SGN * acc[i][j][r];
        }
    }
  }
}

// block index -> tile (XCD-aware 8x8 super tiles: block b runs on XCD b % 8) and the tile's GEMM
template <bool ABL, bool PIPE = false, bool CACC = true, int CKS = GEMM_COMMIT_KS, int FLAGS = 0>
__device__ __forceinline__ void gemm_block(const GemmArgs& g, double (*lds)[2][GT * GPITCH], int64_t b, unsigned* s_next = nullptr) {
  const int64_t b_launch = b;
  if (b < g.tiles2) {  // second problem (workgroup-uniform branch)
    const int tn2 = (int)((g.N2 + GT - 1) / GT);
    const int64_t ti = b / tn2, tj = b - ti * tn2;
    if (ti < tn2 && tj > ti) return;
    GemmArgs h = g;
    h.A = g.A2; h.B = g.B2; h.C = g.C2; h.M = g.M2; h.N = g.N2; h.K = g.K2;
    h.aligned = ((reinterpret_cast<uintptr_t>(g.A2) | reinterpret_cast<uintptr_t>(g.B2)) & 15) == 0 && g.aligned;
    const int64_t row0 = ti * GT, col0 = tj * GT;
    const bool full = (row0 + GT <= h.M) && (col0 + GT <= h.N) && ((h.K & (GBK - 1)) == 0) && h.aligned;
    if (full)
      gemm_tile_body<true, ABL, PIPE, CACC, CKS, (FLAGS & 2)>(h, lds, row0, col0, 0, s_next);
    else
      gemm_tile_body<false, ABL, PIPE, CACC, CKS, (FLAGS & 2)>(h, lds, row0, col0);
    if (g.ready && ti < g.ready_rows && tj < g.ready_rows) {
      __threadfence();
      __syncthreads();
      if (threadIdx.x == 0) atomicAdd(g.ready, 1);
    }
    return;
  }
  b -= g.tiles2;
  const int64_t xcd = b & 7, loc = b >> 3;
  const int64_t s = g.s_begin + (loc >> 6) * 8 + xcd;
  const int within = (int)(loc & 63);
  if (s >= g.n_super) return;
  int64_t SI, SJ;
  if (g.lower) {
    int64_t sr = s, shift = 0;
    const int64_t sm = (g.tiles_m + 7) / 8;
    if (g.col0_first) {  // column SJ = 0 (all SI) first, then the lower triangle of the remaining sm - 1 super rows
      if (s < sm) { SI = s; SJ = 0; sr = -1; }
      else { sr = s - sm; shift = 1; }
    }
    if (sr >= 0) {
      // sr = SI (SI+1)/2 + SJ
      SI = (int64_t)((sqrt(8.0 * (double)sr + 1.0) - 1.0) * 0.5);
      while (SI * (SI + 1) / 2 > sr) --SI;
      while ((SI + 1) * (SI + 2) / 2 <= sr) ++SI;
      SJ = sr - SI * (SI + 1) / 2;
      SI += shift; SJ += shift;
    }
  } else {
    SI = s / g.super_n;
    SJ = s - SI * g.super_n;
  }
  const int64_t ti = SI * 8 + (within >> 3), tj = SJ * 8 + (within & 7);
  if (ti >= g.tiles_m || tj >= g.tiles_n) return;
  if (g.lower && tj > ti) return;
  const int64_t row0 = ti * GT, col0 = tj * GT;
  if (g.cyc_W > 0 && row0 + GT <= g.cyc_block_rows) {  // tiles that reach into the carried rhs row take all columns
    const int64_t gb = (g.cyc_lb0 + row0 / g.cyc_nb) * g.cyc_W + g.cyc_rank;  // (tiles never straddle blocks: nb % GT == 0)
    const int64_t grow_last = gb * g.cyc_nb + (row0 + GT - 1) % g.cyc_nb;
    if (g.cyc_col0 + col0 > grow_last) return;
  }
  const bool full = (row0 + GT <= g.M) && (col0 + GT <= g.N) && ((g.K & (GBK - 1)) == 0) && g.aligned;
  if (full)
    gemm_tile_body<true, ABL, PIPE, CACC, CKS, FLAGS>(g, lds, row0, col0, b_launch, s_next);
  else
    gemm_tile_body<false, ABL, PIPE, CACC, CKS, (FLAGS & 6)>(g, lds, row0, col0);
  if (g.ready && g.tiles2 == 0 && ti < g.ready_rows && tj < g.ready_rows) {  // publish the tile (release: every thread's stores, then one count)
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(g.ready, 1);
  }
}

template <bool ABL>
__global__ void __launch_bounds__(256, 2) gemm_nt_sub_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) double lds[2][2][GT * GPITCH];
  // production instantiation: the 16-byte LDS layout with read-ahead of the fused kernel (gemm.lds16 = 3); the ablation
  // instantiation keeps the 8-byte layout its masks were written for
  gemm_block<ABL, !ABL, true, ABL ? GEMM_COMMIT_KS : 5>(g, lds, blockIdx.x);
}


// C = -A B^T: the production loop with the accumulators starting at zero and the epilogue's stores only -- for callers whose C
// is a fresh work buffer (the prediction contractions cleared 1.8 GB per mat-vec at configs[3] only to have it read back)
__global__ void __launch_bounds__(256, 2) gemm_nt_neg_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) double lds[2][2][GT * GPITCH];
  gemm_block<false, true, true, 5, 4>(g, lds, blockIdx.x);
}

// f0, f1: the launch covers the super tiles [f0 * n_super, f1 * n_super) (whole update: 0, 1);
// timed: bracket with the per-kernel timers (only launches on the timing stream)
struct DiagJob {
  double* A = nullptr;  // top-left of the next panel's diagonal block
  int nbw = 0;
  int64_t off = 0;
  // merged launch: the block's own tiles are part of this launch (GemmArgs::col0_first); they count into *ready
  int* ready = nullptr;
  int ready_target = 0;
  bool col0_first = false;
  // second problem of the launch (GemmArgs::A2 ...)
  const double* A2 = nullptr;
  const double* B2 = nullptr;
  double* C2 = nullptr;
  int64_t M2 = 0, N2 = 0, K2 = 0;
};

static bool gemm_use_n64(gdml_ctx* ctx) { return ctx_opt_i(ctx, "gemm.n64", 0) != 0; }

static int launch_gemm_nt_sub_part(gdml_ctx* ctx, hipStream_t st, const double* A, int64_t lda,
                                   const double* B, int64_t ldb, double* C, int64_t ldc, int64_t M,
                                   int64_t N, int64_t K, int lower, double f0, double f1, bool timed,
                                   const DiagJob* diag = nullptr, const CyclicLower* cyc = nullptr, int tile_n64 = -1,
                                   bool overwrite = false) {
  if (M <= 0 || N <= 0 || K <= 0) return GDML_OK;
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.lower = lower;
  g.diagA = nullptr; g.diag_nbw = 0; g.diag_off = 0; g.diag_info = nullptr;
  g.cyc_W = 0; g.cyc_rank = 0; g.cyc_lb0 = 0; g.cyc_nb = 0; g.cyc_col0 = 0; g.cyc_block_rows = 0;
  g.col0_first = (diag && diag->col0_first) ? 1 : 0;
  g.ready = nullptr; g.ready_target = 0; g.ready_rows = 0;
  g.A2 = g.B2 = nullptr; g.C2 = nullptr; g.M2 = g.N2 = g.K2 = 0; g.tiles2 = 0;
  g.trace = nullptr; g.queue = nullptr; g.n_items = 0;
  if (cyc) { g.cyc_W = cyc->W; g.cyc_rank = cyc->rank; g.cyc_lb0 = cyc->lb0; g.cyc_nb = cyc->nb; g.cyc_col0 = cyc->col0; g.cyc_block_rows = cyc->block_rows; }
  g.dbg = ctx_opt_i(ctx, "gemm.debug", 0);
  g.nt_c = ctx_opt_i(ctx, "gemm.nt_c", 0);
  g.aligned = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0 &&
              (lda % 2 == 0) && (ldb % 2 == 0);
  // gemm.n64: 128 x 64 tiles, three workgroups per CU (gemm_nt_sub_n64_kernel); super tiles stay 1024 x 1024
  const bool n64 = (tile_n64 >= 0 ? tile_n64 != 0 : gemm_use_n64(ctx)) && !cyc && g.dbg == 0 && !overwrite;
  const int TN = n64 ? G6N : GT, super_cols = n64 ? 16 : 8;
  g.tiles_m = (int)((M + GT - 1) / GT);
  g.tiles_n = (int)((N + TN - 1) / TN);
  int64_t sm = (g.tiles_m + 7) / 8, sn = (g.tiles_n + super_cols - 1) / super_cols;
  g.super_n = (int)sn;
  const int64_t n_super_all = lower ? sm * (sm + 1) / 2 : sm * sn;
  g.s_begin = (int64_t)(f0 * (double)n_super_all);
  g.n_super = (f1 >= 1.0) ? n_super_all : (int64_t)(f1 * (double)n_super_all);
  const bool has_diag = diag && diag->A;
  if (g.n_super <= g.s_begin) {
    if (!has_diag) return GDML_OK;
    g.n_super = g.s_begin;  // empty tile range: the launch still carries the diagonal-block workgroup
  }
  int64_t groups = (g.n_super - g.s_begin + 7) / 8;  // each group of 8 super tiles = 8 XCDs x 64 blocks
  int64_t blocks = groups * (n64 ? 1024 : 512);
  const int slot = (timed && st == (ctx->kt_stream ? ctx->kt_stream : ctx->stream)) ? ktime_begin(ctx) : -1;
  if (has_diag) {
    g.diagA = diag->A; g.diag_nbw = diag->nbw; g.diag_off = diag->off; g.diag_info = ctx->d_info;
    if (diag->ready) { g.ready = diag->ready; g.ready_target = diag->ready_target; g.ready_rows = diag->nbw * 64 / GT; }
    if (diag->A2 && diag->M2 > 0) {
      g.A2 = diag->A2; g.B2 = diag->B2; g.C2 = diag->C2; g.M2 = diag->M2; g.N2 = diag->N2; g.K2 = diag->K2;
      g.tiles2 = (int)(ceil_div(g.M2, GT) * ceil_div(g.N2, TN));
    }
    const dim3 grid((unsigned)(blocks + 1 + g.tiles2));
    if (n64) hipLaunchKernelGGL(gemm_nt_sub_n64_kernel, grid, dim3(256), 0, st, g);
    else {
    // gemm.trace = k > 0: the k-th fused launch since the option was set runs the traced instantiation and leaves
    // gemm_trace.bin (header: blocks, tiles2, tiles_m, s_begin, n_super, col0_first; then 4 x 8 words per workgroup)
    const int trace_k = ctx_opt_i(ctx, "gemm.trace", 0);
    if (trace_k > 0 && ++ctx->gemm_trace_seen == trace_k) {
      const size_t words = (size_t)(blocks + g.tiles2) * 32;
      unsigned long long* d_tr = nullptr;
      GDML_TRY(ctx_alloc(ctx, (void**)&d_tr, (int64_t)(words * 8)));
      HIP_CHECK(ctx, hipMemsetAsync(d_tr, 0, words * 8, st));
      g.trace = d_tr;
      const int64_t resident_t = 2 * (int64_t)ctx->num_cus;
      if (ctx_opt_i(ctx, "gemm.persist", 0) != 0 && blocks + g.tiles2 >= 4 * resident_t && ctx->gemm_queue != nullptr) {
        if (ctx->gemm_queue_next >= ctx->gemm_queue_sets) {
          HIP_CHECK(ctx, hipMemsetAsync(ctx->gemm_queue, 0, (size_t)ctx->gemm_queue_sets * 512, st));
          ctx->gemm_queue_next = 0;
        }
        g.queue = ctx->gemm_queue + (size_t)(ctx->gemm_queue_next++) * 128;
        g.n_items = blocks + g.tiles2;
        hipLaunchKernelGGL((gemm_nt_sub_persist_kernel<5, 3>), dim3((unsigned)resident_t), dim3(256), 0, st, g);
      } else
      hipLaunchKernelGGL(gemm_nt_sub_diag_trace_kernel, grid, dim3(256), 0, st, g);
      std::vector<unsigned long long> h(words + 8);
      HIP_CHECK(ctx, hipMemcpyAsync(h.data() + 8, d_tr, words * 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(ctx, hipStreamSynchronize(st));
      h[0] = (unsigned long long)blocks; h[1] = (unsigned long long)g.tiles2; h[2] = (unsigned long long)g.tiles_m;
      h[3] = (unsigned long long)g.s_begin; h[4] = (unsigned long long)g.n_super; h[5] = (unsigned long long)g.col0_first;
      h[6] = (unsigned long long)K; h[7] = (unsigned long long)M;
      if (FILE* f = fopen("gemm_trace.bin", "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
      GDML_TRY(ctx_free(ctx, d_tr));
      ktime_end(ctx, slot, "gemm_nt_sub_diag_traced", 0.0);
      ctx->launch_counter++;
      return GDML_OK;
    }
    // gemm.persist (default 0: measured 1.5 % slower, profiles/r06_gemm_variants.txt): resident workgroups pulling tiles from per-XCD counters, for launches of at least four rounds
    const int64_t resident = 2 * (int64_t)ctx->num_cus;
    if (ctx_opt_i(ctx, "gemm.persist", 0) != 0 && blocks + g.tiles2 >= 4 * resident && ctx->gemm_queue != nullptr) {
      if (ctx->gemm_queue_next >= ctx->gemm_queue_sets) {  // (chol_factor_device zeroes the ring; a caller that does not gets a fresh one here)
        HIP_CHECK(ctx, hipMemsetAsync(ctx->gemm_queue, 0, (size_t)ctx->gemm_queue_sets * 512, st));
        ctx->gemm_queue_next = 0;
      }
      g.queue = ctx->gemm_queue + (size_t)(ctx->gemm_queue_next++) * 128;
      g.n_items = blocks + g.tiles2;
      hipLaunchKernelGGL((gemm_nt_sub_persist_kernel<5>), dim3((unsigned)resident), dim3(256), 0, st, g);
    } else
    // gemm.lds16 = 2: the loop with its last k-tile inside (A/B reference of the peeled production loop)
    if (ctx_opt_i(ctx, "gemm.lds16", 3) == 2) hipLaunchKernelGGL((gemm_nt_sub_diag_kernel<true, true, 4>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_nt_sub_diag_kernel<true, true, 5>), grid, dim3(256), 0, st, g);
    }
  } else if (overwrite)
    hipLaunchKernelGGL(gemm_nt_neg_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g);
  else if (n64)
    hipLaunchKernelGGL(gemm_nt_sub_n64_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g);
  else if (g.dbg)
    hipLaunchKernelGGL(gemm_nt_sub_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, g);
  else
    hipLaunchKernelGGL(gemm_nt_sub_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, g);
  const double part = (double)(g.n_super - g.s_begin) / (double)n_super_all;  // share of the tile list in this launch
  // the fused launches (trailing update + diagonal-block workgroup) are the dominant kernel: timed under their own name
  ktime_end(ctx, slot, has_diag ? "gemm_nt_sub_diag" : "gemm_nt_sub",
            part * (lower ? (double)M * (double)(M + 1) * (double)K : 2.0 * (double)M * (double)N * (double)K) +
                2.0 * (double)g.M2 * (double)g.N2 * (double)g.K2);
  ctx->launch_counter++;
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}

int launch_gemm_nt_sub(gdml_ctx* ctx, hipStream_t st, const double* A, int64_t lda,
                              const double* B, int64_t ldb, double* C, int64_t ldc, int64_t M,
                              int64_t N, int64_t K, int lower) {
  return launch_gemm_nt_sub_part(ctx, st, A, lda, B, ldb, C, ldc, M, N, K, lower, 0.0, 1.0, true);
}

// Plain product whose tile shape is chosen by how well the tile count fills the chip: 128 x 128 tiles run two workgroups per CU
// (512 slots), 128 x 64 tiles three (768 slots) at 0.93 of the wide tile's rate inside the k loop (profiles/r06_gemm_n64_ab.txt).
// A launch of 1128 wide tiles (the configs[4] mat-vec: 6016 x 3072) is 2.2 rounds of 512 -- 0.73 full; as 2256 narrow tiles it
// is 2.94 rounds of 768 -- 0.98 full.  Only for callers that ask (the prediction contractions), and OFF by default (option
// gemm.fill_tiles): measured on exactly that launch the narrow tiles lose -- mat-vec 9.37 ms against 8.43 ms with wide tiles
// (profiles/r06_matvec_probe.txt): the wide launch's third round is short, not a full round long.
int launch_gemm_nt_neg(gdml_ctx* ctx, hipStream_t st, const double* A, int64_t lda, const double* B, int64_t ldb,
                       double* C, int64_t ldc, int64_t M, int64_t N, int64_t K) {  // C = -A B^T, C not read
  return launch_gemm_nt_sub_part(ctx, st, A, lda, B, ldb, C, ldc, M, N, K, 0, 0.0, 1.0, true, nullptr, nullptr, 0, true);
}

int launch_gemm_nt_sub_fill(gdml_ctx* ctx, hipStream_t st, const double* A, int64_t lda, const double* B, int64_t ldb,
                            double* C, int64_t ldc, int64_t M, int64_t N, int64_t K) {
  const int64_t slots = 2 * (int64_t)ctx->num_cus;
  const int64_t t_w = (int64_t)ceil_div(M, GT) * ceil_div(N, GT), t_n = (int64_t)ceil_div(M, GT) * ceil_div(N, G6N);
  const double fill_w = (double)t_w / (double)((int64_t)ceil_div(t_w, slots) * slots);
  const double fill_n = 0.93 * (double)t_n / (double)((int64_t)ceil_div(t_n, slots * 3 / 2) * (slots * 3 / 2));
  const int pick = ctx_opt_i(ctx, "gemm.fill_tiles", 0) != 0 && fill_n > fill_w ? 1 : 0;
  return launch_gemm_nt_sub_part(ctx, st, A, lda, B, ldb, C, ldc, M, N, K, 0, 0.0, 1.0, true, nullptr, nullptr, pick);
}

int launch_gemm_nt_sub_cyclic(gdml_ctx* ctx, hipStream_t st, const double* A, int64_t lda, const double* B, int64_t ldb,
                              double* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const CyclicLower& cyc) {
  return launch_gemm_nt_sub_part(ctx, st, A, lda, B, ldb, C, ldc, M, N, K, 0, 0.0, 1.0, true, nullptr, &cyc);
}

// ------------------------------------------------------------------------------------------
// potrf on a w x w (w <= 64) diagonal block, in place (lower).  ONE wavefront: lane r keeps row r of
// the block in registers (fully unrolled right-looking elimination, no barriers); per step the
// scaled column is exchanged through a 64-entry LDS vector (broadcast reads).  Entries above the
// diagonal are scratch.  info (device int): first failing pivot, global 1-based (LAPACK dpotrf).
// ------------------------------------------------------------------------------------------
// One wavefront factors the 64 x 64 tile in T (pitch 65, identity padding outside w x w) in registers: lane = row.
// Returns 0 or the 1-based index of the first non-positive pivot.  Right-looking elimination blocked by 8 columns (round 4;
// bit-identical to the rounds 1-3 form, which sent every scaled column through LDS and waited for the round trip before any
// entry of the next column -- the next pivot included -- could be updated): the columns of the current 8-block are
// updated from lane broadcasts (v_readlane of the scaled column: no memory on the pivot chain), and the columns right of
// the block get the block's 8 rank-1 updates at once from an LDS copy of the 8 finished columns (cb: 64 rows x 8,
// broadcast 16-byte reads).  1/sqrt(d): hardware estimate + two Newton steps (full double precision), then one multiply
// per lane instead of a correctly rounded sqrt and a division on the 64-step critical path.
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int potrf64_wave(double* T, double* cb /* 64 x 8 */, int lane) {
  double row[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) row[c] = T[lane * 65 + c];
  int fail = 0;
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int j = 8 * jb + jj;
      const double d = readlane_f64(row[j], j);
      if (!(d > 0.0) && fail == 0) fail = j + 1;
      double ri = __builtin_amdgcn_rsq(d);
      ri = ri * (1.5 - 0.5 * d * ri * ri);
      ri = ri * (1.5 - 0.5 * d * ri * ri);
      const double lr = row[j] * ri;  // lane r: L[r][j] (lane j: d * ri = the pivot's square root; lanes r < j: scratch)
      row[j] = lr;
#pragma unroll
      for (int c = j + 1; c < 8 * jb + 8; ++c) row[c] -= lr * readlane_f64(lr, c);
    }
    if (jb < 7) {
      d2* wr = reinterpret_cast<d2*>(cb + lane * 8);
#pragma unroll
      for (int k = 0; k < 4; ++k) wr[k] = (d2){row[8 * jb + 2 * k], row[8 * jb + 2 * k + 1]};
      __builtin_amdgcn_s_waitcnt(0xc07f);
      // 4 columns at a time, the 8 rank-1 updates applied across them (k outer): four independent multiply-add chains
      // (column by column the compiler emits 8 dependent multiply-adds behind each batch of reads and a lone wavefront
      // pays the full instruction latency 1792 times; 8 columns at a time need > 256 registers: copies through AGPRs)
#pragma unroll
      for (int c0 = 8 * jb + 8; c0 < 64; c0 += 4) {
        d2 l[4][4];
        double v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const d2* rd = reinterpret_cast<const d2*>(cb + (c0 + g) * 8);
#pragma unroll
          for (int q = 0; q < 4; ++q) l[g][q] = rd[q];
          v[g] = row[c0 + g];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int g = 0; g < 4; ++g) v[g] -= row[8 * jb + k] * ((k & 1) ? l[g][k >> 1].y : l[g][k >> 1].x);
#pragma unroll
        for (int g = 0; g < 4; ++g) row[c0 + g] = v[g];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);  // the reads of this block are done before the next block overwrites cb
    }
    __builtin_amdgcn_sched_barrier(0);  // keeps the scheduler from hoisting later blocks' work (and its registers) up here
  }
#pragma unroll
  for (int c = 0; c < 64; ++c) T[lane * 65 + c] = row[c];
  __builtin_amdgcn_s_waitcnt(0xc07f);
  return fail;
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): compile-time expansion of a loop body
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// DPP moves of a double (both halves): quad broadcast of lane q, row shifts
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_bcast(double v, int q) {
  switch (q) {
    case 0: return dpp_mov<0x00>(v);
    case 1: return dpp_mov<0x55>(v);
    case 2: return dpp_mov<0xaa>(v);
    default: return dpp_mov<0xff>(v);
  }
}

// Exact substitution x <- x L^-T of one 64-column block for a row held by 8 threads (thread sq owns the columns k = sq mod 8
// in t[0..7]); Lt = L^T with pitch lp and rinv = 1 / diag(L) in LDS.  What this costs is its 64-step dependent chain, so
// the solved entry travels by DPP (quad broadcast, then a 4-lane row shift into the row's other quad: VALU latency instead
// of an LDS permute round trip), the L values of column c + 1 are fetched from LDS while column c is applied, and the
// column steps are expanded at compile time (a 64-trip loop of this size is only partially unrolled even under
// #pragma unroll, which turns t[c >> 3] into a dynamically indexed register array).
template <int LP, bool TRL = false>
__device__ __forceinline__ void subst64_row8(double (&t)[8], const double* __restrict__ Lt, const double* __restrict__ rinv,
                                             int sq) {
  double ln[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ln[i] = TRL ? Lt[(sq + 8 * i) * LP] : Lt[sq + 8 * i];
  double rn = rinv[0];
  auto column = [&](auto cc) __attribute__((always_inline)) {
    constexpr int c = decltype(cc)::value;
    constexpr int qc = c & 7, ic = c >> 3;
    double lc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) lc[i] = ln[i];
    const double rc = rn;
    if constexpr (c + 1 < 64) {
      rn = rinv[c + 1];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i >= ((c + 1) >> 3)) ln[i] = TRL ? Lt[(sq + 8 * i) * LP + (c + 1)] : Lt[(c + 1) * LP + sq + 8 * i];
    }
    const double xq = quad_bcast(t[ic] * rc, qc & 3);                      // lane (qc & 3) of the own quad
    const double xo = (qc < 4) ? dpp_mov<0x114>(xq) : dpp_mov<0x104>(xq);  // the other quad's: row_shr:4 / row_shl:4
    const double x = ((sq >> 2) == (qc >> 2)) ? xq : xo;
    if (sq == qc) t[ic] = x;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i > ic || (i == ic && sq > qc)) t[i] -= x * lc[i];
    if constexpr ((c & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // bounds the scheduler's hoisting of later columns' reads
  };
  static_for<64>(column);
}

// ------------------------------------------------------------------------------------------
// Diagonal block of a panel, factored by ONE workgroup (256 threads) inside the trailing-update launch.
// The 64-wide step chain (potrf64, rows below by substitution, rank-64 update) of the nb x nb block is a
// dependent sequence of tiny kernels that the hardware does not overlap with a pending GEMM grid from another
// stream (profiles/r02_sched_probe.txt) and that ran ~650 us per panel exposed.  As workgroup 0 of the SYRK launch
// of the PREVIOUS panel -- which never touches these columns -- it is dispatched first and its ~0.5 ms on one CU
// disappear behind the other workgroups' tiles.  Phases talk through global memory (the block is L2 resident);
// all of them are executed by the same workgroup, so __syncthreads() orders them (waves of a workgroup share the
// CU's vector L1).
//   D: top-left of the block (row-major, ld), nbw = nb / 64.  lds: >= 67 KB (aliased onto the GEMM tile buffers).
// ------------------------------------------------------------------------------------------

// 64 x 64 Cholesky (lower, in place in the LDS tile T, pitch 65) by a whole workgroup of 256 threads with 16
// doubles of state per thread (the one-wavefront version keeps a row of 64 per lane, too many registers next to
// the GEMM role's budget): thread (row = tid / 4, q = tid % 4) holds the columns k = q (mod 4) of its row.
// Right-looking; per step the current column goes through a double-buffered LDS vector: one barrier per step.
// Returns 0 or the 1-based index of the first non-positive pivot (same value in every thread).
__device__ __forceinline__ int potrf64_wg(double* T, double* colbuf /* 2 x 64 */, int tid) {
  const int row = tid >> 2, q = tid & 3;
  double a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = T[row * 65 + q + 4 * i];
  int fail = 0;
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    const int qj = j & 3, ij = j >> 2;
    double* col = colbuf + (j & 1) * 64;
    if (q == qj) col[row] = a[ij];
    __syncthreads();
    const double d = col[j];
    if (!(d > 0.0) && fail == 0) fail = j + 1;
    double ri = __builtin_amdgcn_rsq(d);
    ri = ri * (1.5 - 0.5 * d * ri * ri);
    ri = ri * (1.5 - 0.5 * d * ri * ri);
    const double lr = col[row] * ri;  // L[row][j] (meaningful for row > j)
    if (q == qj) a[ij] = (row == j) ? d * ri : lr;
#pragma unroll
    for (int i = ij; i < 16; ++i) {
      const int c = q + 4 * i;  // columns right of j only
      const double lc = col[c < 64 ? c : 63] * ri;
      if (i > ij || q > qj) a[i] -= lr * lc;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) T[row * 65 + q + 4 * i] = a[i];
  return fail;
}

// SMALL: no transposed copy of L_jj (the substitution reads the factored block in T column-wise: pitch 65 is conflict free both
// ways) -- 35 KB of LDS instead of 68, for the 128 x 64-tile kernel that runs three workgroups per CU on 48 KB each
template <bool SMALL = false>
__device__ __forceinline__ void diag_block_role(double* __restrict__ D, int64_t ld, int nbw, int64_t global_off,
                                             int* __restrict__ info, double* lds) {
  double* T = lds;                   // 64 x 65 (potrf); afterwards reused as S: 32 x 66 row block of the substitution
  double* Lt = lds + 64 * 65;        // L_jj^T, pitch 65
  double* rinv = SMALL ? lds + 64 * 65 : Lt + 64 * 65;  // 64
  double* col = rinv + 64;           // 2 x 64
  double* S = T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int srow = tid >> 3, sq = tid & 7;
  const int nb = 64 * nbw;
  for (int jj = 0; jj < nbw; ++jj) {
    const int c0 = 64 * jj;
    double* Ad = D + (int64_t)c0 * ld + c0;
    for (int r = wave; r < 64; r += 4) T[r * 65 + lane] = Ad[(int64_t)r * ld + lane];
    __syncthreads();
    {
      const int fail = potrf64_wg(T, col, tid);
      if (fail != 0 && tid == 0) atomicCAS(info, 0, (int)(global_off + c0 + fail));
    }
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 256) {
      const int r = e >> 6, c = e & 63;
      const double v = (c <= r) ? T[r * 65 + c] : 0.0;
      if constexpr (!SMALL) Lt[c * 65 + r] = v;
      if (c <= r) Ad[(int64_t)r * ld + c] = v;
      if (c == r) rinv[r] = 1.0 / v;
    }
    __syncthreads();
    // rows below inside the block: x <- x L_jj^-T by exact substitution, 32 rows per pass, 8 threads per row
    // (thread q of a row holds its columns k = q mod 8; same scheme as panel_trsm_kernel)
    const int mrows = nb - c0 - 64;
    for (int base = 0; base < mrows; base += 32) {
      double* xr = D + (int64_t)(c0 + 64 + base + srow) * ld + c0;
      double t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = xr[sq + 8 * i];
      if constexpr (SMALL) subst64_row8<65, true>(t, T, rinv, sq);
      else subst64_row8<65>(t, Lt, rinv, sq);
#pragma unroll
      for (int i = 0; i < 8; ++i) xr[sq + 8 * i] = t[i];
    }
    __syncthreads();
    // rank-64 update of the rest of the block, C -= X_r X_c^T with X = columns [c0, c0+64): lower 32 x 32 macro tiles
    // (2 x 2 MFMA tiles, each operand run loaded once for two tiles), one macro tile per wavefront and turn
    const int nt = mrows / 32;
    const int ntiles = nt * (nt + 1) / 2;
    for (int t = wave; t < ntiles; t += 4) {
      int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
      while (ti * (ti + 1) / 2 > t) --ti;
      while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
      const int tj = t - ti * (ti + 1) / 2;
      const int r0 = c0 + 64 + 32 * ti, q0 = c0 + 64 + 32 * tj;
      double* Cp = D + (int64_t)(r0 + lk) * ld + q0 + li;  // C layout of tile (i, j): row = 16 i + lk + 4 r, col = 16 j + li
      d4 acc[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = Cp[(int64_t)(16 * i + 4 * r) * ld + 16 * j];
      const double* Ap = D + (int64_t)(r0 + li) * ld + c0 + 4 * lk;
      const double* Bp = D + (int64_t)(q0 + li) * ld + c0 + 4 * lk;
      d4 a[2][4], bb[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          a[i][ch] = *reinterpret_cast<const d4*>(Ap + (int64_t)(16 * i) * ld + 16 * ch);
          bb[i][ch] = *reinterpret_cast<const d4*>(Bp + (int64_t)(16 * i) * ld + 16 * ch);
        }
#pragma unroll
      for (int ch = 0; ch < 4; ++ch)
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[i][ch][sidx], bb[j][ch][sidx], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) Cp[(int64_t)(16 * i + 4 * r) * ld + 16 * j] = acc[i][j][r];
    }
    __syncthreads();
  }
  (void)S;
}

// Trailing update + (workgroup 0) the next panel's diagonal block.
template <bool PIPE, bool CACC, int CKS>
__global__ void __launch_bounds__(256, 2) gemm_nt_sub_diag_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) double lds[2][2][GT * GPITCH];
  static_assert(sizeof(double) * 2 * 2 * GT * GPITCH >= sizeof(double) * (2 * 64 * 65 + 64 + 128), "LDS of the diagonal role");
  if (blockIdx.x == 0) {
    if (g.ready) {  // the block's tiles are computed by workgroups of this launch (dispatched right behind this one)
      if (threadIdx.x == 0) {
        int spins = 0;  // bounded: a target that can never be reached must not hang the GPU (flag -> GDML_ERR_HIP)
        while (__hip_atomic_load(g.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.ready_target) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1 << 24)) {
            atomicExch(g.ready + 1, 1);
            break;
          }
        }
      }
      __syncthreads();
      __threadfence();
    }
    diag_block_role(g.diagA, g.ldc, g.diag_nbw, g.diag_off, g.diag_info, &lds[0][0][0]);
    return;
  }
  gemm_block<false, PIPE, CACC, CKS>(g, lds, (int64_t)blockIdx.x - 1);
}

// ------------------------------------------------------------------------------------------
// 128 x 64 tiles, THREE workgroups per CU (round 6, option gemm.n64).  The traced 128 x 128 kernel loses its time in the state
// "one of the CU's two workgroups is between tiles, the other runs alone at 0.76 of the pipe" (14 % of the time,
// profiles/r06_gemm_trace.txt); with three narrower workgroups (64 accumulator registers per lane instead of 128: 168 VGPRs,
// 48 KB of LDS each) two of them are still in their k loops while the third turns over.  Price: 1.5 x the operand traffic
// per flop from L2 and a barrier every 32 MFMAs instead of 64.  Same loop as the production kernel's (16-byte LDS layout,
// operand pairs of the next half k-tile read ahead, last k-tile peeled, schedule pinned by sched_barriers); wave (wm, wn)
// owns 64 rows x 32 columns = 4 x 2 MFMA tiles.
// ------------------------------------------------------------------------------------------
#define G6_STAGE ((GT + G6N) * GBK)  // doubles per LDS stage: A tile (128 x 16), then B tile (64 x 16)

template <bool FULL>
__device__ __forceinline__ void gemm_tile_body_n64(const GemmArgs& g, double* __restrict__ lds, int64_t row0, int64_t col0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c16 = lane & 15, gq = lane >> 4;
  const int wu = __builtin_amdgcn_readfirstlane(wave);
  double* const Ct = g.C + (row0 + (wu >> 1) * 64) * g.ldc + col0 + (wu & 1) * 32;
  const unsigned coff = (unsigned)gq * (unsigned)g.ldc + (unsigned)c16;
  d4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (FULL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = -(Ct + (int64_t)(i * 16 + 4 * r) * g.ldc)[coff + j * 16];
      } else {
        acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
      }
    }
  const int nk32 = (int)((g.K + GBK - 1) / GBK);
  unsigned offA[4], offB[2];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const int cidx = tid + 256 * s4;
    offA[s4] = (unsigned)(((int64_t)(cidx >> 3) * g.lda + (cidx & 7) * 2) * 8);
    if (s4 < 2) offB[s4] = (unsigned)(((int64_t)(cidx >> 3) * g.ldb + (cidx & 7) * 2) * 8);
  }
  const double* const Abase = g.A + row0 * g.lda;
  const double* const Bbase = g.B + col0 * g.ldb;
  d2 ra[4], rb[2];
  auto load_ab = [&](int kt_) {
    const int64_t ko = (int64_t)kt_ * GBK;
    if constexpr (FULL) {
      typedef const __attribute__((address_space(1))) char* gcptr;
      typedef const __attribute__((address_space(1))) d2* gd2ptr;
      const uint64_t pa = reinterpret_cast<uint64_t>(Abase + ko), pb = reinterpret_cast<uint64_t>(Bbase + ko);
      const uint32_t alo = __builtin_amdgcn_readfirstlane((uint32_t)pa), ahi = __builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32));
      const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)pb), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32));
      gcptr ba = (gcptr)(((uint64_t)ahi << 32) | alo);
      gcptr bb = (gcptr)(((uint64_t)bhi << 32) | blo);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) ra[s4] = *(gd2ptr)(ba + offA[s4]);
#pragma unroll
      for (int s4 = 0; s4 < 2; ++s4) rb[s4] = *(gd2ptr)(bb + offB[s4]);
    } else {
      // branch-free guarded loads: clamp the address, then zero what is out of range
      auto guarded = [&](const double* __restrict__ G, int64_t ld, int64_t r0, int64_t nrows, int cidx) -> d2 {
        const int64_t gr = r0 + (cidx >> 3), gk = ko + (cidx & 7) * 2;
        const int64_t cr = gr < nrows ? gr : nrows - 1;
        const int64_t ck0 = gk < g.K ? gk : g.K - 1, ck1 = gk + 1 < g.K ? gk + 1 : g.K - 1;
        const double v0 = G[cr * ld + ck0], v1 = G[cr * ld + ck1];
        d2 v;
        v.x = (gr < nrows && gk < g.K) ? v0 : 0.0;
        v.y = (gr < nrows && gk + 1 < g.K) ? v1 : 0.0;
        return v;
      };
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) ra[s4] = guarded(g.A, g.lda, row0, g.M, tid + 256 * s4);
#pragma unroll
      for (int s4 = 0; s4 < 2; ++s4) rb[s4] = guarded(g.B, g.ldb, col0, g.N, tid + 256 * s4);
    }
  };
  // element (row, kq = k / 2) of a tile = the pair (k, k + 1) at d2 index kq * rows + (row ^ kq)  (gemm_store_tile16's layout)
  auto commit = [&](int buf) {
    d2* SA = reinterpret_cast<d2*>(lds + buf * G6_STAGE);
    d2* SB = SA + GT * GBK / 2;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int cidx = tid + 256 * s4;
      const int row = cidx >> 3, kq = cidx & 7;
      SA[kq * GT + (row ^ kq)] = ra[s4];
      if (s4 < 2) SB[kq * G6N + (row ^ kq)] = rb[s4];
    }
  };
  auto read_half = [&](d2 (&a_)[4], d2 (&b_)[2], int buf, int h) {
    const int kq = 2 * gq + h;
    const int cx = c16 ^ kq;
    const d2* Ap = reinterpret_cast<const d2*>(lds + buf * G6_STAGE) + kq * GT + wm * 64 + cx;
    const d2* Bp = reinterpret_cast<const d2*>(lds + buf * G6_STAGE) + GT * GBK / 2 + kq * G6N + wn * 32 + cx;
#pragma unroll
    for (int i = 0; i < 4; ++i) a_[i] = Ap[16 * i];
#pragma unroll
    for (int j = 0; j < 2; ++j) b_[j] = Bp[16 * j];
  };
  auto mfma8 = [&](const d2 (&a_)[4], const d2 (&b_)[2], int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(t ? a_[i].y : a_[i].x, t ? b_[j].y : b_[j].x, acc[i][j], 0, 0, 0);
  };
  load_ab(0);
  commit(0);
  __syncthreads();
  d2 a0[4], b0[2], a1[4], b1[2];
  read_half(a0, b0, 0, 0);
  auto step = [&](int kt, auto has_next_t) {
    constexpr bool HN = decltype(has_next_t)::value;
    const int cur = kt & 1;
    if constexpr (HN) load_ab(kt + 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(a0, b0, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_half(a1, b1, cur, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(a0, b0, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(a1, b1, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (HN) {
      commit(cur ^ 1);
      __syncthreads();
      read_half(a0, b0, cur ^ 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma8(a1, b1, 1);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int kt = 0; kt + 1 < nk32; ++kt) step(kt, std::true_type{});
  step(nk32 - 1, std::false_type{});

  // ---- epilogue: the accumulator holds -C + A B^T (full tiles) or A B^T (edge tiles)
  if constexpr (FULL) {
    double* Ct2g = Ct;
    unsigned coff2 = coff;
    asm volatile("" : "+v"(Ct2g), "+v"(coff2));
    __attribute__((address_space(1))) double* Ct2 = (__attribute__((address_space(1))) double*)Ct2g;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) (Ct2 + (int64_t)(i * 16 + 4 * r) * g.ldc)[coff2 + j * 16] = -acc[i][j][r];
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t gc = col0 + wn * 32 + j * 16 + c16;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t gr = row0 + wm * 64 + i * 16 + gq + 4 * r;
          if (gc < g.N && gr < g.M) g.C[gr * g.ldc + gc] -= acc[i][j][r];
        }
    }
  }
}

// block index -> tile: super tiles of 8 x 16 tiles (1024 x 1024 elements, 128 workgroups; block b runs on XCD b % 8)
__device__ __forceinline__ void gemm_block_n64(const GemmArgs& g, double* __restrict__ lds, int64_t b) {
  if (b < g.tiles2) {  // second problem
    const int tn2 = (int)((g.N2 + G6N - 1) / G6N);
    const int64_t ti = b / tn2, tj = b - ti * tn2;
    if (ti * GT < g.N2 && tj * G6N > ti * GT + GT - 1) return;  // above the diagonal of its leading N2 x N2 block
    GemmArgs h = g;
    h.A = g.A2; h.B = g.B2; h.C = g.C2; h.M = g.M2; h.N = g.N2; h.K = g.K2;
    h.aligned = ((reinterpret_cast<uintptr_t>(g.A2) | reinterpret_cast<uintptr_t>(g.B2)) & 15) == 0 && g.aligned;
    const int64_t row0 = ti * GT, col0 = tj * G6N;
    const bool full = (row0 + GT <= h.M) && (col0 + G6N <= h.N) && ((h.K & (GBK - 1)) == 0) && h.aligned;
    if (full) gemm_tile_body_n64<true>(h, lds, row0, col0);
    else gemm_tile_body_n64<false>(h, lds, row0, col0);
    if (g.ready && ti < g.ready_rows && col0 < (int64_t)g.ready_rows * GT) {
      __threadfence();
      __syncthreads();
      if (threadIdx.x == 0) atomicAdd(g.ready, 1);
    }
    return;
  }
  b -= g.tiles2;
  const int64_t xcd = b & 7, loc = b >> 3;
  const int64_t s = g.s_begin + (loc >> 7) * 8 + xcd;
  const int within = (int)(loc & 127);
  if (s >= g.n_super) return;
  int64_t SI, SJ;
  if (g.lower) {
    int64_t sr = s, shift = 0;
    const int64_t sm = (g.tiles_m + 7) / 8;
    if (g.col0_first) {
      if (s < sm) { SI = s; SJ = 0; sr = -1; }
      else { sr = s - sm; shift = 1; }
    }
    if (sr >= 0) {
      SI = (int64_t)((sqrt(8.0 * (double)sr + 1.0) - 1.0) * 0.5);
      while (SI * (SI + 1) / 2 > sr) --SI;
      while ((SI + 1) * (SI + 2) / 2 <= sr) ++SI;
      SJ = sr - SI * (SI + 1) / 2;
      SI += shift; SJ += shift;
    }
  } else {
    SI = s / g.super_n;
    SJ = s - SI * g.super_n;
  }
  const int64_t ti = SI * 8 + (within >> 4), tj = SJ * 16 + (within & 15);
  if (ti >= g.tiles_m || tj >= g.tiles_n) return;
  const int64_t row0 = ti * GT, col0 = tj * G6N;
  if (g.lower && col0 > row0 + GT - 1) return;
  const bool full = (row0 + GT <= g.M) && (col0 + G6N <= g.N) && ((g.K & (GBK - 1)) == 0) && g.aligned;
  if (full) gemm_tile_body_n64<true>(g, lds, row0, col0);
  else gemm_tile_body_n64<false>(g, lds, row0, col0);
  if (g.ready && g.tiles2 == 0 && ti < g.ready_rows && col0 < (int64_t)g.ready_rows * GT) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(g.ready, 1);
  }
}

__global__ void __launch_bounds__(256, 3) gemm_nt_sub_n64_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) double lds[2 * G6_STAGE];  // 48 KB
  static_assert(2 * G6_STAGE >= 64 * 65 + 64 + 128, "LDS of the diagonal role (SMALL form)");
  int64_t b = blockIdx.x;
  if (g.diagA != nullptr) {
    if (b == 0) {
      if (g.ready) {
        if (threadIdx.x == 0) {
          int spins = 0;
          while (__hip_atomic_load(g.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.ready_target) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1 << 24)) { atomicExch(g.ready + 1, 1); break; }
          }
        }
        __syncthreads();
        __threadfence();
      }
      diag_block_role<true>(g.diagA, g.ldc, g.diag_nbw, g.diag_off, g.diag_info, lds);
      return;
    }
    b -= 1;
  }
  gemm_block_n64(g, lds, b);
}

// Persistent form of the fused launch (round 6, option gemm.persist): 2 workgroups per CU stay resident and pull tiles
// from per-XCD counters in the order the hardware dispatcher would have started them (item i of XCD x = block 8 i + x:
// the same super-tile -> L2 mapping), so a slot never waits for a workgroup launch between two tiles (measured with the
// traced kernel: 8-22 k cycles from a tile's last store to the first instruction of the next workgroup on that CU, during
// which the CU's other workgroup runs alone at ~0.7 of the pipe -- profiles/r06_gemm_trace.txt).  The index of the NEXT
// tile is requested before a tile's last k-tile (gemm_tile_body); workgroup 0 factors the diagonal block first and then joins.
template <int CKS, int FLAGS>
__global__ void __launch_bounds__(256, 2) gemm_nt_sub_persist_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) double lds[2][2][GT * GPITCH];
  __shared__ unsigned s_item[2];
  if (blockIdx.x == 0 && g.diagA != nullptr) {
    if (g.ready) {
      if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(g.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.ready_target) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1 << 24)) { atomicExch(g.ready + 1, 1); break; }
        }
      }
      __syncthreads();
      __threadfence();
    }
    diag_block_role(g.diagA, g.ldc, g.diag_nbw, g.diag_off, g.diag_info, &lds[0][0][0]);
    __syncthreads();
  }
  const unsigned xcd = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;  // XCC_ID: the XCD this workgroup really runs on
  unsigned* const q = g.queue + xcd * 16;
  // items of this XCD: first the second problem's blocks v = 8 i + xcd < tiles2, then the tile list's blocks
  const unsigned n2 = (unsigned)g.tiles2 > xcd ? ((unsigned)g.tiles2 - xcd + 7u) / 8u : 0u;
  const unsigned n_mine = n2 + (unsigned)((g.n_items - g.tiles2) / 8);
  if (threadIdx.x == 0) s_item[0] = atomicAdd(q, 1u);
  __syncthreads();
  unsigned item = __builtin_amdgcn_readfirstlane(s_item[0]);
  while (item < n_mine) {
    if (threadIdx.x == 0) s_item[1] = 0u;  // set by the tile body once it has requested the next item
    // the arguments are re-read from the kernarg segment per tile (an opaque pointer): kept live across the tile body they
    // cost ~300 SGPR spills
    // (constant address space: scalar loads; through a generic pointer they would be vector loads and every tile
    //  parameter a lane value)
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass of hipcc parses kernel bodies too and has no address spaces)
    typedef const __attribute__((address_space(4))) GemmArgs* kernarg_ptr;
    kernarg_ptr gp = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();  // GemmArgs is the kernel's only argument
    asm volatile("" : "+s"(gp));
    GemmArgs gl = *gp;
#else
    GemmArgs gl = g;
#endif
    gl.queue = q;  // this XCD's counter
    const int64_t v = item < n2 ? (int64_t)item * 8 + xcd : (int64_t)gl.tiles2 + (int64_t)(item - n2) * 8 + xcd;
    gemm_block<false, true, true, CKS, FLAGS>(gl, lds, v, s_item);
    // every wavefront is done with the LDS tiles; s_item is visible.  LDS-only barriers: __syncthreads() carries a vmcnt(0),
    // i.e. it would wait for the tile's 64 stores per lane to drain (5-12 k cycles) before the next tile's loads are issued
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (s_item[1] == 0u) {  // a skipped or ragged tile: nothing was requested on the way (workgroup-uniform)
      if (threadIdx.x == 0) s_item[0] = atomicAdd(q, 1u);
      __syncthreads();
    }
    item = __builtin_amdgcn_readfirstlane(s_item[0]);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (s_item is rewritten at the top)
  }
}

// Traced instantiation of the production loop (option gemm.trace): the same launch, every full tile leaves its stamps.
__global__ void __launch_bounds__(256, 2) gemm_nt_sub_diag_trace_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) double lds[2][2][GT * GPITCH];
  if (blockIdx.x == 0) {
    if (g.ready) {
      if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(g.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.ready_target) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1 << 24)) { atomicExch(g.ready + 1, 1); break; }
        }
      }
      __syncthreads();
      __threadfence();
    }
    diag_block_role(g.diagA, g.ldc, g.diag_nbw, g.diag_off, g.diag_info, &lds[0][0][0]);
    return;
  }
  gemm_block<false, true, true, 5, 1>(g, lds, (int64_t)blockIdx.x - 1);
}

__global__ void __launch_bounds__(64) potrf64_kernel(double* __restrict__ A, int64_t ld, int w,
                                                     int64_t global_off, int* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) double T[64 * 65];
  __shared__ __attribute__((aligned(16))) double col[64 * 8];
  const int lane = threadIdx.x;
  // coalesced load into LDS, identity padding outside the w x w block; all 64 row loads in flight at once (clamped
  // addresses + selects: as a loop around a conditional load each row waited for its own round trip, ~50 of ~80 us)
  {
    double tv[64];
    const int lc = lane < w ? lane : w - 1;
#pragma unroll
    for (int r = 0; r < 64; ++r) tv[r] = A[(int64_t)(r < w ? r : w - 1) * ld + lc];
#pragma unroll
    for (int r = 0; r < 64; ++r) T[r * 65 + lane] = (r < w && lane < w) ? tv[r] : ((r == lane) ? 1.0 : 0.0);
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  const int fail = potrf64_wave(T, col, lane);
  if (fail != 0 && fail <= w && lane == 0) atomicCAS(info, 0, (int)(global_off + fail));
  for (int r = 0; r < w; ++r)
    if (lane <= r && lane < w) A[(int64_t)r * ld + lane] = T[r * 65 + lane];
}

// Fused step of the panel factorisation: every workgroup factors the diagonal block itself (one wavefront,
// redundantly -- cheaper than a separate launch on the panel's dependent chain) and then solves its 256 rows
// below against it.  The diagonal block cannot be written back in place by this launch (other workgroups are
// still reading the unfactored block), so workgroup 0 saves L to `Lsave` and the NEXT launch of the chain (or
// writeback_block_kernel at the end of the panel) stores it: nothing reads a factored diagonal block again
// before the triangular solves.
__global__ void __launch_bounds__(256) potrf_trsm64_kernel(double* __restrict__ A, double* __restrict__ X,
                                                           int64_t ld, int w, int64_t m, int64_t global_off,
                                                           int* __restrict__ info, double* __restrict__ Lsave,
                                                           const double* __restrict__ Lprev,
                                                           double* __restrict__ Aprev, int wprev) {
  __shared__ __attribute__((aligned(16))) double T[64 * 65];
  __shared__ __attribute__((aligned(16))) double Ls[64 * 64];  // L^T (row c = column c of L)
  __shared__ __attribute__((aligned(16))) double col[64 * 8];
  __shared__ double rinv[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Every global load of this prologue is issued before the first one is used (clamped addresses + selects, no branches
  // around loads): as loops with a conditional load inside, the 16 rows of the block per wavefront and the 16 entries per
  // thread of the deferred write-back each waited for their own round trip -- ~30 of the kernel's ~50 us at the head of
  // every 64-column step of the panel chain (profiles/r04_step_chain.txt).
  double pv[16];
  const bool wb = blockIdx.x == 0 && Lprev != nullptr;  // deferred write-back of the previous diagonal block
  if (wb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) pv[i] = Lprev[tid + 256 * i];
  }
  {
    double tv[16];
    const int lc = lane < w ? lane : w - 1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = wave + 4 * i;
      tv[i] = A[(int64_t)(r < w ? r : w - 1) * ld + lc];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {  // identity padding outside the w x w block
      const int r = wave + 4 * i;
      T[r * 65 + lane] = (r < w && lane < w) ? tv[i] : ((r == lane) ? 1.0 : 0.0);
    }
  }
  if (wb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + 256 * i, r = e >> 6, c = e & 63;
      if (r < wprev && c <= r) Aprev[(int64_t)r * ld + c] = pv[i];
    }
  }
  // this thread's row of the strip: requested here, needed after the factorisation (row-per-thread accesses are 64
  // separate 16-byte segments per instruction; measured neutral at n = 6300, -1 ms at n = 63 000)
  const int64_t r = (int64_t)blockIdx.x * 256 + tid;
  double* xr = X + (r < m ? r : m - 1) * ld;
  double x[64];
  const bool al16 = ((reinterpret_cast<uintptr_t>(xr) & 15) == 0) && (w == 64);
  if (al16) {
#pragma unroll
    for (int c = 0; c < 64; c += 2) {
      d2 v = *reinterpret_cast<const d2*>(xr + c);
      x[c] = v.x;
      x[c + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 64; ++c) x[c] = (c < w) ? xr[c] : 0.0;
  }
  __syncthreads();
  if (wave == 0) {
    const int fail = potrf64_wave(T, col, lane);
    if (blockIdx.x == 0 && fail != 0 && fail <= w && lane == 0) atomicCAS(info, 0, (int)(global_off + fail));
  }
  __syncthreads();
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    const double v = (c <= r) ? T[r * 65 + c] : 0.0;  // lower triangle incl. diagonal; identity padding kept
    if (blockIdx.x == 0) Lsave[e] = v;
    if (r == c) rinv[r] = 1.0 / v;
  }
  for (int e = tid; e < 64 * 64; e += 256) {  // L^T: consecutive threads read a column of T (pitch 65), write a row of Ls
    const int c = e >> 6, r = e & 63;
    Ls[e] = (c <= r) ? T[r * 65 + c] : 0.0;
  }
  __syncthreads();
  if (r >= m) return;
  // right-looking substitution: the same multiply-adds in the same order for every entry as a left-looking row loop, but the
  // dependent chain is one multiply-add + one multiply per column instead of the whole dot product (c terms)
#pragma unroll
  for (int c = 0; c < 64; ++c) {
    const double xc = x[c] * rinv[c];
    x[c] = xc;
    const double* Lc = Ls + c * 64;  // L[c'][c], c' = 0..63
    if ((c + 1) & 1) {  // c + 1 odd: one single entry, then aligned pairs
      if (c + 1 < 64) x[c + 1] -= xc * Lc[c + 1];
#pragma unroll
      for (int k = c + 2; k < 64; k += 2) {
        const d2 l = *reinterpret_cast<const d2*>(Lc + k);
        x[k] -= xc * l.x;
        x[k + 1] -= xc * l.y;
      }
    } else {
#pragma unroll
      for (int k = c + 1; k < 64; k += 2) {
        const d2 l = *reinterpret_cast<const d2*>(Lc + k);
        x[k] -= xc * l.x;
        x[k + 1] -= xc * l.y;
      }
    }
  }
  if (al16) {
#pragma unroll
    for (int c = 0; c < 64; c += 2) {
      d2 v = {x[c], x[c + 1]};
      *reinterpret_cast<d2*>(xr + c) = v;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 64; ++c)
      if (c < w) xr[c] = x[c];
  }
}

__global__ void __launch_bounds__(256) writeback_block_kernel(const double* __restrict__ Lprev,
                                                              double* __restrict__ Aprev, int64_t ld, int wprev) {
  double pv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) pv[i] = Lprev[threadIdx.x + 256 * i];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = threadIdx.x + 256 * i, r = e >> 6, c = e & 63;
    if (r < wprev && c <= r) Aprev[(int64_t)r * ld + c] = pv[i];
  }
}

// ------------------------------------------------------------------------------------------
// X[r, 0:w] <- X[r, 0:w] * L^-T for m rows (L = w x w lower block at Ld, w <= 64).  Exact forward
// substitution per row (no explicit inverse): one thread owns one row, holds it in registers
// (fully unrolled), L is broadcast from LDS.  x_c = (x_c - sum_{k<c} x_k L[c][k]) / L[c][c].
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) trsm64_kernel(const double* __restrict__ Ld,
                                                     double* __restrict__ X, int64_t ld, int w,
                                                     int64_t m, int64_t ldl) {
  __shared__ __attribute__((aligned(16))) double Ls[64 * 64];
  const int tid = threadIdx.x;
  for (int e = tid; e < 64 * 64; e += 256) {
    int r = e >> 6, c = e & 63;
    double v = (r == c) ? 1.0 : 0.0;  // identity padding for w < 64
    if (r < w && c < w && c <= r) v = Ld[r * ldl + c];
    Ls[e] = v;
  }
  __syncthreads();
  __shared__ double rinv[64];  // reciprocals of the diagonal: 64 divisions per workgroup instead of 64 per row
  if (tid < 64) rinv[tid] = 1.0 / Ls[tid * 64 + tid];
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * 256 + tid;
  if (r >= m) return;
  double* xr = X + r * ld;
  double x[64];
  const bool al16 = ((reinterpret_cast<uintptr_t>(xr) & 15) == 0) && (w == 64);
  if (al16) {
#pragma unroll
    for (int c = 0; c < 64; c += 2) {
      d2 v = *reinterpret_cast<const d2*>(xr + c);
      x[c] = v.x;
      x[c + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 64; ++c) x[c] = (c < w) ? xr[c] : 0.0;
  }
#pragma unroll
  for (int c = 0; c < 64; ++c) {
    double s = x[c];
#pragma unroll
    for (int k = 0; k < c; ++k) s -= x[k] * Ls[c * 64 + k];
    x[c] = s * rinv[c];
  }
  if (al16) {
#pragma unroll
    for (int c = 0; c < 64; c += 2) {
      d2 v = {x[c], x[c + 1]};
      *reinterpret_cast<d2*>(xr + c) = v;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 64; ++c)
      if (c < w) xr[c] = x[c];
  }
}

// ------------------------------------------------------------------------------------------
// Row-local panel solve:  X[r, 0:nb] <- X[r, 0:nb] * L^-T  for m rows, L = nb x nb lower (already factored
// diagonal block of the panel), nb = 64 * nbw <= 512.  Replaces the 8 x (trsm64 + K=64 GEMM over the whole
// panel strip) of the panel chain by ONE launch in which every workgroup finishes its own 32 rows: the strip
// is read once and written once instead of eight read-modify-write passes, and the products run on the MFMA pipe.
//
// Workgroup = 32 rows, 4 wavefronts.  The 32 x nb strip lives in REGISTERS in MFMA C-layout: wavefront w owns
// the 64-column blocks {w, 7 - w} (each block 2 x 4 tiles of 16 x 16) -- the right-looking update work of
// block c is c block products, so every wavefront does 7.  Step jj:
//   (1) the owner of block jj writes its (fully updated) block to LDS in row-major form;
//   (2) exact substitution with the 64 x 64 diagonal block L_jj (no explicit inverse, cond(K) ~ 1/lam): 8 threads
//       per row, thread q holds the columns k = q (mod 8); column by column the solved entry is broadcast inside
//       the 8-lane group and the remaining entries are updated (right-looking: independent FMAs); L_jj^T from LDS;
//   (3) the solved block goes to global memory and stays in LDS as the A operand of
//   (4) acc_c -= X_jj * L[c, jj]^T for the blocks c > jj of every wavefront: v_mfma_f64_16x16x4_f64, B operand
//       straight from global memory (L is 2 MB at most and L2 resident).  Both operands are fetched as 32-byte
//       runs (4 consecutive k per lane): MFMA step s of a 16-deep chunk contracts k = 4 (lane >> 4) + s, the same
//       bijection on both sides.
// LDS 50 KB, ~200 VGPRs: two workgroups per CU, the second one covers the substitution phases of the first.
// ------------------------------------------------------------------------------------------
#define PT_ROWS 32
#define PT_SP 66   // pitch of the row-major block in LDS (doubles): 16-byte aligned rows, 16-lane b128 reads conflict-free
#define PT_LP 65   // pitch of L_jj^T

#define PT_TB (64 * PT_LP + 64)  // per 64-block image in the prep buffer: L_jj^T (pitch PT_LP) followed by 1 / diag

// One workgroup per 64 x 64 diagonal block of L: transposed copy (zero above the diagonal) + reciprocal pivots, in the
// exact LDS image of panel_trsm_kernel.  Every row block of the solve used to redo this transpose (and its 64 divisions)
// per step; it was 13 of the kernel's 53 ms (profiles/r02_panel_trsm_breakdown.txt).
__global__ void __launch_bounds__(256) panel_trsm_prep_kernel(const double* __restrict__ L, int64_t ldl,
                                                              double* __restrict__ Tb) {
  const int jj = blockIdx.x, tid = threadIdx.x;
  double* T = Tb + (int64_t)jj * PT_TB;
  double lv[16];  // all 16 loads of a thread in flight at once (a conditional load in the loop waits for each round trip)
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = tid + 256 * i;
    lv[i] = L[(int64_t)(jj * 64 + (e >> 6)) * ldl + jj * 64 + (e & 63)];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = tid + 256 * i, r = e >> 6, c = e & 63;  // L_jj[r][c], c <= r
    const double v = (c <= r) ? lv[i] : 0.0;
    T[c * PT_LP + r] = v;
    if (r == c) T[64 * PT_LP + r] = 1.0 / v;
  }
  if (tid < 64) T[tid * PT_LP + 64] = 0.0;  // pitch padding (copied along, never read)
}

template <bool ABL>
__global__ void __launch_bounds__(256, 2) panel_trsm_kernel(const double* __restrict__ Tb, const double* __restrict__ L, int64_t ldl,
                                                            double* __restrict__ X, int64_t ld, int nbw, int64_t m, int dbg) {
  __shared__ __attribute__((aligned(16))) double S[PT_ROWS * PT_SP];
  __shared__ __attribute__((aligned(16))) double Lt[PT_TB];
  double* const rinv = Lt + 64 * PT_LP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * PT_ROWS;
  // ---- strip into registers (C layout: row = lk + 4 r + 16 i, col = li + 16 j); two separately named register
  // blocks (an array indexed by the owner test would be demoted to scratch memory)
  d4 accA[2][4], accB[2][4];
  const int blkA = wave, blkB = 7 - wave;
  // every load of the strip is issued before the first use: clamped addresses + selects, no branch around a load (with the
  // `blk < nbw` test inside the tile loop the compiler emitted 16 branches, each 4 loads + s_waitcnt vmcnt(0): 16 memory
  // round trips per workgroup before the first step)
  auto load_block = [&](d4 (&acc)[2][4], int blk) {
    const bool have = blk < nbw && !(ABL && (dbg & 8));
    const double* pb = X + (have ? blk : 0) * 64 + li;
    double w[2][4][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t gr = row0 + 16 * i + lk;
      // rows past m read row m-1 (clamped index + select instead of predicated loads)
      const int64_t r0 = gr < m ? gr : m - 1, r1 = gr + 4 < m ? gr + 4 : m - 1;
      const int64_t r2 = gr + 8 < m ? gr + 8 : m - 1, r3 = gr + 12 < m ? gr + 12 : m - 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double* pc = pb + 16 * j;
        w[i][j][0] = pc[r0 * ld];  // (non-temporal loads of the strip: measured neutral, profiles/r04_gemm_peel_stagger_ab.txt)
        w[i][j][1] = pc[r1 * ld];
        w[i][j][2] = pc[r2 * ld];
        w[i][j][3] = pc[r3 * ld];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t gr = row0 + 16 * i + lk;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = (d4){(have && gr < m) ? w[i][j][0] : 0.0, (have && gr + 4 < m) ? w[i][j][1] : 0.0,
                         (have && gr + 8 < m) ? w[i][j][2] : 0.0, (have && gr + 12 < m) ? w[i][j][3] : 0.0};
    }
  };
  load_block(accA, blkA);
  load_block(accB, blkB);

  auto to_lds = [&](const d4 (&acc)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(16 * i + lk + 4 * r) * PT_SP + 16 * j + li] = acc[i][j][r];
  };
  auto update = [&](d4 (&acc)[2][4], int c, int jj) {
    const double* Lc = L + (int64_t)(c * 64 + li) * ldl + jj * 64 + 4 * lk;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {  // 16-deep chunks of the 64 contraction indices
      d4 a[2], bb[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const d4*>(&S[(16 * i + li) * PT_SP + 16 * ch + 4 * lk]);
#pragma unroll
      for (int j = 0; j < 4; ++j) bb[j] = *reinterpret_cast<const d4*>(Lc + (int64_t)(16 * j) * ldl + 16 * ch);
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)  // acc -= a b^T  ==  acc += (-a) b^T
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[i][sidx], bb[j][sidx], acc[i][j], 0, 0, 0);
    }
  };

  for (int jj = 0; jj < nbw; ++jj) {
    // (1) owner's block -> LDS row-major; L_jj^T and reciprocal diagonal -> LDS
    if (jj < 4) {
      if (wave == jj) to_lds(accA);
    } else {
      if (wave == 7 - jj) to_lds(accB);
    }
    if (!(ABL && (dbg & 4))) {  // L_jj^T and reciprocal pivots: straight copy of the prepared image
      const d2* src = reinterpret_cast<const d2*>(Tb + (int64_t)jj * PT_TB);
      d2* dst = reinterpret_cast<d2*>(Lt);
#pragma unroll
      for (int k = 0; k < (PT_TB / 2 + 255) / 256; ++k) {
        const int e = tid + 256 * k;
        if (e < PT_TB / 2) dst[e] = src[e];
      }
    }
    __syncthreads();
    // (2) substitution: x_c = t_c / L[c][c];  t_k -= x_c L[k][c] for k > c.  8 threads per row (subst64_row8)
    if (!ABL || !(dbg & 1)) {
      const int srow = tid >> 3, sq = tid & 7;
      double t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = S[srow * PT_SP + sq + 8 * i];
      if (!(ABL && (dbg & 32))) subst64_row8<PT_LP>(t, Lt, rinv, sq);
      const int64_t gr = row0 + srow;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        S[srow * PT_SP + sq + 8 * i] = t[i];
        if (gr < m && !(ABL && (dbg & 16))) X[gr * ld + jj * 64 + sq + 8 * i] = t[i];
      }
    }
    __syncthreads();
    // (4) right-looking update of the blocks c > jj this wavefront owns
    if (!ABL || !(dbg & 2)) {
      if (blkA > jj && blkA < nbw) update(accA, blkA, jj);
      if (blkB > jj && blkB < nbw) update(accB, blkB, jj);
    }
    __syncthreads();  // S and Lt are rewritten by the next step
  }
}

// L: nb x nb lower block (leading dimension ldl; 0 = the same matrix as X), X: m rows of leading dimension ld
int launch_panel_trsm(gdml_ctx* ctx, hipStream_t st, const double* L, double* X, int64_t ld, int nb, int64_t m,
                      int64_t ldl) {
  if (m <= 0) return GDML_OK;
  const int dbg = ctx_opt_i(ctx, "trsm.debug", 0);  // timing-only ablation bits: 1 no substitution, 2 no MFMA update,
                                                    // 4 no L_jj staging, 8 no strip load, 16 no stores
  double* Tb = nullptr;
  GDML_TRY(ctx_slot(ctx, 8, (int64_t)8 * PT_TB * 8, &Tb));
  const int slot = (st == (ctx->kt_stream ? ctx->kt_stream : ctx->stream)) ? ktime_begin(ctx) : -1;
  hipLaunchKernelGGL(panel_trsm_prep_kernel, dim3((unsigned)(nb / 64)), dim3(256), 0, st, L, ldl > 0 ? ldl : ld, Tb);
  if (dbg)
    hipLaunchKernelGGL(panel_trsm_kernel<true>, dim3((unsigned)ceil_div(m, PT_ROWS)), dim3(256), 0, st, Tb, L, ldl > 0 ? ldl : ld,
                       X, ld, nb / 64, m, dbg);
  else
    hipLaunchKernelGGL(panel_trsm_kernel<false>, dim3((unsigned)ceil_div(m, PT_ROWS)), dim3(256), 0, st, Tb, L, ldl > 0 ? ldl : ld,
                       X, ld, nb / 64, m, 0);
  ctx->launch_counter++;
  ktime_end(ctx, slot, "panel_trsm", (double)m * (double)nb * (double)nb);
  ctx->launch_counter++;
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}

// A <- -A + lam I on the lower triangle (analytic.py:65,82), tiled copy-free.
__global__ void __launch_bounds__(256) negate_shift_kernel(double* __restrict__ A, int64_t n,
                                                           int64_t ld, double lam) {
  const int64_t r = blockIdx.x;
  double* row = A + r * ld;
  for (int64_t c = threadIdx.x; c <= r; c += 256) {
    double v = -row[c];
    if (c == r) v += lam;
    row[c] = v;
  }
}

int launch_trsm64(gdml_ctx* ctx, hipStream_t st, const double* Ld, double* X, int64_t ld, int w,
                  int64_t m, int64_t ldl) {
  if (m <= 0) return GDML_OK;
  hipLaunchKernelGGL(trsm64_kernel, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, st, Ld, X, ld, w, m, ldl > 0 ? ldl : ld);
  ctx->launch_counter++;
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}

// Rank-64 update inside the panel chain:  C[m x ncols] -= A[m x 64] B[ncols x 64]^T with a handful of tiles (the rest of an
// nb x nb diagonal block after one 64-column step: m, ncols <= 448).  The 128 x 128 GEMM tile kernel is the wrong tool
// there: most tiles are ragged (its guarded path loads element-wise and runs a load-subtract-store epilogue in four
// serialised batches) and a launch took 20-60 us for < 30 MFLOP, on the dependent chain of every panel.  Here one wavefront
// owns a 32 x 32 tile (2 x 2 MFMA tiles), both operands come straight from global memory (L2) in MFMA layout as 32-byte
// runs and ALL loads of the tile -- C included -- are issued before the first use: one memory round trip + 64 MFMAs.
// Ragged edges: clamped row indices on the loads, guarded stores.  skip_upper: tiles strictly above the diagonal are not
// computed (C's origin lies on the matrix diagonal; the strict upper triangle is scratch).
__global__ void __launch_bounds__(256) rank64_update_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                            double* __restrict__ C, int64_t ld, int m, int ncols,
                                                            int skip_upper) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int tn = (ncols + 31) >> 5;
  const int t = blockIdx.x * 4 + wave;
  const int ti = t / tn, tj = t - ti * tn;
  if (32 * ti >= m) return;
  if (skip_upper && tj > ti) return;
  const int r0 = 32 * ti, q0 = 32 * tj;
  d4 acc[2][2], a[2][4], bb[2][4];
  int crow[2][4], ccol[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) ccol[j] = q0 + 16 * j + li;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) crow[i][r] = r0 + 16 * i + lk + 4 * r;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = r0 + 16 * i + li, rb = q0 + 16 * i + li;
    const double* Ap = A + (int64_t)(ra < m ? ra : m - 1) * ld + 4 * lk;
    const double* Bp = B + (int64_t)(rb < ncols ? rb : ncols - 1) * ld + 4 * lk;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      a[i][ch] = *reinterpret_cast<const d4*>(Ap + 16 * ch);
      bb[i][ch] = *reinterpret_cast<const d4*>(Bp + 16 * ch);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = crow[i][r] < m ? crow[i][r] : m - 1, cc = ccol[j] < ncols ? ccol[j] : ncols - 1;
        acc[i][j][r] = C[(int64_t)rr * ld + cc];
      }
#pragma unroll
  for (int ch = 0; ch < 4; ++ch)
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[i][ch][sidx], bb[j][ch][sidx], acc[i][j], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (crow[i][r] < m && ccol[j] < ncols) C[(int64_t)crow[i][r] * ld + ccol[j]] = acc[i][j][r];
}

// Factor one panel: columns [k0, k0+nb), rows [k0, n), 64-wide sub-steps (potrf64 / trsm64 / K=64 gemm).
int panel_factor_steps(gdml_ctx* ctx, hipStream_t st, double* A, int64_t n, int64_t ld, int64_t k0, int64_t nb);

// Panel = diagonal block (the 64-wide step chain, but only over the nb rows of the block) + ONE row-local solve of
// all rows below (option chol.panel_kernel = 1, default); the step chain over the whole strip otherwise.
static int panel_factor(gdml_ctx* ctx, hipStream_t st, double* A, int64_t n, int64_t ld, int64_t k0,
                        int64_t nb) {
  const int64_t below = n - k0 - nb;
  const bool aligned = (reinterpret_cast<uintptr_t>(A) & 31) == 0 && (ld % 4 == 0) && (k0 % 4 == 0);
  if (ctx_opt_i(ctx, "chol.panel_kernel", 1) && nb % 64 == 0 && nb <= 512 && below > 0 && aligned) {
    GDML_TRY(panel_factor_steps(ctx, st, A, k0 + nb, ld, k0, nb));
    return launch_panel_trsm(ctx, st, A + k0 * ld + k0, A + (k0 + nb) * ld + k0, ld, (int)nb, below);
  }
  return panel_factor_steps(ctx, st, A, n, ld, k0, nb);
}

int panel_factor_steps(gdml_ctx* ctx, hipStream_t st, double* A, int64_t n, int64_t ld, int64_t k0,
                       int64_t nb) {
  const int fused = ctx_opt_i(ctx, "chol.panel_fused", 1);  // 0: separate potrf64 / trsm64 launches
  const int small_upd = ctx_opt_i(ctx, "chol.small_update", 1);  // 0: the rank-64 updates of the chain through the GEMM tile kernel (A/B)
  double* save = nullptr;  // two 64 x 64 slots for the deferred write-back of the diagonal blocks
  if (fused) GDML_TRY(ctx_slot(ctx, 5, 2 * 4096 * 8, &save));
  const double* Lprev = nullptr;
  double* Aprev = nullptr;
  int wprev = 0, flip = 0;
  for (int64_t jj = 0; jj < nb; jj += 64) {
    const int64_t c0 = k0 + jj;
    const int w = (int)((nb - jj < 64) ? nb - jj : 64);
    double* Ad = A + c0 * ld + c0;
    const int64_t m = n - c0 - w;
    if (fused && m > 0) {
      double* X = A + (c0 + w) * ld + c0;
      double* Lsave = save + flip * 4096;
      hipLaunchKernelGGL(potrf_trsm64_kernel, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, st, Ad, X, ld, w, m,
                         c0, ctx->d_info, Lsave, Lprev, Aprev, wprev);
      ctx->launch_counter++;
      Lprev = Lsave;
      Aprev = Ad;
      wprev = w;
      flip ^= 1;
    } else {
      if (Lprev) {  // flush the pending diagonal block before a plain step
        hipLaunchKernelGGL(writeback_block_kernel, dim3(1), dim3(256), 0, st, Lprev, Aprev, ld, wprev);
        ctx->launch_counter++;
        Lprev = nullptr;
      }
      hipLaunchKernelGGL(potrf64_kernel, dim3(1), dim3(64), 0, st, Ad, ld, w, c0, ctx->d_info);
      ctx->launch_counter++;
      if (m > 0) {
        double* X = A + (c0 + w) * ld + c0;
        hipLaunchKernelGGL(trsm64_kernel, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, st, Ad, X, ld,
                           w, m, ld);
        ctx->launch_counter++;
      }
    }
    if (m > 0) {
      double* X = A + (c0 + w) * ld + c0;
      const int64_t ncols = k0 + nb - (c0 + w);
      if (ncols > 0) {
        // rest of the panel:  C[c0+w:n, c0+w:k0+nb] -= X[c0+w:n, :] X[c0+w:k0+nb, :]^T
        const bool al32 = (reinterpret_cast<uintptr_t>(X) & 31) == 0 && (ld % 4 == 0);
        if (w == 64 && m <= 2048 && al32 && small_upd) {  // a handful of tiles: one wavefront per 32 x 32 tile
          const int tiles = (int)(ceil_div(m, 32) * ceil_div(ncols, 32));
          hipLaunchKernelGGL(rank64_update_kernel, dim3((unsigned)ceil_div(tiles, 4)), dim3(256), 0, st, X, X,
                             A + (c0 + w) * ld + (c0 + w), ld, (int)m, (int)ncols, 1);
          ctx->launch_counter++;
        } else {
          GDML_TRY(launch_gemm_nt_sub(ctx, st, X, ld, X, ld, A + (c0 + w) * ld + (c0 + w), ld, m, ncols, w, 0));
        }
      }
    }
  }
  if (Lprev) {
    hipLaunchKernelGGL(writeback_block_kernel, dim3(1), dim3(256), 0, st, Lprev, Aprev, ld, wprev);
    ctx->launch_counter++;
  }
  return GDML_OK;
}

// n: order of the matrix; n_rows >= n: rows n..n_rows-1 are carried along (right-hand sides stored as extra
// rows: they go through the panel solves and trailing updates, i.e. through the forward substitution)
static int chol_factor_level(gdml_ctx* ctx, double* A, int64_t n, int64_t ld, int* info_out, int64_t n_rows) {
  if (n_rows < n) n_rows = n;
  HIP_CHECK(ctx, hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->stream));
  if (ctx->gemm_queue) {  // tile counters of the persistent trailing updates: one zeroed set per launch
    HIP_CHECK(ctx, hipMemsetAsync(ctx->gemm_queue, 0, (size_t)ctx->gemm_queue_sets * 512, ctx->stream));
    ctx->gemm_queue_next = 0;
  }
  int64_t NB = (int64_t)ctx_opt(ctx, "chol.nb", 512);  // outer panel width (multiple of 64)
  if (NB < 64 || NB % 64 || NB > 512) NB = 512;  // the diagonal-block role and the row-local solve hold at most 8 x 64 columns
  const bool lookahead = ctx_opt_i(ctx, "chol.lookahead", 1) != 0;
  // ---- default schedule: ONE stream.  Per panel k:  GEMM1 (columns of panel k+1)  ->  SYRK of the rest, whose
  // workgroup 0 factors the diagonal block of panel k+1 meanwhile  ->  row-local solve of panel k+1's rows.
  // The 64-wide step chain of the diagonal block is hidden inside the SYRK launch; only the row-local solve
  // (one launch) stays between two GEMM launches.  Near the end (SYRK shorter than the single-workgroup block
  // factorisation) the block is factored by the multi-workgroup step chain instead.
  if (ctx_opt_i(ctx, "chol.fused_diag", 1) && lookahead) {
    hipStream_t st = ctx->stream;
    const int64_t min_rows = (int64_t)ctx_opt(ctx, "chol.fused_min_rows", 12288);
    // Panel PAIRS: while the trailing matrix is large, two NB-wide panels a | b form an outer panel of OB = 2 NB columns
    // and the bulk of the trailing update runs with K = OB, which halves the C read-modify-write traffic per flop of the
    // SYRK (its epilogue is ~6 % of the launch at K = 512).  The bulk launch is split in two halves of its tile list so
    // that each of the two NB x NB diagonal blocks still has a long launch to hide behind:
    //   GEMM1 (columns of a | b, K = OB)  ->  bulk half 1 + [block a]  ->  solve rows below a  ->  K = NB GEMM onto b's columns
    //   ->  bulk half 2 + [block b]  ->  solve rows below b.
    // Once the bulk gets too short (n - t1 < chol.outer_min_rows) new panels are single NB-wide ones again.
    int64_t OB = (int64_t)ctx_opt(ctx, "chol.outer", 1024);
    if (OB != 2 * NB) OB = NB;
    const int64_t outer_min_rows = (int64_t)ctx_opt(ctx, "chol.outer_min_rows", 16384);
    // "how much trailing matrix is left behind column t": its columns for the square systems of rounds 1-5; for a TALL block
    // (round 6, two-level schedule below: n columns, n_rows >> n rows carried along) the geometric mean of rows and columns,
    // so that the long launches of a tall block keep the paired / fused forms.  floor(sqrt((n_rows - t)(n - t))) = n - t for
    // n_rows = n and n_rows = n + 1 (carried right-hand side): the square schedule is unchanged.
    const double tall_f = ctx_opt(ctx, "chol.block_f", 2.0);
    auto left_at = [&](int64_t t) -> int64_t {
      if (t >= n) return 0;
      if (n_rows - n <= 1) return n - t;
      // (tall: the update is a full rectangle, the square case's a triangle; chol.block_f weighs that -- tuned by measurement)
      return (int64_t)sqrt(tall_f * (double)(n_rows - t) * (double)(n - t));
    };
    auto width_at = [&](int64_t c0) -> int64_t {  // width of the panel that starts at column c0
      const int64_t w = (OB > NB && n - (c0 + OB) > 0 && left_at(c0 + OB) >= outer_min_rows) ? OB : NB;
      return (n - c0 < w) ? n - c0 : w;
    };
    // first panel: always one level (nothing to hide its diagonal block behind)
    int64_t k0 = 0, nb = (n < NB) ? n : NB;
    int ready_count = 0;  // cumulative target of the tile counter d_info[6]
    HIP_CHECK(ctx, hipMemsetAsync(ctx->d_info + 6, 0, 2 * sizeof(int), ctx->stream));  // counter, wait-timeout flag
    GDML_TRY(panel_factor(ctx, st, A, n_rows, ld, 0, nb));
    for (;;) {
      const int64_t t0 = k0 + nb;
      if (t0 >= n) break;
      const int64_t nb2 = width_at(t0);
      const int64_t t1 = t0 + nb2;
      const double* P = A + t0 * ld + k0;
      const bool fuse = (nb2 % 64 == 0) && (left_at(t1) >= min_rows) && (n_rows - t1 > 0);
      // the merged schedule counts finished GT x GT tiles of a diagonal block: NB must be a whole number of tiles, or the
      // counter target is too small and the block is factored before its last update arrived
      const bool merged = fuse && nb2 == 2 * NB && NB % GT == 0 && ctx_opt_i(ctx, "chol.merge_gemm1", 1) != 0;
      if (!merged) GDML_TRY(launch_gemm_nt_sub(ctx, st, P, ld, P, ld, A + t0 * ld + t0, ld, n_rows - t0, nb2, nb, 0));
      if (merged) {
        // ONE lower SYRK over everything right of the finished panel, in two launches of its super-tile list.  The first
        // super-tile column is exactly the columns of a | b (2 NB = 8 tiles): it is enumerated first, workgroup 0 of the
        // first launch waits for the 10 tiles of block a (counter) and factors it while the rest of the launch runs.
        const int64_t ta = t0 + NB;  // first row / column of b
        const int64_t sm = (ceil_div(n_rows - t0, GT) + 7) / 8, n_super_all = sm * (sm + 1) / 2;
        int64_t s_split = n_super_all / 2;
        if (s_split < sm) s_split = sm;
        const double fs = ((double)s_split + 0.5) / (double)n_super_all;
        DiagJob dj;
        dj.A = A + t0 * ld + t0; dj.nbw = (int)(NB / 64); dj.off = t0;
        dj.col0_first = true;
        dj.ready = ctx->d_info + 6;
        // (tiles of a diagonal block: lower 128 x 128 tiles, or with gemm.n64 the 128 x 64 tiles that touch its lower triangle)
        const int block_tiles = gemm_use_n64(ctx) ? (int)((NB / GT) * (NB / GT + 1)) : (int)((NB / GT) * (NB / GT + 1) / 2);
        ready_count += block_tiles;
        dj.ready_target = ready_count;
        GDML_TRY(launch_gemm_nt_sub_part(ctx, st, P, ld, P, ld, A + t0 * ld + t0, ld, n_rows - t0, n - t0, nb, 1, 0.0, fs, true,
                                         &dj));
        double* Xa = A + ta * ld + t0;  // rows below block a (they include b's rows of the outer panel)
        GDML_TRY(launch_panel_trsm(ctx, st, A + t0 * ld + t0, Xa, ld, (int)NB, n_rows - ta));
        // second launch: the K = NB update of b's columns by panel a rides in front of the remaining SYRK tiles; workgroup 0
        // waits for the tiles of block b and factors it
        dj.A = A + ta * ld + ta; dj.off = ta;
        dj.A2 = Xa; dj.B2 = Xa; dj.C2 = A + ta * ld + ta; dj.M2 = n_rows - ta; dj.N2 = NB; dj.K2 = NB;
        ready_count += block_tiles;
        dj.ready_target = ready_count;
        GDML_TRY(launch_gemm_nt_sub_part(ctx, st, P, ld, P, ld, A + t0 * ld + t0, ld, n_rows - t0, n - t0, nb, 1, fs, 1.0, true,
                                         &dj));
        GDML_TRY(launch_panel_trsm(ctx, st, A + ta * ld + ta, A + t1 * ld + ta, ld, (int)NB, n_rows - t1));
      } else if (fuse && nb2 == 2 * NB) {
        const double* P1 = A + t1 * ld + k0;
        const int64_t ta = t0 + NB;  // first row / column of b
        DiagJob dj;
        dj.A = A + t0 * ld + t0; dj.nbw = (int)(NB / 64); dj.off = t0;
        GDML_TRY(launch_gemm_nt_sub_part(ctx, st, P1, ld, P1, ld, A + t1 * ld + t1, ld, n_rows - t1, n - t1, nb, 1, 0.0,
                                         0.5, true, &dj));
        double* Xa = A + ta * ld + t0;  // rows below block a (they include b's rows of the outer panel)
        GDML_TRY(launch_panel_trsm(ctx, st, A + t0 * ld + t0, Xa, ld, (int)NB, n_rows - ta));
        GDML_TRY(launch_gemm_nt_sub(ctx, st, Xa, ld, Xa, ld, A + ta * ld + ta, ld, n_rows - ta, NB, NB, 0));
        dj.A = A + ta * ld + ta; dj.off = ta;
        GDML_TRY(launch_gemm_nt_sub_part(ctx, st, P1, ld, P1, ld, A + t1 * ld + t1, ld, n_rows - t1, n - t1, nb, 1, 0.5,
                                         1.0, true, &dj));
        GDML_TRY(launch_panel_trsm(ctx, st, A + ta * ld + ta, A + t1 * ld + ta, ld, (int)NB, n_rows - t1));
      } else if (fuse) {
        DiagJob dj;
        dj.A = A + t0 * ld + t0;
        dj.nbw = (int)(nb2 / 64);
        dj.off = t0;
        const double* P1 = A + t1 * ld + k0;
        GDML_TRY(launch_gemm_nt_sub_part(ctx, st, P1, ld, P1, ld, A + t1 * ld + t1, ld, n_rows - t1, n - t1, nb, 1, 0.0,
                                         1.0, true, &dj));
        GDML_TRY(launch_panel_trsm(ctx, st, A + t0 * ld + t0, A + t1 * ld + t0, ld, (int)nb2, n_rows - t1));
      } else if (ctx_opt_i(ctx, "chol.tail_lookahead", 1) && t1 < n) {
        // tail: the SYRK is too short to hide the single-workgroup block factorisation; the multi-workgroup step chain and
        // the row-local solve of panel k+1 run on the high-priority stream next to the (small) SYRK grid of panel k
        hipStream_t sp = ctx->stream2;
        HIP_CHECK(ctx, hipEventRecord(ctx->ev_la[0], st));
        HIP_CHECK(ctx, hipStreamWaitEvent(sp, ctx->ev_la[0], 0));
        GDML_TRY(panel_factor(ctx, sp, A, n_rows, ld, t0, nb2));
        HIP_CHECK(ctx, hipEventRecord(ctx->ev_la[1], sp));
        const double* P1 = A + t1 * ld + k0;
        GDML_TRY(launch_gemm_nt_sub(ctx, st, P1, ld, P1, ld, A + t1 * ld + t1, ld, n_rows - t1, n - t1, nb, 1));
        HIP_CHECK(ctx, hipStreamWaitEvent(st, ctx->ev_la[1], 0));
      } else {
        GDML_TRY(panel_factor(ctx, st, A, n_rows, ld, t0, nb2));
        if (t1 < n) {
          const double* P1 = A + t1 * ld + k0;
          GDML_TRY(launch_gemm_nt_sub(ctx, st, P1, ld, P1, ld, A + t1 * ld + t1, ld, n_rows - t1, n - t1, nb, 1));
        }
      }
      k0 = t0;
      nb = nb2;
    }
    HIP_CHECK(ctx, hipGetLastError());
    int info8[8] = {0};
    HIP_CHECK(ctx, hipMemcpyAsync(info8, ctx->d_info, sizeof(info8), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (info8[7] != 0)
      return gdml_fail(ctx, GDML_ERR_HIP, "Cholesky: the diagonal-block workgroup gave up waiting for its tiles (schedule bug)");
    if (info_out) *info_out = info8[0];
    return GDML_OK;
  }
  // ---- round-1 schedule (option chol.fused_diag = 0; A/B reference): the step chain of panel k+1 on a second
  // stream with one panel of look-ahead (the hardware hardly overlaps it: profiles/r02_sched_probe.txt), or fully
  // sequential on one stream with chol.lookahead = 0
  hipStream_t sm = ctx->stream, sp = ctx->stream2;
  hipEvent_t evA = ctx->ev_la[0], evB = ctx->ev_la[1];
  GDML_TRY(panel_factor(ctx, sm, A, n_rows, ld, 0, n < NB ? n : NB));
  for (int64_t k0 = 0; k0 < n; k0 += NB) {
    const int64_t nb = (n - k0 < NB) ? n - k0 : NB;
    const int64_t t0 = k0 + nb;
    if (t0 >= n) break;
    const int64_t nb2 = (n - t0 < NB) ? n - t0 : NB;
    const int64_t t1 = t0 + nb2;
    const double* P = A + t0 * ld + k0;  // rows t0.. of panel k
    // (1) next panel's columns: C[t0:n, t0:t1] -= P[t0:n] P[t0:t1]^T
    GDML_TRY(launch_gemm_nt_sub(ctx, sm, P, ld, P, ld, A + t0 * ld + t0, ld, n_rows - t0, nb2, nb, 0));
    if (lookahead) {
      HIP_CHECK(ctx, hipEventRecord(evA, sm));
      HIP_CHECK(ctx, hipStreamWaitEvent(sp, evA, 0));
      GDML_TRY(panel_factor(ctx, sp, A, n_rows, ld, t0, nb2));
      HIP_CHECK(ctx, hipEventRecord(evB, sp));
    }
    // (2) rest of the trailing matrix: C[t1:n, t1:n] -= P[t1:n] P[t1:n]^T  (lower)
    if (t1 < n) {
      const double* P1 = A + t1 * ld + k0;
      GDML_TRY(launch_gemm_nt_sub(ctx, sm, P1, ld, P1, ld, A + t1 * ld + t1, ld, n_rows - t1, n - t1, nb, 1));
    }
    if (lookahead)
      HIP_CHECK(ctx, hipStreamWaitEvent(sm, evB, 0));
    else
      GDML_TRY(panel_factor(ctx, sm, A, n_rows, ld, t0, nb2));
  }
  HIP_CHECK(ctx, hipGetLastError());
  int info = 0;
  HIP_CHECK(ctx, hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (info_out) *info_out = info;
  return GDML_OK;
}

// Two-level schedule (round 6, option chol.block = W > 0): the matrix is factored in column blocks of W columns.  A block --
// W columns, ALL rows below its diagonal carried along -- goes through the schedule above (its trailing updates stay inside the
// block's columns), then ONE lower update of depth W brings everything right of the block up to date.  The bulk of the flops
// (1 - 1.5 W / n of them) runs in products of depth W instead of 1024: a C tile is read and written n / W times instead of
// n / 1024 times and the tile turnover that costs the K = 1024 update 14 % of its time shrinks with it (by shape, idle chip, zero
// operands: K = 1024 0.885, K = 4096 0.927, K = 16384 0.934 of the peak: profiles/r06_gemm_shapes.txt).
int chol_factor_device(gdml_ctx* ctx, double* A, int64_t n, int64_t ld, int* info_out, int64_t n_rows) {
  if (n_rows < n) n_rows = n;
  int64_t W = (int64_t)ctx_opt(ctx, "chol.block", 0);
  if (W % 1024 != 0 || W < 2048) W = 0;
  if (W == 0 || n < 3 * W || ctx_opt_i(ctx, "chol.fused_diag", 1) == 0 || ctx_opt_i(ctx, "chol.lookahead", 1) == 0)
    return chol_factor_level(ctx, A, n, ld, info_out, n_rows);
  if (info_out) *info_out = 0;
  for (int64_t c0 = 0; c0 < n;) {
    int64_t w = (n - c0 < W) ? n - c0 : W;
    if (n - (c0 + w) < W / 2) w = n - c0;  // no sliver at the end: the last block takes it
    int inf = 0;
    GDML_TRY(chol_factor_level(ctx, A + c0 * ld + c0, w, ld, &inf, n_rows - c0));
    if (inf != 0) {
      if (info_out) *info_out = (int)(c0 + inf);
      return GDML_OK;
    }
    const int64_t t1 = c0 + w;
    if (t1 < n_rows && t1 < n) {
      const double* P = A + t1 * ld + c0;  // rows below the block, the block's columns
      GDML_TRY(launch_gemm_nt_sub(ctx, ctx->stream, P, ld, P, ld, A + t1 * ld + t1, ld, n_rows - t1, n - t1, w, 1));
    }
    c0 = t1;
  }
  return GDML_OK;
}

// ------------------------------------------------------------------------------------------
// Triangular solves with the factor (cho_solve, analytic.py:97-99), blocked by 64.
// Forward  L z = b : per block J the kernel first solves the 64x64 diagonal system in LDS (every
// workgroup redundantly: 32 KB from L2), then updates its rows below: b[r] -= L[r,J] z_J.
// Backward L^T x = z : per block I (from the bottom) solve L_II^T x_I = z_I, then
// z[c] -= sum_{r in I} L[r,c] x[r] for the columns c left of the block (one thread per column).
// ------------------------------------------------------------------------------------------
// 64x64 triangular solve by ONE wavefront, operands in registers.
//   TRANS = false: L z = b, lane r holds row r of L;    TRANS = true: L^T x = b, lane c holds column c.
// Ls: LDS copy of the block ([64][65], identity-padded), b: this lane's right-hand side entry.
template <bool TRANS>
__device__ __forceinline__ double tri_solve64_wave(const double* Ls, double b, int lane) {
  double v[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) v[k] = TRANS ? Ls[k * 65 + lane] : Ls[lane * 65 + k];
  double sol = 0.0;
  if (!TRANS) {
#pragma unroll
    for (int c = 0; c < 64; ++c) {
      const double t = b / v[c];            // meaningful on lane c
      const double zc = __shfl(t, c, 64);
      if (lane == c) sol = zc;
      b -= v[c] * zc;                       // lanes r > c use it; others ignore their b afterwards
    }
  } else {
#pragma unroll
    for (int r = 63; r >= 0; --r) {
      const double t = b / v[r];            // lane r: v[r] = L[r][r]
      const double xr = __shfl(t, r, 64);
      if (lane == r) sol = xr;
      b -= v[r] * xr;                       // lane c < r: L[r][c] x_r
    }
  }
  return sol;
}

__device__ __forceinline__ void load_block64(const double* __restrict__ L, int64_t ld, int64_t c0,
                                             int w, double* Ls, int tid, int T) {
  for (int e = tid; e < 64 * 64; e += T) {
    int r = e >> 6, c = e & 63;
    double v = (r == c) ? 1.0 : 0.0;
    if (r < w && c < w && c <= r) v = L[(c0 + r) * ld + c0 + c];
    Ls[r * 65 + c] = v;
  }
}

__global__ void __launch_bounds__(256) trsv_fwd_kernel(const double* __restrict__ L, int64_t ld,
                                                       int64_t n, int64_t c0, int w,
                                                       double* __restrict__ b,
                                                       double* __restrict__ z_out) {
  __shared__ __attribute__((aligned(16))) double Ls[64 * 65];
  __shared__ double z[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  load_block64(L, ld, c0, w, Ls, tid, 256);
  __syncthreads();
  if (wave == 0) {
    const double bi = lane < w ? b[c0 + lane] : 0.0;
    const double zi = tri_solve64_wave<false>(Ls, bi, lane);
    z[lane] = zi;
    if (blockIdx.x == 0 && lane < w) z_out[c0 + lane] = zi;
  }
  __syncthreads();
  // rows below: one wave per row, lanes over the w columns
  const int64_t first = c0 + w;
  const double zl = lane < w ? z[lane] : 0.0;
  for (int64_t r = first + (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
    double v = lane < w ? L[r * ld + c0 + lane] * zl : 0.0;
    v = wave_sum(v);
    if (lane == 0) b[r] -= v;
  }
}

__global__ void __launch_bounds__(256) trsv_bwd_kernel(const double* __restrict__ L, int64_t ld,
                                                       int64_t c0, int w, double* __restrict__ b,
                                                       double* __restrict__ x_out) {
  __shared__ __attribute__((aligned(16))) double Ls[64 * 65];
  __shared__ double x[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  load_block64(L, ld, c0, w, Ls, tid, 256);
  __syncthreads();
  if (wave == 0) {
    const double bi = lane < w ? b[c0 + lane] : 0.0;
    const double xi = tri_solve64_wave<true>(Ls, bi, lane);
    x[lane] = xi;
    if (blockIdx.x == 0 && lane < w) x_out[c0 + lane] = xi;
  }
  __syncthreads();
  // columns left of the block: z[c] -= sum_{r in block} L[r,c] x[r]   (one thread per column)
  for (int64_t c = (int64_t)blockIdx.x * 256 + tid; c < c0; c += (int64_t)gridDim.x * 256) {
    double s = 0.0;
    for (int r = 0; r < w; ++r) s += L[(c0 + r) * ld + c] * x[r];
    b[c] -= s;
  }
}

// d_b: right-hand side (destroyed), d_z: scratch (n), d_x: solution (n).  All device vectors.
static int chol_fwd_device(gdml_ctx* ctx, const double* L, int64_t n, int64_t ld, double* d_b, double* d_z) {
  for (int64_t c0 = 0; c0 < n; c0 += 64) {
    int w = (int)((n - c0 < 64) ? n - c0 : 64);
    int64_t rows = n - c0 - w;
    int grid = (int)((rows + 3) / 4);
    if (grid < 1) grid = 1;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(trsv_fwd_kernel, dim3(grid), dim3(256), 0, ctx->stream, L, ld, n, c0, w, d_b,
                       d_z);
    ctx->launch_counter++;
  }
  return GDML_OK;
}

// ------------------------------------------------------------------------------------------
// Backward substitution L^T x = z as ONE persistent launch (default for n >= 2048; option trsv.persist = 0
// restores the per-block launches).  Left-looking: the workgroup that owns 64-block k accumulates
//   s = sum_{c > k} L[c,k]^T x_c      (rows below the block, 512-byte row segments, 4 wavefronts over the blocks c)
// as the x_c become available, then solves the transposed diagonal block and publishes x_k.  Blocks are
// owned cyclically (from the bottom) by one workgroup per CU, all resident at once.  The solution vector itself is the
// availability signal: it is pre-filled with a NaN sentinel, x_k is published with relaxed agent-scope 8-byte stores and
// the consumers poll the elements they need (x crosses
// XCDs, i.e. L2s).  Blocks are dealt round-robin, so a workgroup also waits for blocks of workgroups with a HIGHER
// index (workgroup 0 at round 2 needs the last workgroup's block of round 1): all min(nbk, CUs) workgroups must
// be resident at once.  One workgroup per CU on an otherwise idle stream satisfies that; where it does not (GPU
// shared with another process) the bounded spins give up, `err` is set and the host falls back to the
// per-block launches below.
// ------------------------------------------------------------------------------------------
// Sentinel of "not yet solved" in the solution vector of the persistent backward substitution: a quiet NaN with a
// payload no arithmetic produces (hardware NaNs are 0x7FF8000000000000 / 0xFFF8...).
#define TRSV_PENDING 0x7FF8A5A5C3C30001ull

__global__ void __launch_bounds__(256) trsv_fill_pending_kernel(unsigned long long* __restrict__ x, int64_t n) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n) x[t] = TRSV_PENDING;
}

__global__ void __launch_bounds__(256) trsv_bwd_persist_kernel(const double* __restrict__ L, int64_t ld,
                                                               int64_t n, int nbk, const double* __restrict__ z,
                                                               double* x, int* err) {
  __shared__ __attribute__((aligned(16))) double Ls[64 * 65];
  __shared__ double part[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned long long* xb = reinterpret_cast<unsigned long long*>(x);
  for (int kk = blockIdx.x; kk < nbk; kk += gridDim.x) {  // kk counts blocks from the bottom
    const int k = nbk - 1 - kk;
    const int64_t c0 = (int64_t)k * 64;
    const int wk = (int)((n - c0 < 64) ? n - c0 : 64);
    load_block64(L, ld, c0, wk, Ls, tid, 256);  // diagonal block (independent of x): overlaps the waiting
    __syncthreads();
    const double rdiag = 1.0 / Ls[lane * 65 + lane];  // reciprocal pivots off the critical path (identity on padding)
    double acc = 0.0;
    for (int cc = wv; cc < kk; cc += 4) {  // blocks below, bottom first; this wavefront takes every 4th
      const int c = nbk - 1 - cc;
      const int64_t r0 = (int64_t)c * 64;
      const int rows = (int)((n - r0 < 64) ? n - r0 : 64);
      // the L block does not depend on x: its 64 row segments are in flight while the wavefront waits for x_c
      const double* Lp = L + r0 * ld + c0 + (lane < wk ? lane : 0);
      double lreg[64];
#pragma unroll
      for (int i = 0; i < 64; ++i) lreg[i] = (i < rows) ? Lp[(int64_t)i * ld] : 0.0;
      // x_c is published element by element (relaxed agent-scope stores of the 8-byte values over the sentinel): the
      // consumer polls the data itself -- one memory round trip, no separate flag, no fences
      unsigned long long xv = 0;
      int spins = 0;
      for (;;) {
        xv = (lane < rows) ? __hip_atomic_load(xb + r0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        if (!__any(xv == TRSV_PENDING)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 26)) {  // ~1 min: only reachable if the GPU is shared with another process for that long
          if (lane == 0) atomicExch(err, 1);
          xv = 0;
          break;
        }
      }
      const double xd = __longlong_as_double((long long)xv);
#pragma unroll
      for (int i = 0; i < 64; ++i) acc += lreg[i] * __shfl(xd, i, 64);
    }
    part[wv][lane] = (lane < wk) ? acc : 0.0;
    __syncthreads();
    if (wv == 0) {
      const double s = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
      double bi = lane < wk ? z[c0 + lane] - s : 0.0;
      // transposed 64 x 64 solve, row r of L broadcast from lane r: x_r = b_r / L_rr; b_c -= L_rc x_r (c < r)
      double sol = 0.0;
#pragma unroll
      for (int r = 63; r >= 0; --r) {
        const double xr = __shfl(bi * rdiag, r, 64);
        if (lane == r) sol = xr;
        bi -= Ls[r * 65 + lane] * xr;
      }
      if (lane < wk)
        __hip_atomic_store(xb + c0 + lane, (unsigned long long)__double_as_longlong(sol), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
}

// d_z is destroyed
int chol_bwd_device(gdml_ctx* ctx, const double* L, int64_t n, int64_t ld, double* d_z, double* d_x) {
  const int persist = ctx_opt_i(ctx, "trsv.persist", 1);
  if (persist && n >= 2048) {
    const int nbk = (int)((n + 63) / 64);
    int* err = ctx->d_info + 5;
    HIP_CHECK(ctx, hipMemsetAsync(err, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(trsv_fill_pending_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<unsigned long long*>(d_x), n);
    const int grid = nbk < ctx->num_cus ? nbk : ctx->num_cus;  // one workgroup per CU, all resident
    hipLaunchKernelGGL(trsv_bwd_persist_kernel, dim3(grid), dim3(256), 0, ctx->stream, L, ld, n, nbk, d_z, d_x, err);
    ctx->launch_counter++;
    int h_err = 0;
    HIP_CHECK(ctx, hipMemcpyAsync(&h_err, err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h_err == 0) return GDML_OK;
    // a workgroup gave up waiting (the persistent launch needs all its workgroups co-resident: not guaranteed when
    // the GPU is shared): d_z was only read so far, redo the substitution with one launch per block
  }
  int64_t last = ((n - 1) / 64) * 64;
  for (int64_t c0 = last; c0 >= 0; c0 -= 64) {
    int w = (int)((n - c0 < 64) ? n - c0 : 64);
    int grid = (int)((c0 + 255) / 256);
    if (grid < 1) grid = 1;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(trsv_bwd_kernel, dim3(grid), dim3(256), 0, ctx->stream, L, ld, c0, w, d_z,
                       d_x);
    ctx->launch_counter++;
  }
  HIP_CHECK(ctx, hipGetLastError());
  return GDML_OK;
}

int chol_solve_device(gdml_ctx* ctx, const double* L, int64_t n, int64_t ld, double* d_b,
                      double* d_z, double* d_x) {
  GDML_TRY(chol_fwd_device(ctx, L, n, ld, d_b, d_z));
  return chol_bwd_device(ctx, L, n, ld, d_z, d_x);
}

// ------------------------------------------------------------------------------------------
extern "C" int gdml_chol_set_rhs(gdml_ctx* ctx, const double* y, int64_t n) {
  if (!ctx || !y) return GDML_ERR_INVALID;
  if (!ctx->K || ctx->K_rows != ctx->K_cols)
    return gdml_fail(ctx, GDML_ERR_STATE, "gdml_chol_set_rhs: assemble a square K first");
  if (ctx->K_factored) return gdml_fail(ctx, GDML_ERR_STATE, "gdml_chol_set_rhs: K is already factored");
  if (ctx->K_extra < 1)
    return gdml_fail(ctx, GDML_ERR_STATE, "gdml_chol_set_rhs: K was assembled without an extra row");
  if (n != ctx->K_rows) return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_chol_set_rhs: n mismatch");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (ctx->d_rhs) GDML_TRY(ctx_free(ctx, ctx->d_rhs));
  ctx->d_rhs = nullptr;
  GDML_TRY(ctx_alloc(ctx, (void**)&ctx->d_rhs, n * 8));
  HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_rhs, y, n * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_CHECK(ctx, hipMemcpyAsync(ctx->K + n * ctx->K_ld, ctx->d_rhs, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // y is a pageable host array owned by the caller
  ctx->K_rhs_row = true;
  return GDML_OK;
}

extern "C" int gdml_chol_factor(gdml_ctx* ctx, double lam, int* info) {
  if (!ctx) return GDML_ERR_INVALID;
  if (!ctx->K) return gdml_fail(ctx, GDML_ERR_STATE, "gdml_chol_factor: assemble K first");
  if (ctx->K_rows != ctx->K_cols)
    return gdml_fail(ctx, GDML_ERR_STATE, "gdml_chol_factor: resident K is %lld x %lld, not square",
                     (long long)ctx->K_rows, (long long)ctx->K_cols);
  if (ctx->K_factored) return gdml_fail(ctx, GDML_ERR_STATE, "gdml_chol_factor: already factored");
  if (ctx->K_destroyed)
    return gdml_fail(ctx, GDML_ERR_STATE, "gdml_chol_factor: the resident matrix was consumed by a failed factorisation: assemble again");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int64_t n = ctx->K_rows;
  if (ctx->K_is_A && lam != ctx->K_lam)
    return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_chol_factor: lam (%g) differs from the one gdml_assemble_A used (%g)",
                     lam, ctx->K_lam);
  phase_begin(ctx);
  if (!ctx->K_is_A) {  // un-negated K from gdml_assemble_K: A = -K + lam I on the lower triangle (analytic.py:65,82)
    hipLaunchKernelGGL(negate_shift_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, ctx->K, n,
                       ctx->K_ld, lam);
    ctx->launch_counter++;
  }
  int inf = 0;
  GDML_TRY(chol_factor_device(ctx, ctx->K, n, ctx->K_ld, &inf, ctx->K_rhs_row ? n + 1 : n));
  GDML_TRY(phase_end(ctx, "factor"));
  if (info) *info = inf;
  ctx->K_lam = lam;
  if (inf != 0) {
    ctx->K_factored = false;
    ctx->K_destroyed = true;
    return gdml_fail(ctx, GDML_ERR_NOT_PD,
                     "%d-th leading minor of the array is not positive definite", inf);
  }
  ctx->K_factored = true;
  return GDML_OK;
}

__global__ void __launch_bounds__(256) scale_kernel(double* __restrict__ x, int64_t n, double s) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) x[t] *= s;
}
__global__ void __launch_bounds__(256) axpy_kernel(double* __restrict__ y, const double* __restrict__ x,
                                                   int64_t n, double a) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) y[t] += a * x[t];
}
// r = y - (-(Kx - lam x))... helper: r = y + kv  where kv = K x - lam x  (A x = -(K x - lam x))
__global__ void __launch_bounds__(256) resid_kernel(const double* __restrict__ y,
                                                    const double* __restrict__ kv, int64_t n,
                                                    double* __restrict__ r) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) r[t] = y[t] + kv[t];
}

int operator_model_from_trainset(gdml_ctx* ctx, double sig);

extern "C" int gdml_chol_solve(gdml_ctx* ctx, const double* y, int64_t n, int n_refine,
                               double* alphas_out) {
  if (!ctx || !alphas_out) return GDML_ERR_INVALID;
  if (!ctx->K || !ctx->K_factored)
    return gdml_fail(ctx, GDML_ERR_STATE, "gdml_chol_solve: no Cholesky factor resident");
  if (!y && !ctx->K_rhs_row)
    return gdml_fail(ctx, GDML_ERR_STATE, "gdml_chol_solve(y = NULL) needs gdml_chol_set_rhs before the factorisation");
  if (n != ctx->K_rows) return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_chol_solve: n mismatch");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  void* buf = nullptr;
  GDML_TRY(ctx_alloc(ctx, &buf, 6 * n * 8));
  double* dx = (double*)buf;   // solution
  double* dy = dx + n;         // right-hand side (kept)
  double* db = dy + n;         // work copy of a right-hand side (destroyed by the solve)
  double* dz = db + n;         // forward-substitution result
  double* dr = dz + n;         // refinement correction
  double* dkv = dr + n;        // K x - lam x
  int rc = GDML_OK;
  hipError_t e;
  if (y) {
    e = hipMemcpyAsync(dy, y, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(db, dy, n * 8, hipMemcpyDeviceToDevice, ctx->stream);
  } else {  // forward substitution already done by the factorisation: z is row n of the factor buffer
    e = hipMemcpyAsync(dy, ctx->d_rhs, n * 8, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(dz, ctx->K + n * ctx->K_ld, n * 8, hipMemcpyDeviceToDevice, ctx->stream);
  }
  if (e != hipSuccess) rc = gdml_fail(ctx, GDML_ERR_HIP, "copy: %s", hipGetErrorString(e));
  if (rc == GDML_OK) {
    phase_begin(ctx);
    rc = y ? chol_solve_device(ctx, ctx->K, n, ctx->K_ld, db, dz, dx)
           : chol_bwd_device(ctx, ctx->K, n, ctx->K_ld, dz, dx);
  }
  if (rc == GDML_OK && n_refine > 0) rc = operator_model_from_trainset(ctx, ctx->K_sig);
  for (int it = 0; rc == GDML_OK && it < n_refine; ++it) {
    // r = y - A x,  A x = -(K x - lam x)  =>  r = y + (K x - lam x)
    rc = matvec_device(ctx, ctx->K_lam, ctx->K_use_E, dx, n, dkv);
    if (rc != GDML_OK) break;
    hipLaunchKernelGGL(resid_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, dy, dkv, n, db);
    rc = chol_solve_device(ctx, ctx->K, n, ctx->K_ld, db, dz, dr);
    if (rc != GDML_OK) break;
    hipLaunchKernelGGL(axpy_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, dx, dr, n, 1.0);
  }
  if (rc == GDML_OK) {
    hipLaunchKernelGGL(scale_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, dx, n, -1.0);
    rc = phase_end(ctx, "solve");
  }
  if (rc == GDML_OK) {
    e = hipMemcpyAsync(alphas_out, dx, n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = gdml_fail(ctx, GDML_ERR_HIP, "D2H: %s", hipGetErrorString(e));
  }
  int rc2 = ctx_free(ctx, buf);
  return rc != GDML_OK ? rc : rc2;
}
