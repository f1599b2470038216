// Context, memory, phase timers, raw buffer helpers.
#include <stdarg.h>

#include <mutex>

#include "common.h"

static std::string g_create_err;

int gdml_fail(gdml_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx)
    ctx->err = buf;
  else
    g_create_err = buf;
  return code;
}

extern "C" int gdml_abi_version(void) { return 4; }

// ---- options --------------------------------------------------------------------------------
// Every tuning / ablation switch of the library is a (key, value) pair of the context, read at the
// point of use (no process-wide read-once caches).  Keys are listed in include/gdml_hip.h.  The only
// environment variable the library looks at is GDML_OPTIONS="key=value,key=value", applied once when
// a context is created (lab convenience for the probes under tools/).
static const char* kKnownOptions[] = {
    "asm.wave", "asm.j_chunk", "asm.lower", "asm.strip", "asm.i_chunk",
    "asm.pts", "asm.pts_nv", "asm.pts_nt", "asm.pts_xcd", "asm.pts_i_chunk", "asm.pts_debug", "asm.perm_debug", "asm.perm_w", "asm.perm_lds_kb", "asm.perm_level", "asm.perm_nimg", "asm.perm_pg", "asm.perm_na", "asm.perm_fast_store", "asm.perm_i_chunk", "asm.perm_compact", "asm.perm_lds_rows", "asm.big1", "asm.perm2", "asm.perm2_min_n", "asm.perm2_min_p", "asm.perm2_split", "asm.perm2_post", "asm.perm2_ed", "asm.perm2_es", "asm.perm2_direct", "asm.perm2_chunk", "asm.perm2_i_chunk", "asm.perm2_debug",
    "gemm.debug", "gemm.trace", "gemm.persist", "gemm.n64", "nys.trsm_left", "nys.syrk_split", "gemm.fill_tiles", "predict.wide_pad", "predict.tn_fill", "gemm.nt_c", "gemm.lds16", "chol.nb", "chol.block", "chol.block_f", "chol.small_update", "pcg.gemv_plain", "chol.lookahead", "chol.panel_fused", "chol.panel_kernel", "chol.fused_diag",
    "chol.fused_min_rows", "chol.outer", "chol.outer_min_rows", "chol.merge_gemm1", "chol.tail_lookahead", "trsm.debug",
    "trsv.persist", "predict.wave_only", "predict.mfma", "predict.fill", "predict.mfma_wide", "predict.fused", "predict.fused_rows", "predict.fused_spin",
    "lu.nb", "comm.force_collectives", "nys.force_qr", "nys.force_fail", "dist.nb", "dist.lookahead", "dist.force_panels", "pcg.depth", "pcg.precon_form", "pcg.f32_rows_per", "pcg.f32_rw", "pcg.f32_min_pivot", "pcg.f32_last_min_pivot", "pcg.f32_gram_rows", "pcg.f32_inplace"};

double ctx_opt(const gdml_ctx* ctx, const char* key, double dflt) {
  auto it = ctx->opts.find(key);
  return it == ctx->opts.end() ? dflt : it->second;
}

extern "C" int gdml_set_option(gdml_ctx* ctx, const char* key, double value) {
  if (!ctx || !key) return GDML_ERR_INVALID;
  for (const char* k : kKnownOptions)
    if (strcmp(k, key) == 0) {
      ctx->opts[key] = value;
      return GDML_OK;
    }
  return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_set_option: unknown option '%s'", key);
}

extern "C" int gdml_get_option(gdml_ctx* ctx, const char* key, double* value_out, int* is_set_out) {
  if (!ctx || !key) return GDML_ERR_INVALID;
  auto it = ctx->opts.find(key);
  if (value_out) *value_out = it == ctx->opts.end() ? 0.0 : it->second;
  if (is_set_out) *is_set_out = it != ctx->opts.end();
  return GDML_OK;
}

static void options_from_env(gdml_ctx* ctx) {
  const char* e = getenv("GDML_OPTIONS");
  if (!e) return;
  std::string s(e);
  size_t pos = 0;
  while (pos < s.size()) {
    size_t end = s.find(',', pos);
    if (end == std::string::npos) end = s.size();
    const std::string kv = s.substr(pos, end - pos);
    const size_t eq = kv.find('=');
    if (eq != std::string::npos) (void)gdml_set_option(ctx, kv.substr(0, eq).c_str(), atof(kv.c_str() + eq + 1));
    pos = end + 1;
  }
}

// ---- process-level device arena (gdml_mem_reserve) ------------------------------------------------------------------
// hipMalloc of the large kernel matrices costs seconds on this driver once a request passes what the runtime has at hand
// (3.5-6.5 s for 128-200 GB, erratic below: profiles/r03_malloc_probe.txt, r04_malloc_probe.txt -- hipMallocAsync pools and
// hipMemCreate / hipMemMap pay the same per byte), and hipFree gives it back.  A long-lived process therefore reserves
// ONE block per device once and keeps it: the large buffers of every context on the device (the kernel matrix / the
// Nystroem matrix, the fp32 copy of the preconditioner factor, its m x m companions) are carved from it first-fit, a
// buffer handed back is not freed, and the block survives contexts.  A request that fits no gap goes to hipMalloc.
// (Round 4 had one tenant at a time; the idle remainder could not be handed out, which the memory models had to know.)
struct ArenaBlock {
  int64_t off, bytes;
  gdml_ctx* owner;
};
struct DeviceArena {
  void* base = nullptr;
  int64_t bytes = 0;
  std::vector<ArenaBlock> blocks;  // sorted by offset
};
static DeviceArena g_arena[64];
// Contexts of one process live on several threads (a solve in a worker thread, a Python __del__ closing a context from
// another): every read or write of an arena entry happens under this lock.  It is never held across a kernel launch;
// hipMalloc / hipFree of the block itself run under it (seconds, once per process).
static std::mutex g_arena_mu;
static const int64_t kArenaMinRequest = (int64_t)1 << 30;  // small work buffers stay with hipMalloc
static const int64_t kArenaAlign = (int64_t)2 << 20;

static DeviceArena* arena_of(const gdml_ctx* ctx) { return (ctx->device >= 0 && ctx->device < 64) ? &g_arena[ctx->device] : nullptr; }

// first-fit carve; caller holds g_arena_mu.  Returns nullptr when no gap is large enough.
static void* arena_carve(DeviceArena* a, gdml_ctx* ctx, int64_t bytes) {
  const int64_t need = (bytes + kArenaAlign - 1) / kArenaAlign * kArenaAlign;
  int64_t off = 0;
  size_t pos = 0;
  for (; pos <= a->blocks.size(); ++pos) {
    const int64_t end = pos < a->blocks.size() ? a->blocks[pos].off : a->bytes;
    if (end - off >= need) break;
    if (pos < a->blocks.size()) off = a->blocks[pos].off + a->blocks[pos].bytes;
  }
  if (pos > a->blocks.size()) return nullptr;
  a->blocks.insert(a->blocks.begin() + (long)pos, ArenaBlock{off, need, ctx});
  return (char*)a->base + off;
}
static bool arena_owns(const DeviceArena* a, const void* p) {
  return a && a->base && (const char*)p >= (const char*)a->base && (const char*)p < (const char*)a->base + a->bytes;
}
static void arena_release(DeviceArena* a, const void* p) {  // caller holds g_arena_mu
  const int64_t off = (const char*)p - (const char*)a->base;
  for (size_t i = 0; i < a->blocks.size(); ++i)
    if (a->blocks[i].off == off) {
      a->blocks.erase(a->blocks.begin() + (long)i);
      return;
    }
}
static void arena_release_owner(gdml_ctx* ctx) {  // caller holds g_arena_mu
  DeviceArena* a = arena_of(ctx);
  if (!a) return;
  for (size_t i = a->blocks.size(); i-- > 0;)
    if (a->blocks[i].owner == ctx) a->blocks.erase(a->blocks.begin() + (long)i);
}
static int64_t arena_largest_gap(const DeviceArena* a) {  // caller holds g_arena_mu; blocks are sorted by offset
  int64_t best = 0, off = 0;
  for (const ArenaBlock& b : a->blocks) {
    if (b.off - off > best) best = b.off - off;
    off = b.off + b.bytes;
  }
  if (a->bytes - off > best) best = a->bytes - off;
  return best;
}

extern "C" int gdml_device_count(int* n_out) {
  if (!n_out) return GDML_ERR_INVALID;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *n_out = 0;
    return gdml_fail(nullptr, GDML_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *n_out = n;
  return GDML_OK;
}

extern "C" int gdml_device_pci_bus_id(int device, char* out, int len) {
  if (!out || len < 16) return GDML_ERR_INVALID;
  out[0] = 0;
  hipError_t e = hipDeviceGetPCIBusId(out, len, device);
  if (e != hipSuccess) return gdml_fail(nullptr, GDML_ERR_HIP, "hipDeviceGetPCIBusId(%d): %s", device, hipGetErrorString(e));
  return GDML_OK;
}

extern "C" int gdml_ctx_create(int device, gdml_ctx** ctx_out) {
  if (!ctx_out) return GDML_ERR_INVALID;
  *ctx_out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    return gdml_fail(nullptr, GDML_ERR_HIP, "no HIP device visible (%s)",
                     e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  if (device < 0 || device >= n)
    return gdml_fail(nullptr, GDML_ERR_INVALID, "device %d out of range [0,%d)", device, n);
  gdml_ctx* ctx = new gdml_ctx();
  ctx->device = device;
  // Two streams: `stream` carries the bulk kernels, `stream2` (highest priority) the latency-bound panel
  // steps of the look-ahead Cholesky.
  auto make_streams = [&]() -> hipError_t {
    hipError_t err;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if ((err = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return err;
    return hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, hi);
  };
  {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0)
      ctx->num_cus = ncu;
  }
  if ((e = hipSetDevice(device)) != hipSuccess || (e = make_streams()) != hipSuccess ||
      (e = hipEventCreate(&ctx->ev0)) != hipSuccess ||
      (e = hipEventCreate(&ctx->ev1)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&ctx->ev_la[0], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&ctx->ev_la[1], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&ctx->ev_la[2], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&ctx->ev_la[3], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&ctx->ev_la[4], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&ctx->ev_la[5], hipEventDisableTiming)) != hipSuccess ||
      (e = hipMalloc((void**)&ctx->d_info, 64)) != hipSuccess ||
      (e = hipMalloc((void**)&ctx->gemm_queue, (size_t)ctx->gemm_queue_sets * 512)) != hipSuccess ||
      (e = hipMemset(ctx->gemm_queue, 0, (size_t)ctx->gemm_queue_sets * 512)) != hipSuccess) {
    gdml_fail(nullptr, GDML_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
    delete ctx;
    return GDML_ERR_HIP;
  }
  options_from_env(ctx);
  *ctx_out = ctx;
  return GDML_OK;
}

extern "C" int gdml_ctx_destroy(gdml_ctx* ctx) {
  if (!ctx) return GDML_OK;
  hipSetDevice(ctx->device);
  hipDeviceSynchronize();
  comm_destroy(ctx);
  for (auto& t : ctx->pending) {
    hipEventDestroy(t.e0);
    hipEventDestroy(t.e1);
  }
  for (auto e : ctx->event_pool) hipEventDestroy(e);
  {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    DeviceArena* a = arena_of(ctx);
    for (auto& kv : ctx->allocs)
      if (!arena_owns(a, kv.first)) hipFree(kv.first);
    ctx->allocs.clear();
    arena_release_owner(ctx);  // the block itself stays with the process
  }
  if (ctx->d_info) hipFree(ctx->d_info);
  if (ctx->gemm_queue) hipFree(ctx->gemm_queue);
  if (ctx->ev0) hipEventDestroy(ctx->ev0);
  if (ctx->ev1) hipEventDestroy(ctx->ev1);
  for (hipEvent_t ev : ctx->ev_la)
    if (ev) hipEventDestroy(ev);
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  if (ctx->stream2) hipStreamDestroy(ctx->stream2);
  if (ctx->h_pin) hipHostFree(ctx->h_pin);
  if (ctx->h_map) hipHostFree(ctx->h_map);
  if (ctx->h_coll) hipHostFree(ctx->h_coll);
  delete ctx;
  return GDML_OK;
}

extern "C" const char* gdml_last_error(const gdml_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

extern "C" int gdml_sync(gdml_ctx* ctx) {
  if (!ctx) return GDML_ERR_INVALID;
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream2));
  return GDML_OK;
}

extern "C" int gdml_mem_info(gdml_ctx* ctx, int64_t* held, int64_t* free_b, int64_t* total_b) {
  if (!ctx) return GDML_ERR_INVALID;
  size_t f = 0, t = 0;
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  HIP_CHECK(ctx, hipMemGetInfo(&f, &t));
  if (held) *held = ctx->held;
  // what large buffers of this context could get: driver-free memory plus the LARGEST contiguous gap of the process arena
  // (a carve needs one gap: the sum of the idle pieces -- the round-5 figure -- promised memory that a fragmented arena
  // cannot hand out; the context's own resident matrix is re-used in place by the next assembly: callers add
  // resident_K_bytes themselves)
  {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    const DeviceArena* a = arena_of(ctx);
    if (a && a->base) f += (size_t)arena_largest_gap(a);
  }
  if (free_b) *free_b = (int64_t)f;
  if (total_b) *total_b = (int64_t)t;
  return GDML_OK;
}

extern "C" int gdml_mem_reserve(gdml_ctx* ctx, int64_t bytes, int64_t* reserved_out) {
  if (!ctx) return GDML_ERR_INVALID;
  DeviceArena* a = arena_of(ctx);
  if (!a) return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_mem_reserve: device index out of range");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> lk(g_arena_mu);
  if (bytes != a->bytes || bytes == 0) {
    if (!a->blocks.empty()) return gdml_fail(ctx, GDML_ERR_STATE, "gdml_mem_reserve: the arena is in use (a context holds a buffer carved from it)");
    if (a->base) {
      HIP_CHECK(ctx, hipFree(a->base));
      a->base = nullptr;
      a->bytes = 0;
    }
    if (bytes > 0) {
      void* p = nullptr;
      hipError_t e = hipMalloc(&p, (size_t)bytes);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        return gdml_fail(ctx, GDML_ERR_OOM, "gdml_mem_reserve: hipMalloc(%lld bytes) failed: %s", (long long)bytes, hipGetErrorString(e));
      }
      a->base = p;
      a->bytes = bytes;
    }
  }
  if (reserved_out) *reserved_out = a->bytes;
  return GDML_OK;
}

int ctx_alloc(gdml_ctx* ctx, void** p, int64_t bytes) {
  *p = nullptr;
  if (bytes <= 0) bytes = 8;
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  DeviceArena* a = arena_of(ctx);
  {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    if (a && a->base && bytes >= kArenaMinRequest && bytes <= a->bytes) {
      void* q = arena_carve(a, ctx, bytes);
      if (q) {
        *p = q;
        ctx->allocs[*p] = bytes;
        ctx->held += bytes;
        return GDML_OK;
      }
    }
  }
  hipError_t e = hipMalloc(p, (size_t)bytes);
  if (e == hipErrorOutOfMemory && a) {
    // an idle arena must not be the reason a larger request fails: give it back and try once more
    std::lock_guard<std::mutex> lk(g_arena_mu);
    if (a->base && a->blocks.empty()) {
      (void)hipGetLastError();
      (void)hipFree(a->base);
      a->base = nullptr;
      a->bytes = 0;
      e = hipMalloc(p, (size_t)bytes);
    }
  }
  if (e != hipSuccess) {
    *p = nullptr;
    (void)hipGetLastError();
    return gdml_fail(ctx, GDML_ERR_OOM, "hipMalloc(%lld bytes) failed: %s", (long long)bytes,
                     hipGetErrorString(e));
  }
  ctx->allocs[*p] = bytes;
  ctx->held += bytes;
  return GDML_OK;
}

int ctx_free(gdml_ctx* ctx, void* p) {
  if (!p) return GDML_OK;
  auto it = ctx->allocs.find(p);
  if (it == ctx->allocs.end()) return gdml_fail(ctx, GDML_ERR_INVALID, "free of unknown pointer");
  HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->held -= it->second;
  ctx->allocs.erase(it);
  {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    DeviceArena* a = arena_of(ctx);
    if (arena_owns(a, p)) {  // back to the arena, not to the driver
      arena_release(a, p);
      return GDML_OK;
    }
  }
  HIP_CHECK(ctx, hipFree(p));
  return GDML_OK;
}

int ctx_scratch(gdml_ctx* ctx, int64_t bytes, double** out) {
  if (bytes > ctx->scratch_bytes) {
    if (ctx->scratch) GDML_TRY(ctx_free(ctx, ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    int64_t want = bytes + bytes / 4;
    GDML_TRY(ctx_alloc(ctx, (void**)&ctx->scratch, want));
    ctx->scratch_bytes = want;
  }
  *out = ctx->scratch;
  return GDML_OK;
}

// Cached work buffer that only grows (avoids hipMalloc/hipFree in hot paths).
int ctx_slot(gdml_ctx* ctx, int slot, int64_t bytes, double** out) {
  if (bytes > ctx->slot_bytes[slot]) {
    if (ctx->slot[slot]) GDML_TRY(ctx_free(ctx, ctx->slot[slot]));
    ctx->slot[slot] = nullptr;
    ctx->slot_bytes[slot] = 0;
    int64_t want = bytes + bytes / 8;
    GDML_TRY(ctx_alloc(ctx, (void**)&ctx->slot[slot], want));
    ctx->slot_bytes[slot] = want;
  }
  *out = ctx->slot[slot];
  return GDML_OK;
}

// Phase timers are read lazily: phase_end only records the end event, so a call that returns device-side
// results (or the single-geometry prediction path) does not pay a host synchronisation for the timer.
int phase_resolve(gdml_ctx* ctx) {
  if (ctx->phase_pending.empty()) return GDML_OK;
  HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  PhaseStat& s = ctx->phases[ctx->phase_pending];
  s.ms = ms;
  s.launches = ctx->phase_pending_launches;
  ctx->phase_pending.clear();
  return GDML_OK;
}

void phase_begin(gdml_ctx* ctx) {
  (void)phase_resolve(ctx);
  ctx->launch_counter = 0;
  (void)hipEventRecord(ctx->ev0, ctx->stream);
}

int phase_end(gdml_ctx* ctx, const char* name) {
  HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ctx->phase_pending = name;
  ctx->phase_pending_launches = ctx->launch_counter;
  return GDML_OK;
}

static hipEvent_t pool_get(gdml_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

int ktime_begin(gdml_ctx* ctx) {
  if (!ctx->profiling) return -1;
  PendingTiming t;
  t.e0 = pool_get(ctx);
  t.e1 = pool_get(ctx);
  t.work = 0;
  (void)hipEventRecord(t.e0, ctx->kt_stream ? ctx->kt_stream : ctx->stream);
  ctx->pending.push_back(t);
  return (int)ctx->pending.size() - 1;
}

void ktime_end(gdml_ctx* ctx, int slot, const char* name, double work) {
  if (slot < 0) return;
  PendingTiming& t = ctx->pending[slot];
  t.name = name;
  t.work = work;
  (void)hipEventRecord(t.e1, ctx->kt_stream ? ctx->kt_stream : ctx->stream);
}

int ktime_collect(gdml_ctx* ctx) {
  if (ctx->pending.empty()) return GDML_OK;
  HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& t : ctx->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) {
      KernelStat& k = ctx->kstats[t.name];
      k.ms += ms;
      k.launches += 1;
      k.work += t.work;
    }
    ctx->event_pool.push_back(t.e0);
    ctx->event_pool.push_back(t.e1);
  }
  ctx->pending.clear();
  return GDML_OK;
}

extern "C" int gdml_profile(gdml_ctx* ctx, int enable) {
  if (!ctx) return GDML_ERR_INVALID;
  GDML_TRY(ktime_collect(ctx));
  ctx->profiling = enable != 0;
  ctx->kstats.clear();
  return GDML_OK;
}

extern "C" int gdml_kernel_stat(gdml_ctx* ctx, const char* kernel, double* ms_out,
                                int64_t* launches_out, double* work_out) {
  if (!ctx || !kernel) return GDML_ERR_INVALID;
  GDML_TRY(ktime_collect(ctx));
  auto it = ctx->kstats.find(kernel);
  KernelStat k;
  if (it != ctx->kstats.end()) k = it->second;
  if (ms_out) *ms_out = k.ms;
  if (launches_out) *launches_out = k.launches;
  if (work_out) *work_out = k.work;
  return GDML_OK;
}

extern "C" int gdml_phase_ms(gdml_ctx* ctx, const char* phase, double* ms_out,
                             int64_t* launches_out) {
  if (!ctx || !phase) return GDML_ERR_INVALID;
  GDML_TRY(phase_resolve(ctx));
  auto it = ctx->phases.find(phase);
  if (it == ctx->phases.end()) return gdml_fail(ctx, GDML_ERR_STATE, "phase '%s' never ran", phase);
  if (ms_out) *ms_out = it->second.ms;
  if (launches_out) *launches_out = it->second.launches;
  return GDML_OK;
}

extern "C" int gdml_dev_alloc(gdml_ctx* ctx, int64_t bytes, void** dev_out) {
  if (!ctx || !dev_out) return GDML_ERR_INVALID;
  return ctx_alloc(ctx, dev_out, bytes);
}
extern "C" int gdml_dev_free(gdml_ctx* ctx, void* dev) {
  if (!ctx) return GDML_ERR_INVALID;
  return ctx_free(ctx, dev);
}
extern "C" int gdml_memcpy_h2d(gdml_ctx* ctx, void* dev, const void* host, int64_t bytes) {
  if (!ctx) return GDML_ERR_INVALID;
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  HIP_CHECK(ctx, hipMemcpyAsync(dev, host, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return GDML_OK;
}
extern "C" int gdml_memcpy_d2h(gdml_ctx* ctx, void* host, const void* dev, int64_t bytes) {
  if (!ctx) return GDML_ERR_INVALID;
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  HIP_CHECK(ctx, hipMemcpyAsync(host, dev, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return GDML_OK;
}

extern "C" int gdml_K_dev(gdml_ctx* ctx, double** K_dev_out, int64_t* ld_out) {
  if (!ctx) return GDML_ERR_INVALID;
  if (!ctx->K) return gdml_fail(ctx, GDML_ERR_STATE, "no kernel matrix resident");
  if (K_dev_out) *K_dev_out = ctx->K;
  if (ld_out) *ld_out = ctx->K_ld;
  return GDML_OK;
}

extern "C" int gdml_K_shape(gdml_ctx* ctx, int64_t* n_rows, int64_t* n_cols, int64_t* extra) {
  if (!ctx) return GDML_ERR_INVALID;
  if (!ctx->K) return gdml_fail(ctx, GDML_ERR_STATE, "no kernel matrix resident");
  if (n_rows) *n_rows = ctx->K_rows;
  if (n_cols) *n_cols = ctx->K_cols;
  if (extra) *extra = ctx->K_extra;
  return GDML_OK;
}
