// RCCL plumbing for the sharded iterative solver (new relative to the reference, which has no
// collective at all: SURVEY.md section 2a).  librccl is bound at run time with dlopen/dlsym so that
// the library has no link-time dependency on it and shares the copy another component of the
// process (e.g. torch.distributed's "nccl" backend, which IS RCCL on ROCm) may already have loaded.
//
// Sharding model (one process per GPU): training points are split into W contiguous shards of
// pts_per = ceil(M / W) points.  Vectors of length n = 3N M are replicated on every rank in buffers
// padded to W * pts_per * 3N doubles so that all-gathers are in place with equal chunk sizes.
#include <dlfcn.h>

#include "common.h"

typedef int ncclResult_t_;
typedef struct { char internal[128]; } ncclUniqueId_;
typedef void* ncclComm_t_;

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t_ (*GetUniqueId)(ncclUniqueId_*) = nullptr;
  ncclResult_t_ (*CommInitRank)(ncclComm_t_*, int, ncclUniqueId_, int) = nullptr;
  ncclResult_t_ (*CommDestroy)(ncclComm_t_) = nullptr;
  ncclResult_t_ (*CommAbort)(ncclComm_t_) = nullptr;
  ncclResult_t_ (*CommSplit)(ncclComm_t_, int, int, ncclComm_t_*, void*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t_) = nullptr;
  ncclResult_t_ (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
  ncclResult_t_ (*AllGather)(const void*, void*, size_t, int, ncclComm_t_, hipStream_t) = nullptr;
  ncclResult_t_ (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
  ncclResult_t_ (*CommCount)(ncclComm_t_, int*) = nullptr;  // optional
};
static RcclApi g_rccl;
static const int kNcclDouble = 8, kNcclSum = 0;

static bool rccl_load(std::string* err) {
  if (g_rccl.handle) return true;
  // A copy already mapped into the process wins (a caller that brought PyTorch's bundled RCCL keeps ONE RCCL); otherwise the
  // ROCm installation's own library is asked for by path BEFORE the loader's search order -- on a box with PyTorch on
  // LD_LIBRARY_PATH the bare name resolved to torch/lib/librccl.so, which made "no PyTorch in the ranks" depend on the
  // environment (round-5 review).
  const char* names[] = {"/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1", "librccl.so", "librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL)) != nullptr) break;
  if (!h)
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
  if (!h) {
    if (err) *err = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found");
    return false;
  }
  RcclApi a;
  a.handle = h;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
  a.CommAbort = (decltype(a.CommAbort))dlsym(h, "ncclCommAbort");  // optional: comm_abort falls back to destroy
  a.CommSplit = (decltype(a.CommSplit))dlsym(h, "ncclCommSplit");  // optional: without it comm2 = comm
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
  a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
  a.Broadcast = (decltype(a.Broadcast))dlsym(h, "ncclBroadcast");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.AllGather || !a.Broadcast) {
    if (err) *err = "librccl is missing required symbols";
    return false;
  }
  g_rccl = a;
  return true;
}

static int rccl_fail(gdml_ctx* ctx, const char* what, ncclResult_t_ r) {
  return gdml_fail(ctx, GDML_ERR_COMM, "%s failed: %s", what,
                   g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
}

extern "C" int gdml_comm_unique_id(void* id128_out) {
  if (!id128_out) return GDML_ERR_INVALID;
  std::string err;
  if (!rccl_load(&err)) return gdml_fail(nullptr, GDML_ERR_COMM, "%s", err.c_str());
  ncclUniqueId_ id;
  ncclResult_t_ r = g_rccl.GetUniqueId(&id);
  if (r != 0) return rccl_fail(nullptr, "ncclGetUniqueId", r);
  memcpy(id128_out, id.internal, 128);
  return GDML_OK;
}

extern "C" int gdml_comm_init(gdml_ctx* ctx, const void* id128, int rank, int world) {
  if (!ctx) return GDML_ERR_INVALID;
  if (world < 1 || rank < 0 || rank >= world)
    return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_comm_init: rank %d / world %d", rank, world);
  if (ctx->comm || ctx->host_allreduce || ctx->parked.on)
    return gdml_fail(ctx, GDML_ERR_STATE, "gdml_comm_init: communicator already initialised");
  HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (id128 == nullptr) {
    // "virtual rank": shard arithmetic without a communicator (collectives are skipped); used by the
    // single-GPU tests that stitch the shards of several contexts together on the host
    ctx->rank = rank;
    ctx->world = world;
    ctx->virtual_rank = true;
    return GDML_OK;
  }
  std::string err;
  if (!rccl_load(&err)) return gdml_fail(ctx, GDML_ERR_COMM, "%s", err.c_str());
  ncclUniqueId_ id;
  memcpy(id.internal, id128, 128);
  ncclComm_t_ comm = nullptr;
  ncclResult_t_ r = g_rccl.CommInitRank(&comm, world, id, rank);
  if (r != 0) return rccl_fail(ctx, "ncclCommInitRank", r);
  ctx->comm = comm;
  ctx->comm2 = nullptr;  // created on demand (comm_ensure_second): only the look-ahead distributed Cholesky wants it
  ctx->comm_aborted = false;
  ctx->rank = rank;
  ctx->world = world;
  {  // what RCCL itself says about the communicator (read back with gdml_get_option "comm.rccl_ranks": bench.py's rccl_ranks_seen)
    int cnt = -1;
    if (g_rccl.CommCount && g_rccl.CommCount(comm, &cnt) == 0) ctx->opts["comm.rccl_ranks"] = (double)cnt;
  }
  ctx->virtual_rank = false;
  ctx->coll_calls = 0;
  ctx->coll_bytes = 0.0;
  return GDML_OK;
}

// Second communicator over the same ranks (ncclCommSplit, a collective call: every rank must get here together).  Without
// ncclCommSplit in the loaded librccl, or when the split fails, the callers keep using the first communicator.
int comm_ensure_second(gdml_ctx* ctx) {
  if (!ctx->comm || ctx->comm2 || ctx->world <= 1 || !g_rccl.CommSplit) return GDML_OK;
  ncclComm_t_ c2 = nullptr;
  if (g_rccl.CommSplit((ncclComm_t_)ctx->comm, 0, ctx->rank, &c2, nullptr) == 0) ctx->comm2 = c2;
  return GDML_OK;
}

void comm_destroy(gdml_ctx* ctx) {
  if (ctx->parked.on) (void)gdml_comm_suspend(ctx, 0);
  if (ctx->comm2 && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t_)ctx->comm2);
  if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t_)ctx->comm);
  ctx->comm = ctx->comm2 = nullptr;
}

// Park / restore the communicator: while suspended the context behaves like a single GPU without a communicator (every
// entry point: unsharded assembly, local solves, no collective).  For work every rank does redundantly -- the LU fallback,
// which the sharded solvers do not carry (analytic.py:101-114).
extern "C" int gdml_comm_suspend(gdml_ctx* ctx, int suspend) {
  if (!ctx) return GDML_ERR_INVALID;
  if (suspend && !ctx->parked.on) {
    ctx->parked.on = true;
    ctx->parked.comm = ctx->comm; ctx->parked.comm2 = ctx->comm2;
    ctx->parked.rank = ctx->rank; ctx->parked.world = ctx->world;
    ctx->parked.virtual_rank = ctx->virtual_rank;
    ctx->parked.ar = ctx->host_allreduce; ctx->parked.ag = ctx->host_allgather;
    ctx->comm = ctx->comm2 = nullptr;
    ctx->rank = 0; ctx->world = 1;
    ctx->virtual_rank = false;
    ctx->host_allreduce = nullptr; ctx->host_allgather = nullptr;
  } else if (!suspend && ctx->parked.on) {
    ctx->comm = ctx->parked.comm; ctx->comm2 = ctx->parked.comm2;
    ctx->rank = ctx->parked.rank; ctx->world = ctx->parked.world;
    ctx->virtual_rank = ctx->parked.virtual_rank;
    ctx->host_allreduce = ctx->parked.ar; ctx->host_allgather = ctx->parked.ag;
    ctx->parked.on = false;
  }
  return GDML_OK;
}

void comm_abort(gdml_ctx* ctx) {
  if (!comm_active(ctx) || ctx->world <= 1) return;
  for (void** c : {&ctx->comm2, &ctx->comm})
    if (*c) {
      if (g_rccl.CommAbort) g_rccl.CommAbort((ncclComm_t_)*c);
      else if (g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t_)*c);
      *c = nullptr;
    }
  ctx->comm_aborted = true;  // (host-staged backend: the peers' gloo collective times out on its own)
}

extern "C" int gdml_comm_info(gdml_ctx* ctx, int* rank_out, int* world_out) {
  if (!ctx) return GDML_ERR_INVALID;
  if (rank_out) *rank_out = ctx->rank;
  if (world_out) *world_out = ctx->world;
  return GDML_OK;
}

// ---- shard arithmetic ---------------------------------------------------------------------
void shard_points(const gdml_ctx* ctx, int64_t M, int64_t* p0, int64_t* p1, int64_t* pts_per) {
  const int64_t W = ctx->world > 0 ? ctx->world : 1;
  const int64_t per = (M + W - 1) / W;
  int64_t a = per * ctx->rank, b = a + per;
  if (a > M) a = M;
  if (b > M) b = M;
  *p0 = a;
  *p1 = b;
  if (pts_per) *pts_per = per;
}

VecLayout vec_layout(const gdml_ctx* ctx, int use_E_cstr) {
  VecLayout L;
  const TrainSet& ts = ctx->ts;
  L.M = ts.M; L.N3 = 3 * (int64_t)ts.N; L.n_ff = L.M * L.N3;
  L.n = L.n_ff + (use_E_cstr ? L.M : 0);
  if (ctx->world > 1) {
    int64_t p0, p1, per;
    shard_points(ctx, ts.M, &p0, &p1, &per);
    L.per = per;
    L.two_seg = use_E_cstr ? 1 : 0;
    L.chunk = per * (L.N3 + (use_E_cstr ? 1 : 0));
    L.n_pad = L.chunk * ctx->world;
    L.row0 = use_E_cstr ? (int64_t)ctx->rank * L.chunk : p0 * L.N3;
    L.n_loc = (p1 - p0) * (L.N3 + (use_E_cstr ? 1 : 0));
  } else {
    L.per = L.M;
    L.chunk = L.n_pad = L.n_loc = L.n;
  }
  return L;
}

int vec_upload(gdml_ctx* ctx, const VecLayout& L, const double* host_ref, double* dev) {
  HIP_CHECK(ctx, hipMemsetAsync(dev, 0, L.n_pad * 8, ctx->stream));
  if (!L.two_seg) {
    HIP_CHECK(ctx, hipMemcpyAsync(dev, host_ref, L.n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return GDML_OK;
  }
  std::vector<double> h((size_t)L.n_pad, 0.0);
  for (int64_t g = 0; g < L.n; ++g) h[(size_t)L.pos(g)] = host_ref[g];
  HIP_CHECK(ctx, hipMemcpyAsync(dev, h.data(), L.n_pad * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return GDML_OK;
}

int vec_download(gdml_ctx* ctx, const VecLayout& L, const double* dev, double* host_ref, hipStream_t st) {
  if (!st) st = ctx->stream;
  if (!L.two_seg) {
    HIP_CHECK(ctx, hipMemcpyAsync(host_ref, dev, L.n * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(ctx, hipStreamSynchronize(st));
    return GDML_OK;
  }
  std::vector<double> h((size_t)L.n_pad);
  HIP_CHECK(ctx, hipMemcpyAsync(h.data(), dev, L.n_pad * 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(ctx, hipStreamSynchronize(st));
  for (int64_t g = 0; g < L.n; ++g) host_ref[g] = h[(size_t)L.pos(g)];
  return GDML_OK;
}

extern "C" int gdml_comm_init_host(gdml_ctx* ctx, int rank, int world, gdml_host_allreduce allreduce,
                                   gdml_host_allgather allgather, void* user) {
  if (!ctx || !allreduce || !allgather) return GDML_ERR_INVALID;
  if (world < 1 || rank < 0 || rank >= world)
    return gdml_fail(ctx, GDML_ERR_INVALID, "gdml_comm_init_host: rank %d / world %d", rank, world);
  if (ctx->comm || ctx->host_allreduce)
    return gdml_fail(ctx, GDML_ERR_STATE, "gdml_comm_init_host: communicator already initialised");
  ctx->host_allreduce = allreduce;
  ctx->host_allgather = allgather;
  ctx->host_coll_user = user;
  ctx->rank = rank;
  ctx->world = world;
  ctx->virtual_rank = false;
  ctx->coll_calls = 0;
  ctx->coll_bytes = 0.0;
  return GDML_OK;
}

extern "C" int gdml_comm_stats(gdml_ctx* ctx, int64_t* calls_out, double* bytes_out) {
  if (!ctx) return GDML_ERR_INVALID;
  if (calls_out) *calls_out = ctx->coll_calls;
  if (bytes_out) *bytes_out = ctx->coll_bytes;
  return GDML_OK;
}

// ---- collectives on the compute stream -----------------------------------------------------------
// Backend order: host-staged callbacks (gdml_comm_init_host) > RCCL communicator (gdml_comm_init with an id;
// used for EVERY world size including 1, so a one-rank communicator exercises the same calls) > nothing
// (single context without communicator, or a "virtual rank": the caller stitches the shards).
static int host_stage(gdml_ctx* ctx, int64_t bytes) {
  if (bytes <= ctx->h_coll_bytes) return GDML_OK;
  if (ctx->h_coll) (void)hipHostFree(ctx->h_coll);
  ctx->h_coll = nullptr;
  ctx->h_coll_bytes = 0;
  HIP_CHECK(ctx, hipHostMalloc((void**)&ctx->h_coll, (size_t)bytes, hipHostMallocDefault));
  ctx->h_coll_bytes = bytes;
  return GDML_OK;
}

int comm_allgather_inplace(gdml_ctx* ctx, double* buf, int64_t chunk) { return comm_allgather_inplace_on(ctx, buf, chunk, ctx->stream); }

int comm_allgather_inplace_on(gdml_ctx* ctx, double* buf, int64_t chunk, hipStream_t st) {
  if (ctx->virtual_rank) return GDML_OK;
  if (ctx->comm_aborted) return gdml_fail(ctx, GDML_ERR_COMM, "the communicator was aborted after a local failure on this rank");
  if (ctx->host_allgather) {
    const int64_t tot = chunk * ctx->world;
    GDML_TRY(host_stage(ctx, tot * 8));
    double* mine = ctx->h_coll + (int64_t)ctx->rank * chunk;
    HIP_CHECK(ctx, hipMemcpyAsync(mine, buf + (int64_t)ctx->rank * chunk, chunk * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (ctx->host_allgather(ctx->h_coll, chunk, ctx->host_coll_user) != 0)
      return gdml_fail(ctx, GDML_ERR_COMM, "host all-gather callback failed");
    HIP_CHECK(ctx, hipMemcpyAsync(buf, ctx->h_coll, tot * 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(ctx, hipStreamSynchronize(st));  // the staging buffer is reused by the next collective
  } else if (ctx->comm) {
    ncclResult_t_ r = g_rccl.AllGather(buf + (int64_t)ctx->rank * chunk, buf, (size_t)chunk, kNcclDouble,
                                       (ncclComm_t_)ctx->comm, st);
    if (r != 0) return rccl_fail(ctx, "ncclAllGather", r);
  } else {
    return GDML_OK;
  }
  ctx->coll_calls++;
  ctx->coll_bytes += 8.0 * (double)chunk;
  return GDML_OK;
}

int comm_allreduce_sum(gdml_ctx* ctx, double* buf, int64_t count) {
  if (ctx->virtual_rank) return GDML_OK;
  if (ctx->comm_aborted) return gdml_fail(ctx, GDML_ERR_COMM, "the communicator was aborted after a local failure on this rank");
  if (!ctx->host_allreduce && !ctx->comm) return GDML_OK;
  // large buffers go in pieces of 2^27 doubles (1 GiB) to stay inside comfortable message sizes
  const int64_t piece = (int64_t)1 << 27;
  for (int64_t off = 0; off < count; off += piece) {
    const int64_t c = (count - off < piece) ? count - off : piece;
    if (ctx->host_allreduce) {
      GDML_TRY(host_stage(ctx, c * 8));
      HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_coll, buf + off, c * 8, hipMemcpyDeviceToHost, ctx->stream));
      HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->host_allreduce(ctx->h_coll, c, ctx->host_coll_user) != 0)
        return gdml_fail(ctx, GDML_ERR_COMM, "host all-reduce callback failed");
      HIP_CHECK(ctx, hipMemcpyAsync(buf + off, ctx->h_coll, c * 8, hipMemcpyHostToDevice, ctx->stream));
      HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    } else {
      ncclResult_t_ r = g_rccl.AllReduce(buf + off, buf + off, (size_t)c, kNcclDouble, kNcclSum,
                                         (ncclComm_t_)ctx->comm, ctx->stream);
      if (r != 0) return rccl_fail(ctx, "ncclAllReduce", r);
    }
    ctx->coll_calls++;
    ctx->coll_bytes += 8.0 * (double)c;
  }
  return GDML_OK;
}

// buf (count doubles) of rank `root` to every rank, on stream st.  RCCL: ncclBroadcast.  Host-staged backend (two
// callbacks only: all-reduce, all-gather): the other ranks contribute zeros to an all-reduce.
int comm_broadcast_on(gdml_ctx* ctx, double* buf, int64_t count, int root, hipStream_t st, bool second_comm) {
  if (ctx->virtual_rank) return GDML_OK;
  if (ctx->comm_aborted) return gdml_fail(ctx, GDML_ERR_COMM, "the communicator was aborted after a local failure on this rank");
  if (ctx->host_allreduce) {
    GDML_TRY(host_stage(ctx, count * 8));
    if (ctx->rank == root) {
      HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_coll, buf, count * 8, hipMemcpyDeviceToHost, st));
      HIP_CHECK(ctx, hipStreamSynchronize(st));
    } else {
      HIP_CHECK(ctx, hipStreamSynchronize(st));  // the previous user of the staging buffer
      memset(ctx->h_coll, 0, (size_t)count * 8);
    }
    if (ctx->host_allreduce(ctx->h_coll, count, ctx->host_coll_user) != 0)
      return gdml_fail(ctx, GDML_ERR_COMM, "host all-reduce callback failed (broadcast)");
    if (ctx->rank != root) HIP_CHECK(ctx, hipMemcpyAsync(buf, ctx->h_coll, count * 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(ctx, hipStreamSynchronize(st));
  } else if (ctx->comm) {
    void* cm = (second_comm && ctx->comm2) ? ctx->comm2 : ctx->comm;
    ncclResult_t_ r = g_rccl.Broadcast(buf, buf, (size_t)count, kNcclDouble, root, (ncclComm_t_)cm, st);
    if (r != 0) return rccl_fail(ctx, "ncclBroadcast", r);
  } else {
    return GDML_OK;
  }
  ctx->coll_calls++;
  ctx->coll_bytes += 8.0 * (double)count;
  return GDML_OK;
}
