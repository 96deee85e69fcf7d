// Native transport of the mesh-partition exchanges: the ctx owns an RCCL communicator (created from a unique id the host
// program distributes once) and performs the halo exchanges and reductions of a cycle by itself, stream-ordered on its own
// stream -- no host callback, no Python between the kernels of a smoother iteration.
//
// What it replaces in the reference: PETSc's VecScatter (PetscSF) forward / reverse-add around PCApply_PATCH and MatMult and
// the MPI_Allreduce of KSPFGMRES's Gram-Schmidt step [3P], reached through alfi/solver.py:604-605 (distribution parameters)
// and alfi/relaxation.py:120-121 (patches of owned vertices only).
//
// MI355X: xGMI is point to point, every GPU has a direct link to each of its 7 peers.  A halo exchange is ONE group of
// ncclSend / ncclRecv pairs with the level's actual neighbours (a box partition of the mesh has at most 7 of them on a
// node), so each message travels over its own link; the reductions are one small ncclAllReduce (<= 33 doubles) each.
// librccl is resolved at run time (dlopen): the library itself has no link-time dependency on it and loads on hosts
// without RCCL; whichever copy the process already holds (PyTorch ships one with the same SONAME) is reused.
#include <dlfcn.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <rccl/rccl.h>
#include "common.h"

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;        // optional
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_api;
std::string g_api_error;

template <typename F>
bool resolve(void* h, const char* name, F* out) {
  *out = reinterpret_cast<F>(dlsym(h, name));
  if (!*out) g_api_error = std::string("librccl lacks ") + name;
  return *out != nullptr;
}

std::mutex g_api_mutex;     // ctxs of different threads may create their communicators concurrently

// nullptr + g_api_error on failure
RcclApi* rccl() {
  std::lock_guard<std::mutex> lock(g_api_mutex);
  if (g_api.handle) return &g_api;
  const char* names[] = {getenv("ALFI_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    if (!n || !*n) continue;
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    const char* e = dlerror();          // (one call: dlerror() clears the message it returns)
    g_api_error = std::string("cannot load librccl: ") + (e ? e : "not found");
    return nullptr;
  }
  RcclApi a;
  a.handle = h;
  if (!(resolve(h, "ncclGetUniqueId", &a.GetUniqueId) && resolve(h, "ncclCommInitRank", &a.CommInitRank) &&
        resolve(h, "ncclCommDestroy", &a.CommDestroy) && resolve(h, "ncclGroupStart", &a.GroupStart) &&
        resolve(h, "ncclGroupEnd", &a.GroupEnd) && resolve(h, "ncclSend", &a.Send) && resolve(h, "ncclRecv", &a.Recv) &&
        resolve(h, "ncclAllReduce", &a.AllReduce) && resolve(h, "ncclGetErrorString", &a.GetErrorString) &&
        resolve(h, "ncclCommCount", &a.CommCount) && resolve(h, "ncclCommUserRank", &a.CommUserRank)))
    return nullptr;
  a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(h, "ncclCommAbort"));
  g_api = a;
  return &g_api;
}

}  // namespace

struct NativeComm {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  bool own_dred = false;
  bool failed = false;             // an exchange failed: the peers may be blocked in their half; abort instead of destroy
  hipStream_t side = nullptr;      // asynchronous exchanges (overlap with interior work) run here
  hipEvent_t ev_ready = nullptr;   // ctx stream -> side stream: buffers packed
  hipEvent_t ev_done = nullptr;    // side stream -> ctx stream: exchange finished
};

#define ALFI_NCCL_CHECK(ctx, api, call)                                                                      \
  do {                                                                                                       \
    ncclResult_t r_ = (call);                                                                                \
    if (r_ != ncclSuccess)                                                                                   \
      return alfi_set_error(ctx, ALFI_E_COMM, "%s failed: %s (%s:%d)", #call, (api)->GetErrorString(r_), __FILE__, \
                            __LINE__);                                                                       \
  } while (0)

void native_destroy(alfi_ctx* ctx) {
  NativeComm* N = ctx->nat;
  if (!N) return;
  // after a failed exchange the streams may hold an RCCL kernel that waits for a peer: abort the communicator first
  const bool abort = N->failed && N->comm && g_api.handle && g_api.CommAbort;
  if (abort) (void)g_api.CommAbort(N->comm);
  (void)hipStreamSynchronize(ctx->stream);
  if (N->side) (void)hipStreamSynchronize(N->side);
  if (!abort && N->comm && g_api.handle) (void)g_api.CommDestroy(N->comm);
  if (N->ev_ready) (void)hipEventDestroy(N->ev_ready);
  if (N->ev_done) (void)hipEventDestroy(N->ev_done);
  if (N->side) (void)hipStreamDestroy(N->side);
  if (N->own_dred && ctx->dred) {
    (void)hipFree(ctx->dred);
    ctx->dred = nullptr;
  }
  delete N;
  ctx->nat = nullptr;
}

int native_allreduce(alfi_ctx* ctx, int64_t offset, int64_t count) {
  NativeComm* N = ctx->nat;
  RcclApi* api = &g_api;
  if (N->nranks == 1) return 0;      // the sum over one rank
  double* p = ctx->dred + offset;
  ALFI_NCCL_CHECK(ctx, api, api->AllReduce(p, p, (size_t)count, ncclDouble, ncclSum, N->comm, ctx->stream));
  return 0;
}

int native_exchange(alfi_level* L, int dir, bool async) {
  alfi_ctx* ctx = L->ctx;
  NativeComm* N = ctx->nat;
  RcclApi* api = &g_api;
  hipStream_t s = ctx->stream;
  if (async) {
    ALFI_HIP_CHECK(ctx, hipEventRecord(N->ev_ready, ctx->stream));
    ALFI_HIP_CHECK(ctx, hipStreamWaitEvent(N->side, N->ev_ready, 0));
    s = N->side;
  }
  const std::vector<int>& nbr = dir == 2 ? L->sum_rank : L->nbr_rank;
  const size_t nn = nbr.size();
  if (nn > 0) {
    // forward: my send-buffer segments go out, my receive-buffer segments come in; reverse: the other way round; sum
    // exchange (dir 2): its own buffers, the same segment layout both ways
    const double* out = dir == 2 ? L->sum_sendbuf : dir == 0 ? L->halo_sendbuf : L->halo_recvbuf;
    double* in = dir == 2 ? L->sum_recvbuf : dir == 0 ? L->halo_recvbuf : L->halo_sendbuf;
    const std::vector<int64_t>& ooff = dir == 2 ? L->sum_off : dir == 0 ? L->nbr_send_off : L->nbr_recv_off;
    const std::vector<int64_t>& ocnt = dir == 2 ? L->sum_cnt : dir == 0 ? L->nbr_send_cnt : L->nbr_recv_cnt;
    const std::vector<int64_t>& ioff = dir == 2 ? L->sum_off : dir == 0 ? L->nbr_recv_off : L->nbr_send_off;
    const std::vector<int64_t>& icnt = dir == 2 ? L->sum_cnt : dir == 0 ? L->nbr_recv_cnt : L->nbr_send_cnt;
    ALFI_NCCL_CHECK(ctx, api, api->GroupStart());
    // a failing Send / Recv must not leave the group open on this thread (every later call on the communicator, CommDestroy
    // included, would be queued behind it): remember the first error, close the group, then report
    ncclResult_t first = ncclSuccess;
    for (size_t i = 0; i < nn && first == ncclSuccess; ++i) {
      if (ocnt[i] > 0) first = api->Send(out + ooff[i], (size_t)ocnt[i], ncclDouble, nbr[i], N->comm, s);
      if (first == ncclSuccess && icnt[i] > 0)
        first = api->Recv(in + ioff[i], (size_t)icnt[i], ncclDouble, nbr[i], N->comm, s);
    }
    const ncclResult_t closed = api->GroupEnd();
    if (first != ncclSuccess || closed != ncclSuccess) {
      N->failed = true;
      return alfi_set_error(ctx, ALFI_E_COMM, "halo exchange with %zu neighbours failed: %s", nn,
                            api->GetErrorString(first != ncclSuccess ? first : closed));
    }
  }
  if (async) ALFI_HIP_CHECK(ctx, hipEventRecord(N->ev_done, N->side));
  return 0;
}

int native_wait(alfi_ctx* ctx) {
  ALFI_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->nat->ev_done, 0));
  return 0;
}

extern "C" {

int alfi_comm_unique_id(void* id_out, int64_t len) {
  if (!id_out || len < ALFI_COMM_ID_BYTES) return alfi_set_error(nullptr, ALFI_E_ARG, "id buffer must hold %d bytes", ALFI_COMM_ID_BYTES);
  RcclApi* api = rccl();
  if (!api) return alfi_set_error(nullptr, ALFI_E_COMM, "%s", g_api_error.c_str());
  static_assert(sizeof(ncclUniqueId) == ALFI_COMM_ID_BYTES, "unique id size");
  ncclUniqueId id;
  ncclResult_t r = api->GetUniqueId(&id);
  if (r != ncclSuccess) return alfi_set_error(nullptr, ALFI_E_COMM, "ncclGetUniqueId failed: %s", api->GetErrorString(r));
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

int alfi_ctx_comm_init(alfi_ctx* ctx, const void* id, int rank, int nranks) {
  if (!ctx || !id) return alfi_set_error(ctx, ALFI_E_ARG, "NULL argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return alfi_set_error(ctx, ALFI_E_ARG, "rank %d of %d", rank, nranks);
  if (ctx->nat) return alfi_set_error(ctx, ALFI_E_STATE, "the ctx already has a communicator");
  if (ctx->comm) return alfi_set_error(ctx, ALFI_E_STATE, "the ctx already exchanges through a callback (alfi_ctx_set_comm)");
  RcclApi* api = rccl();
  if (!api) return alfi_set_error(ctx, ALFI_E_COMM, "%s", g_api_error.c_str());
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  NativeComm* N = new NativeComm();
  N->rank = rank;
  N->nranks = nranks;
  ctx->nat = N;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclResult_t r = api->CommInitRank(&N->comm, nranks, uid, rank);
  if (r != ncclSuccess) {
    N->comm = nullptr;
    native_destroy(ctx);
    return alfi_set_error(ctx, ALFI_E_COMM, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, api->GetErrorString(r));
  }
  hipError_t e = hipStreamCreateWithFlags(&N->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&N->ev_ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&N->ev_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->dred, sizeof(double) * 2 * RED_MAXV);
  if (e == hipSuccess) {
    N->own_dred = true;
    e = hipMemsetAsync(ctx->dred, 0, sizeof(double) * 2 * RED_MAXV, ctx->stream);
  }
  if (e != hipSuccess) {
    native_destroy(ctx);
    return alfi_set_error(ctx, ALFI_E_HIP, "communicator resources: %s", hipGetErrorString(e));
  }
  // |w - V h| of the partitioned FGMRES by its own all-reduce, as PETSc's VecNorm (the default).  ALFI_DIST_EXACT_NORM=0: from
  // |w|^2 - |h|^2 with the dots' all-reduce (one reduction fewer per iteration; absolute error eps |w|^2, so a new direction
  // that is small against w is mis-normalised -- measurements only)
  const char* en = getenv("ALFI_DIST_EXACT_NORM");
  ctx->exact_norm = !(en && atoi(en) == 0);
  return 0;
}

// what the COMMUNICATOR says (ncclCommUserRank / ncclCommCount on the ctx's ncclComm_t), not what alfi_ctx_comm_init was told
int alfi_ctx_comm_size(alfi_ctx* ctx, int* rank, int* nranks) {
  if (!ctx->nat || !ctx->nat->comm) return alfi_set_error(ctx, ALFI_E_STATE, "no communicator (alfi_ctx_comm_init)");
  RcclApi* api = &g_api;
  int r = -1, n = -1;
  ALFI_NCCL_CHECK(ctx, api, api->CommUserRank(ctx->nat->comm, &r));
  ALFI_NCCL_CHECK(ctx, api, api->CommCount(ctx->nat->comm, &n));
  if (rank) *rank = r;
  if (nranks) *nranks = n;
  return 0;
}

// TEST HOOK: a rank may name ITSELF as a neighbour.  RCCL accepts a grouped ncclSend + ncclRecv to one's own rank, so the one
// GPU of a test box runs the real point-to-point path -- pack, group of send / receive pairs on the library's stream, unpack --
// with non-empty buffers (tests/test_gpu_dist.py::test_real_rccl_send_recv_to_self).  No partition of a mesh produces this.
int alfi_ctx_comm_allow_self(alfi_ctx* ctx, int on) {
  ctx->comm_allow_self = on != 0;
  return 0;
}

int alfi_ctx_comm_destroy(alfi_ctx* ctx) {
  native_destroy(ctx);
  return 0;
}

int alfi_level_set_neighbours(alfi_level* L, int nnbr, const int32_t* ranks, const int64_t* send_nodes,
                              const int64_t* recv_nodes) {
  alfi_ctx* ctx = L->ctx;
  if (!L->has_halo) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_set_neighbours before alfi_level_set_partition");
  if (nnbr < 0 || (nnbr > 0 && (!ranks || !send_nodes || !recv_nodes))) return alfi_set_error(ctx, ALFI_E_ARG, "NULL neighbour arrays");
  const int me = ctx->nat ? ctx->nat->rank : -1, world = ctx->nat ? ctx->nat->nranks : INT32_MAX;
  int64_t so = 0, ro = 0;
  std::vector<int> nr;
  std::vector<int64_t> soff, scnt, roff, rcnt;
  for (int i = 0; i < nnbr; ++i) {
    if (ranks[i] < 0 || ranks[i] >= world || (ranks[i] == me && !ctx->comm_allow_self) || (i > 0 && ranks[i] <= ranks[i - 1]))
      return alfi_set_error(ctx, ALFI_E_ARG, "neighbour ranks must be ascending, distinct from this rank and inside the group");
    if (send_nodes[i] < 0 || recv_nodes[i] < 0) return alfi_set_error(ctx, ALFI_E_ARG, "negative neighbour count");
    nr.push_back(ranks[i]);
    soff.push_back(so * L->bs);
    scnt.push_back(send_nodes[i] * L->bs);
    roff.push_back(ro * L->bs);
    rcnt.push_back(recv_nodes[i] * L->bs);
    so += send_nodes[i];
    ro += recv_nodes[i];
  }
  if (so != L->halo_nsend || ro != L->halo_nghost)
    return alfi_set_error(ctx, ALFI_E_ARG, "neighbour counts (%lld send, %lld receive nodes) do not add up to the level's halo "
                          "(%lld, %lld)", (long long)so, (long long)ro, (long long)L->halo_nsend, (long long)L->halo_nghost);
  L->nbr_rank = nr;
  L->nbr_send_off = soff;
  L->nbr_send_cnt = scnt;
  L->nbr_recv_off = roff;
  L->nbr_recv_cnt = rcnt;
  return 0;
}

int alfi_level_set_sum_exchange(alfi_level* L, int nnbr, const int32_t* ranks, const int64_t* counts,
                                const int32_t* send_nodes, int64_t nshared, const int32_t* sum_nodes, const int32_t* sum_ptr,
                                const int32_t* sum_src) {
  alfi_ctx* ctx = L->ctx;
  if (!L->has_halo) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_set_sum_exchange before alfi_level_set_partition");
  if (!ctx->nat) return alfi_set_error(ctx, ALFI_E_STATE, "the sum exchange needs the native transport (alfi_ctx_comm_init)");
  if (nnbr < 0 || nshared < 0 || (nnbr > 0 && (!ranks || !counts || !send_nodes)) || (nshared > 0 && (!sum_nodes || !sum_ptr || !sum_src)))
    return alfi_set_error(ctx, ALFI_E_ARG, "NULL sum-exchange arrays");
  const int me = ctx->nat->rank, world = ctx->nat->nranks;
  const int64_t nloc = L->n / L->bs;
  std::vector<int> nr;
  std::vector<int64_t> off, cnt;
  int64_t total = 0;
  for (int i = 0; i < nnbr; ++i) {
    if (ranks[i] < 0 || ranks[i] >= world || (ranks[i] == me && !ctx->comm_allow_self) || (i > 0 && ranks[i] <= ranks[i - 1]))
      return alfi_set_error(ctx, ALFI_E_ARG, "neighbour ranks must be ascending, distinct from this rank and inside the group");
    if (counts[i] <= 0) return alfi_set_error(ctx, ALFI_E_ARG, "a sum-exchange neighbour shares no node");
    nr.push_back(ranks[i]);
    off.push_back(total * L->bs);
    cnt.push_back(counts[i] * L->bs);
    total += counts[i];
  }
  for (int64_t i = 0; i < total; ++i)
    if (send_nodes[i] < 0 || send_nodes[i] >= nloc) return alfi_set_error(ctx, ALFI_E_ARG, "sum exchange: send node out of range");
  for (int64_t u = 0; u < nshared; ++u) {
    if (sum_nodes[u] < 0 || sum_nodes[u] >= nloc || sum_ptr[u + 1] <= sum_ptr[u])
      return alfi_set_error(ctx, ALFI_E_ARG, "sum exchange: shared node %lld malformed", (long long)u);
    int own = 0;
    for (int32_t q = sum_ptr[u]; q < sum_ptr[u + 1]; ++q) {
      if (sum_src[q] >= total) return alfi_set_error(ctx, ALFI_E_ARG, "sum exchange: source beyond the receive buffer");
      own += sum_src[q] < 0;
    }
    if (own != 1) return alfi_set_error(ctx, ALFI_E_ARG, "sum exchange: every shared node sums exactly one own value");
  }
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (void* p : {(void*)L->sum_send_nodes, (void*)L->sum_nodes, (void*)L->sum_ptr, (void*)L->sum_src, (void*)L->sum_sendbuf,
                  (void*)L->sum_recvbuf})
    if (p) (void)hipFree(p);
  L->sum_send_nodes = L->sum_nodes = L->sum_ptr = L->sum_src = nullptr;
  L->sum_sendbuf = L->sum_recvbuf = nullptr;
  L->sum_ready = false;
  auto up = [&](int32_t** d, const int32_t* h, int64_t n) {
    if (hipMalloc((void**)d, (size_t)std::max<int64_t>(n, 1) * 4) != hipSuccess) return false;
    return n == 0 || hipMemcpy(*d, h, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess;
  };
  bool ok = up(&L->sum_send_nodes, send_nodes, total) && up(&L->sum_nodes, sum_nodes, nshared) &&
            up(&L->sum_ptr, sum_ptr, nshared + 1) && up(&L->sum_src, sum_src, nshared > 0 ? sum_ptr[nshared] : 0);
  ok = ok && hipMalloc((void**)&L->sum_sendbuf, (size_t)std::max<int64_t>(total * L->bs, 1) * 8) == hipSuccess;
  ok = ok && hipMalloc((void**)&L->sum_recvbuf, (size_t)std::max<int64_t>(total * L->bs, 1) * 8) == hipSuccess;
  if (!ok) return alfi_set_error(ctx, ALFI_E_HIP, "sum-exchange buffers: %s", hipGetErrorString(hipGetLastError()));
  L->sum_rank = nr;
  L->sum_off = off;
  L->sum_cnt = cnt;
  L->sum_nsend = total;
  L->sum_nshared = nshared;
  L->sum_ready = true;
  return 0;
}

}  // extern "C"
