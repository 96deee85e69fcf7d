// C ABI of libalfi_hip.so (include/alfi_hip.h): handles, device memory, and the stream-ordered orchestration of the
// smoother (PETSc KSPFGMRES + PCPATCH), the Schoeberl transfers and the PCMG V / full cycles.  No host synchronisation
// happens inside the smoother or the cycles: every scalar (norms, Hessenberg, Givens) stays on the device.
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include "common.h"
#include "hs_layout.h"

static thread_local std::string g_create_error;

bool alfi_test_large_paths() {
  static const bool on = getenv("ALFI_TEST_LARGE_PATHS") && atoi(getenv("ALFI_TEST_LARGE_PATHS")) == 1;
  return on;
}

int alfi_set_error(alfi_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx)
    ctx->err = buf;
  else
    g_create_error = buf;
  return code;
}

// ---- profiling ---------------------------------------------------------------------------------------------------------
int alfi_prof_begin(alfi_ctx* ctx, int kind) {
  if (!ctx->prof) return -1;
  if (ctx->prof == 2 && kind != ALFI_EV_PATCH_APPLY && kind != ALFI_EV_COMM) return -1;
  if (ctx->prof == 3 && kind != ALFI_EV_PATCH_APPLY) return -1;
  if (ctx->ev_used == ctx->ev_pool.size()) {
    alfi_ctx::EvPair p;
    if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return -1;
    ctx->ev_pool.push_back(p);
  }
  const int t = (int)ctx->ev_used++;
  ctx->ev_pool[t].kind = kind;
  ctx->ev_pool[t].tag = ctx->cur_tag;
  (void)hipEventRecord(ctx->ev_pool[t].a, ctx->stream);
  return t;
}
int alfi_prof_end(alfi_ctx* ctx, int token) {
  if (token < 0) return 0;
  (void)hipEventRecord(ctx->ev_pool[token].b, ctx->stream);
  return 0;
}

template <typename T>
static int dev_alloc(alfi_ctx* ctx, T** p, int64_t count) {
  *p = nullptr;
  if (count <= 0) count = 1;
  ALFI_HIP_CHECK(ctx, hipMalloc((void**)p, (size_t)count * sizeof(T)));
  return 0;
}
template <typename T>
static int dev_upload(alfi_ctx* ctx, T** p, const T* host, int64_t count) {
  ALFI_CHECK(dev_alloc(ctx, p, count));
  if (count > 0) ALFI_HIP_CHECK(ctx, hipMemcpy(*p, host, (size_t)count * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}
static void dev_free(void* p) {
  if (p) (void)hipFree(p);
}
static int upload_csr(alfi_ctx* ctx, DevCSR* d, const alfi_csr_host* h);
static void free_csr(DevCSR* d);
static void free_mult_schedule(alfi_level* L);
// the sticky device-side error word (bounded waits of persistent kernels), read after a synchronisation
static int check_dev_err(alfi_ctx* ctx) {
  if (!ctx->dev_err) return 0;
  int32_t e = 0;
  if (hipMemcpy(&e, ctx->dev_err, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return 0;
  if (e != 0) return alfi_set_error(ctx, ALFI_E_STATE, "a device-side dependency wait ran into its bound (multiplicative sweep "
                                    "schedule): results since then are incomplete");
  return 0;
}

// Chunk tables of the flat layout.  Large matrices: equal chunks of SPMV_CHUNK blocks (rows may continue into the next
// chunk; the fix-up launch completes them).  Small ones (nnzb <= SPMV_ALIGNED_MAX, no row longer than a chunk, unless
// ALFI_SPMV_ALIGNED=0): chunks of whole block rows, greedily packed, so the product is ONE launch.  break_row >= 0: a chunk
// boundary is forced in front of that block row (the owned prefix of a partitioned level); returns the number of chunks
// before it in *nchunks_before.
static int build_chunk_tables(alfi_ctx* ctx, DevBSR* d, const int32_t* rowptr, int64_t nbrows, int64_t break_row,
                              int64_t* nchunks_before = nullptr) {
  dev_free(d->chunk_row);
  dev_free(d->chunk_start);
  d->chunk_row = nullptr;
  d->chunk_start = nullptr;
  bool aligned = !alfi_test_large_paths() && d->nnzb <= SPMV_ALIGNED_MAX;
  for (int64_t i = 0; i < nbrows && aligned; ++i) aligned = rowptr[i + 1] - rowptr[i] <= SPMV_CHUNK;
  d->aligned = aligned;
  if (!aligned) {
    d->nchunks = (d->nnzb + SPMV_CHUNK - 1) / SPMV_CHUNK;
    std::vector<int32_t> chunk_row(d->nchunks);
    int64_t row = 0;
    for (int64_t c = 0; c < d->nchunks; ++c) {
      const int64_t k = c * SPMV_CHUNK;
      while (rowptr[row + 1] <= k) ++row;
      chunk_row[c] = (int32_t)row;
    }
    ALFI_CHECK(dev_upload(ctx, &d->chunk_row, chunk_row.data(), d->nchunks));
    if (nchunks_before) *nchunks_before = -1;
    return 0;
  }
  std::vector<int64_t> start;
  std::vector<int32_t> chunk_row;
  int64_t i = 0;
  while (i < nbrows) {
    start.push_back(rowptr[i]);
    chunk_row.push_back((int32_t)i);
    if (nchunks_before && i == break_row) *nchunks_before = (int64_t)start.size() - 1;
    int64_t e = i;
    while (e < nbrows && rowptr[e + 1] - rowptr[i] <= SPMV_CHUNK && (e == i || e != break_row)) ++e;
    i = e;
  }
  if (nchunks_before && break_row >= nbrows) *nchunks_before = (int64_t)start.size();
  start.push_back(rowptr[nbrows]);
  d->nchunks = (int64_t)chunk_row.size();
  ALFI_CHECK(dev_upload(ctx, &d->chunk_row, chunk_row.data(), d->nchunks));
  ALFI_CHECK(dev_upload(ctx, &d->chunk_start, start.data(), d->nchunks + 1));
  return 0;
}

static int upload_bsr(alfi_ctx* ctx, DevBSR* d, const alfi_bsr_host* h, int bs) {
  d->nbrows = h->nbrows;
  d->nbcols = h->nbcols;
  d->bs = bs;
  d->nnzb = h->rowptr[h->nbrows];
  // lane-major layout + segmented SpMV needs every block row non-empty (the row of a block is found by counting row
  // starts); matrices with empty rows (transfer pieces) keep the host layout and the row-per-lane-group kernel
  bool flat = d->nnzb > 0;
  for (int64_t i = 0; i < h->nbrows && flat; ++i) flat = h->rowptr[i + 1] > h->rowptr[i];
  d->flat = flat ? 1 : 0;
  ALFI_CHECK(dev_upload(ctx, &d->rowptr, h->rowptr, h->nbrows + 1));
  if (!flat) {
    ALFI_CHECK(dev_upload(ctx, &d->colidx, h->colidx, d->nnzb));
    ALFI_CHECK(dev_alloc(ctx, &d->vals, d->nnzb * bs * bs));
    if (!h->vals) ALFI_HIP_CHECK(ctx, hipMemsetAsync(d->vals, 0, sizeof(double) * d->nnzb * bs * bs, ctx->stream));
  } else {
    std::vector<int32_t> cf(h->colidx, h->colidx + d->nnzb);
    for (int64_t i = 0; i < h->nbrows; ++i) cf[h->rowptr[i]] |= (int32_t)0x80000000;
    ALFI_CHECK(dev_upload(ctx, &d->colidx, cf.data(), d->nnzb));
    ALFI_CHECK(build_chunk_tables(ctx, d, h->rowptr, h->nbrows, -1));
    ALFI_CHECK(build_spmv_dedup(ctx, d));        // (device-side: sorts every group of SPMV_WG columns in LDS)
    ALFI_CHECK(dev_alloc(ctx, &d->carry, d->nchunks * bs));
    ALFI_CHECK(dev_alloc(ctx, &d->carry_row, d->nchunks));
    const int64_t padded = ((d->nnzb + 63) / 64) * 64 * bs * bs;
    ALFI_CHECK(dev_alloc(ctx, &d->vals, padded));
    // on the ctx stream: a memset on the null stream is asynchronous and NOT ordered against this (non-blocking) stream,
    // it could land after the value upload below
    ALFI_HIP_CHECK(ctx, hipMemsetAsync(d->vals, 0, sizeof(double) * padded, ctx->stream));
  }
  return h->vals ? upload_bsr_values(ctx, d, h->vals) : 0;     // (no values: zeros, see alfi_level_create)
}
static void free_assembly(AssemblyDev* S);
static void free_bsr(DevBSR* d) {
  dev_free(d->rowptr);
  dev_free(d->colidx);
  dev_free(d->vals);
  dev_free(d->chunk_row);
  dev_free(d->chunk_start);
  dev_free(d->carry);
  dev_free(d->carry_row);
  dev_free(d->lidx);
  dev_free(d->ucol);
  dev_free(d->uptr);
  *d = DevBSR();
}

// ---- mesh-partition exchanges: the ctx's own RCCL communicator (alfi_ctx_comm_init, comm.hip) or, as the test transport,
// a host callback (alfi_ctx_set_comm) ----------------------------------------------------------------------------------
static int comm_call(alfi_level* L, int op, int64_t offset, int64_t count) {
  alfi_ctx* ctx = L->ctx;
  switch (op) {             // bookkeeping for alfi_ctx_comm_stats
    case ALFI_COMM_ALLREDUCE:
      ++ctx->comm_nred;
      ctx->comm_sent += count;
      break;
    case ALFI_COMM_HALO_FWD:
    case ALFI_COMM_HALO_FWD_BEGIN:
      ++ctx->comm_nhalo;
      ctx->comm_sent += L->halo_nsend * L->bs;
      break;
    case ALFI_COMM_HALO_REV:
    case ALFI_COMM_HALO_REV_BEGIN:
      ++ctx->comm_nhalo;
      ctx->comm_sent += L->halo_nghost * L->bs;
      break;
    case ALFI_COMM_HALO_SUM:
      ++ctx->comm_nhalo;
      ctx->comm_sent += L->sum_nsend * L->bs;
      break;
    default: break;
  }
  if (ctx->nat) {
    switch (op) {
      case ALFI_COMM_HALO_SUM: return native_exchange(L, 2, false);
      case ALFI_COMM_ALLREDUCE: return native_allreduce(ctx, offset, count);
      case ALFI_COMM_HALO_FWD: return native_exchange(L, 0, false);
      case ALFI_COMM_HALO_REV: return native_exchange(L, 1, false);
      case ALFI_COMM_HALO_FWD_BEGIN: return native_exchange(L, 0, true);
      case ALFI_COMM_HALO_REV_BEGIN: return native_exchange(L, 1, true);
      default: return native_wait(ctx);                      // ALFI_COMM_HALO_FWD_END / _REV_END
    }
  }
  if (!ctx->comm) return alfi_set_error(ctx, ALFI_E_STATE, "partitioned level used before alfi_ctx_comm_init / alfi_ctx_set_comm");
  const int rc = ctx->comm(ctx->comm_user, op, L->id, offset, count);
  if (rc != 0) return alfi_set_error(ctx, ALFI_E_STATE, "communication callback failed (op %d, rc %d)", op, rc);
  return 0;
}

// sum dred[offset .. offset+count) over the ranks
static int comm_allreduce(alfi_level* L, int64_t offset, int64_t count) {
  alfi_ctx* ctx = L->ctx;
  int t = alfi_prof_begin(ctx, ALFI_EV_COMM);
  ALFI_CHECK(comm_call(L, ALFI_COMM_ALLREDUCE, offset, count));
  alfi_prof_end(ctx, t);
  return 0;
}

// owner -> ghost copies of level vector v (ghost slots of v are overwritten)
static int halo_fwd(alfi_level* L, double* v) {
  alfi_ctx* ctx = L->ctx;
  if (!L->has_halo) return alfi_set_error(ctx, ALFI_E_STATE, "halo exchange on a level without alfi_level_set_partition");
  int t = alfi_prof_begin(ctx, ALFI_EV_COMM);
  ALFI_CHECK(launch_halo_pack(ctx, L->halo_sendbuf, v, L->halo_send_nodes, L->halo_nsend, L->bs));
  ALFI_CHECK(comm_call(L, ALFI_COMM_HALO_FWD, 0, 0));
  // (an own copy kernel: the runtime's device-to-device copy costs more per call than these surface-sized buffers take)
  ALFI_CHECK(launch_copy(ctx, v + L->n_own, L->halo_recvbuf, L->halo_nghost * L->bs));
  alfi_prof_end(ctx, t);
  return 0;
}

// the same in two halves: pack + start the exchange | (caller launches work that needs no ghost value) | wait + unpack
static int halo_fwd_begin(alfi_level* L, const double* v) {
  alfi_ctx* ctx = L->ctx;
  int t = alfi_prof_begin(ctx, ALFI_EV_COMM);
  ALFI_CHECK(launch_halo_pack(ctx, L->halo_sendbuf, v, L->halo_send_nodes, L->halo_nsend, L->bs));
  ALFI_CHECK(comm_call(L, ALFI_COMM_HALO_FWD_BEGIN, 0, 0));
  alfi_prof_end(ctx, t);
  return 0;
}
static int halo_fwd_end(alfi_level* L, double* v) {
  alfi_ctx* ctx = L->ctx;
  int t = alfi_prof_begin(ctx, ALFI_EV_COMM);
  ALFI_CHECK(comm_call(L, ALFI_COMM_HALO_FWD_END, 0, 0));
  // (an own copy kernel: the runtime's device-to-device copy costs more per call than these surface-sized buffers take)
  ALFI_CHECK(launch_copy(ctx, v + L->n_own, L->halo_recvbuf, L->halo_nghost * L->bs));
  alfi_prof_end(ctx, t);
  return 0;
}

// ghost contributions of v added onto their owners (ghost slots of v keep their local values)
static int halo_rev(alfi_level* L, double* v) {
  alfi_ctx* ctx = L->ctx;
  if (!L->has_halo) return alfi_set_error(ctx, ALFI_E_STATE, "halo exchange on a level without alfi_level_set_partition");
  int t = alfi_prof_begin(ctx, ALFI_EV_COMM);
  ALFI_CHECK(launch_copy(ctx, L->halo_recvbuf, v + L->n_own, L->halo_nghost * L->bs));
  ALFI_CHECK(comm_call(L, ALFI_COMM_HALO_REV, 0, 0));
  ALFI_CHECK(launch_halo_add(ctx, v, L->halo_sendbuf, L->rev_nodes, L->rev_ptr, L->rev_pos, L->rev_nuniq, L->bs));
  alfi_prof_end(ctx, t);
  return 0;
}

// merged reverse-add + forward (alfi_level_set_sum_exchange): afterwards every holder of a shared node -- owner and ghost
// copies -- has the sum of all holders' values, added in one fixed order
static int halo_sum(alfi_level* L, double* v) {
  alfi_ctx* ctx = L->ctx;
  int t = alfi_prof_begin(ctx, ALFI_EV_COMM);
  ALFI_CHECK(launch_halo_pack(ctx, L->sum_sendbuf, v, L->sum_send_nodes, L->sum_nsend, L->bs));
  ALFI_CHECK(comm_call(L, ALFI_COMM_HALO_SUM, 0, 0));
  ALFI_CHECK(launch_halo_sum(ctx, v, L->sum_recvbuf, L->sum_nodes, L->sum_ptr, L->sum_src, L->sum_nshared, L->bs));
  alfi_prof_end(ctx, t);
  return 0;
}

// reverse route in two halves (see halo_rev): ghost slots -> buffer, start | ... | wait, add onto the owners
static int halo_rev_begin(alfi_level* L, const double* v) {
  alfi_ctx* ctx = L->ctx;
  int t = alfi_prof_begin(ctx, ALFI_EV_COMM);
  ALFI_CHECK(launch_copy(ctx, L->halo_recvbuf, v + L->n_own, L->halo_nghost * L->bs));
  ALFI_CHECK(comm_call(L, ALFI_COMM_HALO_REV_BEGIN, 0, 0));
  alfi_prof_end(ctx, t);
  return 0;
}
static int halo_rev_end(alfi_level* L, double* v) {
  alfi_ctx* ctx = L->ctx;
  int t = alfi_prof_begin(ctx, ALFI_EV_COMM);
  ALFI_CHECK(comm_call(L, ALFI_COMM_HALO_REV_END, 0, 0));
  ALFI_CHECK(launch_halo_add(ctx, v, L->halo_sendbuf, L->rev_nodes, L->rev_ptr, L->rev_pos, L->rev_nuniq, L->bs));
  alfi_prof_end(ctx, t);
  return 0;
}

static void free_cond(alfi_level* L) {
  for (void* q : L->cond_allocs) dev_free(q);
  L->cond_allocs.clear();
  L->cd = CondDev();
  L->cond = false;
  L->h_sptr.clear();
  L->h_cond_gptr.clear();
  L->cond_ngroups = L->cond_mat_doubles = L->cond_sinv_doubles = 0;
  L->cond_lds_bytes = L->cond_max_s = L->cond_umax = 0;
}

template <typename T>
static int cond_upload(alfi_level* L, const T** dst, const std::vector<T>& src) {
  T* d = nullptr;
  ALFI_CHECK(dev_upload(L->ctx, &d, src.data(), (int64_t)src.size()));
  L->cond_allocs.push_back(d);
  *dst = d;
  return 0;
}

extern "C" {

// ---- context -------------------------------------------------------------------------------------------------------------
int alfi_ctx_create(int device, void* stream, alfi_ctx** out) {
  if (!out) return alfi_set_error(nullptr, ALFI_E_ARG, "out is NULL");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return alfi_set_error(nullptr, ALFI_E_HIP, "no HIP device available");
  if (device < 0 || device >= ndev) return alfi_set_error(nullptr, ALFI_E_ARG, "device %d out of range", device);
  alfi_ctx* ctx = new alfi_ctx();
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess) {
    delete ctx;
    return alfi_set_error(nullptr, ALFI_E_HIP, "hipSetDevice(%d) failed", device);
  }
  if (stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      delete ctx;
      return alfi_set_error(nullptr, ALFI_E_HIP, "hipStreamCreate failed");
    }
    ctx->own_stream = true;
  }
  if (hipMalloc((void**)&ctx->red_partial, sizeof(double) * RED_BLOCKS * RED_MAXV) != hipSuccess ||
      hipMalloc((void**)&ctx->red_partial2, sizeof(double) * RED_BLOCKS * RED_MAXV) != hipSuccess) {
    delete ctx;
    return alfi_set_error(nullptr, ALFI_E_HIP, "hipMalloc failed");
  }
  if (hipMalloc((void**)&ctx->dev_err, 16) != hipSuccess || hipMemset(ctx->dev_err, 0, 16) != hipSuccess) {
    delete ctx;
    return alfi_set_error(nullptr, ALFI_E_HIP, "hipMalloc failed");
  }
  *out = ctx;
  return 0;
}

int alfi_ctx_destroy(alfi_ctx* ctx) {
  if (!ctx) return 0;
  (void)hipStreamSynchronize(ctx->stream);
  native_destroy(ctx);
  for (auto& p : ctx->ev_pool) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  dev_free(ctx->red_partial);
  dev_free(ctx->red_partial2);
  dev_free(ctx->dev_err);
  (void)hipFree(ctx->big_arena);
  (void)hipFree(ctx->asm_scratch);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return 0;
}

int alfi_ctx_sync(alfi_ctx* ctx) {
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return check_dev_err(ctx);
}

const char* alfi_last_error(alfi_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int alfi_malloc(alfi_ctx* ctx, int64_t bytes, void** dptr) {
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipMalloc(dptr, (size_t)(bytes > 0 ? bytes : 8)));
  return 0;
}
int alfi_free(alfi_ctx* ctx, void* dptr) {
  if (dptr) ALFI_HIP_CHECK(ctx, hipFree(dptr));
  return 0;
}
int alfi_memcpy_h2d(alfi_ctx* ctx, void* dst, const void* src, int64_t bytes) {
  ALFI_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}
int alfi_memcpy_d2h(alfi_ctx* ctx, void* dst, const void* src, int64_t bytes) {
  ALFI_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return check_dev_err(ctx);
}
int alfi_memset0(alfi_ctx* ctx, void* dst, int64_t bytes) {
  ALFI_HIP_CHECK(ctx, hipMemsetAsync(dst, 0, (size_t)bytes, ctx->stream));
  return 0;
}

int alfi_ctx_set_comm(alfi_ctx* ctx, alfi_comm_fn fn, void* user, double* dred, int64_t dred_len) {
  if (fn && (!dred || dred_len < 2 * RED_MAXV))
    return alfi_set_error(ctx, ALFI_E_ARG, "alfi_ctx_set_comm needs a device buffer of >= %d doubles", 2 * RED_MAXV);
  if (ctx->nat) return alfi_set_error(ctx, ALFI_E_STATE, "the ctx already owns a communicator (alfi_ctx_comm_init)");
  ctx->comm = fn;
  ctx->comm_user = user;
  ctx->dred = dred;
  const char* e = getenv("ALFI_DIST_EXACT_NORM");   // default: |w - V h| by its own all-reduce, as PETSc's VecNorm; 0: see comm.hip
  ctx->exact_norm = !(e && atoi(e) == 0);
  return 0;
}

int alfi_prof_enable(alfi_ctx* ctx, int on) {
  ctx->prof = on;
  return 0;
}
int alfi_prof_reset(alfi_ctx* ctx) {
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->ev_used = 0;
  return 0;
}
int alfi_prof_get(alfi_ctx* ctx, int ev, double* total_ms, int64_t* count) {
  return alfi_prof_get_level(ctx, ev, -1, total_ms, count);
}
int alfi_prof_get_level(alfi_ctx* ctx, int ev, int level_id, double* total_ms, int64_t* count) {
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  double tot = 0;
  int64_t cnt = 0;
  for (size_t i = 0; i < ctx->ev_used; ++i) {
    if (ctx->ev_pool[i].kind != ev) continue;
    if (level_id >= 0 && ctx->ev_pool[i].tag != level_id) continue;
    float ms = 0;
    ALFI_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[i].a, ctx->ev_pool[i].b));
    tot += ms;
    ++cnt;
  }
  if (total_ms) *total_ms = tot;
  if (count) *count = cnt;
  return 0;
}

// ---- level ---------------------------------------------------------------------------------------------------------------
int alfi_level_create(alfi_ctx* ctx, int64_t nbrows, int bs, const int32_t* browptr, const int32_t* bcolidx,
                      const double* bvals, const int32_t* bc_dofs, int64_t nbc, alfi_level** out) {
  // bvals == NULL: the operator starts as zeros and is formed on the device (alfi_level_set_assembly + alfi_level_assemble)
  if (!ctx || !out || !browptr || !bcolidx) return alfi_set_error(ctx, ALFI_E_ARG, "NULL argument");
  if (bs != 2 && bs != 3) return alfi_set_error(ctx, ALFI_E_ARG, "block size must be 2 or 3, got %d", bs);
  if (nbrows * bs > INT32_MAX) return alfi_set_error(ctx, ALFI_E_ARG, "level too large for int32 dof indices");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  alfi_level* L = new alfi_level();
  L->ctx = ctx;
  L->id = ctx->next_level_id++;
  L->bs = bs;
  L->n = nbrows * bs;
  alfi_bsr_host h{nbrows, nbrows, browptr, bcolidx, bvals};
  int rc = upload_bsr(ctx, &L->A, &h, bs);
  if (rc == 0) rc = dev_upload(ctx, &L->bc_dofs, bc_dofs, nbc);
  if (rc == 0) {
    std::vector<uint8_t> mask((size_t)std::max<int64_t>(L->n, 1), 0);
    for (int64_t i = 0; i < nbc && rc == 0; ++i) {
      if (bc_dofs[i] < 0 || bc_dofs[i] >= L->n) rc = alfi_set_error(ctx, ALFI_E_ARG, "Dirichlet dof %d out of range", bc_dofs[i]);
      else mask[bc_dofs[i]] = 1;
    }
    if (rc == 0) rc = dev_upload(ctx, &L->bc_mask, mask.data(), (int64_t)mask.size());
  }
  if (rc == 0) rc = dev_alloc(ctx, &L->status, 1);
  if (rc == 0 && hipMemsetAsync(L->status, 0, sizeof(int), ctx->stream) != hipSuccess) rc = ALFI_E_HIP;
  L->nbc = nbc;
  if (rc != 0) {
    alfi_level_destroy(L);
    return rc;
  }
  L->n_own = L->n;
  L->A_own = L->A;
  for (int64_t i = 0; i < nbrows; ++i) L->max_row_blocks = std::max<int>(L->max_row_blocks, browptr[i + 1] - browptr[i]);
  *out = L;
  return 0;
}

int alfi_level_set_partition(alfi_level* L, int64_t nb_owned, int distributed, int64_t nsend,
                             const int32_t* send_nodes, double* d_sendbuf, double* d_recvbuf, int64_t nb_ghost) {
  alfi_ctx* ctx = L->ctx;
  if (nb_owned < 0 || nb_ghost < 0 || nb_owned + nb_ghost != L->A.nbrows)
    return alfi_set_error(ctx, ALFI_E_ARG, "owned (%lld) + ghost (%lld) nodes != %lld block rows", (long long)nb_owned,
                          (long long)nb_ghost, (long long)L->A.nbrows);
  if (nsend < 0 || (nsend > 0 && !send_nodes) || (!d_sendbuf) != (!d_recvbuf))
    return alfi_set_error(ctx, ALFI_E_ARG, "bad halo arguments (pass both buffers or neither)");
  for (int64_t i = 0; i < nsend; ++i)
    if (send_nodes[i] < 0 || send_nodes[i] >= nb_owned)
      return alfi_set_error(ctx, ALFI_E_ARG, "send node %d is not an owned node", send_nodes[i]);
  if (L->patch_ptr || L->V) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_set_partition must precede patches");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(L->halo_send_nodes);
  dev_free(L->rev_nodes);
  dev_free(L->rev_ptr);
  dev_free(L->rev_pos);
  L->halo_send_nodes = L->rev_nodes = L->rev_ptr = L->rev_pos = nullptr;
  if (L->own_halo_bufs) {
    dev_free(L->halo_sendbuf);
    dev_free(L->halo_recvbuf);
  }
  L->halo_sendbuf = L->halo_recvbuf = nullptr;
  L->own_halo_bufs = false;
  L->nbr_rank.clear();
  if (!d_sendbuf) {                       // the library's own buffers (native transport)
    ALFI_CHECK(dev_alloc(ctx, &d_sendbuf, nsend * L->bs));
    ALFI_CHECK(dev_alloc(ctx, &d_recvbuf, nb_ghost * L->bs));
    L->own_halo_bufs = true;
  }
  // reverse-add plan: positions of the send buffer grouped by node, in buffer order (fixed summation order)
  std::vector<int32_t> order(nsend);
  for (int64_t i = 0; i < nsend; ++i) order[i] = (int32_t)i;
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return send_nodes[a] < send_nodes[b]; });
  std::vector<int32_t> rev_nodes, rev_ptr;
  for (int64_t i = 0; i < nsend; ++i) {
    if (i == 0 || send_nodes[order[i]] != send_nodes[order[i - 1]]) {
      rev_nodes.push_back(send_nodes[order[i]]);
      rev_ptr.push_back((int32_t)i);
    }
  }
  rev_ptr.push_back((int32_t)nsend);
  ALFI_CHECK(dev_upload(ctx, &L->halo_send_nodes, send_nodes, nsend));
  ALFI_CHECK(dev_upload(ctx, &L->rev_nodes, rev_nodes.data(), (int64_t)rev_nodes.size()));
  ALFI_CHECK(dev_upload(ctx, &L->rev_ptr, rev_ptr.data(), (int64_t)rev_ptr.size()));
  ALFI_CHECK(dev_upload(ctx, &L->rev_pos, order.data(), nsend));
  L->rev_nuniq = (int64_t)rev_nodes.size();
  L->halo_nsend = nsend;
  L->halo_nghost = nb_ghost;
  L->halo_sendbuf = d_sendbuf;
  L->halo_recvbuf = d_recvbuf;
  L->has_halo = true;
  L->distributed = distributed != 0;
  L->n_own = nb_owned * L->bs;
  int32_t nnz_own = 0;
  ALFI_HIP_CHECK(ctx, hipMemcpy(&nnz_own, L->A.rowptr + nb_owned, sizeof(int32_t), hipMemcpyDeviceToHost));
  int64_t nchunks_own = -1;
  if (L->A.aligned) {      // whole-row chunks: a chunk boundary in front of the first ghost row
    const int64_t nb = L->A.nbrows;
    std::vector<int32_t> rp(nb + 1);
    ALFI_HIP_CHECK(ctx, hipMemcpy(rp.data(), L->A.rowptr, sizeof(int32_t) * (nb + 1), hipMemcpyDeviceToHost));
    ALFI_CHECK(build_chunk_tables(ctx, &L->A, rp.data(), nb, nb_owned, &nchunks_own));
  }
  L->A_own = L->A;
  L->A_own.nbrows = nb_owned;
  L->A_own.nnzb = nnz_own;
  if (L->A.aligned) L->A_own.nchunks = nchunks_own;
  return 0;
}

// block-row range [r0, r1) of the flat upload as a view with its own chunk table and carry arrays
static int make_row_view(alfi_ctx* ctx, const DevBSR& A, const std::vector<int32_t>& rowptr, int64_t r0, int64_t r1,
                         DevBSR* V) {
  *V = A;
  V->view = true;
  V->aligned = false;                   // a view brings its own equal-sized chunks (+ fix-up)
  V->chunk_start = nullptr;
  V->nbrows = r1;                       // rows are addressed absolutely; nbrows only has to cover the range
  V->kbase = rowptr[r0];
  V->nnzb = rowptr[r1] - rowptr[r0];
  V->nchunks = (V->nnzb + SPMV_CHUNK - 1) / SPMV_CHUNK;
  V->chunk_row = nullptr;
  V->carry = nullptr;
  V->carry_row = nullptr;
  V->dedup = false;                     // the de-duplication groups are counted from block 0 of the upload
  std::vector<int32_t> chunk_row((size_t)std::max<int64_t>(V->nchunks, 1));
  int64_t row = r0;
  for (int64_t c = 0; c < V->nchunks; ++c) {
    const int64_t k = V->kbase + c * SPMV_CHUNK;
    while (rowptr[row + 1] <= k) ++row;
    chunk_row[c] = (int32_t)row;
  }
  ALFI_CHECK(dev_upload(ctx, &V->chunk_row, chunk_row.data(), V->nchunks));
  ALFI_CHECK(dev_alloc(ctx, &V->carry, V->nchunks * A.bs));
  ALFI_CHECK(dev_alloc(ctx, &V->carry_row, V->nchunks));
  return 0;
}
static void free_row_view(DevBSR* V) {
  dev_free(V->chunk_row);
  dev_free(V->carry);
  dev_free(V->carry_row);
  *V = DevBSR();
}

int alfi_level_set_overlap(alfi_level* L, int64_t nb_interior, int64_t npatch_interior) {
  alfi_ctx* ctx = L->ctx;
  const int64_t nb_own = L->n_own / L->bs;
  if (!L->has_halo || !L->distributed)
    return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_set_overlap needs a distributed level (alfi_level_set_partition)");
  if (nb_interior < 0 || nb_interior > nb_own || npatch_interior < 0 || npatch_interior > L->npatch)
    return alfi_set_error(ctx, ALFI_E_ARG, "interior counts out of range");
  if (!L->A.flat) return 0;   // row-per-lane-group layout: no row-range views; keep the plain exchange
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  // verify the claims: interior rows hold owned columns only, interior patches hold owned dofs only
  const int64_t nb = L->A.nbrows;
  std::vector<int32_t> rowptr(nb + 1);
  ALFI_HIP_CHECK(ctx, hipMemcpy(rowptr.data(), L->A.rowptr, sizeof(int32_t) * (nb + 1), hipMemcpyDeviceToHost));
  const int64_t kint = rowptr[nb_interior];
  if (kint > 0) {
    std::vector<int32_t> col((size_t)kint);
    ALFI_HIP_CHECK(ctx, hipMemcpy(col.data(), L->A.colidx, sizeof(int32_t) * kint, hipMemcpyDeviceToHost));
    for (int64_t k = 0; k < kint; ++k)
      if ((col[k] & 0x7fffffff) >= nb_own)
        return alfi_set_error(ctx, ALFI_E_ARG, "an operator row declared interior holds a ghost column");
  }
  for (int64_t p = 0; p < npatch_interior; ++p)
    for (int64_t q = L->h_patch_ptr[p]; q < L->h_patch_ptr[p + 1]; ++q)
      if (L->h_patch_dofs[q] >= L->n_own)
        return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld declared interior holds a ghost dof", (long long)p);
  free_row_view(&L->A_int);
  free_row_view(&L->A_bnd);
  ALFI_CHECK(make_row_view(ctx, L->A, rowptr, 0, nb_interior, &L->A_int));
  ALFI_CHECK(make_row_view(ctx, L->A, rowptr, nb_interior, nb_own, &L->A_bnd));
  L->npatch_int = npatch_interior;
  L->overlap = true;
  return 0;
}

int alfi_level_destroy(alfi_level* L) {
  if (!L) return 0;
  (void)hipStreamSynchronize(L->ctx->stream);
  free_row_view(&L->A_int);
  free_row_view(&L->A_bnd);
  free_assembly(&L->asmb);
  free_bsr(&L->A);
  dev_free(L->bc_dofs);
  dev_free(L->bc_mask);
  dev_free(L->halo_send_nodes);
  if (L->own_halo_bufs) {
    dev_free(L->halo_sendbuf);
    dev_free(L->halo_recvbuf);
  }
  dev_free(L->rev_nodes);
  dev_free(L->rev_ptr);
  dev_free(L->rev_pos);
  dev_free(L->patch_ptr);
  dev_free(L->patch_dofs);
  dev_free(L->inv_ptr);
  dev_free(L->stage_ptr);
  dev_free(L->inv);
  dev_free(L->inv_il);
  dev_free(L->stage);
  dev_free(L->dof_ptr);
  dev_free(L->dof_pos);
  dev_free(L->mult_seq);
  free_mult_schedule(L);
  dev_free(L->status);
  dev_free(L->chk);
  dev_free(L->chk_list);
  free_cond(L);
  dev_free(L->V);
  dev_free(L->Z);
  dev_free(L->w);
  dev_free(L->hs);
  if (L->cinv_owned) dev_free(L->cinv);
  mf_free(L->mf);
  dev_free(L->mg_b);
  dev_free(L->mg_x);
  dev_free(L->mg_r);
  delete L;
  return 0;
}

int alfi_level_update_values(alfi_level* L, const double* bvals) {
  alfi_ctx* ctx = L->ctx;
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ALFI_CHECK(upload_bsr_values(ctx, &L->A, bvals));
  L->A_own.vals = L->A.vals;
  L->A_int.vals = L->A_bnd.vals = L->A.vals;
  L->factored = false;
  return 0;
}

static void free_assembly(AssemblyDev* S) {
  dev_free(S->cptr); dev_free(S->ccell); dev_free(S->cba); dev_free(S->cell_nodes); dev_free(S->grad); dev_free(S->vol);
  dev_free(S->etab); dev_free(S->bItab); dev_free(S->bc_code); dev_free(S->bc_all);
  dev_free(S->wq); dev_free(S->phi); dev_free(S->dphi); dev_free(S->d2phi); dev_free(S->hcell); dev_free(S->diag);
  dev_free(S->wq8); dev_free(S->qtab);
  *S = AssemblyDev();
}

int alfi_level_set_assembly(alfi_level* L, int64_t ncell, int nloc, const int32_t* cell_nodes, const double* grad,
                            const double* vol, const double* Sref, const double* bIref, const double* T1ref, int full_div,
                            const int64_t* cptr, const int32_t* ccell, const uint16_t* cba) {
  alfi_ctx* ctx = L->ctx;
  if (!cell_nodes || !grad || !vol || !Sref || !bIref || !T1ref || !cptr || !ccell || !cba)
    return alfi_set_error(ctx, ALFI_E_ARG, "NULL argument");
  if (ncell < 1 || nloc < 1 || nloc * nloc > 65535) return alfi_set_error(ctx, ALFI_E_ARG, "bad cell counts (%lld cells, %d nodes each)", (long long)ncell, nloc);
  if (!L->A.flat) return alfi_set_error(ctx, ALFI_E_STATE, "device assembly needs the lane-major operator layout (no empty block rows)");
  if (!element_kernel_exists(L->bs, nloc))
    return alfi_set_error(ctx, ALFI_E_ARG, "no element kernel for %d nodes per cell in %d-D", nloc, L->bs);
  const int64_t nnzb = L->A.nnzb, nb = L->A.nbrows;
  const int d = L->bs, nv = d + 1;
  // Unpartitioned: the cells are the mesh, every (cell, a, b) pair contributes to a block.  Partitioned: the cells that touch
  // a local node; their other nodes follow the local ones in the numbering of the state vector, and only the pairs with both
  // nodes local contribute (the ghost rows of the local operator are restricted to local columns).
  const bool part = L->has_halo;
  const int64_t npairs = cptr[nnzb];
  if (cptr[0] != 0 || (part ? npairs > ncell * nloc * nloc : npairs != ncell * nloc * nloc))
    return alfi_set_error(ctx, ALFI_E_ARG, "contributor lists hold %lld pairs, expected %scells x nodes^2 = %lld", (long long)npairs,
                          part ? "at most " : "", (long long)(ncell * nloc * nloc));
  // (the checks below walk 200 M contributors at config-4 size: over host threads, each reporting its first finding)
  std::atomic<int64_t> bad_block(-1), bad_pair(-1), bad_node(-1), max_node(nb - 1);
  host_parallel_ranges(nnzb, [&](int64_t k0, int64_t k1) {
    for (int64_t k = k0; k < k1; ++k)
      if (cptr[k + 1] <= cptr[k]) { bad_block = k; return; }
  });
  if (bad_block >= 0) return alfi_set_error(ctx, ALFI_E_ARG, "block %lld has no contributing cell", (long long)bad_block.load());
  host_parallel_ranges(npairs, [&](int64_t q0, int64_t q1) {
    for (int64_t q = q0; q < q1; ++q)
      if (ccell[q] < 0 || ccell[q] >= ncell || cba[q] >= nloc * nloc) { bad_pair = q; return; }
  });
  if (bad_pair >= 0) return alfi_set_error(ctx, ALFI_E_ARG, "contributor %lld out of range", (long long)bad_pair.load());
  host_parallel_ranges(ncell * nloc, [&](int64_t i0, int64_t i1) {
    int64_t m = 0;
    for (int64_t i = i0; i < i1; ++i) {
      if (cell_nodes[i] < 0 || (!part && cell_nodes[i] >= nb)) { bad_node = i; return; }
      m = std::max<int64_t>(m, cell_nodes[i]);
    }
    int64_t cur = max_node.load();
    while (m > cur && !max_node.compare_exchange_weak(cur, m)) {}
  });
  if (bad_node >= 0) return alfi_set_error(ctx, ALFI_E_ARG, "cell node out of range");
  const int64_t nstate = max_node.load() + 1;
  // the diagonal block of every block row (its contributor list = the cells around the node: the gather of element vectors);
  // the two nodes of a contributing pair are rows / columns of the local operator
  std::vector<int32_t> diag((size_t)nb, -1);
  {
    std::vector<int32_t> rowptr(nb + 1), colidx((size_t)std::max<int64_t>(nnzb, 1));
    ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ALFI_HIP_CHECK(ctx, hipMemcpy(rowptr.data(), L->A.rowptr, sizeof(int32_t) * (nb + 1), hipMemcpyDeviceToHost));
    ALFI_HIP_CHECK(ctx, hipMemcpy(colidx.data(), L->A.colidx, sizeof(int32_t) * nnzb, hipMemcpyDeviceToHost));
    std::atomic<int64_t> stray(-1), stray_block(-1), no_diag(-1);
    host_parallel_ranges(nb, [&](int64_t r0, int64_t r1) {
      for (int64_t r = r0; r < r1; ++r) {
        for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
          const int32_t c = colidx[k] & 0x7fffffff;       // (the sign bit marks the first block of a block row)
          if (c == r) diag[r] = (int32_t)k;
          for (int64_t q = cptr[k]; q < cptr[k + 1]; ++q) {
            const int32_t* cn = cell_nodes + (int64_t)ccell[q] * nloc;
            if (cn[cba[q] % nloc] != r || cn[cba[q] / nloc] != c) { stray = q; stray_block = k; return; }
          }
        }
        if (diag[r] < 0) { no_diag = r; return; }
      }
    });
    if (stray >= 0)
      return alfi_set_error(ctx, ALFI_E_ARG, "contributor %lld does not belong to block %lld", (long long)stray.load(),
                            (long long)stray_block.load());
    if (no_diag >= 0) return alfi_set_error(ctx, ALFI_E_ARG, "block row %lld has no diagonal block", (long long)no_diag.load());
  }
  // per (a, b) the slices of the reference tensors the cell kernel reads (kernels_assemble.hip):
  // T1[k,i,b,a] | T1[b,i,k,a] | S[a,b,i,j];  T1: (nloc, d+1, nloc, nloc), S: (nloc, nloc, d+1, d+1)
  const int ts = 2 * nv * nloc + nv * nv;
  std::vector<double> etab((size_t)nloc * nloc * ts);
  for (int a = 0; a < nloc; ++a)
    for (int b = 0; b < nloc; ++b) {
      double* t = etab.data() + (size_t)(a * nloc + b) * ts;
      for (int i = 0; i < nv; ++i)
        for (int k = 0; k < nloc; ++k) {
          t[i * nloc + k] = T1ref[(((size_t)k * nv + i) * nloc + b) * nloc + a];
          t[nv * nloc + i * nloc + k] = T1ref[(((size_t)b * nv + i) * nloc + k) * nloc + a];
        }
      for (int i = 0; i < nv; ++i)
        for (int j = 0; j < nv; ++j) t[2 * nv * nloc + i * nv + j] = Sref[(((size_t)a * nloc + b) * nv + i) * nv + j];
    }
  free_assembly(&L->asmb);
  AssemblyDev S;
  S.nloc = nloc;
  S.ncell = ncell;
  S.npairs = npairs;
  S.nstate = nstate;
  S.full_div = full_div != 0;
  int rc = dev_upload(ctx, &S.cptr, cptr, nnzb + 1);
  if (rc == 0) rc = dev_upload(ctx, &S.ccell, ccell, npairs);
  if (rc == 0) rc = dev_upload(ctx, &S.cba, cba, npairs);
  if (rc == 0) rc = dev_upload(ctx, &S.cell_nodes, cell_nodes, ncell * nloc);
  if (rc == 0) rc = dev_upload(ctx, &S.grad, grad, ncell * nv * d);
  if (rc == 0) rc = dev_upload(ctx, &S.vol, vol, ncell);
  if (rc == 0) rc = dev_upload(ctx, &S.etab, etab.data(), (int64_t)etab.size());
  if (rc == 0) rc = dev_upload(ctx, &S.bItab, bIref, (int64_t)nloc * nv);
  if (rc == 0) rc = dev_upload(ctx, &S.diag, diag.data(), nb);
  if (rc != 0) {
    free_assembly(&S);
    return rc;
  }
  S.ready = true;
  L->asmb = S;
  return 0;
}

int alfi_ctx_set_assembly_scratch(alfi_ctx* ctx, int64_t max_bytes) {
  if (max_bytes < 1) return alfi_set_error(ctx, ALFI_E_ARG, "scratch limit must be positive");
  ctx->asm_scratch_limit = max_bytes;
  return 0;
}

int alfi_level_assemble(alfi_level* L, double nu, double gamma, double adv, const double* d_state, int apply_bc) {
  alfi_ctx* ctx = L->ctx;
  if (!L->asmb.ready) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_assemble before alfi_level_set_assembly");
  if (adv != 0.0 && !d_state) return alfi_set_error(ctx, ALFI_E_ARG, "advection needs the state");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ctx->cur_tag = L->id;
  int t = alfi_prof_begin(ctx, ALFI_EV_PATCH_FACTOR);       // PCPatchComputeOp
  ALFI_CHECK(launch_operator_refresh(L, nu, gamma, adv, d_state, true, false, 0.0, 0.0, false, apply_bc != 0, L->A.vals));
  alfi_prof_end(ctx, t);
  L->factored = false;
  return 0;
}

// The refresh of a stabilised run in one pass: A = nu K + gamma D + adv N(state) + the linearised SUPG term, boundary conditions
int alfi_level_assemble_supg(alfi_level* L, double nu, double gamma, double adv, const double* d_state, double weight,
                             double magic, int apply_bc) {
  alfi_ctx* ctx = L->ctx;
  if (!L->asmb.supg_ready) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_assemble_supg before alfi_level_set_supg");
  if (!d_state) return alfi_set_error(ctx, ALFI_E_ARG, "SUPG needs the state");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ctx->cur_tag = L->id;
  int t = alfi_prof_begin(ctx, ALFI_EV_PATCH_FACTOR);
  ALFI_CHECK(launch_operator_refresh(L, nu, gamma, adv, d_state, true, true, weight, magic, false, apply_bc != 0, L->A.vals));
  alfi_prof_end(ctx, t);
  L->factored = false;
  return 0;
}

static int level_spmv(alfi_level* L, const double* dx, double* dy, const double* db, int mode, bool ghosts_current = false);

// Partitioned levels: the Dirichlet dofs among ALL local dofs (the level was created with the owned ones only: its smoother
// copies x to y there, which must happen on the owner alone).  The refresh turns the rows / columns of these dofs into
// identity, as firedrake.assemble(a, bcs) does on every rank's rows [3P].
int alfi_level_set_assembly_bc(alfi_level* L, const int32_t* bc_dofs, int64_t nbc) {
  alfi_ctx* ctx = L->ctx;
  if (!L->asmb.ready) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_set_assembly_bc before alfi_level_set_assembly");
  if (nbc < 0 || (nbc > 0 && !bc_dofs)) return alfi_set_error(ctx, ALFI_E_ARG, "NULL argument");
  std::vector<uint8_t> mask((size_t)std::max<int64_t>(L->n, 1), 0);
  for (int64_t i = 0; i < nbc; ++i) {
    if (bc_dofs[i] < 0 || bc_dofs[i] >= L->n) return alfi_set_error(ctx, ALFI_E_ARG, "Dirichlet dof %d out of range", bc_dofs[i]);
    mask[(size_t)bc_dofs[i]] = 1;
  }
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(L->asmb.bc_all);
  dev_free(L->asmb.bc_code);
  L->asmb.bc_all = nullptr;
  L->asmb.bc_code = nullptr;
  return dev_upload(ctx, &L->asmb.bc_all, mask.data(), (int64_t)mask.size());
}

int alfi_level_assembly_state_size(alfi_level* L, int64_t* n) {
  if (!L->asmb.ready) return alfi_set_error(L->ctx, ALFI_E_STATE, "alfi_level_assembly_state_size before alfi_level_set_assembly");
  *n = L->asmb.nstate * L->bs;
  return 0;
}

// y = A(state) x with A = nu K + gamma D + adv N(state) WITHOUT boundary conditions, matrix-free: every cell multiplies its
// element matrix with its entries of x as it forms it, the element vectors are gathered per node in a fixed order.  The level's
// own operator (the Jacobian the patches were factored from) is not touched.  The nonlinear residual of
// alfi/solver.py:565-568 is one such product: F_u = (nu K + gamma D) u + 1/2 N(u) u = A(u; adv / 2) u.
int alfi_level_assemble_mult(alfi_level* L, double nu, double gamma, double adv, const double* d_state, const double* dx,
                             double* dy) {
  alfi_ctx* ctx = L->ctx;
  AssemblyDev& S = L->asmb;
  if (!S.ready) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_assemble_mult before alfi_level_set_assembly");
  if (adv != 0.0 && !d_state) return alfi_set_error(ctx, ALFI_E_ARG, "advection needs the state");
  if (!dx || !dy) return alfi_set_error(ctx, ALFI_E_ARG, "NULL argument");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ctx->cur_tag = L->id;
  int t = alfi_prof_begin(ctx, ALFI_EV_PATCH_FACTOR);
  ALFI_CHECK(launch_element_mult(L, nu, gamma, adv, d_state, dx, dy));
  alfi_prof_end(ctx, t);
  return 0;
}

int alfi_level_set_supg(alfi_level* L, int nq, const double* wq, const double* phi, const double* dphi, const double* d2phi,
                        const double* hcell) {
  alfi_ctx* ctx = L->ctx;
  AssemblyDev& S = L->asmb;
  if (!S.ready) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_set_supg before alfi_level_set_assembly");
  if (nq < 1 || !wq || !phi || !dphi || !d2phi || !hcell) return alfi_set_error(ctx, ALFI_E_ARG, "NULL / empty SUPG tables");
  if (S.nloc > 16) return alfi_set_error(ctx, ALFI_E_ARG, "SUPG kernels handle elements of at most 16 nodes, got %d", S.nloc);
  for (int64_t c = 0; c < S.ncell; ++c)
    if (!(hcell[c] > 0.0)) return alfi_set_error(ctx, ALFI_E_ARG, "cell size of cell %lld is not positive", (long long)c);
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(S.wq); dev_free(S.phi); dev_free(S.dphi); dev_free(S.d2phi); dev_free(S.hcell); dev_free(S.wq8); dev_free(S.qtab);
  S.wq = S.phi = S.dphi = S.d2phi = S.hcell = S.wq8 = S.qtab = nullptr;
  S.supg_ready = false;
  const int nv = L->bs + 1, nloc = S.nloc;
  ALFI_CHECK(dev_upload(ctx, &S.wq, wq, nq));
  ALFI_CHECK(dev_upload(ctx, &S.phi, phi, (int64_t)nq * nloc));
  ALFI_CHECK(dev_upload(ctx, &S.dphi, dphi, (int64_t)nq * nloc * nv));
  ALFI_CHECK(dev_upload(ctx, &S.d2phi, d2phi, (int64_t)nq * nloc * nv * nv));
  ALFI_CHECK(dev_upload(ctx, &S.hcell, hcell, S.ncell));
  {
    // the linearisation kernel takes eight points per chunk (two matrix-core steps of four): pad with zero-weight copies of point 0, and pack per
    // (point, node) phi | dphi | the upper triangle of the (symmetric) second derivatives
    const int nq8 = (nq + 7) / 8 * 8, nh2 = nv * (nv + 1) / 2, qt = 1 + nv + nh2;
    std::vector<double> w8((size_t)nq8, 0.0), tab((size_t)nq8 * nloc * qt);
    for (int p = 0; p < nq8; ++p) {
      const int ps = p < nq ? p : 0;
      if (p < nq) w8[p] = wq[p];
      for (int a = 0; a < nloc; ++a) {
        double* t = tab.data() + ((size_t)p * nloc + a) * qt;
        t[0] = phi[(size_t)ps * nloc + a];
        for (int m = 0; m < nv; ++m) t[1 + m] = dphi[((size_t)ps * nloc + a) * nv + m];
        int at = 1 + nv;
        for (int m = 0; m < nv; ++m)
          for (int n = m; n < nv; ++n) t[at++] = d2phi[(((size_t)ps * nloc + a) * nv + m) * nv + n];
      }
    }
    ALFI_CHECK(dev_upload(ctx, &S.wq8, w8.data(), nq8));
    ALFI_CHECK(dev_upload(ctx, &S.qtab, tab.data(), (int64_t)tab.size()));
    S.nq8 = nq8;
  }
  S.nq = nq;
  S.supg_ready = true;
  return 0;
}

int alfi_level_supg(alfi_level* L, double nu, double weight, double magic, const double* d_state, int add_to_operator, double* d_F) {
  alfi_ctx* ctx = L->ctx;
  if (!L->asmb.supg_ready) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_supg before alfi_level_set_supg");
  if (!d_state) return alfi_set_error(ctx, ALFI_E_ARG, "SUPG needs the state");
  if (!add_to_operator && !d_F) return 0;
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ctx->cur_tag = L->id;
  int t = alfi_prof_begin(ctx, ALFI_EV_PATCH_FACTOR);       // PCPatchComputeOp
  if (add_to_operator)
    ALFI_CHECK(launch_operator_refresh(L, nu, 0.0, 0.0, d_state, false, true, weight, magic, true, false, L->A.vals));
  if (d_F) ALFI_CHECK(launch_supg_residual(L, nu, weight, magic, d_state, d_F));
  alfi_prof_end(ctx, t);
  if (add_to_operator) L->factored = false;
  return 0;
}

int alfi_level_apply_bc(alfi_level* L) {
  if (!L->asmb.ready) return alfi_set_error(L->ctx, ALFI_E_STATE, "alfi_level_apply_bc before alfi_level_set_assembly");
  L->factored = false;
  return launch_apply_bc(L);
}

int alfi_level_get_values(alfi_level* L, double* bvals) {
  alfi_ctx* ctx = L->ctx;
  const int bb = L->bs * L->bs;
  if (L->A.nnzb == 0) return 0;
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!L->A.flat) {
    ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ALFI_HIP_CHECK(ctx, hipMemcpy(bvals, L->A.vals, (size_t)L->A.nnzb * bb * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
  }
  double* tmp = nullptr;
  ALFI_CHECK(dev_alloc(ctx, &tmp, L->A.nnzb * bb));
  int rc = launch_vals_from_lanes(ctx, L->A, tmp);
  if (rc == 0 && hipMemcpyAsync(bvals, tmp, (size_t)L->A.nnzb * bb * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = ALFI_E_HIP;
  if (rc == 0 && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ALFI_E_HIP;
  dev_free(tmp);
  return rc;
}

int alfi_level_size(alfi_level* L, int64_t* n) {
  *n = L->n;
  return 0;
}
int alfi_level_id(alfi_level* L, int* id) {
  *id = L->id;
  return 0;
}

// Partitioned levels: x's ghost slots are refreshed from their owners first (they are scratch: see alfi_hip.h), the
// product is formed on the owned rows.
// y = A x (mode 0) or y = b - A x (mode 1) on the owned rows of a level.  With alfi_level_set_overlap the rows without
// ghost columns are multiplied while the forward halo of x is in flight.
static int level_spmv(alfi_level* L, const double* dx, double* dy, const double* db, int mode, bool ghosts_current) {
  alfi_ctx* ctx = L->ctx;
  ctx->cur_tag = L->id;
  int t;
  if (L->distributed && L->overlap && !ghosts_current) {
    ALFI_CHECK(halo_fwd_begin(L, dx));
    t = alfi_prof_begin(ctx, ALFI_EV_MATMULT);
    ALFI_CHECK(launch_bsr_spmv(ctx, L->A_int, dx, dy, db, 1.0, mode));
    alfi_prof_end(ctx, t);
    ALFI_CHECK(halo_fwd_end(L, const_cast<double*>(dx)));
    t = alfi_prof_begin(ctx, ALFI_EV_MATMULT);
    ALFI_CHECK(launch_bsr_spmv(ctx, L->A_bnd, dx, dy, db, 1.0, mode));
    alfi_prof_end(ctx, t);
    return 0;
  }
  if (L->distributed && !ghosts_current) ALFI_CHECK(halo_fwd(L, const_cast<double*>(dx)));
  t = alfi_prof_begin(ctx, ALFI_EV_MATMULT);
  ALFI_CHECK(launch_bsr_spmv(ctx, L->A_own, dx, dy, db, 1.0, mode));
  alfi_prof_end(ctx, t);
  return 0;
}

int alfi_spmv(alfi_level* L, const double* dx, double* dy) { return level_spmv(L, dx, dy, nullptr, 0); }

int alfi_residual(alfi_level* L, const double* db, const double* dx, double* dr) { return level_spmv(L, dx, dr, db, 1); }

// PCApply_PATCH on a (possibly partitioned) level: ghost values in, local patch solves, ghost contributions back to
// their owners, Dirichlet dofs copied
// ghosts_current (smoother only): with a sum-exchange plan the one exchange after the local solves leaves the total on the
// ghost copies too; *ghosts_current tells the caller that the product A y needs no forward exchange
static int level_patch_apply(alfi_level* L, const double* dx, double* dy, bool* ghosts_current = nullptr) {
  alfi_ctx* ctx = L->ctx;
  if (ghosts_current) *ghosts_current = false;
  if (L->mult) {
    // multiplicative sweep (PCApply_PATCH, local_type multiplicative [3P]): y = 0, then wavefront by wavefront in
    // iteration order and, with symmetrise_sweep, back again in reverse order
    if (L->distributed) ALFI_CHECK(halo_fwd(L, const_cast<double*>(dx)));
    ALFI_HIP_CHECK(ctx, hipMemsetAsync(dy, 0, sizeof(double) * L->n, ctx->stream));
    int t = alfi_prof_begin(ctx, ALFI_EV_PATCH_APPLY);
    const int64_t nw = (int64_t)L->mult_wave_ptr.size() - 1;
    // ALFI_MULT_PERSISTENT=0: one launch per dependency wavefront (the schedule of rounds 1-3; kept for the bitwise comparison)
    static const bool persistent = !(getenv("ALFI_MULT_PERSISTENT") && atoi(getenv("ALFI_MULT_PERSISTENT")) == 0);
    if (persistent && L->mult_nitems > 0) {
      // (a wait that runs into its bound sets the ctx's sticky error word, reported by the next synchronising call: no host
      // synchronisation inside the smoother)
      ALFI_CHECK(launch_patch_mult_persistent(L, dx, dy));
      alfi_prof_end(ctx, t);
      if (L->distributed) ALFI_CHECK(halo_rev(L, dy));
      if (L->nbc > 0) {
        t = alfi_prof_begin(ctx, ALFI_EV_PATCH_SCATTER);
        ALFI_CHECK(launch_copy_dofs(ctx, dy, dx, L->bc_dofs, L->nbc));
        alfi_prof_end(ctx, t);
      }
      return 0;
    }
    for (int64_t w = 0; w < nw; ++w)
      ALFI_CHECK(launch_patch_mult_wave(L, L->mult_seq + L->mult_wave_ptr[w], L->mult_wave_ptr[w + 1] - L->mult_wave_ptr[w],
                                        dx, dy));
    if (L->mult_symmetrise)
      for (int64_t w = nw - 1; w >= 0; --w)
        ALFI_CHECK(launch_patch_mult_wave(L, L->mult_seq + L->mult_wave_ptr[w],
                                          L->mult_wave_ptr[w + 1] - L->mult_wave_ptr[w], dx, dy));
    alfi_prof_end(ctx, t);
    if (L->distributed) ALFI_CHECK(halo_rev(L, dy));
    if (L->nbc > 0) {
      t = alfi_prof_begin(ctx, ALFI_EV_PATCH_SCATTER);
      ALFI_CHECK(launch_copy_dofs(ctx, dy, dx, L->bc_dofs, L->nbc));
      alfi_prof_end(ctx, t);
    }
    return 0;
  }
  if (L->distributed && L->overlap) {
    // Both exchanges hidden behind the patches that hold no ghost dof: half of them run while the forward halo of x is
    // in flight; then the patches with ghost dofs (the only ones contributing to ghost slots), the sums on the ghost
    // slots, and the reverse exchange starts; the other half of the interior patches and the sums on the owned dofs run
    // while it is in flight.
    const int64_t half = L->npatch_int / 2;
    ALFI_CHECK(halo_fwd_begin(L, dx));
    ALFI_CHECK(launch_patch_apply_range(L, 0, half, dx));
    ALFI_CHECK(halo_fwd_end(L, const_cast<double*>(dx)));
    ALFI_CHECK(launch_patch_apply_range(L, L->npatch_int, L->npatch, dx));
    ALFI_CHECK(launch_patch_sum_range(L, L->n_own, L->n, dx, dy));
    ALFI_CHECK(halo_rev_begin(L, dy));
    ALFI_CHECK(launch_patch_apply_range(L, half, L->npatch_int, dx));
    ALFI_CHECK(launch_patch_sum_range(L, 0, L->n_own, dx, dy));
    return halo_rev_end(L, dy);
  }
  if (L->distributed) ALFI_CHECK(halo_fwd(L, const_cast<double*>(dx)));
  ALFI_CHECK(launch_patch_apply(L, dx, dy));    // includes y[bc] = x[bc] (no patch holds a Dirichlet dof, so the
                                                // exchange brings nothing to those entries)
  if (L->distributed && ghosts_current && L->sum_ready && !L->pou) {
    ALFI_CHECK(halo_sum(L, dy));
    *ghosts_current = true;
  } else if (L->distributed) {
    ALFI_CHECK(halo_rev(L, dy));
  }
  return 0;
}

// ---- patches -------------------------------------------------------------------------------------------------------------
int alfi_patches_set(alfi_level* L, int64_t npatch, const int64_t* pptr, const int32_t* pdofs) {
  alfi_ctx* ctx = L->ctx;
  if (npatch < 0 || (npatch > 0 && (!pptr || !pdofs))) return alfi_set_error(ctx, ALFI_E_ARG, "NULL patch arrays");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(L->patch_ptr);
  dev_free(L->patch_dofs);
  dev_free(L->inv_ptr);
  dev_free(L->stage_ptr);
  dev_free(L->inv);
  dev_free(L->inv_il);
  L->inv_il = nullptr;
  L->il_doubles = 0;
  L->il_valid = false;
  dev_free(L->stage);
  dev_free(L->dof_ptr);
  dev_free(L->dof_pos);
  L->patch_ptr = nullptr; L->patch_dofs = nullptr; L->inv_ptr = nullptr; L->stage_ptr = nullptr;
  L->inv = nullptr; L->stage = nullptr; L->dof_ptr = nullptr; L->dof_pos = nullptr;
  L->factored = false;
  free_cond(L);
  L->inv_shrunk = false;
  L->npatch = npatch;
  const int64_t sum_n = npatch > 0 ? pptr[npatch] : 0;
  if (sum_n > INT32_MAX) return alfi_set_error(ctx, ALFI_E_ARG, "too many patch dofs for int32 staging indices");
  std::vector<int64_t> inv_ptr(npatch + 1), stage_ptr(npatch + 1);
  int64_t ip = 0, sp = 0, sum_n2 = 0;
  int max_np = 0;
  for (int64_t p = 0; p < npatch; ++p) {
    const int64_t n = pptr[p + 1] - pptr[p];
    if (n <= 0 || n > PATCH_MAX)
      return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld has %lld dofs; supported range is 1..%d", (long long)p,
                            (long long)n, PATCH_MAX);
    for (int64_t q = pptr[p]; q < pptr[p + 1]; ++q) {
      if (pdofs[q] < 0 || pdofs[q] >= L->n)
        return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: dof %d out of range", (long long)p, pdofs[q]);
      if (q > pptr[p] && pdofs[q] <= pdofs[q - 1])
        return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: dofs must be strictly ascending", (long long)p);
    }
    const int64_t ld = (n + 1) & ~(int64_t)1;
    inv_ptr[p] = ip;
    stage_ptr[p] = sp;
    ip += (n * ld + 15) & ~(int64_t)15;
    sp += ld;
    sum_n2 += n * n;
    max_np = std::max<int>(max_np, (int)n);
  }
  inv_ptr[npatch] = ip;
  stage_ptr[npatch] = sp;
  if (sp > INT32_MAX) return alfi_set_error(ctx, ALFI_E_ARG, "staging buffer exceeds int32 indexing");
  L->sum_n = sum_n;
  L->sum_n2 = sum_n2;
  L->max_np = max_np;
  L->inv_doubles = ip;
  L->stage_len = sp;
  // dof -> staged positions (counting sort; patch order = fixed summation order)
  std::vector<int32_t> dof_ptr(L->n + 1, 0), dof_pos(sum_n > 0 ? sum_n : 1);
  for (int64_t q = 0; q < sum_n; ++q) dof_ptr[pdofs[q] + 1]++;
  for (int64_t i = 0; i < L->n; ++i) dof_ptr[i + 1] += dof_ptr[i];
  {
    std::vector<int32_t> fill(dof_ptr.begin(), dof_ptr.end() - 1);
    for (int64_t p = 0; p < npatch; ++p)
      for (int64_t q = pptr[p]; q < pptr[p + 1]; ++q)
        dof_pos[fill[pdofs[q]]++] = (int32_t)(stage_ptr[p] + (q - pptr[p]));
  }
  ALFI_CHECK(dev_upload(ctx, &L->patch_ptr, pptr, npatch + 1));
  ALFI_CHECK(dev_upload(ctx, &L->patch_dofs, pdofs, sum_n));
  ALFI_CHECK(dev_upload(ctx, &L->inv_ptr, inv_ptr.data(), npatch + 1));
  ALFI_CHECK(dev_upload(ctx, &L->stage_ptr, stage_ptr.data(), npatch + 1));
  ALFI_CHECK(dev_upload(ctx, &L->dof_ptr, dof_ptr.data(), L->n + 1));
  ALFI_CHECK(dev_upload(ctx, &L->dof_pos, dof_pos.data(), sum_n));
  // the dense inverses (8 sum n_p^2 bytes) are allocated by the first alfi_patches_factor that needs them: a level that
  // gets condensed factors (alfi_patches_set_groups) never holds them -- 265 GB for the 3.4 M-dof Scott-Vogelius level
  ALFI_CHECK(dev_alloc(ctx, &L->inv, 16));
  L->inv_shrunk = true;
  ALFI_CHECK(dev_alloc(ctx, &L->stage, sp));
  ALFI_HIP_CHECK(ctx, hipMemsetAsync(L->stage, 0, (size_t)std::max<int64_t>(sp, 1) * sizeof(double), ctx->stream));
  L->h_patch_ptr.assign(pptr, pptr + npatch + 1);
  L->h_patch_dofs.assign(pdofs, pdofs + sum_n);
  L->h_inv_ptr = inv_ptr;
  dev_free(L->mult_seq);
  L->mult_seq = nullptr;
  L->mult = false;
  L->mult_wave_ptr.clear();
  free_mult_schedule(L);
  // a new patch set invalidates the interior-patch count of alfi_level_set_overlap: back to the plain exchange until the
  // caller declares the new one
  L->overlap = false;
  L->npatch_int = 0;
  return 0;
}

int alfi_patches_set_groups(alfi_level* L, const int32_t* group) {
  alfi_ctx* ctx = L->ctx;
  if (!L->patch_ptr) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_patches_set_groups before alfi_patches_set");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  free_cond(L);
  L->factored = false;
  if (!group) return 0;                                  // back to dense inverses (allocated by alfi_patches_factor)
  if (L->mult) return alfi_set_error(ctx, ALFI_E_STATE, "condensed patch factors do not support multiplicative sweeps");
  const int bs = L->bs;
  const int64_t npatch = L->npatch, nb = L->A.nbrows, nnzb = L->A.nnzb;
  std::vector<int32_t> rowptr(nb + 1), colidx(nnzb > 0 ? nnzb : 1);
  ALFI_HIP_CHECK(ctx, hipMemcpy(rowptr.data(), L->A.rowptr, sizeof(int32_t) * (nb + 1), hipMemcpyDeviceToHost));
  if (nnzb > 0) ALFI_HIP_CHECK(ctx, hipMemcpy(colidx.data(), L->A.colidx, sizeof(int32_t) * nnzb, hipMemcpyDeviceToHost));
  const std::vector<int64_t>& pp = L->h_patch_ptr;
  const std::vector<int32_t>& pd = L->h_patch_dofs;
  const int64_t sum_n = pp[npatch];
  std::vector<int32_t> c_dofs(sum_n), c_slot(sum_n), g_off, g_m, g_sc, g_uoff, sidx, p_nI(npatch), s_uptr, s_uidx;
  std::vector<int64_t> gptr(npatch + 1, 0), g_mat, g_sidx, sptr(npatch + 1, 0), sinv_ptr(npatch + 1, 0);
  std::vector<int32_t> node_pos(nb, -1);               // node -> condensed NODE position inside the current patch
  int64_t mat_off = 0, sinv_off = 0;
  int lds_max = 0, umax = 0, smax = 0;
  s_uptr.push_back(0);
  for (int64_t p = 0; p < npatch; ++p) {
    const int64_t off = pp[p];
    const int n = (int)(pp[p + 1] - off);
    if (n % bs != 0) return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: condensed factors need patches of whole nodes", (long long)p);
    const int nn = n / bs;
    // labels per node; groups = distinct non-negative labels in ascending order
    std::vector<int32_t> lab(nn);
    for (int i = 0; i < nn; ++i) {
      lab[i] = group[off + (int64_t)i * bs];
      for (int c = 0; c < bs; ++c) {
        if (pd[off + (int64_t)i * bs + c] != (pd[off + (int64_t)i * bs] / bs) * bs + c || group[off + (int64_t)i * bs + c] != lab[i])
          return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: entries of a node must be adjacent and carry one group label", (long long)p);
      }
    }
    std::vector<int32_t> labels;
    for (int i = 0; i < nn; ++i) if (lab[i] >= 0) labels.push_back(lab[i]);
    std::sort(labels.begin(), labels.end());
    labels.erase(std::unique(labels.begin(), labels.end()), labels.end());
    const int ng = (int)labels.size();
    // condensed node order: groups (ascending label, ascending node), then the skeleton nodes
    std::vector<int32_t> order;                         // condensed node position -> sorted node position
    order.reserve(nn);
    std::vector<int32_t> gstart(ng + 1, 0);
    for (int g = 0; g < ng; ++g) {
      for (int i = 0; i < nn; ++i) if (lab[i] == labels[g]) order.push_back(i);
      gstart[g + 1] = (int32_t)order.size();
    }
    const int nIn = (int)order.size();                  // interior nodes
    for (int i = 0; i < nn; ++i) if (lab[i] < 0) order.push_back(i);
    for (int q = 0; q < nn; ++q) {
      node_pos[pd[off + (int64_t)order[q] * bs] / bs] = q;
      for (int c = 0; c < bs; ++c) {
        c_dofs[off + (int64_t)q * bs + c] = pd[off + (int64_t)order[q] * bs + c];
        c_slot[off + (int64_t)q * bs + c] = order[q] * bs + c;
      }
    }
    const int sn = nn - nIn, s = sn * bs;
    p_nI[p] = nIn * bs;
    sptr[p + 1] = sptr[p] + s;
    const int64_t ld_s = (s + 1) & ~1;
    sinv_ptr[p] = sinv_off;
    sinv_off += ((int64_t)s * ld_s + 15) & ~(int64_t)15;
    if (s > smax) smax = s;
    gptr[p + 1] = gptr[p] + ng;
    // per group: the skeleton nodes its rows couple to; a column inside another group is an error
    std::vector<std::vector<int32_t>> row_contrib(sn);  // skeleton node -> (u position of its first component) per group
    int uoff = 0;
    std::vector<char> mark(sn);
    for (int g = 0; g < ng; ++g) {
      std::fill(mark.begin(), mark.end(), 0);
      for (int q = gstart[g]; q < gstart[g + 1]; ++q) {
        const int node = pd[off + (int64_t)order[q] * bs] / bs;
        for (int32_t k = rowptr[node]; k < rowptr[node + 1]; ++k) {
          const int cq = node_pos[colidx[k] & 0x7fffffff];
          if (cq < 0) continue;
          if (cq >= nIn) mark[cq - nIn] = 1;
          else if (cq < gstart[g] || cq >= gstart[g + 1]) {
            for (int i = 0; i < nn; ++i) node_pos[pd[off + (int64_t)i * bs] / bs] = -1;
            return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: groups %d and another one are coupled by an operator entry "
                                  "(a group may touch the rest of the patch only through unlabelled dofs)", (long long)p, labels[g]);
          }
        }
      }
      const int m = (gstart[g + 1] - gstart[g]) * bs;
      int scn = 0;
      g_sidx.push_back((int64_t)sidx.size());
      for (int j = 0; j < sn; ++j)
        if (mark[j]) {
          for (int c = 0; c < bs; ++c) sidx.push_back(j * bs + c);
          row_contrib[j].push_back(uoff + scn * bs);
          ++scn;
        }
      const int sc = scn * bs;
      if (m > 64 || sc > 64) {
        for (int i = 0; i < nn; ++i) node_pos[pd[off + (int64_t)i * bs] / bs] = -1;
        return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: group %d holds %d entries coupled to %d skeleton entries; the "
                              "condensed factors handle at most 64 of each", (long long)p, labels[g], m, sc);
      }
      g_off.push_back(gstart[g] * bs);
      g_m.push_back(m);
      g_sc.push_back(sc);
      g_uoff.push_back(uoff);
      g_mat.push_back(mat_off);
      mat_off += cond_group_doubles(m, sc);
      uoff += sc;
    }
    if (uoff > umax) umax = uoff;
    for (int j = 0; j < sn; ++j)
      for (int c = 0; c < bs; ++c) {
        for (int32_t u : row_contrib[j]) s_uidx.push_back(u + c);
        s_uptr.push_back((int32_t)s_uidx.size());
      }
    for (int i = 0; i < nn; ++i) node_pos[pd[off + (int64_t)i * bs] / bs] = -1;
    const int lds = (n + uoff + s + 2) * (int)sizeof(double);
    if (lds > lds_max) lds_max = lds;
  }
  sinv_ptr[npatch] = sinv_off;
  if (lds_max > 150 * 1024)
    return alfi_set_error(ctx, ALFI_E_ARG, "condensed apply would need %d bytes of LDS per patch", lds_max);
  if (g_off.empty()) return alfi_set_error(ctx, ALFI_E_ARG, "no group label >= 0: nothing to condense");
  if (sidx.empty()) sidx.push_back(0);
  if (s_uidx.empty()) s_uidx.push_back(0);
  CondDev cd;
  ALFI_CHECK(cond_upload(L, &cd.dofs, c_dofs));
  ALFI_CHECK(cond_upload(L, &cd.slot, c_slot));
  ALFI_CHECK(cond_upload(L, &cd.gptr, gptr));
  ALFI_CHECK(cond_upload(L, &cd.g_off, g_off));
  ALFI_CHECK(cond_upload(L, &cd.g_m, g_m));
  ALFI_CHECK(cond_upload(L, &cd.g_sc, g_sc));
  ALFI_CHECK(cond_upload(L, &cd.g_uoff, g_uoff));
  ALFI_CHECK(cond_upload(L, &cd.g_mat, g_mat));
  ALFI_CHECK(cond_upload(L, &cd.g_sidx, g_sidx));
  ALFI_CHECK(cond_upload(L, &cd.sidx, sidx));
  ALFI_CHECK(cond_upload(L, &cd.p_nI, p_nI));
  ALFI_CHECK(cond_upload(L, &cd.sptr, sptr));
  ALFI_CHECK(cond_upload(L, &cd.sinv_ptr, sinv_ptr));
  ALFI_CHECK(cond_upload(L, &cd.s_uptr, s_uptr));
  ALFI_CHECK(cond_upload(L, &cd.s_uidx, s_uidx));
  {
    // dispatch order of a full-range apply: descending factor bytes (group matrices + inv(Sigma)), ties by index
    std::vector<int64_t> pbytes((size_t)L->npatch, 0);
    for (int64_t p = 0; p < L->npatch; ++p) {
      const int64_t s = sptr[p + 1] - sptr[p];
      pbytes[p] = s * s;
      for (int64_t g = gptr[p]; g < gptr[p + 1]; ++g) pbytes[p] += cond_group_doubles(g_m[g], g_sc[g]);
    }
    std::vector<int32_t> order((size_t)L->npatch);
    for (int64_t p = 0; p < L->npatch; ++p) order[p] = (int32_t)p;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return pbytes[a] > pbytes[b]; });
    ALFI_CHECK(cond_upload(L, &cd.order, order));
  }
  {
    // tables of the three-launch apply (kernels_bigpatch.hip): chunks of <= 256 rows of inv(Sigma) for the sigma kernel
    // (COND_SIGMA_ROWS), row pairs of the group matrices, the row-sorted order of the u buffer
    std::vector<int32_t> ch_patch, ch_row, xp_grp, bp_grp, g_xp(g_m.size()), g_bp(g_m.size()), u_dst;
    std::vector<int64_t> chptr((size_t)npatch + 1, 0), uptr((size_t)npatch + 1, 0), xp_ptr((size_t)npatch + 1, 0),
        bp_ptr((size_t)npatch + 1, 0);
    int lds_front = 0, lds_back = 0;
    for (int64_t p = 0; p < npatch; ++p) {
      const int s = (int)(sptr[p + 1] - sptr[p]), ld = (s + 1) & ~1;
      for (int r = 0; r < ld; r += COND_SIGMA_ROWS) {
        ch_patch.push_back((int32_t)p);
        ch_row.push_back(r);
      }
      chptr[p + 1] = (int64_t)ch_patch.size();
      int uo = 0, xp = 0, bp = 0;
      for (int64_t g = gptr[p]; g < gptr[p + 1]; ++g) {
        g_xp[g] = xp;
        g_bp[g] = bp;
        for (int i = 0; i < cond_pairs(g_m[g]); ++i) xp_grp.push_back((int32_t)g);
        for (int j = 0; j < cond_pairs(g_sc[g]); ++j) bp_grp.push_back((int32_t)g);
        xp += cond_pairs(g_m[g]);
        bp += cond_pairs(g_sc[g]);
        uo += g_sc[g];
      }
      uptr[p + 1] = uptr[p] + uo;
      xp_ptr[p + 1] = xp_ptr[p] + xp;
      bp_ptr[p + 1] = bp_ptr[p] + bp;
      // every entry of the u buffer is one contribution to one skeleton row: s_uidx restricted to the patch is a permutation
      const int32_t qb = s_uptr[sptr[p]], qe = s_uptr[sptr[p + 1]];
      if (qe - qb != uo) return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: inconsistent skeleton contributions", (long long)p);
      u_dst.resize((size_t)uptr[p + 1]);
      for (int32_t q = qb; q < qe; ++q) u_dst[uptr[p] + s_uidx[q]] = q - qb;
      lds_front = std::max(lds_front, (int)((pp[p + 1] - pp[p] + p_nI[p] + uo + 2) * (int64_t)sizeof(double)));
      lds_back = std::max(lds_back, (int)((s + uo + 2) * (int64_t)sizeof(double)));
    }
    if (ch_patch.empty()) { ch_patch.push_back(0); ch_row.push_back(0); }
    if (xp_grp.empty()) xp_grp.push_back(0);
    if (bp_grp.empty()) bp_grp.push_back(0);
    if (u_dst.empty()) u_dst.push_back(0);
    if (lds_front > 150 * 1024)
      return alfi_set_error(ctx, ALFI_E_ARG, "condensed apply would need %d bytes of LDS per patch", lds_front);
    ALFI_CHECK(cond_upload(L, &cd.ch_patch, ch_patch));
    ALFI_CHECK(cond_upload(L, &cd.ch_row, ch_row));
    ALFI_CHECK(cond_upload(L, &cd.uptr, uptr));
    ALFI_CHECK(cond_upload(L, &cd.u_dst, u_dst));
    ALFI_CHECK(cond_upload(L, &cd.xp_ptr, xp_ptr));
    ALFI_CHECK(cond_upload(L, &cd.xp_grp, xp_grp));
    ALFI_CHECK(cond_upload(L, &cd.g_xp, g_xp));
    ALFI_CHECK(cond_upload(L, &cd.bp_ptr, bp_ptr));
    ALFI_CHECK(cond_upload(L, &cd.bp_grp, bp_grp));
    ALFI_CHECK(cond_upload(L, &cd.g_bp, g_bp));
    {
      // chunks of consecutive groups: at most 256 row pairs of X / W and of B each (a lane per pair), one descriptor per chunk
      // (CondChunk, common.h)
      std::vector<CondChunk> gc;
      std::vector<int64_t> gcptr((size_t)npatch + 1, 0), stage_off((size_t)npatch + 1, 0);
      for (int64_t p = 0; p < npatch; ++p)          // the staging layout of alfi_patches_set: ld_p = n_p rounded up to even
        stage_off[p + 1] = stage_off[p] + ((pp[p + 1] - pp[p] + 1) & ~(int64_t)1);
      int lds_gf = 0, lds_gb = 0;
      for (int64_t p = 0; p < npatch; ++p) {
        int64_t g = gptr[p];
        while (g < gptr[p + 1]) {
          CondChunk c;
          const int64_t ga = g;
          int xp = 0, bp = 0, ne = 0, nu = 0;
          while (g < gptr[p + 1] && xp + cond_pairs(g_m[g]) <= 256 && bp + cond_pairs(g_sc[g]) <= 256) {
            xp += cond_pairs(g_m[g]);
            bp += cond_pairs(g_sc[g]);
            ne += g_m[g];
            nu += g_sc[g];
            ++g;
          }
          c.off = pp[p]; c.ubase = uptr[p]; c.sidx0 = g_sidx[ga]; c.stage_off = stage_off[p];
          c.xq0 = (int32_t)(xp_ptr[p] + g_xp[ga]); c.xq1 = c.xq0 + xp;
          c.bq0 = (int32_t)(bp_ptr[p] + g_bp[ga]); c.bq1 = c.bq0 + bp;
          c.e0 = g_off[ga]; c.ne = ne; c.u0 = g_uoff[ga]; c.nu = nu; c.nI = p_nI[p]; c.pad = 0;
          c.xp0 = (int32_t)xp_ptr[p]; c.bp0 = (int32_t)bp_ptr[p];
          gc.push_back(c);
          lds_gf = std::max(lds_gf, (int)(2 * ne * sizeof(double)));
          lds_gb = std::max(lds_gb, (int)(nu * sizeof(double)));
        }
        gcptr[p + 1] = (int64_t)gc.size();
      }
      if (xp_ptr[npatch] > INT32_MAX || bp_ptr[npatch] > INT32_MAX)
        return alfi_set_error(ctx, ALFI_E_ARG, "condensed factors: too many row pairs on one level");
      if (gc.empty()) gc.push_back(CondChunk());
      ALFI_CHECK(cond_upload(L, &cd.gc, gc));
      L->h_cond_gcptr = gcptr;
      L->cond_lds_gfront = lds_gf + 16;
      L->cond_lds_gback = lds_gb + 16;
      ALFI_CHECK(dev_alloc(ctx, &cd.ubuf, uptr[npatch] > 0 ? uptr[npatch] : 1));
      L->cond_allocs.push_back(cd.ubuf);
    }
    L->h_cond_chptr = chptr;
    L->cond_lds_front = lds_front;
    L->cond_lds_back = lds_back;
    ALFI_CHECK(dev_alloc(ctx, &cd.tmp, sum_n > 0 ? sum_n : 1));
    L->cond_allocs.push_back(cd.tmp);
  }
  ALFI_CHECK(dev_alloc(ctx, &cd.mat, mat_off));
  L->cond_allocs.push_back(cd.mat);
  ALFI_CHECK(dev_alloc(ctx, &cd.sinv, sinv_off));
  L->cond_allocs.push_back(cd.sinv);
  // the dense inverses are not needed any more
  if (!L->inv_shrunk) {
    dev_free(L->inv);
    L->inv = nullptr;
    ALFI_CHECK(dev_alloc(ctx, &L->inv, 16));
    L->inv_shrunk = true;
  }
  L->cd = cd;
  L->cond = true;
  L->il_valid = false;
  L->h_sptr = sptr;
  L->h_cond_gptr = gptr;
  L->cond_ngroups = (int64_t)g_off.size();
  L->cond_mat_doubles = mat_off;
  L->cond_sinv_doubles = sinv_off;
  L->cond_lds_bytes = lds_max;
  L->cond_umax = umax;
  L->cond_max_s = smax;
  return 0;
}

int alfi_patches_factor_bytes(alfi_level* L, int64_t* bytes) {
  *bytes = L->cond ? 8 * (L->cond_mat_doubles + L->cond_sinv_doubles) : 8 * L->inv_doubles;
  return 0;
}

static void free_mult_schedule(alfi_level* L) {
  dev_free(L->mult_items); dev_free(L->mult_pred0); dev_free(L->mult_pred); dev_free(L->mult_succ_ptr);
  dev_free(L->mult_succ); dev_free(L->mult_ctl); dev_free(L->mult_rowtab);
  L->mult_rowtab = nullptr;
  L->mult_items = L->mult_pred0 = L->mult_pred = L->mult_succ_ptr = L->mult_succ = L->mult_ctl = nullptr;
  L->mult_nitems = 0;
}

int alfi_patches_set_multiplicative(alfi_level* L, int64_t nit, const int64_t* iterset, int symmetrise) {
  alfi_ctx* ctx = L->ctx;
  if (!L->patch_ptr) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_patches_set_multiplicative before alfi_patches_set");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(L->mult_seq);
  L->mult_seq = nullptr;
  L->mult = false;
  L->mult_wave_ptr.clear();
  free_mult_schedule(L);
  // a new schedule starts with a clean device-side error word: after a timeout every later sweep on this ctx would stop at its
  // first wait (the word is tested inside the spin loop); (re)setting the sweeps is how a context recovers
  if (ctx->dev_err) ALFI_HIP_CHECK(ctx, hipMemset(ctx->dev_err, 0, 16));
  if (nit == 0) return 0;
  if (nit < 0 || !iterset) return alfi_set_error(ctx, ALFI_E_ARG, "bad iteration set");
  if (L->cond) return alfi_set_error(ctx, ALFI_E_STATE, "multiplicative sweeps need dense patch inverses (alfi_patches_set_groups(NULL))");
  if (L->pou) return alfi_set_error(ctx, ALFI_E_STATE, "partition of unity applies to the additive smoother");
  // partitioned levels: every rank sweeps over its own patches with the residual of its local vector (ghost slots hold
  // the rank's own contributions only) and the ghost contributions are added onto their owners at the end -- what
  // PCPATCH does under MPI: local Gauss-Seidel, additive between ranks [3P]
  if (nit > INT32_MAX) return alfi_set_error(ctx, ALFI_E_ARG, "iteration set too long");
  const int bs = L->bs;
  // patches must be unions of whole nodes (the sweep works on block rows); up to 64 nodes and 160 dofs a wave sweeps a
  // patch, beyond (macro stars) a workgroup does
  L->mult_big = L->max_np > SMALL_PATCH_MAX;
  for (int64_t p = 0; p < L->npatch; ++p) {
    const int64_t a = L->h_patch_ptr[p], b = L->h_patch_ptr[p + 1];
    if ((b - a) / bs > 64) L->mult_big = true;
    if ((b - a) % bs != 0)
      return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: multiplicative sweeps need patches of whole nodes", (long long)p);
    for (int64_t q = a; q < b; ++q)
      if (L->h_patch_dofs[q] != (L->h_patch_dofs[a + ((q - a) / bs) * bs] / bs) * bs + (int32_t)((q - a) % bs))
        return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld does not consist of whole nodes", (long long)p);
  }
  for (int64_t t = 0; t < nit; ++t)
    if (iterset[t] < 0 || iterset[t] >= L->npatch) return alfi_set_error(ctx, ALFI_E_ARG, "iteration set entry out of range");
  // sparsity of the operator on the host (row starts are marked in the sign bit of the flat layout)
  const int64_t nb = L->A.nbrows, nnzb = L->A.nnzb;
  std::vector<int32_t> rowptr(nb + 1), colidx(nnzb > 0 ? nnzb : 1);
  ALFI_HIP_CHECK(ctx, hipMemcpy(rowptr.data(), L->A.rowptr, sizeof(int32_t) * (nb + 1), hipMemcpyDeviceToHost));
  if (nnzb > 0) ALFI_HIP_CHECK(ctx, hipMemcpy(colidx.data(), L->A.colidx, sizeof(int32_t) * nnzb, hipMemcpyDeviceToHost));
  // wavefront of position t = 1 + max wavefront of earlier positions whose patch holds a node in the closure of patch t
  // (closure = columns of the patch's block rows).  node_wave[c] = last wavefront that wrote node c.
  std::vector<int32_t> node_wave(nb, -1), wave_of(nit);
  int32_t nwave = 0;
  for (int64_t t = 0; t < nit; ++t) {
    const int64_t p = iterset[t];
    const int64_t a = L->h_patch_ptr[p], b = L->h_patch_ptr[p + 1];
    int32_t w = -1;
    for (int64_t q = a; q < b; q += bs) {
      const int32_t node = L->h_patch_dofs[q] / bs;
      for (int32_t k = rowptr[node]; k < rowptr[node + 1]; ++k) {
        const int32_t c = colidx[k] & 0x7fffffff;
        if (node_wave[c] > w) w = node_wave[c];
      }
    }
    ++w;
    wave_of[t] = w;
    if (w + 1 > nwave) nwave = w + 1;
    for (int64_t q = a; q < b; q += bs) node_wave[L->h_patch_dofs[q] / bs] = w;
  }
  // counting sort of the positions by wavefront (stable)
  L->mult_wave_ptr.assign(nwave + 1, 0);
  for (int64_t t = 0; t < nit; ++t) L->mult_wave_ptr[wave_of[t] + 1]++;
  for (int32_t w = 0; w < nwave; ++w) L->mult_wave_ptr[w + 1] += L->mult_wave_ptr[w];
  std::vector<int32_t> seq(nit);
  {
    std::vector<int64_t> fill(L->mult_wave_ptr.begin(), L->mult_wave_ptr.end() - 1);
    for (int64_t t = 0; t < nit; ++t) seq[fill[wave_of[t]]++] = (int32_t)iterset[t];
  }
  ALFI_CHECK(dev_upload(ctx, &L->mult_seq, seq.data(), nit));
  L->mult = true;
  L->mult_symmetrise = symmetrise != 0;
  // ---- the persistent schedule (wave-per-patch levels): items = the forward sweep in wavefront-major order, then (symmetrised)
  // the wavefronts in reverse order, each in its listed order -- exactly the launch sequence of the per-wavefront schedule.
  // Predecessors of an item = the LAST WRITERS (earlier items) of the nodes it reads: patches writing the same node conflict
  // with each other, so earlier writers are ordered before the last one transitively; a patch that READS a node this item
  // writes has, by the symmetric sparsity, nodes in this item's closure, whose last writer is that patch or a later conflicting
  // one.  The list order is a topological order of these dependencies.
  free_mult_schedule(L);
  {
    if (!L->mult_big) {
      // row table of the sweep kernels (kernels_patch.hip, mult_wg_rows): per patch node its first block, block count and node
      std::vector<int32_t> rowtab((size_t)L->npatch * 64 * 3, 0);
      for (int64_t p = 0; p < L->npatch; ++p) {
        const int64_t a = L->h_patch_ptr[p], b = L->h_patch_ptr[p + 1];
        for (int64_t q = a, i = 0; q < b; q += bs, ++i) {
          const int32_t node = L->h_patch_dofs[q] / bs;
          int32_t* rt = &rowtab[((size_t)p * 64 + (size_t)i) * 3];
          rt[0] = rowptr[node];
          rt[1] = rowptr[node + 1] - rowptr[node];
          rt[2] = node;
        }
      }
      ALFI_CHECK(dev_upload(ctx, &L->mult_rowtab, rowtab.data(), (int64_t)rowtab.size()));
    }
    std::vector<int32_t> items(seq);
    if (symmetrise)
      for (int32_t w = nwave - 1; w >= 0; --w)
        for (int64_t q = L->mult_wave_ptr[w]; q < L->mult_wave_ptr[w + 1]; ++q) items.push_back(seq[q]);
    const int64_t N = (int64_t)items.size();
    if (N > INT32_MAX / 2) return alfi_set_error(ctx, ALFI_E_ARG, "iteration set too long for the persistent schedule");
    std::vector<int32_t> last_writer(nb, -1), pred0(N, 0), tmp;
    std::vector<std::vector<int32_t>> succ_of;     // built as (pred, item) pairs to keep memory flat
    std::vector<int32_t> e_from, e_to;
    e_from.reserve((size_t)N * 32);
    e_to.reserve((size_t)N * 32);
    for (int64_t t = 0; t < N; ++t) {
      const int64_t p = items[t];
      const int64_t a = L->h_patch_ptr[p], b = L->h_patch_ptr[p + 1];
      tmp.clear();
      for (int64_t q = a; q < b; q += bs) {
        const int32_t node = L->h_patch_dofs[q] / bs;
        for (int32_t k = rowptr[node]; k < rowptr[node + 1]; ++k) {
          const int32_t lw = last_writer[colidx[k] & 0x7fffffff];
          if (lw >= 0) tmp.push_back(lw);
        }
      }
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      pred0[t] = (int32_t)tmp.size();
      for (int32_t f : tmp) {
        e_from.push_back(f);
        e_to.push_back((int32_t)t);
      }
      for (int64_t q = a; q < b; q += bs) last_writer[L->h_patch_dofs[q] / bs] = (int32_t)t;
    }
    if (e_from.size() > (size_t)INT32_MAX) return alfi_set_error(ctx, ALFI_E_ARG, "too many dependencies for int32 offsets");
    std::vector<int32_t> succ_ptr(N + 1, 0), succ(e_from.size() > 0 ? e_from.size() : 1);
    for (int32_t f : e_from) succ_ptr[f + 1]++;
    for (int64_t t = 0; t < N; ++t) succ_ptr[t + 1] += succ_ptr[t];
    {
      std::vector<int32_t> fill(succ_ptr.begin(), succ_ptr.end() - 1);
      for (size_t e = 0; e < e_from.size(); ++e) succ[fill[e_from[e]]++] = e_to[e];
    }
    ALFI_CHECK(dev_upload(ctx, &L->mult_items, items.data(), N));
    ALFI_CHECK(dev_upload(ctx, &L->mult_pred0, pred0.data(), N));
    ALFI_CHECK(dev_alloc(ctx, &L->mult_pred, N));
    ALFI_CHECK(dev_upload(ctx, &L->mult_succ_ptr, succ_ptr.data(), N + 1));
    ALFI_CHECK(dev_upload(ctx, &L->mult_succ, succ.data(), (int64_t)succ.size()));
    ALFI_CHECK(dev_alloc(ctx, &L->mult_ctl, 4));
    L->mult_nitems = (int32_t)N;
  }
  return 0;
}

int alfi_patches_set_partition_of_unity(alfi_level* L, int on) {
  if (on && L->mult) return alfi_set_error(L->ctx, ALFI_E_STATE, "partition of unity applies to the additive smoother");
  L->pou = on != 0;
  return 0;
}

int alfi_patches_multiplicative_levels(alfi_level* L, int64_t* nwave) {
  *nwave = L->mult ? (int64_t)L->mult_wave_ptr.size() - 1 : 0;
  return 0;
}

int alfi_patches_factor(alfi_level* L) {
  alfi_ctx* ctx = L->ctx;
  if (!L->patch_ptr) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_patches_factor before alfi_patches_set");
  int t = alfi_prof_begin(ctx, ALFI_EV_PATCH_FACTOR);
  ALFI_HIP_CHECK(ctx, hipMemsetAsync(L->status, 0, sizeof(int), ctx->stream));
  if (!L->cond && L->inv_shrunk) {                 // first dense factorisation of this patch set
    ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    dev_free(L->inv);
    L->inv = nullptr;
    ALFI_CHECK(dev_alloc(ctx, &L->inv, L->inv_doubles));
    L->inv_shrunk = false;
  }
  if (L->cond) {
    ALFI_CHECK(launch_cond_factor(L));            // condensed factors: group inverses + Schur complements
  } else if (L->max_np > SMALL_PATCH_MAX) {
    ALFI_CHECK(launch_big_factor(L));             // macro-star sized patches: blocked Gauss-Jordan on the matrix cores
  } else {
    ALFI_CHECK(launch_patch_gather_dense(L));
    ALFI_CHECK(launch_patch_invert(L));
  }
  ALFI_CHECK(build_patch_il(L));                  // small-patch levels: the wave-contiguous copy the apply streams
  alfi_prof_end(ctx, t);
  int st = 0;
  ALFI_HIP_CHECK(ctx, hipMemcpyAsync(&st, L->status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  // every stored inverse is probed (|| A_p X_p e - e ||); the ones that fail -- an unpivoted elimination met a zero or
  // tiny pivot -- are re-inverted with partial pivoting, as the reference's LAPACK / UMFPACK factorisations would
  ALFI_CHECK(patch_verify_and_repair(L, st));
  L->factored = true;
  return 0;
}

int alfi_patches_check(alfi_level* L, double* worst_residual, int64_t* flagged, int64_t* repaired, double* worst_after) {
  if (!L->factored) return alfi_set_error(L->ctx, ALFI_E_STATE, "alfi_patches_check before alfi_patches_factor");
  if (worst_residual) *worst_residual = L->chk_worst;
  if (flagged) *flagged = L->chk_flagged;
  if (repaired) *repaired = L->chk_repaired;
  if (worst_after) *worst_after = L->chk_flagged > 0 ? L->chk_worst_after : L->chk_worst;
  return 0;
}

int alfi_patch_apply(alfi_level* L, const double* dx, double* dy) {
  if (!L->factored) return alfi_set_error(L->ctx, ALFI_E_STATE, "alfi_patch_apply before alfi_patches_factor");
  if (dx == dy) return alfi_set_error(L->ctx, ALFI_E_ARG, "alfi_patch_apply: x and y must not alias");
  L->ctx->cur_tag = L->id;
  return level_patch_apply(L, dx, dy);
}

int alfi_patches_stats(alfi_level* L, int64_t* npatch, int64_t* sum_n, int64_t* sum_n2) {
  if (npatch) *npatch = L->npatch;
  if (sum_n) *sum_n = L->sum_n;
  if (sum_n2) *sum_n2 = L->sum_n2;
  return 0;
}

int alfi_patch_get_inverse(alfi_level* L, int64_t p, double* out) {
  alfi_ctx* ctx = L->ctx;
  if (!L->factored) return alfi_set_error(ctx, ALFI_E_STATE, "patches not factored");
  if (p < 0 || p >= L->npatch) return alfi_set_error(ctx, ALFI_E_ARG, "patch index out of range");
  if (L->cond) return alfi_set_error(ctx, ALFI_E_STATE, "condensed patch factors hold no dense inverse");
  const int64_t n = L->h_patch_ptr[p + 1] - L->h_patch_ptr[p];
  const int64_t ld = (n + 1) & ~(int64_t)1;
  std::vector<double> tmp(n * ld);
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ALFI_HIP_CHECK(ctx, hipMemcpy(tmp.data(), L->inv + L->h_inv_ptr[p], tmp.size() * sizeof(double),
                                hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < n; ++j) out[i * n + j] = tmp[patch_inv_index((int)i, (int)j, (int)n, (int)ld)];
  return 0;
}

// ---- FGMRES(k) smoother ----------------------------------------------------------------------------------------------------
static int ensure_fgmres_workspace(alfi_level* L, int k) {
  alfi_ctx* ctx = L->ctx;
  if (k <= L->kmax) return 0;
  if (k > RED_MAXV - 1) return alfi_set_error(ctx, ALFI_E_ARG, "k = %d exceeds the supported maximum %d", k, RED_MAXV - 1);
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(L->V);
  dev_free(L->Z);
  dev_free(L->w);
  dev_free(L->hs);
  L->V = L->Z = L->w = L->hs = nullptr;
  L->kmax = 0;
  // stride of the Krylov bases: the local length rounded up to even, so that every basis vector starts on a 16-byte
  // boundary (the BLAS-1 kernels read entry pairs; n = 3 x nodes is odd on half of the 3-D levels)
  L->ldv = (L->n + 1) & ~(int64_t)1;
  ALFI_CHECK(dev_alloc(ctx, &L->V, (int64_t)(k + 1) * L->ldv));
  ALFI_CHECK(dev_alloc(ctx, &L->Z, (int64_t)k * L->ldv));
  ALFI_CHECK(dev_alloc(ctx, &L->w, L->ldv));
  HsLayout hl(k);
  ALFI_CHECK(dev_alloc(ctx, &L->hs, hl.total));
  ALFI_HIP_CHECK(ctx, hipMemsetAsync(L->hs, 0, sizeof(double) * hl.total, ctx->stream));
  L->kmax = k;
  return 0;
}

// FGMRES(k) on a small, unpartitioned level with the additive smoother: four launches per iteration (see the kernels in
// kernels_vec.hip).  Same algorithm as the general path below -- right-preconditioned FGMRES, classical Gram-Schmidt,
// explicit norms -- with the normalisation of the new Krylov vector moved behind the patch solves (they are linear).
static int smooth_fgmres_fused(alfi_level* L, int k, const double* db, double* dx, int nonzero_guess, bool flat_spmv) {
  alfi_ctx* ctx = L->ctx;
  const int K = L->kmax;
  const int64_t n = L->n, ldv = L->ldv;
  HsLayout hl(K);
  double *V = L->V, *Z = L->Z, *w = L->w, *hs = L->hs;
  double* hdots = hs + hl.hd;
  int t;
  if (nonzero_guess) {
    ALFI_CHECK(alfi_residual(L, db, dx, w));
  } else {
    ALFI_HIP_CHECK(ctx, hipMemsetAsync(dx, 0, sizeof(double) * n, ctx->stream));
    t = alfi_prof_begin(ctx, ALFI_EV_BLAS1);
    ALFI_CHECK(launch_copy(ctx, w, db, n));
    alfi_prof_end(ctx, t);
  }
  t = alfi_prof_begin(ctx, ALFI_EV_BLAS1);
  ALFI_CHECK(launch_norm_partials(ctx, w, n));
  alfi_prof_end(ctx, t);
  const int G = red_blocks_for(n);
  const double* normpart = ctx->red_partial;
  for (int j = 0; j < k; ++j) {
    double* zj = Z + (int64_t)j * ldv;
    ALFI_CHECK(launch_patch_apply_range(L, 0, L->npatch, w));                                   // stage <- patch solves of w
    t = alfi_prof_begin(ctx, ALFI_EV_PATCH_SCATTER);
    ALFI_CHECK(launch_patch_sum_scale(L, w, zj, V + (int64_t)j * ldv, normpart, G, hdots, hs, j, K));   // z_j, v_j, H column j-1
    alfi_prof_end(ctx, t);
    int nb = 0;
    if (!flat_spmv) {
      t = alfi_prof_begin(ctx, ALFI_EV_MATMULT);
      ALFI_CHECK(launch_bsr_spmv_dot(ctx, L->A_own, zj, w, V, ldv, j + 1, ctx->red_partial, &nb));  // w = A z_j, V^T w partials
      alfi_prof_end(ctx, t);
    } else {
      // long or very uneven block rows (the 3-D operators): the nnz-balanced product, then the dots as their own pass; up to
      // 256 partials per vector the projection kernel sums them itself, beyond a one-block reduction does
      t = alfi_prof_begin(ctx, ALFI_EV_MATMULT);
      ALFI_CHECK(launch_bsr_spmv(ctx, L->A_own, zj, w, nullptr, 1.0, 0));
      alfi_prof_end(ctx, t);
      t = alfi_prof_begin(ctx, ALFI_EV_BLAS1);
      const bool in_consumer = G <= 256;
      ALFI_CHECK(launch_multi_dot(ctx, V, ldv, j + 1, w, in_consumer ? nullptr : hdots, n));
      alfi_prof_end(ctx, t);
      nb = in_consumer ? G : 0;
    }
    t = alfi_prof_begin(ctx, ALFI_EV_BLAS1);
    ALFI_CHECK(launch_multi_axpy_norm(ctx, V, ldv, j + 1, hdots, w, n, ctx->red_partial2, nb));     // h, w -= V h, |w|^2 partials
    alfi_prof_end(ctx, t);
    normpart = ctx->red_partial2;
  }
  t = alfi_prof_begin(ctx, ALFI_EV_BLAS1);
  ALFI_CHECK(launch_fgmres_finish_fused(ctx, normpart, G, hdots, hs, k, K));                       // H column k-1, y
  ALFI_CHECK(launch_update_solution(ctx, dx, Z, ldv, k, hs + hl.y, n));
  alfi_prof_end(ctx, t);
  return 0;
}

int alfi_smooth_fgmres(alfi_level* L, int k, const double* db, double* dx, int nonzero_guess) {
  alfi_ctx* ctx = L->ctx;
  if (k < 1) return alfi_set_error(ctx, ALFI_E_ARG, "k must be >= 1");
  if (!L->factored) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_smooth_fgmres before alfi_patches_factor");
  ALFI_CHECK(ensure_fgmres_workspace(L, k));
  ctx->cur_tag = L->id;
  {
    // unpartitioned levels with short operator rows: the four-launch iteration (on the small levels a smoother iteration is
    // launch latency, on the large 2-D ones the folded vector passes save BLAS-1 traffic, which is comparable to the patch
    // traffic there): patch_sum_scale_kernel writes z_j and v_j in one pass (the normalisation follows the linear patch
    // solves), the product is the lanes-per-row kernel with the dots folded in.
    // Short rows = the 2-D operators (<= 32 blocks).  With the 50 .. 125 blocks per row of the 3-D ones the flat segmented
    // product of the general path below is the faster kernel (config 3 19.0 against 18.9 ms, config 2 5.10 against 5.43, same
    // box; with the bimodal rows of [P1+FB]^3 the lanes-per-row product idles most lanes: config 6 188.8 ms fused, 175.0
    // general) ... except on levels of <= 50 000 dofs, which are bound by the number of dependent launches whatever the rows
    // look like (config 3 18.66 -> 18.12-18.23 ms).  Sending every level here was measured too: nothing (config 3 17.33 /
    // 17.35 ms, config 4 163.1 / 163.3, config 5 27.24 / 27.10).
    constexpr int64_t small_n = 50000;
    const bool fusable = !alfi_test_large_paths() && !L->distributed && L->n_own == L->n && !L->mult && k + 1 <= 16 &&
                         L->A_own.flat;
    const bool short_rows = L->max_row_blocks <= 32 || L->n <= small_n;
    if (fusable && short_rows) return smooth_fgmres_fused(L, k, db, dx, nonzero_guess, !short_rows);
  }
  const int K = L->kmax;
  const int64_t n = L->n_own;    // vector kernels and reductions run on the owned prefix
  const int64_t ldv = L->ldv;    // stride of the Krylov bases (local length incl. ghost slots, rounded up to even)
  const bool par = L->distributed;
  HsLayout hl(K);
  double* V = L->V;
  double* Z = L->Z;
  double* w = L->w;
  double* hs = L->hs;
  // partitioned level: dots / norms are reduced into the caller's buffer and all-reduced there
  double* hdots = par ? ctx->dred : hs + hl.hd;
  double* nrm2 = par ? ctx->dred + RED_MAXV : nullptr;
  int t;
  // r0 = b - A x (MatMult), beta = |r0|, v0 = r0 / beta
  if (nonzero_guess) {
    ALFI_CHECK(alfi_residual(L, db, dx, w));
  } else {
    ALFI_HIP_CHECK(ctx, hipMemsetAsync(dx, 0, sizeof(double) * L->n, ctx->stream));
    t = alfi_prof_begin(ctx, ALFI_EV_BLAS1);
    ALFI_CHECK(launch_copy(ctx, w, db, n));
    alfi_prof_end(ctx, t);
  }
  // reductions: red_blocks_for(n) partials per vector.  Up to 256 of them (levels of <= 1 M dofs, where a smoother
  // iteration is a chain of launches of a few microseconds each) the kernel that needs a reduced value sums the partials
  // itself -- every block in the same fixed order -- instead of waiting for a one-block reduction launch.
  const int G = red_blocks_for(n);
  // (G = n / 4096 partials per vector; a limit of 512 would include config 3's finest level, 319 partials: measured 18.31
  // against 18.13 ms per cycle, the re-summation in every consumer block costs more than the launch)
  const bool fused = !par && G <= 256 && k + 1 <= 16;
  t = alfi_prof_begin(ctx, ALFI_EV_BLAS1);
  ALFI_CHECK(launch_norm_partials(ctx, w, n));
  if (par) ALFI_CHECK(launch_reduce_partials(ctx, ctx->red_partial, G, 1, nrm2));
  alfi_prof_end(ctx, t);
  if (par) ALFI_CHECK(comm_allreduce(L, RED_MAXV, 1));
  t = alfi_prof_begin(ctx, ALFI_EV_BLAS1);
  ALFI_CHECK(launch_norm_init_finish(ctx, par ? nrm2 : ctx->red_partial, par ? 1 : G, hs, K));
  ALFI_CHECK(launch_scale_by_inv(ctx, V, w, hs + hl.beta, n));
  alfi_prof_end(ctx, t);
  for (int j = 0; j < k; ++j) {
    bool zghosts = false;
    ALFI_CHECK(level_patch_apply(L, V + (int64_t)j * ldv, Z + (int64_t)j * ldv, &zghosts));   // z_j = M^-1 v_j
    ALFI_CHECK(level_spmv(L, Z + (int64_t)j * ldv, w, nullptr, 0, zghosts));                 // w = A z_j
    t = alfi_prof_begin(ctx, ALFI_EV_BLAS1);
    // h = V^T w (classical GS); fused: the partials stay in red_partial and the projection kernel sums them
    ALFI_CHECK(launch_multi_dot(ctx, V, ldv, j + 1, w, fused ? nullptr : hdots, n));
    // partitioned: |w|^2 rides along in the same all-reduce; |w - V h|^2 = |w|^2 - |h|^2 then needs no second one
    const bool pyth = par && !ctx->exact_norm;
    const double* ww = pyth ? hdots + (j + 1) : nullptr;
    if (pyth) {
      ALFI_CHECK(launch_norm_partials(ctx, w, n));
      ALFI_CHECK(launch_reduce_partials(ctx, ctx->red_partial, G, 1, hdots + (j + 1)));
    }
    alfi_prof_end(ctx, t);
    if (par) ALFI_CHECK(comm_allreduce(L, 0, pyth ? j + 2 : j + 1));
    t = alfi_prof_begin(ctx, ALFI_EV_BLAS1);
    // w -= V h, |w|^2 partials (into the second partial buffer: the dot partials are still being read)
    ALFI_CHECK(launch_multi_axpy_norm(ctx, V, ldv, j + 1, hdots, w, n, ctx->red_partial2, fused ? G : 0));
    if (par && !pyth) ALFI_CHECK(launch_reduce_partials(ctx, ctx->red_partial2, G, 1, nrm2));
    alfi_prof_end(ctx, t);
    if (par && !pyth) ALFI_CHECK(comm_allreduce(L, RED_MAXV, 1));
    t = alfi_prof_begin(ctx, ALFI_EV_BLAS1);
    const double* part = par ? nrm2 : ctx->red_partial2;
    const int nblk = par ? 1 : G;
    if (j + 1 < k)   // Hessenberg column + v_{j+1} = w / |w| in one launch
      ALFI_CHECK(launch_hessenberg_scale(ctx, part, nblk, hdots, hs, j, K, V + (int64_t)(j + 1) * ldv, w, n, ww));
    else
      ALFI_CHECK(launch_hessenberg_update(ctx, part, nblk, hdots, hs, j, K, ww));
    alfi_prof_end(ctx, t);
  }
  t = alfi_prof_begin(ctx, ALFI_EV_BLAS1);
  ALFI_CHECK(launch_fgmres_finish(ctx, hs, k, K));
  ALFI_CHECK(launch_update_solution(ctx, dx, Z, ldv, k, hs + hl.y, n));
  alfi_prof_end(ctx, t);
  return 0;
}

// ---- coarse solve ------------------------------------------------------------------------------------------------------------
int alfi_coarse_set_inverse(alfi_level* L, const double* inv, int inv_is_device) {
  alfi_ctx* ctx = L->ctx;
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (L->cinv_owned) dev_free(L->cinv);
  L->cinv = nullptr;
  L->cinv_owned = false;
  mf_free(L->mf);
  L->mf = nullptr;
  if (inv_is_device) {
    L->cinv = const_cast<double*>(inv);
  } else {
    ALFI_CHECK(dev_upload(ctx, &L->cinv, inv, L->n * L->n));
    L->cinv_owned = true;
  }
  return 0;
}

// the coarse residual probe fails the setup beyond this (ALFI_COARSE_CHECK_FAIL; round 2 failed at 1e-5: ADVICE r2 -- at
// Re 10 000 on fine coarse grids a merely ill-conditioned operator would have become a setup error)
static double coarse_probe_fail() {
  static const double v = getenv("ALFI_COARSE_CHECK_FAIL") ? atof(getenv("ALFI_COARSE_CHECK_FAIL")) : 1e-2;
  return v;
}

// +-1 pattern for the residual probe of the coarse inverse
int alfi_coarse_factor(alfi_level* L) {
  alfi_ctx* ctx = L->ctx;
  if (L->has_halo && L->n_own != L->n)
    return alfi_set_error(ctx, ALFI_E_STATE, "alfi_coarse_factor needs a level owned by one rank");
  if (L->n > ((int64_t)1 << 17)) return alfi_set_error(ctx, ALFI_E_ARG, "coarse level of %lld dofs: the dense inverse would "
                                                      "take %.0f GB", (long long)L->n, 8e-9 * (double)L->n * (double)L->n);
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (L->cinv_owned) dev_free(L->cinv);
  L->cinv = nullptr;
  L->cinv_owned = false;
  mf_free(L->mf);
  L->mf = nullptr;
  double* inv = nullptr;
  ALFI_CHECK(dev_alloc(ctx, &inv, L->n * L->n));
  ALFI_HIP_CHECK(ctx, hipMemsetAsync(L->status, 0, sizeof(int), ctx->stream));
  int rc = launch_coarse_factor(L, inv);
  int st = 0;
  if (rc == 0 && hipMemcpy(&st, L->status, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) rc = ALFI_E_HIP;
  if (rc == 0 && st != 0) rc = alfi_set_error(ctx, ALFI_E_SINGULAR, "zero pivot block while inverting the coarse operator");
  // residual probe: || A (X e) - e ||_inf for a +-1 vector e
  if (rc == 0) {
    double *de = nullptr, *dy = nullptr, *dr = nullptr;
    double worst = 0.0;
    rc = dev_alloc(ctx, &de, L->n);
    if (rc == 0) rc = dev_alloc(ctx, &dy, L->n);
    if (rc == 0) rc = dev_alloc(ctx, &dr, L->n);
    if (rc == 0) rc = launch_probe_fill(ctx, de, L->n);
    if (rc == 0) rc = launch_dense_gemv(ctx, inv, de, dy, L->n);
    if (rc == 0) rc = launch_bsr_spmv(ctx, L->A, dy, dr, nullptr, 1.0, 0);
    if (rc == 0) rc = launch_probe_residual(ctx, dr, L->n, &worst);
    else (void)hipStreamSynchronize(ctx->stream);
    dev_free(de);
    dev_free(dy);
    dev_free(dr);
    if (rc == 0) {
      L->cinv_residual = worst;
      // (cond(A_0) ~ 1e8 at config 4: cond * eps * |X| |A| |e| leaves ~1e-7 even for a perfectly rounded inverse; the
      // residual of a backward-stable factorisation scales with the condition number, so a large one is REPORTED through
      // alfi_coarse_residual and only one beyond coarse_probe_fail() -- default 1e-2, or a non-finite one -- fails the setup)
      if (!(worst <= coarse_probe_fail()))
        rc = alfi_set_error(ctx, ALFI_E_SINGULAR, "coarse inverse fails the residual probe: || A X e - e || = %.3e", worst);
    }
  }
  if (rc != 0) {
    dev_free(inv);
    return rc;
  }
  L->cinv = inv;
  L->cinv_owned = true;
  return 0;
}

// || A x - e ||_inf for x = solve(e), e a +-1 vector: the residual probe shared by the dense and the sparse coarse solver
static int coarse_probe(alfi_level* L, double* worst_out) {
  alfi_ctx* ctx = L->ctx;
  double *de = nullptr, *dy = nullptr, *dr = nullptr;
  int rc = dev_alloc(ctx, &de, L->n);
  if (rc == 0) rc = dev_alloc(ctx, &dy, L->n);
  if (rc == 0) rc = dev_alloc(ctx, &dr, L->n);
  if (rc == 0) rc = launch_probe_fill(ctx, de, L->n);
  if (rc == 0) rc = L->mf ? mf_solve(L, de, dy) : launch_dense_gemv(ctx, L->cinv, de, dy, L->n);
  if (rc == 0) rc = launch_bsr_spmv(ctx, L->A, dy, dr, nullptr, 1.0, 0);
  if (rc == 0) rc = launch_probe_residual(ctx, dr, L->n, worst_out);
  else (void)hipStreamSynchronize(ctx->stream);
  dev_free(de);
  dev_free(dy);
  dev_free(dr);
  return rc;
}

// Sparse direct factorisation of the level operator (multifrontal, mf_coarse.h) for coarse grids beyond the dense inverse.
// coords: nbrows x dim node coordinates (host) for the geometric nested dissection, or NULL (BFS level sets of the graph).
int alfi_coarse_factor_sparse(alfi_level* L, const double* coords, int dim, int leaf_nodes) {
  alfi_ctx* ctx = L->ctx;
  if (L->has_halo && L->n_own != L->n)
    return alfi_set_error(ctx, ALFI_E_STATE, "alfi_coarse_factor_sparse needs a level owned by one rank");
  if (coords && (dim < 1 || dim > 3)) return alfi_set_error(ctx, ALFI_E_ARG, "node coordinates of dimension %d", dim);
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (L->cinv_owned) dev_free(L->cinv);
  L->cinv = nullptr;
  L->cinv_owned = false;
  ALFI_HIP_CHECK(ctx, hipMemsetAsync(L->status, 0, sizeof(int), ctx->stream));
  ALFI_CHECK(mf_factor(L, coords, dim, leaf_nodes));
  int st = 0;
  ALFI_HIP_CHECK(ctx, hipMemcpy(&st, L->status, sizeof(int), hipMemcpyDeviceToHost));
  int rc = 0;
  if (st != 0) rc = alfi_set_error(ctx, ALFI_E_SINGULAR, "zero pivot block in a front of the coarse factorisation");
  double worst = 0.0;
  if (rc == 0) rc = coarse_probe(L, &worst);
  if (rc == 0) {
    L->cinv_residual = worst;
    if (!(worst <= coarse_probe_fail()))
      rc = alfi_set_error(ctx, ALFI_E_SINGULAR, "sparse coarse factorisation fails the residual probe: || A x - e || = %.3e", worst);
  }
  if (rc != 0) {
    mf_free(L->mf);
    L->mf = nullptr;
  }
  return rc;
}

int alfi_coarse_factor_bytes(alfi_level* L, int64_t* bytes) {
  if (L->mf) *bytes = mf_bytes(L->mf);
  else if (L->cinv) *bytes = L->n * L->n * 8;
  else return alfi_set_error(L->ctx, ALFI_E_STATE, "no coarse factorisation");
  return 0;
}

int alfi_coarse_residual(alfi_level* L, double* worst) {
  if (!L->cinv && !L->mf) return alfi_set_error(L->ctx, ALFI_E_STATE, "no coarse inverse");
  *worst = L->cinv_residual;
  return 0;
}

int alfi_coarse_solve(alfi_level* L, const double* db, double* dx) {
  if (!L->cinv && !L->mf) return alfi_set_error(L->ctx, ALFI_E_STATE, "alfi_coarse_solve before alfi_coarse_set_inverse");
  L->ctx->cur_tag = L->id;
  int t = alfi_prof_begin(L->ctx, ALFI_EV_COARSE);
  if (L->mf) ALFI_CHECK(mf_solve(L, db, dx));
  else ALFI_CHECK(launch_dense_gemv(L->ctx, L->cinv, db, dx, L->n));
  alfi_prof_end(L->ctx, t);
  return 0;
}

// ---- transfer ------------------------------------------------------------------------------------------------------------------
int alfi_transfer_create(alfi_ctx* ctx, alfi_level* coarse, alfi_level* fine, const alfi_bsr_host* P,
                         const alfi_bsr_host* PT, const alfi_bsr_host* PT_plain, const alfi_bsr_host* D_I,
                         const alfi_bsr_host* D_IT, int64_t nblk, int m, const int32_t* blk_dofs, const double* K_II,
                         const double* D_II, alfi_transfer** out) {
  if (!ctx || !coarse || !fine || !P || !PT || !D_I || !D_IT || !blk_dofs || !K_II || !D_II || !out)
    return alfi_set_error(ctx, ALFI_E_ARG, "NULL argument");
  const int bs = fine->bs;
  if (coarse->bs != bs) return alfi_set_error(ctx, ALFI_E_ARG, "block size mismatch");
  if (m < 1 || m > PATCH_MAX)
    return alfi_set_error(ctx, ALFI_E_ARG, "interior block size %d not in 1..%d", m, PATCH_MAX);
  // partitioned fine level: P and D_I^T hold the owned fine rows, P^T the owned fine columns (its rows are partial sums
  // over the local coarse numbering, reverse-added to their owners); serial: n_own == n
  if (P->nbrows * bs != fine->n_own || P->nbcols * bs != coarse->n || PT->nbrows * bs != coarse->n ||
      (PT->nbcols * bs != fine->n_own && PT->nbcols * bs != fine->n))
    return alfi_set_error(ctx, ALFI_E_ARG, "prolongation shape does not match the levels");
  if ((nblk * m) % bs != 0 || D_I->nbrows * bs != nblk * m || D_I->nbcols * bs != fine->n ||
      D_IT->nbrows * bs != fine->n_own || D_IT->nbcols * bs != nblk * m)
    return alfi_set_error(ctx, ALFI_E_ARG, "grad-div interior rows shape mismatch");
  if (fine->distributed && !coarse->has_halo)
    return alfi_set_error(ctx, ALFI_E_STATE, "coarse level of a partitioned transfer needs alfi_level_set_partition");
  for (int64_t i = 0; i < nblk * m; ++i)
    if (blk_dofs[i] < 0 || blk_dofs[i] >= fine->n) return alfi_set_error(ctx, ALFI_E_ARG, "blk_dofs out of range");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  alfi_transfer* T = new alfi_transfer();
  T->ctx = ctx;
  T->coarse = coarse;
  T->fine = fine;
  T->bs = bs;
  T->nblk = nblk;
  T->m = m;
  T->ld = (m + 1) & ~1;
  int rc = upload_bsr(ctx, &T->P, P, bs);
  if (rc == 0) rc = upload_bsr(ctx, &T->PT, PT, bs);
  if (rc == 0) {
    if (PT_plain) {
      rc = upload_bsr(ctx, &T->PTp, PT_plain, bs);
    } else {
      T->PTp = T->PT;
      T->ptp_alias = true;
    }
  }
  if (rc == 0) rc = upload_bsr(ctx, &T->DI, D_I, bs);
  if (rc == 0) rc = upload_bsr(ctx, &T->DIT, D_IT, bs);
  if (rc == 0) rc = dev_upload(ctx, &T->blk_dofs, blk_dofs, nblk * m);
  if (rc == 0) rc = dev_upload(ctx, &T->KII, K_II, nblk * m * m);
  if (rc == 0) rc = dev_upload(ctx, &T->DII, D_II, nblk * m * m);
  if (m > 32) {
    // macro-cell blocks (Scott-Vogelius transfer, transfer.py:49-88): inverses in the row-piece layout, solved with the
    // patch smoother's kernels; compact vectors keep stride m, the kernels write row pairs with stride ld (for odd m the
    // result passes through pm_tmp)
    T->patch_mode = true;
    T->bstride = ((int64_t)m * T->ld + 15) & ~(int64_t)15;
    std::vector<int64_t> ptr(nblk + 1), iptr(nblk + 1), sptr(nblk + 1);
    std::vector<int32_t> iota((size_t)nblk * m);
    for (int64_t b = 0; b <= nblk; ++b) {
      ptr[b] = b * m;
      iptr[b] = b * T->bstride;
      sptr[b] = b * T->ld;
    }
    for (int64_t i = 0; i < nblk * m; ++i) iota[i] = (int32_t)i;
    if (rc == 0) rc = dev_upload(ctx, &T->pm_ptr, ptr.data(), nblk + 1);
    if (rc == 0) rc = dev_upload(ctx, &T->pm_inv_ptr, iptr.data(), nblk + 1);
    if (rc == 0) rc = dev_upload(ctx, &T->pm_stage_ptr, sptr.data(), nblk + 1);
    if (rc == 0) rc = dev_upload(ctx, &T->pm_iota, iota.data(), nblk * m);
    if (rc == 0) rc = dev_alloc(ctx, &T->binv, nblk * T->bstride);
    if (rc == 0 && T->ld != m) rc = dev_alloc(ctx, &T->pm_tmp, nblk * T->ld);
    if (rc == 0) rc = dev_alloc(ctx, &T->pm_res, nblk * m);       // one step of iterative refinement per interior solve
    if (rc == 0) rc = dev_alloc(ctx, &T->pm_cor, nblk * m);
    if (rc == 0) rc = dev_alloc(ctx, &T->AIIt, nblk * m * T->ld);
  } else if (rc == 0) {
    rc = dev_alloc(ctx, &T->binv, nblk * m * T->ld);
  }
  if (rc == 0) rc = dev_alloc(ctx, &T->tI, nblk * m);
  if (rc == 0) rc = dev_alloc(ctx, &T->bI, nblk * m);
  if (rc == 0) rc = dev_alloc(ctx, &T->tmp_f, fine->n);
  if (rc == 0) rc = dev_alloc(ctx, &T->status, 1);
  if (rc != 0) {
    alfi_transfer_destroy(T);
    return rc;
  }
  *out = T;
  return 0;
}

int alfi_transfer_destroy(alfi_transfer* T) {
  if (!T) return 0;
  (void)hipStreamSynchronize(T->ctx->stream);
  free_bsr(&T->P);
  free_bsr(&T->PT);
  if (!T->ptp_alias) free_bsr(&T->PTp);
  free_bsr(&T->DI);
  free_bsr(&T->DIT);
  dev_free(T->blk_dofs);
  dev_free(T->KII);
  dev_free(T->DII);
  dev_free(T->binv);
  dev_free(T->pm_ptr);
  dev_free(T->pm_inv_ptr);
  dev_free(T->pm_stage_ptr);
  dev_free(T->pm_iota);
  dev_free(T->pm_tmp);
  dev_free(T->pm_res);
  dev_free(T->pm_cor);
  dev_free(T->AIIt);
  dev_free(T->tI);
  dev_free(T->bI);
  dev_free(T->tmp_f);
  dev_free(T->inj);
  if (T->injm) {
    free_csr(T->injm);
    delete T->injm;
  }
  dev_free(T->status);
  delete T;
  return 0;
}

int alfi_transfer_set_injection(alfi_transfer* T, const int32_t* fine_node) {
  alfi_ctx* ctx = T->ctx;
  if (!fine_node) return alfi_set_error(ctx, ALFI_E_ARG, "NULL injection map");
  if (T->fine->has_halo || T->coarse->has_halo)
    return alfi_set_error(ctx, ALFI_E_ARG, "alfi_inject is not available on partitioned levels");
  const int64_t nc = T->coarse->n / T->bs, nf = T->fine->n / T->bs;
  for (int64_t i = 0; i < nc; ++i)
    if (fine_node[i] < 0 || fine_node[i] >= nf) return alfi_set_error(ctx, ALFI_E_ARG, "injection map entry out of range");
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(T->inj);
  T->inj = nullptr;
  return dev_upload(ctx, &T->inj, fine_node, nc);
}

int alfi_transfer_set_injection_matrix(alfi_transfer* T, const alfi_csr_host* J) {
  alfi_ctx* ctx = T->ctx;
  if (!J) return alfi_set_error(ctx, ALFI_E_ARG, "NULL injection matrix");
  if (T->fine->has_halo || T->coarse->has_halo)
    return alfi_set_error(ctx, ALFI_E_ARG, "alfi_inject is not available on partitioned levels");
  const int64_t nc = T->coarse->n / T->bs, nf = T->fine->n / T->bs;
  if (J->nrows != nc || J->ncols != nf)
    return alfi_set_error(ctx, ALFI_E_ARG, "injection matrix must be %lld x %lld (coarse nodes x fine nodes)", (long long)nc, (long long)nf);
  for (int64_t i = 0; i < nc; ++i)
    if (J->rowptr[i + 1] < J->rowptr[i]) return alfi_set_error(ctx, ALFI_E_ARG, "row pointer not monotone");
  for (int64_t k = 0; k < J->rowptr[nc]; ++k)
    if (J->colidx[k] < 0 || J->colidx[k] >= nf) return alfi_set_error(ctx, ALFI_E_ARG, "injection matrix column out of range");
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (T->injm) {
    free_csr(T->injm);
    delete T->injm;
    T->injm = nullptr;
  }
  DevCSR* M = new DevCSR();
  const int rc = upload_csr(ctx, M, J);
  if (rc != 0) {
    free_csr(M);
    delete M;
    return rc;
  }
  T->injm = M;
  return 0;
}

int alfi_inject(alfi_transfer* T, const double* dxf, double* dxc) {
  if (T->injm) return launch_inject_csr(T->ctx, *T->injm, T->bs, dxf, dxc);
  if (!T->inj) return alfi_set_error(T->ctx, ALFI_E_STATE, "alfi_inject before alfi_transfer_set_injection");
  return launch_halo_pack(T->ctx, dxc, dxf, T->inj, T->coarse->n / T->bs, T->bs);   // coarse[i] = fine[inj[i]]
}

int alfi_transfer_update(alfi_transfer* T, double nu, double gamma) {
  alfi_ctx* ctx = T->ctx;
  T->nu = nu;
  T->gamma = gamma;
  ALFI_HIP_CHECK(ctx, hipMemsetAsync(T->status, 0, sizeof(int), ctx->stream));
  ALFI_CHECK(launch_block_build_invert(T));
  int st = 0;
  ALFI_HIP_CHECK(ctx, hipMemcpyAsync(&st, T->status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (st != 0) return alfi_set_error(ctx, ALFI_E_SINGULAR, "zero pivot while inverting a coarse-cell interior block");
  T->ready = true;
  return 0;
}

int alfi_transfer_get_block_inverse(alfi_transfer* T, int64_t blk, double* out) {
  alfi_ctx* ctx = T->ctx;
  if (!T->ready) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_transfer_get_block_inverse before alfi_transfer_update");
  if (blk < 0 || blk >= T->nblk) return alfi_set_error(ctx, ALFI_E_ARG, "block index out of range");
  const int m = T->m, ld = T->ld;
  const int64_t stride = T->patch_mode ? T->bstride : (int64_t)m * ld;
  std::vector<double> tmp((size_t)m * ld);
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ALFI_HIP_CHECK(ctx, hipMemcpy(tmp.data(), T->binv + blk * stride, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j)
      out[(int64_t)i * m + j] = T->patch_mode ? tmp[patch_inv_index(i, j, m, ld)] : tmp[(int64_t)j * ld + i];
  return 0;
}

int alfi_prolong(alfi_transfer* T, const double* dxc, double* dxf) {
  alfi_ctx* ctx = T->ctx;
  if (!T->ready) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_prolong before alfi_transfer_update");
  ctx->cur_tag = T->fine->id;
  // partitioned fine level: coarse ghosts in, rhs on the owned fine rows, fine ghosts in; the interior solves of every
  // coarse cell touching an owned node run locally (redundantly at partition boundaries), writes to ghost slots are scratch
  const bool par = T->fine->distributed;
  if (par) ALFI_CHECK(halo_fwd(T->coarse, const_cast<double*>(dxc)));
  int t = alfi_prof_begin(ctx, ALFI_EV_PROLONG);
  ALFI_CHECK(launch_bsr_spmv(ctx, T->P, dxc, dxf, nullptr, 0.0, 0));                 // rhs = P coarse        :247
  alfi_prof_end(ctx, t);
  if (par) ALFI_CHECK(halo_fwd(T->fine, dxf));
  t = alfi_prof_begin(ctx, ALFI_EV_PROLONG);
  ALFI_CHECK(launch_bsr_spmv(ctx, T->DI, dxf, T->bI, nullptr, 0.0, 0));              // b_I = (D rhs)_I       :249
  ALFI_CHECK(launch_block_gemv(T, T->bI, T->tI, false));                              // t = inv(A_II) b_I     :254-257
  ALFI_CHECK(launch_scatter_sub(ctx, dxf, T->blk_dofs, T->tI, T->gamma, T->nblk * T->m));  // fine = rhs - gamma t  :259
  ALFI_CHECK(launch_zero_dofs(ctx, dxf, T->fine->bc_dofs, T->fine->nbc));
  alfi_prof_end(ctx, t);
  return 0;
}

int alfi_restrict(alfi_transfer* T, const double* drf, double* drc, int robust) {
  alfi_ctx* ctx = T->ctx;
  if (robust && !T->ready) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_restrict before alfi_transfer_update");
  ctx->cur_tag = T->fine->id;
  const bool par = T->fine->distributed;
  if (par && robust) ALFI_CHECK(halo_fwd(T->fine, const_cast<double*>(drf)));
  int t = alfi_prof_begin(ctx, ALFI_EV_RESTRICT);
  if (robust) {
    ALFI_CHECK(launch_block_gemv(T, drf, T->tI, true));                               // t = inv(A_II) r_I     :265-270
    ALFI_CHECK(launch_bsr_spmv(ctx, T->DIT, T->tI, T->tmp_f, drf, T->gamma, 1));      // s = r - gamma D t     :272-274
    ALFI_CHECK(launch_bsr_spmv(ctx, T->PT, T->tmp_f, drc, nullptr, 0.0, 0));          // coarse = P^T s        :275
  } else {
    ALFI_CHECK(launch_bsr_spmv(ctx, T->PTp, drf, drc, nullptr, 0.0, 0));              // firedrake.restrict
  }
  alfi_prof_end(ctx, t);
  if (par) ALFI_CHECK(halo_rev(T->coarse, drc));   // partial sums over the owned fine nodes -> the coarse owners
  t = alfi_prof_begin(ctx, ALFI_EV_RESTRICT);
  ALFI_CHECK(launch_zero_dofs(ctx, drc, T->coarse->bc_dofs, T->coarse->nbc));
  alfi_prof_end(ctx, t);
  return 0;
}

// ---- multigrid --------------------------------------------------------------------------------------------------------------------
int alfi_mg_create(alfi_ctx* ctx, int nlevels, alfi_level** levels, alfi_transfer** transfers, int k,
                   int robust_restriction, alfi_mg** out) {
  if (!ctx || nlevels < 1 || !levels || (nlevels > 1 && !transfers) || !out)
    return alfi_set_error(ctx, ALFI_E_ARG, "bad arguments");
  for (int l = 1; l < nlevels; ++l) {
    if (transfers[l - 1]->coarse != levels[l - 1] || transfers[l - 1]->fine != levels[l])
      return alfi_set_error(ctx, ALFI_E_ARG, "transfer %d does not link levels %d and %d", l - 1, l - 1, l);
  }
  // (patch factors and the coarse inverse are asked for when a cycle runs -- mg_ready -- not here: a hierarchy whose operators
  // are formed on the device is linked before its first refresh)
  alfi_mg* mg = new alfi_mg();
  mg->ctx = ctx;
  mg->levels.assign(levels, levels + nlevels);
  if (nlevels > 1) mg->transfers.assign(transfers, transfers + nlevels - 1);
  mg->k = k;
  mg->robust = robust_restriction;
  for (int l = 0; l < nlevels; ++l) {
    alfi_level* L = levels[l];
    int rc = 0;
    if (!L->mg_b) rc = dev_alloc(ctx, &L->mg_b, L->n);
    if (rc == 0 && !L->mg_x) rc = dev_alloc(ctx, &L->mg_x, L->n);
    if (rc == 0 && !L->mg_r) rc = dev_alloc(ctx, &L->mg_r, L->n);
    if (rc == 0 && l > 0 && L->n_own > 0) rc = ensure_fgmres_workspace(L, k);
    if (rc != 0) {
      delete mg;
      return rc;
    }
  }
  *out = mg;
  return 0;
}

int alfi_mg_destroy(alfi_mg* mg) {
  if (!mg) return 0;
  (void)hipStreamSynchronize(mg->ctx->stream);
  for (CycleGraph& g : mg->graphs)
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
  delete mg;
  return 0;
}

// PCMGMCycle_Private [3P]: x_l <- V(b_l, x_l).  x_zero: the incoming x is known to be zero (every level below the one the
// cycle starts on): the pre-smoother then runs with a zero initial guess -- r0 = b, no residual SpMV -- as PCMG does by
// switching KSPSetInitialGuessNonzero off for the down-smoother of those levels; the result is the same bit for bit.
static int vcycle(alfi_mg* mg, int l, const double* b, double* x, bool x_zero) {
  alfi_ctx* ctx = mg->ctx;
  alfi_level* L = mg->levels[l];
  // partitioned hierarchies: a rank's lowest level is either the coarse grid it owns or ghost copies of a level another
  // rank owns and solves (then there is nothing to do here); every level above is owned in part by every rank
  if (l == 0) return L->n_own > 0 ? alfi_coarse_solve(L, b, x) : 0;
  alfi_level* C = mg->levels[l - 1];
  alfi_transfer* T = mg->transfers[l - 1];
  ALFI_CHECK(alfi_smooth_fgmres(L, mg->k, b, x, x_zero ? 0 : 1));    // pre-smooth
  ALFI_CHECK(alfi_residual(L, b, x, L->mg_r));                       // r = b - A x
  ALFI_CHECK(alfi_restrict(T, L->mg_r, C->mg_b, mg->robust));        // b_{l-1} = R r
  // x_{l-1} = 0: the coarse solve overwrites it, a smoother with a zero initial guess zeroes it itself
  ALFI_CHECK(vcycle(mg, l - 1, C->mg_b, C->mg_x, true));
  ALFI_CHECK(alfi_prolong(T, C->mg_x, L->mg_r));                     // x += P x_{l-1}
  ctx->cur_tag = L->id;
  int t = alfi_prof_begin(ctx, ALFI_EV_BLAS1);
  ALFI_CHECK(launch_axpy(ctx, x, L->mg_r, 1.0, L->n_own));
  alfi_prof_end(ctx, t);
  ALFI_CHECK(alfi_smooth_fgmres(L, mg->k, b, x, 1));                 // post-smooth
  return 0;
}

// PCMGFCycle_Private [3P]
static int fcycle(alfi_mg* mg, const double* db, double* dx) {
  const int Lmax = (int)mg->levels.size() - 1;
  if (Lmax == 0) return mg->levels[0]->n_own > 0 ? alfi_coarse_solve(mg->levels[0], db, dx) : 0;
  // restrict the right-hand side through all levels
  const double* bf = db;
  for (int l = Lmax; l >= 1; --l) {
    ALFI_CHECK(alfi_restrict(mg->transfers[l - 1], bf, mg->levels[l - 1]->mg_b, mg->robust));
    bf = mg->levels[l - 1]->mg_b;
  }
  for (int l = 0; l < Lmax; ++l) {
    alfi_level* L = mg->levels[l];
    // note: inside vcycle(l) the coarser levels' mg_b / mg_x are overwritten; level l's own b must survive, and it
    // does: vcycle(l) only writes mg_b of levels < l.  But the restricted rhs of levels < l is then gone -- it is
    // not needed any more at that point (levels are visited in increasing order).
    ALFI_CHECK(vcycle(mg, l, L->mg_b, L->mg_x, l == 0));    // l >= 1: x_l is the prolonged coarser solution
    double* xnext = (l + 1 == Lmax) ? dx : mg->levels[l + 1]->mg_x;
    ALFI_CHECK(alfi_prolong(mg->transfers[l], L->mg_x, xnext));
  }
  return vcycle(mg, Lmax, db, dx, false);
}

// ---- whole cycles as hipGraphs (alfi_ctx_set_graph) -----------------------------------------------------------------------
// A cycle on the small levels of a hierarchy is a few hundred short kernels whose launch cost exceeds their run time
// (ldc2d at 1.2 M dofs: 6 ms per V-cycle against 1.8 ms of HBM time; multiplicative sweeps launch one kernel per
// wavefront of patches).  With graphs on, the second call of a cycle with the same (b, x) pair captures its launch
// sequence from the library's stream and every later call replays it; the first call runs eagerly (it may still allocate
// the smoother's workspace).  Anything a captured argument depends on is part of the entry's signature, so new operator
// values in place are picked up for free and new (nu, gamma), patches or a new coarse inverse re-capture.  Profiling
// events and the exchange callbacks of partitioned levels cannot be captured: those runs stay eager.
static void cycle_signature(alfi_mg* mg, std::vector<uint64_t>* sig) {
  auto push = [&](const void* q) { sig->push_back((uint64_t)(uintptr_t)q); };
  auto pushd = [&](double v) {
    uint64_t u;
    memcpy(&u, &v, sizeof(u));
    sig->push_back(u);
  };
  sig->clear();
  sig->push_back((uint64_t)mg->k);
  sig->push_back((uint64_t)mg->robust);
  for (alfi_level* L : mg->levels) {
    push(L->A.vals);
    push(L->inv);
    push(L->patch_ptr);
    push(L->patch_dofs);
    push(L->cinv);
    push(L->mf);
    push(L->V);
    push(L->mult_seq);
    sig->push_back((uint64_t)L->npatch);
    sig->push_back((uint64_t)L->kmax);
    sig->push_back((uint64_t)(L->mult ? 1 + (L->mult_symmetrise ? 1 : 0) + 4 * L->mult_wave_ptr.size() : 0));
  }
  for (alfi_transfer* T : mg->transfers) {
    push(T->binv);
    pushd(T->nu);
    pushd(T->gamma);
  }
}

// every level has what a cycle multiplies with: factored patches, a coarse inverse
static int mg_ready(alfi_mg* mg) {
  alfi_ctx* ctx = mg->ctx;
  for (size_t l = 1; l < mg->levels.size(); ++l)
    if (mg->levels[l]->n_own > 0 && !mg->levels[l]->factored)
      return alfi_set_error(ctx, ALFI_E_STATE, "level %d: patches not factored", (int)l);
  // a rank that only holds ghost copies of its lowest level (the owner solves it) needs no coarse inverse
  alfi_level* C = mg->levels[0];
  if (C->n_own > 0 && !C->cinv && !C->mf) return alfi_set_error(ctx, ALFI_E_STATE, "coarse level has no inverse");
  return 0;
}

static int run_cycle(alfi_mg* mg, int kind, const double* db, double* dx) {
  alfi_ctx* ctx = mg->ctx;
  ALFI_CHECK(mg_ready(mg));
  auto eager = [&]() { return kind ? fcycle(mg, db, dx) : vcycle(mg, (int)mg->levels.size() - 1, db, dx, false); };
  bool ok = ctx->use_graph && ctx->prof == 0;
  for (alfi_level* L : mg->levels) ok = ok && !L->has_halo;
  if (!ok) return eager();
  std::vector<uint64_t> sig;
  cycle_signature(mg, &sig);
  CycleGraph* G = nullptr;
  for (CycleGraph& g : mg->graphs)
    if (g.kind == kind && g.b == db && g.x == dx) G = &g;
  if (!G) {                       // first call with this pair: eager, remember it
    if (mg->graphs.size() >= 16) {
      for (CycleGraph& g : mg->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
      mg->graphs.clear();
    }
    CycleGraph g;
    g.kind = kind;
    g.b = db;
    g.x = dx;
    mg->graphs.push_back(g);
    return eager();
  }
  if (G->failed) return eager();
  if (G->exec && G->sig != sig) {
    (void)hipGraphExecDestroy(G->exec);
    G->exec = nullptr;
  }
  if (!G->exec) {
    // the workspace may have changed between the eager call and now: compare after a possible eager warm-up
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError();
      G->failed = true;
      return eager();
    }
    const int rc = eager();
    const hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    if (rc != 0 || e != hipSuccess || !graph) {
      if (graph) (void)hipGraphDestroy(graph);
      (void)hipGetLastError();
      G->failed = true;
      return rc != 0 ? rc : eager();
    }
    const hipError_t ei = hipGraphInstantiate(&G->exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) {
      (void)hipGetLastError();
      G->exec = nullptr;
      G->failed = true;
      return eager();
    }
    cycle_signature(mg, &G->sig);
  }
  ALFI_HIP_CHECK(ctx, hipGraphLaunch(G->exec, ctx->stream));
  return 0;
}

int alfi_mg_vcycle(alfi_mg* mg, const double* db, double* dx) { return run_cycle(mg, 0, db, dx); }

int alfi_mg_fcycle(alfi_mg* mg, const double* db, double* dx) { return run_cycle(mg, 1, db, dx); }

int alfi_ctx_comm_stats(alfi_ctx* ctx, int64_t* halo_exchanges, int64_t* allreduces, int64_t* doubles_sent, int reset) {
  if (halo_exchanges) *halo_exchanges = ctx->comm_nhalo;
  if (allreduces) *allreduces = ctx->comm_nred;
  if (doubles_sent) *doubles_sent = ctx->comm_sent;
  if (reset) ctx->comm_nhalo = ctx->comm_nred = ctx->comm_sent = 0;
  return 0;
}

int alfi_ctx_set_graph(alfi_ctx* ctx, int on) {
  ctx->use_graph = on != 0;
  return 0;
}

// ---- outer saddle-point solve (alfi/solver.py:386-422) ----------------------------------------------------------------------------
static int upload_csr(alfi_ctx* ctx, DevCSR* d, const alfi_csr_host* h) {
  d->nrows = h->nrows;
  d->ncols = h->ncols;
  d->nnz = h->rowptr[h->nrows];
  ALFI_CHECK(dev_upload(ctx, &d->rowptr, h->rowptr, h->nrows + 1));
  ALFI_CHECK(dev_upload(ctx, &d->colidx, h->colidx, d->nnz));
  ALFI_CHECK(dev_upload(ctx, &d->vals, h->vals, d->nnz));
  return 0;
}
static void free_csr(DevCSR* d) {
  dev_free(d->rowptr);
  dev_free(d->colidx);
  dev_free(d->vals);
  *d = DevCSR();
}

// ---- building blocks of the outer solve on partitioned levels (alfi_amd/dist.py: DistSaddle drives them from the host) -------
int alfi_level_halo_forward(alfi_level* L, double* dv) {
  if (!L->distributed) return 0;
  L->ctx->cur_tag = L->id;
  return halo_fwd(L, dv);
}

int alfi_level_halo_reverse_add(alfi_level* L, double* dv) {
  if (!L->distributed) return 0;
  L->ctx->cur_tag = L->id;
  return halo_rev(L, dv);
}

// ---- small vector operations for the Newton loop around the path: state, update and residual stay in HBM (the reference's
// NonlinearVariationalSolver keeps z as a Function, alfi/solver.py:245-273); only scalars come back ------------------------------
int alfi_transfer_stats(int64_t* h2d_bytes, int64_t* d2h_bytes, int reset) {
  if (h2d_bytes) *h2d_bytes = g_alfi_h2d_bytes.load();
  if (d2h_bytes) *d2h_bytes = g_alfi_d2h_bytes.load();
  if (reset) {
    g_alfi_h2d_bytes = 0;
    g_alfi_d2h_bytes = 0;
  }
  return 0;
}

int alfi_vec_axpy(alfi_ctx* ctx, double* dy, const double* dx, double a, int64_t n) {
  if (n < 0 || (n > 0 && (!dy || !dx))) return alfi_set_error(ctx, ALFI_E_ARG, "bad vector arguments");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return n == 0 ? 0 : launch_axpy(ctx, dy, dx, a, n);
}

int alfi_vec_copy(alfi_ctx* ctx, double* dy, const double* dx, int64_t n) {
  if (n < 0 || (n > 0 && (!dy || !dx))) return alfi_set_error(ctx, ALFI_E_ARG, "bad vector arguments");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return n == 0 ? 0 : launch_copy(ctx, dy, dx, n);
}

// dst[i] = src[idx[i]] (bs consecutive doubles per index): a level's state from a vector that holds the values it needs
int alfi_vec_gather(alfi_ctx* ctx, double* dst, const double* src, const int32_t* d_idx, int64_t nidx, int bs) {
  if (nidx < 0 || bs < 1 || (nidx > 0 && (!dst || !src || !d_idx))) return alfi_set_error(ctx, ALFI_E_ARG, "bad gather arguments");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return nidx == 0 ? 0 : launch_halo_pack(ctx, dst, src, d_idx, nidx, bs);
}

// dv[Dirichlet dofs of the level] = 0 (bc.zero(F), alfi/solver.py:282-286)
int alfi_level_zero_bc(alfi_level* L, double* dv) {
  alfi_ctx* ctx = L->ctx;
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return L->nbc == 0 ? 0 : launch_zero_dofs(ctx, dv, L->bc_dofs, L->nbc);
}

// x . y of two vectors of the outer solve, (velocity | pressure) -- on a partitioned finest level the owned entries, summed over
// the ranks: every rank gets the same value.  Fixed summation order (two-stage reduction).
int alfi_saddle_dot(alfi_saddle* S, const double* dx, const double* dy, double* out_host) {
  alfi_ctx* ctx = S->ctx;
  if (!dx || !dy || !out_host) return alfi_set_error(ctx, ALFI_E_ARG, "NULL argument");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int64_t n = S->nu_dofs + S->np_dofs;
  if (!S->dotbuf) ALFI_CHECK(dev_alloc(ctx, &S->dotbuf, 2));
  double* out = S->par ? ctx->dred : S->dotbuf;
  ALFI_CHECK(launch_multi_dot(ctx, dx, n, 1, dy, out, n));
  if (S->par) {
    ctx->cur_tag = S->fine->id;
    ALFI_CHECK(comm_allreduce(S->fine, 0, 1));
  }
  ALFI_HIP_CHECK(ctx, hipMemcpyAsync(out_host, out, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return check_dev_err(ctx);
}

// the merged exchange of the smoother (alfi_level_set_sum_exchange): every holder of a shared node ends with the sum of all
// holders' values
int alfi_level_halo_sum(alfi_level* L, double* dv) {
  if (!L->distributed) return 0;
  if (!L->sum_ready) return alfi_set_error(L->ctx, ALFI_E_STATE, "alfi_level_halo_sum before alfi_level_set_sum_exchange");
  L->ctx->cur_tag = L->id;
  return halo_sum(L, dv);
}

struct alfi_csr {
  alfi_ctx* ctx = nullptr;
  DevCSR M;
};

int alfi_csr_create(alfi_ctx* ctx, const alfi_csr_host* h, alfi_csr** out) {
  if (!ctx || !h || !out) return alfi_set_error(ctx, ALFI_E_ARG, "NULL argument");
  for (int64_t i = 0; i < h->nrows; ++i)
    if (h->rowptr[i + 1] < h->rowptr[i]) return alfi_set_error(ctx, ALFI_E_ARG, "row pointer not monotone");
  const int64_t nnz = h->rowptr[h->nrows];
  for (int64_t k = 0; k < nnz; ++k)
    if (h->colidx[k] < 0 || h->colidx[k] >= h->ncols) return alfi_set_error(ctx, ALFI_E_ARG, "column index out of range");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  alfi_csr* C = new alfi_csr();
  C->ctx = ctx;
  int rc = upload_csr(ctx, &C->M, h);
  if (rc != 0) {
    free_csr(&C->M);
    delete C;
    return rc;
  }
  *out = C;
  return 0;
}

int alfi_csr_destroy(alfi_csr* C) {
  if (!C) return 0;
  (void)hipStreamSynchronize(C->ctx->stream);
  free_csr(&C->M);
  delete C;
  return 0;
}

// mode 0: y = M x;  1: y = b - alpha M x;  2: y += M x;  3: y = alpha M x
int alfi_csr_mult(alfi_csr* C, const double* dx, double* dy, const double* db, double alpha, int mode) {
  if (mode < 0 || mode > 3 || (mode == 1 && !db)) return alfi_set_error(C->ctx, ALFI_E_ARG, "bad mode / missing b");
  return launch_csr_spmv(C->ctx, C->M, dx, dy, db, alpha, mode);
}

// On a PARTITIONED finest level (alfi_level_set_partition, distributed) the call is collective and the matrices are the rank's
// pieces: B = the rank's pressure rows over all LOCAL velocity dofs (owned + ghost: n_loc columns), BT = its transpose (n_loc
// rows; the ghost rows hold contributions for their owners, reverse-added).  Vectors then hold (owned velocity dofs | owned
// pressure dofs) and every reduction is one all-reduce -- the same Krylov loop as on one GPU, no host arithmetic.
int alfi_saddle_create(alfi_mg* mg, const alfi_csr_host* B, const alfi_csr_host* BT, const double* mass_diag,
                       double nu, double gamma, int remove_constant_nullspace, alfi_saddle** out) {
  if (!mg || !B || !BT || !mass_diag || !out) return alfi_set_error(mg ? mg->ctx : nullptr, ALFI_E_ARG, "NULL argument");
  alfi_ctx* ctx = mg->ctx;
  alfi_level* F = mg->levels.back();
  const bool par = F->distributed;
  if (F->has_halo && !par)
    return alfi_set_error(ctx, ALFI_E_ARG, "the finest level has a halo but is not distributed: no outer solve on it");
  if (B->ncols != F->n || BT->nrows != F->n || BT->ncols != B->nrows)
    return alfi_set_error(ctx, ALFI_E_ARG, "divergence matrix shape does not match the finest level");
  for (int64_t i = 0; i < B->nrows; ++i)
    if (!(mass_diag[i] > 0.0)) return alfi_set_error(ctx, ALFI_E_ARG, "pressure mass matrix entry %lld is not positive", (long long)i);
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  alfi_saddle* S = new alfi_saddle();
  S->ctx = ctx;
  S->mg = mg;
  S->fine = F;
  S->par = par;
  S->n_loc = F->n;
  S->nu_dofs = par ? F->n_own : F->n;
  S->np_dofs = B->nrows;
  S->np_global = (double)B->nrows;
  S->nu = nu;
  S->gamma = gamma;
  S->remove_nullspace = remove_constant_nullspace != 0;
  std::vector<double> minv(std::max<int64_t>(B->nrows, 1), 1.0);
  for (int64_t i = 0; i < B->nrows; ++i) minv[i] = 1.0 / mass_diag[i];
  int rc = upload_csr(ctx, &S->B, B);
  if (rc == 0) rc = upload_csr(ctx, &S->BT, BT);
  if (rc == 0) rc = dev_upload(ctx, &S->minv, minv.data(), (int64_t)minv.size());
  if (rc == 0) rc = dev_alloc(ctx, &S->tmp_u, S->n_loc);
  if (rc == 0) rc = dev_alloc(ctx, &S->tmp_p, std::max<int64_t>(S->np_dofs, 1));
  if (rc == 0 && par) {
    rc = dev_alloc(ctx, &S->wa, S->n_loc);
    if (rc == 0) rc = dev_alloc(ctx, &S->wb, S->n_loc);
    if (rc == 0) rc = dev_alloc(ctx, &S->wc, S->n_loc);
    if (rc == 0 && hipMemsetAsync(S->wa, 0, sizeof(double) * S->n_loc, ctx->stream) != hipSuccess) rc = ALFI_E_HIP;
    // pressure dofs of all ranks: one all-reduce at setup
    if (rc == 0) {
      const double mine = (double)B->nrows;
      if (hipMemcpyAsync(ctx->dred + RED_MAXV, &mine, sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess)
        rc = alfi_set_error(ctx, ALFI_E_HIP, "alfi_saddle_create: copy to the reduction buffer failed");
    }
    if (rc == 0) {
      ctx->cur_tag = F->id;
      rc = comm_allreduce(F, RED_MAXV, 1);
    }
    if (rc == 0) {
      double tot = 0.0;
      if (hipMemcpyAsync(&tot, ctx->dred + RED_MAXV, sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess)
        rc = alfi_set_error(ctx, ALFI_E_HIP, "alfi_saddle_create: copy from the reduction buffer failed");
      S->np_global = tot;
    }
  }
  if (rc != 0) {
    alfi_saddle_destroy(S);
    return rc;
  }
  *out = S;
  return 0;
}

int alfi_saddle_destroy(alfi_saddle* S) {
  if (!S) return 0;
  (void)hipStreamSynchronize(S->ctx->stream);
  free_csr(&S->B);
  free_csr(&S->BT);
  if (S->has_Minv) free_csr(&S->Minv);
  dev_free(S->minv);
  dev_free(S->V);
  dev_free(S->Z);
  dev_free(S->w);
  dev_free(S->hs);
  dev_free(S->tmp_u);
  dev_free(S->tmp_p);
  dev_free(S->wa);
  dev_free(S->wb);
  dev_free(S->wc);
  dev_free(S->dotbuf);
  delete S;
  return 0;
}

int alfi_saddle_update(alfi_saddle* S, double nu, double gamma) {
  S->nu = nu;
  S->gamma = gamma;
  return 0;
}

// y = [A B^T; B 0] x
int alfi_saddle_mult(alfi_saddle* S, const double* dx, double* dy) {
  alfi_ctx* ctx = S->ctx;
  const int64_t nu = S->nu_dofs;
  if (S->par) {
    alfi_level* F = S->fine;
    ctx->cur_tag = F->id;
    ALFI_CHECK(launch_copy(ctx, S->wa, dx, nu));
    ALFI_CHECK(level_spmv(F, S->wa, S->wb, nullptr, 0));                          // ghosts of wa filled; owned rows of A
    ALFI_CHECK(launch_csr_spmv(ctx, S->B, S->wa, dy + nu, nullptr, 0.0, 0));      // y_p = B u (needs the ghosts)
    ALFI_CHECK(launch_csr_spmv(ctx, S->BT, dx + nu, S->wc, nullptr, 0.0, 0));     // partial B^T p on all local dofs
    ALFI_CHECK(halo_rev(F, S->wc));                                               // ... summed onto their owners
    return launch_add(ctx, dy, S->wb, S->wc, nu);
  }
  ALFI_CHECK(alfi_spmv(S->fine, dx, dy));                                       // y_u = A x_u
  ALFI_CHECK(launch_csr_spmv(ctx, S->BT, dx + nu, dy, nullptr, 0.0, 2));        // y_u += B^T x_p
  ALFI_CHECK(launch_csr_spmv(ctx, S->B, dx, dy + nu, nullptr, 0.0, 0));         // y_p = B x_u
  return 0;
}

// PCFIELDSPLIT, Schur, full factorisation [3P] with the sub-solvers of solver.py:359-391
int alfi_saddle_set_mass_inverse(alfi_saddle* S, const alfi_csr_host* Minv) {
  alfi_ctx* ctx = S->ctx;
  if (!Minv || Minv->nrows != S->np_dofs || Minv->ncols != S->np_dofs)
    return alfi_set_error(ctx, ALFI_E_ARG, "mass inverse must be %lld x %lld", (long long)S->np_dofs, (long long)S->np_dofs);
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (S->has_Minv) free_csr(&S->Minv);
  S->has_Minv = false;
  ALFI_CHECK(upload_csr(ctx, &S->Minv, Minv));
  S->has_Minv = true;
  return 0;
}

// y_p = -(nu + gamma) M^-1 q
static int saddle_schur(alfi_saddle* S, const double* q, double* yp) {
  if (S->has_Minv) return launch_csr_spmv(S->ctx, S->Minv, q, yp, nullptr, -(S->nu + S->gamma), 3);
  return launch_scale_rows(S->ctx, yp, q, S->minv, -(S->nu + S->gamma), S->np_dofs);
}

int alfi_saddle_precond(alfi_saddle* S, const double* dx, double* dy) {
  alfi_ctx* ctx = S->ctx;
  const int64_t nu = S->nu_dofs, np = S->np_dofs;
  if (S->par) {
    alfi_level* F = S->fine;
    ALFI_CHECK(launch_copy(ctx, S->wa, dx, nu));
    ALFI_CHECK(alfi_mg_fcycle(S->mg, S->wa, S->wb));                              // y_u = MG(b_u)
    ctx->cur_tag = F->id;
    ALFI_CHECK(halo_fwd(F, S->wb));
    ALFI_CHECK(launch_csr_spmv(ctx, S->B, S->wb, S->tmp_p, dx + nu, 1.0, 1));     // q = b_p - B y_u
    ALFI_CHECK(saddle_schur(S, S->tmp_p, dy + nu));
    ALFI_CHECK(launch_csr_spmv(ctx, S->BT, dy + nu, S->wc, nullptr, 0.0, 0));     // B^T y_p: partial sums on all local dofs
    ALFI_CHECK(halo_rev(F, S->wc));
    ALFI_CHECK(launch_xmy(ctx, S->wc, dx, nu));                                   // t = b_u - B^T y_p
    ALFI_CHECK(alfi_mg_fcycle(S->mg, S->wc, S->wb));                              // y_u = MG(t)
    ALFI_CHECK(launch_copy(ctx, dy, S->wb, nu));
    if (S->remove_nullspace) {                                                    // y_p -= (sum over all ranks) / (all pressure dofs)
      ctx->cur_tag = F->id;
      ALFI_CHECK(launch_sum_to(ctx, dy + nu, np, ctx->dred + RED_MAXV));
      ALFI_CHECK(comm_allreduce(F, RED_MAXV, 1));
      ALFI_CHECK(launch_sub_scaled(ctx, dy + nu, np, ctx->dred + RED_MAXV, 1.0 / S->np_global));
    }
    return 0;
  }
  ALFI_CHECK(alfi_mg_fcycle(S->mg, dx, dy));                                    // y_u = MG(b_u)
  ALFI_CHECK(launch_csr_spmv(ctx, S->B, dy, S->tmp_p, dx + nu, 1.0, 1));        // q = b_p - B y_u
  ALFI_CHECK(saddle_schur(S, S->tmp_p, dy + nu));                               // y_p = -(nu+gamma) M^-1 q
  ALFI_CHECK(launch_csr_spmv(ctx, S->BT, dy + nu, S->tmp_u, dx, 1.0, 1));       // t = b_u - B^T y_p
  ALFI_CHECK(alfi_mg_fcycle(S->mg, S->tmp_u, dy));                              // y_u = MG(t)
  if (S->remove_nullspace) ALFI_CHECK(launch_remove_mean(ctx, dy + nu, np));
  return 0;
}

int alfi_saddle_solve(alfi_saddle* S, const double* db, double* dx, double rtol, double atol, int max_it, int restart,
                      int* iterations, double* residual_norm) {
  alfi_ctx* ctx = S->ctx;
  alfi_level* F = S->fine;
  const bool par = S->par;
  const int64_t n = S->nu_dofs + S->np_dofs;
  if (restart < 1 || restart > RED_MAXV - 2) return alfi_set_error(ctx, ALFI_E_ARG, "restart must be in 1..%d", RED_MAXV - 2);
  if (restart != S->restart) {
    ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    dev_free(S->V);
    dev_free(S->Z);
    dev_free(S->w);
    dev_free(S->hs);
    S->V = S->Z = S->w = S->hs = nullptr;
    S->restart = 0;
    S->ldv = (n + 1) & ~(int64_t)1;           // every basis vector on a 16-byte boundary
    ALFI_CHECK(dev_alloc(ctx, &S->V, (int64_t)(restart + 1) * S->ldv));
    ALFI_CHECK(dev_alloc(ctx, &S->Z, (int64_t)restart * S->ldv));
    ALFI_CHECK(dev_alloc(ctx, &S->w, S->ldv));
    HsLayout hl0(restart);
    ALFI_CHECK(dev_alloc(ctx, &S->hs, hl0.total));
    S->restart = restart;
  }
  const int K = restart;
  const int64_t ldv = S->ldv;
  HsLayout hl(K);
  double *V = S->V, *Z = S->Z, *w = S->w, *hs = S->hs;
  // partitioned: dots and norms are reduced into the ctx's reduction buffer and all-reduced there (as in the level smoother);
  // the values every rank reads back are identical, so all ranks take the same branches
  double* hdots = par ? ctx->dred : hs + hl.hd;
  double* nrm2 = par ? ctx->dred + RED_MAXV : nullptr;
  const int G = red_blocks_for(n);
  ALFI_HIP_CHECK(ctx, hipMemsetAsync(hs, 0, sizeof(double) * hl.total, ctx->stream));
  ALFI_HIP_CHECK(ctx, hipMemsetAsync(dx, 0, sizeof(double) * n, ctx->stream));
  auto read = [&](const double* p, double* out) -> int {
    ALFI_HIP_CHECK(ctx, hipMemcpyAsync(out, p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
  };
  // beta = |w| into hs (and the rotated right-hand side)
  auto norm_init = [&]() -> int {
    ALFI_CHECK(launch_norm_partials(ctx, w, n));
    if (par) {
      ctx->cur_tag = F->id;
      ALFI_CHECK(launch_reduce_partials(ctx, ctx->red_partial, G, 1, nrm2));
      ALFI_CHECK(comm_allreduce(F, RED_MAXV, 1));
      return launch_norm_init_finish(ctx, nrm2, 1, hs, K);
    }
    return launch_norm_init_finish(ctx, ctx->red_partial, G, hs, K);
  };
  int its = 0;
  double bnorm = 0.0, rnorm = 0.0;
  // r = b (zero initial guess)
  ALFI_CHECK(launch_copy(ctx, w, db, n));
  ALFI_CHECK(norm_init());
  ALFI_CHECK(read(hs + hl.beta, &bnorm));
  rnorm = bnorm;
  const double tol = std::max(rtol * bnorm, atol);
  bool converged = rnorm <= tol;
  while (!converged && its < max_it) {
    ALFI_CHECK(launch_scale_by_inv(ctx, V, w, hs + hl.beta, n));            // v_0 = r / |r|
    int j = 0;
    for (; j < K && its < max_it; ++j) {
      double* zj = Z + (int64_t)j * ldv;
      ALFI_CHECK(alfi_saddle_precond(S, V + (int64_t)j * ldv, zj));          // z_j = P^-1 v_j
      ALFI_CHECK(alfi_saddle_mult(S, zj, w));                               // w = K z_j
      ALFI_CHECK(launch_multi_dot(ctx, V, ldv, j + 1, w, hdots, n));
      if (par) {
        ctx->cur_tag = F->id;
        ALFI_CHECK(comm_allreduce(F, 0, j + 1));
      }
      ALFI_CHECK(launch_multi_axpy_norm(ctx, V, ldv, j + 1, hdots, w, n, ctx->red_partial2, 0));
      if (par) {                                                            // |w - V h| by its own all-reduce (VecNorm)
        ALFI_CHECK(launch_reduce_partials(ctx, ctx->red_partial2, G, 1, nrm2));
        ALFI_CHECK(comm_allreduce(F, RED_MAXV, 1));
      }
      ALFI_CHECK(launch_hessenberg_update(ctx, par ? nrm2 : ctx->red_partial2, par ? 1 : G, hdots, hs, j, K, nullptr));
      ++its;
      double g = 0.0;
      ALFI_CHECK(read(hs + hl.grs + j + 1, &g));                            // |rotated rhs| = residual norm estimate
      rnorm = std::fabs(g);
      if (rnorm <= tol) {
        converged = true;
        ++j;
        break;
      }
      if (j + 1 < K) ALFI_CHECK(launch_scale_by_inv(ctx, V + (int64_t)(j + 1) * ldv, w, hs + hl.tt, n));
    }
    ALFI_CHECK(launch_fgmres_finish(ctx, hs, j, K));
    ALFI_CHECK(launch_update_solution(ctx, dx, Z, ldv, j, hs + hl.y, n));
    if (converged || its >= max_it) break;
    // restart: true residual
    ALFI_CHECK(alfi_saddle_mult(S, dx, w));
    ALFI_CHECK(launch_xmy(ctx, w, db, n));                                   // w = b - K x
    ALFI_CHECK(norm_init());
    ALFI_CHECK(read(hs + hl.beta, &rnorm));
    converged = rnorm <= tol;
  }
  // final true residual norm
  ALFI_CHECK(alfi_saddle_mult(S, dx, w));
  ALFI_CHECK(launch_xmy(ctx, w, db, n));
  ALFI_CHECK(norm_init());
  double tn = 0.0;
  ALFI_CHECK(read(hs + hl.beta, &tn));
  if (iterations) *iterations = its;
  if (residual_norm) *residual_norm = tn;
  // a persistent multiplicative sweep inside a cycle may have run into its wait bound: the iterates above were then computed
  // from an incomplete smoother result -- report it instead of returning counts and norms of garbage
  return check_dev_err(ctx);
}

}  // extern "C"
