// C ABI of libalfi_hip.so (include/alfi_hip.h): handles, device memory, and the stream-ordered orchestration of the
// smoother (PETSc KSPFGMRES + PCPATCH), the Schoeberl transfers and the PCMG V / full cycles.  No host synchronisation
// happens inside the smoother or the cycles: every scalar (norms, Hessenberg, Givens) stays on the device.
// This file: errors, profiling events, the helpers the other api_*.hip files share (api_internal.h), the halo / all-reduce entry
// points of partitioned levels, and the context.
#include "api_internal.h"

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

// the sticky device-side error word (bounded waits of persistent kernels), read after a synchronisation
int check_dev_err(alfi_ctx* ctx) {
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
int build_chunk_tables(alfi_ctx* ctx, DevBSR* d, const int32_t* rowptr, int64_t nbrows, int64_t break_row,
                              int64_t* nchunks_before) {
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

int upload_bsr(alfi_ctx* ctx, DevBSR* d, const alfi_bsr_host* h, int bs) {
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
void free_bsr(DevBSR* d) {
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
int comm_allreduce(alfi_level* L, int64_t offset, int64_t count) {
  alfi_ctx* ctx = L->ctx;
  int t = alfi_prof_begin(ctx, ALFI_EV_COMM);
  ALFI_CHECK(comm_call(L, ALFI_COMM_ALLREDUCE, offset, count));
  alfi_prof_end(ctx, t);
  return 0;
}

// owner -> ghost copies of level vector v (ghost slots of v are overwritten)
int halo_fwd(alfi_level* L, double* v) {
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
int halo_fwd_begin(alfi_level* L, const double* v) {
  alfi_ctx* ctx = L->ctx;
  int t = alfi_prof_begin(ctx, ALFI_EV_COMM);
  ALFI_CHECK(launch_halo_pack(ctx, L->halo_sendbuf, v, L->halo_send_nodes, L->halo_nsend, L->bs));
  ALFI_CHECK(comm_call(L, ALFI_COMM_HALO_FWD_BEGIN, 0, 0));
  alfi_prof_end(ctx, t);
  return 0;
}
int halo_fwd_end(alfi_level* L, double* v) {
  alfi_ctx* ctx = L->ctx;
  int t = alfi_prof_begin(ctx, ALFI_EV_COMM);
  ALFI_CHECK(comm_call(L, ALFI_COMM_HALO_FWD_END, 0, 0));
  // (an own copy kernel: the runtime's device-to-device copy costs more per call than these surface-sized buffers take)
  ALFI_CHECK(launch_copy(ctx, v + L->n_own, L->halo_recvbuf, L->halo_nghost * L->bs));
  alfi_prof_end(ctx, t);
  return 0;
}

// ghost contributions of v added onto their owners (ghost slots of v keep their local values)
int halo_rev(alfi_level* L, double* v) {
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
int halo_sum(alfi_level* L, double* v) {
  alfi_ctx* ctx = L->ctx;
  int t = alfi_prof_begin(ctx, ALFI_EV_COMM);
  ALFI_CHECK(launch_halo_pack(ctx, L->sum_sendbuf, v, L->sum_send_nodes, L->sum_nsend, L->bs));
  ALFI_CHECK(comm_call(L, ALFI_COMM_HALO_SUM, 0, 0));
  ALFI_CHECK(launch_halo_sum(ctx, v, L->sum_recvbuf, L->sum_nodes, L->sum_ptr, L->sum_src, L->sum_nshared, L->bs));
  alfi_prof_end(ctx, t);
  return 0;
}

// reverse route in two halves (see halo_rev): ghost slots -> buffer, start | ... | wait, add onto the owners
int halo_rev_begin(alfi_level* L, const double* v) {
  alfi_ctx* ctx = L->ctx;
  int t = alfi_prof_begin(ctx, ALFI_EV_COMM);
  ALFI_CHECK(launch_copy(ctx, L->halo_recvbuf, v + L->n_own, L->halo_nghost * L->bs));
  ALFI_CHECK(comm_call(L, ALFI_COMM_HALO_REV_BEGIN, 0, 0));
  alfi_prof_end(ctx, t);
  return 0;
}
int halo_rev_end(alfi_level* L, double* v) {
  alfi_ctx* ctx = L->ctx;
  int t = alfi_prof_begin(ctx, ALFI_EV_COMM);
  ALFI_CHECK(comm_call(L, ALFI_COMM_HALO_REV_END, 0, 0));
  ALFI_CHECK(launch_halo_add(ctx, v, L->halo_sendbuf, L->rev_nodes, L->rev_ptr, L->rev_pos, L->rev_nuniq, L->bs));
  alfi_prof_end(ctx, t);
  return 0;
}

void free_cond(alfi_level* L) {
  for (void* q : L->cond_allocs) dev_free(q);
  L->cond_allocs.clear();
  L->cd = CondDev();
  L->cond = false;
  L->h_sptr.clear();
  L->h_cond_gptr.clear();
  L->cond_ngroups = L->cond_mat_doubles = L->cond_sinv_doubles = 0;
  L->cond_lds_bytes = L->cond_max_s = L->cond_umax = 0;
}

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
