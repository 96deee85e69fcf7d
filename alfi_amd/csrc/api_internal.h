// Shared by the api_*.hip files (the C ABI of libalfi_hip.so, include/alfi_hip.h): device allocation helpers and the functions one
// concern's file offers the others.  Not part of the ABI.  The public entry points get their C linkage from alfi_hip.h.
#pragma once
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include "common.h"
#include "hs_layout.h"

template <typename T>
inline int dev_alloc(alfi_ctx* ctx, T** p, int64_t count) {
  *p = nullptr;
  if (count <= 0) count = 1;
  ALFI_HIP_CHECK(ctx, hipMalloc((void**)p, (size_t)count * sizeof(T)));
  return 0;
}
template <typename T>
inline int dev_upload(alfi_ctx* ctx, T** p, const T* host, int64_t count) {
  ALFI_CHECK(dev_alloc(ctx, p, count));
  if (count > 0) ALFI_HIP_CHECK(ctx, hipMemcpy(*p, host, (size_t)count * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}
inline void dev_free(void* p) {
  if (p) (void)hipFree(p);
}
// an array of a level's condensed patch factors: uploaded, remembered for free_cond
template <typename T>
inline int cond_upload(alfi_level* L, const T** dst, const std::vector<T>& src) {
  T* d = nullptr;
  ALFI_CHECK(dev_upload(L->ctx, &d, src.data(), (int64_t)src.size()));
  L->cond_allocs.push_back(d);
  *dst = d;
  return 0;
}

// api_ctx.hip
int check_dev_err(alfi_ctx* ctx);      // the sticky device-side error word (bounded waits of persistent kernels), read after a sync
int build_chunk_tables(alfi_ctx* ctx, DevBSR* d, const int32_t* rowptr, int64_t nbrows, int64_t break_row,
                       int64_t* nchunks_before = nullptr);
int upload_bsr(alfi_ctx* ctx, DevBSR* d, const alfi_bsr_host* h, int bs);
void free_bsr(DevBSR* d);
void free_cond(alfi_level* L);
int comm_allreduce(alfi_level* L, int64_t offset, int64_t count);
int halo_fwd(alfi_level* L, double* v);
int halo_fwd_begin(alfi_level* L, const double* v);
int halo_fwd_end(alfi_level* L, double* v);
int halo_rev(alfi_level* L, double* v);
int halo_sum(alfi_level* L, double* v);
int halo_rev_begin(alfi_level* L, const double* v);
int halo_rev_end(alfi_level* L, double* v);
// api_level.hip
void free_assembly(AssemblyDev* S);
int level_spmv(alfi_level* L, const double* dx, double* dy, const double* db, int mode, bool ghosts_current = false);
int level_patch_apply(alfi_level* L, const double* dx, double* dy, bool* ghosts_current = nullptr);
// api_patches.hip
void free_mult_schedule(alfi_level* L);
// api_smoother.hip
int ensure_fgmres_workspace(alfi_level* L, int k);
// api_saddle.hip
int upload_csr(alfi_ctx* ctx, DevCSR* d, const alfi_csr_host* h);
void free_csr(DevCSR* d);
