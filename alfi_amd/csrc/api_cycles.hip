// C ABI of libalfi_hip.so (include/alfi_hip.h), coarse solve, Schoeberl transfers, PCMG V / full cycles and their hipGraph replay.
// (One file per concern since round 5: api_ctx / api_level / api_patches / api_smoother / api_cycles / api_saddle; the helpers they
// share are declared in api_internal.h.)
#include "api_internal.h"

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
