// C ABI of libalfi_hip.so (include/alfi_hip.h), the outer saddle-point solve and the vector operations of the Newton loop around the path.
// (One file per concern since round 5: api_ctx / api_level / api_patches / api_smoother / api_cycles / api_saddle; the helpers they
// share are declared in api_internal.h.)
#include "api_internal.h"

// ---- outer saddle-point solve (alfi/solver.py:386-422) ----------------------------------------------------------------------------
int upload_csr(alfi_ctx* ctx, DevCSR* d, const alfi_csr_host* h) {
  d->nrows = h->nrows;
  d->ncols = h->ncols;
  d->nnz = h->rowptr[h->nrows];
  ALFI_CHECK(dev_upload(ctx, &d->rowptr, h->rowptr, h->nrows + 1));
  ALFI_CHECK(dev_upload(ctx, &d->colidx, h->colidx, d->nnz));
  ALFI_CHECK(dev_upload(ctx, &d->vals, h->vals, d->nnz));
  return 0;
}
void free_csr(DevCSR* d) {
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
