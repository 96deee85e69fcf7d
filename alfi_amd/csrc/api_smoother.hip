// C ABI of libalfi_hip.so (include/alfi_hip.h), the FGMRES(k) level smoother (KSPFGMRES + PCPATCH, alfi/solver.py:309-317).
// (One file per concern since round 5: api_ctx / api_level / api_patches / api_smoother / api_cycles / api_saddle; the helpers they
// share are declared in api_internal.h.)
#include "api_internal.h"

// ---- FGMRES(k) smoother ----------------------------------------------------------------------------------------------------
int ensure_fgmres_workspace(alfi_level* L, int k) {
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
