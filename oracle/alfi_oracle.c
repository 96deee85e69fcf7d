/* TEST INFRASTRUCTURE ONLY -- plain C (OpenMP) restatement of the reference's multigrid hot path, used (a) as a second
 * checker next to oracle/alfi_oracle.py and (b) as the timed CPU baseline of bench.py (cpu_baseline.kind = "port").
 *
 * PARITY UNPINNED: the reference has no tests or golden vectors and its arithmetic lives in un-vendored PETSc /
 * Firedrake code; this file follows the reference's call sites:
 *   - MatMult on the BAIJ level operator                      alfi/solver.py:512
 *   - PCPATCH setup: A_p = A[dofs_p, dofs_p], dense inverse    alfi/solver.py:320, 599-602
 *   - PCPATCH additive apply, y[bc] = x[bc]                    alfi/solver.py:318-328
 *   - KSPFGMRES(k), classical Gram-Schmidt, no convergence test alfi/solver.py:314-317
 *   - coarse-cell block solves of the Schoeberl transfer       alfi/transfer.py:254-257, 267-270
 * The product (alfi_amd/) never links or loads this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

int oracle_num_threads(void) { return omp_get_max_threads(); }
int oracle_set_num_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
}

/* NUMA first touch for the timed baseline: copy into freshly allocated (untouched) memory with the static schedule the
 * kernels below use, so every thread later reads mostly node-local pages (PETSc gets the same from MPI-rank-local
 * allocation). */
void oracle_parallel_copy(void* dst, const void* src, int64_t nbytes) {
  const int64_t chunk = 1 << 16;
  const int64_t nchunk = (nbytes + chunk - 1) / chunk;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < nchunk; ++c) {
    const int64_t o = c * chunk;
    memcpy((char*)dst + o, (const char*)src + o, (size_t)(nbytes - o < chunk ? nbytes - o : chunk));
  }
}

/* y = A x (mode 0) or y = b - alpha A x (mode 1); A block-CSR with bs x bs row-major blocks */
void oracle_bsr_spmv(int64_t nbrows, int bs, const int32_t* rowptr, const int32_t* colidx, const double* vals,
                     const double* x, double* y, const double* b, double alpha, int mode) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nbrows; ++r) {
    double acc[3] = {0, 0, 0};
    if (bs == 3) {                       /* the hot case, unrolled so that the compiler keeps it in registers */
      double a0 = 0, a1 = 0, a2 = 0;
      for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const double* v = vals + (int64_t)k * 9;
        const double* xv = x + (int64_t)colidx[k] * 3;
        a0 += v[0] * xv[0] + v[1] * xv[1] + v[2] * xv[2];
        a1 += v[3] * xv[0] + v[4] * xv[1] + v[5] * xv[2];
        a2 += v[6] * xv[0] + v[7] * xv[1] + v[8] * xv[2];
      }
      acc[0] = a0; acc[1] = a1; acc[2] = a2;
    } else {
      for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const double* v = vals + (int64_t)k * bs * bs;
        const double* xv = x + (int64_t)colidx[k] * bs;
        for (int i = 0; i < bs; ++i)
          for (int j = 0; j < bs; ++j) acc[i] += v[i * bs + j] * xv[j];
      }
    }
    for (int i = 0; i < bs; ++i) y[r * bs + i] = mode ? b[r * bs + i] - alpha * acc[i] : acc[i];
  }
}

/* in-place inverse of a dense n x n row-major matrix by Gauss-Jordan with partial pivoting (what getrf + getri
 * deliver up to rounding); work: n ints.  returns 0, or -1 if singular */
static int dense_inverse(int n, double* a, int* piv) {
  for (int k = 0; k < n; ++k) {
    int p = k;
    double best = fabs(a[k * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (fabs(a[i * n + k]) > best) { best = fabs(a[i * n + k]); p = i; }
    if (best == 0.0) return -1;
    piv[k] = p;
    if (p != k)
      for (int j = 0; j < n; ++j) { double t = a[k * n + j]; a[k * n + j] = a[p * n + j]; a[p * n + j] = t; }
    const double ip = 1.0 / a[k * n + k];
    a[k * n + k] = 1.0;
    for (int j = 0; j < n; ++j) a[k * n + j] *= ip;
    for (int i = 0; i < n; ++i) {
      if (i == k) continue;
      const double f = a[i * n + k];
      if (f == 0.0) continue;
      a[i * n + k] = 0.0;
      for (int j = 0; j < n; ++j) a[i * n + j] -= f * a[k * n + j];
    }
  }
  for (int k = n - 1; k >= 0; --k) {   /* undo the row swaps as column swaps */
    const int p = piv[k];
    if (p != k)
      for (int i = 0; i < n; ++i) { double t = a[i * n + k]; a[i * n + k] = a[i * n + p]; a[i * n + p] = t; }
  }
  return 0;
}

/* PCPATCH setup: inv + inv_ptr[p] receives inv(A[dofs_p, dofs_p]) row-major (inv_ptr[p+1]-inv_ptr[p] = n_p^2) */
int oracle_invert_patches(int64_t npatch, const int64_t* patch_ptr, const int32_t* patch_dofs, int bs,
                          const int32_t* rowptr, const int32_t* colidx, const double* vals, const int64_t* inv_ptr,
                          double* inv) {
  int err = 0;
  int max_n = 1;
  for (int64_t p = 0; p < npatch; ++p)
    if (patch_ptr[p + 1] - patch_ptr[p] > max_n) max_n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
#pragma omp parallel
  {
    int* piv = (int*)malloc(sizeof(int) * max_n);
#pragma omp for schedule(dynamic, 1)
    for (int64_t p = 0; p < npatch; ++p) {
      const int32_t* dofs = patch_dofs + patch_ptr[p];
      const int n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
      double* a = inv + inv_ptr[p];
      memset(a, 0, sizeof(double) * (size_t)n * n);
      for (int i = 0; i < n; ++i) {
        const int64_t r = dofs[i] / bs, rc = dofs[i] % bs;
        for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
          const int64_t c0 = (int64_t)colidx[k] * bs;
          int lo = 0, hi = n;                 /* first local dof >= c0 (dofs ascending) */
          while (lo < hi) { int mid = (lo + hi) / 2; if (dofs[mid] < c0) lo = mid + 1; else hi = mid; }
          for (; lo < n && dofs[lo] < c0 + bs; ++lo) a[i * n + lo] = vals[(int64_t)k * bs * bs + rc * bs + (dofs[lo] - c0)];
        }
      }
      if (dense_inverse(n, a, piv) != 0) err = 1;
    }
    free(piv);
  }
  return err ? -1 : 0;
}

/* PCApply_PATCH additive: y = sum_p R_p^T inv_p R_p x, y[bc] = x[bc].  stage: sum n_p doubles; dof_ptr/dof_pos:
 * CSR dof -> staged positions (fixed summation order, as in the product). */
void oracle_patch_apply(int64_t npatch, const int64_t* patch_ptr, const int32_t* patch_dofs, const int64_t* inv_ptr,
                        const double* inv, int64_t n, const int32_t* dof_ptr, const int32_t* dof_pos,
                        const int32_t* bc_dofs, int64_t nbc, const double* x, double* y, double* stage) {
  int max_n = 1;
  for (int64_t p = 0; p < npatch; ++p)
    if (patch_ptr[p + 1] - patch_ptr[p] > max_n) max_n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
  const int chunk = max_n > 256 ? 1 : 16;
#pragma omp parallel
  {
    double* xp = (double*)malloc(sizeof(double) * max_n);
#pragma omp for schedule(dynamic, chunk)
    for (int64_t p = 0; p < npatch; ++p) {
      const int32_t* dofs = patch_dofs + patch_ptr[p];
      const int np = (int)(patch_ptr[p + 1] - patch_ptr[p]);
      const double* a = inv + inv_ptr[p];
      for (int j = 0; j < np; ++j) xp[j] = x[dofs[j]];
      double* out = stage + patch_ptr[p];
      for (int i = 0; i < np; ++i) {     /* dgemv, row-major: vectorised dot products (PETSc calls BLAS here) */
        double s = 0.0;
        const double* ai = a + (int64_t)i * np;
#pragma omp simd reduction(+ : s)
        for (int j = 0; j < np; ++j) s += ai[j] * xp[j];
        out[i] = s;
      }
    }
    free(xp);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      double s = 0.0;
      for (int32_t q = dof_ptr[i]; q < dof_ptr[i + 1]; ++q) s += stage[dof_pos[q]];
      y[i] = s;
    }
  }
  for (int64_t i = 0; i < nbc; ++i) y[bc_dofs[i]] = x[bc_dofs[i]];
}

static double dotp(const double* a, const double* b, int64_t n) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

typedef struct {
  int64_t nbrows; int bs; const int32_t* rowptr; const int32_t* colidx; const double* vals;
  int64_t npatch; const int64_t* patch_ptr; const int32_t* patch_dofs; const int64_t* inv_ptr; const double* inv;
  const int32_t* dof_ptr; const int32_t* dof_pos; const int32_t* bc_dofs; int64_t nbc; double* stage;
} oracle_level;

/* KSPFGMRES(k), right-preconditioned by the additive patch smoother, classical Gram-Schmidt, exactly k iterations.
 * work: (2k + 2) * n doubles.  x updated in place. */
void oracle_fgmres(const oracle_level* L, int k, const double* b, double* x, int nonzero_guess, double* work) {
  const int64_t n = L->nbrows * L->bs;
  double* V = work;                     /* (k+1) x n */
  double* Z = work + (int64_t)(k + 1) * n; /* k x n */
  double* w = Z + (int64_t)k * n;
  double H[33][32], cs[32], sn[32], grs[33], yv[32], h[33];
  if (nonzero_guess)
    oracle_bsr_spmv(L->nbrows, L->bs, L->rowptr, L->colidx, L->vals, x, w, b, 1.0, 1);
  else {
    memcpy(w, b, sizeof(double) * n);
    memset(x, 0, sizeof(double) * n);
  }
  const double beta = sqrt(dotp(w, w, n));
  if (beta == 0.0) return;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) V[i] = w[i] / beta;
  memset(grs, 0, sizeof(grs));
  grs[0] = beta;
  int its = 0;
  for (int j = 0; j < k; ++j) {
    double* zj = Z + (int64_t)j * n;
    oracle_patch_apply(L->npatch, L->patch_ptr, L->patch_dofs, L->inv_ptr, L->inv, n, L->dof_ptr, L->dof_pos,
                       L->bc_dofs, L->nbc, V + (int64_t)j * n, zj, L->stage);
    oracle_bsr_spmv(L->nbrows, L->bs, L->rowptr, L->colidx, L->vals, zj, w, NULL, 0.0, 0);
    for (int i = 0; i <= j; ++i) h[i] = dotp(V + (int64_t)i * n, w, n);
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < n; ++q) {
      double s = w[q];
      for (int i = 0; i <= j; ++i) s -= h[i] * V[(int64_t)i * n + q];
      w[q] = s;
    }
    const double tt = sqrt(dotp(w, w, n));
    for (int i = 0; i <= j; ++i) H[i][j] = h[i];
    H[j + 1][j] = tt;
    its = j + 1;
    for (int i = 0; i < j; ++i) {
      const double t = H[i][j];
      H[i][j] = cs[i] * t + sn[i] * H[i + 1][j];
      H[i + 1][j] = -sn[i] * t + cs[i] * H[i + 1][j];
    }
    const double den = hypot(H[j][j], H[j + 1][j]);
    if (den == 0.0) break;
    cs[j] = H[j][j] / den;
    sn[j] = H[j + 1][j] / den;
    grs[j + 1] = -sn[j] * grs[j];
    grs[j] = cs[j] * grs[j];
    H[j][j] = den;
    H[j + 1][j] = 0.0;
    if (tt == 0.0) break;
    if (j + 1 < k) {
      double* vn = V + (int64_t)(j + 1) * n;
#pragma omp parallel for schedule(static)
      for (int64_t q = 0; q < n; ++q) vn[q] = w[q] / tt;
    }
  }
  for (int i = its - 1; i >= 0; --i) {
    double s = grs[i];
    for (int q = i + 1; q < its; ++q) s -= H[i][q] * yv[q];
    yv[i] = s / H[i][i];
  }
#pragma omp parallel for schedule(static)
  for (int64_t q = 0; q < n; ++q) {
    double s = x[q];
    for (int i = 0; i < its; ++i) s += yv[i] * Z[(int64_t)i * n + q];
    x[q] = s;
  }
}

/* out[blk*m + i] = sum_j binv[blk][i][j] * in[...]; binv row-major (nblk, m, m); gather: in indexed by blk_dofs */
void oracle_block_gemv(int64_t nblk, int m, const double* binv, const int32_t* blk_dofs, const double* in, double* out,
                       int gather) {
#pragma omp parallel
  {
    double* xb = (double*)malloc(sizeof(double) * m);
#pragma omp for schedule(static)
    for (int64_t b = 0; b < nblk; ++b) {
      for (int j = 0; j < m; ++j) xb[j] = gather ? in[blk_dofs[b * m + j]] : in[b * m + j];
      for (int i = 0; i < m; ++i) {
        double s = 0.0;
        for (int j = 0; j < m; ++j) s += binv[(b * m + i) * m + j] * xb[j];
        out[b * m + i] = s;
      }
    }
    free(xb);
  }
}

/* binv[b] = inv(nu K[b] + gamma D[b]) */
int oracle_block_invert(int64_t nblk, int m, const double* K, const double* D, double nu, double gamma, double* binv) {
  int err = 0;
#pragma omp parallel
  {
    int* piv = (int*)malloc(sizeof(int) * m);
#pragma omp for schedule(static)
    for (int64_t b = 0; b < nblk; ++b) {
      double* a = binv + b * m * m;
      for (int e = 0; e < m * m; ++e) a[e] = nu * K[b * m * m + e] + gamma * D[b * m * m + e];
      if (dense_inverse(m, a, piv) != 0) err = 1;
    }
    free(piv);
  }
  return err ? -1 : 0;
}
