// FGMRES(k) + additive patch smoother of a TINY level as ONE launch of ONE workgroup (PETSc KSPFGMRES around PCPATCH,
// alfi/solver.py:313-328, with k = 6 / 10 iterations, solver.py:309-317).
//
// On the lowest levels of a hierarchy (ldc2d: 1 250 and 4 802 dofs under 1.2 M) a smoother call is a chain of ~30 dependent
// launches of 5-8 us each plus ~3 us between them, although the data of the whole level -- operator, patch inverses, 2k+2
// vectors -- is a megabyte or two.  Here 1024 threads of one workgroup run the whole call: every phase of the fused
// iteration of kernels_vec.hip (patch solves of the un-normalised direction | dof-wise sums + normalisation | w = A z_j with
// the Gram-Schmidt dots | projection + norm) separated by __syncthreads() only, the reductions in LDS, the Hessenberg /
// Givens recurrences on one thread.  No grid-wide barrier (4-16 us on this chip), no launch boundary.
// One CU streams ~150 GB/s from L2, so the path is taken only while a level's operator + inverses stay below a few MB
// (alfi_smooth_fgmres decides); above that the multi-workgroup launch chain wins.
//
// Same algorithm and data as smooth_fgmres_fused (api.hip): right-preconditioned FGMRES, classical Gram-Schmidt, explicit
// norms, the normalisation of the new Krylov vector moved behind the (linear) patch solves; only the order of the
// floating-point sums inside the reductions differs.
#include "common.h"
#include "hs_layout.h"

namespace {

constexpr int TINY_THREADS = 1024;
constexpr int TINY_WAVES = TINY_THREADS / 64;
constexpr int TINY_NV = 16;       // k + 1 <= 16 (as for the fused iteration)

struct TinyArgs {
  int64_t n, nbrows, npatch;
  int k, K, nonzero_guess, lpr, G, pou;
  const int32_t* rowptr;
  const int32_t* colflag;
  const double* vals;
  const int64_t* patch_ptr;
  const int32_t* patch_dofs;
  const int64_t* inv_ptr;
  const int64_t* stage_ptr;
  const double* inv;
  double* stage;
  const int32_t* dof_ptr;
  const int32_t* dof_pos;
  const uint8_t* bc_mask;
  const double* b;
  double* x;
  double* V;
  double* Z;
  double* w;
  double* hs;
};

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// sums of nv <= TINY_NV per-thread values over the workgroup, in one fixed order; every thread gets all of them in out[]
__device__ __forceinline__ void block_sums(double (&acc)[TINY_NV], int nv, double (*red)[TINY_NV], double* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int v = 0; v < TINY_NV; ++v) {
    if (v < nv) {
      const double s = wave_sum64(acc[v]);
      if (lane == 0) red[wave][v] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x < nv) {
    double s = 0.0;
#pragma unroll
    for (int wv = 0; wv < TINY_WAVES; ++wv) s += red[wv][threadIdx.x];
    out[threadIdx.x] = s;
  }
  __syncthreads();
}

// y = A x (mode 0) or y = b - A x (mode 1) on all block rows, lpr lanes per block row; mode 2: y = A x and acc[v] += V_v . y
template <int BS>
__device__ __forceinline__ void tiny_spmv(const TinyArgs& a, const double* __restrict__ x, double* __restrict__ y, int mode,
                                          int nv, double (&acc)[TINY_NV]) {
  constexpr int BB = BS * BS;
  const int lpr = a.lpr;
  const int l = threadIdx.x & (lpr - 1);
  const int rows_per_pass = TINY_THREADS / lpr;
  for (int64_t row0 = 0; row0 < a.nbrows; row0 += rows_per_pass) {     // uniform trip count: the shuffles need all lanes
    const int64_t row = row0 + threadIdx.x / lpr;
    double s[BS];
#pragma unroll
    for (int r = 0; r < BS; ++r) s[r] = 0.0;
    if (row < a.nbrows) {
      const int32_t lo = a.rowptr[row], hi = a.rowptr[row + 1];
      for (int32_t k = lo + l; k < hi; k += lpr) {
        const int64_t col = a.colflag[k] & 0x7fffffff;
        const double* vb = a.vals + ((int64_t)k >> 6) * (64 * BB);
        const int kl = k & 63;
        double m[BB], xv[BS];
#pragma unroll
        for (int q = 0; q < BB / 2; ++q) {
          const double2 t = *(reinterpret_cast<const double2*>(vb + q * 128) + kl);
          m[2 * q] = t.x;
          m[2 * q + 1] = t.y;
        }
        if (BB & 1) m[BB - 1] = vb[(BB / 2) * 128 + kl];
#pragma unroll
        for (int c = 0; c < BS; ++c) xv[c] = x[col * BS + c];
#pragma unroll
        for (int r = 0; r < BS; ++r)
#pragma unroll
          for (int c = 0; c < BS; ++c) s[r] = __builtin_fma(m[r * BS + c], xv[c], s[r]);
      }
    }
    for (int o = lpr >> 1; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < BS; ++r) s[r] += __shfl_xor(s[r], o);
    if (row < a.nbrows && l == 0) {
#pragma unroll
      for (int r = 0; r < BS; ++r) {
        const int64_t i = row * BS + r;
        if (mode == 1) {
          y[i] = a.b[i] - s[r];
        } else {
          y[i] = s[r];
          if (mode == 2) {
#pragma unroll
            for (int v = 0; v < TINY_NV; ++v)
              if (v < nv) acc[v] = __builtin_fma(a.V[(int64_t)v * a.n + i], s[r], acc[v]);
          }
        }
      }
    }
  }
}

// stage <- inv(A_p) x_p for every patch: G lanes per patch (a power of two, 8 .. 64, 2 G >= max_np / 2 ... see the launcher),
// lane l owns the row pairs 2l, 2l + 2G, ... of the row-piece layout (patch_inv_index).  x_p is fetched ONCE per group -- lane
// l loads the entries l, l + G, l + 2G, l + 3G -- and broadcast by shuffles, so that a patch costs two dependent global
// loads (index, value) and then only independent loads of the inverse: with one workgroup on the whole level nothing else
// hides the latency of a longer chain.
__device__ __forceinline__ void tiny_patch_solve(const TinyArgs& a, const double* __restrict__ x) {
  const int G = a.G;
  const int l = threadIdx.x & (G - 1);
  const int per_pass = TINY_THREADS / G;
  const int64_t npass = (a.npatch + per_pass - 1) / per_pass;
  for (int64_t pass = 0; pass < npass; ++pass) {          // uniform trip count: the shuffles need all lanes
    const int64_t p = pass * per_pass + threadIdx.x / G;
    const bool live = p < a.npatch;
    int n = 0;
    int64_t off = 0;
    if (live) {
      off = a.patch_ptr[p];
      n = (int)(a.patch_ptr[p + 1] - off);
    }
    const int ld = (n + 1) & ~1;
    double xr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xr[u] = (l + u * G < n) ? x[a.patch_dofs[off + l + u * G]] : 0.0;
    const double* ip = a.inv + (live ? a.inv_ptr[p] : 0);
    double* st = a.stage + (live ? a.stage_ptr[p] : 0);
    int nmax = n;                                          // longest patch among the groups of this wave
    for (int o = G; o < 64; o <<= 1) nmax = max(nmax, __shfl_xor(nmax, o));
    const int npairs = (nmax + 2 * G - 1) / (2 * G);       // row-pair rounds (1 up to 2 G dofs)
    for (int rp = 0; rp < npairs; ++rp) {
      const int r = 2 * l + rp * 2 * G;
      const bool active = live && r < ld;
      const double* base = ip;
      int rows = 2;
      if (active) {
        base = ip + patch_inv_index(r, 0, n, ld);
        rows = (int)(patch_inv_index(r, 1, n, ld) - patch_inv_index(r, 0, n, ld));
      }
      double acc0 = 0.0, acc1 = 0.0;
      for (int c0 = 0; c0 < nmax; c0 += 4) {
        double2 v[4];
        double xc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = c0 + u;
          const int src = c & (G - 1), which = c / G;      // entry c sits in register `which` of lane `src` of the group
          const double x0 = __shfl(xr[0], src, G), x1 = __shfl(xr[1], src, G), x2 = __shfl(xr[2], src, G),
                       x3 = __shfl(xr[3], src, G);
          xc[u] = which == 0 ? x0 : which == 1 ? x1 : which == 2 ? x2 : x3;
          v[u] = (active && c < n) ? *reinterpret_cast<const double2*>(base + (int64_t)c * rows) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc0 = __builtin_fma(v[u].x, xc[u], acc0);
          acc1 = __builtin_fma(v[u].y, xc[u], acc1);
        }
      }
      if (active) *reinterpret_cast<double2*>(st + r) = make_double2(acc0, acc1);
    }
  }
}

template <int BS>
__global__ __launch_bounds__(TINY_THREADS) void smooth_tiny_kernel(TinyArgs a) {
  __shared__ double red[TINY_WAVES][TINY_NV];
  __shared__ double sums[TINY_NV];
  __shared__ double ysol[TINY_NV];
  // the Hessenberg / Givens data of the call lives in LDS: its recurrences run on ONE thread while the others wait at the
  // next barrier, and in global memory every dependent access of that chain costs a microsecond (measured: 38 us per
  // iteration on a 1 250-dof level, most of it this chain)
  __shared__ double hsl[8 + TINY_NV * (TINY_NV + 1) + 6 * TINY_NV + 8];
  const int t = threadIdx.x;
  const int64_t n = a.n;
  const int k = a.k, K = a.K;
  HsLayout hl(K);
  double acc[TINY_NV];
#pragma unroll
  for (int v = 0; v < TINY_NV; ++v) acc[v] = 0.0;

  // r0 = b - A x (MatMult) or r0 = b with x = 0
  if (a.nonzero_guess) {
    tiny_spmv<BS>(a, a.x, a.w, 1, 0, acc);
  } else {
    for (int64_t i = t; i < n; i += TINY_THREADS) {
      a.w[i] = a.b[i];
      a.x[i] = 0.0;
    }
  }
  __syncthreads();
  acc[0] = 0.0;
  for (int64_t i = t; i < n; i += TINY_THREADS) acc[0] = __builtin_fma(a.w[i], a.w[i], acc[0]);
  block_sums(acc, 1, red, sums);
  double norm2 = sums[0];

  for (int j = 0; j < k; ++j) {
    double* zj = a.Z + (int64_t)j * n;
    double* vj = a.V + (int64_t)j * n;
    const double tt = sqrt(norm2);
    const double f = tt != 0.0 ? 1.0 / tt : 0.0;
    if (t == 0) {
      if (j == 0) {                       // beta = |r0|, rotated rhs = beta e_1
        hsl[hl.beta] = tt;
        hsl[hl.grs] = tt;
        for (int i = 1; i <= K; ++i) hsl[hl.grs + i] = 0.0;
      } else {
        hessenberg_column(hsl, K, j - 1, sums, tt);       // sums = the dots h of iteration j - 1 (untouched since)
      }
    }
    tiny_patch_solve(a, a.w);                                           // stage <- patch solves of w
    __syncthreads();
    for (int64_t i = t; i < n; i += TINY_THREADS) {                     // z_j = f * M^-1 w, v_j = f * w
      double s = 0.0;
      const int32_t q0 = a.dof_ptr[i], q1 = a.dof_ptr[i + 1];
      for (int32_t q = q0; q < q1; ++q) s += a.stage[a.dof_pos[q]];
      if (a.pou && q1 - q0 > 1) s /= (double)(q1 - q0);
      const double wi = a.w[i];
      zj[i] = f * (a.bc_mask[i] ? wi : s);
      vj[i] = f * wi;
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < TINY_NV; ++v) acc[v] = 0.0;
    tiny_spmv<BS>(a, zj, a.w, 2, j + 1, acc);                           // w = A z_j, partial V_i . w
    // h = V^T w; the norm of the projected w rides in the next reduction.  (sums[] is read by thread 0 above only in the
    // NEXT iteration's head, i.e. after the barriers below.)
    double hsum[TINY_NV];
    {
      __shared__ double hsh[TINY_NV];
      block_sums(acc, j + 1, red, hsh);
#pragma unroll
      for (int v = 0; v < TINY_NV; ++v) hsum[v] = v <= j ? hsh[v] : 0.0;
    }
    double nacc = 0.0;
    for (int64_t i = t; i < n; i += TINY_THREADS) {                     // w -= V h, |w|^2
      double wi = a.w[i];
#pragma unroll
      for (int v = 0; v < TINY_NV; ++v)
        if (v <= j) wi = __builtin_fma(-hsum[v], a.V[(int64_t)v * n + i], wi);
      a.w[i] = wi;
      nacc = __builtin_fma(wi, wi, nacc);
    }
    // one reduction carrying |w|^2 in slot j + 1 behind the dots (which stay available to thread 0 for the Hessenberg column)
#pragma unroll
    for (int v = 0; v < TINY_NV; ++v) acc[v] = 0.0;
    {
      const double s = wave_sum64(nacc);
      if ((t & 63) == 0) red[t >> 6][0] = s;
      __syncthreads();
      if (t == 0) {
        double tot = 0.0;
#pragma unroll
        for (int wv = 0; wv < TINY_WAVES; ++wv) tot += red[wv][0];
        for (int v = 0; v <= j; ++v) sums[v] = hsum[v];
        sums[TINY_NV - 1] = tot;
      }
      __syncthreads();
      norm2 = sums[TINY_NV - 1];
    }
  }
  if (t == 0) {
    hessenberg_column(hsl, K, k - 1, sums, sqrt(norm2));
    fgmres_back_substitution(hsl, k, K);
    for (int v = 0; v < k; ++v) ysol[v] = hsl[hl.y + v];
  }
  __syncthreads();
  for (int i = t; i < hl.total; i += TINY_THREADS) a.hs[i] = hsl[i];     // (the level's array mirrors the last call, as elsewhere)
  for (int64_t i = t; i < n; i += TINY_THREADS) {                       // x += Z y
    double xi = a.x[i];
#pragma unroll
    for (int v = 0; v < TINY_NV; ++v)
      if (v < k) xi = __builtin_fma(ysol[v], a.Z[(int64_t)v * n + i], xi);
    a.x[i] = xi;
  }
}

}  // namespace

// bytes one smoother iteration streams on this level (operator + dense patch inverses): the criterion of the one-workgroup path
int64_t tiny_level_bytes(const alfi_level* L) {
  return (int64_t)L->A_own.nnzb * (8 * L->bs * L->bs + 4) + 8 * L->inv_doubles;
}

int launch_smooth_tiny(alfi_level* L, int k, const double* db, double* dx, int nonzero_guess) {
  alfi_ctx* ctx = L->ctx;
  TinyArgs a;
  a.n = L->n;
  a.nbrows = L->A_own.nbrows;
  a.npatch = L->npatch;
  a.k = k;
  a.K = L->kmax;
  a.nonzero_guess = nonzero_guess;
  const double avg = a.nbrows > 0 ? (double)L->A_own.nnzb / (double)a.nbrows : 1.0;
  int lpr = 2;
  while (lpr < 64 && 2 * lpr <= avg) lpr <<= 1;          // ~2 blocks per lane: short dependent chains, few idle lanes
  a.lpr = lpr;
  int G = 8;
  while (G < 64 && 4 * G < L->max_np) G <<= 1;           // four x entries per lane: G >= max_np / 4 (max_np <= 160 -> G <= 64)
  a.G = G;
  a.pou = L->pou ? 1 : 0;
  a.rowptr = L->A_own.rowptr;
  a.colflag = L->A_own.colidx;
  a.vals = L->A_own.vals;
  a.patch_ptr = L->patch_ptr;
  a.patch_dofs = L->patch_dofs;
  a.inv_ptr = L->inv_ptr;
  a.stage_ptr = L->stage_ptr;
  a.inv = L->inv;
  a.stage = L->stage;
  a.dof_ptr = L->dof_ptr;
  a.dof_pos = L->dof_pos;
  a.bc_mask = L->bc_mask;
  a.b = db;
  a.x = dx;
  a.V = L->V;
  a.Z = L->Z;
  a.w = L->w;
  a.hs = L->hs;
  int t = alfi_prof_begin(ctx, ALFI_EV_KSP_TINY);
  if (L->bs == 2)
    hipLaunchKernelGGL(smooth_tiny_kernel<2>, dim3(1), dim3(TINY_THREADS), 0, ctx->stream, a);
  else
    hipLaunchKernelGGL(smooth_tiny_kernel<3>, dim3(1), dim3(TINY_THREADS), 0, ctx->stream, a);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  alfi_prof_end(ctx, t);
  return 0;
}
