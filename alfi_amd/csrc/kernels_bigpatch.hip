// Large patches (160 < n_p <= 4096): the macro-star patches of the reference's high-order discretisations
// (alfi/relaxation.py:163-177; 405 / 1275 dofs for Scott-Vogelius P2 / P3) need a different setup
// and a different work split than the vertex stars of kernels_patch.hip:
//
//  * setup (PCSetUp_PATCH with dense_inverse): 2 n^3 flops per patch are now a GEMM-shaped problem -- 4.1 GFLOP for
//    n = 1275, against 13 MB of matrix -- so the inversion is a BLOCKED Gauss-Jordan on a row-major scratch copy
//    (identity-padded to a multiple of 64): per 64-column step a panel kernel (one workgroup per patch: inverse of the
//    64 x 64 pivot block in LDS, the scaled pivot row panel R = D^-1 S[K,:], the saved column panel F = S[:,K] and
//    S[:,K] <- -F D^-1) and the rank-64 trailing update S -= F R on the FP64 matrix cores
//    (v_mfma_f64_16x16x4_f64: 95 % of the flops).  No pivoting across blocks, like the register kernel for small patches.
//  * apply: one WORKGROUP per patch; the row pieces of the inverse (same storage as for small patches,
//    patch_inv_index) are dealt round-robin to the four waves, x_p sits in LDS.
#include <algorithm>
#include <cstdlib>
#include <vector>
#include "common.h"

constexpr int BIG_NB = 64;          // pivot block / GEMM k-extent
constexpr int BIG_MAX_NP = PATCH_MAX;

// ---------------------------------------------------------------------------------------------------------------------
// 1. gather A_p = A[dofs_p, dofs_p] into the row-major scratch (N x N, N = n rounded up to 64, identity padding)
// ---------------------------------------------------------------------------------------------------------------------
template <int BS>
__global__ __launch_bounds__(256) void big_gather_kernel(int64_t p0, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ colidx,
                                                          const double* __restrict__ vals, int flat,
                                                          const int64_t* __restrict__ patch_ptr,
                                                          const int32_t* __restrict__ patch_dofs,
                                                          const int64_t* __restrict__ scr_ptr, double* __restrict__ scr) {
  __shared__ int32_t dofs_s[BIG_MAX_NP];
  const int64_t p = p0 + blockIdx.y;
  const int64_t off = patch_ptr[p];
  const int n = (int)(patch_ptr[p + 1] - off);
  const int N = (n + BIG_NB - 1) / BIG_NB * BIG_NB;
  double* S = scr + scr_ptr[blockIdx.y];
  for (int i = threadIdx.x; i < n; i += 256) dofs_s[i] = patch_dofs[off + i];
  // this workgroup's rows: zero + identity padding, then the matrix entries
  const int chunk = (N + gridDim.x - 1) / gridDim.x;
  const int ra = blockIdx.x * chunk, rb = min(N, ra + chunk);
  for (int64_t e = (int64_t)ra * N + threadIdx.x; e < (int64_t)rb * N; e += 256) {
    const int r = (int)(e / N), c = (int)(e % N);
    S[e] = (r == c && r >= n) ? 1.0 : 0.0;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int r = ra + wave; r < min(rb, n); r += 4) {
    const int gr = dofs_s[r];
    const int brow = gr / BS, rr = gr % BS;
    const int32_t lo = rowptr[brow], hi = rowptr[brow + 1];
    const int nent = (hi - lo) * BS;
    for (int e = lane; e < nent; e += 64) {
      const int blk = e / BS, cc = e % BS;
      const int gcol = (colidx[lo + blk] & 0x7fffffff) * BS + cc;
      int a = 0, b = n;
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (dofs_s[mid] < gcol) a = mid + 1; else b = mid;
      }
      if (a < n && dofs_s[a] == gcol) S[(int64_t)r * N + a] = vals[bsr_val_index(flat, lo + blk, rr * BS + cc, BS * BS)];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2a. panel step for pivot block K = [k0, k0 + 64), two launches:
//       big_pivot_kernel (one workgroup per patch):  D^-1 = inv(S[K,K]) by Gauss-Jordan in LDS, written to dinv;
//       big_panel_kernel (2 N/64 workgroups per patch, one 64 x 64 x 64 product each on the matrix cores):
//         column tile J:  R[:,J] = D^-1 S[K,J] saved (columns K zeroed);  S[K,J] <- R[:,J],  S[K,K] <- D^-1;
//         row tile I:     F[I] = S[I,K] saved (rows K zeroed);            S[I,K] <- -F[I] D^-1  (I != K).
//     After the trailing update S[I,J] -= F[I] R[J] (2b) the scratch holds the state of block Gauss-Jordan after step K.
//     (First version: everything in one workgroup per patch with FMA products -- 0.69 s of the 3.0 s setup of config 5.)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void big_pivot_kernel(int k0, const int64_t* __restrict__ patch_ptr, int64_t p0,
                                                         const int64_t* __restrict__ scr_ptr, const double* __restrict__ scr,
                                                         double* __restrict__ dinv, int* __restrict__ status) {
  // Gauss-Jordan on the 64 x 64 pivot block held in REGISTERS: thread (row i = t / 4, q = t % 4) owns D[i][q + 4 m],
  // m = 0..15.  Per elimination step the four threads of a row get their multiplier D[i][k] by a shuffle inside their
  // lane quad, the pivot row travels through a double-buffered LDS line, and there is ONE barrier (the first version
  // kept D in LDS with three barriers per step: 184 us per launch).  The k loop is fully unrolled so that every register
  // index is a compile-time constant.
  __shared__ double rowbuf[2][BIG_NB];
  const int64_t p = p0 + blockIdx.x;
  const int n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
  const int N = (n + BIG_NB - 1) / BIG_NB * BIG_NB;
  if (k0 >= N) return;                          // smaller patch of the batch: already done
  const double* S = scr + scr_ptr[blockIdx.x];
  const int t = threadIdx.x;
  const int gi = t >> 2, gq = t & 3, lane = t & 63;
  double a[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) a[m] = S[(int64_t)(k0 + gi) * N + k0 + gq + 4 * m];
  bool bad = false;
#pragma unroll
  for (int k = 0; k < BIG_NB; ++k) {
    const int kq = k & 3, km = k >> 2;
    const double f = __shfl(a[km], (lane & ~3) | kq, 64);     // D[gi][k]
    double* rb = rowbuf[k & 1];
    if (gi == k) {                                            // wave-uniform for 63 of 64 rows' waves: one quad diverges
      if (f == 0.0) bad = true;
      const double ip = 1.0 / f;
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        a[m] = (m == km && gq == kq) ? ip : a[m] * ip;
        rb[gq + 4 * m] = a[m];
      }
    }
    __syncthreads();
    if (gi != k) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const double rk = rb[gq + 4 * m];
        a[m] = (m == km && gq == kq) ? -f * rk : __builtin_fma(-f, rk, a[m]);
      }
    }
  }
  if (bad) atomicExch(status, 1);
  double* Dg = dinv + (int64_t)blockIdx.x * BIG_NB * BIG_NB;
#pragma unroll
  for (int m = 0; m < 16; ++m) Dg[gi * BIG_NB + gq + 4 * m] = a[m];
}

typedef double big_d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void big_panel_kernel(int k0, const int64_t* __restrict__ patch_ptr, int64_t p0,
                                                         const int64_t* __restrict__ scr_ptr, double* __restrict__ scr,
                                                         const int64_t* __restrict__ pan_ptr, double* __restrict__ panF,
                                                         double* __restrict__ panR, const double* __restrict__ dinv,
                                                         int tiles_max) {
  const int64_t p = p0 + blockIdx.y;
  const int n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
  const int N = (n + BIG_NB - 1) / BIG_NB * BIG_NB;
  const int nt = N / BIG_NB;
  if (k0 >= N) return;
  const bool col = (int)blockIdx.x < tiles_max;                 // column tile of the pivot row panel / row tile of the pivot column panel
  const int tile = col ? blockIdx.x : blockIdx.x - tiles_max;
  if (tile >= nt) return;
  double* S = scr + scr_ptr[blockIdx.y];
  double* F = panF + pan_ptr[blockIdx.y];       // (N, 64) row-major
  double* R = panR + pan_ptr[blockIdx.y];       // (64, N) row-major
  const double* Dg = dinv + (int64_t)blockIdx.y * BIG_NB * BIG_NB;
  const int t = threadIdx.x;
  const int e0 = tile * BIG_NB;
  if (e0 == k0) {                                // the pivot block itself (uniform per workgroup)
    if (col) {
      for (int e = t; e < BIG_NB * BIG_NB; e += 256) {
        S[(int64_t)(k0 + e / BIG_NB) * N + k0 + e % BIG_NB] = Dg[e];
        R[(int64_t)(e / BIG_NB) * N + k0 + e % BIG_NB] = 0.0;
      }
    } else {
      for (int e = t; e < BIG_NB * BIG_NB; e += 256) F[(int64_t)(k0 + e / BIG_NB) * BIG_NB + e % BIG_NB] = 0.0;
    }
    return;
  }
  // operands: column tile  A = D^-1 (ld 64),  B = S[K, J] (ld N);   row tile  A = S[I, K] (ld N),  B = D^-1 (ld 64)
  const double* A = col ? Dg : S + (int64_t)e0 * N + k0;
  const double* B = col ? S + (int64_t)k0 * N + e0 : Dg;
  const int lda = col ? BIG_NB : N, ldb = col ? N : BIG_NB;
  const int wave = t >> 6, lane = t & 63;
  const int r0 = (wave >> 1) * 32, c0 = (wave & 1) * 32;
  const int lm = lane & 15, lk = lane >> 4;
  if (!col)                                       // save F[I] = S[I, K] before it is overwritten
    for (int e = t; e < BIG_NB * BIG_NB; e += 256)
      F[(int64_t)(e0 + e / BIG_NB) * BIG_NB + e % BIG_NB] = S[(int64_t)(e0 + e / BIG_NB) * N + k0 + e % BIG_NB];
  big_d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (big_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int kk = 0; kk < BIG_NB; kk += 4) {
    const double a0 = A[(int64_t)(r0 + lm) * lda + kk + lk];
    const double a1 = A[(int64_t)(r0 + 16 + lm) * lda + kk + lk];
    const double b0 = B[(int64_t)(kk + lk) * ldb + c0 + lm];
    const double b1 = B[(int64_t)(kk + lk) * ldb + c0 + 16 + lm];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
  __syncthreads();                                // the product overwrites one of its operands: all reads first
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = r0 + 16 * a + lk + 4 * g, c = c0 + 16 * b + lm;
        if (col) {
          S[(int64_t)(k0 + r) * N + e0 + c] = acc[a][b][g];
          R[(int64_t)r * N + e0 + c] = acc[a][b][g];
        } else {
          S[(int64_t)(e0 + r) * N + k0 + c] = -acc[a][b][g];
        }
      }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2b. trailing update S[I,J] -= F[I] R[J] on the FP64 matrix cores: workgroup = one 64 x 64 tile of one patch, K = 64.
//     History (config-5 setup, 1765 + 303 patches): fragments read straight from global memory 0.75 s (16 TFLOP/s, 24 %
//     MFMA-busy); staging BOTH whole 64 x 64 operand tiles through LDS (64 KB per workgroup, 2 waves per SIMD) was slower,
//     and so was a 64 x 64 register tile per wave (128 accumulator VGPRs); the 16-deep double-buffered slices of
//     big_tile_product (32 KB) are faster: 0.62 s.  The rank-64 update re-streams S once per step -- (n/64) 16 n^2 bytes
//     per patch, 0.24 s of HBM time here -- so a 128-wide pivot block is what would help next.
//     XCD-aware placement was tried for this kernel and the polish products (all tiles of a patch on one XCD so that the
//     shared operand panels stay in one L2): handing each XCD a contiguous eighth of the batch 0.64 s / 0.99 s, dealing
//     patches round-robin to the XCDs 0.52 s / 0.79 s, against 0.45 s / 0.66 s with the hardware's own round-robin of
//     workgroups -- which stays.  Storing F k-major like R (so that both fragments load coalesced) and dropping the LDS
//     stage again: 0.54 s, and the transposing panel writes cost another 0.05 s.
// ---------------------------------------------------------------------------------------------------------------------
// 64 x 64 tile product on the matrix cores, shared by the trailing update (K = 64) and the polish products (K = N):
// acc += A[0:64, 0:K] B[0:K, 0:64], A row-major with leading dimension lda, B row-major with ldb.  Both operands pass
// through LDS in 16-deep k slices, double-buffered (one barrier per slice); the left operand is transposed on the way in
// (k-major, so that the 16 lanes of an MFMA A fragment read consecutive doubles); global loads are 32 B per lane along the
// rows.  Wave w owns the 32 x 32 quarter (w >> 1, w & 1) = 2 x 2 MFMA blocks.
// Operand maps (cdna_hip_programming.md): A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15],
// C/D: col = lane & 15, row = (lane >> 4) + 4 reg.
constexpr int BIG_BK = 16;
__device__ __forceinline__ void big_tile_product(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                                 int K, double (*As)[BIG_BK][BIG_NB], double (*Bs)[BIG_BK][BIG_NB],
                                                 big_d4 (&acc)[2][2]) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int wr0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;
  const int lm = lane & 15, lk = lane >> 4;
  // staging roles: A slice 64 rows x 16 k (thread: row t / 4, 4 k's), B slice 16 k x 64 columns (thread: k t / 16, 4 columns)
  const int arow = t >> 2, akq = (t & 3) * 4;
  const int bk = t >> 4, bnq = (t & 15) * 4;
  const double* Ag = A + (int64_t)arow * lda + akq;
  const double* Bg = B + (int64_t)bk * ldb + bnq;
  big_d4 ra = *reinterpret_cast<const big_d4*>(Ag);
  big_d4 rb = *reinterpret_cast<const big_d4*>(Bg);
#pragma unroll
  for (int i = 0; i < 4; ++i) As[0][akq + i][arow] = ra[i];
  *reinterpret_cast<big_d4*>(&Bs[0][bk][bnq]) = rb;
  __syncthreads();
  int cur = 0;
  for (int kk = 0; kk < K; kk += BIG_BK) {
    const bool more = kk + BIG_BK < K;
    if (more) {
      ra = *reinterpret_cast<const big_d4*>(Ag + kk + BIG_BK);
      rb = *reinterpret_cast<const big_d4*>(Bg + (int64_t)(kk + BIG_BK) * ldb);
    }
#pragma unroll
    for (int s4 = 0; s4 < BIG_BK; s4 += 4) {
      const double a0 = As[cur][s4 + lk][wr0 + lm];
      const double a1 = As[cur][s4 + lk][wr0 + 16 + lm];
      const double b0 = Bs[cur][s4 + lk][wc0 + lm];
      const double b1 = Bs[cur][s4 + lk][wc0 + 16 + lm];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < 4; ++i) As[cur ^ 1][akq + i][arow] = ra[i];
      *reinterpret_cast<big_d4*>(&Bs[cur ^ 1][bk][bnq]) = rb;
    }
    __syncthreads();
    cur ^= 1;
  }
}

__global__ __launch_bounds__(256) void big_update_kernel(int k0, const int64_t* __restrict__ patch_ptr, int64_t p0,
                                                          const int64_t* __restrict__ scr_ptr, double* __restrict__ scr,
                                                          const int64_t* __restrict__ pan_ptr,
                                                          const double* __restrict__ panF,
                                                          const double* __restrict__ panR, int tiles_max) {
  __shared__ double As[2][BIG_BK][BIG_NB];
  __shared__ double Bs[2][BIG_BK][BIG_NB];
  const int64_t p = p0 + blockIdx.y;
  const int n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
  const int N = (n + BIG_NB - 1) / BIG_NB * BIG_NB;
  const int nt = N / BIG_NB;
  if (k0 >= N) return;
  const int ti = blockIdx.x / tiles_max, tj = blockIdx.x % tiles_max;
  if (ti >= nt || tj >= nt) return;
  if (ti * BIG_NB == k0 || tj * BIG_NB == k0) return;      // F rows / R columns of the pivot block are zero: nothing to do
  double* S = scr + scr_ptr[blockIdx.y];
  const double* F = panF + pan_ptr[blockIdx.y];
  const double* R = panR + pan_ptr[blockIdx.y];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r0 = ti * BIG_NB + (wave >> 1) * 32, c0 = tj * BIG_NB + (wave & 1) * 32;
  const int lm = lane & 15, lk = lane >> 4;
  big_d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (big_d4){0.0, 0.0, 0.0, 0.0};
  // the tile of S is fetched before the product so that its latency hides behind the matrix-core work
  double sold[2][2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) sold[a][b][g] = S[(int64_t)(r0 + 16 * a + lk + 4 * g) * N + c0 + 16 * b + lm];
  big_tile_product(F + (int64_t)ti * BIG_NB * BIG_NB, BIG_NB, R + tj * BIG_NB, N, BIG_NB, As, Bs, acc);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t at = (int64_t)(r0 + 16 * a + lk + 4 * g) * N + c0 + 16 * b + lm;
        S[at] = sold[a][b][g] - acc[a][b][g];
      }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2b'. Newton-Schulz polish X <- X + X (I - A X) (two N x N x N products on the matrix cores).  Block Gauss-Jordan forms
//      the inverses of its 64 x 64 pivot blocks explicitly; with patch condition numbers of 1e7 that costs accuracy
//      (relative difference to LAPACK's inverse 2e-5 against 1e-10 for the scalar register kernel of the small patches,
//      measured).  One polish step squares the error (measured afterwards: see DESIGN.md).
//        mode 0: C = I - A B        mode 1: C = A0 + A B   (A0 = the left operand itself)
//      big_tile_product with K = N: 36 TFLOP/s, 62 % MFMA-busy (the first version, fragments straight from global memory:
//      19.5; a 64 x 128 workgroup tile with 2 x 4 MFMA blocks per wave and 48 KB of LDS: 33 -- two waves per SIMD).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void big_gemm_kernel(int mode, const int64_t* __restrict__ patch_ptr, int64_t p0,
                                                        const int64_t* __restrict__ scr_ptr,
                                                        const double* __restrict__ Aall, const double* __restrict__ Ball,
                                                        double* __restrict__ Call, int tiles_max) {
  __shared__ double As[2][BIG_BK][BIG_NB];            // [k][m]
  __shared__ double Bs[2][BIG_BK][BIG_NB];            // [k][n]
  const int64_t p = p0 + blockIdx.y;
  const int n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
  const int N = (n + BIG_NB - 1) / BIG_NB * BIG_NB;
  const int nt = N / BIG_NB;
  const int ti = blockIdx.x / tiles_max, tj = blockIdx.x % tiles_max;
  if (ti >= nt || tj >= nt) return;
  const double* A = Aall + scr_ptr[blockIdx.y];
  const double* B = Ball + scr_ptr[blockIdx.y];
  double* C = Call + scr_ptr[blockIdx.y];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wr0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;
  const int lm = lane & 15, lk = lane >> 4;
  big_d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (big_d4){0.0, 0.0, 0.0, 0.0};
  big_tile_product(A + (int64_t)ti * BIG_NB * N, N, B + tj * BIG_NB, N, N, As, Bs, acc);
  const int r0 = ti * BIG_NB + wr0, c0 = tj * BIG_NB + wc0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = r0 + 16 * a + lk + 4 * g, col = c0 + 16 * b + lm;
        const int64_t at = (int64_t)row * N + col;
        C[at] = mode == 0 ? (row == col ? 1.0 : 0.0) - acc[a][b][g] : A[at] + acc[a][b][g];
      }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2c. scratch (row-major inverse) -> the level's inverse storage (row pieces, patch_inv_index), incl. the zero pad row
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void big_store_kernel(int64_t p0, const int64_t* __restrict__ patch_ptr,
                                                         const int64_t* __restrict__ inv_ptr,
                                                         const int64_t* __restrict__ scr_ptr,
                                                         const double* __restrict__ scr, double* __restrict__ inv) {
  const int64_t p = p0 + blockIdx.y;
  const int n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
  const int N = (n + BIG_NB - 1) / BIG_NB * BIG_NB;
  const int ld = (n + 1) & ~1;
  const double* S = scr + scr_ptr[blockIdx.y];
  double* T = inv + inv_ptr[p];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)ld * n; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / n), c = (int)(e % n);
    T[patch_inv_index(r, c, n, ld)] = r < n ? S[(int64_t)r * N + c] : 0.0;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3. additive apply, stage 1: one workgroup per patch, row pieces dealt to the four waves
// ---------------------------------------------------------------------------------------------------------------------
typedef double big_d2 __attribute__((ext_vector_type(2)));
template <int G, bool NT>
__device__ __forceinline__ void big_piece(const double* __restrict__ T, int n, const double* __restrict__ xs, int lane,
                                          double* __restrict__ out) {
  constexpr int C = 64 / G, U = 8;
  const int cg = lane / G, l = lane % G;
  const double* base = T + 2 * l;
  double acc0 = 0.0, acc1 = 0.0;
  int j = cg;
  for (; j + (U - 1) * C < n; j += U * C) {
    big_d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const big_d2* q = reinterpret_cast<const big_d2*>(base + (int64_t)(j + u * C) * (2 * G));
      v[u] = NT ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double xj = xs[j + u * C];
      acc0 = __builtin_fma(v[u].x, xj, acc0);
      acc1 = __builtin_fma(v[u].y, xj, acc1);
    }
  }
  for (; j < n; j += C) {
    const big_d2* q = reinterpret_cast<const big_d2*>(base + (int64_t)j * (2 * G));
    const big_d2 v = NT ? __builtin_nontemporal_load(q) : *q;
    const double xj = xs[j];
    acc0 = __builtin_fma(v.x, xj, acc0);
    acc1 = __builtin_fma(v.y, xj, acc1);
  }
  if (C > 1) {
#pragma unroll
    for (int o = G; o < 64; o <<= 1) {
      acc0 += __shfl_xor(acc0, o);
      acc1 += __shfl_xor(acc1, o);
    }
  }
  if (cg == 0) *reinterpret_cast<double2*>(out + 2 * l) = make_double2(acc0, acc1);
}

// Rows [r0, r1) of y = T x for a wave, T in the row-piece layout (pieces of 128 rows, then the binary digits of the rest;
// piece (row0, P rows): entry (r, c) at T[row0 * n + c * P + (r - row0)]).  The range is cut into aligned power-of-two
// blocks of rows inside the stored pieces: a block of 2 G rows takes G lanes per column and 64 / G columns per load
// instruction.  Used where whole pieces dealt round-robin leave waves idle (a Schur complement of 312 rows is
// 2 x 128 + 32 + 16 + 8: two of eight waves would stream 82 % of it).
template <int G, bool NT>
__device__ __forceinline__ void big_block(const double* __restrict__ Tp, int P, int n, const double* __restrict__ xs, int lane,
                                          double* __restrict__ out, int lim) {
  constexpr int C = 64 / G, U = 8;
  const int cg = lane / G, l = lane % G;
  const double* base = Tp + 2 * l;
  double acc0 = 0.0, acc1 = 0.0;
  int j = cg;
  for (; j + (U - 1) * C < n; j += U * C) {
    big_d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const big_d2* q = reinterpret_cast<const big_d2*>(base + (int64_t)(j + u * C) * P);
      v[u] = NT ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double xj = xs[j + u * C];
      acc0 = __builtin_fma(v[u].x, xj, acc0);
      acc1 = __builtin_fma(v[u].y, xj, acc1);
    }
  }
  for (; j < n; j += C) {
    const big_d2* q = reinterpret_cast<const big_d2*>(base + (int64_t)j * P);
    const big_d2 v = NT ? __builtin_nontemporal_load(q) : *q;
    const double xj = xs[j];
    acc0 = __builtin_fma(v.x, xj, acc0);
    acc1 = __builtin_fma(v.y, xj, acc1);
  }
  if (C > 1) {
#pragma unroll
    for (int o = G; o < 64; o <<= 1) {
      acc0 += __shfl_xor(acc0, o);
      acc1 += __shfl_xor(acc1, o);
    }
  }
  if (cg == 0) {                       // lim: rows of the range that exist (the pad row of an odd n is not written)
    if (2 * l < lim) out[2 * l] = acc0;
    if (2 * l + 1 < lim) out[2 * l + 1] = acc1;
  }
}

template <bool NT>
__device__ __forceinline__ void big_rows(const double* __restrict__ T, int n, int ld, int r0, int r1,
                                         const double* __restrict__ xs, int lane, double* __restrict__ ys) {
  // ys: n doubles (LDS or global); rows >= n (the zero pad row) are not stored
  const int full = ld & ~127;
  int r = r0;
  while (r < r1) {
    int row0, P;                       // the stored piece that holds row r
    if (r < full) {
      row0 = r & ~127;
      P = 128;
    } else {
      row0 = full;
      P = 64;
      const int rem = ld - full;
      for (; P >= 2; P >>= 1)
        if (rem & P) {
          if (r < row0 + P) break;
          row0 += P;
        }
    }
    const int o = r - row0;
    int R = P;                         // the largest aligned power-of-two block at o that ends inside the piece and the range
    while (R > 2 && ((o & (R - 1)) != 0 || o + R > P || r + R > r1)) R >>= 1;
    const double* Tp = T + (int64_t)row0 * n + o;
    switch (R) {
      case 128: big_block<64, NT>(Tp, P, n, xs, lane, ys + r, n - r); break;
      case 64: big_block<32, NT>(Tp, P, n, xs, lane, ys + r, n - r); break;
      case 32: big_block<16, NT>(Tp, P, n, xs, lane, ys + r, n - r); break;
      case 16: big_block<8, NT>(Tp, P, n, xs, lane, ys + r, n - r); break;
      case 8: big_block<4, NT>(Tp, P, n, xs, lane, ys + r, n - r); break;
      case 4: big_block<2, NT>(Tp, P, n, xs, lane, ys + r, n - r); break;
      default: big_block<1, NT>(Tp, P, n, xs, lane, ys + r, n - r); break;
    }
    r += R;
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void big_apply_kernel(int64_t p0, int64_t npatch, const int64_t* __restrict__ patch_ptr,
                                                         const int32_t* __restrict__ patch_dofs,
                                                         const int64_t* __restrict__ inv_ptr,
                                                         const int64_t* __restrict__ stage_ptr,
                                                         const double* __restrict__ inv, const double* __restrict__ x,
                                                         double* __restrict__ stage, int split) {
  __shared__ double xs[BIG_MAX_NP];
  // ``split`` consecutive workgroups share a patch (levels with few patches: 303 macro stars would leave half the CUs idle)
  const int64_t p = p0 + blockIdx.x / split;
  const int part = blockIdx.x % split, nwave = 4 * split;
  if (p >= npatch) return;
  const int64_t off = patch_ptr[p];
  const int n = (int)(patch_ptr[p + 1] - off);
  for (int i = threadIdx.x; i < n; i += 256) xs[i] = x[patch_dofs[off + i]];
  __syncthreads();
  const int wave = part * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ld = (n + 1) & ~1;
  const double* T = inv + inv_ptr[p];
  double* out = stage + stage_ptr[p];
  int piece = 0, row0 = 0;
  for (; row0 + 128 <= ld; row0 += 128, ++piece)
    if (piece % nwave == wave) big_piece<64, NT>(T + (int64_t)row0 * n, n, xs, lane, out + row0);
  const int rem = ld - row0;
#define ALFI_BIG_PIECE(R)                                                                   \
  if (rem & R) {                                                                            \
    if (piece % nwave == wave) big_piece<R / 2, NT>(T + (int64_t)row0 * n, n, xs, lane, out + row0); \
    row0 += R;                                                                              \
    ++piece;                                                                                \
  }
  ALFI_BIG_PIECE(64)
  ALFI_BIG_PIECE(32)
  ALFI_BIG_PIECE(16)
  ALFI_BIG_PIECE(8)
  ALFI_BIG_PIECE(4)
  ALFI_BIG_PIECE(2)
#undef ALFI_BIG_PIECE
}

// ---------------------------------------------------------------------------------------------------------------------
// 3b. multiplicative sweep over large patches (macro stars, alfi/solver.py:322-324 with 339-342: --patch macro
//     --patch-composition multiplicative): one dependency wavefront per launch as in kernels_patch.hip, but a WORKGROUP per
//     patch.  r_p = x_p - (A y)_p: a wave per patch node, lanes over the blocks of its operator row (coalesced value
//     planes), shuffle-reduced; then y_p += inv(A_p) r_p with the row pieces dealt to the four waves as in the additive
//     apply.  Patches of one wavefront neither read what another one writes nor write the same dofs: plain stores, the
//     result is that of the sequential sweep.
// ---------------------------------------------------------------------------------------------------------------------
// one patch of a multiplicative sweep by a workgroup of 4 waves: r = (x - A y) on the patch's rows (a wave per node, lanes over
// the row's blocks), ys = inv(A_p) r by row pieces, y += ys
template <int BS, bool NT>
__device__ __forceinline__ void big_mult_patch(int64_t p, double* __restrict__ rs, double* __restrict__ ys,
                                               const int64_t* __restrict__ patch_ptr, const int32_t* __restrict__ patch_dofs,
                                               const int64_t* __restrict__ inv_ptr, const double* __restrict__ inv,
                                               const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                               const double* __restrict__ vals, int flat, const double* __restrict__ x,
                                               double* y) {
  constexpr int BB = BS * BS;
  const int64_t off = patch_ptr[p];
  const int n = (int)(patch_ptr[p + 1] - off);
  const int nn = n / BS;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = wave; i < nn; i += 4) {
    const int node = patch_dofs[off + (int64_t)i * BS] / BS;
    const int32_t lo = rowptr[node], hi = rowptr[node + 1];
    double s[BS];
#pragma unroll
    for (int r = 0; r < BS; ++r) s[r] = 0.0;
    for (int32_t k = lo + lane; k < hi; k += 64) {
      const int64_t col = colidx[k] & 0x7fffffff;
      double a[BB], yv[BS];
#pragma unroll
      for (int e = 0; e < BB; ++e) {
        const double* v = vals + bsr_val_index(flat, k, e, BB);
        a[e] = NT ? __builtin_nontemporal_load(v) : *v;
      }
#pragma unroll
      for (int c = 0; c < BS; ++c) yv[c] = y[col * BS + c];
#pragma unroll
      for (int r = 0; r < BS; ++r)
#pragma unroll
        for (int c = 0; c < BS; ++c) s[r] = __builtin_fma(a[r * BS + c], yv[c], s[r]);
    }
#pragma unroll
    for (int r = 0; r < BS; ++r) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s[r] += __shfl_xor(s[r], o);
    }
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < BS; ++r) rs[i * BS + r] = x[(int64_t)node * BS + r] - s[r];
    }
  }
  __syncthreads();
  const int ld = (n + 1) & ~1;
  const double* T = inv + inv_ptr[p];
  int piece = 0, row0 = 0;
  for (; row0 + 128 <= ld; row0 += 128, ++piece)
    if (piece % 4 == wave) big_piece<64, NT>(T + (int64_t)row0 * n, n, rs, lane, ys + row0);
  const int rem = ld - row0;
#define ALFI_BIG_PIECE(R)                                                                      \
  if (rem & R) {                                                                               \
    if (piece % 4 == wave) big_piece<R / 2, NT>(T + (int64_t)row0 * n, n, rs, lane, ys + row0); \
    row0 += R;                                                                                 \
    ++piece;                                                                                   \
  }
  ALFI_BIG_PIECE(64)
  ALFI_BIG_PIECE(32)
  ALFI_BIG_PIECE(16)
  ALFI_BIG_PIECE(8)
  ALFI_BIG_PIECE(4)
  ALFI_BIG_PIECE(2)
#undef ALFI_BIG_PIECE
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) y[patch_dofs[off + i]] += ys[i];
}


template <int BS, bool NT>
__global__ __launch_bounds__(256) void big_mult_kernel(int64_t count, const int32_t* __restrict__ seq,
                                                        const int64_t* __restrict__ patch_ptr,
                                                        const int32_t* __restrict__ patch_dofs,
                                                        const int64_t* __restrict__ inv_ptr, const double* __restrict__ inv,
                                                        const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                        const double* __restrict__ vals, int flat,
                                                        const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double rs[BIG_MAX_NP];
  __shared__ double ys[BIG_MAX_NP];
  big_mult_patch<BS, NT>(seq[blockIdx.x], rs, ys, patch_ptr, patch_dofs, inv_ptr, inv, rowptr, colidx, vals, flat, x, y);
}

// The whole (symmetrised) sweep of a level of LARGE patches (macro stars: more than 64 nodes) as ONE launch of a resident grid
// (round 5; rounds 2-4 launched once per dependency wavefront): the schedule, the ticket counter and the per-item dependency
// counters of patch_mult_persistent_kernel (kernels_patch.hip; alfi_patches_set_multiplicative builds them for every level), a
// workgroup per item.  Release / acquire at agent scope: the y entries an item reads were written by items on other CUs / XCDs.
template <int BS, bool NT>
__global__ __launch_bounds__(256) void big_mult_persistent_kernel(
    int32_t nitems, const int32_t* __restrict__ items, int32_t* __restrict__ pred, const int32_t* __restrict__ succ_ptr,
    const int32_t* __restrict__ succ, int32_t* __restrict__ head, int32_t* __restrict__ err,
    const int64_t* __restrict__ patch_ptr, const int32_t* __restrict__ patch_dofs, const int64_t* __restrict__ inv_ptr,
    const double* __restrict__ inv, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
    const double* __restrict__ vals, int flat, const double* __restrict__ x, double* y) {
  __shared__ double rs[BIG_MAX_NP];
  __shared__ double ys[BIG_MAX_NP];
  __shared__ int32_t s_ticket, s_ok;
  for (;;) {
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int32_t t = s_ticket;
    if (t >= nitems) break;
    if (threadIdx.x == 0) {
      int ok = 1;
      unsigned spins = 0;
      while (__hip_atomic_load(pred + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 24) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          ok = 0;
          break;
        }
      }
      s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) {
      if (threadIdx.x == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // every wave: forget what other CUs have rewritten
    big_mult_patch<BS, NT>(items[t], rs, ys, patch_ptr, patch_dofs, inv_ptr, inv, rowptr, colidx, vals, flat, x, y);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // this wave's y stores are out
    __syncthreads();                                            // ... and those of the other waves (rs / ys free again)
    // (the release is the fence above, once per wave: a release on every decrement costs an L2 write-back each)
    for (int32_t e = succ_ptr[t] + threadIdx.x; e < succ_ptr[t + 1]; e += 256)
      __hip_atomic_fetch_sub(pred + succ[e], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int launch_big_mult_persistent(alfi_level* L, const double* x, double* y) {
  alfi_ctx* ctx = L->ctx;
  if (ctx->big_mult_ncu == 0) {
    hipDeviceProp_t prop;
    ALFI_HIP_CHECK(ctx, hipGetDeviceProperties(&prop, ctx->device));
    const auto k2 = &big_mult_persistent_kernel<2, true>;
    const auto k3 = &big_mult_persistent_kernel<3, true>;
    ALFI_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&ctx->big_mult_per_cu[0], k2, 256, 0));
    ALFI_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&ctx->big_mult_per_cu[1], k3, 256, 0));
    ctx->big_mult_ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 64;
  }
  const int pc = std::max(1, ctx->big_mult_per_cu[L->bs == 3 ? 1 : 0]);
  dim3 grid((unsigned)std::min<int64_t>(L->mult_nitems, (int64_t)ctx->big_mult_ncu * pc)), block(256);
#define ALFI_BPM(BSV)                                                                                                     \
  hipLaunchKernelGGL((big_mult_persistent_kernel<BSV, true>), grid, block, 0, ctx->stream, L->mult_nitems, L->mult_items, \
                     L->mult_pred, L->mult_succ_ptr, L->mult_succ, L->mult_ctl, ctx->dev_err, L->patch_ptr, L->patch_dofs, \
                     L->inv_ptr, L->inv, L->A.rowptr, L->A.colidx, L->A.vals, L->A.flat, x, y)
  if (L->bs == 2) ALFI_BPM(2);
  else if (L->bs == 3) ALFI_BPM(3);
  else return alfi_set_error(ctx, ALFI_E_ARG, "unsupported block size %d", L->bs);
#undef ALFI_BPM
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

int launch_big_mult_wave(alfi_level* L, const int32_t* seq, int64_t count, const double* x, double* y) {
  alfi_ctx* ctx = L->ctx;
  if (count == 0) return 0;
  constexpr bool nt = true;      // factors are read once per apply: nontemporal loads
  dim3 grid((unsigned)count), block(256);
#define ALFI_BMULT(BSV, NTV)                                                                                          \
  hipLaunchKernelGGL((big_mult_kernel<BSV, NTV>), grid, block, 0, ctx->stream, count, seq, L->patch_ptr, L->patch_dofs, \
                     L->inv_ptr, L->inv, L->A.rowptr, L->A.colidx, L->A.vals, L->A.flat, x, y)
  if (L->bs == 2) {
    if (nt) ALFI_BMULT(2, true); else ALFI_BMULT(2, false);
  } else if (L->bs == 3) {
    if (nt) ALFI_BMULT(3, true); else ALFI_BMULT(3, false);
  } else {
    return alfi_set_error(ctx, ALFI_E_ARG, "unsupported block size %d", L->bs);
  }
#undef ALFI_BMULT
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------------------------------
// workgroups per patch: enough of them to fill 256 CUs several times over, but no more waves than 128-row pieces
static int big_split(int64_t npatch, int max_np) {
  const int64_t want = (2048 + npatch - 1) / npatch;
  const int pieces = (max_np + 127) / 128;
  const int cap = (pieces + 3) / 4;
  return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

int launch_big_apply_range(alfi_level* L, int64_t p0, int64_t p1, const double* x) {
  alfi_ctx* ctx = L->ctx;
  if (p1 <= p0) return 0;
  constexpr bool nt = true;      // factors are read once per apply: nontemporal loads
  const int split = big_split(p1 - p0, L->max_np);
  dim3 grid((unsigned)((p1 - p0) * split)), block(256);
  if (nt)
    hipLaunchKernelGGL(big_apply_kernel<true>, grid, block, 0, ctx->stream, p0, p1, L->patch_ptr, L->patch_dofs, L->inv_ptr,
                       L->stage_ptr, L->inv, x, L->stage, split);
  else
    hipLaunchKernelGGL(big_apply_kernel<false>, grid, block, 0, ctx->stream, p0, p1, L->patch_ptr, L->patch_dofs, L->inv_ptr,
                       L->stage_ptr, L->inv, x, L->stage, split);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

// nu K + gamma D of the interior blocks of a Schoeberl transfer (dense m x m inputs, transfer.py:186-232) into the scratch
__global__ __launch_bounds__(256) void big_dense_fill_kernel(int64_t p0, int m, const double* __restrict__ K,
                                                              const double* __restrict__ D, double nu, double gamma,
                                                              const int64_t* __restrict__ scr_ptr, double* __restrict__ scr) {
  const int64_t p = p0 + blockIdx.y;
  const int N = (m + BIG_NB - 1) / BIG_NB * BIG_NB;
  double* S = scr + scr_ptr[blockIdx.y];
  const double* Kp = K + p * (int64_t)m * m;
  const double* Dp = D + p * (int64_t)m * m;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)N * N; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / N), c = (int)(e % N);
    S[e] = (r < m && c < m) ? nu * Kp[(int64_t)r * m + c] + gamma * Dp[(int64_t)r * m + c] : (r == c ? 1.0 : 0.0);
  }
}

int launch_big_apply_arrays(alfi_ctx* ctx, int64_t npatch, int max_np, const int64_t* patch_ptr,
                            const int32_t* patch_dofs, const int64_t* inv_ptr, const int64_t* stage_ptr, const double* inv,
                            const double* x, double* stage) {
  if (npatch == 0) return 0;
  constexpr bool nt = true;      // factors are read once per apply: nontemporal loads
  const int split = big_split(npatch, max_np);
  dim3 grid((unsigned)(npatch * split)), block(256);
  if (nt)
    hipLaunchKernelGGL(big_apply_kernel<true>, grid, block, 0, ctx->stream, (int64_t)0, npatch, patch_ptr, patch_dofs,
                       inv_ptr, stage_ptr, inv, x, stage, split);
  else
    hipLaunchKernelGGL(big_apply_kernel<false>, grid, block, 0, ctx->stream, (int64_t)0, npatch, patch_ptr, patch_dofs,
                       inv_ptr, stage_ptr, inv, x, stage, split);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// 4. Condensed patch factors (CondDev, common.h): setup = fill (operator entries into the group matrices and the Schur
//    scratch), group step (X_g = inv(A_gg), W_g = X_g A[g, S_g]), Schur step (Sigma -= B_g W_g), blocked inversion of
//    Sigma by the kernels above; apply = three launches (cond_front / cond_sigma / cond_back below).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int COND_GMAX = 64;     // a group / its coupled skeleton set holds at most 64 entries (a lane per row)

__device__ __forceinline__ int cond_find(const int32_t* a, int n, int v) {   // position of v in the ascending list a, or -1
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return (lo < n && a[lo] == v) ? lo : -1;
}

// workgroup per patch of the batch [p0, p0 + nb): zero its group matrices and its padded Schur scratch, then scatter the
// operator entries.  Dynamic LDS: sorted dofs (n), condensed position of every sorted entry (n), group of every interior
// condensed position (nI) as int32.
template <int BS>
__global__ __launch_bounds__(256) void cond_fill_kernel(int64_t p0, CondDev cd, const int64_t* __restrict__ patch_ptr,
                                                         const int32_t* __restrict__ patch_dofs,
                                                         const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                         const double* __restrict__ vals, int flat,
                                                         const int64_t* __restrict__ scr_ptr, double* __restrict__ scr,
                                                         int* __restrict__ status) {
  extern __shared__ int32_t cond_smem[];
  const int64_t p = p0 + blockIdx.x;
  const int64_t off = patch_ptr[p];
  const int n = (int)(patch_ptr[p + 1] - off);
  const int nI = cd.p_nI[p];
  const int s = n - nI;
  const int N = (s + BIG_NB - 1) / BIG_NB * BIG_NB;
  int32_t* dofs_s = cond_smem;          // ascending dofs of the patch
  int32_t* cpos = dofs_s + n;           // sorted entry -> condensed position
  int32_t* grp = cpos + n;              // condensed interior position -> group (index relative to the patch's first group)
  const int64_t g0 = cd.gptr[p], g1 = cd.gptr[p + 1];
  for (int i = threadIdx.x; i < n; i += 256) {
    dofs_s[i] = patch_dofs[off + i];
    cpos[cd.slot[off + i]] = i;         // condensed entry i sits at sorted position slot[i]
  }
  for (int64_t g = g0 + threadIdx.x; g < g1; g += 256)
    for (int i = 0; i < cd.g_m[g]; ++i) grp[cd.g_off[g] + i] = (int)(g - g0);
  // zero the patch's group matrices (contiguous) and its Schur scratch (identity padding)
  const int64_t m0 = g1 > g0 ? cd.g_mat[g0] : 0;
  int64_t m1 = m0;
  if (g1 > g0) {
    const int64_t gl = g1 - 1;
    m1 = cd.g_mat[gl] + cond_group_doubles(cd.g_m[gl], cd.g_sc[gl]);
  }
  for (int64_t e = m0 + threadIdx.x; e < m1; e += 256) cd.mat[e] = 0.0;
  double* S = scr + scr_ptr[blockIdx.x];
  for (int64_t e = threadIdx.x; e < (int64_t)N * N; e += 256) {
    const int r = (int)(e / N), c = (int)(e % N);
    S[e] = (r == c && r >= s) ? 1.0 : 0.0;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = wave; i < n; i += 4) {            // sorted row i
    const int r = cpos[i];                       // its condensed position
    const int gr = dofs_s[i];
    const int brow = gr / BS, rr = gr % BS;
    const int32_t lo = rowptr[brow], hi = rowptr[brow + 1];
    const int nent = (hi - lo) * BS;
    for (int e = lane; e < nent; e += 64) {
      const int blk = e / BS, cc = e % BS;
      const int j = cond_find(dofs_s, n, (colidx[lo + blk] & 0x7fffffff) * BS + cc);
      if (j < 0) continue;
      const int c = cpos[j];
      const double v = vals[bsr_val_index(flat, lo + blk, rr * BS + cc, BS * BS)];
      if (r < nI && c < nI) {
        const int64_t g = g0 + grp[r];
        if (grp[c] != grp[r]) { atomicExch(status, 2); continue; }      // two groups are coupled: not a valid condensation
        const int ldm = cond_ldim(cd.g_m[g]);
        cd.mat[cd.g_mat[g] + (int64_t)(c - cd.g_off[g]) * ldm + (r - cd.g_off[g])] = v;                   // X storage <- A_gg
      } else if (r < nI) {                       // A[g, S]: into the W storage
        const int64_t g = g0 + grp[r];
        const int m = cd.g_m[g], sc = cd.g_sc[g];
        const int jj = cond_find(cd.sidx + cd.g_sidx[g], sc, c - nI);
        if (jj < 0) { atomicExch(status, 2); continue; }
        const int ldm = cond_ldim(m);
        cd.mat[cd.g_mat[g] + (int64_t)ldm * m + (int64_t)cond_ldim(sc) * m + (int64_t)jj * ldm + (r - cd.g_off[g])] = v;
      } else if (c < nI) {                       // A[S, g]: B storage (sc x m, column-major)
        const int64_t g = g0 + grp[c];
        const int m = cd.g_m[g], sc = cd.g_sc[g];
        const int jj = cond_find(cd.sidx + cd.g_sidx[g], sc, r - nI);
        if (jj < 0) { atomicExch(status, 2); continue; }
        cd.mat[cd.g_mat[g] + (int64_t)cond_ldim(m) * m + (int64_t)(c - cd.g_off[g]) * cond_ldim(sc) + jj] = v;
      } else {
        S[(int64_t)(r - nI) * N + (c - nI)] = v;
      }
    }
  }
}

// one wave per group: X = inv(A_gg) by Gauss-Jordan (lane = row, the row in registers), then W = X A[g, S_g] in place.
// GM: the register extent, the smallest of 4 / 12 / 24 / 46 / 64 that holds the group (the groups of a [P3]^3 macro star have
// 3, 9 or 45 entries; with the full 64 x 64 elimination for every group this kernel was the largest of config 5's
// re-factorisation, 84 of 173 ms).  The padding rows are identity rows: their pivots are 1 and their multipliers 0, so the
// entries of the group come out the same for every GM.
template <int GM>
__device__ __noinline__ void cond_group_body(int m, int sc, int ldm, double* __restrict__ X, double* __restrict__ W, int lane,
                                                int* __restrict__ status) {
  double a[GM], a0[GM];                      // row `lane` of the matrix being inverted / of A_gg itself
#pragma unroll
  for (int j = 0; j < GM; ++j) {
    a[j] = (lane < m && j < m) ? X[(int64_t)j * ldm + lane] : (lane == j ? 1.0 : 0.0);
    a0[j] = a[j];
  }
  bool bad = false;
#pragma unroll
  for (int k = 0; k < GM; ++k) {
    const double piv = __shfl(a[k], k, 64);
    if (piv == 0.0) bad = true;
    const double ip = 1.0 / piv;
    const double mk = (lane == k) ? 0.0 : a[k] * ip;
#pragma unroll
    for (int j = 0; j < GM; ++j) {
      const double rk = __shfl(a[j], k, 64);
      if (j == k)
        a[j] = (lane == k) ? ip : -mk;
      else
        a[j] = (lane == k) ? rk * ip : __builtin_fma(-mk, rk, a[j]);
    }
  }
  if (bad && lane == 0) atomicExch(status, 1);
  if (lane < m) {
#pragma unroll
    for (int j = 0; j < GM; ++j)
      if (j < m) X[(int64_t)j * ldm + lane] = a[j];
  }
  // W[:, c] solves A_gg w = A[g, S_g][:, c]: w = X b, then two steps of iterative refinement w += X (b - A_gg w).  The
  // explicit inverse alone leaves a residual of cond(A_gg) * eps * |b|, and |b| ~ gamma here (the grad-div coupling to
  // the skeleton): measured 2.6e-5 in the patch probe for [P3]^3 against 1e-9 with the refinement.
  for (int c = 0; c < sc; ++c) {
    const double b = lane < m ? W[(int64_t)c * ldm + lane] : 0.0;
    double w = 0.0;
#pragma unroll
    for (int k = 0; k < GM; ++k) w = __builtin_fma(a[k], __shfl(b, k, 64), w);
    for (int it = 0; it < 2; ++it) {
      double r = b;
#pragma unroll
      for (int k = 0; k < GM; ++k) r = __builtin_fma(-a0[k], __shfl(w, k, 64), r);
      if (lane >= m) r = 0.0;
      double dw = 0.0;
#pragma unroll
      for (int k = 0; k < GM; ++k) dw = __builtin_fma(a[k], __shfl(r, k, 64), dw);
      w += dw;
    }
    if (lane < m) W[(int64_t)c * ldm + lane] = w;
  }
}

__global__ __launch_bounds__(256) void cond_group_kernel(int64_t g_begin, int64_t g_end, CondDev cd, int* __restrict__ status) {
  const int64_t g = g_begin + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= g_end) return;
  const int lane = threadIdx.x & 63;
  const int m = cd.g_m[g], sc = cd.g_sc[g];
  const int ldm = cond_ldim(m);
  double* X = cd.mat + cd.g_mat[g];
  double* W = X + (int64_t)ldm * m + (int64_t)cond_ldim(sc) * m;
  if (m <= 4) cond_group_body<4>(m, sc, ldm, X, W, lane, status);
  else if (m <= 12) cond_group_body<12>(m, sc, ldm, X, W, lane, status);
  else if (m <= 24) cond_group_body<24>(m, sc, ldm, X, W, lane, status);
  else if (m <= 46) cond_group_body<46>(m, sc, ldm, X, W, lane, status);     // (48: the compiler gives up unrolling)
  else cond_group_body<COND_GMAX>(m, sc, ldm, X, W, lane, status);
}

// RARE PATH (kernels_check.hip: cond_repair): the groups of a flagged patch by LU WITH PARTIAL PIVOTING -- what the reference's
// sparse patch factorisation does (UMFPACK, alfi/solver.py:655-659); the elimination above does not pivot, and a zero or tiny
// leading pivot inside one macro-cell group used to send the whole level back to dense inverses (6.8 x the memory).  One wave
// per group: P A_gg = L U in LDS (lane = row), then X_g = inv(A_gg) and W_g = inv(A_gg) A[g, S_g] column by column through
// the factors (lane = column; backward stable per column, no refinement needed).  Dynamic LDS: m * m + 64 * m doubles, m ints.
__global__ __launch_bounds__(64) void cond_group_pivot_kernel(int64_t g_begin, CondDev cd, int* __restrict__ status) {
  extern __shared__ double cgp_s[];
  const int64_t g = g_begin + blockIdx.x;
  const int lane = threadIdx.x;
  const int m = cd.g_m[g], sc = cd.g_sc[g];
  const int ldm = cond_ldim(m);
  double* X = cd.mat + cd.g_mat[g];
  double* W = X + (int64_t)ldm * m + (int64_t)cond_ldim(sc) * m;
  double* A = cgp_s;                               // m x m, row-major: L \ U
  double* Y = A + (int64_t)m * m;                   // m x 64: work column of lane l at Y[i * 64 + l]
  int* perm = reinterpret_cast<int*>(Y + (int64_t)m * 64);
  for (int e = lane; e < m * m; e += 64) A[(e % m) * m + e / m] = X[(int64_t)(e / m) * ldm + e % m];
  __syncthreads();
  bool bad = false;
  for (int k = 0; k < m; ++k) {
    double v = (lane >= k && lane < m) ? fabs(A[lane * m + k]) : -1.0;
    if (!(v == v)) v = INFINITY;                     // NaN
    int idx = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(v, o);
      const int oi = __shfl_xor(idx, o);
      if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    if (!(v > 0.0) || v == INFINITY) { bad = true; break; }     // wave-uniform
    if (lane == 0) perm[k] = idx;
    if (idx != k && lane < m) {
      const double t = A[k * m + lane];
      A[k * m + lane] = A[idx * m + lane];
      A[idx * m + lane] = t;
    }
    __syncthreads();
    const double ip = 1.0 / A[k * m + k];
    if (lane > k && lane < m) {
      const double l = A[lane * m + k] * ip;
      A[lane * m + k] = l;
      for (int j = k + 1; j < m; ++j) A[lane * m + j] = __builtin_fma(-l, A[k * m + j], A[lane * m + j]);
    }
    __syncthreads();
  }
  if (bad) {
    if (lane == 0) atomicExch(status, 1);
    return;
  }
  // column `lane` of the right-hand side B (nrhs columns, leading dimension ldm, in place): x = U^-1 L^-1 P b
  auto solve = [&](double* B, int nrhs, bool identity) {
    const bool act = lane < nrhs;
    for (int i = 0; i < m; ++i) Y[i * 64 + lane] = act ? (identity ? (i == lane ? 1.0 : 0.0) : B[(int64_t)lane * ldm + i]) : 0.0;
    for (int k = 0; k < m; ++k) {                    // P: the row exchanges in the order they were made
      const int pk = perm[k];
      if (pk != k) {
        const double t = Y[k * 64 + lane];
        Y[k * 64 + lane] = Y[pk * 64 + lane];
        Y[pk * 64 + lane] = t;
      }
    }
    for (int i = 1; i < m; ++i) {                    // L (unit diagonal)
      double acc = Y[i * 64 + lane];
      for (int j = 0; j < i; ++j) acc = __builtin_fma(-A[i * m + j], Y[j * 64 + lane], acc);
      Y[i * 64 + lane] = acc;
    }
    for (int i = m - 1; i >= 0; --i) {               // U
      double acc = Y[i * 64 + lane];
      for (int j = i + 1; j < m; ++j) acc = __builtin_fma(-A[i * m + j], Y[j * 64 + lane], acc);
      Y[i * 64 + lane] = acc / A[i * m + i];
    }
    if (act)
      for (int i = 0; i < m; ++i) B[(int64_t)lane * ldm + i] = Y[i * 64 + lane];
  };
  solve(W, sc, false);                               // W_g = inv(A_gg) A[g, S_g]   (reads the raw A[g, S_g] the fill put there)
  solve(X, m, true);                                 // X_g = inv(A_gg)
}

// workgroup per patch of the batch: Sigma -= B_g W_g, group after group (a fixed order: deterministic)
__global__ __launch_bounds__(256) void cond_schur_kernel(int64_t p0, CondDev cd, const int64_t* __restrict__ patch_ptr,
                                                          const int64_t* __restrict__ scr_ptr, double* __restrict__ scr) {
  const int64_t p = p0 + blockIdx.x;
  const int n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
  const int s = n - cd.p_nI[p];
  const int N = (s + BIG_NB - 1) / BIG_NB * BIG_NB;
  double* S = scr + scr_ptr[blockIdx.x];
  for (int64_t g = cd.gptr[p]; g < cd.gptr[p + 1]; ++g) {
    const int m = cd.g_m[g], sc = cd.g_sc[g];
    const int ldm = cond_ldim(m), ldsc = cond_ldim(sc);
    const double* B = cd.mat + cd.g_mat[g] + (int64_t)ldm * m;
    const double* W = B + (int64_t)ldsc * m;
    const int32_t* si = cd.sidx + cd.g_sidx[g];
    for (int e = threadIdx.x; e < sc * sc; e += 256) {
      const int i = e % sc, j = e / sc;
      double acc = 0.0;
      for (int k = 0; k < m; ++k) acc = __builtin_fma(B[(int64_t)k * ldsc + i], W[(int64_t)j * ldm + k], acc);
      S[(int64_t)si[i] * N + si[j]] -= acc;
    }
    __syncthreads();
  }
}

// additive apply, stage 1, condensed patches [p0, p1): one workgroup of COND_WAVES waves per patch.  Dynamic LDS (doubles):
// xs (n: gathered x, condensed order; interior slices become t_g, the skeleton slice becomes x_S - sum B_g t_g), us
// (sum_g s_g), ys (s).  The small matrices are streamed with COND_U loads in flight per lane (they are read once).
constexpr int COND_U = 8;

// acc += sum_k M[k * ld + lane] * bcast_k, k < m, for the lanes < rows; bcast_k = lane k's value of `v`
template <bool NT>
__device__ __forceinline__ double cond_gemv_shfl(const double* __restrict__ M, int ld, int rows, int m, double v, int lane,
                                                 double acc) {
  const bool act = lane < rows;
  int k = 0;
  for (; k + COND_U <= m; k += COND_U) {
    double a[COND_U];
#pragma unroll
    for (int u = 0; u < COND_U; ++u) {
      const double* q = M + (int64_t)(k + u) * ld + lane;
      a[u] = act ? (NT ? __builtin_nontemporal_load(q) : *q) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < COND_U; ++u) acc = __builtin_fma(a[u], __shfl(v, k + u, 64), acc);
  }
  for (; k < m; ++k) {
    const double* q = M + (int64_t)k * ld + lane;
    const double a = act ? (NT ? __builtin_nontemporal_load(q) : *q) : 0.0;
    acc = __builtin_fma(a, __shfl(v, k, 64), acc);
  }
  return acc;
}

// The apply as THREE launches (rounds 1-2: one launch per patch, a wave per group with a lane per row -- dropped).  Measured on config 5's
// finest level (1765 macro stars, 4.85 GB of factors; kernel trace, same box): the one-launch kernel takes 949 us; its
// group phases -- a wave per group, a lane per row, 8-byte loads, 30-95 % of the lanes without a row for groups of 3, 9 or
// 45 entries -- run at 4.0 TB/s and its Schur phase leaves waves idle (312 rows = 2 x 128 + 32 + 16 + 8 over 8 waves).
// Split: (front) t_g = X_g x_g, u_g = B_g t_g and the Schur right-hand side, a workgroup per patch, a lane per PAIR of rows
// with the pairs of different groups side by side in a wave; (sigma) y_S = inv(Sigma) rhs with a workgroup per <= 256 ROWS
// (3 workgroups for the largest patches, the rows dealt in equal shares to the 4 waves: 6.4 TB/s); (back)
// y_g = t_g - W_g y_S[S_g] and the staging, a workgroup per patch, row pairs again.  t and y_S travel through cd.tmp
// (sum_n doubles), the right-hand side through the patch's own staging slots (overwritten by the back kernel): 16 (n + s)
// bytes of extra traffic per patch against megabytes of factors.

// wave-wide maximum of a small non-negative int
__device__ __forceinline__ int cond_wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// TWO ROWS of a group product per lane (rows 2 i, 2 i + 1 of a column-major block with an even leading dimension: one
// 16-byte load per column):   acc += sum_{k < kn} M[k * ld + (0, 1)] * v[k],   M -> the lane's rows, v -> LDS (the group's
// operand).  kn differs between the lanes of a wave; the loop runs to the wave's maximum with the loads predicated.
template <bool NT, int RU>
__device__ __forceinline__ void cond_row2_dot(const double* __restrict__ M, int ld, int kn, const double* __restrict__ v,
                                              double& acc0, double& acc1) {
  const int kmax = cond_wave_max(kn);
  for (int k = 0; k < kmax; k += RU) {
    big_d2 a[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const big_d2* q = reinterpret_cast<const big_d2*>(M + (int64_t)(k + u) * ld);
      const big_d2 z = {0.0, 0.0};
      a[u] = (k + u < kn) ? (NT ? __builtin_nontemporal_load(q) : *q) : z;
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const double vk = (k + u < kn) ? v[k + u] : 0.0;
      acc0 = __builtin_fma(a[u].x, vk, acc0);
      acc1 = __builtin_fma(a[u].y, vk, acc1);
    }
  }
}

template <bool NT, int COND_WAVES, int RU>
__global__ __launch_bounds__(64 * COND_WAVES) void cond_front_kernel(int64_t p0, int64_t p1, CondDev cd,
                                                                     const int64_t* __restrict__ patch_ptr,
                                                                     const int64_t* __restrict__ stage_ptr,
                                                                     const double* __restrict__ x, double* __restrict__ stage,
                                                                     int ordered) {
  extern __shared__ double cond_dsmem[];
  constexpr int NT_ = 64 * COND_WAVES;
  if (p0 + blockIdx.x >= p1) return;
  const int64_t p = ordered ? cd.order[blockIdx.x] : p0 + blockIdx.x;
  const int64_t off = patch_ptr[p];
  const int n = (int)(patch_ptr[p + 1] - off);
  const int nI = cd.p_nI[p];
  const int s = n - nI;
  const int64_t uo0 = cd.uptr[p];
  double* xs = cond_dsmem;          // n
  double* ts = xs + n;              // nI
  double* us = ts + nI;             // u buffer, in the order the right-hand side sums it (u_dst)
  // gather x: 4 index loads, then 4 value loads in flight per thread (the two loads of an entry depend on each other)
  for (int i0 = threadIdx.x; i0 < n; i0 += 4 * NT_) {
    int32_t d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) d[u] = (i0 + u * NT_ < n) ? cd.dofs[off + i0 + u * NT_] : 0;
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (i0 + u * NT_ < n) ? x[d[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * NT_ < n) xs[i0 + u * NT_] = v[u];
  }
  __syncthreads();
  double* tmp = cd.tmp + off;
  // t = X x, a lane per pair of interior entries (whole waves stay in cond_row2_dot: the loop bounds are uniform)
  const int64_t xp0 = cd.xp_ptr[p], bp0 = cd.bp_ptr[p];
  const int nxp = (int)(cd.xp_ptr[p + 1] - xp0), nbp = (int)(cd.bp_ptr[p + 1] - bp0);
  for (int q0 = 0; q0 < nxp; q0 += NT_) {
    const int q = q0 + threadIdx.x;
    const bool act = q < nxp;
    const int32_t g = act ? cd.xp_grp[xp0 + q] : (int32_t)cd.gptr[p];
    const int m = cd.g_m[g], o = cd.g_off[g], i = 2 * (q - cd.g_xp[g]);
    double t0 = 0.0, t1 = 0.0;
    cond_row2_dot<NT, RU>(cd.mat + cd.g_mat[g] + i, cond_ldim(m), act ? m : 0, xs + o, t0, t1);
    if (act) {
      ts[o + i] = t0;
      tmp[o + i] = t0;
      if (i + 1 < m) {
        ts[o + i + 1] = t1;
        tmp[o + i + 1] = t1;
      }
    }
  }
  __syncthreads();
  // u = B t, a lane per pair of entries of the patch's u buffer
  for (int q0 = 0; q0 < nbp; q0 += NT_) {
    const int q = q0 + threadIdx.x;
    const bool act = q < nbp;
    const int32_t g = act ? cd.bp_grp[bp0 + q] : (int32_t)cd.gptr[p];
    const int m = cd.g_m[g], sc = cd.g_sc[g], j = 2 * (q - cd.g_bp[g]);
    const int64_t ue = uo0 + cd.g_uoff[g] + j;
    const int32_t d0 = act ? cd.u_dst[ue] : 0, d1 = (act && j + 1 < sc) ? cd.u_dst[ue + 1] : 0;
    double u0 = 0.0, u1 = 0.0;
    cond_row2_dot<NT, RU>(cd.mat + cd.g_mat[g] + (int64_t)cond_ldim(m) * m + j, cond_ldim(sc), act ? m : 0,
                          ts + cd.g_off[g], u0, u1);
    if (act) {
      us[d0] = u0;
      if (j + 1 < sc) us[d1] = u1;
    }
  }
  __syncthreads();
  // right-hand side of the Schur system: the contributions to a row are adjacent in us, summed in ascending order
  const int64_t srow0 = cd.sptr[p];
  const int32_t qb = cd.s_uptr[srow0];
  double* rhs = stage + stage_ptr[p] + nI;
  for (int i = threadIdx.x; i < s; i += NT_) {
    double acc = xs[nI + i];
    const int32_t qe = cd.s_uptr[srow0 + i + 1] - qb;
    for (int32_t q = cd.s_uptr[srow0 + i] - qb; q < qe; ++q) acc -= us[q];
    rhs[i] = acc;
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void cond_sigma_kernel(int64_t c0, CondDev cd, const int64_t* __restrict__ patch_ptr,
                                                         const int64_t* __restrict__ stage_ptr,
                                                         const double* __restrict__ stage) {
  extern __shared__ double cond_dsmem[];
  const int64_t c = c0 + blockIdx.x;
  const int64_t p = cd.ch_patch[c];
  const int r0 = cd.ch_row[c];
  const int64_t off = patch_ptr[p];
  const int n = (int)(patch_ptr[p + 1] - off);
  const int nI = cd.p_nI[p];
  const int s = n - nI;
  const int ld = (s + 1) & ~1;
  const double* rhs = stage + stage_ptr[p] + nI;
  for (int i = threadIdx.x; i < s; i += 256) cond_dsmem[i] = rhs[i];
  __syncthreads();
  const int r1 = min(ld, r0 + COND_SIGMA_ROWS);
  const int share = (((r1 - r0) + 3) / 4 + 15) & ~15;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int a = r0 + wave * share, b = min(r1, a + share);
  if (a < b) big_rows<NT>(cd.sinv + cd.sinv_ptr[p], s, ld, a, b, cond_dsmem, lane, cd.tmp + off + nI);
}

template <bool NT, int COND_WAVES, int RU>
__global__ __launch_bounds__(64 * COND_WAVES) void cond_back_kernel(int64_t p0, int64_t p1, CondDev cd,
                                                                    const int64_t* __restrict__ patch_ptr,
                                                                    const int64_t* __restrict__ stage_ptr,
                                                                    double* __restrict__ stage, int ordered) {
  extern __shared__ double cond_dsmem[];
  constexpr int NT_ = 64 * COND_WAVES;
  if (p0 + blockIdx.x >= p1) return;
  const int64_t p = ordered ? cd.order[blockIdx.x] : p0 + blockIdx.x;
  const int64_t off = patch_ptr[p];
  const int n = (int)(patch_ptr[p + 1] - off);
  const int nI = cd.p_nI[p];
  const int s = n - nI;
  const int uo = (int)(cd.uptr[p + 1] - cd.uptr[p]);
  const int64_t g0 = cd.gptr[p];
  double* ys = cond_dsmem;          // s
  double* yg = ys + s;              // uo: y_S[S_g], group after group (the layout of the u buffer)
  const double* tmp = cd.tmp + off;
  for (int i = threadIdx.x; i < s; i += NT_) ys[i] = tmp[nI + i];
  // the positions S_g of all groups (adjacent in sidx), 4 loads in flight per thread
  const int32_t* si = cd.sidx + (uo > 0 ? cd.g_sidx[g0] : 0);
  int32_t sq[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) sq[u] = (threadIdx.x + u * NT_ < uo) ? si[threadIdx.x + u * NT_] : 0;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (threadIdx.x + u * NT_ < uo) yg[threadIdx.x + u * NT_] = ys[sq[u]];
  for (int q = threadIdx.x + 4 * NT_; q < uo; q += NT_) yg[q] = ys[si[q]];
  __syncthreads();
  double* out = stage + stage_ptr[p];
  // y_g = t_g - W_g y_S[S_g], a lane per pair of interior entries
  const int64_t xp0 = cd.xp_ptr[p];
  const int nxp = (int)(cd.xp_ptr[p + 1] - xp0);
  for (int q0 = 0; q0 < nxp; q0 += NT_) {
    const int q = q0 + threadIdx.x;
    const bool act = q < nxp;
    const int32_t g = act ? cd.xp_grp[xp0 + q] : (int32_t)g0;
    const int m = cd.g_m[g], sc = cd.g_sc[g], o = cd.g_off[g], i = 2 * (q - cd.g_xp[g]);
    const int ldm = cond_ldim(m);
    const double tg0 = act ? tmp[o + i] : 0.0, tg1 = (act && i + 1 < m) ? tmp[o + i + 1] : 0.0;
    const int32_t sl0 = act ? cd.slot[off + o + i] : 0, sl1 = (act && i + 1 < m) ? cd.slot[off + o + i + 1] : 0;
    double y0 = 0.0, y1 = 0.0;
    cond_row2_dot<NT, RU>(cd.mat + cd.g_mat[g] + (int64_t)ldm * m + (int64_t)cond_ldim(sc) * m + i, ldm, act ? sc : 0,
                          yg + cd.g_uoff[g], y0, y1);
    if (act) {
      out[sl0] = tg0 - y0;
      if (i + 1 < m) out[sl1] = tg1 - y1;
    }
  }
  for (int i = threadIdx.x; i < s; i += NT_) out[cd.slot[off + nI + i]] = ys[i];
}

// ---- the group products by CHUNKS of groups (the default since the end of round 3; ALFI_COND_SPLIT=2 keeps the per-patch
// front / back kernels above).  With a workgroup per patch the front and back launches on config 5's finest level spend their
// second half on the ~1000 small stars of the level -- short chains of dependent phases, 3 workgroups per CU because the LDS
// is sized for the largest star -- at a third of the bandwidth of the first half.  Here the unit of work is a run of
// consecutive groups of one patch holding at most 256 row pairs of X / W and of B (8 macro-cell groups of a [P3]^3 star):
// every workgroup is 256 threads with a few KB of LDS, a lane owns at most one row pair per phase, big and small stars
// make the same kind of workgroup.  What needed the whole patch moves: the u buffer goes to global memory (cd.ubuf, in the
// row-sorted order) and the Schur right-hand side is formed by the sigma workgroups themselves (each of the <= 3 per patch
// forms all of it, in the same fixed order: bitwise the same), which also stage the skeleton part of the result.
// the first COND_U columns of a row pair, requested early (before a barrier the loads do not depend on)
template <bool NT>
__device__ __forceinline__ void cond_row2_prefetch(const double* __restrict__ M, int ld, int kn, big_d2 (&a)[COND_U]) {
#pragma unroll
  for (int u = 0; u < COND_U; ++u) {
    const big_d2* q = reinterpret_cast<const big_d2*>(M + (int64_t)u * ld);
    const big_d2 z = {0.0, 0.0};
    a[u] = (u < kn) ? (NT ? __builtin_nontemporal_load(q) : *q) : z;
  }
}
// ... and the product continued from them
template <bool NT>
__device__ __forceinline__ void cond_row2_finish(const double* __restrict__ M, int ld, int kn, const double* __restrict__ v,
                                                 const big_d2 (&a)[COND_U], double& acc0, double& acc1) {
#pragma unroll
  for (int u = 0; u < COND_U; ++u) {
    const double vk = (u < kn) ? v[u] : 0.0;
    acc0 = __builtin_fma(a[u].x, vk, acc0);
    acc1 = __builtin_fma(a[u].y, vk, acc1);
  }
  const int rest = kn > COND_U ? kn - COND_U : 0;
  cond_row2_dot<NT, COND_U>(M + (int64_t)COND_U * ld, ld, rest, v + COND_U, acc0, acc1);
}

// a lane's row pair of X / W (q: its number in the level's xp_grp order), decoded from the group arrays
__device__ __forceinline__ CondXPair cond_xpair(const CondDev& cd, const CondChunk& c, int32_t q, bool act) {
  CondXPair d = {0, 0, 2, 0, 0, 0, 0, 0, -1, -1};
  if (!act) return d;      // a lane without a pair: nothing is read through this descriptor
  const int32_t g = cd.xp_grp[q];
  const int m = cd.g_m[g], sc = cd.g_sc[g], o = cd.g_off[g];
  const int i = 2 * (q - c.xp0 - cd.g_xp[g]);
  const int64_t mat = cd.g_mat[g];
  d.ld = cond_ldim(m);
  d.m = m;
  d.sc = sc;
  d.o = o;
  d.uo = cd.g_uoff[g];
  d.i = i;
  d.xoff = mat + i;
  d.woff = mat + (int64_t)d.ld * m + (int64_t)cond_ldim(sc) * m + i;
  d.sl0 = cd.slot[c.off + o + i];
  d.sl1 = i + 1 < m ? cd.slot[c.off + o + i + 1] : -1;
  return d;
}
// ... and of B
__device__ __forceinline__ CondBPair cond_bpair(const CondDev& cd, const CondChunk& c, int32_t q, bool act) {
  CondBPair d = {0, 2, 0, 0, -1, -1, 0};
  if (!act) return d;
  const int32_t g = cd.bp_grp[q];
  const int m = cd.g_m[g], sc = cd.g_sc[g];
  const int j = 2 * (q - c.bp0 - cd.g_bp[g]);
  const int64_t ue = c.ubase + cd.g_uoff[g] + j;
  d.boff = cd.g_mat[g] + (int64_t)cond_ldim(m) * m + j;
  d.ld = cond_ldim(sc);
  d.m = m;
  d.o = cd.g_off[g];
  d.d0 = cd.u_dst[ue];
  d.d1 = j + 1 < sc ? cd.u_dst[ue + 1] : -1;
  d.pad = 0;
  return d;
}

template <bool NT>
__global__ __launch_bounds__(256) void cond_gfront_kernel(int64_t k0, CondDev cd, const double* __restrict__ x) {
  extern __shared__ double cond_dsmem[];
  const CondChunk c = cd.gc[k0 + blockIdx.x];
  double* xs = cond_dsmem;                           // c.ne
  double* ts = xs + c.ne;                            // c.ne
  const int tid = threadIdx.x;
  const bool xact = c.xq0 + tid < c.xq1, bact = c.bq0 + tid < c.bq1;
  // the lane's row pairs of the two phases, decoded while the gather of x is under way
  const CondXPair xd = cond_xpair(cd, c, c.xq0 + tid, xact);
  const CondBPair bd = cond_bpair(cd, c, c.bq0 + tid, bact);
  for (int i = tid; i < c.ne; i += 256) xs[i] = x[cd.dofs[c.off + c.e0 + i]];
  // the first columns of both products do not depend on anything computed here
  big_d2 xa[COND_U], ba[COND_U];
  cond_row2_prefetch<NT>(cd.mat + xd.xoff, xd.ld, xact ? xd.m : 0, xa);
  cond_row2_prefetch<NT>(cd.mat + bd.boff, bd.ld, bact ? bd.m : 0, ba);
  __syncthreads();
  double* tmp = cd.tmp + c.off;
  {
    double t0 = 0.0, t1 = 0.0;
    cond_row2_finish<NT>(cd.mat + xd.xoff, xd.ld, xact ? xd.m : 0, xs + (xd.o - c.e0), xa, t0, t1);
    if (xact) {
      const int e = xd.o + xd.i;
      ts[e - c.e0] = t0;
      tmp[e] = t0;
      if (xd.sl1 >= 0) {
        ts[e - c.e0 + 1] = t1;
        tmp[e + 1] = t1;
      }
    }
  }
  __syncthreads();
  {
    double v0 = 0.0, v1 = 0.0;
    cond_row2_finish<NT>(cd.mat + bd.boff, bd.ld, bact ? bd.m : 0, ts + (bd.o - c.e0), ba, v0, v1);
    if (bact) {
      cd.ubuf[c.ubase + bd.d0] = v0;
      if (bd.d1 >= 0) cd.ubuf[c.ubase + bd.d1] = v1;
    }
  }
}

// y_S = inv(Sigma) rhs for a chunk of <= 256 rows, the right-hand side formed here: rhs_i = x_i - sum of the row's
// contributions in cd.ubuf (adjacent, ascending: the order of the other forms of the apply); the chunk's rows of y_S go to
// cd.tmp AND to their staging slots.
template <bool NT>
__global__ __launch_bounds__(256) void cond_gsigma_kernel(int64_t c0, CondDev cd, const int64_t* __restrict__ patch_ptr,
                                                          const int64_t* __restrict__ stage_ptr, const double* __restrict__ x,
                                                          double* __restrict__ stage) {
  extern __shared__ double cond_dsmem[];
  const int64_t c = c0 + blockIdx.x;
  const int64_t p = cd.ch_patch[c];
  const int r0 = cd.ch_row[c];
  const int64_t off = patch_ptr[p];
  const int n = (int)(patch_ptr[p + 1] - off);
  const int nI = cd.p_nI[p];
  const int s = n - nI;
  const int ld = (s + 1) & ~1;
  const int64_t srow0 = cd.sptr[p];
  const int32_t qb = cd.s_uptr[srow0];
  const double* ub = cd.ubuf + cd.uptr[p];
  double* rhs = cond_dsmem;                          // s
  double* ys = rhs + ld;                             // the chunk's rows (<= 256)
  for (int i = threadIdx.x; i < s; i += 256) {
    double acc = x[cd.dofs[off + nI + i]];
    const int32_t qe = cd.s_uptr[srow0 + i + 1] - qb;
    for (int32_t q = cd.s_uptr[srow0 + i] - qb; q < qe; ++q) acc -= ub[q];
    rhs[i] = acc;
  }
  __syncthreads();
  const int r1 = min(ld, r0 + COND_SIGMA_ROWS);
  const int share = (((r1 - r0) + 3) / 4 + 15) & ~15;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int a = r0 + wave * share, b = min(r1, a + share);
  // (big_rows indexes its output by the row number and skips the pad row r = s: the chunk's slice, shifted)
  if (a < b) big_rows<NT>(cd.sinv + cd.sinv_ptr[p], s, ld, a, b, rhs, lane, ys - r0);
  __syncthreads();
  double* out = stage + stage_ptr[p];
  double* tmp = cd.tmp + off + nI;
  for (int i = r0 + threadIdx.x; i < min(r1, s); i += 256) {
    const double v = ys[i - r0];
    tmp[i] = v;
    out[cd.slot[off + nI + i]] = v;
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void cond_gback_kernel(int64_t k0, CondDev cd, double* __restrict__ stage) {
  extern __shared__ double cond_dsmem[];
  const CondChunk c = cd.gc[k0 + blockIdx.x];
  const int tid = threadIdx.x;
  const bool act = c.xq0 + tid < c.xq1;
  const CondXPair xd = cond_xpair(cd, c, c.xq0 + tid, act);
  const double* tmp = cd.tmp + c.off;
  // y_S[S_g] of the chunk's groups (adjacent in sidx and in the u layout)
  const int32_t* si = cd.sidx + c.sidx0;
  double* yg = cond_dsmem;                           // c.nu
  for (int q = tid; q < c.nu; q += 256) yg[q] = tmp[c.nI + si[q]];
  const int e = xd.o + xd.i;
  const double tg0 = act ? tmp[e] : 0.0, tg1 = (act && xd.sl1 >= 0) ? tmp[e + 1] : 0.0;
  big_d2 wa[COND_U];
  cond_row2_prefetch<NT>(cd.mat + xd.woff, xd.ld, act ? xd.sc : 0, wa);
  __syncthreads();
  double y0 = 0.0, y1 = 0.0;
  cond_row2_finish<NT>(cd.mat + xd.woff, xd.ld, act ? xd.sc : 0, yg + (xd.uo - c.u0), wa, y0, y1);
  if (act) {
    double* out = stage + c.stage_off;
    out[xd.sl0] = tg0 - y0;
    if (xd.sl1 >= 0) out[xd.sl1] = tg1 - y1;
  }
}

// out (n x n, row-major) = the leading n x n part of the padded N x N scratch
__global__ void dense_unpad_kernel(int64_t n, int64_t N, const double* __restrict__ S, double* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n * n; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = S[(e / n) * N + e % n];
}

// the whole operator of a (coarse) level as ONE dense N x N matrix (N = n rounded up to 64, identity padding): zero / pad,
// then scatter the bs x bs blocks
__global__ void coarse_pad_kernel(int64_t n, int64_t N, double* __restrict__ S) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < N * N; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / N, c = e % N;
    S[e] = (r == c && r >= n) ? 1.0 : 0.0;
  }
}
template <int BS>
__global__ void coarse_scatter_kernel(int64_t nbrows, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                      const double* __restrict__ vals, int flat, int64_t N, double* __restrict__ S) {
  const int64_t total = (int64_t)rowptr[nbrows] * BS * BS;
  // block row of entry k by binary search in rowptr (setup only)
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = e / (BS * BS);
    const int rc = (int)(e % (BS * BS));
    int64_t a = 0, b = nbrows;
    while (b - a > 1) {
      const int64_t mid = (a + b) >> 1;
      if (rowptr[mid] <= k) a = mid; else b = mid;
    }
    const int64_t col = colidx[k] & 0x7fffffff;
    S[(a * BS + rc / BS) * N + col * BS + rc % BS] = vals[bsr_val_index(flat, k, rc, BS * BS)];
  }
}

// Where the matrices of a batch come from: the level's operator (patch smoother), the dense blocks of a transfer, or the
// whole operator of a level (C: the coarse grid, one matrix).
struct MfFill;     // multifrontal coarse factorisation (mf_coarse.h): the fronts of one tree height
void mf_fill_dispatch(const MfFill* f, alfi_ctx* ctx, int64_t p0, int64_t nb, const int64_t* d_scr_ptr, double* dst);
void mf_store_dispatch(const MfFill* f, alfi_ctx* ctx, int64_t p0, int64_t nb, const int64_t* d_scr_ptr, const double* res);

struct BigSource {
  const MfFill* M = nullptr;
  alfi_level* L = nullptr;
  alfi_transfer* T = nullptr;
  alfi_level* C = nullptr;
  alfi_level* K = nullptr;        // condensed patches: the Schur complements of the level's patches
  bool pivot_groups = false;      // K: the group inverses by LU with partial pivoting (repair of flagged patches)
  void fill(alfi_ctx* ctx, int64_t p0, int64_t nb, const int64_t* d_scr_ptr, double* dst) const {
    dim3 block(256);
    if (M) {
      mf_fill_dispatch(M, ctx, p0, nb, d_scr_ptr, dst);
    } else if (K) {
      const size_t lds = (size_t)(3 * K->max_np) * sizeof(int32_t);
      if (K->bs == 2)
        hipLaunchKernelGGL(cond_fill_kernel<2>, dim3((unsigned)nb), block, lds, ctx->stream, p0, K->cd, K->patch_ptr,
                           K->patch_dofs, K->A.rowptr, K->A.colidx, K->A.vals, K->A.flat, d_scr_ptr, dst, K->status);
      else
        hipLaunchKernelGGL(cond_fill_kernel<3>, dim3((unsigned)nb), block, lds, ctx->stream, p0, K->cd, K->patch_ptr,
                           K->patch_dofs, K->A.rowptr, K->A.colidx, K->A.vals, K->A.flat, d_scr_ptr, dst, K->status);
      // groups of the patches [p0, p0 + nb)
      const int64_t ga = K->h_cond_gptr[p0], gb = K->h_cond_gptr[p0 + nb];
      if (gb > ga && pivot_groups) {
        const size_t glds = (size_t)(COND_GMAX * COND_GMAX + 64 * COND_GMAX) * sizeof(double) + COND_GMAX * sizeof(int);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cond_group_pivot_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds);
        hipLaunchKernelGGL(cond_group_pivot_kernel, dim3((unsigned)(gb - ga)), dim3(64), glds, ctx->stream, ga, K->cd, K->status);
      } else if (gb > ga)
        hipLaunchKernelGGL(cond_group_kernel, dim3((unsigned)((gb - ga + 3) / 4)), block, 0, ctx->stream, ga, gb, K->cd,
                           K->status);
      hipLaunchKernelGGL(cond_schur_kernel, dim3((unsigned)nb), block, 0, ctx->stream, p0, K->cd, K->patch_ptr, d_scr_ptr, dst);
    } else if (C) {
      const int64_t n = C->n, N = (n + BIG_NB - 1) / BIG_NB * BIG_NB;
      hipLaunchKernelGGL(coarse_pad_kernel, dim3(8192), block, 0, ctx->stream, n, N, dst);
      if (C->bs == 2)
        hipLaunchKernelGGL(coarse_scatter_kernel<2>, dim3(4096), block, 0, ctx->stream, C->A.nbrows, C->A.rowptr, C->A.colidx,
                           C->A.vals, C->A.flat, N, dst);
      else
        hipLaunchKernelGGL(coarse_scatter_kernel<3>, dim3(4096), block, 0, ctx->stream, C->A.nbrows, C->A.rowptr, C->A.colidx,
                           C->A.vals, C->A.flat, N, dst);
    } else if (T) {
      hipLaunchKernelGGL(big_dense_fill_kernel, dim3(64, (unsigned)nb), block, 0, ctx->stream, p0, T->m, T->KII, T->DII,
                         T->nu, T->gamma, d_scr_ptr, dst);
    } else if (L->bs == 2) {
      hipLaunchKernelGGL(big_gather_kernel<2>, dim3(16, (unsigned)nb), block, 0, ctx->stream, p0, L->A.rowptr, L->A.colidx,
                         L->A.vals, L->A.flat, L->patch_ptr, L->patch_dofs, d_scr_ptr, dst);
    } else {
      hipLaunchKernelGGL(big_gather_kernel<3>, dim3(16, (unsigned)nb), block, 0, ctx->stream, p0, L->A.rowptr, L->A.colidx,
                         L->A.vals, L->A.flat, L->patch_ptr, L->patch_dofs, d_scr_ptr, dst);
    }
  }
};

// fill + blocked inversion of npatch matrices, in batches bounded by the scratch budget
// dense_out != nullptr (one matrix): the inverse is delivered row-major n x n there instead of the row-piece layout
static int big_factor_core(alfi_ctx* ctx, const BigSource& src, int64_t npatch, const int64_t* h_patch_ptr,
                           const int64_t* d_patch_ptr, const int64_t* d_inv_ptr, double* inv, int* status,
                           double* dense_out = nullptr) {
  if (npatch == 0) return 0;
  const int64_t budget = (int64_t)6 << 30;          // bytes of scratch (matrices + panels) per batch
  const char* env = getenv("ALFI_BIG_SCRATCH_MB");
  const int64_t limit = env ? (int64_t)atoll(env) << 20 : budget;
  constexpr bool polish = true;   // one Newton-Schulz step after the blocked elimination (explicit pivot-block inverses lose
                                  // cond * eps: 2e-5 against LAPACK without it, 1e-8 with)
  const int64_t nscr = polish ? 3 : 1;            // X | a second copy of A_p | I - A X
  // batches [p0, p1) within the scratch budget
  struct Batch {
    int64_t p0, p1, sdoubles, pdoubles;
    int Nmax;
    std::vector<int64_t> scr_ptr, pan_ptr;
  };
  std::vector<Batch> batches;
  int64_t smax = 0, pmax = 0, nbmax = 0;
  for (int64_t p0 = 0; p0 < npatch;) {
    Batch B;
    B.p0 = p0;
    B.sdoubles = B.pdoubles = 0;
    B.Nmax = 0;
    int64_t p1 = p0;
    while (p1 < npatch) {
      const int n = (int)(h_patch_ptr[p1 + 1] - h_patch_ptr[p1]);
      const int N = (n + BIG_NB - 1) / BIG_NB * BIG_NB;
      const int64_t need = ((int64_t)nscr * N * N + 2 * (int64_t)N * BIG_NB) * 8;
      if (p1 > p0 && (nscr * B.sdoubles + 2 * B.pdoubles) * 8 + need > limit) break;
      B.scr_ptr.push_back(B.sdoubles);
      B.pan_ptr.push_back(B.pdoubles);
      B.sdoubles += (int64_t)N * N;
      B.pdoubles += (int64_t)N * BIG_NB;
      if (N > B.Nmax) B.Nmax = N;
      ++p1;
    }
    B.p1 = p1;
    smax = std::max(smax, B.sdoubles);
    pmax = std::max(pmax, B.pdoubles);
    nbmax = std::max(nbmax, p1 - p0);
    p0 = p1;
    batches.push_back(std::move(B));
  }
  // one arena for all batches, kept by the context (patches are re-factored every Newton step; hipMalloc / hipFree of
  // gigabytes per batch cost more wall time than the kernels)
  auto up = [](int64_t b) { return (size_t)((b + 255) & ~(int64_t)255); };
  const size_t need = nscr * up(smax * 8) + 2 * up(pmax * 8) + up(nbmax * BIG_NB * BIG_NB * 8) + 2 * up(nbmax * 8);
  if (ctx->big_arena_bytes < need) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->big_arena);
    ctx->big_arena = nullptr;
    ctx->big_arena_bytes = 0;
    hipError_t e = hipMalloc(&ctx->big_arena, need);
    if (e != hipSuccess)
      return alfi_set_error(ctx, ALFI_E_HIP, "large-block scratch of %zu bytes: %s", need, hipGetErrorString(e));
    ctx->big_arena_bytes = need;
  }
  char* cur = static_cast<char*>(ctx->big_arena);
  auto carve = [&](size_t bytes) { char* q = cur; cur += bytes; return q; };
  double* scr = reinterpret_cast<double*>(carve(up(smax * 8)));
  double* scrA = polish ? reinterpret_cast<double*>(carve(up(smax * 8))) : nullptr;
  double* scrR = polish ? reinterpret_cast<double*>(carve(up(smax * 8))) : nullptr;
  double* panF = reinterpret_cast<double*>(carve(up(pmax * 8)));
  double* panR = reinterpret_cast<double*>(carve(up(pmax * 8)));
  double* dinv = reinterpret_cast<double*>(carve(up(nbmax * BIG_NB * BIG_NB * 8)));
  int64_t* d_scr_ptr = reinterpret_cast<int64_t*>(carve(up(nbmax * 8)));
  int64_t* d_pan_ptr = reinterpret_cast<int64_t*>(carve(up(nbmax * 8)));
  hipError_t e = hipSuccess;
  const dim3 block(256);
  for (const Batch& B : batches) {
    const int64_t p0 = B.p0, nb = B.p1 - B.p0;
    // (pageable host source: the copy has left the host buffer when the call returns; the stream orders it after the
    // previous batch's kernels)
    e = hipMemcpyAsync(d_scr_ptr, B.scr_ptr.data(), (size_t)nb * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pan_ptr, B.pan_ptr.data(), (size_t)nb * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) break;
    src.fill(ctx, p0, nb, d_scr_ptr, scr);
    if (polish)   // the elimination overwrites its copy of A_p; the polish needs A_p again
      (void)hipMemcpyAsync(scrA, scr, (size_t)B.sdoubles * 8, hipMemcpyDeviceToDevice, ctx->stream);
    const int tiles = B.Nmax / BIG_NB;
    for (int k0 = 0; k0 < B.Nmax; k0 += BIG_NB) {
      hipLaunchKernelGGL(big_pivot_kernel, dim3((unsigned)nb), block, 0, ctx->stream, k0, d_patch_ptr, p0, d_scr_ptr, scr,
                         dinv, status);
      hipLaunchKernelGGL(big_panel_kernel, dim3((unsigned)(2 * tiles), (unsigned)nb), block, 0, ctx->stream, k0,
                         d_patch_ptr, p0, d_scr_ptr, scr, d_pan_ptr, panF, panR, dinv, tiles);
      hipLaunchKernelGGL(big_update_kernel, dim3((unsigned)(tiles * tiles), (unsigned)nb), block, 0, ctx->stream, k0,
                         d_patch_ptr, p0, d_scr_ptr, scr, d_pan_ptr, panF, panR, tiles);
    }
    const double* result = scr;
    if (polish) {
      // R = I - A X, X <- X + X R (into the buffer that held the copy of A_p)
      hipLaunchKernelGGL(big_gemm_kernel, dim3((unsigned)(tiles * tiles), (unsigned)nb), block, 0, ctx->stream, 0,
                         d_patch_ptr, p0, d_scr_ptr, scrA, scr, scrR, tiles);
      hipLaunchKernelGGL(big_gemm_kernel, dim3((unsigned)(tiles * tiles), (unsigned)nb), block, 0, ctx->stream, 1,
                         d_patch_ptr, p0, d_scr_ptr, scr, scrR, scrA, tiles);
      result = scrA;
      constexpr bool mf_ns2 = true;
      if (dense_out || (src.M && mf_ns2)) {
        // the coarse operator (one matrix, hundreds of pivot blocks; or the fronts of its sparse factorisation, whose errors
        // travel up the elimination tree through the Schur complements): a second Newton-Schulz step.  The matrix is
        // filled again (into the buffer that held the first iterate's predecessor), R = I - A X, X <- X + X R.
        src.fill(ctx, p0, nb, d_scr_ptr, scr);
        hipLaunchKernelGGL(big_gemm_kernel, dim3((unsigned)(tiles * tiles), (unsigned)nb), block, 0, ctx->stream, 0,
                           d_patch_ptr, p0, d_scr_ptr, scr, scrA, scrR, tiles);
        hipLaunchKernelGGL(big_gemm_kernel, dim3((unsigned)(tiles * tiles), (unsigned)nb), block, 0, ctx->stream, 1,
                           d_patch_ptr, p0, d_scr_ptr, scrA, scrR, scr, tiles);
        result = scr;
      }
    }
    if (src.M) {
      mf_store_dispatch(src.M, ctx, p0, nb, d_scr_ptr, result);
    } else if (dense_out) {
      // (a kernel, not hipMemcpy2DAsync: the runtime splits a pitched device copy into one copy per row -- 17 320 copy
      // launches, 0.45 s, for the 23 355 rows of config 4's coarse inverse)
      const int64_t n = h_patch_ptr[1] - h_patch_ptr[0];
      hipLaunchKernelGGL(dense_unpad_kernel, dim3(8192), block, 0, ctx->stream, n, (int64_t)B.Nmax, result, dense_out);
    } else {
      hipLaunchKernelGGL(big_store_kernel, dim3(64, (unsigned)nb), block, 0, ctx->stream, p0, d_patch_ptr, d_inv_ptr,
                         d_scr_ptr, result, inv);
    }
    e = hipGetLastError();
    if (e != hipSuccess) break;
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return alfi_set_error(ctx, ALFI_E_HIP, "large-block factorisation failed: %s", hipGetErrorString(e));
  return 0;
}

int launch_big_factor(alfi_level* L) {
  BigSource src;
  src.L = L;
  return big_factor_core(L->ctx, src, L->npatch, L->h_patch_ptr.data(), L->patch_ptr, L->inv_ptr, L->inv, L->status);
}

// interior blocks of a transfer with 160 < m <= 4096 (macro-cell blocks of the 3-D Scott-Vogelius transfer, m = 390 for P3)
int launch_big_factor_transfer(alfi_transfer* T) {
  BigSource src;
  src.T = T;
  std::vector<int64_t> hptr(T->nblk + 1);
  for (int64_t b = 0; b <= T->nblk; ++b) hptr[b] = b * T->m;
  return big_factor_core(T->ctx, src, T->nblk, hptr.data(), T->pm_ptr, T->pm_inv_ptr, T->binv, T->status);
}

// Dense inverse of a whole level operator (the coarse grid: AssembledPC + LU in the reference, alfi/solver.py:369-378) with
// the same blocked Gauss-Jordan on the FP64 matrix cores + Newton-Schulz polish as the macro-star patches: one N x N
// matrix, N/64 pivot steps, no library GEMM and no host LAPACK.  out: n x n doubles, row-major, on the device.
int launch_coarse_factor(alfi_level* L, double* out) {
  alfi_ctx* ctx = L->ctx;
  BigSource src;
  src.C = L;
  const int64_t hptr[2] = {0, L->n};
  int64_t* dptr = nullptr;
  ALFI_HIP_CHECK(ctx, hipMalloc((void**)&dptr, sizeof(hptr)));
  hipError_t e = hipMemcpy(dptr, hptr, sizeof(hptr), hipMemcpyHostToDevice);
  int rc = e == hipSuccess ? big_factor_core(ctx, src, 1, hptr, dptr, nullptr, nullptr, L->status, out)
                           : alfi_set_error(ctx, ALFI_E_HIP, "hipMemcpy: %s", hipGetErrorString(e));
  (void)hipFree(dptr);
  // the 3 N^2 scratch arena of a large coarse grid is not needed again (patch re-factorisations are far smaller)
  if (rc == 0 && ctx->big_arena_bytes > ((size_t)1 << 30)) {
    (void)hipFree(ctx->big_arena);
    ctx->big_arena = nullptr;
    ctx->big_arena_bytes = 0;
  }
  return rc;
}

// condensed patches: group inverses + Schur complements, the latter inverted by the blocked Gauss-Jordan above
int launch_cond_factor(alfi_level* L) {
  BigSource src;
  src.K = L;
  return big_factor_core(L->ctx, src, L->npatch, L->h_sptr.data(), L->cd.sptr, L->cd.sinv_ptr, L->cd.sinv, L->status);
}

// the group matrices of ONE condensed patch again -- X_g, W_g by LU with partial pivoting -- and its Schur complement into scr
// (N x N row-major, N = s rounded up to BIG_NB, identity padding);
// d_zero: a device int64 holding 0 (the patch's offset in scr).  The repair path of kernels_check.hip.
int launch_cond_schur_one(alfi_level* L, int64_t p, const int64_t* d_zero, double* scr) {
  BigSource src;
  src.K = L;
  src.pivot_groups = true;       // the patch failed the probe: its group inverses by pivoted LU as well
  src.fill(L->ctx, p, 1, d_zero, scr);
  ALFI_HIP_CHECK(L->ctx, hipGetLastError());
  return 0;
}

// One condensed apply = three launches (front / sigma / back).  Two forms of the group products:
//   * launches of fewer than 1024 patches (the lower levels: config 5's level 1 has 303 stars; ranges of an overlapped exchange):
//     chunks of <= 8 consecutive groups per workgroup, one descriptor per workgroup, a lane's row pair decoded in the kernel, the
//     Schur right-hand side formed by the sigma workgroups (cond_gfront / cond_gsigma / cond_gback): front + back 48 -> 39 us
//     against a workgroup per patch there;
//   * larger launches (the finest levels): a workgroup of 8 waves per patch (cond_front / cond_sigma / cond_back).  The chunked
//     form on config 5's finest level, same box: 25.1 against 24.6 ms per cycle (round 4, with the decoded descriptors; round
//     3 with a descriptor per lane: 338 against 326 us).
// The sigma launch takes chunks of <= 64 rows (COND_SIGMA_ROWS; 256 in round 3: config 5 24.8 -> 24.1 ms per cycle, the apply
// from 0.70 to 0.73 of the HBM roofline, profiles/r04_ab_cond_cfg5.txt).
// Measured and dropped (rounds 2-3, profiles/r03_cond_apply_ab_cfg5.txt): the whole apply as ONE launch per patch (941 against
// 860 us on config 5's finest level), 4 or 16 waves per patch on large launches, 16 columns in flight per lane, nontemporal
// loads for the group matrices (a column of 45 doubles shares its first and last line with its neighbours: 218 against 208 us).
int launch_cond_apply_range(alfi_level* L, int64_t p0, int64_t p1, const double* x) {
  alfi_ctx* ctx = L->ctx;
  if (p1 <= p0) return 0;
  if (!L->cd.tmp) return alfi_set_error(ctx, ALFI_E_STATE, "condensed apply without its scratch (alfi_patches_set_groups)");
  dim3 grid((unsigned)(p1 - p0));
  // full-range launches walk the patches largest first (workgroups are dispatched in index order: with ~7 patches per CU at
  // config 5's size the big patches must not come last); range launches (overlapped exchanges) keep the natural order
  const int ordered = p0 == 0 && p1 == L->npatch && L->cd.order ? 1 : 0;
  const int64_t c0 = L->h_cond_chptr[p0], c1 = L->h_cond_chptr[p1];
  if (p1 - p0 < 1024) {
    const int64_t k0 = L->h_cond_gcptr[p0], k1 = L->h_cond_gcptr[p1];
    const size_t lds_f = (size_t)L->cond_lds_gfront, lds_b = (size_t)L->cond_lds_gback;
    const size_t lds_s = (size_t)(L->cond_max_s + 2 + COND_SIGMA_ROWS) * sizeof(double);
    if (k1 > k0)
      hipLaunchKernelGGL((cond_gfront_kernel<false>), dim3((unsigned)(k1 - k0)), dim3(256), lds_f, ctx->stream, k0, L->cd, x);
    if (c1 > c0)
      hipLaunchKernelGGL((cond_gsigma_kernel<true>), dim3((unsigned)(c1 - c0)), dim3(256), lds_s, ctx->stream, c0, L->cd,
                         L->patch_ptr, L->stage_ptr, x, L->stage);
    if (k1 > k0)
      hipLaunchKernelGGL((cond_gback_kernel<false>), dim3((unsigned)(k1 - k0)), dim3(256), lds_b, ctx->stream, k0, L->cd,
                         L->stage);
    ALFI_HIP_CHECK(ctx, hipGetLastError());
    return 0;
  }
  const size_t lds_f = (size_t)L->cond_lds_front, lds_s = (size_t)(L->cond_max_s + 2) * sizeof(double);
  const size_t lds_b = (size_t)L->cond_lds_back;
#define ALFI_COND_LAUNCH3(WV)                                                                                             \
  do {                                                                                                                    \
    if (lds_f > 64 * 1024)                                                                                                \
      ALFI_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&cond_front_kernel<false, WV, 8>),            \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));                   \
    hipLaunchKernelGGL((cond_front_kernel<false, WV, 8>), dim3(grid.x), dim3(64 * WV), lds_f, ctx->stream, p0, p1, L->cd, \
                       L->patch_ptr, L->stage_ptr, x, L->stage, ordered);                                                 \
    if (c1 > c0)                                                                                                          \
      hipLaunchKernelGGL((cond_sigma_kernel<true>), dim3((unsigned)(c1 - c0)), dim3(256), lds_s, ctx->stream, c0, L->cd,  \
                         L->patch_ptr, L->stage_ptr, L->stage);                                                           \
    if (lds_b > 64 * 1024)                                                                                                \
      ALFI_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&cond_back_kernel<false, WV, 8>),             \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));                   \
    hipLaunchKernelGGL((cond_back_kernel<false, WV, 8>), dim3(grid.x), dim3(64 * WV), lds_b, ctx->stream, p0, p1, L->cd,  \
                       L->patch_ptr, L->stage_ptr, L->stage, ordered);                                                    \
  } while (0)
  ALFI_COND_LAUNCH3(8);
#undef ALFI_COND_LAUNCH3
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

#include "mf_coarse.h"
