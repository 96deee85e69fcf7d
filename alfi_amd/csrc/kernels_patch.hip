// Patch smoother kernels for gfx950 (wave64): PCPATCH setup (gather + dense inversion) and additive apply.
//
// Storage of the inverses (owned layout, chosen for the apply kernel; patch_inv_index in common.h): patch p holds an
// n_p x n_p inverse at inv + inv_ptr[p] (a multiple of 16 doubles = 128 B) as row pieces of 128, 64, ..., 2 rows, each
// piece stored [column][rows of the piece].  Lane l of a wave reads a row pair of one column as one aligned 16-byte load,
// a wave instruction covers 1 KiB of consecutive bytes, and consecutive instructions continue where the previous one
// stopped: every 128-B line of the inverse is fetched exactly once and fully used.  (Round 1 stored plain column-major
// with ld = 154: the 26-row tail of each column was read in a second pass through lines shared with the next column,
// FETCH_SIZE showed 1.19 x the algorithmic bytes -- profiles/pmc_patch_apply_cfg4.json.)
//
// Roofline: apply is a GEMV streaming every inverse once (0.25 flop/B) -> HBM-bound; algorithmic bytes per patch
// 8 n_p^2 + 28 n_p.  Inversion is 2 n_p^3 flops per patch in FP64; gfx950's FP64 MFMA rate
// equals its FP64 vector rate (78.6 TF), so the inversion runs as a register-tiled Gauss-Jordan on the vector ALUs.
#include <algorithm>
#include <cstdlib>
#include "common.h"

// ---------------------------------------------------------------------------------------------------------------------
// 1. gather A_p = A[dofs_p, dofs_p] from the BSR operator into row-major dense storage (ld_p columns per row)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int MAX_NP = 160;
// same-wave LDS hand-off: a wave that writes an LDS array and reads it through other lanes needs no workgroup barrier (the LDS
// serves a wave's instructions in order), only the compiler must keep the order
#define ALFI_WAVE_LDS_ORDER_GATHER()                           \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                           \
  } while (0)

template <int BS>
__global__ __launch_bounds__(256) void patch_gather_dense_kernel(const int32_t* __restrict__ rowptr,
                                                                  const int32_t* __restrict__ colidx,
                                                                  const double* __restrict__ vals,
                                                                  const int64_t* __restrict__ patch_ptr,
                                                                  const int32_t* __restrict__ patch_dofs,
                                                                  const int64_t* __restrict__ inv_ptr,
                                                                  double* __restrict__ inv, int flat) {
  __shared__ int32_t dofs_s[MAX_NP];
  __shared__ double rowbuf[4][MAX_NP];
  const int64_t p = blockIdx.x;
  const int64_t off = patch_ptr[p];
  const int n = (int)(patch_ptr[p + 1] - off);
  const int ld = (n + 1) & ~1;
  double* S = inv + inv_ptr[p];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < n; i += 256) dofs_s[i] = patch_dofs[off + i];
  __syncthreads();
  // every wave takes rows of its own through its own row buffer: no workgroup barrier inside the loop (the LDS serves a wave's
  // accesses in order; until round 5 the four waves met at two barriers per four rows, 76 per patch of 153 dofs)
  for (int r = wave; r < n; r += 4) {
    for (int c = lane; c < ld; c += 64) rowbuf[wave][c] = 0.0;
    ALFI_WAVE_LDS_ORDER_GATHER();
    const int gr = dofs_s[r];
    const int brow = gr / BS, rr = gr % BS;
    const int32_t lo = rowptr[brow], hi = rowptr[brow + 1];
    const int nent = (hi - lo) * BS;
    for (int e = lane; e < nent; e += 64) {
      const int blk = e / BS, cc = e % BS;
      const int gcol = (colidx[lo + blk] & 0x7fffffff) * BS + cc;  // sign bit = row-start mark (flat layout)
      // binary search gcol in dofs_s[0..n)
      int a = 0, b = n;
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (dofs_s[mid] < gcol) a = mid + 1; else b = mid;
      }
      if (a < n && dofs_s[a] == gcol) rowbuf[wave][a] = vals[bsr_val_index(flat, lo + blk, rr * BS + cc, BS * BS)];
    }
    ALFI_WAVE_LDS_ORDER_GATHER();
    for (int c = lane; c < ld; c += 64) S[(int64_t)r * ld + c] = rowbuf[wave][c];
    ALFI_WAVE_LDS_ORDER_GATHER();        // the copy out has read the buffer before the next row clears it
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2a. in-place inversion, big patches (n <= 16 B): one 256-thread workgroup per patch, the matrix lives in registers as
//     a 16 x 16 grid of B x B tiles (thread (ti, tj) owns rows ti*B.., cols tj*B..); Gauss-Jordan without pivoting (the
//     patch operators are principal sub-blocks of an SPD-dominated operator); per step the pivot row and column travel
//     through double-buffered LDS (one barrier per step).  Reads row-major, writes column-major (transposed store), both
//     with leading dimension ld; the result is written in the row-piece layout of patch_inv_index.
// ---------------------------------------------------------------------------------------------------------------------
template <int B>
__global__ __launch_bounds__(256) void patch_invert_big_kernel(const int64_t* __restrict__ patch_ptr,
                                                                const int64_t* __restrict__ inv_ptr,
                                                                double* __restrict__ inv, int* __restrict__ status) {
  constexpr int NMAX = 16 * B;
  __shared__ double rowbuf[2][NMAX];
  __shared__ double colbuf[2][NMAX];
  const int64_t p = blockIdx.x;
  const int n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
  const int ld = (n + 1) & ~1;
  double* S = inv + inv_ptr[p];
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  const int r0 = ti * B, c0 = tj * B;
  double a[B][B];
#pragma unroll
  for (int li = 0; li < B; ++li)
#pragma unroll
    for (int lj = 0; lj < B; ++lj) {
      const int r = r0 + li, c = c0 + lj;
      a[li][lj] = (r < n && c < n) ? S[(int64_t)r * ld + c] : (r == c ? 1.0 : 0.0);
    }
  __syncthreads();  // all loads done before anyone stores (in-place transposition at the end)
  const int nkb = (n + B - 1) / B;
  bool bad = false;
  for (int kb = 0; kb < nkb; ++kb) {
    const bool own_row = (ti == kb), own_col = (tj == kb);
#pragma unroll
    for (int kl = 0; kl < B; ++kl) {
      const int k = kb * B + kl;
      const int buf = k & 1;
      if (own_row) {
#pragma unroll
        for (int lj = 0; lj < B; ++lj) rowbuf[buf][c0 + lj] = a[kl][lj];
      }
      if (own_col) {
#pragma unroll
        for (int li = 0; li < B; ++li) colbuf[buf][r0 + li] = a[li][kl];
      }
      __syncthreads();
      const double piv = rowbuf[buf][k];
      if (piv == 0.0) bad = true;
      const double ip = 1.0 / piv;
      double rr[B], cc[B], rm[B], cm[B];
#pragma unroll
      for (int l = 0; l < B; ++l) {
        rr[l] = rowbuf[buf][c0 + l];
        cc[l] = colbuf[buf][r0 + l] * ip;
        rm[l] = (own_col && l == kl) ? 0.0 : rr[l];
        cm[l] = (own_row && l == kl) ? 0.0 : cc[l];
      }
#pragma unroll
      for (int li = 0; li < B; ++li)
#pragma unroll
        for (int lj = 0; lj < B; ++lj) a[li][lj] = __builtin_fma(-cm[li], rm[lj], a[li][lj]);
      if (own_row) {
#pragma unroll
        for (int lj = 0; lj < B; ++lj) a[kl][lj] = rr[lj] * ip;
      }
      if (own_col) {
#pragma unroll
        for (int li = 0; li < B; ++li) a[li][kl] = -cm[li];
      }
      if (own_row && own_col) a[kl][kl] = ip;
    }
  }
  if (bad && threadIdx.x == 0) atomicExch(status, 1);
#pragma unroll
  for (int li = 0; li < B; ++li)
#pragma unroll
    for (int lj = 0; lj < B; ++lj) {
      const int r = r0 + li, c = c0 + lj;
      // r == n < ld: the padding row; its entries stayed 0 through the elimination (identity padding) and must be stored,
      // the apply kernel reads row pairs
      if (r < ld && c < n) S[patch_inv_index(r, c, n, ld)] = a[li][lj];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2b. in-place inversion, small matrices (n <= N <= 32): N lanes per matrix, lane i holds row i in registers, 64/N
//     matrices per wave, pivot rows broadcast with shuffles.  Output: TILED = the patch layout (patch_inv_index),
//     otherwise column-major with leading dimension ld (the Schoeberl interior blocks, kernels_vec.hip).
// ---------------------------------------------------------------------------------------------------------------------
template <int N, bool TILED>
__global__ __launch_bounds__(256) void invert_small_kernel(int64_t nmat, const int64_t* __restrict__ ptr,
                                                            const int64_t* __restrict__ inv_ptr, int fixed_n,
                                                            int64_t fixed_stride, double* __restrict__ inv,
                                                            int* __restrict__ status) {
  const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) / N;
  const int i = threadIdx.x % N;
  const bool valid = g < nmat;
  int n = 0;
  double* S = inv;
  if (valid) {
    if (ptr) {
      n = (int)(ptr[g + 1] - ptr[g]);
      S = inv + inv_ptr[g];
    } else {
      n = fixed_n;
      S = inv + g * fixed_stride;
    }
  }
  const int ld = (n + 1) & ~1;
  double a[N];
#pragma unroll
  for (int j = 0; j < N; ++j) a[j] = (i < n && j < n) ? S[(int64_t)i * ld + j] : (i == j ? 1.0 : 0.0);
  bool bad = false;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double piv = __shfl(a[k], k, N);
    if (piv == 0.0) bad = true;
    const double ip = 1.0 / piv;
    const double m = (i == k) ? 0.0 : a[k] * ip;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const double rk = __shfl(a[j], k, N);
      if (j == k)
        a[j] = (i == k) ? ip : -m;
      else
        a[j] = (i == k) ? rk * ip : __builtin_fma(-m, rk, a[j]);
    }
  }
  // all lanes of the group have read their rows before the first shuffle round completed -> safe to store transposed
  if (valid) {
    if (bad && i == 0) atomicExch(status, 1);
#pragma unroll
    for (int j = 0; j < N; ++j)
      if (i < ld && j < n) S[TILED ? patch_inv_index(i, j, n, ld) : (int64_t)j * ld + i] = a[j];  // incl. zero pad row
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3. additive apply, stage 1: one wave per patch.  stage[patch_ptr[p] + r] = sum_c inv_p[r][c] * x[dofs_p[c]]
// ---------------------------------------------------------------------------------------------------------------------
// one row piece of 2G rows, stored [column][2G]: G lanes per column, 64/G columns per wave instruction
typedef double alfi_d2 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 load_pair(const double* p) {   // one aligned 16-byte load
  const alfi_d2* q = reinterpret_cast<const alfi_d2*>(p);
  const alfi_d2 v = NT ? __builtin_nontemporal_load(q) : *q;
  return make_double2(v.x, v.y);
}

#ifndef ALFI_APPLY_U
#define ALFI_APPLY_U 8                    // tuning builds: -DALFI_APPLY_U=16
#endif
// U = loads kept in flight per lane.  8 for the additive apply (a CU holds 16-32 waves there: 8 KiB in flight per wave fill the
// memory pipeline); the multiplicative sweep runs ONE wave per patch with ~2 waves per CU, where the wave's own bytes in flight
// bound it (8 KiB / ~2 us of loaded latency = 4 GB/s per wave: the 46 us per wavefront of round 3) -- it asks for 32.
template <int G, bool NT, int U = ALFI_APPLY_U>
__device__ __forceinline__ void apply_piece(const double* __restrict__ T, int n, const double* __restrict__ xs,
                                            int lane, double* __restrict__ out) {
  constexpr int C = 64 / G;  // columns handled per wave instruction
  const int cg = lane / G, l = lane % G;
  const double* base = T + 2 * l;
  double acc0 = 0.0, acc1 = 0.0;
  int j = cg;
  for (; j + (U - 1) * C < n; j += U * C) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = load_pair<NT>(base + (int64_t)(j + u * C) * (2 * G));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double xj = xs[j + u * C];
      acc0 = __builtin_fma(v[u].x, xj, acc0);
      acc1 = __builtin_fma(v[u].y, xj, acc1);
    }
  }
  if (j < n) {       // the last, partial group: its loads are requested together too (same summation order: ascending columns)
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      v[u] = j + u * C < n ? load_pair<NT>(base + (int64_t)(j + u * C) * (2 * G)) : make_double2(0.0, 0.0);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (j + u * C < n) {
        const double xj = xs[j + u * C];
        acc0 = __builtin_fma(v[u].x, xj, acc0);
        acc1 = __builtin_fma(v[u].y, xj, acc1);
      }
  }
  if (C > 1) {
#pragma unroll
    for (int o = G; o < 64; o <<= 1) {
      acc0 += __shfl_xor(acc0, o);
      acc1 += __shfl_xor(acc1, o);
    }
  }
  // the padding row (n odd) of the inverse is zero, so storing both rows is always valid (stage has ld slots)
  if (cg == 0) *reinterpret_cast<double2*>(out + 2 * l) = make_double2(acc0, acc1);
}

// A WORKGROUP of APPLY_W waves per patch: wave w takes the w-th share of the columns of every row piece, the partial results
// are added in the order of the waves (deterministic).  Rounds 1-3 gave a patch to ONE wave, four patches to a workgroup: a
// wave then streams its patch for as long as a CU's share of the bandwidth lets it (config 3: 78 KB at 32 waves per CU =
// ~110 us, config 4 ~240 us), workgroups end when their slowest wave ends, and the launch ends with a ramp-down of that length.
// Cut four times finer (same box, apply + sum, profiles/r04_ab_apply_split.txt): config 4's finest level 5316 -> 5256 us,
// its level 2 636 -> 615 us, config 3's level 3 (4913 patches) 72.8 -> 63.4 us; giving only the LAST patches of a level to
// workgroups (one resident round, 8192 patches) gained less than half of that.
// Waves per patch, same box: config 4 (153 dofs) 2: 4929 / 4: 4912 / 8: 4856 us (level 2: 628 / 624 / 608); config 3 (111 dofs)
// 2: 517 / 4: 487-509 / 8: 516 us -- eight above 128 dofs, four down to 65, one wave below.
#ifndef ALFI_APPLY_W
#define ALFI_APPLY_W 4
#endif
constexpr int APPLY_W = ALFI_APPLY_W;     // waves per patch of 65 .. 128 dofs (twice as many above, one wave below)
template <bool NT, int W>
__global__ __launch_bounds__(64 * W) void patch_apply_kernel(int64_t p0, int64_t p1,
                                                                    const int64_t* __restrict__ patch_ptr,
                                                                    const int32_t* __restrict__ patch_dofs,
                                                                    const int64_t* __restrict__ inv_ptr,
                                                                    const int64_t* __restrict__ stage_ptr,
                                                                    const double* __restrict__ inv,
                                                                    const double* __restrict__ x, double* __restrict__ stage) {
  __shared__ double xs[MAX_NP];
  __shared__ double part[W][MAX_NP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t p = p0 + blockIdx.x;
  if (p >= p1) return;
  const int64_t off = patch_ptr[p];
  const int n = (int)(patch_ptr[p + 1] - off);
  for (int i = threadIdx.x; i < n; i += 64 * W) xs[i] = x[patch_dofs[off + i]];
  __syncthreads();
  const int ld = (n + 1) & ~1;
  const double* T = inv + inv_ptr[p];
  const int ca = (int)(((int64_t)wave * n) / W), cn = (int)(((int64_t)(wave + 1) * n) / W) - ca;
  double* out = part[wave];
  int row0 = 0;
  for (; row0 + 128 <= ld; row0 += 128)
    apply_piece<64, NT>(T + (int64_t)row0 * n + (int64_t)ca * 128, cn, xs + ca, lane, out + row0);
  const int rem = ld - row0;  // even, < 128: one piece per binary digit
#define ALFI_PIECE(R)                                                                             \
  if (rem & R) {                                                                                  \
    apply_piece<R / 2, NT>(T + (int64_t)row0 * n + (int64_t)ca * R, cn, xs + ca, lane, out + row0); \
    row0 += R;                                                                                    \
  }
  ALFI_PIECE(64)
  ALFI_PIECE(32)
  ALFI_PIECE(16)
  ALFI_PIECE(8)
  ALFI_PIECE(4)
  ALFI_PIECE(2)
#undef ALFI_PIECE
  __syncthreads();
  double* dst = stage + stage_ptr[p];
  for (int i = threadIdx.x; i < ld; i += 64 * W) {
    double d = part[0][i];
#pragma unroll
    for (int w = 1; w < W; ++w) d += part[w][i];
    dst[i] = d;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3a. additive apply for levels of SMALL patches (every n_p <= 32: the 2-D stars with n_p = 14) from the INTERLEAVED copy of
//     the inverses (patch_il_index, common.h).  A wave per patch would stream 1.5 KB per wave, so G lanes share a patch
//     (lane l owns the row pair 2l, 2l + 1), PW = 64 / G patches share a wave, and column c of the wave's PW patches is ONE
//     contiguous request of PW * G * 16 bytes, the whole wave group nc such requests back to back.  (Rounds 2-3 read the
//     row-piece storage with 8 lanes per patch: a wave instruction touched 3 pieces x 8 patches = 24 segments of 16 .. 64
//     bytes in as many lines; same box, ldc2d's finest level: 4.93 -> 4.37 ms per V-cycle, profiles/r04_ab_cfg2.txt.)
//     G = ceil(max_np / 2) need not be a power of two (n_p = 14: G = 7, 9 patches per wave).
// ---------------------------------------------------------------------------------------------------------------------
template <int G, bool NT>
__global__ __launch_bounds__(256) void patch_apply_il_kernel(int64_t p0, int64_t p1, int nc,
                                                              const int64_t* __restrict__ patch_ptr,
                                                              const int32_t* __restrict__ patch_dofs,
                                                              const int64_t* __restrict__ stage_ptr,
                                                              const double* __restrict__ il,
                                                              const double* __restrict__ x, double* __restrict__ stage) {
  constexpr int PW = 64 / G, LW = PW * G;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t w = p0 / PW + (int64_t)blockIdx.x * 4 + wave;     // wave group; the grid covers [p0 / PW, (p1 - 1) / PW]
  const int q = lane / G, l = lane - q * G;
  const int64_t p = w * PW + q;
  const bool inwave = lane < LW && w * PW < p1;                    // the group exists (its storage is allocated whole)
  const bool live = inwave && p >= p0 && p < p1;
  int n = 0;
  int64_t off = 0;
  if (live) {
    off = patch_ptr[p];
    n = (int)(patch_ptr[p + 1] - off);
  }
  double xa = 0.0, xb = 0.0;   // x_p[l], x_p[l + G]
  if (l < n) xa = x[patch_dofs[off + l]];
  if (l + G < n) xb = x[patch_dofs[off + l + G]];
  // all nc columns of the lane's row pair are requested before the first one is used
  const double* base = il + ((w * nc) * (int64_t)LW + lane) * 2;
  double2 v[2 * G];
#pragma unroll
  for (int c = 0; c < 2 * G; ++c)
    v[c] = (inwave && c < nc) ? load_pair<NT>(base + (int64_t)c * (2 * LW)) : make_double2(0.0, 0.0);
  double acc0 = 0.0, acc1 = 0.0;
  const int src0 = q * G;
#pragma unroll
  for (int c = 0; c < 2 * G; ++c) {
    const double xc = __shfl(c < G ? xa : xb, src0 + (c < G ? c : c - G), 64);
    acc0 = __builtin_fma(v[c].x, xc, acc0);
    acc1 = __builtin_fma(v[c].y, xc, acc1);
  }
  // rows >= n of the interleaved copy are zero, so storing the pair is valid whenever its first row exists (stage has ld slots)
  if (live && 2 * l < n) *reinterpret_cast<double2*>(stage + stage_ptr[p] + 2 * l) = make_double2(acc0, acc1);
}

// interleaved copy <- row-piece storage: one thread per (wave group, column, lane) writes its 16 bytes
__global__ __launch_bounds__(256) void patch_il_build_kernel(int64_t npatch, int G, int nc, int64_t total,
                                                              const int64_t* __restrict__ patch_ptr,
                                                              const int64_t* __restrict__ inv_ptr,
                                                              const double* __restrict__ inv, double* __restrict__ il) {
  const int PW = 64 / G, LW = PW * G;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int lane = (int)(t % LW);
    const int64_t wc = t / LW;
    const int c = (int)(wc % nc);
    const int64_t w = wc / nc;
    const int q = lane / G, l = lane - q * G;
    const int64_t p = w * PW + q;
    double2 o = make_double2(0.0, 0.0);
    if (p < npatch) {
      const int n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
      const int ld = (n + 1) & ~1;
      const int r = 2 * l;
      if (c < n && r < ld) {
        const double* S = inv + inv_ptr[p];
        o.x = S[patch_inv_index(r, c, n, ld)];
        o.y = r + 1 < n ? S[patch_inv_index(r + 1, c, n, ld)] : 0.0;
      }
    }
    *reinterpret_cast<double2*>(il + t * 2) = o;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3b. multiplicative sweep, one dependency wavefront: one wave per patch p of the wavefront,
//       r_p = x_p - (A y)_p ,   y_p += inv(A_p) r_p .
//     The patches of a wavefront are mutually uncoupled (no operator entry links them), so they neither read what another
//     one writes nor write the same dofs: plain stores, result identical to the sequential sweep.
//     Residual: the block rows of the patch's nodes are treated as ONE flat list of blocks (prefix sums of the row
//     lengths in LDS, lane = block, coalesced value planes because a row's blocks are consecutive), row sums by the same
//     segmented wave scan as bsr_spmv_flat_kernel.  Then the apply pieces of section 3 with r_p in LDS.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int MAX_PNODES = 64;

#ifndef ALFI_MULT_U
#define ALFI_MULT_U 16
#endif
#ifndef ALFI_MULT_G
#define ALFI_MULT_G 3
#endif
#ifndef ALFI_MULT_NG
#define ALFI_MULT_NG 3
#endif
#ifndef ALFI_MULT_W
#define ALFI_MULT_W 4
#endif
// -DALFI_MULT_TIMING (measurement builds, scripts/mult_stamps.py): thread 0 of a workgroup of the persistent sweep records the
// 100 MHz wall clock at seven points of every item (ticket | tables | wait over | residual | apply | stores drained | successors
// released); alfi_debug_mult_stamps copies them out
#ifdef ALFI_MULT_TIMING
#define ALFI_MULT_STAMP_PARAM , int64_t* __restrict__ stamps
#define ALFI_MULT_STAMP_ARG , stamps, t
#define ALFI_MULT_STAMP(K)                                                   \
  do {                                                                       \
    if (stamps && threadIdx.x == 0) stamps[(int64_t)t * 8 + (K)] = (int64_t)wall_clock64(); \
  } while (0)
#else
#define ALFI_MULT_STAMP_PARAM
#define ALFI_MULT_STAMP_ARG
#define ALFI_MULT_STAMP(K) do { } while (0)
#endif
#ifndef ALFI_MULT_WG_PER_CU
#define ALFI_MULT_WG_PER_CU 4           // cap on the resident workgroups per CU of the persistent sweep
#endif
#ifndef ALFI_MULT_OCC
#define ALFI_MULT_OCC 2                   // waves per SIMD the sweep kernels are compiled for: two workgroups per CU (three --
                                          // 168 registers, 100-400 bytes of scratch per lane -- ran 27.2 against 25.3 ms)
#endif
#define ALFI_MULT_OCC_ATTR __attribute__((amdgpu_waves_per_eu(ALFI_MULT_OCC, ALFI_MULT_OCC)))
constexpr int MULT_W = ALFI_MULT_W;         // waves per patch
constexpr int MULT_NTHR = 64 * MULT_W;
static_assert(MULT_NTHR >= MAX_NP, "one thread per row entry of a patch in the residual");
constexpr int MULT_U = ALFI_MULT_U;         // 16-byte loads in flight per lane in the sweep's inverse apply (see apply_piece)
constexpr int MULT_G = ALFI_MULT_G;         // operator blocks per thread whose loads leave together (one group)
constexpr int MULT_NG = ALFI_MULT_NG;       // groups per round (G x NG blocks per thread: 2560 blocks per round at 4 waves)
constexpr int MULT_MAXU = MULT_G * MULT_NG;

// same-wave LDS hand-off: the LDS serves a wave's instructions in order, so a wave that writes an array and then reads it
// through other lanes needs no barrier -- only the compiler must keep the order
#define ALFI_WAVE_LDS_ORDER()                                  \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                           \
  } while (0)

// A sweep is a chain of dependent patches (config 4: 337 wavefronts per direction), so the time of an apply is at least the
// number of wavefronts times the latency of ONE patch, whatever the occupancy.  Rounds 1-3 gave a patch to one wave: ~35
// dependent passes over its operator rows and ~150 columns of its inverse behind each other, 75-85 us per patch.  Here a
// WORKGROUP of MULT_W waves shares the patch and the dependent round trips to memory are counted:
//   * tables (before the wait of the persistent schedule -- nothing of it depends on y): the rows' block ranges, and for
//     every operator block of the patch's rows, dealt thread by thread, its index and its column (registers);
//   * residual: the blocks' values and y entries leave in two groups of loads, each thread writes its blocks' products
//     A_k y_col to LDS, then one thread per row entry adds the row's products in ascending order -- the order of the CSR row,
//     whatever the thread layout;
//   * apply: row pieces of >= 64 rows by COLUMN shares (wave w multiplies its share of the columns; the partial results are
//     added in the order of the waves), the small pieces a wave each, at most two groups of loads per wave.
// Clock stamps of config 4's finest level (scripts/mult_stamps.py, profiles/r04_mult_stamps_cfg4.txt) guided this.
struct MultLds {
  double rs[MAX_NP];                          // r_p
  double part[MULT_W][MAX_NP];                // partial products of the waves
  double prod[2][MULT_NTHR * MULT_G * 3];     // products of one group of blocks (BS doubles each), double-buffered
  int32_t pre[MAX_PNODES + 1];                // exclusive prefix of the rows' block counts
  int32_t k0[MAX_PNODES];                     // first block of each row
  int32_t nd[MAX_PNODES];                     // node (block row) of each patch node
  int32_t ticket, ok;
};

// tables of patch p: rows (wave 0; from the level's row table: first block, block count and node of every patch node, written
// once by alfi_patches_set_multiplicative -- patch_ptr -> patch_dofs -> rowptr would be three dependent round trips per item),
// then -- after a barrier -- every thread's blocks of the round starting at block ``base``
template <int BS>
__device__ __forceinline__ void mult_wg_rows(int64_t p, MultLds& S, const int64_t* __restrict__ patch_ptr,
                                             const int32_t* __restrict__ rowtab) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int32_t* rt = rowtab + (p * MAX_PNODES + lane) * 3;
  const int32_t r0 = rt[0], r1 = rt[1], r2 = rt[2];          // entries beyond the patch's nodes are zero
  const int n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
  for (int i = lane; i < ((n + 1) & ~1); i += 64) S.part[wave][i] = 0.0;      // rows this wave does not touch in the apply
  if (wave != 0) return;
  const int nn = n / BS;
  if (lane < nn) {
    S.k0[lane] = r0;
    S.nd[lane] = r2;
  }
  int incl = lane < nn ? r1 : 0;   // inclusive wave scan
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (lane < nn) S.pre[lane + 1] = incl;
  if (lane == 0) S.pre[0] = 0;
}

__device__ __forceinline__ void mult_wg_blocks(const MultLds& S, int nn, int base, const int32_t* __restrict__ colidx,
                                               int32_t (&kk)[MULT_MAXU], int32_t (&cc)[MULT_MAXU]) {
  const int total = S.pre[nn];
#pragma unroll
  for (int u = 0; u < MULT_MAXU; ++u) {
    const int f = base + (int)threadIdx.x + MULT_NTHR * u;
    kk[u] = -1;
    cc[u] = 0;
    if (f < total) {
      int lo = 0, hi = nn;          // row of block f: largest i with pre[i] <= f
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (S.pre[mid] <= f) lo = mid; else hi = mid;
      }
      kk[u] = S.k0[lo] + (f - S.pre[lo]);
    }
  }
#pragma unroll
  for (int u = 0; u < MULT_MAXU; ++u)
    if (kk[u] >= 0) cc[u] = colidx[kk[u]] & 0x7fffffff;
}

// group GI of a round: the loads of its blocks (values and y entries) ...
template <int BS, bool NT>
__device__ __forceinline__ void mult_wg_load(int GI, const int32_t (&kk)[MULT_MAXU], const int32_t (&cc)[MULT_MAXU],
                                             const double* __restrict__ vals, int flat, const double* __restrict__ y,
                                             double (&a_)[MULT_G][BS * BS], double (&yv_)[MULT_G][BS]) {
  constexpr int BB = BS * BS;
#pragma unroll
  for (int g = 0; g < MULT_G; ++g) {
    const int32_t k = kk[GI * MULT_G + g];
#pragma unroll
    for (int e = 0; e < BB; ++e) {
      const double* v = vals + bsr_val_index(flat, k >= 0 ? k : 0, e, BB);
      a_[g][e] = k >= 0 ? (NT ? __builtin_nontemporal_load(v) : *v) : 0.0;
    }
  }
#pragma unroll
  for (int g = 0; g < MULT_G; ++g) {
#pragma unroll
    for (int c = 0; c < BS; ++c) yv_[g][c] = kk[GI * MULT_G + g] >= 0 ? y[(int64_t)cc[GI * MULT_G + g] * BS + c] : 0.0;
  }
}
// ... and their products A_k y_col into prod[buffer GI & 1][block of the group][0 .. BS)
template <int BS>
__device__ __forceinline__ void mult_wg_store(int GI, MultLds& S, const int32_t (&kk)[MULT_MAXU], const double (&a_)[MULT_G][BS * BS],
                                              const double (&yv_)[MULT_G][BS]) {
#pragma unroll
  for (int g = 0; g < MULT_G; ++g) {
    if (kk[GI * MULT_G + g] >= 0) {
      double* q = S.prod[GI & 1] + ((int)threadIdx.x + MULT_NTHR * g) * BS;
#pragma unroll
      for (int r = 0; r < BS; ++r) {
        double sacc = 0.0;
#pragma unroll
        for (int c = 0; c < BS; ++c) sacc = __builtin_fma(a_[g][r * BS + c], yv_[g][c], sacc);
        q[r] = sacc;
      }
    }
  }
}
// the products of the blocks [fa, fz) of a row that lie in the group starting at block gbase, added in ascending order
template <int BS>
__device__ __forceinline__ double mult_wg_rowsum(const double* __restrict__ q, int fa, int fz, int gbase, double racc) {
  const int f1 = min(fz, gbase + MULT_NTHR * MULT_G) - gbase;
  int f = max(fa, gbase) - gbase;
  for (; f + 8 <= f1; f += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = q[(f + u) * BS];
#pragma unroll
    for (int u = 0; u < 8; ++u) racc += v[u];
  }
  for (; f < f1; ++f) racc += q[f * BS];
  return racc;
}

// r_p = x_p - (A y)_p, y_p += inv(A_p) r_p by the whole workgroup, in three pieces (the persistent schedule puts the loads of the
// NEXT item's tables between them); kk, cc: the blocks of round 0 (mult_wg_blocks).
// (1) residual into S.rs -- the caller's barrier follows
template <int BS, bool NT>
__device__ __forceinline__ void mult_wg_residual(int64_t p, MultLds& S, int32_t (&kk)[MULT_MAXU], int32_t (&cc)[MULT_MAXU],
                                                 const int64_t* __restrict__ patch_ptr, const int32_t* __restrict__ colidx,
                                                 const double* __restrict__ vals, int flat, const double* __restrict__ x,
                                                 const double* __restrict__ y) {
  const int n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
  const int nn = n / BS;
  const int total = S.pre[nn];
  // the row entry of this thread (thread e < n: row e / BS, component e % BS) and its right-hand side
  const int e = threadIdx.x, row = e / BS, comp = e % BS;
  const bool mine = e < n;
  const double xe = mine ? x[(int64_t)S.nd[mine ? row : 0] * BS + comp] : 0.0;
  const int fa = mine ? S.pre[row] : 0, fz = mine ? S.pre[row + 1] : 0;
  double racc = 0.0;
  constexpr int GB = MULT_NTHR * MULT_G;      // blocks per group
  for (int base = 0; base < total; base += MULT_NTHR * MULT_MAXU) {
    if (base > 0) {
      __syncthreads();            // the previous round's last products have been added
      mult_wg_blocks(S, nn, base, colidx, kk, cc);
    }
    // the groups of the round, software-pipelined: the loads of group g + 1 are in flight while the rows add the products of
    // group g (two product buffers: a group's buffer is rewritten two barriers after its sums)
    double a_[2][MULT_G][BS * BS], yv_[2][MULT_G][BS];
    mult_wg_load<BS, NT>(0, kk, cc, vals, flat, y, a_[0], yv_[0]);
#pragma unroll
    for (int gi = 0; gi < MULT_NG; ++gi) {
      if (base + gi * GB >= total) break;                                  // uniform
      mult_wg_store<BS>(gi, S, kk, a_[gi & 1], yv_[gi & 1]);
      __syncthreads();
      if (gi + 1 < MULT_NG && base + (gi + 1) * GB < total)
        mult_wg_load<BS, NT>(gi + 1, kk, cc, vals, flat, y, a_[(gi + 1) & 1], yv_[(gi + 1) & 1]);
      racc = mult_wg_rowsum<BS>(S.prod[gi & 1] + comp, fa, fz, base + gi * GB, racc);
    }
  }
  if (mine) S.rs[e] = xe - racc;
}
// (2) the waves' partial products inv(A_p) r_p into S.part -- the caller's barrier follows
template <bool NT>
__device__ __forceinline__ void mult_wg_apply(int64_t p, MultLds& S, const int64_t* __restrict__ patch_ptr,
                                              const int64_t* __restrict__ inv_ptr, const double* __restrict__ inv) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
  const int ld = (n + 1) & ~1;
  const double* T = inv + inv_ptr[p];
  double* out = S.part[wave];
  int row0 = 0;
  // pieces of >= 64 rows: column shares
  const int ca = (int)(((int64_t)wave * n) / MULT_W), cn = (int)(((int64_t)(wave + 1) * n) / MULT_W) - ca;
  for (; row0 + 128 <= ld; row0 += 128)
    apply_piece<64, NT, MULT_U>(T + (int64_t)row0 * n + (int64_t)ca * 128, cn, S.rs + ca, lane, out + row0);
  const int rem = ld - row0;
  if (rem & 64) {
    apply_piece<32, NT, MULT_U>(T + (int64_t)row0 * n + (int64_t)ca * 64, cn, S.rs + ca, lane, out + row0);
    row0 += 64;
  }
  // smaller pieces: a wave each, all columns
  int piece = 0;
#define ALFI_PIECE(R)                                                                    \
  if (rem & R) {                                                                         \
    if (piece % MULT_W == wave) apply_piece<R / 2, NT, MULT_U>(T + (int64_t)row0 * n, n, S.rs, lane, out + row0); \
    row0 += R;                                                                           \
    ++piece;                                                                             \
  }
  ALFI_PIECE(32)
  ALFI_PIECE(16)
  ALFI_PIECE(8)
  ALFI_PIECE(4)
  ALFI_PIECE(2)
#undef ALFI_PIECE
}
// (3) y_p += the sum of the partial products, in the order of the waves.  PUBLISH (persistent schedule): the new y entries leave
// with agent-scope write-through stores, so that a wave on another CU / XCD that acquires afterwards reads them
// (cdna_hip_programming.md Guideline 16, R1).
template <bool PUBLISH>
__device__ __forceinline__ void mult_wg_update(int64_t p, const MultLds& S, const int64_t* __restrict__ patch_ptr,
                                               const int32_t* __restrict__ patch_dofs, double* __restrict__ y) {
  const int64_t off = patch_ptr[p];
  const int n = (int)(patch_ptr[p + 1] - off);
  for (int i = threadIdx.x; i < n; i += MULT_NTHR) {
    const int64_t dof = patch_dofs[off + i];
    double d = S.part[0][i];
#pragma unroll
    for (int w = 1; w < MULT_W; ++w) d += S.part[w][i];
    const double v = y[dof] + d;
    if (PUBLISH)
      __hip_atomic_store(y + dof, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // global_store_dwordx2 ... sc1
    else
      y[dof] = v;
  }
}

// one dependency wavefront per launch, a workgroup per patch
template <int BS, bool NT>
__global__ __launch_bounds__(MULT_NTHR) ALFI_MULT_OCC_ATTR void patch_mult_kernel(int64_t count, const int32_t* __restrict__ seq,
                                                               const int64_t* __restrict__ patch_ptr,
                                                               const int32_t* __restrict__ patch_dofs,
                                                               const int64_t* __restrict__ inv_ptr,
                                                               const double* __restrict__ inv,
                                                               const int32_t* __restrict__ rowtab,
                                                               const int32_t* __restrict__ colidx,
                                                               const double* __restrict__ vals, int flat,
                                                               const double* __restrict__ x, double* __restrict__ y) {
  __shared__ MultLds S;
  const int64_t p = seq[blockIdx.x];
  mult_wg_rows<BS>(p, S, patch_ptr, rowtab);
  __syncthreads();
  int32_t kk[MULT_MAXU], cc[MULT_MAXU];
  mult_wg_blocks(S, (int)(patch_ptr[p + 1] - patch_ptr[p]) / BS, 0, colidx, kk, cc);
  mult_wg_residual<BS, NT>(p, S, kk, cc, patch_ptr, colidx, vals, flat, x, y);
  __syncthreads();
  mult_wg_apply<NT>(p, S, patch_ptr, inv_ptr, inv);
  __syncthreads();
  mult_wg_update<false>(p, S, patch_ptr, patch_dofs, y);
}

// The whole sweep (both directions of a symmetrised one) as ONE launch of a resident grid: workgroups draw the items of the
// wavefront-major schedule in order (a ticket counter) and start an item when its predecessor count has reached zero -- the
// predecessors of an item are the LAST WRITERS of the nodes it reads (host: alfi_patches_set_multiplicative), each finished
// item decrements its successors.  The schedule order is a topological order and every ticket is held by a running workgroup,
// so every wait ends; the independent patches of wavefront k + 1 start while the tail of wavefront k is still running, a
// workgroup reads its patch's tables while it waits, and the 674 launch boundaries of config 4's symmetrised sweep are gone.
// Results are those of the launch-per-wavefront schedule bit for bit: the same patches see the same y entries (conflicting
// patches keep their order), the same code computes each.
// Hand-off (Guideline 16, R1): y entries leave with write-through stores, every storing wave drains them (vmcnt(0)), the
// workgroup meets at a barrier, then the successors are decremented; a consumer polls its own counter relaxed (one lane), then
// every wave does ONE agent-scope acquire, then plain loads.
// err[0]: set when a wait ran into its bound (a broken schedule would otherwise spin until the watchdog).
// Round 5: the tables of the NEXT item are fetched in the shadow of the current one.  A resident workgroup used to spend 7.4 us
// of its 34 us per item on the chain ticket -> item -> row table -> block columns before it even looked at its predecessor count
// (profiles/r04_mult_stamps_cfg4.txt), and the sweep is bound by slots x time per item.  Now the ticket of the next item is drawn
// when the wait of the current one ends, its patch number is read after the residual, its row table during the apply, and the rows
// go to LDS behind the update (the residual is the only reader of the row tables) -- each of the dependent round trips behind one
// phase of the current item; only the block columns (one round trip) stay in front of the next wait.  Drawing a ticket early
// keeps the schedule safe: a workgroup still works its tickets in order, every drawn ticket belongs to a resident workgroup, and
// the predecessors of ticket t carry smaller tickets.
template <int BS, bool NT>
__global__ __launch_bounds__(MULT_NTHR) ALFI_MULT_OCC_ATTR void patch_mult_persistent_kernel(
    int32_t nitems, const int32_t* __restrict__ items, int32_t* __restrict__ pred, const int32_t* __restrict__ succ_ptr,
    const int32_t* __restrict__ succ, int32_t* __restrict__ head, int32_t* __restrict__ err,
    const int64_t* __restrict__ patch_ptr, const int32_t* __restrict__ patch_dofs, const int64_t* __restrict__ inv_ptr,
    const double* __restrict__ inv, const int32_t* __restrict__ rowtab, const int32_t* __restrict__ colidx,
    const double* __restrict__ vals, int flat, const double* __restrict__ x, double* __restrict__ y ALFI_MULT_STAMP_PARAM) {
  __shared__ MultLds S;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // the first item of this workgroup: tables in the open
  if (threadIdx.x == 0) S.ticket = __hip_atomic_fetch_add(head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  int32_t t = S.ticket;
  if (t >= nitems) return;
  int64_t p = items[t];
  mult_wg_rows<BS>(p, S, patch_ptr, rowtab);
  __syncthreads();
  int32_t kk[MULT_MAXU], cc[MULT_MAXU];
  mult_wg_blocks(S, (int)(patch_ptr[p + 1] - patch_ptr[p]) / BS, 0, colidx, kk, cc);
  for (;;) {
    ALFI_MULT_STAMP(1);
    // wait for the predecessors: ONE lane polls ONE word, relaxed
    if (threadIdx.x == 0) {
      int ok = 1;
      unsigned spins = 0;
      // (a RELAXED poll: an acquire load invalidates the caches at every iteration of the spin -- measured on the macro-star
      // sweep: 3.6 x slower; the acquire is the agent fence every wave executes after the barrier, pairing with the RELEASE
      // decrement of the predecessors below -- ADVICE r4)
      while (__hip_atomic_load(pred + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 24) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          ok = 0;
          break;
        }
      }
      S.ok = ok;
    }
    __syncthreads();
    if (!S.ok) {
      if (threadIdx.x == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    ALFI_MULT_STAMP(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // this CU's L1 forgets what other CUs have rewritten
    // (next item, 1) its ticket: the atomic returns behind the residual
    int32_t tk = 0;
    if (threadIdx.x == 0) tk = __hip_atomic_fetch_add(head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mult_wg_residual<BS, NT>(p, S, kk, cc, patch_ptr, colidx, vals, flat, x, y);
    if (threadIdx.x == 0) S.ticket = tk;
    __syncthreads();                                            // S.rs complete; the ticket is there
    ALFI_MULT_STAMP(3);
    const int32_t tn = S.ticket;
    const bool more = tn < nitems;
    // (next item, 2) its patch and, in wave 0, its row table: requested now, consumed behind the apply
    const int64_t pn = more ? items[tn] : 0;
    mult_wg_apply<NT>(p, S, patch_ptr, inv_ptr, inv);
    int32_t r0 = 0, r1 = 0, r2 = 0, nnx = 0;
    if (more && wave == 0) {
      const int32_t* rt = rowtab + (pn * MAX_PNODES + lane) * 3;
      r0 = rt[0];
      r1 = rt[1];
      r2 = rt[2];          // entries beyond the patch's nodes are zero
      nnx = (int)(patch_ptr[pn + 1] - patch_ptr[pn]) / BS;
    }
    __syncthreads();                                            // the partial products are complete
    ALFI_MULT_STAMP(4);
    mult_wg_update<true>(p, S, patch_ptr, patch_dofs, y);
    // (next item, 3) its rows into the tables: their last reader was this item's residual
    if (more && wave == 0) {
      if (lane < nnx) {
        S.k0[lane] = r0;
        S.nd[lane] = r2;
      }
      int incl = lane < nnx ? r1 : 0;   // inclusive wave scan
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
      }
      if (lane < nnx) S.pre[lane + 1] = incl;
      if (lane == 0) S.pre[0] = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's y stores have left
    __syncthreads();                                            // ... and those of the other waves; the next tables are there
    ALFI_MULT_STAMP(5);
    // release, at the level of the ISA: the y stores of this item are WRITE-THROUGH (sc1: they bypass this XCD's L2 as far as
    // other agents' reads are concerned), drained by the s_waitcnt above, and the barrier orders the other waves' stores before
    // the decrements -- so a RELAXED decrement publishes them.  (A __ATOMIC_RELEASE decrement, as ADVICE r4 proposed, makes the
    // compiler put an L2 write-back, buffer_wbl2 sc1, in front of EVERY atomic: config 4's sweep 25.5 -> 41.8 ms.  The acquire side
    // is the agent fence each consumer wave executes after its wait.)
    for (int32_t e = succ_ptr[t] + threadIdx.x; e < succ_ptr[t + 1]; e += MULT_NTHR)
      __hip_atomic_fetch_sub(pred + succ[e], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ALFI_MULT_STAMP(6);
    if (!more) break;
    // (next item, 4) the partial-product rows its waves will not touch, and its blocks of round 0
    t = tn;
    p = pn;
    ALFI_MULT_STAMP(0);
    {
      const int n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
      for (int i = lane; i < ((n + 1) & ~1); i += 64) S.part[wave][i] = 0.0;
      mult_wg_blocks(S, n / BS, 0, colidx, kk, cc);
    }
  }
}

// stage 2: dof-wise sum of the staged patch results in a fixed order (deterministic; replaces PETSc's scatter-add)
__global__ __launch_bounds__(256) void patch_sum_kernel(int64_t i0, int64_t n, const int32_t* __restrict__ dof_ptr,
                                                         const int32_t* __restrict__ dof_pos,
                                                         const double* __restrict__ stage,
                                                         const uint8_t* __restrict__ bc_mask,
                                                         const double* __restrict__ x, double* __restrict__ y, int pou) {
  const int64_t i = i0 + (int64_t)blockIdx.x * 256 + threadIdx.x;   // dofs [i0, n)
  if (i >= n) return;
  double s = 0.0;
  const int32_t q0 = dof_ptr[i], q1 = dof_ptr[i + 1];
  for (int32_t q = q0; q < q1; ++q) s += stage[dof_pos[q]];
  if (pou && q1 - q0 > 1) s /= (double)(q1 - q0);   // patch_pc_patch_partition_of_unity: weight 1 / (patches holding the dof)
  y[i] = bc_mask[i] ? x[i] : s;   // Dirichlet dofs: y[bc] = x[bc]
}

// ---------------------------------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------------------------------
int launch_patch_gather_dense(alfi_level* L) {
  alfi_ctx* ctx = L->ctx;
  if (L->npatch == 0) return 0;
  dim3 grid((unsigned)L->npatch), block(256);
  if (L->bs == 2)
    hipLaunchKernelGGL(patch_gather_dense_kernel<2>, grid, block, 0, ctx->stream, L->A.rowptr, L->A.colidx, L->A.vals,
                       L->patch_ptr, L->patch_dofs, L->inv_ptr, L->inv, L->A.flat);
  else if (L->bs == 3)
    hipLaunchKernelGGL(patch_gather_dense_kernel<3>, grid, block, 0, ctx->stream, L->A.rowptr, L->A.colidx, L->A.vals,
                       L->patch_ptr, L->patch_dofs, L->inv_ptr, L->inv, L->A.flat);
  else
    return alfi_set_error(ctx, ALFI_E_ARG, "unsupported block size %d", L->bs);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

template <int N>
static int launch_invert_small(alfi_ctx* ctx, int64_t nmat, const int64_t* ptr, const int64_t* inv_ptr, int fixed_n,
                               int64_t fixed_stride, double* inv, int* status) {
  const int per_block = 256 / N;
  dim3 grid((unsigned)((nmat + per_block - 1) / per_block)), block(256);
  if (ptr)   // patches: row-piece layout
    hipLaunchKernelGGL((invert_small_kernel<N, true>), grid, block, 0, ctx->stream, nmat, ptr, inv_ptr, fixed_n,
                       fixed_stride, inv, status);
  else
    hipLaunchKernelGGL((invert_small_kernel<N, false>), grid, block, 0, ctx->stream, nmat, ptr, inv_ptr, fixed_n,
                       fixed_stride, inv, status);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

int launch_invert_small_any(alfi_ctx* ctx, int nmax, int64_t nmat, const int64_t* ptr, const int64_t* inv_ptr,
                            int fixed_n, int64_t fixed_stride, double* inv, int* status) {
  if (nmat == 0) return 0;
  if (nmax <= 8) return launch_invert_small<8>(ctx, nmat, ptr, inv_ptr, fixed_n, fixed_stride, inv, status);
  if (nmax <= 16) return launch_invert_small<16>(ctx, nmat, ptr, inv_ptr, fixed_n, fixed_stride, inv, status);
  if (nmax <= 32) return launch_invert_small<32>(ctx, nmat, ptr, inv_ptr, fixed_n, fixed_stride, inv, status);
  return alfi_set_error(ctx, ALFI_E_ARG, "small inversion supports n <= 32, got %d", nmax);
}

// in-place inversion of npatch dense matrices (row-major n x ld at inv + inv_ptr[p], n = patch_ptr[p+1] - patch_ptr[p] <=
// max_np <= 160) into the row-piece layout: the level's patches, and the interior blocks of a Schoeberl transfer whose
// block size exceeds 32
int launch_patch_invert_arrays(alfi_ctx* ctx, int64_t npatch, int max_np, const int64_t* patch_ptr,
                               const int64_t* inv_ptr, double* inv, int* status) {
  if (npatch == 0) return 0;
  if (max_np <= 32) return launch_invert_small_any(ctx, max_np, npatch, patch_ptr, inv_ptr, 0, 0, inv, status);
  {
    int handled = 0;       // 33 .. 160 dofs: block Gauss-Jordan on the FP64 matrix cores (kernels_invert.hip)
    ALFI_CHECK(launch_patch_invert_mfma(ctx, npatch, max_np, patch_ptr, inv_ptr, inv, status, &handled));
    if (handled) return 0;
  }
  dim3 grid((unsigned)npatch), block(256);
  if (max_np <= 112)
    hipLaunchKernelGGL(patch_invert_big_kernel<7>, grid, block, 0, ctx->stream, patch_ptr, inv_ptr, inv, status);
  else if (max_np <= SMALL_PATCH_MAX)
    hipLaunchKernelGGL(patch_invert_big_kernel<10>, grid, block, 0, ctx->stream, patch_ptr, inv_ptr, inv, status);
  else
    return alfi_set_error(ctx, ALFI_E_ARG, "register inversion handles sizes <= %d, got %d", SMALL_PATCH_MAX, max_np);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

int launch_patch_invert(alfi_level* L) {
  return launch_patch_invert_arrays(L->ctx, L->npatch, L->max_np, L->patch_ptr, L->inv_ptr, L->inv, L->status);
}

// stage[stage_ptr[p] + r] = sum_c inv_p[r][c] x[patch_dofs[patch_ptr[p] + c]] for npatch row-piece inverses of size <= 160
int launch_patch_apply_arrays(alfi_ctx* ctx, int64_t npatch, const int64_t* patch_ptr, const int32_t* patch_dofs,
                              const int64_t* inv_ptr, const int64_t* stage_ptr, const double* inv, const double* x,
                              double* stage) {
  if (npatch == 0) return 0;
  hipLaunchKernelGGL((patch_apply_kernel<true, APPLY_W>), dim3((unsigned)npatch), dim3(64 * APPLY_W), 0, ctx->stream, (int64_t)0,
                     npatch, patch_ptr, patch_dofs, inv_ptr, stage_ptr, inv, x, stage);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

int launch_patch_mult_wave(alfi_level* L, const int32_t* seq, int64_t count, const double* x, double* y) {
  alfi_ctx* ctx = L->ctx;
  if (count == 0) return 0;
  if (L->mult_big) return launch_big_mult_wave(L, seq, count, x, y);
  dim3 grid((unsigned)count), block(MULT_NTHR);
#define ALFI_MULT(BSV, NTV)                                                                                           \
  hipLaunchKernelGGL((patch_mult_kernel<BSV, NTV>), grid, block, 0, ctx->stream, count, seq, L->patch_ptr,            \
                     L->patch_dofs, L->inv_ptr, L->inv, L->mult_rowtab, L->A.colidx, L->A.vals, L->A.flat, x, y)
  if (L->bs == 2) {
    ALFI_MULT(2, true);
  } else if (L->bs == 3) {
    ALFI_MULT(3, true);
  } else {
    return alfi_set_error(ctx, ALFI_E_ARG, "unsupported block size %d", L->bs);
  }
#undef ALFI_MULT
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

#ifdef ALFI_MULT_TIMING
static int64_t* g_mult_stamps = nullptr;
static int64_t g_mult_stamps_n = 0;
extern "C" int64_t alfi_debug_mult_stamps(int64_t* out, int64_t nitems) {
  if (!g_mult_stamps || nitems > g_mult_stamps_n) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpy(out, g_mult_stamps, sizeof(int64_t) * 8 * (size_t)nitems, hipMemcpyDeviceToHost) != hipSuccess) return -3;
  return nitems;
}
#endif

int launch_patch_mult_persistent(alfi_level* L, const double* x, double* y) {
  alfi_ctx* ctx = L->ctx;
  if (L->mult_nitems == 0) return 0;
  // fresh counters for this apply: predecessor counts and the ticket (the error word is the ctx's, sticky)
  ALFI_HIP_CHECK(ctx, hipMemcpyAsync(L->mult_pred, L->mult_pred0, sizeof(int32_t) * (size_t)L->mult_nitems,
                                     hipMemcpyDeviceToDevice, ctx->stream));
  ALFI_HIP_CHECK(ctx, hipMemsetAsync(L->mult_ctl, 0, 4 * sizeof(int32_t), ctx->stream));
  if (L->mult_big) return launch_big_mult_persistent(L, x, y);      // macro stars: a workgroup per item (kernels_bigpatch.hip)
  // a resident grid: as many workgroups per CU as the kernel's registers and LDS admit (every ticket holder must be running;
  // more workgroups in flight = more of the next wavefronts' independent patches started early)
  // (per ctx -- ADVICE r4: function-local statics sized every later ctx's grid from the first ctx's device)
  if (ctx->mult_ncu == 0) {
    hipDeviceProp_t prop;
    ALFI_HIP_CHECK(ctx, hipGetDeviceProperties(&prop, ctx->device));
    const auto k2 = &patch_mult_persistent_kernel<2, true>;
    const auto k3 = &patch_mult_persistent_kernel<3, true>;
    ALFI_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&ctx->mult_per_cu[0], k2, MULT_NTHR, 0));
    ALFI_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&ctx->mult_per_cu[1], k3, MULT_NTHR, 0));
    ctx->mult_ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 64;
  }
  const int pc = std::max(1, std::min(ctx->mult_per_cu[L->bs == 3 ? 1 : 0], ALFI_MULT_WG_PER_CU));
  dim3 grid((unsigned)std::min<int64_t>(L->mult_nitems, (int64_t)ctx->mult_ncu * pc)), block(MULT_NTHR);
#ifdef ALFI_MULT_TIMING
  if (g_mult_stamps_n < L->mult_nitems) {
    if (g_mult_stamps) (void)hipFree(g_mult_stamps);
    ALFI_HIP_CHECK(ctx, hipMalloc(&g_mult_stamps, sizeof(int64_t) * 8 * (size_t)L->mult_nitems));
    g_mult_stamps_n = L->mult_nitems;
  }
#define ALFI_PMULT_STAMPS , g_mult_stamps
#else
#define ALFI_PMULT_STAMPS
#endif
#define ALFI_PMULT(BSV)                                                                                                   \
  hipLaunchKernelGGL((patch_mult_persistent_kernel<BSV, true>), grid, block, 0, ctx->stream, L->mult_nitems, L->mult_items, \
                     L->mult_pred, L->mult_succ_ptr, L->mult_succ, L->mult_ctl, ctx->dev_err, L->patch_ptr,               \
                     L->patch_dofs, L->inv_ptr, L->inv, L->mult_rowtab, L->A.colidx, L->A.vals, L->A.flat, x, y ALFI_PMULT_STAMPS)
  if (L->bs == 2) ALFI_PMULT(2);
  else if (L->bs == 3) ALFI_PMULT(3);
  else return alfi_set_error(ctx, ALFI_E_ARG, "unsupported block size %d", L->bs);
#undef ALFI_PMULT
#undef ALFI_PMULT_STAMPS
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

// stage 1 for the patches [p0, p1) of the level
int launch_patch_apply_range(alfi_level* L, int64_t p0, int64_t p1, const double* x) {
  alfi_ctx* ctx = L->ctx;
  if (p1 <= p0) return 0;
  int t = alfi_prof_begin(ctx, ALFI_EV_PATCH_APPLY);
  if (L->cond) {                                // condensed factors (kernels_bigpatch.hip)
    ALFI_CHECK(launch_cond_apply_range(L, p0, p1, x));
    alfi_prof_end(ctx, t);
    return 0;
  }
  // levels with FEW star patches of 3-D size (the lower levels of a hierarchy: 125 and 729 patches under config 3's 35 937)
  // leave most CUs empty: up to 1000 patches each patch gets 8 waves whatever its size (729 patches: 13.8 us against 15.6 us
  // through the macro stars' kernel, whose row pieces -- 64 + 32 + 8 + 4 + 2 rows for 111 dofs -- are dealt unevenly to 4
  // waves); with fewer patches than CUs the macro stars' kernel and its several workgroups per patch win (125 patches: 7.5
  // against 8.6 us).  Config 3: 17.27 -> 17.1 ms per cycle, same box.
  // (ALFI_TEST_LARGE_PATHS: no level counts as small, so that test hierarchies take the kernels of the large levels)
  const int64_t few = alfi_test_large_paths() ? 0 : 1000, fewer = alfi_test_large_paths() ? 0 : 256;
  if (L->max_np > SMALL_PATCH_MAX || (L->max_np > 64 && L->npatch <= fewer)) {     // kernels_bigpatch.hip
    ALFI_CHECK(launch_big_apply_range(L, p0, p1, x));
    alfi_prof_end(ctx, t);
    return 0;
  }
  const int64_t cnt = p1 - p0;
  dim3 grid((unsigned)((cnt + 3) / 4)), block(256);
  // (the inverses are read once per apply: nontemporal loads keep x, the staging buffer and the index arrays in cache)
  if (L->il_valid) {
    // small patches, interleaved copy of the inverses: G lanes per patch, 64 / G patches per wave
    const int PW = 64 / L->il_G;
    const int64_t nwave = (p1 - 1) / PW - p0 / PW + 1;
    dim3 igrid((unsigned)((nwave + 3) / 4));
#define ALFI_IL(GV)                                                                                                      \
  case GV:                                                                                                               \
    hipLaunchKernelGGL((patch_apply_il_kernel<GV, true>), igrid, block, 0, ctx->stream, p0, p1, L->il_nc, L->patch_ptr,  \
                       L->patch_dofs, L->stage_ptr, L->inv_il, x, L->stage);                                             \
    break;
    switch (L->il_G) {
      ALFI_IL(4) ALFI_IL(7) ALFI_IL(8) ALFI_IL(12) ALFI_IL(16)
      default: return alfi_set_error(ctx, ALFI_E_STATE, "interleaved patch storage with %d lanes per patch", L->il_G);
    }
#undef ALFI_IL
  } else {
    if (L->max_np > 128 || (L->max_np > 64 && L->npatch <= few))
      hipLaunchKernelGGL((patch_apply_kernel<true, 2 * APPLY_W>), dim3((unsigned)cnt), dim3(128 * APPLY_W), 0, ctx->stream, p0,
                         p1, L->patch_ptr, L->patch_dofs, L->inv_ptr, L->stage_ptr, L->inv, x, L->stage);
    else if (L->max_np > 64)
      hipLaunchKernelGGL((patch_apply_kernel<true, APPLY_W>), dim3((unsigned)cnt), dim3(64 * APPLY_W), 0, ctx->stream, p0, p1,
                         L->patch_ptr, L->patch_dofs, L->inv_ptr, L->stage_ptr, L->inv, x, L->stage);
    else
      hipLaunchKernelGGL((patch_apply_kernel<true, 1>), dim3((unsigned)cnt), dim3(64), 0, ctx->stream, p0, p1, L->patch_ptr,
                         L->patch_dofs, L->inv_ptr, L->stage_ptr, L->inv, x, L->stage);
  }
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  alfi_prof_end(ctx, t);
  return 0;
}

// stage 2: dof-wise sum of the staged results (+ Dirichlet copy) for the dofs [i0, i1)
int launch_patch_sum_range(alfi_level* L, int64_t i0, int64_t i1, const double* x, double* y) {
  alfi_ctx* ctx = L->ctx;
  if (i1 <= i0) return 0;
  int t = alfi_prof_begin(ctx, ALFI_EV_PATCH_SCATTER);
  dim3 grid((unsigned)((i1 - i0 + 255) / 256)), block(256);
  hipLaunchKernelGGL(patch_sum_kernel, grid, block, 0, ctx->stream, i0, i1, L->dof_ptr, L->dof_pos, L->stage, L->bc_mask,
                     x, y, L->pou ? 1 : 0);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  alfi_prof_end(ctx, t);
  return 0;
}

int launch_patch_sum(alfi_level* L, const double* x, double* y) { return launch_patch_sum_range(L, 0, L->n, x, y); }

// Small-patch levels (every n_p <= 32, dense inverses): (re)build the interleaved copy the additive apply streams.  Called
// at the end of every factorisation and after a pivoted repair has rewritten some inverses.
int build_patch_il(alfi_level* L) {
  alfi_ctx* ctx = L->ctx;
  L->il_valid = false;
  if (L->cond || L->npatch == 0 || L->max_np > 32 || L->inv_shrunk) return 0;
  const int need = (L->max_np + 1) / 2;
  const int G = need <= 4 ? 4 : need <= 7 ? 7 : need <= 8 ? 8 : need <= 12 ? 12 : 16;
  const int PW = 64 / G, LW = PW * G, nc = L->max_np;
  const int64_t ngroups = (L->npatch + PW - 1) / PW;
  const int64_t total = ngroups * nc * LW;          // 16-byte entries
  if (!L->inv_il || L->il_doubles != 2 * total) {
    ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (L->inv_il) (void)hipFree(L->inv_il);
    L->inv_il = nullptr;
    L->il_doubles = 0;
    ALFI_HIP_CHECK(ctx, hipMalloc((void**)&L->inv_il, sizeof(double) * (size_t)(2 * total)));
    L->il_doubles = 2 * total;
  }
  const int64_t blocks = std::min<int64_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(patch_il_build_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, L->npatch, G, nc, total,
                     L->patch_ptr, L->inv_ptr, L->inv, L->inv_il);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  L->il_G = G;
  L->il_nc = nc;
  L->il_valid = true;
  return 0;
}

int launch_patch_apply(alfi_level* L, const double* x, double* y) {
  ALFI_CHECK(launch_patch_apply_range(L, 0, L->npatch, x));
  return launch_patch_sum(L, x, y);
}
