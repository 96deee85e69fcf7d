// S5: batched in-place inversion of the star-patch matrices (33 .. 160 dofs) on the FP64 matrix cores.
// (PCPATCH's `dense_inverse`, alfi/solver.py:599-602: LAPACK getrf + getri per patch in the reference, paid on every
// Newton step.)
//
// Block Gauss-Jordan with 4 x 4 pivot blocks -- the k depth of v_mfma_f64_16x16x4 -- on a register-resident matrix:
// a workgroup (8 waves) holds one padded patch matrix as NT x NT tiles of 16 x 16 in the MFMA accumulator layout, wave w
// the tiles (ti, tj) with ti % 4 == w >> 1, tj % 2 == w & 1 (<= 15 tiles = 60 doubles per lane for NT = 10).  Step s
// (pivot rows / columns K = 4 s .. 4 s + 3):
//     panels      the owners put the raw row panel A[K, :] (4 x N) and column panel A[:, K] (N x 4) into LDS   -- barrier
//     pivot       every lane factors the 4 x 4 block D = A[K, K] = L U (unpivoted, redundantly: no second barrier)
//     operands    the four scalar Gauss-Jordan steps of the block, delayed: B = rows r_t = the scaled pivot rows at the time of
//                 step t (L^-1 A[K, :], scaled), A-operand = columns c_t at the time of step t (A[:, K] U^-1, unscaled);
//                 columns K of B and rows K of A zeroed
//     update      ONE rank-4 MFMA per tile:  A <- A - sum_t c_t r_t^T               (all tiles, no special cases)
//     fix-up      rows K <- D^-1 A[K, :] (back substitution on the lane's own column), columns K <- - (A[:, K] U^-1) L^-1 (one
//                 more MFMA per tile of that tile column), A[K, K] <- U^-1 L^-1
// D^-1 is never formed: see Lu4 below.
// Against the rank-1 register kernel it replaces (patch_invert_big_kernel: 21 LDS operand reads and one barrier per
// pivot for 100 FMAs, 9.5 TFLOP/s = 12 % of peak at config 4): a quarter of the barriers, and the operands of the 2 n^3
// flops travel inside the MFMA instead of through LDS broadcasts.  No pivoting across blocks (as before): every
// factorisation is followed by the residual probe + pivoted repair of kernels_check.hip.
//
// MFMA operand maps (cdna_hip_programming.md): A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15],
// C/D: column = lane & 15, row = (lane >> 4) + 4 * reg.
#include <cstdlib>
#include <type_traits>
#include "common.h"

// -DALFI_INVERT_TIMING (measurement builds, scripts/invert_phases.py): every wave adds up the shader clock spent in the phases
// of a block step (panels + barrier | LU of the pivot block | operands | update MFMAs issued | column fix-up | row fix-up);
// alfi_debug_invert_phases copies the sums out
#ifdef ALFI_INVERT_TIMING
static __device__ long long* g_invert_phases_dev = nullptr;
#define ALFI_INV_PH(I)                                   \
  do {                                                   \
    __builtin_amdgcn_sched_barrier(0);                   \
    const long long now_ = (long long)__builtin_readcyclecounter(); \
    __builtin_amdgcn_sched_barrier(0);                   \
    tacc[I] += now_ - tprev;                             \
    tprev = now_;                                        \
  } while (0)
#else
#define ALFI_INV_PH(I) do { } while (0)
#endif

namespace {

typedef double inv_d4 __attribute__((ext_vector_type(4)));

// Unpivoted LU of the 4 x 4 pivot block D = L U (L unit lower) in registers.  The block step never forms D^-1: with patch
// operators of condition ~ gamma / nu = 1e7 the 4 x 4 diagonal blocks are that ill-conditioned themselves (gamma b b^T + nu K
// per node), and a Schur complement formed as A[:, K] (D^-1 A[K, :]) loses cond(D) eps -- measured: 62 % of config 4's patches
// fail the 1e-6 residual probe (worst 2e-3) -- whereas (A[:, K] U^-1)(L^-1 A[K, :]) is what the scalar elimination computes.
struct Lu4 {
  double l10, l20, l30, l21, l31, l32;        // L
  double u01, u02, u03, u12, u13, u23;        // strict upper part of U
  double i0, i1, i2, i3;                      // 1 / U_tt
  double s01, s02, s03, s12, s13, s23;        // U_ut / U_uu: the scaled pivot rows inside the block
};

// 1 / u by the hardware estimate and two Newton steps (five dependent instructions; the IEEE division the compiler emits for
// 1.0 / u is a chain of twelve with two quarter-rate ones, four times per block step on the critical path of all eight waves).
// Not correctly rounded (<= 1 ulp): the factors stay consistent because every use takes this same value.
__device__ __forceinline__ double inv_rcp(double u) {
  double r = __builtin_amdgcn_rcp(u);
  r = __builtin_fma(__builtin_fma(-u, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-u, r, 1.0), r, r);
  return r;
}

__device__ __forceinline__ void lu4(const double (&d)[4][4], Lu4& f, bool& bad) {
  const double u00 = d[0][0];
  f.i0 = inv_rcp(u00);
  f.u01 = d[0][1]; f.u02 = d[0][2]; f.u03 = d[0][3];
  f.l10 = d[1][0] * f.i0; f.l20 = d[2][0] * f.i0; f.l30 = d[3][0] * f.i0;
  const double u11 = __builtin_fma(-f.l10, f.u01, d[1][1]);
  f.u12 = __builtin_fma(-f.l10, f.u02, d[1][2]);
  f.u13 = __builtin_fma(-f.l10, f.u03, d[1][3]);
  f.i1 = inv_rcp(u11);
  f.l21 = __builtin_fma(-f.l20, f.u01, d[2][1]) * f.i1;
  f.l31 = __builtin_fma(-f.l30, f.u01, d[3][1]) * f.i1;
  const double u22 = __builtin_fma(-f.l21, f.u12, __builtin_fma(-f.l20, f.u02, d[2][2]));
  f.u23 = __builtin_fma(-f.l21, f.u13, __builtin_fma(-f.l20, f.u03, d[2][3]));
  f.i2 = inv_rcp(u22);
  f.l32 = __builtin_fma(-f.l31, f.u12, __builtin_fma(-f.l30, f.u02, d[3][2])) * f.i2;
  const double u33 = __builtin_fma(-f.l32, f.u23, __builtin_fma(-f.l31, f.u13, __builtin_fma(-f.l30, f.u03, d[3][3])));
  f.i3 = inv_rcp(u33);
  if (u00 == 0.0 || u11 == 0.0 || u22 == 0.0 || u33 == 0.0) bad = true;
  f.s01 = f.u01 * f.i0; f.s02 = f.u02 * f.i0; f.s03 = f.u03 * f.i0;
  f.s12 = f.u12 * f.i1; f.s13 = f.u13 * f.i1;
  f.s23 = f.u23 * f.i2;
}

// one of four values by a per-lane index, as three SELECTS: the nested conditional expression compiled to divergent control
// flow (two s_and_saveexec / s_or pairs and a branch around every pick -- a dozen picks per block step)
__device__ __forceinline__ double sel4(int i, double a0, double a1, double a2, double a3) {
  double r = a0;
  r = i == 1 ? a1 : r;
  r = i == 2 ? a2 : r;
  r = i == 3 ? a3 : r;
  return r;
}

// x = one row of the raw column panel A[i, K].  c[t] = the column t at the time of scalar step t (the A operand)
__device__ __forceinline__ void col_panel(const Lu4& f, const double (&x)[4], double (&c)[4]) {
  c[0] = x[0];
  c[1] = __builtin_fma(-f.s01, c[0], x[1]);
  c[2] = __builtin_fma(-f.s12, c[1], __builtin_fma(-f.s02, c[0], x[2]));
  c[3] = __builtin_fma(-f.s23, c[2], __builtin_fma(-f.s13, c[1], __builtin_fma(-f.s03, c[0], x[3])));
}

// w = x D^-1 = (x U^-1) L^-1 for one row x (4 entries): the final content of the columns K is - w, row lk of D^-1 is this with
// x = e_lk
__device__ __forceinline__ void times_dinv(const Lu4& f, const double (&x)[4], double (&w)[4]) {
  const double v0 = x[0] * f.i0;
  const double v1 = __builtin_fma(-f.u01, v0, x[1]) * f.i1;
  const double v2 = __builtin_fma(-f.u12, v1, __builtin_fma(-f.u02, v0, x[2])) * f.i2;
  const double v3 = __builtin_fma(-f.u23, v2, __builtin_fma(-f.u13, v1, __builtin_fma(-f.u03, v0, x[3]))) * f.i3;
  w[3] = v3;
  w[2] = __builtin_fma(-f.l32, w[3], v2);
  w[1] = __builtin_fma(-f.l31, w[3], __builtin_fma(-f.l21, w[2], v1));
  w[0] = __builtin_fma(-f.l30, w[3], __builtin_fma(-f.l20, w[2], __builtin_fma(-f.l10, w[1], v0)));
}

// 8 waves per patch: wave w holds the tiles (ti, tj) with ti % 4 == w >> 1, tj % 2 == w & 1 -- at most 3 x 5 = 15 tiles = 60
// doubles per lane for NT = 10, so the whole working set stays in architectural VGPRs (with 4 waves and 25 tiles per wave the
// accumulators live in AGPRs and every panel extraction / fix-up pays v_accvgpr moves; the compiler spilled 100+ registers),
// and two waves share a SIMD: the scalar section of one (pivot-block LU: four dependent divisions) runs under the MFMAs of
// the other.  The loop over the pivot tiles is FULLY unrolled, so every accumulator index is a compile-time constant and
// "is this my tile row / column" is a wave-uniform branch, not a select over all tiles.
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int NT>
__global__ __launch_bounds__(512) void patch_invert_mfma_kernel(const int64_t* __restrict__ patch_ptr,
                                                                 const int64_t* __restrict__ inv_ptr,
                                                                 double* __restrict__ inv, int* __restrict__ status) {
  constexpr int N = 16 * NT;
  constexpr int NA = (NT + 3) / 4;            // tile rows per wave (ti = 4 a + wr)
  constexpr int NB = (NT + 1) / 2;            // tile columns per wave (tj = 2 b + wc)
  // raw panels, four doubles per matrix column / row side by side: a lane fetches the pivot rows of its column (the pivot
  // columns of its row) with two 16-byte reads
  __shared__ __attribute__((aligned(16))) double Rraw[2][N][4];   // raw row panel, transposed: Rraw[c][i] = A[K + i][c]
  __shared__ __attribute__((aligned(16))) double Craw[2][N][4];   // raw column panel: Craw[r][j] = A[r][K + j]
  const int64_t p = blockIdx.x;
  const int n = (int)(patch_ptr[p + 1] - patch_ptr[p]);
  const int ld = (n + 1) & ~1;
  double* S = inv + inv_ptr[p];
  // wave index as an SGPR value: the conditions below become scalar branches
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int lm = lane & 15, lk = lane >> 4;
  static_assert(NT % 2 == 0, "tile columns 2 b + wc < NT for every b < NT / 2");
  inv_d4 acc[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int ti = 4 * a + wr, tj = 2 * b + wc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = 16 * ti + lk + 4 * g, c = 16 * tj + lm;
        acc[a][b][g] = (r < n && c < n) ? S[(int64_t)r * ld + c] : (r == c ? 1.0 : 0.0);
      }
    }
  __syncthreads();  // all loads done before anyone stores (the result goes back in place, in another layout)
  bool bad = false;
#ifdef ALFI_INVERT_TIMING
  long long tacc[6] = {0, 0, 0, 0, 0, 0};
  long long tprev = (long long)__builtin_readcyclecounter();
  const long long tstart = tprev;
#endif
  // the last tile row of a wave may lie outside the matrix (NT = 10: waves 4-7 hold two tile rows, not three): its panel reads
  // are redirected to a tile row that exists -- straight-line code, the reads of a step leave together -- and only its MFMAs and
  // stores are skipped
  const bool last_row_in = 4 * (NA - 1) + wr < NT;               // wave-uniform
  static_for<0, NT>([&](auto TK) __attribute__((always_inline)) {
    constexpr int tk = decltype(TK)::value;   // the pivot tile: a compile-time constant, so is every accumulator index below
    if (16 * tk >= n) return;                 // uniform: the remaining pivots are identity padding
    const bool own_r = (tk & 3) == wr, own_c = (tk & 1) == wc;     // wave-uniform
    constexpr int ar = tk >> 2, bc = tk >> 1; // my tile index of tile row / column tk (when I own it)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int s = 4 * tk + q;
      if (4 * s >= n) continue;               // uniform
      const int buf = s & 1;
      const bool colgrp = (lm >> 2) == q;     // lanes holding the columns K of a tile of tile column tk (as C/D columns) and
                                              // the rows K of a tile of tile row tk (as A-operand rows)
      // ---- raw panels -> LDS
      if (own_r) {
#pragma unroll
        for (int b = 0; b < NB; ++b) Rraw[buf][16 * (2 * b + wc) + lm][lk] = acc[ar][b][q];
      }
      if (own_c && colgrp) {
#pragma unroll
        for (int a = 0; a < NA; ++a)
          if (a + 1 < NA || last_row_in) {
#pragma unroll
            for (int g = 0; g < 4; ++g) Craw[buf][16 * (4 * a + wr) + lk + 4 * g][lm & 3] = acc[a][bc][g];
          }
      }
      __syncthreads();
      ALFI_INV_PH(0);
      // ---- every panel read of the step, requested together: the pivot block (16 doubles side by side, the same for all
      //      lanes), the pivot rows of my tile columns, the pivot columns of my tile rows
      const inv_d4* Rv = reinterpret_cast<const inv_d4*>(&Rraw[buf][0][0]);
      const inv_d4* Cv = reinterpret_cast<const inv_d4*>(&Craw[buf][0][0]);
      inv_d4 dc[4], xb[NB], xa[NA];
#pragma unroll
      for (int j = 0; j < 4; ++j) dc[j] = Rv[4 * s + j];
#pragma unroll
      for (int b = 0; b < NB; ++b) xb[b] = Rv[16 * (2 * b + wc) + lm];
#pragma unroll
      for (int a = 0; a < NA; ++a) xa[a] = Cv[16 * ((a + 1 < NA || last_row_in) ? 4 * a + wr : wr) + lm];
      // ---- LU of the pivot block, redundantly on every lane (no second barrier)
      double d[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) d[i][j] = dc[j][i];
      Lu4 f;
      lu4(d, f, bad);
#ifdef ALFI_INVERT_TIMING
      asm volatile("" : "+v"(f.i3), "+v"(f.s23));
#endif
      ALFI_INV_PH(1);
      // ---- operands of the rank-4 update  A <- A - sum_t c_t r_t^T  (the four scalar Gauss-Jordan steps, delayed), by
      //      forward substitution with the SAME rounded L, U entries the pivots were computed with.  (Forming L^-1 and
      //      diag(U)^-1 U explicitly and taking each operand as a 4-term dot product is a third of the FP64 work, and fails the
      //      residual probe exactly like an explicit D^-1 does: 1.5e-4 on the [P2+FB]^3 test patches.  The second pivot of a
      //      node's gamma b b^T + nu K block is a cancellation of O(gamma) terms down to O(nu); only substitutions that repeat
      //      the elimination's own roundings stay consistent with it.)
      const double isel = sel4(lk, f.i0, f.i1, f.i2, f.i3);
      double bop[NB], aop[NA], vop[NA];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int tj = 2 * b + wc;
        const double x0 = xb[b][0], x1 = xb[b][1], x2 = xb[b][2], x3 = xb[b][3];
        const double y1 = __builtin_fma(-f.l10, x0, x1);
        const double y2 = __builtin_fma(-f.l21, y1, __builtin_fma(-f.l20, x0, x2));
        const double y3 = __builtin_fma(-f.l32, y2, __builtin_fma(-f.l31, y1, __builtin_fma(-f.l30, x0, x3)));
        const double rv = sel4(lk, x0, y1, y2, y3) * isel;               // r_lk = (L^-1 x)_lk / U_lk,lk
        bop[b] = (tj == tk && colgrp) ? 0.0 : rv;                        // columns K: left to the fix-up
        if (own_r) {
          // rows K <- D^-1 A[K, :] (columns outside K), the back substitution on the same y, straight into register q of the
          // tile: the update below leaves these rows alone (their A operand is zero)
          const double r0 = x0 * f.i0, r1 = y1 * f.i1, r2 = y2 * f.i2, r3 = y3 * f.i3;
          const double z2 = __builtin_fma(-f.s23, r3, r2);
          const double z1 = __builtin_fma(-f.s13, r3, __builtin_fma(-f.s12, z2, r1));
          const double z0 = __builtin_fma(-f.s03, r3, __builtin_fma(-f.s02, z2, __builtin_fma(-f.s01, z1, r0)));
          const double zs = sel4(lk, z0, z1, z2, r3);
          acc[ar][b][q] = (tj == tk && colgrp) ? acc[ar][b][q] : zs;
        }
      }
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const int ti = 4 * a + wr;
        const double x[4] = {xa[a][0], xa[a][1], xa[a][2], xa[a][3]};
        double c[4];
        col_panel(f, x, c);
        const double cv = sel4(lk, c[0], c[1], c[2], c[3]);
        aop[a] = (ti == tk && colgrp) ? 0.0 : cv;                        // rows K: left to the fix-up
        vop[a] = cv * isel;                                              // (A[:, K] U^-1)[row][lk]
      }
#ifdef ALFI_INVERT_TIMING
#pragma unroll
      for (int a = 0; a < NA; ++a) asm volatile("" : "+v"(aop[a]), "+v"(vop[a]));
#pragma unroll
      for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(bop[b]));
#endif
      ALFI_INV_PH(2);
#pragma unroll
      for (int a = 0; a < NA; ++a)
        if (a + 1 < NA || last_row_in) {
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-aop[a], bop[b], acc[a][b], 0, 0, 0);
        }
      ALFI_INV_PH(3);
      // ---- columns K <- - A[:, K] D^-1 = - (A[:, K] U^-1) L^-1 (rows outside K): one more MFMA per tile of tile column tk,
      //      A-operand v = A[:, K] U^-1 (the c_t scaled by 1 / U_tt), B-operand - L^-1 (unit lower triangular, explicit: 6
      //      entries) placed on the columns K of the tile.  A[K, K] <- D^-1 = U^-1 L^-1 on the 16 lanes that hold it.
      //      (Triangular solves per ENTRY instead -- 80 dependent chains per step on a quarter of the lanes -- made the
      //      kernel slower than the rank-1 register kernel it replaces: 22 against 18 ms at config 4's level 2.)
      if (own_c) {
        const int cq = lm & 3;
        const double m10 = -f.l10, m21 = -f.l21, m32 = -f.l32;           // L^-1 (only here: the final columns K)
        const double m20 = __builtin_fma(f.l21, f.l10, -f.l20), m31 = __builtin_fma(f.l32, f.l21, -f.l31);
        const double m30 = __builtin_fma(-m32, f.l20, __builtin_fma(-m31, f.l10, -f.l30));
        const double li0 = sel4(lk, 1.0, m10, m20, m30), li1 = sel4(lk, 0.0, 1.0, m21, m31), li2 = sel4(lk, 0.0, 0.0, 1.0, m32),
                     li3 = lk == 3 ? 1.0 : 0.0;
        const double b2 = colgrp ? -sel4(cq, li0, li1, li2, li3) : 0.0;
        double dkk = 0.0;                                                 // D^-1[lk][cq]
        {
          const double x[4] = {lk == 0 ? 1.0 : 0.0, lk == 1 ? 1.0 : 0.0, lk == 2 ? 1.0 : 0.0, lk == 3 ? 1.0 : 0.0};
          double w[4];
          times_dinv(f, x, w);
          dkk = sel4(cq, w[0], w[1], w[2], w[3]);
        }
#pragma unroll
        for (int a = 0; a < NA; ++a)
          if (a + 1 < NA || last_row_in) {
            const inv_d4 zero = {0.0, 0.0, 0.0, 0.0};
            const inv_d4 t = __builtin_amdgcn_mfma_f64_16x16x4f64(vop[a], b2, zero, 0, 0, 0);
            if (colgrp) {
              const bool pivot_tile = (4 * a + wr) == tk;
#pragma unroll
              for (int g = 0; g < 4; ++g) acc[a][bc][g] = (pivot_tile && g == q) ? dkk : t[g];   // rows K of the pivot tile: g == q
            }
          }
      }
      ALFI_INV_PH(4);
      ALFI_INV_PH(5);
    }
  });
  if (bad && threadIdx.x == 0) atomicExch(status, 1);
#ifdef ALFI_INVERT_TIMING
  if (g_invert_phases_dev && lane == 0 && p < 4096) {
    long long* o = g_invert_phases_dev + (p * 8 + wave) * 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = tacc[i];
    o[6] = (long long)__builtin_readcyclecounter() - tstart;
    o[7] = n;
  }
#endif
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int ti = 4 * a + wr, tj = 2 * b + wc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = 16 * ti + lk + 4 * g, c = 16 * tj + lm;
        // r == n < ld: the padding row; it stayed zero through the elimination and must be stored (the apply reads row pairs)
        if (ti < NT && tj < NT && r < ld && c < n) S[patch_inv_index(r, c, n, ld)] = (r < n) ? acc[a][b][g] : 0.0;
      }
    }
}

// Round 4 built this elimination twice more with LOOK-AHEAD -- the tiles of the next pivot's tile row / column updated first,
// the next panels written, the bulk of the MFMAs issued just before the barrier so that they drain under the next block's LU;
// once with ONE wave factoring the next pivot block and publishing the 28 factors, once with the LU redundant after the
// barrier -- bitwise the same inverses, and SLOWER: 106.4 and 105.1 ms against 98.9 ms on config 4's finest level (13.30
// against 12.65 ms at 24 389 patches, same box).  The PMC counters of this kernel (profiles/r04_pmc_patch_invert.txt: 45 % of
// the wave cycles in s_waitcnt / s_barrier, 29 % issue stalls, 287 VALU + 113 SALU + 12 MFMA instructions per wave and block
// step) say why reordering inside a step does not help: the step is a chain of dependent latencies (panel -> LDS -> barrier ->
// LU -> operands -> MFMA -> fix-up -> panel) that all eight waves walk in lockstep, and no wave of another patch fits beside
// them (232 VGPRs).  Both variants were removed again (CHANGELOG.md).

}  // namespace

#ifdef ALFI_INVERT_TIMING
static long long* g_invert_phases = nullptr;
extern "C" int64_t alfi_debug_invert_phases(int64_t* out) {     // out: 4096 x 8 waves x 8 (6 phase sums, total, n)
  if (!g_invert_phases) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpy(out, g_invert_phases, sizeof(long long) * 4096 * 64, hipMemcpyDeviceToHost) != hipSuccess) return -3;
  return 4096;
}
#endif

// in-place inversion (row-major n x ld in, row-piece layout out) of patches with 33 .. 160 dofs on the matrix cores;
// returns 1 if the sizes are handled here, 0 if the caller should use the register kernel
int launch_patch_invert_mfma(alfi_ctx* ctx, int64_t npatch, int max_np, const int64_t* patch_ptr, const int64_t* inv_ptr,
                             double* inv, int* status, int* handled) {
  static const bool allow = !(getenv("ALFI_INVERT_MFMA") && atoi(getenv("ALFI_INVERT_MFMA")) == 0);
  *handled = 0;
  // up to 112 dofs the rank-1 register kernel (7 x 7 tiles, two workgroups per CU) is the faster one -- measured at
  // [P1+FB]^3's 111 dofs: 5.5 against 9.8 ms for 35 937 patches; ALFI_INVERT_MFMA=2 sends those sizes here too (tests)
  static const bool all_sizes = getenv("ALFI_INVERT_MFMA") && atoi(getenv("ALFI_INVERT_MFMA")) == 2;
  if (!allow || max_np <= 32 || max_np > 160 || (max_np <= 112 && !all_sizes)) return 0;
#ifdef ALFI_INVERT_TIMING
  if (!g_invert_phases) {
    ALFI_HIP_CHECK(ctx, hipMalloc(&g_invert_phases, sizeof(long long) * 4096 * 64));
    ALFI_HIP_CHECK(ctx, hipMemset(g_invert_phases, 0, sizeof(long long) * 4096 * 64));
    ALFI_HIP_CHECK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_invert_phases_dev), &g_invert_phases, sizeof(long long*)));
  }
#endif
  dim3 grid((unsigned)npatch), block(512);
#define ALFI_INV(NTV) \
  hipLaunchKernelGGL(patch_invert_mfma_kernel<NTV>, grid, block, 0, ctx->stream, patch_ptr, inv_ptr, inv, status)
#ifdef ALFI_INVERT_DEV_NT10            // development builds: one instantiation (the file takes minutes per size)
  if (max_np <= 128) return 0;
  ALFI_INV(10);
#else
  if (max_np <= 64) ALFI_INV(4);
  else if (max_np <= 96) ALFI_INV(6);
  else if (max_np <= 128) ALFI_INV(8);
  else ALFI_INV(10);
#endif
#undef ALFI_INV
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  *handled = 1;
  return 0;
}
