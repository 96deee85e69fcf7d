// Operator refresh ON THE DEVICE: the level operator of a Newton step,
//
//     A(w) = nu K + gamma D + adv N(w),    N(w)[u, v] = ((w . grad) u + (u . grad) w, v)           (alfi/solver.py:565-568)
//
// written straight into the lane-major BSR the SpMV and the patch gather read, from the state w resident in HBM.  In the
// reference PatchPC.update recomputes the element tensors and re-assembles the patch operators inside PCPATCH on every Newton
// step (`precompute_element_tensors`, `save_operators`: alfi/solver.py:320, 325; event PCPatchComputeOp, driver.py:80).
//
// CELL-CENTRIC (round 5).  Two passes, no atomics, bitwise reproducible:
//   (1) element_cell_kernel: a LANE per cell forms the cell's element matrix block by block -- viscous, grad-div and advection
//       terms from the reference-cell tensors S, bI, T1 (the arithmetic of csrc/host_assemble.cpp:element_matrix) -- with the
//       cell's state, gradients and volume in registers.  All 64 lanes of a wave work on the same block (a, b) at the same
//       time, so every tensor entry is a wave-uniform operand that arrives through the scalar cache (s_load) and feeds D FMAs;
//       the d x d blocks go to a scratch array E[cell][a][b][d d].  (Rounds 3-4 gathered per BSR block: one thread walked its
//       contributor list and re-derived w . g and both tensor contractions for every (cell, a, b), with per-lane addresses into
//       the 175-KB tensor -- 112 divergent 8-byte loads per 392 FMAs: 67 ms for config 4's finest level, 0.05 of the HBM peak.)
//   (2) element_gather_kernel: one thread per BSR block adds the blocks of its contributing (cell, a, b) -- the fixed-order
//       lists of alfi_host_contributors --, turns Dirichlet rows / columns into identity (firedrake.assemble(a, bcs=...)) and
//       writes the nine value planes coalesced.
// The same cell kernel in its second mode multiplies the element matrix with the cell's entries of a vector instead of storing
// it (alfi_level_assemble_mult: the nonlinear residual needs A(u) u once -- no value array is formed), and the SUPG terms
// are added to E by their own cell kernel before the one gather.
#include <algorithm>
#include <cstdlib>
#include "common.h"

namespace {

// 16-byte accesses to 8-byte aligned addresses (a d x d block of E starts at a multiple of 8 d d bytes): global memory
// instructions need dword alignment only
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

template <int BB>
__device__ inline void store_block(double* dst, const double (&e)[BB]) {
#pragma unroll
  for (int t = 0; t + 1 < BB; t += 2) {
    d2u v;
    v.x = e[t];
    v.y = e[t + 1];
    *(d2u*)(dst + t) = v;
  }
  if (BB & 1) dst[BB - 1] = e[BB - 1];
}
template <int BB>
__device__ inline void load_block(const double* src, double (&e)[BB]) {
#pragma unroll
  for (int t = 0; t + 1 < BB; t += 2) {
    const d2u v = *(const d2u*)(src + t);
    e[t] = v.x;
    e[t + 1] = v.y;
  }
  if (BB & 1) e[BB - 1] = src[BB - 1];
}

// Element table, per (a, b) TS = 2 (d+1) nloc + (d+1)^2 doubles (alfi_level_set_assembly builds it from S, T1):
//   [i * nloc + k]                      T1[k, i, b, a]     ((w . grad) u: the state enters through w_k . g_i)
//   [(d+1) nloc + i * nloc + k]         T1[b, i, k, a]     ((u . grad) w: the state enters through w_k^cc g_i^dd)
//   [2 (d+1) nloc + i * (d+1) + j]      S[a, b, i, j]      (avg d_i phi_a d_j phi_b)
// MODE 0: E[slot][a][b][cc d + dd] = the block;  MODE 1: Fe[slot][a][cc] = sum_b block . x_b (x from LDS, a column per lane).
template <int D, int NLOC, int MODE>
__global__ __launch_bounds__(64) void element_cell_kernel(int64_t c0, int64_t c1, const int32_t* __restrict__ cell_nodes,
                                                           const double* __restrict__ grad, const double* __restrict__ vol,
                                                           const double* __restrict__ etab, const double* __restrict__ bItab,
                                                           const double* __restrict__ U, const double* __restrict__ X, double nu,
                                                           double gamma, double gfull, double adv, double* __restrict__ out,
                                                           int elay) {
  constexpr int NV = D + 1, BB = D * D, TS = 2 * NV * NLOC + NV * NV;
  __shared__ double xs[MODE == 1 ? NLOC * D * 64 : 1];
  const int lane = threadIdx.x;
  const int64_t slot = (int64_t)blockIdx.x * 64 + lane;
  const bool valid = c0 + slot < c1;
  const int64_t cell = valid ? c0 + slot : c1 - 1;     // (idle lanes of the last wave repeat the last cell and store nothing)
  const bool do_adv = adv != 0.0;
  double g[NV][D];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int x = 0; x < D; ++x) g[i][x] = grad[(cell * NV + i) * D + x];
  const double vc = vol[cell];
  const int32_t* cn = cell_nodes + cell * NLOC;
  double w[NLOC][D];
#pragma unroll
  for (int k = 0; k < NLOC; ++k) {
    const int64_t node = cn[k];
#pragma unroll
    for (int x = 0; x < D; ++x) {
      w[k][x] = do_adv ? U[node * D + x] : 0.0;
      if (MODE == 1) xs[(k * D + x) * 64 + lane] = X[node * D + x];
    }
  }
  const double fa = adv * vc, fn = nu * vc, fg = gfull * vc, fd = gamma * vc;
#pragma nounroll
  for (int a = 0; a < NLOC; ++a) {
    double ba[D];                       // gamma vol (1 / vol) int d_cc phi_a
#pragma unroll
    for (int x = 0; x < D; ++x) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < NV; ++i) s = __builtin_fma(g[i][x], bItab[a * NV + i], s);
      ba[x] = fd * s;
    }
    double fe[D];
#pragma unroll
    for (int x = 0; x < D; ++x) fe[x] = 0.0;
#pragma nounroll
    for (int b = 0; b < NLOC; ++b) {
      const double* __restrict__ t = etab + (size_t)(a * NLOC + b) * TS;
      double e[BB];
      {
        // H[p][q] = int d_p phi_a d_q phi_b = sum_ij S_ab[i][j] g_i^p g_j^q, in two steps
        double M1[NV][D];
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
          for (int q = 0; q < D; ++q) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < NV; ++j) s = __builtin_fma(t[2 * NV * NLOC + i * NV + j], g[j][q], s);
            M1[i][q] = s;
          }
        double H[D][D], gab = 0.0;
#pragma unroll
        for (int p = 0; p < D; ++p)
#pragma unroll
          for (int q = 0; q < D; ++q) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < NV; ++i) s = __builtin_fma(g[i][p], M1[i][q], s);
            H[p][q] = s;
            if (p == q) gab += s;
          }
        double bb[D];
#pragma unroll
        for (int x = 0; x < D; ++x) {
          double s = 0.0;
#pragma unroll
          for (int i = 0; i < NV; ++i) s = __builtin_fma(g[i][x], bItab[b * NV + i], s);
          bb[x] = s;
        }
        // K_(a,cc),(b,dd) = delta_{cc,dd} G_ab + int d_dd phi_a d_cc phi_b;  (div, div) = int d_cc phi_a d_dd phi_b;
        // cell-averaged grad-div = vol bvec_a^cc bvec_b^dd                                  (host_assemble.cpp:119-139)
#pragma unroll
        for (int cc = 0; cc < D; ++cc)
#pragma unroll
          for (int dd = 0; dd < D; ++dd) {
            double v = fn * (H[dd][cc] + (cc == dd ? gab : 0.0));
            v = __builtin_fma(fg, H[cc][dd], v);
            e[cc * D + dd] = __builtin_fma(ba[cc], bb[dd], v);
          }
      }
      if (do_adv) {
        double R[NV][D], S2[NV][D];
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
          for (int x = 0; x < D; ++x) R[i][x] = S2[i][x] = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
          for (int k = 0; k < NLOC; ++k) {
            const double ta = t[i * NLOC + k], tb = t[NV * NLOC + i * NLOC + k];
#pragma unroll
            for (int x = 0; x < D; ++x) {
              R[i][x] = __builtin_fma(ta, w[k][x], R[i][x]);
              S2[i][x] = __builtin_fma(tb, w[k][x], S2[i][x]);
            }
          }
        double t1 = 0.0;                 // sum_{k,i} (w_k . g_i) T1[k,i,b,a]
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
          for (int x = 0; x < D; ++x) t1 = __builtin_fma(g[i][x], R[i][x], t1);
#pragma unroll
        for (int cc = 0; cc < D; ++cc)
#pragma unroll
          for (int dd = 0; dd < D; ++dd) {
            double v = cc == dd ? t1 : 0.0;
#pragma unroll
            for (int i = 0; i < NV; ++i) v = __builtin_fma(S2[i][cc], g[i][dd], v);
            e[cc * D + dd] = __builtin_fma(fa, v, e[cc * D + dd]);
          }
      }
      if (MODE == 0) {
        // elay: the blocks (a, b) of the wave's 64 cells side by side (whole lines per store group) instead of a cell's blocks
        if (valid)
          store_block<BB>(out + (elay ? (((int64_t)blockIdx.x * (NLOC * NLOC) + a * NLOC + b) * 64 + lane) * BB
                                      : ((slot * NLOC + a) * NLOC + b) * BB), e);
      } else {
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
          const double xv = xs[(b * D + dd) * 64 + lane];
#pragma unroll
          for (int cc = 0; cc < D; ++cc) fe[cc] = __builtin_fma(e[cc * D + dd], xv, fe[cc]);
        }
      }
    }
    if (MODE == 1 && valid) {
#pragma unroll
      for (int cc = 0; cc < D; ++cc) out[(slot * NLOC + a) * D + cc] = fe[cc];
    }
  }
}

// bits 0 .. D-1: Dirichlet flags of the block's row dofs, bits 3 .. 3+D-1: of its column dofs, bit 6: diagonal block
template <int D>
__global__ __launch_bounds__(256) void bc_code_kernel(int64_t nnzb, int nloc, const int64_t* __restrict__ cptr,
                                                       const int32_t* __restrict__ ccell, const uint16_t* __restrict__ cba,
                                                       const int32_t* __restrict__ cell_nodes, const uint8_t* __restrict__ bc_mask,
                                                       uint8_t* __restrict__ code) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= nnzb) return;
  const int64_t q0 = cptr[k];
  const int32_t cell = ccell[q0];
  const int ba = cba[q0];
  const int64_t rnode = cell_nodes[(int64_t)cell * nloc + ba % nloc], cnode = cell_nodes[(int64_t)cell * nloc + ba / nloc];
  unsigned c = rnode == cnode ? 64u : 0u;
#pragma unroll
  for (int x = 0; x < D; ++x) {
    if (bc_mask[rnode * D + x]) c |= 1u << x;
    if (bc_mask[cnode * D + x]) c |= 8u << x;
  }
  code[k] = (uint8_t)c;
}

// vals (lane-major planes of the level operator) = / += the element blocks of the cells [c0, c1) in E, every block adding its
// contributors in list order; then (apply_bc) identity on Dirichlet rows / columns
template <int D>
__global__ __launch_bounds__(256) void element_gather_kernel(int64_t nnzb, int nloc, int64_t c0, int64_t c1,
                                                              const int64_t* __restrict__ cptr, const int32_t* __restrict__ ccell,
                                                              const uint16_t* __restrict__ cba, const double* __restrict__ E,
                                                              const uint8_t* __restrict__ code, int accumulate, int apply_bc,
                                                              double* __restrict__ vals, int elay) {
  constexpr int BB = D * D;
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= nnzb) return;
  // (accumulate: continue from the value in place, so that batches of cells add up in the order of one pass over the list)
  double acc[BB];
#pragma unroll
  for (int t = 0; t < BB; ++t) acc[t] = accumulate ? vals[bsr_val_index(1, k, t, BB)] : 0.0;
  bool any = false;
  const int64_t q1 = cptr[k + 1];
  for (int64_t q = cptr[k]; q < q1; ++q) {
    const int64_t cell = ccell[q];
    if (cell < c0 || cell >= c1) continue;
    any = true;
    const unsigned ba = cba[q], b = ba / (unsigned)nloc, a = ba - b * (unsigned)nloc;
    double e[BB];
    const int64_t cs = cell - c0;
    load_block<BB>(E + (elay ? (((cs >> 6) * (nloc * nloc) + a * nloc + b) * 64 + (cs & 63)) * BB : ((cs * nloc + a) * nloc + b) * BB), e);
#pragma unroll
    for (int t = 0; t < BB; ++t) acc[t] += e[t];
  }
  if (accumulate && !any && !apply_bc) return;
  const unsigned c = apply_bc ? code[k] : 0u;
#pragma unroll
  for (int cc = 0; cc < D; ++cc)
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
      const int64_t at = bsr_val_index(1, k, cc * D + dd, BB);
      double v = acc[cc * D + dd];
      const bool rb = (c >> cc) & 1u, cb = (c >> (3 + dd)) & 1u;
      if (rb || cb) v = (rb && cb && (c & 64u) && cc == dd) ? 1.0 : 0.0;
      vals[at] = v;
    }
}

// F (the level's n dofs) = / += the element vectors Fe[cell][a][d]: node r through the contributor list of its diagonal block
// (which holds exactly the (cell, a, a) with node a of the cell == r), in list order
template <int D>
__global__ __launch_bounds__(256) void cell_vector_gather_kernel(int64_t nnode, int nloc, const int32_t* __restrict__ diag,
                                                                  const int64_t* __restrict__ cptr, const int32_t* __restrict__ ccell,
                                                                  const uint16_t* __restrict__ cba, const double* __restrict__ Fe,
                                                                  int add, double* __restrict__ F) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= nnode) return;
  const int64_t k = diag[r];
  double acc[D];
#pragma unroll
  for (int i = 0; i < D; ++i) acc[i] = add ? F[r * D + i] : 0.0;
  const int64_t q1 = cptr[k + 1];
  for (int64_t q = cptr[k]; q < q1; ++q) {
    const int64_t cell = ccell[q];
    const int a = cba[q] % nloc;
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] += Fe[(cell * nloc + a) * D + i];
  }
#pragma unroll
  for (int i = 0; i < D; ++i) F[r * D + i] = acc[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// SUPG stabilisation on the device (alfi/stabilisation.py:47-97 with the Shakib-Hughes-Johan coefficient,
// alfi/solver.py:204-234; the reference's production option `--stabilisation-type supg`, examples/generate_submission:18-20):
//     stabilisation_form = weight * beta * inner(Lu, dot(grad(v), u)) * dx(degree 2k),
//     Lu = -nu div(2 sym grad u) + (grad u) u,   beta = (4 u.u / h^2 + magic (4 nu / h^2)^2)^(-1/2)
// by quadrature (beta is not polynomial): residual contribution and its exact Newton linearisation, the same formulas as
// csrc/host_assemble.cpp:alfi_host_supg.
//
// The linearisation (supg_matrix_kernel below): with, per quadrature point q,
//     s_a = u . grad phi_a,   dL_bij = phi_b G_ij - nu H_b[i][j] + delta_ij (- nu lap_b + s_b),   wt = w_q vol weight,
// entry ((a, i), (b, j)) of the element matrix is  sum_q wt (b3 u_j phi_b L_i s_a + beta dL_bij s_a + beta L_i phi_b dphi_a/dx_j)
//     = sum_q SAW[a][q] C1[q][(b,i,j)]  +  sum_q GP[(a,j)][q] C2W[q][(b,i)]
// -- two small GEMMs over the quadrature points, (nloc x nq)(nq x nloc d^2) and (nloc d x nq)(nq x nloc d), on the FP64 matrix
// cores (v_mfma_f64_16x16x4: k = 4 points per step).  A workgroup of TWO waves per cell, eight points per chunk: lane (q, a) of
// wave w forms the physical gradient / Hessian of basis function a at point 4 w + q and its contributions to the state
// quantities (u, grad u, the second-order part of Lu), 15 lanes per point sum them, every lane then has beta, Lu, ... of ITS point
// in registers and writes ITS node's rows of the operands -- wt s_a, the nine C1[q][(a, i, j)], the three C2W[q][(a, i)], static
// indices throughout -- to LDS; then wave 0 runs GEMM 1 (8 accumulator tiles for [P2+FB]^3) and wave 1 GEMM 2 (9 tiles) over the
// chunk's two k-steps: one LDS read per operand tile and the MFMAs, nothing else.
// The result goes through LDS into the scratch of element blocks E, coalesced (added to what element_cell_kernel left there).
// (Rounds 3-4: one wave per cell, every lane accumulated 28 entries with two FMAs and four LDS reads per entry and point behind
// five barriers per POINT with 14-42 lanes busy: 7000 cycles per point, 197 ms for config 4's finest level.  All 17 tiles in one
// wave: the register allocator keeps four of them in VGPRs and moves them to the accumulation registers and back around every
// chunk, 272 moves, at one wave per SIMD: 58.7 ms.  Two waves with the B operands decoded per column in the consuming lane --
// index arithmetic, four look-ups and selects per tile and k-step: 40.0 ms.)
// ---------------------------------------------------------------------------------------------------------------------
typedef double supg_d4 __attribute__((ext_vector_type(4)));

// 1 / sqrt(x): v_rsq_f64 + two Newton steps (the division-and-square-root sequence of `1.0 / sqrt(x)` is 40 instructions)
__device__ inline double supg_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  return y;
}

// qtab: per point p (padded to a multiple of EIGHT with zero-weight copies of point 0) and local node a:
//       [phi | dphi (d+1) | d2phi, upper triangle (d+1)(d+2)/2], derivatives w.r.t. the barycentric coordinates
template <int D, int NLOC>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2))) void supg_matrix_kernel(int64_t c0, int64_t ncell, const int32_t* __restrict__ cell_nodes,
                                                           const double* __restrict__ grad, const double* __restrict__ vol,
                                                           const double* __restrict__ hcell, int nchunk,
                                                           const double* __restrict__ wq8, const double* __restrict__ qtab,
                                                           const double* __restrict__ U, double nu, double weight, double magic,
                                                           int add, double* __restrict__ E) {
  constexpr int NV = D + 1, DD = D * D, ND = NLOC * D, ND2 = NLOC * DD;
  constexpr int NT1 = (ND2 + 15) / 16, NR2 = (ND + 15) / 16;       // column tiles of GEMM 1, row = column tiles of GEMM 2
  constexpr int NA = NT1 > NR2 * NR2 ? NT1 : NR2 * NR2;            // accumulator tiles of a wave
  constexpr int NS = 2 * D + DD, NH2 = NV * (NV + 1) / 2, QT = 1 + NV + NH2;
  static_assert(NLOC <= 16 && NS <= 16, "a lane per (point, local node)");
  constexpr int LDS_CHUNK = 8 * NS * 16 + 2 * 128 * NR2 + 2 * 128 + 128 * NT1;
  constexpr int LDS_N = LDS_CHUNK > NLOC * ND2 ? LDS_CHUNK : NLOC * ND2;
  __shared__ double sm[LDS_N];
  double* CON = sm;                    // [8][NS][16]     contributions of basis function a to state quantity v at point p
  double* GPs = CON + 8 * NS * 16;     // [8][16 NR2]     physical gradients, (a, x) -> a D + x: the A operand of GEMM 2
  double* C2Ws = GPs + 128 * NR2;      // [8][16 NR2]     wt beta L_i phi_b at (b, i) -> b D + i: the B operand of GEMM 2
  double* ST = C2Ws + 128 * NR2;       // [8][16]         u | grad u | second-order part of Lu
  double* SAWT = ST + 128;             // [8][16]         wt s_a: the A operand of GEMM 1
  double* C1s = SAWT + 128;            // [8][16 NT1]     C1 at (b, i, j) -> (b D + i) D + j: the B operand of GEMM 1
                                       // (operand entries beyond the element's are zero: written once, before the loop)
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, q = lane >> 4, r = lane & 15, p = 4 * wv + q;
  const int64_t cell = c0 + blockIdx.x;
  if (cell >= ncell) return;
  const int aa = r < NLOC ? r : NLOC - 1;
  double g[NV][D];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int x = 0; x < D; ++x) g[i][x] = grad[(cell * NV + i) * D + x];
  double Ua[D];
  {
    const int64_t node = cell_nodes[cell * NLOC + aa];
#pragma unroll
    for (int x = 0; x < D; ++x) Ua[x] = r < NLOC ? U[node * D + x] : 0.0;
  }
  const double hc = hcell[cell], ih2 = 1.0 / (hc * hc), vw = vol[cell] * weight, vis = 4.0 * nu * ih2;
  for (int e = tid; e < LDS_CHUNK - 8 * NS * 16; e += 128) GPs[e] = 0.0;
  supg_d4 acc[NA];
#pragma unroll
  for (int t = 0; t < NA; ++t) acc[t] = supg_d4{0.0, 0.0, 0.0, 0.0};
  // tabulated values of this lane's (point, node) pair, requested one chunk ahead
  double qt[QT];
  {
    const double* src = qtab + ((size_t)p * NLOC + aa) * QT;
#pragma unroll
    for (int t = 0; t < QT; ++t) qt[t] = src[t];
  }
  __syncthreads();
#pragma nounroll
  for (int ch = 0; ch < nchunk; ++ch) {
    // ---- basis function aa at point 8 ch + p: physical gradient, Hessian G^T H G (M = H G first), contributions to the state
    double gp[D], hs[D][D], ph, la = 0.0;
    {
#pragma unroll
      for (int x = 0; x < D; ++x) {
        double sgp = 0.0;
#pragma unroll
        for (int m = 0; m < NV; ++m) sgp = __builtin_fma(qt[1 + m], g[m][x], sgp);
        gp[x] = sgp;
      }
      double M[NV][D];
#pragma unroll
      for (int m = 0; m < NV; ++m)
#pragma unroll
        for (int y = 0; y < D; ++y) {
          double sm_ = 0.0;
#pragma unroll
          for (int n = 0; n < NV; ++n) {
            const int lo = m < n ? m : n, hi = m < n ? n : m;
            sm_ = __builtin_fma(qt[1 + NV + lo * NV - lo * (lo - 1) / 2 + hi - lo], g[n][y], sm_);
          }
          M[m][y] = sm_;
        }
#pragma unroll
      for (int x = 0; x < D; ++x)
#pragma unroll
        for (int y = x; y < D; ++y) {
          double sh = 0.0;
#pragma unroll
          for (int m = 0; m < NV; ++m) sh = __builtin_fma(g[m][x], M[m][y], sh);
          hs[x][y] = hs[y][x] = sh;
          if (x == y) la += sh;
        }
      ph = qt[0];
      double* con = CON + (p * NS) * 16 + r;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        con[i * 16] = ph * Ua[i];
#pragma unroll
        for (int x = 0; x < D; ++x) con[(D + i * D + x) * 16] = gp[x] * Ua[i];
        double cl = la * Ua[i];                       // - nu (lap_a U_ai + sum_x H_a[i][x] U_ax)
#pragma unroll
        for (int x = 0; x < D; ++x) cl = __builtin_fma(hs[i][x], Ua[x], cl);
        con[(D + DD + i) * 16] = -nu * cl;
      }
      if (r < NLOC) {
#pragma unroll
        for (int x = 0; x < D; ++x) GPs[p * 16 * NR2 + r * D + x] = gp[x];
      }
    }
    if (ch + 1 < nchunk) {
      const double* src = qtab + ((size_t)(8 * (ch + 1) + p) * NLOC + aa) * QT;
#pragma unroll
      for (int t = 0; t < QT; ++t) qt[t] = src[t];
    }
    __syncthreads();
    // ---- state quantity r of point p: the sum over the basis functions
    if (r < NS) {
      const double* con = CON + (p * NS + r) * 16;
      double sv = 0.0;
#pragma unroll
      for (int a = 0; a < 16; ++a) sv += con[a];
      ST[p * 16 + r] = sv;
    }
    __syncthreads();
    // ---- every lane: the state at ITS point
    {
      double u[D], Gu[D][D], Lu[D], uu = 0.0;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        u[i] = ST[p * 16 + i];
#pragma unroll
        for (int x = 0; x < D; ++x) Gu[i][x] = ST[p * 16 + D + i * D + x];
      }
#pragma unroll
      for (int i = 0; i < D; ++i) {
        double t = ST[p * 16 + D + DD + i];
#pragma unroll
        for (int x = 0; x < D; ++x) t = __builtin_fma(u[x], Gu[i][x], t);
        Lu[i] = t;
        uu = __builtin_fma(u[i], u[i], uu);
      }
      const double beta = supg_rsqrt(__builtin_fma(4.0 * uu, ih2, magic * vis * vis));
      const double b3 = -4.0 * beta * beta * beta * ih2;
      const double wt = wq8[8 * ch + p] * vw;
      double saw = 0.0;
#pragma unroll
      for (int x = 0; x < D; ++x) saw = __builtin_fma(u[x], gp[x], saw);
      // this lane's node as COLUMN node b = r: C1[(b, i, j)] = phi_b (b3 u_j L_i + beta G_ij) + beta (- nu H_b[i][j] + delta_ij
      // (- nu lap_b + s_b)),  C2W[(b, i)] = wt beta L_i phi_b;  as ROW node a = r: wt s_a
      if (r < NLOC) {
        const double dgb = __builtin_fma(-nu, la, saw), nb = -nu * beta, wb = wt * beta * ph;
        SAWT[p * 16 + r] = wt * saw;
#pragma unroll
        for (int i = 0; i < D; ++i) {
#pragma unroll
          for (int j = 0; j < D; ++j) {
            const double wij = __builtin_fma(b3 * u[j], Lu[i], beta * Gu[i][j]);
            double t = nb * hs[i][j];
            if (i == j) t = __builtin_fma(beta, dgb, t);
            C1s[p * 16 * NT1 + (r * D + i) * D + j] = __builtin_fma(ph, wij, t);
          }
          C2Ws[p * 16 * NR2 + r * D + i] = wb * Lu[i];
        }
      }
    }
    __syncthreads();
    if (wv == 0) {
      // ---- GEMM 1: A = SAWT[a = lane & 15][k = point lane >> 4], B = C1s[k][column lane & 15 of tile t]
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int pp = 4 * ks + q;
        const double aop1 = SAWT[pp * 16 + r];
#pragma unroll
        for (int t = 0; t < NT1; ++t)
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop1, C1s[pp * 16 * NT1 + 16 * t + r], acc[t], 0, 0, 0);
      }
    } else {
      // ---- GEMM 2: rows (a, j) of GP, columns (b, i) of C2W
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int pp = 4 * ks + q;
        double aop2[NR2], bop2[NR2];
#pragma unroll
        for (int tr = 0; tr < NR2; ++tr) {
          aop2[tr] = GPs[pp * 16 * NR2 + 16 * tr + r];
          bop2[tr] = C2Ws[pp * 16 * NR2 + 16 * tr + r];
        }
#pragma unroll
        for (int tr = 0; tr < NR2; ++tr)
#pragma unroll
          for (int tc = 0; tc < NR2; ++tc)
            acc[tr * NR2 + tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop2[tr], bop2[tc], acc[tr * NR2 + tc], 0, 0, 0);
      }
    }
    __syncthreads();     // the next chunk overwrites the tables
  }
  // ---- the element matrix in E's layout, E[a][b][i D + j] = EB[a ND2 + (b D + i) D + j], through LDS.  Results of an MFMA:
  //      column = lane & 15, rows (lane >> 4) + 4 reg
  double* EB = sm;
  if (wv == 0) {
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
      const int col = 16 * t + r;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int a = q + 4 * reg;
        if (a < NLOC && col < ND2) EB[a * ND2 + col] = acc[t][reg];
      }
    }
  }
  __syncthreads();
  if (wv == 1) {
#pragma unroll
    for (int tr = 0; tr < NR2; ++tr)
#pragma unroll
      for (int tc = 0; tc < NR2; ++tc) {
        const int c2 = 16 * tc + r;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int r2 = 16 * tr + q + 4 * reg;
          if (r2 < ND && c2 < ND) {
            const int a = r2 / D, j = r2 - a * D;
            EB[a * ND2 + c2 * D + j] += acc[tr * NR2 + tc][reg];
          }
        }
      }
  }
  __syncthreads();
  double* Ec = E + (int64_t)blockIdx.x * (NLOC * ND2);
  for (int e = tid; e < NLOC * ND2; e += 128) Ec[e] = add ? Ec[e] + EB[e] : EB[e];
}

// SUPG residual only (every Newton step evaluates it once more than it refreshes the operator): a LANE per cell, everything in
// registers -- per quadrature point the physical gradients / Hessians of the basis functions one after the other (tabulated
// values are wave-uniform scalar operands), the state quantities accumulated on the way, then the element residual
// F_(a,i) += wt beta Lu_i (u . grad phi_a).  (Round 4: the wave-per-cell kernel with the matrix part compiled out took 99.7 ms
// for config 4's finest level -- five barriers per point with a handful of lanes busy between them.)
template <int D, int NLOC>
__global__ __launch_bounds__(64) void supg_residual_cell_kernel(int64_t ncell, const int32_t* __restrict__ cell_nodes,
                                                                 const double* __restrict__ grad, const double* __restrict__ vol,
                                                                 const double* __restrict__ hcell, int nq,
                                                                 const double* __restrict__ wq, const double* __restrict__ phi,
                                                                 const double* __restrict__ dphi, const double* __restrict__ d2phi,
                                                                 const double* __restrict__ U, double nu, double weight, double magic,
                                                                 double* __restrict__ Fe) {
  constexpr int NV = D + 1;
  const int64_t slot = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool valid = slot < ncell;
  const int64_t cell = valid ? slot : ncell - 1;
  double g[NV][D];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int x = 0; x < D; ++x) g[i][x] = grad[(cell * NV + i) * D + x];
  const int32_t* cn = cell_nodes + cell * NLOC;
  double Uk[NLOC][D], F[NLOC][D];
#pragma unroll
  for (int k = 0; k < NLOC; ++k) {
    const int64_t node = cn[k];
#pragma unroll
    for (int x = 0; x < D; ++x) {
      Uk[k][x] = U[node * D + x];
      F[k][x] = 0.0;
    }
  }
  const double hc = hcell[cell], h2 = hc * hc;
  const double vw = vol[cell] * weight;
  const double vis = 4.0 * nu / h2;
#pragma nounroll
  for (int q = 0; q < nq; ++q) {
    const double* __restrict__ ph = phi + (size_t)q * NLOC;
    const double* __restrict__ dp = dphi + (size_t)q * NLOC * NV;
    const double* __restrict__ hp = d2phi + (size_t)q * NLOC * NV * NV;
    double u[D], Gu[D][D], Ls[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      u[i] = Ls[i] = 0.0;
#pragma unroll
      for (int x = 0; x < D; ++x) Gu[i][x] = 0.0;
    }
#pragma unroll
    for (int a = 0; a < NLOC; ++a) {
      double gp[D];
#pragma unroll
      for (int x = 0; x < D; ++x) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) s = __builtin_fma(dp[a * NV + i], g[i][x], s);
        gp[x] = s;
      }
      // physical Hessian G^T H_a G (symmetric), M = H_a G first
      double M[NV][D];
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int y = 0; y < D; ++y) {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < NV; ++k) s = __builtin_fma(hp[(a * NV + i) * NV + k], g[k][y], s);
          M[i][y] = s;
        }
      double hs[D][D], la = 0.0;
#pragma unroll
      for (int x = 0; x < D; ++x)
#pragma unroll
        for (int y = x; y < D; ++y) {
          double s = 0.0;
#pragma unroll
          for (int i = 0; i < NV; ++i) s = __builtin_fma(g[i][x], M[i][y], s);
          hs[x][y] = hs[y][x] = s;
          if (x == y) la += s;
        }
      const double pa = ph[a];
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const double ui = Uk[a][i];
        u[i] = __builtin_fma(pa, ui, u[i]);
#pragma unroll
        for (int x = 0; x < D; ++x) Gu[i][x] = __builtin_fma(gp[x], ui, Gu[i][x]);
        Ls[i] = __builtin_fma(-nu * la, ui, Ls[i]);                                   // -nu Lap u_i
#pragma unroll
        for (int j = 0; j < D; ++j) Ls[j] = __builtin_fma(-nu * hs[j][i], ui, Ls[j]);   // -nu d_j div u
      }
    }
    double uu = 0.0, c[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double t = Ls[i];
#pragma unroll
      for (int x = 0; x < D; ++x) t = __builtin_fma(u[x], Gu[i][x], t);
      c[i] = t;
      uu = __builtin_fma(u[i], u[i], uu);
    }
    const double beta = 1.0 / sqrt(4.0 * uu / h2 + magic * vis * vis);
    const double wb = wq[q] * vw * beta;
#pragma unroll
    for (int i = 0; i < D; ++i) c[i] *= wb;
#pragma unroll
    for (int a = 0; a < NLOC; ++a) {
      double sa = 0.0;                   // u . grad phi_a (the gradient formed again: 12 FMAs against 3 registers kept per node)
#pragma unroll
      for (int x = 0; x < D; ++x) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) s = __builtin_fma(dp[a * NV + i], g[i][x], s);
        sa = __builtin_fma(u[x], s, sa);
      }
#pragma unroll
      for (int i = 0; i < D; ++i) F[a][i] = __builtin_fma(c[i], sa, F[a][i]);
    }
  }
  if (valid) {
#pragma unroll
    for (int a = 0; a < NLOC; ++a)
#pragma unroll
      for (int i = 0; i < D; ++i) Fe[(slot * NLOC + a) * D + i] = F[a][i];
  }
}

// Dirichlet rows / columns of the level operator -> identity (what firedrake.assemble(a, bcs=...) produces), after the terms
// have been summed; row node of a block from its first contributor
template <int D>
__global__ __launch_bounds__(256) void apply_bc_kernel(int64_t nnzb, int nloc, const int64_t* __restrict__ cptr,
                                                        const int32_t* __restrict__ ccell, const uint16_t* __restrict__ cba,
                                                        const int32_t* __restrict__ cell_nodes, const uint8_t* __restrict__ bc_mask,
                                                        double* __restrict__ vals) {
  constexpr int BB = D * D;
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= nnzb) return;
  const int64_t q0 = cptr[k];
  const int32_t cell = ccell[q0];
  const int ba = cba[q0];
  const int64_t rnode = cell_nodes[(int64_t)cell * nloc + ba % nloc], cnode = cell_nodes[(int64_t)cell * nloc + ba / nloc];
#pragma unroll
  for (int cc = 0; cc < D; ++cc)
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
      const bool rb = bc_mask[rnode * D + cc], cb = bc_mask[cnode * D + dd];
      if (rb || cb) vals[bsr_val_index(1, k, cc * D + dd, BB)] = (rb && cb && rnode == cnode && cc == dd) ? 1.0 : 0.0;
    }
}

// lane-major -> host layout (nnzb, d, d): alfi_level_get_values (tests, diagnostics)
__global__ void vals_from_lanes_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t total, int bb) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    dst[e] = src[bsr_val_index(1, e / bb, (int)(e % bb), bb)];
}


}  // namespace

// ---- launches ----------------------------------------------------------------------------------------------------------
#define ALFI_ELEMENT_DISPATCH(d, nloc, WHAT)                                   \
  do {                                                                         \
    if (d == 2 && nloc == 3) { WHAT(2, 3); }                                   \
    else if (d == 2 && nloc == 6) { WHAT(2, 6); }                              \
    else if (d == 2 && nloc == 10) { WHAT(2, 10); }                            \
    else if (d == 3 && nloc == 4) { WHAT(3, 4); }                              \
    else if (d == 3 && nloc == 8) { WHAT(3, 8); }                              \
    else if (d == 3 && nloc == 10) { WHAT(3, 10); }                            \
    else if (d == 3 && nloc == 14) { WHAT(3, 14); }                            \
    else if (d == 3 && nloc == 20) { WHAT(3, 20); }                            \
    else return alfi_set_error(ctx, ALFI_E_ARG, "no element kernel for %d nodes per cell in %d-D", nloc, d); \
  } while (0)

bool element_kernel_exists(int d, int nloc) {
  return (d == 2 && (nloc == 3 || nloc == 6 || nloc == 10)) || (d == 3 && (nloc == 4 || nloc == 8 || nloc == 10 || nloc == 14 || nloc == 20));
}

static int ensure_scratch(alfi_ctx* ctx, size_t bytes) {
  if (ctx->asm_scratch_bytes >= bytes) return 0;
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(ctx->asm_scratch);
  ctx->asm_scratch = nullptr;
  ctx->asm_scratch_bytes = 0;
  ALFI_HIP_CHECK(ctx, hipMalloc(&ctx->asm_scratch, bytes));
  ctx->asm_scratch_bytes = bytes;
  return 0;
}

static int ensure_bc_code(alfi_level* L) {
  alfi_ctx* ctx = L->ctx;
  AssemblyDev& S = L->asmb;
  if (S.bc_code) return 0;
  const int64_t nnzb = L->A.nnzb;
  ALFI_HIP_CHECK(ctx, hipMalloc((void**)&S.bc_code, (size_t)std::max<int64_t>(nnzb, 1)));
  dim3 grid((unsigned)((nnzb + 255) / 256)), block(256);
  const uint8_t* mask = S.bc_all ? S.bc_all : L->bc_mask;
  if (L->bs == 2)
    hipLaunchKernelGGL(bc_code_kernel<2>, grid, block, 0, ctx->stream, nnzb, S.nloc, S.cptr, S.ccell, S.cba, S.cell_nodes, mask, S.bc_code);
  else
    hipLaunchKernelGGL(bc_code_kernel<3>, grid, block, 0, ctx->stream, nnzb, S.nloc, S.cptr, S.ccell, S.cba, S.cell_nodes, mask, S.bc_code);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

// The level operator from its cells: vals = [vals +] sum over the contributing cells of
//     (nu K_e + gamma D_e + adv N_e(state))   (with_elements)   +   the linearised SUPG term (with_supg),
// Dirichlet rows / columns -> identity (apply_bc).  Cells in batches when the scratch of element blocks would exceed the ctx's
// limit (alfi_ctx_set_assembly_scratch; config 4's finest level: 14.9 GB in one batch).
int launch_operator_refresh(alfi_level* L, double nu, double gamma, double adv, const double* d_state, bool with_elements,
                            bool with_supg, double weight, double magic, bool accumulate, bool apply_bc, double* out_vals) {
  alfi_ctx* ctx = L->ctx;
  AssemblyDev& S = L->asmb;
  const int d = L->bs, nloc = S.nloc, ndof = nloc * d;
  const int64_t nnzb = L->A.nnzb;
  if (apply_bc) ALFI_CHECK(ensure_bc_code(L));
  const int64_t per_cell = (int64_t)ndof * ndof * 8;
  int64_t batch = std::max<int64_t>(1, ctx->asm_scratch_limit / per_cell);
  if (batch > S.ncell) batch = S.ncell;
  // element blocks of a wave's 64 cells interleaved by (a, b) -- unless the SUPG kernel, which writes a cell's matrix as one piece,
  // adds into the same scratch
#ifndef ALFI_ELAY
#define ALFI_ELAY 1
#endif
  const int elay = (ALFI_ELAY && !with_supg) ? 1 : 0;
  ALFI_CHECK(ensure_scratch(ctx, (size_t)(((batch + 63) / 64 * 64) * per_cell)));
  double* E = (double*)ctx->asm_scratch;
  const double gcell = S.full_div ? 0.0 : gamma, gfull = S.full_div ? gamma : 0.0;
  for (int64_t c0 = 0; c0 < S.ncell; c0 += batch) {
    const int64_t c1 = std::min<int64_t>(c0 + batch, S.ncell);
    if (with_elements) {
      dim3 grid((unsigned)((c1 - c0 + 63) / 64)), block(64);
#define ALFI_EL0(DV, NL)                                                                                                          \
  hipLaunchKernelGGL((element_cell_kernel<DV, NL, 0>), grid, block, 0, ctx->stream, c0, c1, S.cell_nodes, S.grad, S.vol, S.etab, \
                     S.bItab, d_state, (const double*)nullptr, nu, gcell, gfull, adv, E, elay)
      ALFI_ELEMENT_DISPATCH(d, nloc, ALFI_EL0);
#undef ALFI_EL0
      ALFI_HIP_CHECK(ctx, hipGetLastError());
    }
    if (with_supg) {
      dim3 grid((unsigned)(c1 - c0)), block(128);
      const int add = with_elements ? 1 : 0;
#define ALFI_SUPG_M(DV, NL)                                                                                                      \
  hipLaunchKernelGGL((supg_matrix_kernel<DV, NL>), grid, block, 0, ctx->stream, c0, c1, S.cell_nodes, S.grad, S.vol, S.hcell,   \
                     S.nq8 / 8, S.wq8, S.qtab, d_state, nu, weight, magic, add, E)
      if (d == 2 && nloc == 3) { ALFI_SUPG_M(2, 3); }
      else if (d == 2 && nloc == 6) { ALFI_SUPG_M(2, 6); }
      else if (d == 3 && nloc == 4) { ALFI_SUPG_M(3, 4); }
      else if (d == 3 && nloc == 8) { ALFI_SUPG_M(3, 8); }
      else if (d == 3 && nloc == 10) { ALFI_SUPG_M(3, 10); }
      else if (d == 3 && nloc == 14) { ALFI_SUPG_M(3, 14); }
      else return alfi_set_error(ctx, ALFI_E_ARG, "no SUPG kernel for %d nodes per cell in %d-D", nloc, d);
#undef ALFI_SUPG_M
      ALFI_HIP_CHECK(ctx, hipGetLastError());
    }
    dim3 g2((unsigned)((nnzb + 255) / 256)), b2(256);
    const int acc = (accumulate || c0 > 0) ? 1 : 0, bc = (apply_bc && c1 == S.ncell) ? 1 : 0;
    if (d == 2)
      hipLaunchKernelGGL(element_gather_kernel<2>, g2, b2, 0, ctx->stream, nnzb, nloc, c0, c1, S.cptr, S.ccell, S.cba, E, S.bc_code, acc, bc, out_vals, elay);
    else
      hipLaunchKernelGGL(element_gather_kernel<3>, g2, b2, 0, ctx->stream, nnzb, nloc, c0, c1, S.cptr, S.ccell, S.cba, E, S.bc_code, acc, bc, out_vals, elay);
    ALFI_HIP_CHECK(ctx, hipGetLastError());
  }
  return 0;
}

static int gather_cell_vectors(alfi_level* L, const double* Fe, int add, double* d_F) {
  alfi_ctx* ctx = L->ctx;
  const AssemblyDev& S = L->asmb;
  const int64_t nnode = L->A.nbrows;
  dim3 g3((unsigned)((nnode + 255) / 256)), b3(256);
  if (L->bs == 2)
    hipLaunchKernelGGL(cell_vector_gather_kernel<2>, g3, b3, 0, ctx->stream, nnode, S.nloc, S.diag, S.cptr, S.ccell, S.cba, Fe, add, d_F);
  else
    hipLaunchKernelGGL(cell_vector_gather_kernel<3>, g3, b3, 0, ctx->stream, nnode, S.nloc, S.diag, S.cptr, S.ccell, S.cba, Fe, add, d_F);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

// dy (the level's rows) = sum over the cells of (nu K_e + gamma D_e + adv N_e(state)) x_e, matrix-free: the element matrices are
// multiplied block by block as they are formed, the element vectors gathered per node in a fixed order
int launch_element_mult(alfi_level* L, double nu, double gamma, double adv, const double* d_state, const double* dx, double* dy) {
  alfi_ctx* ctx = L->ctx;
  AssemblyDev& S = L->asmb;
  const int d = L->bs, nloc = S.nloc;
  ALFI_CHECK(ensure_scratch(ctx, sizeof(double) * (size_t)(S.ncell * nloc * d)));
  double* Fe = (double*)ctx->asm_scratch;
  const double gcell = S.full_div ? 0.0 : gamma, gfull = S.full_div ? gamma : 0.0;
  dim3 grid((unsigned)((S.ncell + 63) / 64)), block(64);
#define ALFI_EL1(DV, NL)                                                                                                         \
  hipLaunchKernelGGL((element_cell_kernel<DV, NL, 1>), grid, block, 0, ctx->stream, (int64_t)0, S.ncell, S.cell_nodes, S.grad,  \
                     S.vol, S.etab, S.bItab, d_state, dx, nu, gcell, gfull, adv, Fe, 0)
  ALFI_ELEMENT_DISPATCH(d, nloc, ALFI_EL1);
#undef ALFI_EL1
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return gather_cell_vectors(L, Fe, 0, dy);
}

// d_F += the SUPG residual contribution about d_state
int launch_supg_residual(alfi_level* L, double nu, double weight, double magic, const double* d_state, double* d_F) {
  alfi_ctx* ctx = L->ctx;
  AssemblyDev& S = L->asmb;
  const int d = L->bs, nloc = S.nloc;
  ALFI_CHECK(ensure_scratch(ctx, sizeof(double) * (size_t)(S.ncell * nloc * d)));
  double* Fe = (double*)ctx->asm_scratch;
  dim3 grid((unsigned)((S.ncell + 63) / 64)), block(64);
#define ALFI_SR(DV, NL)                                                                                                          \
  hipLaunchKernelGGL((supg_residual_cell_kernel<DV, NL>), grid, block, 0, ctx->stream, S.ncell, S.cell_nodes, S.grad, S.vol,    \
                     S.hcell, S.nq, S.wq, S.phi, S.dphi, S.d2phi, d_state, nu, weight, magic, Fe)
  if (d == 2 && nloc == 3) { ALFI_SR(2, 3); }
  else if (d == 2 && nloc == 6) { ALFI_SR(2, 6); }
  else if (d == 3 && nloc == 4) { ALFI_SR(3, 4); }
  else if (d == 3 && nloc == 8) { ALFI_SR(3, 8); }
  else if (d == 3 && nloc == 10) { ALFI_SR(3, 10); }
  else if (d == 3 && nloc == 14) { ALFI_SR(3, 14); }
  else return alfi_set_error(ctx, ALFI_E_ARG, "no SUPG residual kernel for %d nodes per cell in %d-D", nloc, d);
#undef ALFI_SR
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return gather_cell_vectors(L, Fe, 1, d_F);
}

int launch_apply_bc(alfi_level* L) {
  alfi_ctx* ctx = L->ctx;
  const AssemblyDev& S = L->asmb;
  const int64_t nnzb = L->A.nnzb;
  dim3 grid((unsigned)((nnzb + 255) / 256)), block(256);
  if (L->bs == 2)
    hipLaunchKernelGGL(apply_bc_kernel<2>, grid, block, 0, ctx->stream, nnzb, S.nloc, S.cptr, S.ccell, S.cba, S.cell_nodes, S.bc_all ? S.bc_all : L->bc_mask, L->A.vals);
  else
    hipLaunchKernelGGL(apply_bc_kernel<3>, grid, block, 0, ctx->stream, nnzb, S.nloc, S.cptr, S.ccell, S.cba, S.cell_nodes, S.bc_all ? S.bc_all : L->bc_mask, L->A.vals);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

int launch_vals_from_lanes(alfi_ctx* ctx, const DevBSR& A, double* d_out) {
  const int bb = A.bs * A.bs;
  const int64_t total = A.nnzb * bb;
  if (total == 0) return 0;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(vals_from_lanes_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, A.vals, d_out, total, bb);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}
