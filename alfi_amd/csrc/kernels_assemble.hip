// Operator refresh ON THE DEVICE: the level operator of a Newton step,
//
//     A(w) = nu K + gamma D + adv N(w),    N(w)[u, v] = ((w . grad) u + (u . grad) w, v)           (alfi/solver.py:565-568)
//
// written straight into the lane-major BSR the SpMV and the patch gather read, from the state w resident in HBM.  In the
// reference PatchPC.update recomputes the element tensors and re-assembles the patch operators inside PCPATCH on every Newton
// step (`precompute_element_tensors`, `save_operators`: alfi/solver.py:320, 325; event PCPatchComputeOp, driver.py:80);
// here the host generator used to rediscretise every level and re-upload 10 GB per step.
//
// K (viscous) and D (grad-div) do not depend on the state: they are uploaded once, in the operator's own layout.  N(w) is
// assembled by GATHER, one thread per d x d block (r, c): the thread walks the block's contributor list -- the (cell, a, b)
// with r = node a, c = node b of the cell, in a fixed order built on the host (alfi_host_contributors) -- and forms, with the
// reference tensor T1[k, i, b, a] = avg(phi_k d_i phi_b phi_a) (barycentric derivative i),
//
//     N_(a cc),(b dd) = vol [ delta_{cc dd} sum_{k,i} (w_k . g_i) T1[k,i,b,a]  +  sum_i g_i^dd  sum_k T1[b,i,k,a] w_k^cc ] ,
//
// the same arithmetic as csrc/host_assemble.cpp:element_matrix.  No atomics: every entry is summed in the same order on
// every run, consecutive lanes own consecutive blocks, so the nine value planes are written coalesced.  Dirichlet rows and
// columns become identity in the same pass (firedrake.assemble(a, bcs=...)).
#include <algorithm>
#include <cstdlib>
#include "common.h"

namespace {

template <int D>
__global__ __launch_bounds__(256) void assemble_gather_kernel(int64_t nnzb, int nloc, const int64_t* __restrict__ cptr,
                                                               const int32_t* __restrict__ ccell,
                                                               const uint16_t* __restrict__ cba,
                                                               const int32_t* __restrict__ cell_nodes,
                                                               const double* __restrict__ grad, const double* __restrict__ vol,
                                                               const double* __restrict__ Ta, const double* __restrict__ Tb,
                                                               const double* __restrict__ Kv, const double* __restrict__ Dv,
                                                               const double* __restrict__ w, const uint8_t* __restrict__ bc_mask,
                                                               double nu, double gamma, double adv, int apply_bc,
                                                               double* __restrict__ out) {
  constexpr int NV = D + 1, BB = D * D;
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= nnzb) return;
  double acc[D][D];
#pragma unroll
  for (int cc = 0; cc < D; ++cc)
#pragma unroll
    for (int dd = 0; dd < D; ++dd) acc[cc][dd] = 0.0;
  const int64_t q0 = cptr[k], q1 = cptr[k + 1];
  int32_t rnode = 0, cnode = 0;
  {
    const int32_t cell = ccell[q0];
    const int ba = cba[q0];
    rnode = cell_nodes[(int64_t)cell * nloc + ba % nloc];
    cnode = cell_nodes[(int64_t)cell * nloc + ba / nloc];
  }
  if (adv != 0.0) {
    for (int64_t q = q0; q < q1; ++q) {
      const int32_t cell = ccell[q];
      const int ba = cba[q];
      const int32_t* cn = cell_nodes + (int64_t)cell * nloc;
      double g[NV][D];
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int x = 0; x < D; ++x) g[i][x] = grad[((int64_t)cell * NV + i) * D + x];
      const double* ta = Ta + (int64_t)ba * nloc * NV;      // [k][i] = T1[k, i, b, a]
      const double* tb = Tb + (int64_t)ba * nloc * NV;      // [i][k] = T1[b, i, k, a]
      double t1 = 0.0, s[NV][D];
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int cc = 0; cc < D; ++cc) s[i][cc] = 0.0;
      for (int kk = 0; kk < nloc; ++kk) {
        const int64_t node = cn[kk];
        double wv[D];
#pragma unroll
        for (int x = 0; x < D; ++x) wv[x] = w[node * D + x];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          double wg = 0.0;
#pragma unroll
          for (int x = 0; x < D; ++x) wg = __builtin_fma(wv[x], g[i][x], wg);
          t1 = __builtin_fma(wg, ta[kk * NV + i], t1);
          const double tbv = tb[i * nloc + kk];
#pragma unroll
          for (int cc = 0; cc < D; ++cc) s[i][cc] = __builtin_fma(tbv, wv[cc], s[i][cc]);
        }
      }
      const double vc = vol[cell];
#pragma unroll
      for (int cc = 0; cc < D; ++cc)
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
          double v = cc == dd ? t1 : 0.0;
#pragma unroll
          for (int i = 0; i < NV; ++i) v = __builtin_fma(s[i][cc], g[i][dd], v);
          acc[cc][dd] = __builtin_fma(vc, v, acc[cc][dd]);
        }
    }
  }
  // A = nu K + gamma D + adv N, Dirichlet rows / columns -> identity; lane-major planes (bsr_val_index, flat layout)
  uint8_t rb[D], cb[D];
#pragma unroll
  for (int x = 0; x < D; ++x) {
    rb[x] = apply_bc ? bc_mask[(int64_t)rnode * D + x] : 0;
    cb[x] = apply_bc ? bc_mask[(int64_t)cnode * D + x] : 0;
  }
#pragma unroll
  for (int cc = 0; cc < D; ++cc)
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
      const int64_t at = bsr_val_index(1, k, cc * D + dd, BB);
      double v = __builtin_fma(nu, Kv[at], __builtin_fma(gamma, Dv[at], adv * acc[cc][dd]));
      if (rb[cc] || cb[dd]) v = (rb[cc] && cb[dd] && rnode == cnode && cc == dd) ? 1.0 : 0.0;
      out[at] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// SUPG stabilisation on the device (alfi/stabilisation.py:47-97 with the Shakib-Hughes-Johan coefficient,
// alfi/solver.py:204-234; the reference's production option `--stabilisation-type supg`, examples/generate_submission:18-20):
//     stabilisation_form = weight * beta * inner(Lu, dot(grad(v), u)) * dx(degree 2k),
//     Lu = -nu div(2 sym grad u) + (grad u) u,   beta = (4 u.u / h^2 + magic (4 nu / h^2)^2)^(-1/2)
// by quadrature (beta is not polynomial): residual contribution and its exact Newton linearisation, the same formulas as
// csrc/host_assemble.cpp:alfi_host_supg.
//
// Two phases, no atomics.  (1) supg_cell_kernel: one wave per cell forms the element matrix (ndof x ndof, ndof = nloc d) and the
// element residual over the quadrature points: per point the physical gradients / Hessians of the basis (a lane per local
// node), the state quantities u, grad u, Lu (a lane per output), then every lane accumulates its ndof^2 / 64 entries in
// registers; results go to a scratch array (cells x ndof^2).  (2) supg_gather_kernel: one thread per BSR block adds the
// entries of its contributing cells in the fixed order of the contributor lists (the lists of the advection assembly);
// supg_residual_kernel does the same per node through the diagonal block's list.
// ---------------------------------------------------------------------------------------------------------------------
template <int D, int EPL, bool WANT>
__global__ __launch_bounds__(64) void supg_cell_kernel(int64_t cell0, int64_t ncell, int nloc, const int32_t* __restrict__ cell_nodes,
                                                        const double* __restrict__ grad, const double* __restrict__ vol,
                                                        const double* __restrict__ hcell, int nq, const double* __restrict__ wq,
                                                        const double* __restrict__ phi, const double* __restrict__ dphi,
                                                        const double* __restrict__ d2phi, const double* __restrict__ U, double nu,
                                                        double weight, double magic, int want_vals, double* __restrict__ Ae_out,
                                                        double* __restrict__ Fe_out) {
  constexpr int NV = D + 1;
  extern __shared__ double sm[];
  const int64_t cell = cell0 + blockIdx.x;
  if (cell >= ncell) return;
  const int lane = threadIdx.x;
  const int ndof = nloc * D;
  double* Uk = sm;                       // nloc * D
  double* gp = Uk + nloc * D;            // nloc * D     physical gradient of basis a
  double* hs = gp + nloc * D;            // nloc * D * D physical Hessian of basis a
  double* lap = hs + nloc * D * D;       // nloc
  double* ph = lap + nloc;               // nloc
  double* st = ph + nloc;                // u[D], Gu[D][D], Lu[D]
  double* saw = st + 2 * D + D * D;      // nloc          wt * (u . grad phi_a)
  double* c2w = saw + nloc;              // nloc * D      wt * beta * Lu_i * phi_b
  double* c1 = c2w + nloc * D;           // nloc * D * D  the (b, i, j) factor of the entry that multiplies saw_a (see below)
  const int32_t* cn = cell_nodes + cell * nloc;
  double g[NV][D];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int x = 0; x < D; ++x) g[i][x] = grad[(cell * NV + i) * D + x];
  for (int e = lane; e < ndof; e += 64) Uk[e] = U[(int64_t)cn[e / D] * D + e % D];
  const double hc = hcell[cell], h2 = hc * hc;
  const double vw = vol[cell] * weight;
  double acc[WANT ? EPL : 1];
#pragma unroll
  for (int t = 0; t < (WANT ? EPL : 1); ++t) acc[t] = 0.0;
  double facc = 0.0;                     // element residual entry `lane` (ndof <= 64)
  // entry e = (row (a, i), column (b, j)) of this lane's t-th entry, decomposed ONCE: the places of its four factors in the
  // per-point tables, a byte each (saw[a] | c1[(b D + i) D + j] | c2w[b D + i] | gp[a D + j]).  (Round 3 divided e by ndof and
  // by D inside the loop over the quadrature points: ~150 integer instructions per entry and point around 25 flops -- the
  // kernel took 0.9 s per refresh of config 4's finest level, half of a Newton step with SUPG.)
  uint32_t pk[WANT ? EPL : 1];
#pragma unroll
  for (int t = 0; t < (WANT ? EPL : 1); ++t) {
    const int e = lane + 64 * t;
    pk[t] = 0;
    if (e < ndof * ndof) {
      const int row = e / ndof, col = e % ndof;
      const int a = row / D, i = row % D, b = col / D, j = col % D;
      pk[t] = (uint32_t)a | ((uint32_t)((b * D + i) * D + j) << 8) | ((uint32_t)(b * D + i) << 16) | ((uint32_t)(a * D + j) << 24);
    }
  }
  __syncthreads();
  for (int q = 0; q < nq; ++q) {
    // ---- basis at the point: a lane per (local node a, direction y).  Physical gradient component y and column y of the
    //      physical Hessian G^T H_a G in two steps, M = H_a G[:, y] then G^T M: 28 FMAs on nloc D lanes.  (Round 3: a lane per
    //      node summed all 16 x 9 products ha_ik g_ix g_ky -- 290 multiply-adds on 14 of the 64 lanes, most of the time of a
    //      residual-only pass.)
    if (lane < ndof) {
      const int a = lane / D, y = lane % D;
      const double* da = dphi + ((int64_t)q * nloc + a) * NV;
      const double* ha = d2phi + ((int64_t)q * nloc + a) * NV * NV;
      double gy[NV];
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        gy[k] = g[k][0];
#pragma unroll
        for (int yy = 1; yy < D; ++yy) gy[k] = y == yy ? g[k][yy] : gy[k];
      }
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < NV; ++i) t = __builtin_fma(da[i], gy[i], t);
      gp[a * D + y] = t;
      double M[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        double m = 0.0;
#pragma unroll
        for (int k = 0; k < NV; ++k) m = __builtin_fma(ha[i * NV + k], gy[k], m);
        M[i] = m;
      }
#pragma unroll
      for (int x = 0; x < D; ++x) {
        double h = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) h = __builtin_fma(g[i][x], M[i], h);
        hs[(a * D + x) * D + y] = h;
      }
      if (y == 0) ph[a] = phi[(int64_t)q * nloc + a];
    }
    __syncthreads();
    // ---- state at the point: a lane per output (u_i | Gu_ix | the second-order part of Lu_i)
    if (lane < D) {
      double t = 0.0;
      for (int a = 0; a < nloc; ++a) t = __builtin_fma(ph[a], Uk[a * D + lane], t);
      st[lane] = t;
    } else if (lane < D + D * D) {
      const int i = (lane - D) / D, x = (lane - D) % D;
      double t = 0.0;
      for (int a = 0; a < nloc; ++a) t = __builtin_fma(gp[a * D + x], Uk[a * D + i], t);
      st[D + i * D + x] = t;
    } else if (lane < 2 * D + D * D) {
      const int j = lane - D - D * D;     // Lu_j = -nu sum_a (lap_a U_aj + sum_i hs_a[j][i] U_ai)   (+ convective part below)
      double t = 0.0;
      for (int a = 0; a < nloc; ++a) {
        double la = 0.0;                  // lap_a = trace of the Hessian (lap[] is written by other lanes in this phase)
#pragma unroll
        for (int x = 0; x < D; ++x) la += hs[(a * D + x) * D + x];
        t = __builtin_fma(-nu * la, Uk[a * D + j], t);
#pragma unroll
        for (int i = 0; i < D; ++i) t = __builtin_fma(-nu * hs[(a * D + j) * D + i], Uk[a * D + i], t);
      }
      st[D + D * D + j] = t;
    } else if (lane >= 32 && lane < 32 + nloc) {
      const int a = lane - 32;
      double la = 0.0;
#pragma unroll
      for (int x = 0; x < D; ++x) la += hs[(a * D + x) * D + x];
      lap[a] = la;
    }
    __syncthreads();
    double u[D], Gu[D][D], Lu[D];
#pragma unroll
    for (int i = 0; i < D; ++i) u[i] = st[i];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int x = 0; x < D; ++x) Gu[i][x] = st[D + i * D + x];
    double uu = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double t = st[D + D * D + i];
#pragma unroll
      for (int x = 0; x < D; ++x) t = __builtin_fma(u[x], Gu[i][x], t);
      Lu[i] = t;
      uu = __builtin_fma(u[i], u[i], uu);
    }
    const double vis = 4.0 * nu / h2;
    const double beta = 1.0 / sqrt(4.0 * uu / h2 + magic * vis * vis);
    const double b3 = -4.0 * beta * beta * beta / h2;
    const double wt = wq[q] * vw;
    // ---- element residual: entry (a, i) = lane
    if (lane < ndof) {
      const int a = lane / D, i = lane % D;
      double sa = 0.0;
#pragma unroll
      for (int x = 0; x < D; ++x) sa = __builtin_fma(u[x], gp[a * D + x], sa);
      // (component select without dynamic register indexing)
      double Li = Lu[0];
#pragma unroll
      for (int t = 1; t < D; ++t) Li = i == t ? Lu[t] : Li;
      facc = __builtin_fma(wt * beta * Li, sa, facc);
    }
    // ---- element matrix.  Entry ((a, i), (b, j)) at this point is
    //        wt (b3 u_j phi_b L_i s_a + beta dL_bij s_a + beta L_i phi_b dphi_a/dx_j),   s_a = u . grad phi_a,
    //        dL_bij = phi_b G_ij - nu H_b[i][j] + delta_ij (- nu lap_b + s_b)
    //      = saw[a] c1[b, i, j] + c2w[b, i] gp[a, j]   with the three tables below (nloc + nloc D D + nloc D values per point,
    //      formed once by the wave instead of once per entry): two FMAs and four LDS reads per entry.
    if (WANT) {
      if (lane < nloc) {
        double sa = 0.0;
#pragma unroll
        for (int x = 0; x < D; ++x) sa = __builtin_fma(u[x], gp[lane * D + x], sa);
        saw[lane] = sa;                    // (scaled by wt below, after c1 has read the unscaled value)
      }
      __syncthreads();
      for (int e = lane; e < nloc * D * D; e += 64) {
        const int b = e / (D * D), i = (e / D) % D, j = e % D;
        double Li = Lu[0], uj = u[0], Gij = 0.0;
#pragma unroll
        for (int tt = 1; tt < D; ++tt) {
          Li = i == tt ? Lu[tt] : Li;
          uj = j == tt ? u[tt] : uj;
        }
#pragma unroll
        for (int ii = 0; ii < D; ++ii)
#pragma unroll
          for (int jj = 0; jj < D; ++jj) Gij = (i == ii && j == jj) ? Gu[ii][jj] : Gij;
        double dL = __builtin_fma(ph[b], Gij, -nu * hs[e]);
        if (i == j) dL += -nu * lap[b] + saw[b];
        c1[e] = b3 * uj * ph[b] * Li + beta * dL;
        if (j == 0) c2w[b * D + i] = wt * beta * Li * ph[b];
      }
      __syncthreads();
      if (lane < nloc) saw[lane] *= wt;
      __syncthreads();
#pragma unroll
      for (int t = 0; t < EPL; ++t) {
        // (opaque to the optimiser: otherwise the 4 EPL LDS addresses are hoisted out of the loop over the points and held in
        // registers -- 256 + 47 of them, one wave per SIMD, every latency of a point exposed)
        uint32_t k = pk[t];
        asm volatile("" : "+v"(k));
        acc[t] = __builtin_fma(saw[k & 255u], c1[(k >> 8) & 255u], __builtin_fma(c2w[(k >> 16) & 255u], gp[k >> 24], acc[t]));
      }
    }
    __syncthreads();     // the next point overwrites gp / hs / st
  }
  const int64_t slot = blockIdx.x;
  if (lane < ndof) Fe_out[slot * ndof + lane] = facc;
  if (WANT) {
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
      const int e = lane + 64 * t;
      if (e < ndof * ndof) Ae_out[slot * (int64_t)ndof * ndof + e] = acc[t];
    }
  }
}

// vals (lane-major planes of the level operator) += the element matrices of the cells [cell0, cell0 + nslot) -- every block adds
// the entries of its contributors that lie in that cell range, in list order
template <int D>
__global__ __launch_bounds__(256) void supg_gather_kernel(int64_t nnzb, int nloc, int64_t cell0, int64_t nslot,
                                                           const int64_t* __restrict__ cptr, const int32_t* __restrict__ ccell,
                                                           const uint16_t* __restrict__ cba, const double* __restrict__ Ae,
                                                           double* __restrict__ vals) {
  constexpr int BB = D * D;
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= nnzb) return;
  const int ndof = nloc * D;
  double acc[D][D];
#pragma unroll
  for (int cc = 0; cc < D; ++cc)
#pragma unroll
    for (int dd = 0; dd < D; ++dd) acc[cc][dd] = 0.0;
  bool any = false;
  for (int64_t q = cptr[k]; q < cptr[k + 1]; ++q) {
    const int64_t cell = ccell[q];
    if (cell < cell0 || cell >= cell0 + nslot) continue;
    any = true;
    const int ba = cba[q], a = ba % nloc, b = ba / nloc;
    const double* M = Ae + (cell - cell0) * (int64_t)ndof * ndof + (int64_t)(a * D) * ndof + b * D;
#pragma unroll
    for (int cc = 0; cc < D; ++cc)
#pragma unroll
      for (int dd = 0; dd < D; ++dd) acc[cc][dd] += M[cc * ndof + dd];
  }
  if (!any) return;
#pragma unroll
  for (int cc = 0; cc < D; ++cc)
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
      const int64_t at = bsr_val_index(1, k, cc * D + dd, BB);
      vals[at] += acc[cc][dd];
    }
}

// F (n dofs) += the element residuals of the cells [cell0, cell0 + nslot): node r through the contributor list of its diagonal
// block (which holds exactly the (cell, a, a) with node a of the cell == r)
template <int D>
__global__ __launch_bounds__(256) void supg_residual_kernel(int64_t nnode, int nloc, int64_t cell0, int64_t nslot,
                                                             const int32_t* __restrict__ diag, const int64_t* __restrict__ cptr,
                                                             const int32_t* __restrict__ ccell, const uint16_t* __restrict__ cba,
                                                             const double* __restrict__ Fe, double* __restrict__ F) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= nnode) return;
  const int ndof = nloc * D;
  const int64_t k = diag[r];
  double acc[D];
#pragma unroll
  for (int i = 0; i < D; ++i) acc[i] = 0.0;
  for (int64_t q = cptr[k]; q < cptr[k + 1]; ++q) {
    const int64_t cell = ccell[q];
    if (cell < cell0 || cell >= cell0 + nslot) continue;
    const int a = cba[q] % nloc;
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] += Fe[(cell - cell0) * ndof + a * D + i];
  }
#pragma unroll
  for (int i = 0; i < D; ++i) F[r * D + i] += acc[i];
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

int launch_assemble_gather(alfi_level* L, double nu, double gamma, double adv, const double* d_state, int apply_bc,
                           double* out_vals) {
  alfi_ctx* ctx = L->ctx;
  const AssemblyDev& S = L->asmb;
  const int64_t nnzb = L->A.nnzb;
  dim3 grid((unsigned)((nnzb + 255) / 256)), block(256);
  if (L->bs == 2)
    hipLaunchKernelGGL(assemble_gather_kernel<2>, grid, block, 0, ctx->stream, nnzb, S.nloc, S.cptr, S.ccell, S.cba, S.cell_nodes,
                       S.grad, S.vol, S.Ta, S.Tb, S.Kv, S.Dv, d_state, S.bc_all ? S.bc_all : L->bc_mask, nu, gamma, adv, apply_bc, out_vals);
  else
    hipLaunchKernelGGL(assemble_gather_kernel<3>, grid, block, 0, ctx->stream, nnzb, S.nloc, S.cptr, S.ccell, S.cba, S.cell_nodes,
                       S.grad, S.vol, S.Ta, S.Tb, S.Kv, S.Dv, d_state, S.bc_all ? S.bc_all : L->bc_mask, nu, gamma, adv, apply_bc, out_vals);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

// SUPG terms about d_state: the Newton linearisation added to the level operator (add_vals), the residual contribution added to
// d_F (may be NULL); cells in batches bounded by ALFI_SUPG_SCRATCH_MB (default 4096) of element-matrix scratch
int launch_supg(alfi_level* L, double nu, double weight, double magic, const double* d_state, int add_vals, double* d_F) {
  alfi_ctx* ctx = L->ctx;
  const AssemblyDev& S = L->asmb;
  const int d = L->bs, nloc = S.nloc, ndof = nloc * d;
  static const int64_t scratch_mb = getenv("ALFI_SUPG_SCRATCH_MB") ? atoll(getenv("ALFI_SUPG_SCRATCH_MB")) : 4096;
  const int64_t per_cell = (add_vals ? (int64_t)ndof * ndof : 0) + ndof;
  int64_t batch = std::max<int64_t>(1, (scratch_mb << 20) / (8 * per_cell));
  if (batch > S.ncell) batch = S.ncell;
  double *Ae = nullptr, *Fe = nullptr;
  if (add_vals) ALFI_HIP_CHECK(ctx, hipMalloc((void**)&Ae, sizeof(double) * (size_t)(batch * ndof * ndof)));
  hipError_t e = hipMalloc((void**)&Fe, sizeof(double) * (size_t)(batch * ndof));
  if (e != hipSuccess) {
    (void)hipFree(Ae);
    return alfi_set_error(ctx, ALFI_E_HIP, "SUPG scratch: %s", hipGetErrorString(e));
  }
  const size_t lds = sizeof(double) * (size_t)(2 * nloc * d + nloc * d * d + 2 * nloc + 2 * d + d * d + nloc + nloc * d + nloc * d * d);
  const int epl = (ndof * ndof + 63) / 64;
  const int64_t nnzb = L->A.nnzb, nnode = L->A.nbrows;
  int rc = 0;
  for (int64_t c0 = 0; c0 < S.ncell && rc == 0; c0 += batch) {
    const int64_t nb = std::min<int64_t>(batch, S.ncell - c0);
    dim3 grid((unsigned)nb), block(64);
#define ALFI_SUPG_CELL(DV, EV)                                                                                                  \
  do {                                                                                                                          \
    if (add_vals)                                                                                                               \
      hipLaunchKernelGGL((supg_cell_kernel<DV, EV, true>), grid, block, lds, ctx->stream, c0, S.ncell, nloc, S.cell_nodes,      \
                         S.grad, S.vol, S.hcell, S.nq, S.wq, S.phi, S.dphi, S.d2phi, d_state, nu, weight, magic, 1, Ae, Fe);    \
    else                                                                                                                        \
      hipLaunchKernelGGL((supg_cell_kernel<DV, 1, false>), grid, block, lds, ctx->stream, c0, S.ncell, nloc, S.cell_nodes,      \
                         S.grad, S.vol, S.hcell, S.nq, S.wq, S.phi, S.dphi, S.d2phi, d_state, nu, weight, magic, 0, Ae, Fe);    \
  } while (0)
    if (d == 2) {
      if (epl <= 4) ALFI_SUPG_CELL(2, 4); else if (epl <= 16) ALFI_SUPG_CELL(2, 16); else ALFI_SUPG_CELL(2, 64);
    } else {
      if (epl <= 16) ALFI_SUPG_CELL(3, 16); else if (epl <= 32) ALFI_SUPG_CELL(3, 32); else ALFI_SUPG_CELL(3, 64);
    }
#undef ALFI_SUPG_CELL
    if (hipGetLastError() != hipSuccess) rc = alfi_set_error(ctx, ALFI_E_HIP, "supg_cell_kernel launch failed");
    if (rc == 0 && add_vals) {
      dim3 g2((unsigned)((nnzb + 255) / 256)), b2(256);
      if (d == 2)
        hipLaunchKernelGGL(supg_gather_kernel<2>, g2, b2, 0, ctx->stream, nnzb, nloc, c0, nb, S.cptr, S.ccell, S.cba, Ae, L->A.vals);
      else
        hipLaunchKernelGGL(supg_gather_kernel<3>, g2, b2, 0, ctx->stream, nnzb, nloc, c0, nb, S.cptr, S.ccell, S.cba, Ae, L->A.vals);
      if (hipGetLastError() != hipSuccess) rc = alfi_set_error(ctx, ALFI_E_HIP, "supg_gather_kernel launch failed");
    }
    if (rc == 0 && d_F) {
      dim3 g3((unsigned)((nnode + 255) / 256)), b3(256);
      if (d == 2)
        hipLaunchKernelGGL(supg_residual_kernel<2>, g3, b3, 0, ctx->stream, nnode, nloc, c0, nb, S.diag, S.cptr, S.ccell, S.cba, Fe, d_F);
      else
        hipLaunchKernelGGL(supg_residual_kernel<3>, g3, b3, 0, ctx->stream, nnode, nloc, c0, nb, S.diag, S.cptr, S.ccell, S.cba, Fe, d_F);
      if (hipGetLastError() != hipSuccess) rc = alfi_set_error(ctx, ALFI_E_HIP, "supg_residual_kernel launch failed");
    }
  }
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(Ae);
  (void)hipFree(Fe);
  return rc;
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
