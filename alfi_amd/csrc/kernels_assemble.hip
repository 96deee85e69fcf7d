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

// lane-major -> host layout (nnzb, d, d): alfi_level_get_values (tests, diagnostics)
__global__ void vals_from_lanes_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t total, int bb) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    dst[e] = src[bsr_val_index(1, e / bb, (int)(e % bb), bb)];
}

}  // namespace

int launch_assemble_gather(alfi_level* L, double nu, double gamma, double adv, const double* d_state, int apply_bc) {
  alfi_ctx* ctx = L->ctx;
  const AssemblyDev& S = L->asmb;
  const int64_t nnzb = L->A.nnzb;
  dim3 grid((unsigned)((nnzb + 255) / 256)), block(256);
  if (L->bs == 2)
    hipLaunchKernelGGL(assemble_gather_kernel<2>, grid, block, 0, ctx->stream, nnzb, S.nloc, S.cptr, S.ccell, S.cba, S.cell_nodes,
                       S.grad, S.vol, S.Ta, S.Tb, S.Kv, S.Dv, d_state, L->bc_mask, nu, gamma, adv, apply_bc, L->A.vals);
  else
    hipLaunchKernelGGL(assemble_gather_kernel<3>, grid, block, 0, ctx->stream, nnzb, S.nloc, S.cptr, S.ccell, S.cba, S.cell_nodes,
                       S.grad, S.vol, S.Ta, S.Tb, S.Kv, S.Dv, d_state, L->bc_mask, nu, gamma, adv, apply_bc, L->A.vals);
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
