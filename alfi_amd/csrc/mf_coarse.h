// Sparse direct solver for the coarse grid: multifrontal block L D U on a nested-dissection ordering, all fronts dense,
// every product and inversion on the FP64 matrix cores.  Included at the end of kernels_bigpatch.hip (it re-uses
// big_tile_product and the blocked Gauss-Jordan of big_factor_core).
//
// The reference hands the coarse operator to a sparse direct package (AssembledPC + LU through MUMPS / SuperLU_DIST,
// alfi/solver.py:369-378 [3P]); the dense inverse of alfi_coarse_factor stops at ~1e5 dofs (8 n^2 bytes).  This is the
// same exact solve for the coarse grids that no longer fit: memory O(n^{4/3}) in 3-D.
//
//   ordering   recursive bisection of the nodes (by the longest coordinate axis, or by BFS level sets when no coordinates
//              are given); separator = the nodes of the first half with a neighbour in the second.  Tree nodes are numbered
//              in post order; node t owns the nodes s_t it eliminates and carries the boundary b_t = the not yet
//              eliminated nodes its subtree couples to (all in separators of ancestors).
//   factor     bottom-up by height, every height one batch: front F = [ss sb; bs bb] assembled from the operator's blocks
//              and the children's Schur complements (extend-add, one launch per child slot -- no atomics), then
//                X_t = F_ss^-1  (blocked Gauss-Jordan + Newton-Schulz, as for the macro-star patches)
//                W_t = X_t F_sb,  L_t = F_bs X_t,  U_t = F_bb - F_bs W_t  (batched 64 x 64-tile products)
//   solve      forward, heights ascending: v_s = r_s - (contributions of the descendants, gathered in a fixed order),
//              c_t = L_t v_s;  backward, heights descending: x_s = X_t v_s - W_t x_b.  Three launches per height.
//              One step of iterative refinement with the level's own SpMV on top (the explicit inverses lose
//              ~cond * eps, like the condensed patch factors; the refinement squares that).
#include <cmath>
#include <cstring>
#include <type_traits>
#include <utility>

struct MfDev {
  int bs = 0, H = 0;
  int64_t n = 0, nnode = 0, fac_doubles = 0;
  int refine = 1;
  std::vector<int64_t> lp, h_sdof_ptr;      // nodes of height h: [lp[h], lp[h+1]);  s-dof list offsets
  std::vector<int> lvl_max_ns, lvl_max_nb;
  int32_t *ns = nullptr, *nb = nullptr, *sdofs = nullptr, *bdofs = nullptr;
  int64_t *xoff = nullptr, *loff = nullptr, *woff = nullptr, *coff = nullptr, *sdof_ptr = nullptr, *bdof_ptr = nullptr,
          *g_ptr = nullptr, *g_idx = nullptr;
  double *fac = nullptr, *contrib = nullptr, *vloc = nullptr, *tmp_r = nullptr, *tmp_x = nullptr;
  std::vector<void*> allocs;
  // the symbolic plan kept for re-factorisations with new operator values (every Newton step): host copies of the sizes
  // and offsets, the device index arrays of the assembly (MfFactorArgs, owned through allocs)
  std::vector<int32_t> h_ns, h_nb;
  std::vector<int64_t> h_xoff, h_loff, h_woff, h_fsb, h_fbs, h_uo, h_fss;
  int64_t tmp_doubles = 0, nnzb = 0;
  struct MfFactorArgs* fa = nullptr;
  int leaf = 0;
  bool with_coords = false;
};

void mf_free_args(struct MfFactorArgs* a);
void mf_free(MfDev* m) {
  if (!m) return;
  for (void* p : m->allocs) (void)hipFree(p);
  mf_free_args(m->fa);
  delete m;
}
int64_t mf_bytes(const MfDev* m) { return m ? m->fac_doubles * 8 : 0; }

static inline int mf_pad(int v) { return (v + BIG_NB - 1) / BIG_NB * BIG_NB; }

// ---- host: ordering and symbolic factorisation ----------------------------------------------------------------------------
struct MfTreeNode {
  int parent = -1, child[2] = {-1, -1}, height = 0, first = 0;
  std::vector<int32_t> s, b;
};

struct MfBuilder {
  const std::vector<int64_t>& xadj;
  const std::vector<int32_t>& adj;
  const double* coords;
  int dim, leaf;
  std::vector<int32_t> mark, dist, seen;
  int32_t tag = 0;
  std::vector<MfTreeNode> nodes;

  MfBuilder(const std::vector<int64_t>& xa, const std::vector<int32_t>& ad, const double* c, int d, int lf, int64_t nn)
      : xadj(xa), adj(ad), coords(c), dim(d), leaf(lf), mark((size_t)nn, 0), dist((size_t)nn, 0), seen((size_t)nn, 0) {}

  // split key of the nodes of D: the coordinate along the longest axis of D's bounding box, or the BFS distance from a
  // pseudo-peripheral node of D
  void keys(const std::vector<int32_t>& D, std::vector<double>* key) {
    key->resize(D.size());
    if (coords) {
      int axis = 0;
      double best = -1.0;
      for (int a = 0; a < dim; ++a) {
        double lo = INFINITY, hi = -INFINITY;
        for (int32_t i : D) {
          const double c = coords[(int64_t)i * dim + a];
          lo = std::min(lo, c);
          hi = std::max(hi, c);
        }
        if (hi - lo > best) best = hi - lo, axis = a;
      }
      for (size_t q = 0; q < D.size(); ++q) (*key)[q] = coords[(int64_t)D[q] * dim + axis];
      return;
    }
    const int32_t member = ++tag;
    for (int32_t i : D) mark[i] = member;
    std::vector<int32_t> queue;
    int32_t start = D[0];
    for (int pass = 0; pass < 3; ++pass) {
      // BFS inside D from start; nodes of D in other components get the distance "far"
      const int32_t visited = ++tag;
      queue.clear();
      queue.push_back(start);
      seen[start] = visited;
      dist[start] = 0;
      for (size_t h = 0; h < queue.size(); ++h) {
        const int32_t i = queue[h];
        for (int64_t k = xadj[i]; k < xadj[i + 1]; ++k) {
          const int32_t j = adj[k];
          if (mark[j] == member && seen[j] != visited) {
            seen[j] = visited;
            dist[j] = dist[i] + 1;
            queue.push_back(j);
          }
        }
      }
      const int32_t far = dist[queue.back()] + 1;
      for (int32_t i : D)
        if (seen[i] != visited) dist[i] = far;
      start = queue.back();
    }
    for (size_t q = 0; q < D.size(); ++q) (*key)[q] = (double)dist[D[q]];
  }

  int leaf_node(std::vector<int32_t>& D) {
    MfTreeNode t;
    std::sort(D.begin(), D.end());
    t.s = D;
    t.first = (int)nodes.size();
    nodes.push_back(std::move(t));
    return (int)nodes.size() - 1;
  }

  int build(std::vector<int32_t>& D) {
    if ((int)D.size() <= leaf) return leaf_node(D);
    std::vector<double> key;
    keys(D, &key);
    std::vector<std::pair<double, int32_t>> order(D.size());
    for (size_t q = 0; q < D.size(); ++q) order[q] = {key[q], D[q]};
    size_t mid = D.size() / 2;
    std::nth_element(order.begin(), order.begin() + mid, order.end());
    // nodes with the median key stay together (a cut along a mesh plane / one BFS level gives a one-layer separator; a cut
    // through the ties a jagged, thicker one) unless that leaves a side with less than a quarter of the nodes
    {
      const double km = order[mid].first;
      auto below = std::partition(order.begin(), order.end(), [km](const std::pair<double, int32_t>& o) { return o.first < km; });
      auto upto = std::partition(below, order.end(), [km](const std::pair<double, int32_t>& o) { return o.first <= km; });
      const size_t nlt = (size_t)(below - order.begin()), nle = (size_t)(upto - order.begin());
      const size_t quarter = D.size() / 4, half = D.size() / 2;
      const bool ok_lt = nlt >= quarter && D.size() - nlt >= quarter, ok_le = nle >= quarter && D.size() - nle >= quarter;
      const size_t d_lt = half > nlt ? half - nlt : nlt - half, d_le = half > nle ? half - nle : nle - half;
      if (ok_lt && (!ok_le || d_lt <= d_le)) mid = nlt;
      else if (ok_le) mid = nle;
      else std::sort(below, upto);      // a level too large to keep together: ties by node number, cut in the middle
    }
    tag += 2;
    const int32_t in1 = tag - 1, in2 = tag;
    for (size_t q = 0; q < mid; ++q) mark[order[q].second] = in1;
    for (size_t q = mid; q < order.size(); ++q) mark[order[q].second] = in2;
    // two candidate vertex separators of the edge cut: the nodes of the first half with a neighbour in the second, or the
    // other way round -- the smaller one is taken (for a cut along a mesh plane: the plane, not the layer of cells below it)
    size_t n12 = 0, n21 = 0;
    for (size_t q = 0; q < order.size(); ++q) {
      const int32_t i = order[q].second, other = q < mid ? in2 : in1;
      bool sep = false;
      for (int64_t k = xadj[i]; k < xadj[i + 1] && !sep; ++k) sep = mark[adj[k]] == other;
      if (sep) {
        dist[i] = 1;
        ++(q < mid ? n12 : n21);
      } else {
        dist[i] = 0;
      }
    }
    const bool from_first = n12 <= n21;
    std::vector<int32_t> D1, D2, S;
    for (size_t q = 0; q < order.size(); ++q) {
      const int32_t i = order[q].second;
      const bool first = q < mid;
      if (dist[i] == 1 && first == from_first) S.push_back(i);
      else (first ? D1 : D2).push_back(i);
    }
    std::vector<int32_t>().swap(D);
    if (S.empty()) {              // the halves do not touch: any node serves as the (formal) separator
      std::vector<int32_t>& from = D1.size() > 1 || D2.empty() ? D1 : D2;
      S.push_back(from.back());
      from.pop_back();
    }
    int c1 = -1, c2 = -1, first = (int)nodes.size();
    if (!D1.empty()) c1 = build(D1);
    if (!D2.empty()) c2 = build(D2);
    MfTreeNode t;
    std::sort(S.begin(), S.end());
    t.s = std::move(S);
    t.first = first;
    t.child[0] = c1 >= 0 ? c1 : c2;
    t.child[1] = c1 >= 0 ? c2 : -1;
    const int me = (int)nodes.size();
    for (int c : t.child)
      if (c >= 0) {
        nodes[c].parent = me;
        t.height = std::max(t.height, nodes[c].height + 1);
      }
    nodes.push_back(std::move(t));
    return me;
  }
};

// ---- device: assembly of the fronts ------------------------------------------------------------------------------------------
struct MfFactorArgs {
  int bs;
  const int32_t *ns, *nb, *snn;            // dofs in s / b per tree node, nodes in s
  const int64_t *fsb_off, *fbs_off, *u_off, *fss_off;  // into tmp
  double* tmp;
  const int32_t* child;                    // 2 per tree node
  const int64_t* cmap_ptr;
  const int32_t* cmap;
  const int64_t *ss_ptr, *ob_ptr;
  const int4 *ss_ent, *ob_ent;
  const double* vals;
  int flat;
};

void mf_free_args(MfFactorArgs* a) { delete a; }

// zero + identity padding of the N x N scratch matrices of a batch
__global__ void mf_pad_kernel(int64_t pbase, const int32_t* __restrict__ ns, const int64_t* __restrict__ scr_ptr,
                              double* __restrict__ scr) {
  const int n = ns[pbase + blockIdx.y], N = (n + BIG_NB - 1) / BIG_NB * BIG_NB;
  double* S = scr + scr_ptr[blockIdx.y];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)N * N; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / N), c = (int)(e % N);
    S[e] = (r == c && r >= n) ? 1.0 : 0.0;
  }
}

// blocks of the operator into the front: SS = true: (s, s) entries into the scratch matrix, else (s, b) / (b, s) into tmp
template <bool SS>
__global__ void mf_scatter_kernel(int64_t pbase, MfFactorArgs a, const int64_t* __restrict__ scr_ptr,
                                  double* __restrict__ scr) {
  const int64_t p = pbase + blockIdx.y;
  const int bs = a.bs, bb = bs * bs;
  const int Ns = (a.ns[p] + BIG_NB - 1) / BIG_NB * BIG_NB, Nb = (a.nb[p] + BIG_NB - 1) / BIG_NB * BIG_NB;
  const int64_t e0 = SS ? a.ss_ptr[p] : a.ob_ptr[p], e1 = SS ? a.ss_ptr[p + 1] : a.ob_ptr[p + 1];
  const int4* ent = SS ? a.ss_ent : a.ob_ent;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < (e1 - e0) * bb; q += (int64_t)gridDim.x * blockDim.x) {
    const int4 en = ent[e0 + q / bb];
    const int rc = (int)(q % bb);
    const double v = a.vals[bsr_val_index(a.flat, en.x, rc, bb)];
    const int r = en.y * bs + rc / bs, c = en.z * bs + rc % bs;
    if (SS) scr[scr_ptr[blockIdx.y] + (int64_t)r * Ns + c] = v;
    else if (en.w == 1) a.tmp[a.fsb_off[p] + (int64_t)r * Nb + c] = v;
    else a.tmp[a.fbs_off[p] + (int64_t)r * Ns + c] = v;
  }
}

// extend-add of the Schur complement of child `slot` into the parent's front (SS: only the (s, s) part, into the scratch)
template <bool SS>
__global__ void mf_extend_kernel(int64_t pbase, int slot, MfFactorArgs a, const int64_t* __restrict__ scr_ptr,
                                 double* __restrict__ scr) {
  const int64_t p = pbase + blockIdx.y;
  const int c = a.child[2 * p + slot];
  if (c < 0) return;
  const int bs = a.bs;
  const int ns = a.ns[p], Ns = (ns + BIG_NB - 1) / BIG_NB * BIG_NB, Nb = (a.nb[p] + BIG_NB - 1) / BIG_NB * BIG_NB;
  const int nbc = a.nb[c], Nbc = (nbc + BIG_NB - 1) / BIG_NB * BIG_NB;
  const double* U = a.tmp + a.u_off[c];
  const int32_t* cm = a.cmap + a.cmap_ptr[c];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)nbc * nbc; e += (int64_t)gridDim.x * blockDim.x) {
    const int k1 = (int)(e / nbc), k2 = (int)(e % nbc);
    const int r = cm[k1 / bs] * bs + k1 % bs, q = cm[k2 / bs] * bs + k2 % bs;
    const double v = U[(int64_t)k1 * Nbc + k2];
    if (SS) {
      if (r < ns && q < ns) scr[scr_ptr[blockIdx.y] + (int64_t)r * Ns + q] += v;
    } else if (r < ns) {
      if (q >= ns) a.tmp[a.fsb_off[p] + (int64_t)r * Nb + (q - ns)] += v;
    } else if (q < ns) {
      a.tmp[a.fbs_off[p] + (int64_t)(r - ns) * Ns + q] += v;
    } else {
      a.tmp[a.u_off[p] + (int64_t)(r - ns) * Nb + (q - ns)] += v;
    }
  }
}

__global__ void mf_store_kernel(int64_t pbase, const int32_t* __restrict__ ns, const int64_t* __restrict__ xoff,
                                const int64_t* __restrict__ scr_ptr, const double* __restrict__ res, double* __restrict__ fac);

struct MfFill {
  MfFactorArgs a;
  int64_t level_base;
  const int32_t* ns;
  const int64_t* xoff;
  double* fac;
  void fill(alfi_ctx* ctx, int64_t p0, int64_t nb, const int64_t* d_scr_ptr, double* dst) const {
    const dim3 block(256), grid(32, (unsigned)nb);
    const int64_t pb = level_base + p0;
    hipLaunchKernelGGL(mf_pad_kernel, dim3(64, (unsigned)nb), block, 0, ctx->stream, pb, a.ns, d_scr_ptr, dst);
    hipLaunchKernelGGL(mf_scatter_kernel<true>, grid, block, 0, ctx->stream, pb, a, d_scr_ptr, dst);
    hipLaunchKernelGGL(mf_extend_kernel<true>, grid, block, 0, ctx->stream, pb, 0, a, d_scr_ptr, dst);
    hipLaunchKernelGGL(mf_extend_kernel<true>, grid, block, 0, ctx->stream, pb, 1, a, d_scr_ptr, dst);
    // F_ss is needed again for the refinement of W and L
    hipLaunchKernelGGL(mf_store_kernel, dim3(64, (unsigned)nb), block, 0, ctx->stream, pb, a.ns, a.fss_off, d_scr_ptr, dst, a.tmp);
  }
};

void mf_fill_dispatch(const MfFill* f, alfi_ctx* ctx, int64_t p0, int64_t nb, const int64_t* d_scr_ptr, double* dst) {
  f->fill(ctx, p0, nb, d_scr_ptr, dst);
}

__global__ void mf_store_kernel(int64_t pbase, const int32_t* __restrict__ ns, const int64_t* __restrict__ xoff,
                                const int64_t* __restrict__ scr_ptr, const double* __restrict__ res,
                                double* __restrict__ fac) {
  const int64_t p = pbase + blockIdx.y;
  const int N = (ns[p] + BIG_NB - 1) / BIG_NB * BIG_NB;
  const double* S = res + scr_ptr[blockIdx.y];
  double* X = fac + xoff[p];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)N * N; e += (int64_t)gridDim.x * blockDim.x)
    X[e] = S[e];
}

void mf_store_dispatch(const MfFill* f, alfi_ctx* ctx, int64_t p0, int64_t nb, const int64_t* d_scr_ptr, const double* res) {
  hipLaunchKernelGGL(mf_store_kernel, dim3(64, (unsigned)nb), dim3(256), 0, ctx->stream, f->level_base + p0, f->ns, f->xoff,
                     d_scr_ptr, res, f->fac);
}

// batched products on padded row-major operands: mode 0: C = A B, mode 1: C -= A B, mode 2: C += A B
struct MfGemm {
  const double *A, *B;
  double* C;
  int M, N, K, lda, ldb, ldc;
};
__global__ __launch_bounds__(256) void mf_gemm_kernel(const MfGemm* __restrict__ descs, int mode) {
  __shared__ double As[2][BIG_BK][BIG_NB];
  __shared__ double Bs[2][BIG_BK][BIG_NB];
  const MfGemm g = descs[blockIdx.y];
  const int tn = g.N / BIG_NB, tm = g.M / BIG_NB;
  if ((int)blockIdx.x >= tm * tn || g.K == 0) return;
  const int ti = blockIdx.x / tn, tj = blockIdx.x % tn;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lm = lane & 15, lk = lane >> 4;
  big_d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (big_d4){0.0, 0.0, 0.0, 0.0};
  big_tile_product(g.A + (int64_t)ti * BIG_NB * g.lda, g.lda, g.B + tj * BIG_NB, g.ldb, g.K, As, Bs, acc);
  const int r0 = ti * BIG_NB + (wave >> 1) * 32, c0 = tj * BIG_NB + (wave & 1) * 32;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t at = (int64_t)(r0 + 16 * a + lk + 4 * q) * g.ldc + c0 + 16 * b + lm;
        g.C[at] = mode == 0 ? acc[a][b][q] : mode == 1 ? g.C[at] - acc[a][b][q] : g.C[at] + acc[a][b][q];
      }
}

// ---- device: the two sweeps -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ double mf_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ void mf_gather_kernel(int64_t d0, int64_t d1, const int32_t* __restrict__ sdofs, const int64_t* __restrict__ g_ptr,
                                 const int64_t* __restrict__ g_idx, const double* __restrict__ contrib,
                                 const double* __restrict__ r, double* __restrict__ vloc) {
  const int64_t i = d0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d1) return;
  const int32_t dof = sdofs[i];
  double acc = r[dof];
  for (int64_t q = g_ptr[dof]; q < g_ptr[dof + 1]; ++q) acc -= contrib[g_idx[q]];
  vloc[i] = acc;
}

constexpr int MF_ROWS = 4;      // rows per workgroup of the sweeps: one wave per row

// sum_j a[j] v[j] over j < cols, four independent partial sums per lane (loads of four 512 B row segments in flight)
__device__ __forceinline__ double mf_row_dot(const double* __restrict__ a, const double* __restrict__ v, int cols, int lane) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int j = lane;
  for (; j + 192 < cols; j += 256) {
    const double a0 = a[j], a1 = a[j + 64], a2 = a[j + 128], a3 = a[j + 192];
    s0 += a0 * v[j];
    s1 += a1 * v[j + 64];
    s2 += a2 * v[j + 128];
    s3 += a3 * v[j + 192];
  }
  for (; j < cols; j += 64) s0 += a[j] * v[j];
  return (s0 + s1) + (s2 + s3);
}
__device__ __forceinline__ double mf_row_dot_gather(const double* __restrict__ a, const double* __restrict__ x,
                                                    const int32_t* __restrict__ idx, int cols, int lane) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int j = lane;
  for (; j + 192 < cols; j += 256) {
    const double a0 = a[j], a1 = a[j + 64], a2 = a[j + 128], a3 = a[j + 192];
    s0 += a0 * x[idx[j]];
    s1 += a1 * x[idx[j + 64]];
    s2 += a2 * x[idx[j + 128]];
    s3 += a3 * x[idx[j + 192]];
  }
  for (; j < cols; j += 64) s0 += a[j] * x[idx[j]];
  return (s0 + s1) + (s2 + s3);
}

// c_t = L_t v_s   (one wave per row of the boundary)
__global__ __launch_bounds__(256) void mf_fwd_kernel(int64_t pbase, const int32_t* __restrict__ ns,
                                                      const int32_t* __restrict__ nb, const int64_t* __restrict__ loff,
                                                      const int64_t* __restrict__ coff, const int64_t* __restrict__ sdof_ptr,
                                                      const double* __restrict__ fac, const double* __restrict__ vloc,
                                                      double* __restrict__ contrib) {
  const int64_t p = pbase + blockIdx.y;
  const int rows = nb[p], cols = ns[p], ld = (cols + BIG_NB - 1) / BIG_NB * BIG_NB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * MF_ROWS + wave;
  if (row >= rows) return;
  const double acc = mf_wave_sum(mf_row_dot(fac + loff[p] + (int64_t)row * ld, vloc + sdof_ptr[p], cols, lane));
  if (lane == 0) contrib[coff[p] + row] = acc;
}

// x_s = X_t v_s - W_t x_b
__global__ __launch_bounds__(256) void mf_bwd_kernel(int64_t pbase, const int32_t* __restrict__ ns,
                                                      const int32_t* __restrict__ nb, const int64_t* __restrict__ xoff,
                                                      const int64_t* __restrict__ woff, const int64_t* __restrict__ sdof_ptr,
                                                      const int64_t* __restrict__ bdof_ptr, const int32_t* __restrict__ sdofs,
                                                      const int32_t* __restrict__ bdofs, const double* __restrict__ fac,
                                                      const double* __restrict__ vloc, double* __restrict__ x) {
  const int64_t p = pbase + blockIdx.y;
  const int rows = ns[p], nbd = nb[p];
  const int ldx = (rows + BIG_NB - 1) / BIG_NB * BIG_NB, ldw = (nbd + BIG_NB - 1) / BIG_NB * BIG_NB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * MF_ROWS + wave;
  if (row >= rows) return;
  double acc = mf_row_dot(fac + xoff[p] + (int64_t)row * ldx, vloc + sdof_ptr[p], rows, lane);
  acc -= mf_row_dot_gather(fac + woff[p] + (int64_t)row * ldw, x, bdofs + bdof_ptr[p], nbd, lane);
  acc = mf_wave_sum(acc);
  if (lane == 0) x[sdofs[sdof_ptr[p] + row]] = acc;
}

// x = (L D U)^-1 r; r and x may not alias
static int mf_sweeps(alfi_ctx* ctx, const MfDev* m, const double* r, double* x) {
  const dim3 block(256);
  for (int h = 0; h < m->H; ++h) {
    const int64_t p0 = m->lp[h], p1 = m->lp[h + 1];
    const int64_t d0 = m->h_sdof_ptr[p0], d1 = m->h_sdof_ptr[p1];
    hipLaunchKernelGGL(mf_gather_kernel, dim3((unsigned)((d1 - d0 + 255) / 256)), block, 0, ctx->stream, d0, d1, m->sdofs,
                       m->g_ptr, m->g_idx, m->contrib, r, m->vloc);
    if (m->lvl_max_nb[h] > 0)
      hipLaunchKernelGGL(mf_fwd_kernel, dim3((unsigned)((m->lvl_max_nb[h] + MF_ROWS - 1) / MF_ROWS), (unsigned)(p1 - p0)),
                         block, 0, ctx->stream, p0, m->ns, m->nb, m->loff, m->coff, m->sdof_ptr, m->fac, m->vloc,
                         m->contrib);
  }
  for (int h = m->H - 1; h >= 0; --h) {
    const int64_t p0 = m->lp[h], p1 = m->lp[h + 1];
    hipLaunchKernelGGL(mf_bwd_kernel, dim3((unsigned)((m->lvl_max_ns[h] + MF_ROWS - 1) / MF_ROWS), (unsigned)(p1 - p0)),
                       block, 0, ctx->stream, p0, m->ns, m->nb, m->xoff, m->woff, m->sdof_ptr, m->bdof_ptr, m->sdofs,
                       m->bdofs, m->fac, m->vloc, x);
  }
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

int mf_solve(alfi_level* L, const double* b, double* x) {
  alfi_ctx* ctx = L->ctx;
  const MfDev* m = L->mf;
  ALFI_CHECK(mf_sweeps(ctx, m, b, x));
  for (int it = 0; it < m->refine; ++it) {
    ALFI_CHECK(launch_bsr_spmv(ctx, L->A, x, m->tmp_r, b, 1.0, 1));     // r = b - A x
    ALFI_CHECK(mf_sweeps(ctx, m, m->tmp_r, m->tmp_x));
    ALFI_CHECK(launch_axpy(ctx, x, m->tmp_x, 1.0, m->n));
  }
  return 0;
}

// ---- host driver: symbolic + numeric factorisation ---------------------------------------------------------------------------
template <typename T>
static int mf_up(alfi_ctx* ctx, MfDev* m, T** dst, const std::vector<T>& src) {
  T* d = nullptr;
  const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
  ALFI_HIP_CHECK(ctx, hipMalloc((void**)&d, bytes));
  if (m) m->allocs.push_back(d);
  if (!src.empty()) ALFI_HIP_CHECK(ctx, hipMemcpy(d, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  *dst = d;
  return 0;
}

static int mf_numeric(alfi_level* L, MfDev* m);

// ordering, symbolic factorisation and the device index arrays: *out owns everything (mf_free)
static int mf_analyse(alfi_level* L, const double* coords, int dim, int leaf_nodes, MfDev** out) {
  alfi_ctx* ctx = L->ctx;
  const int bs = L->bs;
  const int64_t nbn = L->A.nbrows;
  if (L->A.nbcols != nbn) return alfi_set_error(ctx, ALFI_E_ARG, "sparse coarse factorisation needs a square operator");
  // the operator's graph, symmetrised
  std::vector<int32_t> rowptr((size_t)nbn + 1), colidx((size_t)L->A.nnzb);
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ALFI_HIP_CHECK(ctx, hipMemcpy(rowptr.data(), L->A.rowptr, rowptr.size() * 4, hipMemcpyDeviceToHost));
  ALFI_HIP_CHECK(ctx, hipMemcpy(colidx.data(), L->A.colidx, colidx.size() * 4, hipMemcpyDeviceToHost));
  for (int32_t& c : colidx) c &= 0x7fffffff;
  std::vector<int64_t> xadj((size_t)nbn + 1, 0);
  for (int64_t i = 0; i < nbn; ++i)
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if (colidx[k] != i) ++xadj[i + 1], ++xadj[colidx[k] + 1];
  for (int64_t i = 0; i < nbn; ++i) xadj[i + 1] += xadj[i];
  std::vector<int32_t> adj((size_t)xadj[nbn]);
  {
    std::vector<int64_t> cur(xadj.begin(), xadj.end() - 1);
    for (int64_t i = 0; i < nbn; ++i)
      for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
        if (colidx[k] != i) adj[cur[i]++] = colidx[k], adj[cur[colidx[k]]++] = (int32_t)i;
    // duplicates (i, j) + (j, i) removed
    int64_t w = 0;
    std::vector<int64_t> nx((size_t)nbn + 1, 0);
    for (int64_t i = 0; i < nbn; ++i) {
      std::sort(adj.begin() + xadj[i], adj.begin() + xadj[i + 1]);
      const int64_t e = std::unique(adj.begin() + xadj[i], adj.begin() + xadj[i + 1]) - adj.begin();
      for (int64_t k = xadj[i]; k < e; ++k) adj[w++] = adj[k];
      nx[i + 1] = w;
    }
    adj.resize((size_t)w);
    xadj.swap(nx);
  }
  MfBuilder B(xadj, adj, coords, dim, leaf_nodes, nbn);
  {
    std::vector<int32_t> all((size_t)nbn);
    for (int64_t i = 0; i < nbn; ++i) all[i] = (int32_t)i;
    B.build(all);
  }
  std::vector<MfTreeNode>& T = B.nodes;
  const int nt = (int)T.size();
  std::vector<int32_t> owner((size_t)nbn, -1);
  for (int t = 0; t < nt; ++t)
    for (int32_t i : T[t].s) owner[i] = t;
  // boundaries, children before parents (post order)
  for (int t = 0; t < nt; ++t) {
    std::vector<int32_t>& b = T[t].b;
    for (int32_t i : T[t].s)
      for (int64_t k = xadj[i]; k < xadj[i + 1]; ++k) {
        const int o = owner[adj[k]];
        if (o > t) b.push_back(adj[k]);
        else if (o < T[t].first) return alfi_set_error(ctx, ALFI_E_STATE, "nested dissection: separator violated");
      }
    for (int c : T[t].child)
      if (c >= 0)
        for (int32_t j : T[c].b)
          if (owner[j] != t) b.push_back(j);
    std::sort(b.begin(), b.end());
    b.erase(std::unique(b.begin(), b.end()), b.end());
    // every boundary node is eliminated by an ancestor
    for (int32_t j : b) {
      int a = T[t].parent;
      while (a >= 0 && a != owner[j]) a = T[a].parent;
      if (a < 0) return alfi_set_error(ctx, ALFI_E_STATE, "nested dissection: boundary outside the ancestors");
    }
  }
  // tree nodes ordered by height
  int H = 0;
  for (const MfTreeNode& t : T) H = std::max(H, t.height + 1);
  std::vector<int> order((size_t)nt), pidx((size_t)nt);
  for (int t = 0; t < nt; ++t) order[t] = t;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return T[a].height < T[b].height; });
  for (int p = 0; p < nt; ++p) pidx[order[p]] = p;
  MfDev* m = new MfDev;
  m->bs = bs;
  m->n = L->n;
  m->nnode = nt;
  m->H = H;
  m->refine = 1;
  m->leaf = leaf_nodes;
  m->with_coords = coords != nullptr;
  m->nnzb = L->A.nnzb;
  m->lp.assign((size_t)H + 1, 0);
  for (int p = 0; p < nt; ++p) ++m->lp[T[order[p]].height + 1];
  for (int h = 0; h < H; ++h) m->lp[h + 1] += m->lp[h];
  m->lvl_max_ns.assign((size_t)H, 0);
  m->lvl_max_nb.assign((size_t)H, 0);
  std::vector<int32_t> ns((size_t)nt), nb((size_t)nt), snn((size_t)nt), sdofs, bdofs, child((size_t)2 * nt, -1), cmap;
  std::vector<int64_t> xoff((size_t)nt), loff((size_t)nt), woff((size_t)nt), coff((size_t)nt + 1, 0), sdp((size_t)nt + 1, 0),
      bdp((size_t)nt + 1, 0), fsb((size_t)nt), fbs((size_t)nt), uo((size_t)nt), fss((size_t)nt), cmp((size_t)nt + 1, 0), ssp((size_t)nt + 1, 0),
      obp((size_t)nt + 1, 0);
  std::vector<int4> sse, obe;
  int64_t fac = 0, tmp = 0;
  std::vector<int32_t> loc((size_t)nbn, -1);
  for (int p = 0; p < nt; ++p) {
    const MfTreeNode& t = T[order[p]];
    const int h = t.height;
    ns[p] = (int32_t)t.s.size() * bs;
    nb[p] = (int32_t)t.b.size() * bs;
    snn[p] = (int32_t)t.s.size();
    m->lvl_max_ns[h] = std::max(m->lvl_max_ns[h], (int)ns[p]);
    m->lvl_max_nb[h] = std::max(m->lvl_max_nb[h], (int)nb[p]);
    const int64_t Ns = mf_pad(ns[p]), Nb = nb[p] ? mf_pad(nb[p]) : 0;
    xoff[p] = fac, fac += Ns * Ns;
    loff[p] = fac, fac += Nb * Ns;
    woff[p] = fac, fac += Ns * Nb;
    fsb[p] = tmp, tmp += Ns * Nb;
    fbs[p] = tmp, tmp += Nb * Ns;
    uo[p] = tmp, tmp += Nb * Nb;
    fss[p] = tmp, tmp += Ns * Ns;
    coff[p + 1] = coff[p] + nb[p];
    for (int32_t i : t.s)
      for (int c = 0; c < bs; ++c) sdofs.push_back(i * bs + c);
    for (int32_t i : t.b)
      for (int c = 0; c < bs; ++c) bdofs.push_back(i * bs + c);
    sdp[p + 1] = (int64_t)sdofs.size();
    bdp[p + 1] = (int64_t)bdofs.size();
    for (int j = 0; j < 2; ++j) child[2 * p + j] = t.child[j] >= 0 ? pidx[t.child[j]] : -1;
  }
  // operator entries of every front and the children's index maps (node positions in the parent's [s | b] list)
  std::vector<std::vector<int32_t>> cmap_of((size_t)nt);
  for (int p = 0; p < nt; ++p) {
    const MfTreeNode& t = T[order[p]];
    const int nsn = (int)t.s.size();
    for (int q = 0; q < nsn; ++q) loc[t.s[q]] = q;
    for (size_t q = 0; q < t.b.size(); ++q) loc[t.b[q]] = nsn + (int)q;
    for (int q = 0; q < nsn; ++q) {
      const int32_t i = t.s[q];
      for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        const int lj = loc[colidx[k]];
        if (lj < 0) continue;
        if (lj < nsn) sse.push_back(make_int4(k, q, lj, 0));
        else obe.push_back(make_int4(k, q, lj - nsn, 1));
      }
    }
    for (size_t q = 0; q < t.b.size(); ++q) {
      const int32_t i = t.b[q];
      for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        const int lj = loc[colidx[k]];
        if (lj >= 0 && lj < nsn) obe.push_back(make_int4(k, (int)q, lj, 2));
      }
    }
    ssp[p + 1] = (int64_t)sse.size();
    obp[p + 1] = (int64_t)obe.size();
    for (int c : t.child)
      if (c >= 0) {
        std::vector<int32_t>& cm = cmap_of[pidx[c]];
        for (int32_t j : T[c].b) {
          if (loc[j] < 0) {
            mf_free(m);
            return alfi_set_error(ctx, ALFI_E_STATE, "nested dissection: child boundary outside the parent's front");
          }
          cm.push_back(loc[j]);
        }
      }
    for (int32_t i : t.s) loc[i] = -1;
    for (int32_t i : t.b) loc[i] = -1;
  }
  for (int p = 0; p < nt; ++p) {
    cmap.insert(cmap.end(), cmap_of[p].begin(), cmap_of[p].end());
    cmp[p + 1] = (int64_t)cmap.size();
  }
  // descendants' contributions to every dof, in tree-node order
  std::vector<int64_t> gptr((size_t)L->n + 1, 0), gidx;
  for (int64_t q = 0; q < (int64_t)bdofs.size(); ++q) ++gptr[bdofs[q] + 1];
  for (int64_t i = 0; i < L->n; ++i) gptr[i + 1] += gptr[i];
  gidx.resize((size_t)gptr[L->n]);
  {
    std::vector<int64_t> cur(gptr.begin(), gptr.end() - 1);
    for (int64_t q = 0; q < (int64_t)bdofs.size(); ++q) gidx[cur[bdofs[q]]++] = q;   // contrib is laid out like bdofs
  }
  m->h_sdof_ptr = sdp;
  m->fac_doubles = fac;
  int rc = 0;
#define MF_TRY(call)            \
  do {                          \
    if (rc == 0) rc = (call);   \
  } while (0)
  MF_TRY(mf_up(ctx, m, &m->ns, ns));
  MF_TRY(mf_up(ctx, m, &m->nb, nb));
  MF_TRY(mf_up(ctx, m, &m->sdofs, sdofs));
  MF_TRY(mf_up(ctx, m, &m->bdofs, bdofs));
  MF_TRY(mf_up(ctx, m, &m->xoff, xoff));
  MF_TRY(mf_up(ctx, m, &m->loff, loff));
  MF_TRY(mf_up(ctx, m, &m->woff, woff));
  MF_TRY(mf_up(ctx, m, &m->coff, coff));
  MF_TRY(mf_up(ctx, m, &m->sdof_ptr, sdp));
  MF_TRY(mf_up(ctx, m, &m->bdof_ptr, bdp));
  MF_TRY(mf_up(ctx, m, &m->g_ptr, gptr));
  MF_TRY(mf_up(ctx, m, &m->g_idx, gidx));
  auto dalloc = [&](double** p, int64_t count) {
    hipError_t e = hipMalloc((void**)p, (size_t)std::max<int64_t>(count, 1) * 8);
    if (e != hipSuccess) return alfi_set_error(ctx, ALFI_E_HIP, "hipMalloc of %lld doubles: %s", (long long)count, hipGetErrorString(e));
    m->allocs.push_back(*p);
    return 0;
  };
  MF_TRY(dalloc(&m->fac, fac));
  MF_TRY(dalloc(&m->contrib, coff[nt]));
  MF_TRY(dalloc(&m->vloc, L->n));
  MF_TRY(dalloc(&m->tmp_r, L->n));
  MF_TRY(dalloc(&m->tmp_x, L->n));
  // device index arrays of the assembly, kept with the plan
  MfFactorArgs* fa = new MfFactorArgs;
  std::memset(fa, 0, sizeof(*fa));
  m->fa = fa;
  fa->bs = bs;
  fa->ns = m->ns;
  fa->nb = m->nb;
  {
    int32_t* d32 = nullptr;
    int64_t* d64 = nullptr;
    int4* d4 = nullptr;
    MF_TRY(mf_up(ctx, m, &d32, snn)); fa->snn = d32;
    MF_TRY(mf_up(ctx, m, &d32, child)); fa->child = d32;
    MF_TRY(mf_up(ctx, m, &d32, cmap)); fa->cmap = d32;
    MF_TRY(mf_up(ctx, m, &d64, fsb)); fa->fsb_off = d64;
    MF_TRY(mf_up(ctx, m, &d64, fbs)); fa->fbs_off = d64;
    MF_TRY(mf_up(ctx, m, &d64, uo)); fa->u_off = d64;
    MF_TRY(mf_up(ctx, m, &d64, fss)); fa->fss_off = d64;
    MF_TRY(mf_up(ctx, m, &d64, cmp)); fa->cmap_ptr = d64;
    MF_TRY(mf_up(ctx, m, &d64, ssp)); fa->ss_ptr = d64;
    MF_TRY(mf_up(ctx, m, &d64, obp)); fa->ob_ptr = d64;
    MF_TRY(mf_up(ctx, m, &d4, sse)); fa->ss_ent = d4;
    MF_TRY(mf_up(ctx, m, &d4, obe)); fa->ob_ent = d4;
  }
#undef MF_TRY
  m->h_ns = ns;
  m->h_nb = nb;
  m->h_xoff = xoff;
  m->h_loff = loff;
  m->h_woff = woff;
  m->h_fsb = fsb;
  m->h_fbs = fbs;
  m->h_uo = uo;
  m->h_fss = fss;
  m->tmp_doubles = tmp;
  if (rc != 0) {
    mf_free(m);
    return rc;
  }
  *out = m;
  return 0;
}

// numeric factorisation with the operator's current values
static int mf_numeric(alfi_level* L, MfDev* m) {
  alfi_ctx* ctx = L->ctx;
  const int H = m->H;
  const std::vector<int32_t>&ns = m->h_ns, &nb = m->h_nb;
  const std::vector<int64_t>&xoff = m->h_xoff, &loff = m->h_loff, &woff = m->h_woff, &fsb = m->h_fsb, &fbs = m->h_fbs,
                             &uo = m->h_uo, &fss = m->h_fss;
  const int64_t tmp = m->tmp_doubles, fac = m->fac_doubles;
  MfFactorArgs a = *m->fa;
  a.vals = L->A.vals;
  a.flat = L->A.flat;
  int rc = 0;
  {
    hipError_t e = hipMalloc((void**)&a.tmp, (size_t)std::max<int64_t>(tmp, 1) * 8);
    if (e != hipSuccess) return alfi_set_error(ctx, ALFI_E_HIP, "front storage of %.1f GB: %s", 8e-9 * (double)tmp, hipGetErrorString(e));
  }
  if (rc == 0 && hipMemsetAsync(a.tmp, 0, (size_t)tmp * 8, ctx->stream) != hipSuccess) rc = ALFI_E_HIP;
  if (rc == 0 && hipMemsetAsync(m->fac, 0, (size_t)fac * 8, ctx->stream) != hipSuccess) rc = ALFI_E_HIP;
  const dim3 block(256);
  for (int h = 0; h < H && rc == 0; ++h) {
    const int64_t p0 = m->lp[h], p1 = m->lp[h + 1], cnt = p1 - p0;
    // off-diagonal blocks of the fronts: operator entries, then the children's Schur complements
    hipLaunchKernelGGL(mf_scatter_kernel<false>, dim3(32, (unsigned)cnt), block, 0, ctx->stream, p0, a, nullptr, nullptr);
    if (h > 0) {
      hipLaunchKernelGGL(mf_extend_kernel<false>, dim3(32, (unsigned)cnt), block, 0, ctx->stream, p0, 0, a, nullptr, nullptr);
      hipLaunchKernelGGL(mf_extend_kernel<false>, dim3(32, (unsigned)cnt), block, 0, ctx->stream, p0, 1, a, nullptr, nullptr);
    }
    // X = F_ss^-1
    std::vector<int64_t> hptr((size_t)cnt + 1, 0);
    for (int64_t q = 0; q < cnt; ++q) hptr[q + 1] = hptr[q] + ns[p0 + q];
    int64_t* dptr = nullptr;
    rc = mf_up(ctx, nullptr, &dptr, hptr);
    if (rc != 0) break;
    BigSource src;
    MfFill fill;
    fill.a = a;
    fill.level_base = p0;
    fill.ns = m->ns;
    fill.xoff = m->xoff;
    fill.fac = m->fac;
    src.M = &fill;
    rc = big_factor_core(ctx, src, cnt, hptr.data(), dptr, nullptr, nullptr, L->status);
    (void)hipFree(dptr);
    if (rc != 0) break;
    // W = X F_sb and L = F_bs X, each with one residual correction (the entries of X are large and those of the products
    // O(1): W <- W + X (F_sb - F_ss W), L <- L + (F_bs - L F_ss) X), and U = F_bb - F_bs W
    if (m->lvl_max_nb[h] > 0) {
      std::vector<MfGemm> g[7];
      for (auto& v : g) v.resize((size_t)cnt);
      int tsb = 0, tbb = 0;
      for (int64_t q = 0; q < cnt; ++q) {
        const int64_t p = p0 + q;
        const int Ns = mf_pad(ns[p]), Nb = nb[p] ? mf_pad(nb[p]) : 0;
        const double *X = m->fac + xoff[p], *Fss = a.tmp + fss[p];
        double *W = m->fac + woff[p], *Lm = m->fac + loff[p], *Fsb = a.tmp + fsb[p], *Fbs = a.tmp + fbs[p], *U = a.tmp + uo[p];
        g[0][q] = {X, Fsb, W, Ns, Nb, Ns, Ns, Nb, Nb};        // W = X F_sb
        g[1][q] = {Fss, W, Fsb, Ns, Nb, Ns, Ns, Nb, Nb};      // F_sb -= F_ss W
        g[2][q] = {X, Fsb, W, Ns, Nb, Ns, Ns, Nb, Nb};        // W += X F_sb
        g[3][q] = {Fbs, W, U, Nb, Nb, Ns, Ns, Nb, Nb};        // U -= F_bs W
        g[4][q] = {Fbs, X, Lm, Nb, Ns, Ns, Ns, Ns, Ns};       // L = F_bs X
        g[5][q] = {Lm, Fss, Fbs, Nb, Ns, Ns, Ns, Ns, Ns};     // F_bs -= L F_ss
        g[6][q] = {Fbs, X, Lm, Nb, Ns, Ns, Ns, Ns, Ns};       // L += F_bs X
        tsb = std::max(tsb, (Ns / BIG_NB) * (Nb / BIG_NB));
        tbb = std::max(tbb, (Nb / BIG_NB) * (Nb / BIG_NB));
      }
      constexpr bool corr = true;
      const int modes[7] = {0, 1, 2, 1, 0, 1, 2};
      for (int i = 0; i < 7 && rc == 0; ++i) {
        if (!corr && (i == 1 || i == 2 || i == 5 || i == 6)) continue;
        MfGemm* d = nullptr;
        rc = mf_up(ctx, nullptr, &d, g[i]);
        if (rc != 0) break;
        hipLaunchKernelGGL(mf_gemm_kernel, dim3((unsigned)(i == 3 ? tbb : tsb), (unsigned)cnt), block, 0, ctx->stream, d,
                           modes[i]);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
          rc = alfi_set_error(ctx, ALFI_E_HIP, "multifrontal products failed at height %d", h);
        (void)hipFree(d);
      }
    }
  }
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(a.tmp);
  if (rc == 0 && ctx->big_arena_bytes > ((size_t)1 << 30)) {
    (void)hipFree(ctx->big_arena);
    ctx->big_arena = nullptr;
    ctx->big_arena_bytes = 0;
  }
  return rc;
}

// (Re-)factorisation of the level operator.  The plan of an earlier call is reused when it was made for the same ordering
// request (the sparsity of a level never changes; new values arrive through alfi_level_update_values).
int mf_factor(alfi_level* L, const double* coords, int dim, int leaf_nodes) {
  if (leaf_nodes <= 0) leaf_nodes = 64;
  MfDev* m = L->mf;
  if (m && (m->leaf != leaf_nodes || m->with_coords != (coords != nullptr) || m->nnzb != L->A.nnzb || m->n != L->n)) {
    mf_free(m);
    L->mf = m = nullptr;
  }
  if (!m) {
    ALFI_CHECK(mf_analyse(L, coords, dim, leaf_nodes, &m));
  }
  const int rc = mf_numeric(L, m);
  if (rc != 0) {
    mf_free(m);
    L->mf = nullptr;
    return rc;
  }
  L->mf = m;
  return 0;
}
