// Host-side (CPU, OpenMP) operator generator for the synthetic ldc2d / ldc3d workloads.
//
// The reference obtains its level operators from Firedrake/TSFC/PyOP2 assembly of the UFL forms in
// alfi/solver.py:565-568 (velocity block of the Newton linearisation) and alfi/transfer.py:319-332 (the symmetric
// form used by the Schoeberl transfer).  Neither is available here, so this file assembles the same bilinear form
//
//     a(u, v) = nu (2 sym grad u, grad v) + gamma (cell_avg(div u), div v) + adv ((w . grad) u + (u . grad) w, v)
//
// for nodal vector elements on affine simplices directly into a block-CSR (block size d) matrix, from reference-cell
// tensors computed in alfi_amd/elements.py.  This is input generation (setup time), not part of the measured hot path.
//
// C ABI, plain pointers; called through ctypes from alfi_amd/_hostlib.py.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>
#include <omp.h>

extern "C" {

// Node-to-node adjacency (two nodes are coupled iff they share a cell), sorted columns.
// Pass colidx == nullptr to only fill rowptr (counting pass).
int alfi_host_node_graph(int64_t ncell, int nloc, const int32_t* cell_nodes, int64_t nnode, int32_t* rowptr,
                         int32_t* colidx) {
  // node -> cells
  std::vector<int64_t> nptr(nnode + 1, 0);
  for (int64_t c = 0; c < ncell; ++c)
    for (int a = 0; a < nloc; ++a) nptr[cell_nodes[c * nloc + a] + 1]++;
  for (int64_t i = 0; i < nnode; ++i) nptr[i + 1] += nptr[i];
  std::vector<int32_t> ncells(nptr[nnode]);
  {
    std::vector<int64_t> fill(nptr.begin(), nptr.end() - 1);
    for (int64_t c = 0; c < ncell; ++c)
      for (int a = 0; a < nloc; ++a) ncells[fill[cell_nodes[c * nloc + a]]++] = (int32_t)c;
  }
  const bool counting = (colidx == nullptr);
  if (counting) rowptr[0] = 0;
#pragma omp parallel
  {
    std::vector<int32_t> buf;
#pragma omp for schedule(dynamic, 1024)
    for (int64_t i = 0; i < nnode; ++i) {
      buf.clear();
      for (int64_t p = nptr[i]; p < nptr[i + 1]; ++p) {
        const int32_t* cn = cell_nodes + (int64_t)ncells[p] * nloc;
        buf.insert(buf.end(), cn, cn + nloc);
      }
      std::sort(buf.begin(), buf.end());
      auto last = std::unique(buf.begin(), buf.end());
      int64_t cnt = last - buf.begin();
      if (counting)
        rowptr[i + 1] = (int32_t)cnt;
      else
        std::copy(buf.begin(), last, colidx + rowptr[i]);
    }
  }
  if (counting) {
    int64_t acc = 0;
    for (int64_t i = 0; i < nnode; ++i) {
      acc += rowptr[i + 1];
      if (acc > INT32_MAX) return -1;
      rowptr[i + 1] = (int32_t)acc;
    }
  }
  return 0;
}

static inline int64_t find_col(const int32_t* colidx, int64_t lo, int64_t hi, int32_t col) {
  const int32_t* p = std::lower_bound(colidx + lo, colidx + hi, col);
  return p - colidx;
}

// vals (nnzb, d, d) += nu*K + gamma*D + adv*N(w).  g: (ncell, d+1, d) gradients of the barycentric coordinates,
// vol: (ncell).  S: (nloc,nloc,d+1,d+1) avg d_i phi_a d_j phi_b; bI: (nloc,d+1) avg d_i phi_a;
// T1: (nloc,d+1,nloc,nloc) avg phi_k d_i phi_b phi_a (index order k,i,b,a).  w: (nnode, d) wind or nullptr.
int alfi_host_assemble_bsr(int64_t ncell, int nloc, int d, const int32_t* cell_nodes, const double* g,
                           const double* vol, const double* S, const double* bI, const double* T1, const double* w,
                           double nu, double gamma, double adv, const int32_t* rowptr, const int32_t* colidx,
                           double* vals) {
  const int nv = d + 1;
  const int ndof = nloc * d;
  const bool do_adv = (adv != 0.0) && (w != nullptr);
  int err = 0;
#pragma omp parallel
  {
    std::vector<double> Ae((size_t)ndof * ndof), G((size_t)nv * nv), bvec(ndof), cki((size_t)nloc * nv),
        wk((size_t)nloc * d), Hij((size_t)nv * nv * d * d);
#pragma omp for schedule(dynamic, 256)
    for (int64_t c = 0; c < ncell; ++c) {
      const double* gc = g + c * nv * d;
      const double vc = vol[c];
      const int32_t* cn = cell_nodes + c * nloc;
      std::fill(Ae.begin(), Ae.end(), 0.0);
      // geometry products: G[i][j] = g_i . g_j ; Hij[i][j][dd][cc] = g_i^dd g_j^cc
      for (int i = 0; i < nv; ++i)
        for (int j = 0; j < nv; ++j) {
          double s = 0;
          for (int x = 0; x < d; ++x) s += gc[i * d + x] * gc[j * d + x];
          G[i * nv + j] = s;
          for (int dd = 0; dd < d; ++dd)
            for (int cc = 0; cc < d; ++cc) Hij[((i * nv + j) * d + dd) * d + cc] = gc[i * d + dd] * gc[j * d + cc];
        }
      if (nu != 0.0) {
        for (int a = 0; a < nloc; ++a)
          for (int b = 0; b < nloc; ++b) {
            const double* Sab = S + ((size_t)(a * nloc + b)) * nv * nv;
            double gab = 0;
            double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // h[dd][cc] = int d_dd phi_a d_cc phi_b
            for (int i = 0; i < nv; ++i)
              for (int j = 0; j < nv; ++j) {
                const double s = Sab[i * nv + j];
                if (s == 0.0) continue;
                gab += s * G[i * nv + j];
                const double* hh = &Hij[((i * nv + j) * d) * d];
                for (int q = 0; q < d * d; ++q) h[q] += s * hh[q];
              }
            // K_(a,c),(b,dd) = delta_{c,dd} G_ab + int d_dd phi_a d_c phi_b
            for (int cc = 0; cc < d; ++cc)
              for (int dd = 0; dd < d; ++dd) {
                double v = h[dd * d + cc];
                if (cc == dd) v += gab;
                Ae[(size_t)(a * d + cc) * ndof + b * d + dd] += nu * vc * v;
              }
          }
      }
      if (gamma != 0.0) {
        for (int a = 0; a < nloc; ++a)
          for (int cc = 0; cc < d; ++cc) {
            double s = 0;
            for (int i = 0; i < nv; ++i) s += gc[i * d + cc] * bI[a * nv + i];
            bvec[a * d + cc] = s;  // (1/vol) int d_cc phi_a
          }
        const double f = gamma * vc;
        for (int p = 0; p < ndof; ++p) {
          const double bp = f * bvec[p];
          if (bp == 0.0) continue;
          for (int q = 0; q < ndof; ++q) Ae[(size_t)p * ndof + q] += bp * bvec[q];
        }
      }
      if (do_adv) {
        for (int k = 0; k < nloc; ++k)
          for (int x = 0; x < d; ++x) wk[k * d + x] = w[(int64_t)cn[k] * d + x];
        // term 1: delta_{cd} * vol * sum_{k,i} (w_k . g_i) T1[k,i,b,a]
        for (int k = 0; k < nloc; ++k)
          for (int i = 0; i < nv; ++i) {
            double s = 0;
            for (int x = 0; x < d; ++x) s += wk[k * d + x] * gc[i * d + x];
            cki[k * nv + i] = s;
          }
        const double f = adv * vc;
        for (int k = 0; k < nloc; ++k)
          for (int i = 0; i < nv; ++i) {
            const double cc1 = f * cki[k * nv + i];
            const double* T = T1 + ((size_t)(k * nv + i)) * nloc * nloc;  // T[b][a]
            if (cc1 != 0.0) {
              for (int b = 0; b < nloc; ++b)
                for (int a = 0; a < nloc; ++a) {
                  const double t = cc1 * T[b * nloc + a];
                  for (int x = 0; x < d; ++x) Ae[(size_t)(a * d + x) * ndof + b * d + x] += t;
                }
            }
            // term 2: (a,c),(b,dd) += vol * w_k^c g_i^dd * T1[b,i,k,a]   (here the loop variable k is "k", and
            // T1 is indexed [b][i][k][a])
          }
        for (int b = 0; b < nloc; ++b)
          for (int i = 0; i < nv; ++i) {
            const double* T = T1 + ((size_t)(b * nv + i)) * nloc * nloc;  // T[k][a]
            for (int k = 0; k < nloc; ++k)
              for (int a = 0; a < nloc; ++a) {
                const double t = f * T[k * nloc + a];
                if (t == 0.0) continue;
                for (int cc = 0; cc < d; ++cc) {
                  const double tw = t * wk[k * d + cc];
                  for (int dd = 0; dd < d; ++dd) Ae[(size_t)(a * d + cc) * ndof + b * d + dd] += tw * gc[i * d + dd];
                }
              }
          }
      }
      // scatter
      for (int a = 0; a < nloc; ++a) {
        const int32_t ra = cn[a];
        const int64_t lo = rowptr[ra], hi = rowptr[ra + 1];
        for (int b = 0; b < nloc; ++b) {
          const int64_t pos = find_col(colidx, lo, hi, cn[b]);
          if (pos >= hi || colidx[pos] != cn[b]) {
            err = 1;
            continue;
          }
          double* dst = vals + pos * d * d;
          for (int cc = 0; cc < d; ++cc)
            for (int dd = 0; dd < d; ++dd) {
              const double v = Ae[(size_t)(a * d + cc) * ndof + b * d + dd];
#pragma omp atomic
              dst[cc * d + dd] += v;
            }
        }
      }
    }
  }
  return err ? -2 : 0;
}

// Dirichlet rows and columns -> identity (what firedrake.assemble(a, bcs=...) produces).  bcmask: (nnode*d) bytes.
int alfi_host_apply_bc_bsr(int64_t nnode, int d, const int32_t* rowptr, const int32_t* colidx, double* vals,
                           const uint8_t* bcmask) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nnode; ++r)
    for (int64_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
      const int64_t cnode = colidx[p];
      double* blk = vals + p * d * d;
      for (int cc = 0; cc < d; ++cc)
        for (int dd = 0; dd < d; ++dd) {
          const bool rb = bcmask[r * d + cc], cb = bcmask[cnode * d + dd];
          if (rb || cb) blk[cc * d + dd] = (rb && cb && r == cnode && cc == dd) ? 1.0 : 0.0;
        }
    }
  return 0;
}

// Dense principal sub-blocks A[dofs_b, dofs_b] of a BSR matrix.  blk_ptr: (nblk+1) offsets into blk_dofs;
// out_ptr: (nblk+1) offsets into out (out_ptr[b+1]-out_ptr[b] == n_b^2), row-major blocks.
int alfi_host_extract_blocks(int d, const int32_t* rowptr, const int32_t* colidx, const double* vals, int64_t nblk,
                             const int64_t* blk_ptr, const int32_t* blk_dofs, const int64_t* out_ptr, double* out) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t b = 0; b < nblk; ++b) {
    const int32_t* dofs = blk_dofs + blk_ptr[b];
    const int64_t n = blk_ptr[b + 1] - blk_ptr[b];
    double* o = out + out_ptr[b];
    std::fill(o, o + n * n, 0.0);
    for (int64_t i = 0; i < n; ++i) {
      const int64_t r = dofs[i] / d, rc = dofs[i] % d;
      for (int64_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
        const int64_t c0 = (int64_t)colidx[p] * d;
        // dofs are sorted ascending: locate the first dof >= c0
        const int32_t* q = std::lower_bound(dofs, dofs + n, (int32_t)c0);
        for (; q < dofs + n && *q < c0 + d; ++q) o[i * n + (q - dofs)] = vals[p * d * d + rc * d + (*q - c0)];
      }
    }
  }
  return 0;
}

int alfi_host_num_threads() { return omp_get_max_threads(); }

}  // extern "C"
