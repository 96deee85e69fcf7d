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
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
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

// Element matrix Ae (ndof x ndof, dof = a*d + c) = nu*K + gamma*D + adv*N(w) of one affine simplex.
// gc: (d+1, d) gradients of the barycentric coordinates, vc: volume.  S: (nloc,nloc,d+1,d+1) avg d_i phi_a d_j phi_b;
// bI: (nloc,d+1) avg d_i phi_a; T1: (nloc,d+1,nloc,nloc) avg phi_k d_i phi_b phi_a (index order k,i,b,a);
// wk: (nloc, d) wind at the cell's nodes (only read when adv != 0).
struct ElemWork {
  std::vector<double> bvec, cki, t1, r, out, St;
  // S: (nloc, nloc, d+1, d+1), kept here transposed to (d+1, d+1, nloc, nloc)
  ElemWork(int nloc, int d, const double* S = nullptr)
      : bvec(nloc * d), cki((size_t)nloc * (d + 1)), t1((size_t)nloc * nloc), r((size_t)nloc * nloc * (d + 1) * d),
        out((size_t)(1 + d * d) * nloc * nloc), St((size_t)(d + 1) * (d + 1) * nloc * nloc) {
    const int nv = d + 1, nn = nloc * nloc;
    if (S)
      for (int ab = 0; ab < nn; ++ab)
        for (int ij = 0; ij < nv * nv; ++ij) St[(size_t)ij * nn + ab] = S[(size_t)ab * nv * nv + ij];
  }
};

// gfull: coefficient of the FULL grad-div term (div u, div v) of the Scott-Vogelius forms (alfi/solver.py:609-619,
// alfi/transfer.py:295-309), as opposed to gamma, the coefficient of the cell-averaged one of the PkP0 forms.
static void element_matrix(int nloc, int d, const double* gc, double vc, const double* S, const double* bI,
                           const double* T1, const double* wk, double nu, double gamma, double adv, ElemWork& W,
                           double* Ae, double gfull = 0.0) {
  const int nv = d + 1;
  const int ndof = nloc * d;
  double* bvec = W.bvec.data();
  double* cki = W.cki.data();
  if (nu != 0.0 || gfull != 0.0) {
    // out[q][ab] = sum_ij M[ij][q] St[ij][ab]: q = 0 the Gram entry g_i . g_j (-> G_ab), q = 1 + dd d + cc the product
    // g_i^dd g_j^cc (-> h[dd][cc] = int d_dd phi_a d_cc phi_b); the (a, b) index is innermost and contiguous in St
    const int nn = nloc * nloc, nq = 1 + d * d;
    double* out = W.out.data();
    std::fill(out, out + (size_t)nq * nn, 0.0);
    for (int i = 0; i < nv; ++i)
      for (int j = 0; j < nv; ++j) {
        const double* St = W.St.data() + (size_t)(i * nv + j) * nn;
        double m[10];
        double s = 0;
        for (int x = 0; x < d; ++x) s += gc[i * d + x] * gc[j * d + x];
        m[0] = s;
        for (int dd = 0; dd < d; ++dd)
          for (int cc = 0; cc < d; ++cc) m[1 + dd * d + cc] = gc[i * d + dd] * gc[j * d + cc];
        for (int q = 0; q < nq; ++q) {
          const double mq = m[q];
          double* oq = out + (size_t)q * nn;
#pragma omp simd
          for (int ab = 0; ab < nn; ++ab) oq[ab] += mq * St[ab];
        }
      }
    // K_(a,c),(b,dd) = delta_{c,dd} G_ab + int d_dd phi_a d_c phi_b ;  (div, div)_(a,c),(b,dd) = int d_c phi_a d_dd phi_b
    const double fn = nu * vc, fg = gfull * vc;
    for (int a = 0; a < nloc; ++a)
      for (int cc = 0; cc < d; ++cc) {
        double* row = Ae + (size_t)(a * d + cc) * ndof;
        for (int b = 0; b < nloc; ++b) {
          const int ab = a * nloc + b;
          for (int dd = 0; dd < d; ++dd) {
            double v = out[(size_t)(1 + dd * d + cc) * nn + ab];
            if (cc == dd) v += out[ab];
            row[b * d + dd] = fn * v + fg * out[(size_t)(1 + cc * d + dd) * nn + ab];
          }
        }
      }
  } else {
    std::fill(Ae, Ae + (size_t)ndof * ndof, 0.0);
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
  if (adv != 0.0) {
    // Both terms as small dense products with the (b, a) / a index innermost (contiguous in T1: the loops vectorise), then one
    // pass over Ae.  (Until round 5 the second term updated d x d entries of Ae per (b, i, k, a): 100 k dependent
    // read-modify-writes per cell, 4/5 of the time of the advective assembly.)
    //   t1[b][a]       = sum_{k,i} (w_k . g_i) T1[k,i,b,a]
    //   r[b][i][c][a]  = sum_k w_k^c T1[b,i,k,a]
    //   Ae[(a,c),(b,dd)] += adv vol (delta_{c,dd} t1[b][a] + sum_i r[b][i][c][a] g_i^dd)
    const int nn = nloc * nloc;
    double* t1 = W.t1.data();
    double* r = W.r.data();
    for (int k = 0; k < nloc; ++k)
      for (int i = 0; i < nv; ++i) {
        double s = 0;
        for (int x = 0; x < d; ++x) s += wk[k * d + x] * gc[i * d + x];
        cki[k * nv + i] = s;
      }
    std::fill(t1, t1 + nn, 0.0);
    for (int ki = 0; ki < nloc * nv; ++ki) {
      const double cc1 = cki[ki];
      const double* T = T1 + (size_t)ki * nn;  // T[b][a]
#pragma omp simd
      for (int q = 0; q < nn; ++q) t1[q] += cc1 * T[q];
    }
    std::fill(r, r + (size_t)nn * nv * d, 0.0);
    for (int bi = 0; bi < nloc * nv; ++bi) {
      const double* T = T1 + (size_t)bi * nn;  // T[k][a]
      double* rb = r + (size_t)bi * d * nloc;   // rb[c][a]
      for (int k = 0; k < nloc; ++k) {
        const double* Tk = T + k * nloc;
        for (int cc = 0; cc < d; ++cc) {
          const double wc = wk[k * d + cc];
          double* rc = rb + cc * nloc;
#pragma omp simd
          for (int a = 0; a < nloc; ++a) rc[a] += wc * Tk[a];
        }
      }
    }
    const double f = adv * vc;
    for (int a = 0; a < nloc; ++a)
      for (int cc = 0; cc < d; ++cc) {
        double* row = Ae + (size_t)(a * d + cc) * ndof;
        for (int b = 0; b < nloc; ++b) {
          double acc[3] = {0.0, 0.0, 0.0};
          for (int i = 0; i < nv; ++i) {
            const double rv = r[((size_t)(b * nv + i) * d + cc) * nloc + a];
            for (int dd = 0; dd < d; ++dd) acc[dd] += rv * gc[i * d + dd];
          }
          acc[cc] += t1[b * nloc + a];
          for (int dd = 0; dd < d; ++dd) row[b * d + dd] += f * acc[dd];
        }
      }
  }
}

// vals (nnzb, d, d) += nu*K + gamma*D + adv*N(w), scattered into a BSR matrix whose block rows are the nodes mapped
// through row_map (nullptr: identity; row_map[node] < 0: row skipped) and whose block columns are all nodes.
int alfi_host_assemble_bsr(int64_t ncell, int nloc, int d, const int32_t* cell_nodes, const double* g,
                           const double* vol, const double* S, const double* bI, const double* T1, const double* w,
                           double nu, double gamma, double adv, const int32_t* row_map, const int32_t* rowptr,
                           const int32_t* colidx, double* vals, double gamma_full) {
  const int nv = d + 1;
  const int ndof = nloc * d;
  const bool do_adv = (adv != 0.0) && (w != nullptr);
  int err = 0;
#pragma omp parallel
  {
    ElemWork W(nloc, d, S);
    std::vector<double> Ae((size_t)ndof * ndof), wk((size_t)nloc * d);
#pragma omp for schedule(dynamic, 256)
    for (int64_t c = 0; c < ncell; ++c) {
      const int32_t* cn = cell_nodes + c * nloc;
      if (row_map) {
        bool any = false;
        for (int a = 0; a < nloc; ++a) any |= row_map[cn[a]] >= 0;
        if (!any) continue;
      }
      if (do_adv)
        for (int k = 0; k < nloc; ++k)
          for (int x = 0; x < d; ++x) wk[k * d + x] = w[(int64_t)cn[k] * d + x];
      element_matrix(nloc, d, g + c * nv * d, vol[c], S, bI, T1, wk.data(), nu, gamma, do_adv ? adv : 0.0, W,
                     Ae.data(), gamma_full);
      for (int a = 0; a < nloc; ++a) {
        const int32_t ra = row_map ? row_map[cn[a]] : cn[a];
        if (ra < 0) continue;
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

// Contributor lists for the DEVICE assembly of the level operators (csrc/kernels_assemble.hip): for every block (r, c) of the
// BSR sparsity the (cell, b * nloc + a) pairs with r = node a and c = node b of the cell, cells ascending and, inside a cell, b
// ascending -- a fixed order, so that the device sums every operator entry in the same order on every run (no atomics).
// cptr: (nnzb + 1) int64.  Call with ccell == nullptr to fill cptr only (counting pass).
// nindex >= nnode: cell_nodes may name nodes beyond the nnode rows of the sparsity (a rank's cells reach nodes it holds no
// operator row for); partial != 0: pairs whose block is not in the sparsity are skipped instead of being an error (ghost rows
// restricted to local columns).
int alfi_host_contributors(int64_t ncell, int nloc, const int32_t* cell_nodes, int64_t nnode, const int32_t* rowptr,
                           const int32_t* colidx, int64_t* cptr, int32_t* ccell, uint16_t* cba, int64_t nindex, int partial) {
  if (nindex < nnode) nindex = nnode;
  std::vector<int64_t> nptr(nindex + 1, 0);                 // node -> cells (ascending)
  for (int64_t c = 0; c < ncell; ++c)
    for (int a = 0; a < nloc; ++a) nptr[cell_nodes[c * nloc + a] + 1]++;
  for (int64_t i = 0; i < nindex; ++i) nptr[i + 1] += nptr[i];
  std::vector<int32_t> ncells(nptr[nindex]);
  {
    std::vector<int64_t> fill(nptr.begin(), nptr.end() - 1);
    for (int64_t c = 0; c < ncell; ++c)
      for (int a = 0; a < nloc; ++a) ncells[fill[cell_nodes[c * nloc + a]]++] = (int32_t)c;
  }
  const int64_t nnzb = rowptr[nnode];
  const bool counting = (ccell == nullptr);
  int err = 0;
  if (counting) std::fill(cptr, cptr + nnzb + 1, (int64_t)0);
  // the blocks of block row r are touched by the thread that owns r only: no atomics, a fixed order
#pragma omp parallel
  {
    std::vector<int64_t> cursor;
    std::vector<std::pair<int32_t, int64_t>> byc;        // the row's (column, block) pairs ascending: a rank's local operator
                                                         // numbers its columns owned-first, so its rows are not sorted
#pragma omp for schedule(dynamic, 512)
    for (int64_t r = 0; r < nnode; ++r) {
      const int64_t lo = rowptr[r], hi = rowptr[r + 1];
      if (!counting) cursor.assign(cptr + lo, cptr + hi);
      byc.resize((size_t)(hi - lo));
      for (int64_t k = lo; k < hi; ++k) byc[(size_t)(k - lo)] = std::make_pair(colidx[k], k);
      if (!std::is_sorted(byc.begin(), byc.end())) std::sort(byc.begin(), byc.end());
      for (int64_t q = nptr[r]; q < nptr[r + 1]; ++q) {
        const int32_t cell = ncells[q];
        const int32_t* cn = cell_nodes + (int64_t)cell * nloc;
        int a = 0;
        while (a < nloc && cn[a] != r) ++a;
        for (int b = 0; b < nloc; ++b) {
          const auto it = std::lower_bound(byc.begin(), byc.end(), std::make_pair(cn[b], (int64_t)-1));
          const int64_t pos = (it != byc.end() && it->first == cn[b]) ? it->second : hi;
          if (pos >= hi) {
            if (!partial) err = 1;
            continue;
          }
          if (counting) {
            cptr[pos + 1]++;
          } else {
            const int64_t at = cursor[pos - lo]++;
            ccell[at] = cell;
            cba[at] = (uint16_t)(b * nloc + a);
          }
        }
      }
    }
  }
  if (counting)
    for (int64_t k = 0; k < nnzb; ++k) cptr[k + 1] += cptr[k];
  return err ? -2 : 0;
}

// Dense interior blocks of the Schoeberl transfer, assembled directly: block `blk` (= coarse cell) receives the
// contributions of its nch children (fine cells blk*nch .. blk*nch+nch-1) restricted to the block's interior dofs.
// blk_local[node] = position of the node inside its block (0..m/d-1) or -1.  KII, DII: (nblk, m, m) row-major.
int alfi_host_interior_blocks(int64_t nblk, int nch, int nloc, int d, const int32_t* cell_nodes, const double* g,
                              const double* vol, const double* S, const double* bI, const int32_t* blk_local, int m,
                              double* KII, double* DII, int full_div) {
  const int nv = d + 1;
  const int ndof = nloc * d;
#pragma omp parallel
  {
    ElemWork W(nloc, d, S);
    std::vector<double> Ke((size_t)ndof * ndof), De((size_t)ndof * ndof);
#pragma omp for schedule(static)
    for (int64_t blk = 0; blk < nblk; ++blk) {
      double* Kb = KII + blk * m * m;
      double* Db = DII + blk * m * m;
      std::fill(Kb, Kb + (size_t)m * m, 0.0);
      std::fill(Db, Db + (size_t)m * m, 0.0);
      for (int ch = 0; ch < nch; ++ch) {
        const int64_t c = blk * nch + ch;
        const int32_t* cn = cell_nodes + c * nloc;
        element_matrix(nloc, d, g + c * nv * d, vol[c], S, bI, nullptr, nullptr, 1.0, 0.0, 0.0, W, Ke.data());
        element_matrix(nloc, d, g + c * nv * d, vol[c], S, bI, nullptr, nullptr, 0.0, full_div ? 0.0 : 1.0, 0.0, W,
                       De.data(), full_div ? 1.0 : 0.0);
        for (int a = 0; a < nloc; ++a) {
          const int la = blk_local[cn[a]];
          if (la < 0) continue;
          for (int b = 0; b < nloc; ++b) {
            const int lb = blk_local[cn[b]];
            if (lb < 0) continue;
            for (int cc = 0; cc < d; ++cc)
              for (int dd = 0; dd < d; ++dd) {
                const size_t src = (size_t)(a * d + cc) * ndof + b * d + dd;
                const size_t dst = (size_t)(la * d + cc) * m + lb * d + dd;
                Kb[dst] += Ke[src];
                Db[dst] += De[src];
              }
          }
        }
      }
    }
  }
  return 0;
}

// Transpose of a BSR matrix (counting sort; output columns ascending because rows are visited in order).
int alfi_host_bsr_transpose(int64_t nbrows, int64_t nbcols, int bs, const int32_t* rowptr, const int32_t* colidx,
                            const double* vals, int32_t* rowptr_t, int32_t* colidx_t, double* vals_t) {
  const int64_t nnzb = rowptr[nbrows];
  std::fill(rowptr_t, rowptr_t + nbcols + 1, 0);
  for (int64_t k = 0; k < nnzb; ++k) rowptr_t[colidx[k] + 1]++;
  for (int64_t j = 0; j < nbcols; ++j) rowptr_t[j + 1] += rowptr_t[j];
  std::vector<int32_t> fill(rowptr_t, rowptr_t + nbcols);
  std::vector<int32_t> src(nnzb);
  for (int64_t r = 0; r < nbrows; ++r)
    for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      const int32_t q = fill[colidx[k]]++;
      colidx_t[q] = (int32_t)r;
      src[q] = (int32_t)k;
    }
  const int bb = bs * bs;
#pragma omp parallel for schedule(static)
  for (int64_t q = 0; q < nnzb; ++q) {
    const double* s = vals + (int64_t)src[q] * bb;
    double* t = vals_t + q * bb;
    for (int i = 0; i < bs; ++i)
      for (int j = 0; j < bs; ++j) t[j * bs + i] = s[i * bs + j];
  }
  return 0;
}

// Dirichlet rows and columns -> identity (what firedrake.assemble(a, bcs=...) produces).  bcmask: (all nodes * d) bytes.
// row_ids: node of each of the nrow block rows (a row subset, rank-local generation), or nullptr: row r is node r.
int alfi_host_apply_bc_bsr(int64_t nrow, int d, const int32_t* rowptr, const int32_t* colidx, double* vals,
                           const uint8_t* bcmask, const int32_t* row_ids) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nrow; ++r) {
    const int64_t rnode = row_ids ? row_ids[r] : r;
    for (int64_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
      const int64_t cnode = colidx[p];
      double* blk = vals + p * d * d;
      for (int cc = 0; cc < d; ++cc)
        for (int dd = 0; dd < d; ++dd) {
          const bool rb = bcmask[rnode * d + cc], cb = bcmask[cnode * d + dd];
          if (rb || cb) blk[cc * d + dd] = (rb && cb && rnode == cnode && cc == dd) ? 1.0 : 0.0;
        }
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
// launchers such as torch.distributed.run export OMP_NUM_THREADS=1; the generator's own thread count is set explicitly
int alfi_host_set_num_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
}


// ---------------------------------------------------------------------------------------------------------------------
// SUPG stabilisation of the momentum equation (alfi/stabilisation.py:47-97 with the Shakib-Hughes-Johan coefficient,
// alfi/solver.py:204-234: stabilisation_form = weight * beta * inner(Lu, dot(grad(v), u)) * dx(degree = 2k), state = u):
//   Lu   = -nu div(2 sym grad u) + (grad u) u   (+ grad p, zero for the piecewise constant pressure)
//   beta = (4 u.u / h^2 + magic (4 nu / h^2)^2)^(-1/2)
// by quadrature (beta is not polynomial).  F (n dofs, may be NULL) += the residual contribution; vals (BSR, may be NULL)
// += its Newton linearisation about u:
//   dF[a,i; b,j] = int wgt [ dbeta_j (Lu)_i s_a + beta (dLu)_ij s_a + beta (Lu)_i phi_b d_j phi_a ],
//   s_a = u.grad phi_a,  dbeta_j = -4 beta^3 u_j phi_b / h^2,
//   (dLu)_ij = -nu (delta_ij Lap phi_b + d_i d_j phi_b) + delta_ij s_b + phi_b d_j u_i.
// phi (nq, nloc), dphi (nq, nloc, d+1), d2phi (nq, nloc, d+1, d+1): basis and its derivatives w.r.t. the barycentric
// coordinates at the quadrature points; wq (nq) sums to 1; h (ncell): cell size; U (nnode, d): state.
// ---------------------------------------------------------------------------------------------------------------------
int alfi_host_supg(int64_t ncell, int nloc, int d, const int32_t* cell_nodes, const double* g, const double* vol,
                   const double* h, int nq, const double* wq, const double* phi, const double* dphi, const double* d2phi,
                   const double* U, double nu, double weight, double magic, const int32_t* rowptr, const int32_t* colidx,
                   double* vals, double* F) {
  const int nv = d + 1;
  const int ndof = nloc * d;
  int err = 0;
#pragma omp parallel
  {
    std::vector<double> Ae((size_t)ndof * ndof), Fe(ndof), Uk((size_t)nloc * d), gp((size_t)nloc * d),
        hs((size_t)nloc * d * d), lap(nloc), s(nloc);
#pragma omp for schedule(dynamic, 64)
    for (int64_t c = 0; c < ncell; ++c) {
      const int32_t* cn = cell_nodes + c * nloc;
      const double* gc = g + c * nv * d;
      for (int a = 0; a < nloc; ++a)
        for (int x = 0; x < d; ++x) Uk[a * d + x] = U[(int64_t)cn[a] * d + x];
      std::fill(Ae.begin(), Ae.end(), 0.0);
      std::fill(Fe.begin(), Fe.end(), 0.0);
      const double h2 = h[c] * h[c];
      for (int q = 0; q < nq; ++q) {
        const double* ph = phi + (size_t)q * nloc;
        // physical gradients and Hessians of the basis at this point
        for (int a = 0; a < nloc; ++a) {
          const double* da = dphi + ((size_t)q * nloc + a) * nv;
          const double* ha = d2phi + ((size_t)q * nloc + a) * nv * nv;
          for (int x = 0; x < d; ++x) {
            double t = 0.0;
            for (int i = 0; i < nv; ++i) t += da[i] * gc[i * d + x];
            gp[a * d + x] = t;
          }
          double l = 0.0;
          for (int x = 0; x < d; ++x)
            for (int y = 0; y < d; ++y) {
              double t = 0.0;
              for (int i = 0; i < nv; ++i)
                for (int k = 0; k < nv; ++k) t += ha[i * nv + k] * gc[i * d + x] * gc[k * d + y];
              hs[((size_t)a * d + x) * d + y] = t;
              if (x == y) l += t;
            }
          lap[a] = l;
        }
        // state at the point
        double u[3] = {0, 0, 0}, Gu[3][3] = {{0}}, Lu[3] = {0, 0, 0};
        for (int a = 0; a < nloc; ++a)
          for (int i = 0; i < d; ++i) {
            const double ui = Uk[a * d + i];
            u[i] += ph[a] * ui;
            for (int x = 0; x < d; ++x) Gu[i][x] += gp[a * d + x] * ui;
            Lu[i] -= nu * lap[a] * ui;                                             // -nu Lap u_i
            for (int j = 0; j < d; ++j) Lu[j] -= nu * hs[((size_t)a * d + j) * d + i] * ui;   // -nu d_j div u
          }
        for (int i = 0; i < d; ++i)
          for (int x = 0; x < d; ++x) Lu[i] += u[x] * Gu[i][x];
        double uu = 0.0;
        for (int i = 0; i < d; ++i) uu += u[i] * u[i];
        const double vis = 4.0 * nu / h2;
        const double beta = 1.0 / std::sqrt(4.0 * uu / h2 + magic * vis * vis);
        for (int a = 0; a < nloc; ++a) {
          double t = 0.0;
          for (int x = 0; x < d; ++x) t += u[x] * gp[a * d + x];
          s[a] = t;
        }
        const double wt = wq[q] * vol[c] * weight;
        for (int a = 0; a < nloc; ++a)
          for (int i = 0; i < d; ++i) Fe[a * d + i] += wt * beta * Lu[i] * s[a];
        if (vals) {
          const double b3 = -4.0 * beta * beta * beta / h2;
          for (int a = 0; a < nloc; ++a)
            for (int i = 0; i < d; ++i) {
              double* row = Ae.data() + (size_t)(a * d + i) * ndof;
              for (int b = 0; b < nloc; ++b)
                for (int j = 0; j < d; ++j) {
                  double dL = -nu * hs[((size_t)b * d + i) * d + j] + ph[b] * Gu[i][j];
                  if (i == j) dL += -nu * lap[b] + s[b];
                  row[b * d + j] += wt * (b3 * u[j] * ph[b] * Lu[i] * s[a] + beta * dL * s[a] +
                                          beta * Lu[i] * ph[b] * gp[a * d + j]);
                }
            }
        }
      }
      if (F)
        for (int a = 0; a < nloc; ++a)
          for (int i = 0; i < d; ++i) {
#pragma omp atomic
            F[(int64_t)cn[a] * d + i] += Fe[a * d + i];
          }
      if (vals)
        for (int a = 0; a < nloc; ++a) {
          const int64_t lo = rowptr[cn[a]], hi = rowptr[cn[a] + 1];
          for (int b = 0; b < nloc; ++b) {
            const int64_t pos = find_col(colidx, lo, hi, cn[b]);
            if (pos >= hi || colidx[pos] != cn[b]) {
              err = 1;
              continue;
            }
            double* dst = vals + pos * d * d;
            for (int cc = 0; cc < d; ++cc)
              for (int dd = 0; dd < d; ++dd) {
#pragma omp atomic
                dst[cc * d + dd] += Ae[(size_t)(a * d + cc) * ndof + b * d + dd];
              }
          }
        }
    }
  }
  return err ? -2 : 0;
}

}  // extern "C"
