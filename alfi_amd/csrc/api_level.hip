// C ABI of libalfi_hip.so (include/alfi_hip.h), the level operator: creation, partition views, device-side assembly, products.
// (One file per concern since round 5: api_ctx / api_level / api_patches / api_smoother / api_cycles / api_saddle; the helpers they
// share are declared in api_internal.h.)
#include "api_internal.h"

// ---- level ---------------------------------------------------------------------------------------------------------------
int alfi_level_create(alfi_ctx* ctx, int64_t nbrows, int bs, const int32_t* browptr, const int32_t* bcolidx,
                      const double* bvals, const int32_t* bc_dofs, int64_t nbc, alfi_level** out) {
  // bvals == NULL: the operator starts as zeros and is formed on the device (alfi_level_set_assembly + alfi_level_assemble)
  if (!ctx || !out || !browptr || !bcolidx) return alfi_set_error(ctx, ALFI_E_ARG, "NULL argument");
  if (bs != 2 && bs != 3) return alfi_set_error(ctx, ALFI_E_ARG, "block size must be 2 or 3, got %d", bs);
  if (nbrows * bs > INT32_MAX) return alfi_set_error(ctx, ALFI_E_ARG, "level too large for int32 dof indices");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  alfi_level* L = new alfi_level();
  L->ctx = ctx;
  L->id = ctx->next_level_id++;
  L->bs = bs;
  L->n = nbrows * bs;
  alfi_bsr_host h{nbrows, nbrows, browptr, bcolidx, bvals};
  int rc = upload_bsr(ctx, &L->A, &h, bs);
  if (rc == 0) rc = dev_upload(ctx, &L->bc_dofs, bc_dofs, nbc);
  if (rc == 0) {
    std::vector<uint8_t> mask((size_t)std::max<int64_t>(L->n, 1), 0);
    for (int64_t i = 0; i < nbc && rc == 0; ++i) {
      if (bc_dofs[i] < 0 || bc_dofs[i] >= L->n) rc = alfi_set_error(ctx, ALFI_E_ARG, "Dirichlet dof %d out of range", bc_dofs[i]);
      else mask[bc_dofs[i]] = 1;
    }
    if (rc == 0) rc = dev_upload(ctx, &L->bc_mask, mask.data(), (int64_t)mask.size());
  }
  if (rc == 0) rc = dev_alloc(ctx, &L->status, 1);
  if (rc == 0 && hipMemsetAsync(L->status, 0, sizeof(int), ctx->stream) != hipSuccess) rc = ALFI_E_HIP;
  L->nbc = nbc;
  if (rc != 0) {
    alfi_level_destroy(L);
    return rc;
  }
  L->n_own = L->n;
  L->A_own = L->A;
  for (int64_t i = 0; i < nbrows; ++i) L->max_row_blocks = std::max<int>(L->max_row_blocks, browptr[i + 1] - browptr[i]);
  *out = L;
  return 0;
}

int alfi_level_set_partition(alfi_level* L, int64_t nb_owned, int distributed, int64_t nsend,
                             const int32_t* send_nodes, double* d_sendbuf, double* d_recvbuf, int64_t nb_ghost) {
  alfi_ctx* ctx = L->ctx;
  if (nb_owned < 0 || nb_ghost < 0 || nb_owned + nb_ghost != L->A.nbrows)
    return alfi_set_error(ctx, ALFI_E_ARG, "owned (%lld) + ghost (%lld) nodes != %lld block rows", (long long)nb_owned,
                          (long long)nb_ghost, (long long)L->A.nbrows);
  if (nsend < 0 || (nsend > 0 && !send_nodes) || (!d_sendbuf) != (!d_recvbuf))
    return alfi_set_error(ctx, ALFI_E_ARG, "bad halo arguments (pass both buffers or neither)");
  for (int64_t i = 0; i < nsend; ++i)
    if (send_nodes[i] < 0 || send_nodes[i] >= nb_owned)
      return alfi_set_error(ctx, ALFI_E_ARG, "send node %d is not an owned node", send_nodes[i]);
  if (L->patch_ptr || L->V) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_set_partition must precede patches");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(L->halo_send_nodes);
  dev_free(L->rev_nodes);
  dev_free(L->rev_ptr);
  dev_free(L->rev_pos);
  L->halo_send_nodes = L->rev_nodes = L->rev_ptr = L->rev_pos = nullptr;
  if (L->own_halo_bufs) {
    dev_free(L->halo_sendbuf);
    dev_free(L->halo_recvbuf);
  }
  L->halo_sendbuf = L->halo_recvbuf = nullptr;
  L->own_halo_bufs = false;
  L->nbr_rank.clear();
  if (!d_sendbuf) {                       // the library's own buffers (native transport)
    ALFI_CHECK(dev_alloc(ctx, &d_sendbuf, nsend * L->bs));
    ALFI_CHECK(dev_alloc(ctx, &d_recvbuf, nb_ghost * L->bs));
    L->own_halo_bufs = true;
  }
  // reverse-add plan: positions of the send buffer grouped by node, in buffer order (fixed summation order)
  std::vector<int32_t> order(nsend);
  for (int64_t i = 0; i < nsend; ++i) order[i] = (int32_t)i;
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return send_nodes[a] < send_nodes[b]; });
  std::vector<int32_t> rev_nodes, rev_ptr;
  for (int64_t i = 0; i < nsend; ++i) {
    if (i == 0 || send_nodes[order[i]] != send_nodes[order[i - 1]]) {
      rev_nodes.push_back(send_nodes[order[i]]);
      rev_ptr.push_back((int32_t)i);
    }
  }
  rev_ptr.push_back((int32_t)nsend);
  ALFI_CHECK(dev_upload(ctx, &L->halo_send_nodes, send_nodes, nsend));
  ALFI_CHECK(dev_upload(ctx, &L->rev_nodes, rev_nodes.data(), (int64_t)rev_nodes.size()));
  ALFI_CHECK(dev_upload(ctx, &L->rev_ptr, rev_ptr.data(), (int64_t)rev_ptr.size()));
  ALFI_CHECK(dev_upload(ctx, &L->rev_pos, order.data(), nsend));
  L->rev_nuniq = (int64_t)rev_nodes.size();
  L->halo_nsend = nsend;
  L->halo_nghost = nb_ghost;
  L->halo_sendbuf = d_sendbuf;
  L->halo_recvbuf = d_recvbuf;
  L->has_halo = true;
  L->distributed = distributed != 0;
  L->n_own = nb_owned * L->bs;
  int32_t nnz_own = 0;
  ALFI_HIP_CHECK(ctx, hipMemcpy(&nnz_own, L->A.rowptr + nb_owned, sizeof(int32_t), hipMemcpyDeviceToHost));
  int64_t nchunks_own = -1;
  if (L->A.aligned) {      // whole-row chunks: a chunk boundary in front of the first ghost row
    const int64_t nb = L->A.nbrows;
    std::vector<int32_t> rp(nb + 1);
    ALFI_HIP_CHECK(ctx, hipMemcpy(rp.data(), L->A.rowptr, sizeof(int32_t) * (nb + 1), hipMemcpyDeviceToHost));
    ALFI_CHECK(build_chunk_tables(ctx, &L->A, rp.data(), nb, nb_owned, &nchunks_own));
  }
  L->A_own = L->A;
  L->A_own.nbrows = nb_owned;
  L->A_own.nnzb = nnz_own;
  if (L->A.aligned) L->A_own.nchunks = nchunks_own;
  return 0;
}

// block-row range [r0, r1) of the flat upload as a view with its own chunk table and carry arrays
static int make_row_view(alfi_ctx* ctx, const DevBSR& A, const std::vector<int32_t>& rowptr, int64_t r0, int64_t r1,
                         DevBSR* V) {
  *V = A;
  V->view = true;
  V->aligned = false;                   // a view brings its own equal-sized chunks (+ fix-up)
  V->chunk_start = nullptr;
  V->nbrows = r1;                       // rows are addressed absolutely; nbrows only has to cover the range
  V->kbase = rowptr[r0];
  V->nnzb = rowptr[r1] - rowptr[r0];
  V->nchunks = (V->nnzb + SPMV_CHUNK - 1) / SPMV_CHUNK;
  V->chunk_row = nullptr;
  V->carry = nullptr;
  V->carry_row = nullptr;
  V->dedup = false;                     // the de-duplication groups are counted from block 0 of the upload
  std::vector<int32_t> chunk_row((size_t)std::max<int64_t>(V->nchunks, 1));
  int64_t row = r0;
  for (int64_t c = 0; c < V->nchunks; ++c) {
    const int64_t k = V->kbase + c * SPMV_CHUNK;
    while (rowptr[row + 1] <= k) ++row;
    chunk_row[c] = (int32_t)row;
  }
  ALFI_CHECK(dev_upload(ctx, &V->chunk_row, chunk_row.data(), V->nchunks));
  ALFI_CHECK(dev_alloc(ctx, &V->carry, V->nchunks * A.bs));
  ALFI_CHECK(dev_alloc(ctx, &V->carry_row, V->nchunks));
  return 0;
}
static void free_row_view(DevBSR* V) {
  dev_free(V->chunk_row);
  dev_free(V->carry);
  dev_free(V->carry_row);
  *V = DevBSR();
}

int alfi_level_set_overlap(alfi_level* L, int64_t nb_interior, int64_t npatch_interior) {
  alfi_ctx* ctx = L->ctx;
  const int64_t nb_own = L->n_own / L->bs;
  if (!L->has_halo || !L->distributed)
    return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_set_overlap needs a distributed level (alfi_level_set_partition)");
  if (nb_interior < 0 || nb_interior > nb_own || npatch_interior < 0 || npatch_interior > L->npatch)
    return alfi_set_error(ctx, ALFI_E_ARG, "interior counts out of range");
  if (!L->A.flat) return 0;   // row-per-lane-group layout: no row-range views; keep the plain exchange
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  // verify the claims: interior rows hold owned columns only, interior patches hold owned dofs only
  const int64_t nb = L->A.nbrows;
  std::vector<int32_t> rowptr(nb + 1);
  ALFI_HIP_CHECK(ctx, hipMemcpy(rowptr.data(), L->A.rowptr, sizeof(int32_t) * (nb + 1), hipMemcpyDeviceToHost));
  const int64_t kint = rowptr[nb_interior];
  if (kint > 0) {
    std::vector<int32_t> col((size_t)kint);
    ALFI_HIP_CHECK(ctx, hipMemcpy(col.data(), L->A.colidx, sizeof(int32_t) * kint, hipMemcpyDeviceToHost));
    for (int64_t k = 0; k < kint; ++k)
      if ((col[k] & 0x7fffffff) >= nb_own)
        return alfi_set_error(ctx, ALFI_E_ARG, "an operator row declared interior holds a ghost column");
  }
  for (int64_t p = 0; p < npatch_interior; ++p)
    for (int64_t q = L->h_patch_ptr[p]; q < L->h_patch_ptr[p + 1]; ++q)
      if (L->h_patch_dofs[q] >= L->n_own)
        return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld declared interior holds a ghost dof", (long long)p);
  free_row_view(&L->A_int);
  free_row_view(&L->A_bnd);
  ALFI_CHECK(make_row_view(ctx, L->A, rowptr, 0, nb_interior, &L->A_int));
  ALFI_CHECK(make_row_view(ctx, L->A, rowptr, nb_interior, nb_own, &L->A_bnd));
  L->npatch_int = npatch_interior;
  L->overlap = true;
  return 0;
}

int alfi_level_destroy(alfi_level* L) {
  if (!L) return 0;
  (void)hipStreamSynchronize(L->ctx->stream);
  free_row_view(&L->A_int);
  free_row_view(&L->A_bnd);
  free_assembly(&L->asmb);
  free_bsr(&L->A);
  dev_free(L->bc_dofs);
  dev_free(L->bc_mask);
  dev_free(L->halo_send_nodes);
  if (L->own_halo_bufs) {
    dev_free(L->halo_sendbuf);
    dev_free(L->halo_recvbuf);
  }
  dev_free(L->rev_nodes);
  dev_free(L->rev_ptr);
  dev_free(L->rev_pos);
  dev_free(L->patch_ptr);
  dev_free(L->patch_dofs);
  dev_free(L->inv_ptr);
  dev_free(L->stage_ptr);
  dev_free(L->inv);
  dev_free(L->inv_il);
  dev_free(L->stage);
  dev_free(L->dof_ptr);
  dev_free(L->dof_pos);
  dev_free(L->mult_seq);
  free_mult_schedule(L);
  dev_free(L->status);
  dev_free(L->chk);
  dev_free(L->chk_list);
  free_cond(L);
  dev_free(L->V);
  dev_free(L->Z);
  dev_free(L->w);
  dev_free(L->hs);
  if (L->cinv_owned) dev_free(L->cinv);
  mf_free(L->mf);
  dev_free(L->mg_b);
  dev_free(L->mg_x);
  dev_free(L->mg_r);
  delete L;
  return 0;
}

int alfi_level_update_values(alfi_level* L, const double* bvals) {
  alfi_ctx* ctx = L->ctx;
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ALFI_CHECK(upload_bsr_values(ctx, &L->A, bvals));
  L->A_own.vals = L->A.vals;
  L->A_int.vals = L->A_bnd.vals = L->A.vals;
  L->factored = false;
  return 0;
}

void free_assembly(AssemblyDev* S) {
  dev_free(S->cptr); dev_free(S->ccell); dev_free(S->cba); dev_free(S->cell_nodes); dev_free(S->grad); dev_free(S->vol);
  dev_free(S->etab); dev_free(S->bItab); dev_free(S->bc_code); dev_free(S->bc_all);
  dev_free(S->wq); dev_free(S->phi); dev_free(S->dphi); dev_free(S->d2phi); dev_free(S->hcell); dev_free(S->diag);
  dev_free(S->wq8); dev_free(S->qtab);
  *S = AssemblyDev();
}

int alfi_level_set_assembly(alfi_level* L, int64_t ncell, int nloc, const int32_t* cell_nodes, const double* grad,
                            const double* vol, const double* Sref, const double* bIref, const double* T1ref, int full_div,
                            const int64_t* cptr, const int32_t* ccell, const uint16_t* cba) {
  alfi_ctx* ctx = L->ctx;
  if (!cell_nodes || !grad || !vol || !Sref || !bIref || !T1ref || !cptr || !ccell || !cba)
    return alfi_set_error(ctx, ALFI_E_ARG, "NULL argument");
  if (ncell < 1 || nloc < 1 || nloc * nloc > 65535) return alfi_set_error(ctx, ALFI_E_ARG, "bad cell counts (%lld cells, %d nodes each)", (long long)ncell, nloc);
  if (!L->A.flat) return alfi_set_error(ctx, ALFI_E_STATE, "device assembly needs the lane-major operator layout (no empty block rows)");
  if (!element_kernel_exists(L->bs, nloc))
    return alfi_set_error(ctx, ALFI_E_ARG, "no element kernel for %d nodes per cell in %d-D", nloc, L->bs);
  const int64_t nnzb = L->A.nnzb, nb = L->A.nbrows;
  const int d = L->bs, nv = d + 1;
  // Unpartitioned: the cells are the mesh, every (cell, a, b) pair contributes to a block.  Partitioned: the cells that touch
  // a local node; their other nodes follow the local ones in the numbering of the state vector, and only the pairs with both
  // nodes local contribute (the ghost rows of the local operator are restricted to local columns).
  const bool part = L->has_halo;
  const int64_t npairs = cptr[nnzb];
  if (cptr[0] != 0 || (part ? npairs > ncell * nloc * nloc : npairs != ncell * nloc * nloc))
    return alfi_set_error(ctx, ALFI_E_ARG, "contributor lists hold %lld pairs, expected %scells x nodes^2 = %lld", (long long)npairs,
                          part ? "at most " : "", (long long)(ncell * nloc * nloc));
  // (the checks below walk 200 M contributors at config-4 size: over host threads, each reporting its first finding)
  std::atomic<int64_t> bad_block(-1), bad_pair(-1), bad_node(-1), max_node(nb - 1);
  host_parallel_ranges(nnzb, [&](int64_t k0, int64_t k1) {
    for (int64_t k = k0; k < k1; ++k)
      if (cptr[k + 1] <= cptr[k]) { bad_block = k; return; }
  });
  if (bad_block >= 0) return alfi_set_error(ctx, ALFI_E_ARG, "block %lld has no contributing cell", (long long)bad_block.load());
  host_parallel_ranges(npairs, [&](int64_t q0, int64_t q1) {
    for (int64_t q = q0; q < q1; ++q)
      if (ccell[q] < 0 || ccell[q] >= ncell || cba[q] >= nloc * nloc) { bad_pair = q; return; }
  });
  if (bad_pair >= 0) return alfi_set_error(ctx, ALFI_E_ARG, "contributor %lld out of range", (long long)bad_pair.load());
  host_parallel_ranges(ncell * nloc, [&](int64_t i0, int64_t i1) {
    int64_t m = 0;
    for (int64_t i = i0; i < i1; ++i) {
      if (cell_nodes[i] < 0 || (!part && cell_nodes[i] >= nb)) { bad_node = i; return; }
      m = std::max<int64_t>(m, cell_nodes[i]);
    }
    int64_t cur = max_node.load();
    while (m > cur && !max_node.compare_exchange_weak(cur, m)) {}
  });
  if (bad_node >= 0) return alfi_set_error(ctx, ALFI_E_ARG, "cell node out of range");
  const int64_t nstate = max_node.load() + 1;
  // the diagonal block of every block row (its contributor list = the cells around the node: the gather of element vectors);
  // the two nodes of a contributing pair are rows / columns of the local operator
  std::vector<int32_t> diag((size_t)nb, -1);
  {
    std::vector<int32_t> rowptr(nb + 1), colidx((size_t)std::max<int64_t>(nnzb, 1));
    ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ALFI_HIP_CHECK(ctx, hipMemcpy(rowptr.data(), L->A.rowptr, sizeof(int32_t) * (nb + 1), hipMemcpyDeviceToHost));
    ALFI_HIP_CHECK(ctx, hipMemcpy(colidx.data(), L->A.colidx, sizeof(int32_t) * nnzb, hipMemcpyDeviceToHost));
    std::atomic<int64_t> stray(-1), stray_block(-1), no_diag(-1);
    host_parallel_ranges(nb, [&](int64_t r0, int64_t r1) {
      for (int64_t r = r0; r < r1; ++r) {
        for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
          const int32_t c = colidx[k] & 0x7fffffff;       // (the sign bit marks the first block of a block row)
          if (c == r) diag[r] = (int32_t)k;
          for (int64_t q = cptr[k]; q < cptr[k + 1]; ++q) {
            const int32_t* cn = cell_nodes + (int64_t)ccell[q] * nloc;
            if (cn[cba[q] % nloc] != r || cn[cba[q] / nloc] != c) { stray = q; stray_block = k; return; }
          }
        }
        if (diag[r] < 0) { no_diag = r; return; }
      }
    });
    if (stray >= 0)
      return alfi_set_error(ctx, ALFI_E_ARG, "contributor %lld does not belong to block %lld", (long long)stray.load(),
                            (long long)stray_block.load());
    if (no_diag >= 0) return alfi_set_error(ctx, ALFI_E_ARG, "block row %lld has no diagonal block", (long long)no_diag.load());
  }
  // per (a, b) the slices of the reference tensors the cell kernel reads (kernels_assemble.hip):
  // T1[k,i,b,a] | T1[b,i,k,a] | S[a,b,i,j];  T1: (nloc, d+1, nloc, nloc), S: (nloc, nloc, d+1, d+1)
  const int ts = 2 * nv * nloc + nv * nv;
  std::vector<double> etab((size_t)nloc * nloc * ts);
  for (int a = 0; a < nloc; ++a)
    for (int b = 0; b < nloc; ++b) {
      double* t = etab.data() + (size_t)(a * nloc + b) * ts;
      for (int i = 0; i < nv; ++i)
        for (int k = 0; k < nloc; ++k) {
          t[i * nloc + k] = T1ref[(((size_t)k * nv + i) * nloc + b) * nloc + a];
          t[nv * nloc + i * nloc + k] = T1ref[(((size_t)b * nv + i) * nloc + k) * nloc + a];
        }
      for (int i = 0; i < nv; ++i)
        for (int j = 0; j < nv; ++j) t[2 * nv * nloc + i * nv + j] = Sref[(((size_t)a * nloc + b) * nv + i) * nv + j];
    }
  free_assembly(&L->asmb);
  AssemblyDev S;
  S.nloc = nloc;
  S.ncell = ncell;
  S.npairs = npairs;
  S.nstate = nstate;
  S.full_div = full_div != 0;
  int rc = dev_upload(ctx, &S.cptr, cptr, nnzb + 1);
  if (rc == 0) rc = dev_upload(ctx, &S.ccell, ccell, npairs);
  if (rc == 0) rc = dev_upload(ctx, &S.cba, cba, npairs);
  if (rc == 0) rc = dev_upload(ctx, &S.cell_nodes, cell_nodes, ncell * nloc);
  if (rc == 0) rc = dev_upload(ctx, &S.grad, grad, ncell * nv * d);
  if (rc == 0) rc = dev_upload(ctx, &S.vol, vol, ncell);
  if (rc == 0) rc = dev_upload(ctx, &S.etab, etab.data(), (int64_t)etab.size());
  if (rc == 0) rc = dev_upload(ctx, &S.bItab, bIref, (int64_t)nloc * nv);
  if (rc == 0) rc = dev_upload(ctx, &S.diag, diag.data(), nb);
  if (rc != 0) {
    free_assembly(&S);
    return rc;
  }
  S.ready = true;
  L->asmb = S;
  return 0;
}

int alfi_ctx_set_assembly_scratch(alfi_ctx* ctx, int64_t max_bytes) {
  if (max_bytes < 1) return alfi_set_error(ctx, ALFI_E_ARG, "scratch limit must be positive");
  ctx->asm_scratch_limit = max_bytes;
  return 0;
}

int alfi_level_assemble(alfi_level* L, double nu, double gamma, double adv, const double* d_state, int apply_bc) {
  alfi_ctx* ctx = L->ctx;
  if (!L->asmb.ready) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_assemble before alfi_level_set_assembly");
  if (adv != 0.0 && !d_state) return alfi_set_error(ctx, ALFI_E_ARG, "advection needs the state");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ctx->cur_tag = L->id;
  int t = alfi_prof_begin(ctx, ALFI_EV_PATCH_FACTOR);       // PCPatchComputeOp
  ALFI_CHECK(launch_operator_refresh(L, nu, gamma, adv, d_state, true, false, 0.0, 0.0, false, apply_bc != 0, L->A.vals));
  alfi_prof_end(ctx, t);
  L->factored = false;
  return 0;
}

// The refresh of a stabilised run in one pass: A = nu K + gamma D + adv N(state) + the linearised SUPG term, boundary conditions
int alfi_level_assemble_supg(alfi_level* L, double nu, double gamma, double adv, const double* d_state, double weight,
                             double magic, int apply_bc) {
  alfi_ctx* ctx = L->ctx;
  if (!L->asmb.supg_ready) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_assemble_supg before alfi_level_set_supg");
  if (!d_state) return alfi_set_error(ctx, ALFI_E_ARG, "SUPG needs the state");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ctx->cur_tag = L->id;
  int t = alfi_prof_begin(ctx, ALFI_EV_PATCH_FACTOR);
  ALFI_CHECK(launch_operator_refresh(L, nu, gamma, adv, d_state, true, true, weight, magic, false, apply_bc != 0, L->A.vals));
  alfi_prof_end(ctx, t);
  L->factored = false;
  return 0;
}


// Partitioned levels: the Dirichlet dofs among ALL local dofs (the level was created with the owned ones only: its smoother
// copies x to y there, which must happen on the owner alone).  The refresh turns the rows / columns of these dofs into
// identity, as firedrake.assemble(a, bcs) does on every rank's rows [3P].
int alfi_level_set_assembly_bc(alfi_level* L, const int32_t* bc_dofs, int64_t nbc) {
  alfi_ctx* ctx = L->ctx;
  if (!L->asmb.ready) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_set_assembly_bc before alfi_level_set_assembly");
  if (nbc < 0 || (nbc > 0 && !bc_dofs)) return alfi_set_error(ctx, ALFI_E_ARG, "NULL argument");
  std::vector<uint8_t> mask((size_t)std::max<int64_t>(L->n, 1), 0);
  for (int64_t i = 0; i < nbc; ++i) {
    if (bc_dofs[i] < 0 || bc_dofs[i] >= L->n) return alfi_set_error(ctx, ALFI_E_ARG, "Dirichlet dof %d out of range", bc_dofs[i]);
    mask[(size_t)bc_dofs[i]] = 1;
  }
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(L->asmb.bc_all);
  dev_free(L->asmb.bc_code);
  L->asmb.bc_all = nullptr;
  L->asmb.bc_code = nullptr;
  return dev_upload(ctx, &L->asmb.bc_all, mask.data(), (int64_t)mask.size());
}

int alfi_level_assembly_state_size(alfi_level* L, int64_t* n) {
  if (!L->asmb.ready) return alfi_set_error(L->ctx, ALFI_E_STATE, "alfi_level_assembly_state_size before alfi_level_set_assembly");
  *n = L->asmb.nstate * L->bs;
  return 0;
}

// y = A(state) x with A = nu K + gamma D + adv N(state) WITHOUT boundary conditions, matrix-free: every cell multiplies its
// element matrix with its entries of x as it forms it, the element vectors are gathered per node in a fixed order.  The level's
// own operator (the Jacobian the patches were factored from) is not touched.  The nonlinear residual of
// alfi/solver.py:565-568 is one such product: F_u = (nu K + gamma D) u + 1/2 N(u) u = A(u; adv / 2) u.
int alfi_level_assemble_mult(alfi_level* L, double nu, double gamma, double adv, const double* d_state, const double* dx,
                             double* dy) {
  alfi_ctx* ctx = L->ctx;
  AssemblyDev& S = L->asmb;
  if (!S.ready) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_assemble_mult before alfi_level_set_assembly");
  if (adv != 0.0 && !d_state) return alfi_set_error(ctx, ALFI_E_ARG, "advection needs the state");
  if (!dx || !dy) return alfi_set_error(ctx, ALFI_E_ARG, "NULL argument");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ctx->cur_tag = L->id;
  int t = alfi_prof_begin(ctx, ALFI_EV_PATCH_FACTOR);
  ALFI_CHECK(launch_element_mult(L, nu, gamma, adv, d_state, dx, dy));
  alfi_prof_end(ctx, t);
  return 0;
}

int alfi_level_set_supg(alfi_level* L, int nq, const double* wq, const double* phi, const double* dphi, const double* d2phi,
                        const double* hcell) {
  alfi_ctx* ctx = L->ctx;
  AssemblyDev& S = L->asmb;
  if (!S.ready) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_set_supg before alfi_level_set_assembly");
  if (nq < 1 || !wq || !phi || !dphi || !d2phi || !hcell) return alfi_set_error(ctx, ALFI_E_ARG, "NULL / empty SUPG tables");
  if (S.nloc > 16) return alfi_set_error(ctx, ALFI_E_ARG, "SUPG kernels handle elements of at most 16 nodes, got %d", S.nloc);
  for (int64_t c = 0; c < S.ncell; ++c)
    if (!(hcell[c] > 0.0)) return alfi_set_error(ctx, ALFI_E_ARG, "cell size of cell %lld is not positive", (long long)c);
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(S.wq); dev_free(S.phi); dev_free(S.dphi); dev_free(S.d2phi); dev_free(S.hcell); dev_free(S.wq8); dev_free(S.qtab);
  S.wq = S.phi = S.dphi = S.d2phi = S.hcell = S.wq8 = S.qtab = nullptr;
  S.supg_ready = false;
  const int nv = L->bs + 1, nloc = S.nloc;
  ALFI_CHECK(dev_upload(ctx, &S.wq, wq, nq));
  ALFI_CHECK(dev_upload(ctx, &S.phi, phi, (int64_t)nq * nloc));
  ALFI_CHECK(dev_upload(ctx, &S.dphi, dphi, (int64_t)nq * nloc * nv));
  ALFI_CHECK(dev_upload(ctx, &S.d2phi, d2phi, (int64_t)nq * nloc * nv * nv));
  ALFI_CHECK(dev_upload(ctx, &S.hcell, hcell, S.ncell));
  {
    // the linearisation kernel takes eight points per chunk (two matrix-core steps of four): pad with zero-weight copies of point 0, and pack per
    // (point, node) phi | dphi | the upper triangle of the (symmetric) second derivatives
    const int nq8 = (nq + 7) / 8 * 8, nh2 = nv * (nv + 1) / 2, qt = 1 + nv + nh2;
    std::vector<double> w8((size_t)nq8, 0.0), tab((size_t)nq8 * nloc * qt);
    for (int p = 0; p < nq8; ++p) {
      const int ps = p < nq ? p : 0;
      if (p < nq) w8[p] = wq[p];
      for (int a = 0; a < nloc; ++a) {
        double* t = tab.data() + ((size_t)p * nloc + a) * qt;
        t[0] = phi[(size_t)ps * nloc + a];
        for (int m = 0; m < nv; ++m) t[1 + m] = dphi[((size_t)ps * nloc + a) * nv + m];
        int at = 1 + nv;
        for (int m = 0; m < nv; ++m)
          for (int n = m; n < nv; ++n) t[at++] = d2phi[(((size_t)ps * nloc + a) * nv + m) * nv + n];
      }
    }
    ALFI_CHECK(dev_upload(ctx, &S.wq8, w8.data(), nq8));
    ALFI_CHECK(dev_upload(ctx, &S.qtab, tab.data(), (int64_t)tab.size()));
    S.nq8 = nq8;
  }
  S.nq = nq;
  S.supg_ready = true;
  return 0;
}

int alfi_level_supg(alfi_level* L, double nu, double weight, double magic, const double* d_state, int add_to_operator, double* d_F) {
  alfi_ctx* ctx = L->ctx;
  if (!L->asmb.supg_ready) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_level_supg before alfi_level_set_supg");
  if (!d_state) return alfi_set_error(ctx, ALFI_E_ARG, "SUPG needs the state");
  if (!add_to_operator && !d_F) return 0;
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ctx->cur_tag = L->id;
  int t = alfi_prof_begin(ctx, ALFI_EV_PATCH_FACTOR);       // PCPatchComputeOp
  if (add_to_operator)
    ALFI_CHECK(launch_operator_refresh(L, nu, 0.0, 0.0, d_state, false, true, weight, magic, true, false, L->A.vals));
  if (d_F) ALFI_CHECK(launch_supg_residual(L, nu, weight, magic, d_state, d_F));
  alfi_prof_end(ctx, t);
  if (add_to_operator) L->factored = false;
  return 0;
}

int alfi_level_apply_bc(alfi_level* L) {
  if (!L->asmb.ready) return alfi_set_error(L->ctx, ALFI_E_STATE, "alfi_level_apply_bc before alfi_level_set_assembly");
  L->factored = false;
  return launch_apply_bc(L);
}

int alfi_level_get_values(alfi_level* L, double* bvals) {
  alfi_ctx* ctx = L->ctx;
  const int bb = L->bs * L->bs;
  if (L->A.nnzb == 0) return 0;
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!L->A.flat) {
    ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ALFI_HIP_CHECK(ctx, hipMemcpy(bvals, L->A.vals, (size_t)L->A.nnzb * bb * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
  }
  double* tmp = nullptr;
  ALFI_CHECK(dev_alloc(ctx, &tmp, L->A.nnzb * bb));
  int rc = launch_vals_from_lanes(ctx, L->A, tmp);
  if (rc == 0 && hipMemcpyAsync(bvals, tmp, (size_t)L->A.nnzb * bb * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = ALFI_E_HIP;
  if (rc == 0 && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ALFI_E_HIP;
  dev_free(tmp);
  return rc;
}

int alfi_level_size(alfi_level* L, int64_t* n) {
  *n = L->n;
  return 0;
}
int alfi_level_id(alfi_level* L, int* id) {
  *id = L->id;
  return 0;
}

// Partitioned levels: x's ghost slots are refreshed from their owners first (they are scratch: see alfi_hip.h), the
// product is formed on the owned rows.
// y = A x (mode 0) or y = b - A x (mode 1) on the owned rows of a level.  With alfi_level_set_overlap the rows without
// ghost columns are multiplied while the forward halo of x is in flight.
int level_spmv(alfi_level* L, const double* dx, double* dy, const double* db, int mode, bool ghosts_current) {
  alfi_ctx* ctx = L->ctx;
  ctx->cur_tag = L->id;
  int t;
  if (L->distributed && L->overlap && !ghosts_current) {
    ALFI_CHECK(halo_fwd_begin(L, dx));
    t = alfi_prof_begin(ctx, ALFI_EV_MATMULT);
    ALFI_CHECK(launch_bsr_spmv(ctx, L->A_int, dx, dy, db, 1.0, mode));
    alfi_prof_end(ctx, t);
    ALFI_CHECK(halo_fwd_end(L, const_cast<double*>(dx)));
    t = alfi_prof_begin(ctx, ALFI_EV_MATMULT);
    ALFI_CHECK(launch_bsr_spmv(ctx, L->A_bnd, dx, dy, db, 1.0, mode));
    alfi_prof_end(ctx, t);
    return 0;
  }
  if (L->distributed && !ghosts_current) ALFI_CHECK(halo_fwd(L, const_cast<double*>(dx)));
  t = alfi_prof_begin(ctx, ALFI_EV_MATMULT);
  ALFI_CHECK(launch_bsr_spmv(ctx, L->A_own, dx, dy, db, 1.0, mode));
  alfi_prof_end(ctx, t);
  return 0;
}

int alfi_spmv(alfi_level* L, const double* dx, double* dy) { return level_spmv(L, dx, dy, nullptr, 0); }

int alfi_residual(alfi_level* L, const double* db, const double* dx, double* dr) { return level_spmv(L, dx, dr, db, 1); }

// PCApply_PATCH on a (possibly partitioned) level: ghost values in, local patch solves, ghost contributions back to
// their owners, Dirichlet dofs copied
// ghosts_current (smoother only): with a sum-exchange plan the one exchange after the local solves leaves the total on the
// ghost copies too; *ghosts_current tells the caller that the product A y needs no forward exchange
int level_patch_apply(alfi_level* L, const double* dx, double* dy, bool* ghosts_current) {
  alfi_ctx* ctx = L->ctx;
  if (ghosts_current) *ghosts_current = false;
  if (L->mult) {
    // multiplicative sweep (PCApply_PATCH, local_type multiplicative [3P]): y = 0, then wavefront by wavefront in
    // iteration order and, with symmetrise_sweep, back again in reverse order
    if (L->distributed) ALFI_CHECK(halo_fwd(L, const_cast<double*>(dx)));
    ALFI_HIP_CHECK(ctx, hipMemsetAsync(dy, 0, sizeof(double) * L->n, ctx->stream));
    int t = alfi_prof_begin(ctx, ALFI_EV_PATCH_APPLY);
    const int64_t nw = (int64_t)L->mult_wave_ptr.size() - 1;
    // ALFI_MULT_PERSISTENT=0: one launch per dependency wavefront (the schedule of rounds 1-3; kept for the bitwise comparison)
    static const bool persistent = !(getenv("ALFI_MULT_PERSISTENT") && atoi(getenv("ALFI_MULT_PERSISTENT")) == 0);
    if (persistent && L->mult_nitems > 0) {
      // (a wait that runs into its bound sets the ctx's sticky error word, reported by the next synchronising call: no host
      // synchronisation inside the smoother)
      ALFI_CHECK(launch_patch_mult_persistent(L, dx, dy));
      alfi_prof_end(ctx, t);
      if (L->distributed) ALFI_CHECK(halo_rev(L, dy));
      if (L->nbc > 0) {
        t = alfi_prof_begin(ctx, ALFI_EV_PATCH_SCATTER);
        ALFI_CHECK(launch_copy_dofs(ctx, dy, dx, L->bc_dofs, L->nbc));
        alfi_prof_end(ctx, t);
      }
      return 0;
    }
    for (int64_t w = 0; w < nw; ++w)
      ALFI_CHECK(launch_patch_mult_wave(L, L->mult_seq + L->mult_wave_ptr[w], L->mult_wave_ptr[w + 1] - L->mult_wave_ptr[w],
                                        dx, dy));
    if (L->mult_symmetrise)
      for (int64_t w = nw - 1; w >= 0; --w)
        ALFI_CHECK(launch_patch_mult_wave(L, L->mult_seq + L->mult_wave_ptr[w],
                                          L->mult_wave_ptr[w + 1] - L->mult_wave_ptr[w], dx, dy));
    alfi_prof_end(ctx, t);
    if (L->distributed) ALFI_CHECK(halo_rev(L, dy));
    if (L->nbc > 0) {
      t = alfi_prof_begin(ctx, ALFI_EV_PATCH_SCATTER);
      ALFI_CHECK(launch_copy_dofs(ctx, dy, dx, L->bc_dofs, L->nbc));
      alfi_prof_end(ctx, t);
    }
    return 0;
  }
  if (L->distributed && L->overlap) {
    // Both exchanges hidden behind the patches that hold no ghost dof: half of them run while the forward halo of x is
    // in flight; then the patches with ghost dofs (the only ones contributing to ghost slots), the sums on the ghost
    // slots, and the reverse exchange starts; the other half of the interior patches and the sums on the owned dofs run
    // while it is in flight.
    const int64_t half = L->npatch_int / 2;
    ALFI_CHECK(halo_fwd_begin(L, dx));
    ALFI_CHECK(launch_patch_apply_range(L, 0, half, dx));
    ALFI_CHECK(halo_fwd_end(L, const_cast<double*>(dx)));
    ALFI_CHECK(launch_patch_apply_range(L, L->npatch_int, L->npatch, dx));
    ALFI_CHECK(launch_patch_sum_range(L, L->n_own, L->n, dx, dy));
    ALFI_CHECK(halo_rev_begin(L, dy));
    ALFI_CHECK(launch_patch_apply_range(L, half, L->npatch_int, dx));
    ALFI_CHECK(launch_patch_sum_range(L, 0, L->n_own, dx, dy));
    return halo_rev_end(L, dy);
  }
  if (L->distributed) ALFI_CHECK(halo_fwd(L, const_cast<double*>(dx)));
  ALFI_CHECK(launch_patch_apply(L, dx, dy));    // includes y[bc] = x[bc] (no patch holds a Dirichlet dof, so the
                                                // exchange brings nothing to those entries)
  if (L->distributed && ghosts_current && L->sum_ready && !L->pou) {
    ALFI_CHECK(halo_sum(L, dy));
    *ghosts_current = true;
  } else if (L->distributed) {
    ALFI_CHECK(halo_rev(L, dy));
  }
  return 0;
}
