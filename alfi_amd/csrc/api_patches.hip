// C ABI of libalfi_hip.so (include/alfi_hip.h), PCPATCH: patch sets, condensed factors, multiplicative schedules, factorisation, apply.
// (One file per concern since round 5: api_ctx / api_level / api_patches / api_smoother / api_cycles / api_saddle; the helpers they
// share are declared in api_internal.h.)
#include "api_internal.h"

// ---- patches -------------------------------------------------------------------------------------------------------------
int alfi_patches_set(alfi_level* L, int64_t npatch, const int64_t* pptr, const int32_t* pdofs) {
  alfi_ctx* ctx = L->ctx;
  if (npatch < 0 || (npatch > 0 && (!pptr || !pdofs))) return alfi_set_error(ctx, ALFI_E_ARG, "NULL patch arrays");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(L->patch_ptr);
  dev_free(L->patch_dofs);
  dev_free(L->inv_ptr);
  dev_free(L->stage_ptr);
  dev_free(L->inv);
  dev_free(L->inv_il);
  L->inv_il = nullptr;
  L->il_doubles = 0;
  L->il_valid = false;
  dev_free(L->stage);
  dev_free(L->dof_ptr);
  dev_free(L->dof_pos);
  L->patch_ptr = nullptr; L->patch_dofs = nullptr; L->inv_ptr = nullptr; L->stage_ptr = nullptr;
  L->inv = nullptr; L->stage = nullptr; L->dof_ptr = nullptr; L->dof_pos = nullptr;
  L->factored = false;
  free_cond(L);
  L->inv_shrunk = false;
  L->npatch = npatch;
  const int64_t sum_n = npatch > 0 ? pptr[npatch] : 0;
  if (sum_n > INT32_MAX) return alfi_set_error(ctx, ALFI_E_ARG, "too many patch dofs for int32 staging indices");
  std::vector<int64_t> inv_ptr(npatch + 1), stage_ptr(npatch + 1);
  int64_t ip = 0, sp = 0, sum_n2 = 0;
  int max_np = 0;
  for (int64_t p = 0; p < npatch; ++p) {
    const int64_t n = pptr[p + 1] - pptr[p];
    if (n <= 0 || n > PATCH_MAX)
      return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld has %lld dofs; supported range is 1..%d", (long long)p,
                            (long long)n, PATCH_MAX);
    for (int64_t q = pptr[p]; q < pptr[p + 1]; ++q) {
      if (pdofs[q] < 0 || pdofs[q] >= L->n)
        return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: dof %d out of range", (long long)p, pdofs[q]);
      if (q > pptr[p] && pdofs[q] <= pdofs[q - 1])
        return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: dofs must be strictly ascending", (long long)p);
    }
    const int64_t ld = (n + 1) & ~(int64_t)1;
    inv_ptr[p] = ip;
    stage_ptr[p] = sp;
    ip += (n * ld + 15) & ~(int64_t)15;
    sp += ld;
    sum_n2 += n * n;
    max_np = std::max<int>(max_np, (int)n);
  }
  inv_ptr[npatch] = ip;
  stage_ptr[npatch] = sp;
  if (sp > INT32_MAX) return alfi_set_error(ctx, ALFI_E_ARG, "staging buffer exceeds int32 indexing");
  L->sum_n = sum_n;
  L->sum_n2 = sum_n2;
  L->max_np = max_np;
  L->inv_doubles = ip;
  L->stage_len = sp;
  // dof -> staged positions (counting sort; patch order = fixed summation order)
  std::vector<int32_t> dof_ptr(L->n + 1, 0), dof_pos(sum_n > 0 ? sum_n : 1);
  for (int64_t q = 0; q < sum_n; ++q) dof_ptr[pdofs[q] + 1]++;
  for (int64_t i = 0; i < L->n; ++i) dof_ptr[i + 1] += dof_ptr[i];
  {
    std::vector<int32_t> fill(dof_ptr.begin(), dof_ptr.end() - 1);
    for (int64_t p = 0; p < npatch; ++p)
      for (int64_t q = pptr[p]; q < pptr[p + 1]; ++q)
        dof_pos[fill[pdofs[q]]++] = (int32_t)(stage_ptr[p] + (q - pptr[p]));
  }
  ALFI_CHECK(dev_upload(ctx, &L->patch_ptr, pptr, npatch + 1));
  ALFI_CHECK(dev_upload(ctx, &L->patch_dofs, pdofs, sum_n));
  ALFI_CHECK(dev_upload(ctx, &L->inv_ptr, inv_ptr.data(), npatch + 1));
  ALFI_CHECK(dev_upload(ctx, &L->stage_ptr, stage_ptr.data(), npatch + 1));
  ALFI_CHECK(dev_upload(ctx, &L->dof_ptr, dof_ptr.data(), L->n + 1));
  ALFI_CHECK(dev_upload(ctx, &L->dof_pos, dof_pos.data(), sum_n));
  // the dense inverses (8 sum n_p^2 bytes) are allocated by the first alfi_patches_factor that needs them: a level that
  // gets condensed factors (alfi_patches_set_groups) never holds them -- 265 GB for the 3.4 M-dof Scott-Vogelius level
  ALFI_CHECK(dev_alloc(ctx, &L->inv, 16));
  L->inv_shrunk = true;
  ALFI_CHECK(dev_alloc(ctx, &L->stage, sp));
  ALFI_HIP_CHECK(ctx, hipMemsetAsync(L->stage, 0, (size_t)std::max<int64_t>(sp, 1) * sizeof(double), ctx->stream));
  L->h_patch_ptr.assign(pptr, pptr + npatch + 1);
  L->h_patch_dofs.assign(pdofs, pdofs + sum_n);
  L->h_inv_ptr = inv_ptr;
  dev_free(L->mult_seq);
  L->mult_seq = nullptr;
  L->mult = false;
  L->mult_wave_ptr.clear();
  free_mult_schedule(L);
  // a new patch set invalidates the interior-patch count of alfi_level_set_overlap: back to the plain exchange until the
  // caller declares the new one
  L->overlap = false;
  L->npatch_int = 0;
  return 0;
}

int alfi_patches_set_groups(alfi_level* L, const int32_t* group) {
  alfi_ctx* ctx = L->ctx;
  if (!L->patch_ptr) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_patches_set_groups before alfi_patches_set");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  free_cond(L);
  L->factored = false;
  if (!group) return 0;                                  // back to dense inverses (allocated by alfi_patches_factor)
  if (L->mult) return alfi_set_error(ctx, ALFI_E_STATE, "condensed patch factors do not support multiplicative sweeps");
  const int bs = L->bs;
  const int64_t npatch = L->npatch, nb = L->A.nbrows, nnzb = L->A.nnzb;
  std::vector<int32_t> rowptr(nb + 1), colidx(nnzb > 0 ? nnzb : 1);
  ALFI_HIP_CHECK(ctx, hipMemcpy(rowptr.data(), L->A.rowptr, sizeof(int32_t) * (nb + 1), hipMemcpyDeviceToHost));
  if (nnzb > 0) ALFI_HIP_CHECK(ctx, hipMemcpy(colidx.data(), L->A.colidx, sizeof(int32_t) * nnzb, hipMemcpyDeviceToHost));
  const std::vector<int64_t>& pp = L->h_patch_ptr;
  const std::vector<int32_t>& pd = L->h_patch_dofs;
  const int64_t sum_n = pp[npatch];
  std::vector<int32_t> c_dofs(sum_n), c_slot(sum_n), g_off, g_m, g_sc, g_uoff, sidx, p_nI(npatch), s_uptr, s_uidx;
  std::vector<int64_t> gptr(npatch + 1, 0), g_mat, g_sidx, sptr(npatch + 1, 0), sinv_ptr(npatch + 1, 0);
  std::vector<int32_t> node_pos(nb, -1);               // node -> condensed NODE position inside the current patch
  int64_t mat_off = 0, sinv_off = 0;
  int lds_max = 0, umax = 0, smax = 0;
  s_uptr.push_back(0);
  for (int64_t p = 0; p < npatch; ++p) {
    const int64_t off = pp[p];
    const int n = (int)(pp[p + 1] - off);
    if (n % bs != 0) return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: condensed factors need patches of whole nodes", (long long)p);
    const int nn = n / bs;
    // labels per node; groups = distinct non-negative labels in ascending order
    std::vector<int32_t> lab(nn);
    for (int i = 0; i < nn; ++i) {
      lab[i] = group[off + (int64_t)i * bs];
      for (int c = 0; c < bs; ++c) {
        if (pd[off + (int64_t)i * bs + c] != (pd[off + (int64_t)i * bs] / bs) * bs + c || group[off + (int64_t)i * bs + c] != lab[i])
          return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: entries of a node must be adjacent and carry one group label", (long long)p);
      }
    }
    std::vector<int32_t> labels;
    for (int i = 0; i < nn; ++i) if (lab[i] >= 0) labels.push_back(lab[i]);
    std::sort(labels.begin(), labels.end());
    labels.erase(std::unique(labels.begin(), labels.end()), labels.end());
    const int ng = (int)labels.size();
    // condensed node order: groups (ascending label, ascending node), then the skeleton nodes
    std::vector<int32_t> order;                         // condensed node position -> sorted node position
    order.reserve(nn);
    std::vector<int32_t> gstart(ng + 1, 0);
    for (int g = 0; g < ng; ++g) {
      for (int i = 0; i < nn; ++i) if (lab[i] == labels[g]) order.push_back(i);
      gstart[g + 1] = (int32_t)order.size();
    }
    const int nIn = (int)order.size();                  // interior nodes
    for (int i = 0; i < nn; ++i) if (lab[i] < 0) order.push_back(i);
    for (int q = 0; q < nn; ++q) {
      node_pos[pd[off + (int64_t)order[q] * bs] / bs] = q;
      for (int c = 0; c < bs; ++c) {
        c_dofs[off + (int64_t)q * bs + c] = pd[off + (int64_t)order[q] * bs + c];
        c_slot[off + (int64_t)q * bs + c] = order[q] * bs + c;
      }
    }
    const int sn = nn - nIn, s = sn * bs;
    p_nI[p] = nIn * bs;
    sptr[p + 1] = sptr[p] + s;
    const int64_t ld_s = (s + 1) & ~1;
    sinv_ptr[p] = sinv_off;
    sinv_off += ((int64_t)s * ld_s + 15) & ~(int64_t)15;
    if (s > smax) smax = s;
    gptr[p + 1] = gptr[p] + ng;
    // per group: the skeleton nodes its rows couple to; a column inside another group is an error
    std::vector<std::vector<int32_t>> row_contrib(sn);  // skeleton node -> (u position of its first component) per group
    int uoff = 0;
    std::vector<char> mark(sn);
    for (int g = 0; g < ng; ++g) {
      std::fill(mark.begin(), mark.end(), 0);
      for (int q = gstart[g]; q < gstart[g + 1]; ++q) {
        const int node = pd[off + (int64_t)order[q] * bs] / bs;
        for (int32_t k = rowptr[node]; k < rowptr[node + 1]; ++k) {
          const int cq = node_pos[colidx[k] & 0x7fffffff];
          if (cq < 0) continue;
          if (cq >= nIn) mark[cq - nIn] = 1;
          else if (cq < gstart[g] || cq >= gstart[g + 1]) {
            for (int i = 0; i < nn; ++i) node_pos[pd[off + (int64_t)i * bs] / bs] = -1;
            return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: groups %d and another one are coupled by an operator entry "
                                  "(a group may touch the rest of the patch only through unlabelled dofs)", (long long)p, labels[g]);
          }
        }
      }
      const int m = (gstart[g + 1] - gstart[g]) * bs;
      int scn = 0;
      g_sidx.push_back((int64_t)sidx.size());
      for (int j = 0; j < sn; ++j)
        if (mark[j]) {
          for (int c = 0; c < bs; ++c) sidx.push_back(j * bs + c);
          row_contrib[j].push_back(uoff + scn * bs);
          ++scn;
        }
      const int sc = scn * bs;
      if (m > 64 || sc > 64) {
        for (int i = 0; i < nn; ++i) node_pos[pd[off + (int64_t)i * bs] / bs] = -1;
        return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: group %d holds %d entries coupled to %d skeleton entries; the "
                              "condensed factors handle at most 64 of each", (long long)p, labels[g], m, sc);
      }
      g_off.push_back(gstart[g] * bs);
      g_m.push_back(m);
      g_sc.push_back(sc);
      g_uoff.push_back(uoff);
      g_mat.push_back(mat_off);
      mat_off += cond_group_doubles(m, sc);
      uoff += sc;
    }
    if (uoff > umax) umax = uoff;
    for (int j = 0; j < sn; ++j)
      for (int c = 0; c < bs; ++c) {
        for (int32_t u : row_contrib[j]) s_uidx.push_back(u + c);
        s_uptr.push_back((int32_t)s_uidx.size());
      }
    for (int i = 0; i < nn; ++i) node_pos[pd[off + (int64_t)i * bs] / bs] = -1;
    const int lds = (n + uoff + s + 2) * (int)sizeof(double);
    if (lds > lds_max) lds_max = lds;
  }
  sinv_ptr[npatch] = sinv_off;
  if (lds_max > 150 * 1024)
    return alfi_set_error(ctx, ALFI_E_ARG, "condensed apply would need %d bytes of LDS per patch", lds_max);
  if (g_off.empty()) return alfi_set_error(ctx, ALFI_E_ARG, "no group label >= 0: nothing to condense");
  if (sidx.empty()) sidx.push_back(0);
  if (s_uidx.empty()) s_uidx.push_back(0);
  CondDev cd;
  ALFI_CHECK(cond_upload(L, &cd.dofs, c_dofs));
  ALFI_CHECK(cond_upload(L, &cd.slot, c_slot));
  ALFI_CHECK(cond_upload(L, &cd.gptr, gptr));
  ALFI_CHECK(cond_upload(L, &cd.g_off, g_off));
  ALFI_CHECK(cond_upload(L, &cd.g_m, g_m));
  ALFI_CHECK(cond_upload(L, &cd.g_sc, g_sc));
  ALFI_CHECK(cond_upload(L, &cd.g_uoff, g_uoff));
  ALFI_CHECK(cond_upload(L, &cd.g_mat, g_mat));
  ALFI_CHECK(cond_upload(L, &cd.g_sidx, g_sidx));
  ALFI_CHECK(cond_upload(L, &cd.sidx, sidx));
  ALFI_CHECK(cond_upload(L, &cd.p_nI, p_nI));
  ALFI_CHECK(cond_upload(L, &cd.sptr, sptr));
  ALFI_CHECK(cond_upload(L, &cd.sinv_ptr, sinv_ptr));
  ALFI_CHECK(cond_upload(L, &cd.s_uptr, s_uptr));
  ALFI_CHECK(cond_upload(L, &cd.s_uidx, s_uidx));
  {
    // dispatch order of a full-range apply: descending factor bytes (group matrices + inv(Sigma)), ties by index
    std::vector<int64_t> pbytes((size_t)L->npatch, 0);
    for (int64_t p = 0; p < L->npatch; ++p) {
      const int64_t s = sptr[p + 1] - sptr[p];
      pbytes[p] = s * s;
      for (int64_t g = gptr[p]; g < gptr[p + 1]; ++g) pbytes[p] += cond_group_doubles(g_m[g], g_sc[g]);
    }
    std::vector<int32_t> order((size_t)L->npatch);
    for (int64_t p = 0; p < L->npatch; ++p) order[p] = (int32_t)p;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return pbytes[a] > pbytes[b]; });
    ALFI_CHECK(cond_upload(L, &cd.order, order));
  }
  {
    // tables of the three-launch apply (kernels_bigpatch.hip): chunks of <= 256 rows of inv(Sigma) for the sigma kernel
    // (COND_SIGMA_ROWS), row pairs of the group matrices, the row-sorted order of the u buffer
    std::vector<int32_t> ch_patch, ch_row, xp_grp, bp_grp, g_xp(g_m.size()), g_bp(g_m.size()), u_dst;
    std::vector<int64_t> chptr((size_t)npatch + 1, 0), uptr((size_t)npatch + 1, 0), xp_ptr((size_t)npatch + 1, 0),
        bp_ptr((size_t)npatch + 1, 0);
    int lds_front = 0, lds_back = 0;
    for (int64_t p = 0; p < npatch; ++p) {
      const int s = (int)(sptr[p + 1] - sptr[p]), ld = (s + 1) & ~1;
      for (int r = 0; r < ld; r += COND_SIGMA_ROWS) {
        ch_patch.push_back((int32_t)p);
        ch_row.push_back(r);
      }
      chptr[p + 1] = (int64_t)ch_patch.size();
      int uo = 0, xp = 0, bp = 0;
      for (int64_t g = gptr[p]; g < gptr[p + 1]; ++g) {
        g_xp[g] = xp;
        g_bp[g] = bp;
        for (int i = 0; i < cond_pairs(g_m[g]); ++i) xp_grp.push_back((int32_t)g);
        for (int j = 0; j < cond_pairs(g_sc[g]); ++j) bp_grp.push_back((int32_t)g);
        xp += cond_pairs(g_m[g]);
        bp += cond_pairs(g_sc[g]);
        uo += g_sc[g];
      }
      uptr[p + 1] = uptr[p] + uo;
      xp_ptr[p + 1] = xp_ptr[p] + xp;
      bp_ptr[p + 1] = bp_ptr[p] + bp;
      // every entry of the u buffer is one contribution to one skeleton row: s_uidx restricted to the patch is a permutation
      const int32_t qb = s_uptr[sptr[p]], qe = s_uptr[sptr[p + 1]];
      if (qe - qb != uo) return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: inconsistent skeleton contributions", (long long)p);
      u_dst.resize((size_t)uptr[p + 1]);
      for (int32_t q = qb; q < qe; ++q) u_dst[uptr[p] + s_uidx[q]] = q - qb;
      lds_front = std::max(lds_front, (int)((pp[p + 1] - pp[p] + p_nI[p] + uo + 2) * (int64_t)sizeof(double)));
      lds_back = std::max(lds_back, (int)((s + uo + 2) * (int64_t)sizeof(double)));
    }
    if (ch_patch.empty()) { ch_patch.push_back(0); ch_row.push_back(0); }
    if (xp_grp.empty()) xp_grp.push_back(0);
    if (bp_grp.empty()) bp_grp.push_back(0);
    if (u_dst.empty()) u_dst.push_back(0);
    if (lds_front > 150 * 1024)
      return alfi_set_error(ctx, ALFI_E_ARG, "condensed apply would need %d bytes of LDS per patch", lds_front);
    ALFI_CHECK(cond_upload(L, &cd.ch_patch, ch_patch));
    ALFI_CHECK(cond_upload(L, &cd.ch_row, ch_row));
    ALFI_CHECK(cond_upload(L, &cd.uptr, uptr));
    ALFI_CHECK(cond_upload(L, &cd.u_dst, u_dst));
    ALFI_CHECK(cond_upload(L, &cd.xp_ptr, xp_ptr));
    ALFI_CHECK(cond_upload(L, &cd.xp_grp, xp_grp));
    ALFI_CHECK(cond_upload(L, &cd.g_xp, g_xp));
    ALFI_CHECK(cond_upload(L, &cd.bp_ptr, bp_ptr));
    ALFI_CHECK(cond_upload(L, &cd.bp_grp, bp_grp));
    ALFI_CHECK(cond_upload(L, &cd.g_bp, g_bp));
    {
      // chunks of consecutive groups: at most 256 row pairs of X / W and of B each (a lane per pair), one descriptor per chunk
      // (CondChunk, common.h)
      std::vector<CondChunk> gc;
      std::vector<int64_t> gcptr((size_t)npatch + 1, 0), stage_off((size_t)npatch + 1, 0);
      for (int64_t p = 0; p < npatch; ++p)          // the staging layout of alfi_patches_set: ld_p = n_p rounded up to even
        stage_off[p + 1] = stage_off[p] + ((pp[p + 1] - pp[p] + 1) & ~(int64_t)1);
      int lds_gf = 0, lds_gb = 0;
      for (int64_t p = 0; p < npatch; ++p) {
        int64_t g = gptr[p];
        while (g < gptr[p + 1]) {
          CondChunk c;
          const int64_t ga = g;
          int xp = 0, bp = 0, ne = 0, nu = 0;
          while (g < gptr[p + 1] && xp + cond_pairs(g_m[g]) <= 256 && bp + cond_pairs(g_sc[g]) <= 256) {
            xp += cond_pairs(g_m[g]);
            bp += cond_pairs(g_sc[g]);
            ne += g_m[g];
            nu += g_sc[g];
            ++g;
          }
          c.off = pp[p]; c.ubase = uptr[p]; c.sidx0 = g_sidx[ga]; c.stage_off = stage_off[p];
          c.xq0 = (int32_t)(xp_ptr[p] + g_xp[ga]); c.xq1 = c.xq0 + xp;
          c.bq0 = (int32_t)(bp_ptr[p] + g_bp[ga]); c.bq1 = c.bq0 + bp;
          c.e0 = g_off[ga]; c.ne = ne; c.u0 = g_uoff[ga]; c.nu = nu; c.nI = p_nI[p]; c.pad = 0;
          c.xp0 = (int32_t)xp_ptr[p]; c.bp0 = (int32_t)bp_ptr[p];
          gc.push_back(c);
          lds_gf = std::max(lds_gf, (int)(2 * ne * sizeof(double)));
          lds_gb = std::max(lds_gb, (int)(nu * sizeof(double)));
        }
        gcptr[p + 1] = (int64_t)gc.size();
      }
      if (xp_ptr[npatch] > INT32_MAX || bp_ptr[npatch] > INT32_MAX)
        return alfi_set_error(ctx, ALFI_E_ARG, "condensed factors: too many row pairs on one level");
      if (gc.empty()) gc.push_back(CondChunk());
      ALFI_CHECK(cond_upload(L, &cd.gc, gc));
      L->h_cond_gcptr = gcptr;
      L->cond_lds_gfront = lds_gf + 16;
      L->cond_lds_gback = lds_gb + 16;
      ALFI_CHECK(dev_alloc(ctx, &cd.ubuf, uptr[npatch] > 0 ? uptr[npatch] : 1));
      L->cond_allocs.push_back(cd.ubuf);
    }
    L->h_cond_chptr = chptr;
    L->cond_lds_front = lds_front;
    L->cond_lds_back = lds_back;
    ALFI_CHECK(dev_alloc(ctx, &cd.tmp, sum_n > 0 ? sum_n : 1));
    L->cond_allocs.push_back(cd.tmp);
  }
  ALFI_CHECK(dev_alloc(ctx, &cd.mat, mat_off));
  L->cond_allocs.push_back(cd.mat);
  ALFI_CHECK(dev_alloc(ctx, &cd.sinv, sinv_off));
  L->cond_allocs.push_back(cd.sinv);
  // the dense inverses are not needed any more
  if (!L->inv_shrunk) {
    dev_free(L->inv);
    L->inv = nullptr;
    ALFI_CHECK(dev_alloc(ctx, &L->inv, 16));
    L->inv_shrunk = true;
  }
  L->cd = cd;
  L->cond = true;
  L->il_valid = false;
  L->h_sptr = sptr;
  L->h_cond_gptr = gptr;
  L->cond_ngroups = (int64_t)g_off.size();
  L->cond_mat_doubles = mat_off;
  L->cond_sinv_doubles = sinv_off;
  L->cond_lds_bytes = lds_max;
  L->cond_umax = umax;
  L->cond_max_s = smax;
  return 0;
}

int alfi_patches_factor_bytes(alfi_level* L, int64_t* bytes) {
  *bytes = L->cond ? 8 * (L->cond_mat_doubles + L->cond_sinv_doubles) : 8 * L->inv_doubles;
  return 0;
}

void free_mult_schedule(alfi_level* L) {
  dev_free(L->mult_items); dev_free(L->mult_pred0); dev_free(L->mult_pred); dev_free(L->mult_succ_ptr);
  dev_free(L->mult_succ); dev_free(L->mult_ctl); dev_free(L->mult_rowtab);
  L->mult_rowtab = nullptr;
  L->mult_items = L->mult_pred0 = L->mult_pred = L->mult_succ_ptr = L->mult_succ = L->mult_ctl = nullptr;
  L->mult_nitems = 0;
}

int alfi_patches_set_multiplicative(alfi_level* L, int64_t nit, const int64_t* iterset, int symmetrise) {
  alfi_ctx* ctx = L->ctx;
  if (!L->patch_ptr) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_patches_set_multiplicative before alfi_patches_set");
  ALFI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(L->mult_seq);
  L->mult_seq = nullptr;
  L->mult = false;
  L->mult_wave_ptr.clear();
  free_mult_schedule(L);
  // a new schedule starts with a clean device-side error word: after a timeout every later sweep on this ctx would stop at its
  // first wait (the word is tested inside the spin loop); (re)setting the sweeps is how a context recovers
  if (ctx->dev_err) ALFI_HIP_CHECK(ctx, hipMemset(ctx->dev_err, 0, 16));
  if (nit == 0) return 0;
  if (nit < 0 || !iterset) return alfi_set_error(ctx, ALFI_E_ARG, "bad iteration set");
  if (L->cond) return alfi_set_error(ctx, ALFI_E_STATE, "multiplicative sweeps need dense patch inverses (alfi_patches_set_groups(NULL))");
  if (L->pou) return alfi_set_error(ctx, ALFI_E_STATE, "partition of unity applies to the additive smoother");
  // partitioned levels: every rank sweeps over its own patches with the residual of its local vector (ghost slots hold
  // the rank's own contributions only) and the ghost contributions are added onto their owners at the end -- what
  // PCPATCH does under MPI: local Gauss-Seidel, additive between ranks [3P]
  if (nit > INT32_MAX) return alfi_set_error(ctx, ALFI_E_ARG, "iteration set too long");
  const int bs = L->bs;
  // patches must be unions of whole nodes (the sweep works on block rows); up to 64 nodes and 160 dofs a wave sweeps a
  // patch, beyond (macro stars) a workgroup does
  L->mult_big = L->max_np > SMALL_PATCH_MAX;
  for (int64_t p = 0; p < L->npatch; ++p) {
    const int64_t a = L->h_patch_ptr[p], b = L->h_patch_ptr[p + 1];
    if ((b - a) / bs > 64) L->mult_big = true;
    if ((b - a) % bs != 0)
      return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld: multiplicative sweeps need patches of whole nodes", (long long)p);
    for (int64_t q = a; q < b; ++q)
      if (L->h_patch_dofs[q] != (L->h_patch_dofs[a + ((q - a) / bs) * bs] / bs) * bs + (int32_t)((q - a) % bs))
        return alfi_set_error(ctx, ALFI_E_ARG, "patch %lld does not consist of whole nodes", (long long)p);
  }
  for (int64_t t = 0; t < nit; ++t)
    if (iterset[t] < 0 || iterset[t] >= L->npatch) return alfi_set_error(ctx, ALFI_E_ARG, "iteration set entry out of range");
  // sparsity of the operator on the host (row starts are marked in the sign bit of the flat layout)
  const int64_t nb = L->A.nbrows, nnzb = L->A.nnzb;
  std::vector<int32_t> rowptr(nb + 1), colidx(nnzb > 0 ? nnzb : 1);
  ALFI_HIP_CHECK(ctx, hipMemcpy(rowptr.data(), L->A.rowptr, sizeof(int32_t) * (nb + 1), hipMemcpyDeviceToHost));
  if (nnzb > 0) ALFI_HIP_CHECK(ctx, hipMemcpy(colidx.data(), L->A.colidx, sizeof(int32_t) * nnzb, hipMemcpyDeviceToHost));
  // wavefront of position t = 1 + max wavefront of earlier positions whose patch holds a node in the closure of patch t
  // (closure = columns of the patch's block rows).  node_wave[c] = last wavefront that wrote node c.
  std::vector<int32_t> node_wave(nb, -1), wave_of(nit);
  int32_t nwave = 0;
  for (int64_t t = 0; t < nit; ++t) {
    const int64_t p = iterset[t];
    const int64_t a = L->h_patch_ptr[p], b = L->h_patch_ptr[p + 1];
    int32_t w = -1;
    for (int64_t q = a; q < b; q += bs) {
      const int32_t node = L->h_patch_dofs[q] / bs;
      for (int32_t k = rowptr[node]; k < rowptr[node + 1]; ++k) {
        const int32_t c = colidx[k] & 0x7fffffff;
        if (node_wave[c] > w) w = node_wave[c];
      }
    }
    ++w;
    wave_of[t] = w;
    if (w + 1 > nwave) nwave = w + 1;
    for (int64_t q = a; q < b; q += bs) node_wave[L->h_patch_dofs[q] / bs] = w;
  }
  // counting sort of the positions by wavefront (stable)
  L->mult_wave_ptr.assign(nwave + 1, 0);
  for (int64_t t = 0; t < nit; ++t) L->mult_wave_ptr[wave_of[t] + 1]++;
  for (int32_t w = 0; w < nwave; ++w) L->mult_wave_ptr[w + 1] += L->mult_wave_ptr[w];
  std::vector<int32_t> seq(nit);
  {
    std::vector<int64_t> fill(L->mult_wave_ptr.begin(), L->mult_wave_ptr.end() - 1);
    for (int64_t t = 0; t < nit; ++t) seq[fill[wave_of[t]]++] = (int32_t)iterset[t];
  }
  ALFI_CHECK(dev_upload(ctx, &L->mult_seq, seq.data(), nit));
  L->mult = true;
  L->mult_symmetrise = symmetrise != 0;
  // ---- the persistent schedule (wave-per-patch levels): items = the forward sweep in wavefront-major order, then (symmetrised)
  // the wavefronts in reverse order, each in its listed order -- exactly the launch sequence of the per-wavefront schedule.
  // Predecessors of an item = the LAST WRITERS (earlier items) of the nodes it reads: patches writing the same node conflict
  // with each other, so earlier writers are ordered before the last one transitively; a patch that READS a node this item
  // writes has, by the symmetric sparsity, nodes in this item's closure, whose last writer is that patch or a later conflicting
  // one.  The list order is a topological order of these dependencies.
  free_mult_schedule(L);
  {
    if (!L->mult_big) {
      // row table of the sweep kernels (kernels_patch.hip, mult_wg_rows): per patch node its first block, block count and node
      std::vector<int32_t> rowtab((size_t)L->npatch * 64 * 3, 0);
      for (int64_t p = 0; p < L->npatch; ++p) {
        const int64_t a = L->h_patch_ptr[p], b = L->h_patch_ptr[p + 1];
        for (int64_t q = a, i = 0; q < b; q += bs, ++i) {
          const int32_t node = L->h_patch_dofs[q] / bs;
          int32_t* rt = &rowtab[((size_t)p * 64 + (size_t)i) * 3];
          rt[0] = rowptr[node];
          rt[1] = rowptr[node + 1] - rowptr[node];
          rt[2] = node;
        }
      }
      ALFI_CHECK(dev_upload(ctx, &L->mult_rowtab, rowtab.data(), (int64_t)rowtab.size()));
    }
    std::vector<int32_t> items(seq);
    if (symmetrise)
      for (int32_t w = nwave - 1; w >= 0; --w)
        for (int64_t q = L->mult_wave_ptr[w]; q < L->mult_wave_ptr[w + 1]; ++q) items.push_back(seq[q]);
    const int64_t N = (int64_t)items.size();
    if (N > INT32_MAX / 2) return alfi_set_error(ctx, ALFI_E_ARG, "iteration set too long for the persistent schedule");
    std::vector<int32_t> last_writer(nb, -1), pred0(N, 0), tmp;
    std::vector<std::vector<int32_t>> succ_of;     // built as (pred, item) pairs to keep memory flat
    std::vector<int32_t> e_from, e_to;
    e_from.reserve((size_t)N * 32);
    e_to.reserve((size_t)N * 32);
    for (int64_t t = 0; t < N; ++t) {
      const int64_t p = items[t];
      const int64_t a = L->h_patch_ptr[p], b = L->h_patch_ptr[p + 1];
      tmp.clear();
      for (int64_t q = a; q < b; q += bs) {
        const int32_t node = L->h_patch_dofs[q] / bs;
        for (int32_t k = rowptr[node]; k < rowptr[node + 1]; ++k) {
          const int32_t lw = last_writer[colidx[k] & 0x7fffffff];
          if (lw >= 0) tmp.push_back(lw);
        }
      }
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      pred0[t] = (int32_t)tmp.size();
      for (int32_t f : tmp) {
        e_from.push_back(f);
        e_to.push_back((int32_t)t);
      }
      for (int64_t q = a; q < b; q += bs) last_writer[L->h_patch_dofs[q] / bs] = (int32_t)t;
    }
    if (e_from.size() > (size_t)INT32_MAX) return alfi_set_error(ctx, ALFI_E_ARG, "too many dependencies for int32 offsets");
    std::vector<int32_t> succ_ptr(N + 1, 0), succ(e_from.size() > 0 ? e_from.size() : 1);
    for (int32_t f : e_from) succ_ptr[f + 1]++;
    for (int64_t t = 0; t < N; ++t) succ_ptr[t + 1] += succ_ptr[t];
    {
      std::vector<int32_t> fill(succ_ptr.begin(), succ_ptr.end() - 1);
      for (size_t e = 0; e < e_from.size(); ++e) succ[fill[e_from[e]]++] = e_to[e];
    }
    ALFI_CHECK(dev_upload(ctx, &L->mult_items, items.data(), N));
    ALFI_CHECK(dev_upload(ctx, &L->mult_pred0, pred0.data(), N));
    ALFI_CHECK(dev_alloc(ctx, &L->mult_pred, N));
    ALFI_CHECK(dev_upload(ctx, &L->mult_succ_ptr, succ_ptr.data(), N + 1));
    ALFI_CHECK(dev_upload(ctx, &L->mult_succ, succ.data(), (int64_t)succ.size()));
    ALFI_CHECK(dev_alloc(ctx, &L->mult_ctl, 4));
    L->mult_nitems = (int32_t)N;
  }
  return 0;
}

int alfi_patches_set_partition_of_unity(alfi_level* L, int on) {
  if (on && L->mult) return alfi_set_error(L->ctx, ALFI_E_STATE, "partition of unity applies to the additive smoother");
  L->pou = on != 0;
  return 0;
}

int alfi_patches_multiplicative_levels(alfi_level* L, int64_t* nwave) {
  *nwave = L->mult ? (int64_t)L->mult_wave_ptr.size() - 1 : 0;
  return 0;
}

int alfi_patches_factor(alfi_level* L) {
  alfi_ctx* ctx = L->ctx;
  if (!L->patch_ptr) return alfi_set_error(ctx, ALFI_E_STATE, "alfi_patches_factor before alfi_patches_set");
  int t = alfi_prof_begin(ctx, ALFI_EV_PATCH_FACTOR);
  ALFI_HIP_CHECK(ctx, hipMemsetAsync(L->status, 0, sizeof(int), ctx->stream));
  if (!L->cond && L->inv_shrunk) {                 // first dense factorisation of this patch set
    ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    dev_free(L->inv);
    L->inv = nullptr;
    ALFI_CHECK(dev_alloc(ctx, &L->inv, L->inv_doubles));
    L->inv_shrunk = false;
  }
  if (L->cond) {
    ALFI_CHECK(launch_cond_factor(L));            // condensed factors: group inverses + Schur complements
  } else if (L->max_np > SMALL_PATCH_MAX) {
    ALFI_CHECK(launch_big_factor(L));             // macro-star sized patches: blocked Gauss-Jordan on the matrix cores
  } else {
    ALFI_CHECK(launch_patch_gather_dense(L));
    ALFI_CHECK(launch_patch_invert(L));
  }
  ALFI_CHECK(build_patch_il(L));                  // small-patch levels: the wave-contiguous copy the apply streams
  alfi_prof_end(ctx, t);
  int st = 0;
  ALFI_HIP_CHECK(ctx, hipMemcpyAsync(&st, L->status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  // every stored inverse is probed (|| A_p X_p e - e ||); the ones that fail -- an unpivoted elimination met a zero or
  // tiny pivot -- are re-inverted with partial pivoting, as the reference's LAPACK / UMFPACK factorisations would
  ALFI_CHECK(patch_verify_and_repair(L, st));
  L->factored = true;
  return 0;
}

int alfi_patches_check(alfi_level* L, double* worst_residual, int64_t* flagged, int64_t* repaired, double* worst_after) {
  if (!L->factored) return alfi_set_error(L->ctx, ALFI_E_STATE, "alfi_patches_check before alfi_patches_factor");
  if (worst_residual) *worst_residual = L->chk_worst;
  if (flagged) *flagged = L->chk_flagged;
  if (repaired) *repaired = L->chk_repaired;
  if (worst_after) *worst_after = L->chk_flagged > 0 ? L->chk_worst_after : L->chk_worst;
  return 0;
}

int alfi_patch_apply(alfi_level* L, const double* dx, double* dy) {
  if (!L->factored) return alfi_set_error(L->ctx, ALFI_E_STATE, "alfi_patch_apply before alfi_patches_factor");
  if (dx == dy) return alfi_set_error(L->ctx, ALFI_E_ARG, "alfi_patch_apply: x and y must not alias");
  L->ctx->cur_tag = L->id;
  return level_patch_apply(L, dx, dy);
}

int alfi_patches_stats(alfi_level* L, int64_t* npatch, int64_t* sum_n, int64_t* sum_n2) {
  if (npatch) *npatch = L->npatch;
  if (sum_n) *sum_n = L->sum_n;
  if (sum_n2) *sum_n2 = L->sum_n2;
  return 0;
}

int alfi_patch_get_inverse(alfi_level* L, int64_t p, double* out) {
  alfi_ctx* ctx = L->ctx;
  if (!L->factored) return alfi_set_error(ctx, ALFI_E_STATE, "patches not factored");
  if (p < 0 || p >= L->npatch) return alfi_set_error(ctx, ALFI_E_ARG, "patch index out of range");
  if (L->cond) return alfi_set_error(ctx, ALFI_E_STATE, "condensed patch factors hold no dense inverse");
  const int64_t n = L->h_patch_ptr[p + 1] - L->h_patch_ptr[p];
  const int64_t ld = (n + 1) & ~(int64_t)1;
  std::vector<double> tmp(n * ld);
  ALFI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ALFI_HIP_CHECK(ctx, hipMemcpy(tmp.data(), L->inv + L->h_inv_ptr[p], tmp.size() * sizeof(double),
                                hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < n; ++j) out[i * n + j] = tmp[patch_inv_index((int)i, (int)j, (int)n, (int)ld)];
  return 0;
}
