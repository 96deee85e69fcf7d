// Level SpMV / residual (PETSc MatMult on BAIJ), FGMRES vector kernels, small dense block kernels of the Schoeberl
// transfer and the dense coarse GEMV.  All HBM-bound; wave64, FP64.
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "hs_layout.h"

// ---------------------------------------------------------------------------------------------------------------------
// BSR SpMV: LPR lanes cooperate on one block row (each lane takes whole bs x bs blocks), shuffle-reduce the bs sums.
//   mode 0: y = A x          mode 1: y = b - alpha A x
// Algorithmic bytes: (8 bs^2 + 4) nnzb + 4 (nbrows + 1) + 16 n   (SURVEY.md section 8(d)).
// ---------------------------------------------------------------------------------------------------------------------
template <int BS, int LPR>
__global__ __launch_bounds__(256) void bsr_spmv_kernel(int64_t nbrows, const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ colidx,
                                                        const double* __restrict__ vals, const double* __restrict__ x,
                                                        double* __restrict__ y, const double* __restrict__ b,
                                                        double alpha, int mode) {
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPR;
  const int l = threadIdx.x % LPR;
  double acc[BS];
#pragma unroll
  for (int r = 0; r < BS; ++r) acc[r] = 0.0;
  if (row < nbrows) {
    const int32_t lo = rowptr[row], hi = rowptr[row + 1];
    for (int32_t k = lo + l; k < hi; k += LPR) {
      const int64_t col = colidx[k];
      const double* v = vals + (int64_t)k * BS * BS;
      double xv[BS];
#pragma unroll
      for (int c = 0; c < BS; ++c) xv[c] = x[col * BS + c];
#pragma unroll
      for (int r = 0; r < BS; ++r)
#pragma unroll
        for (int c = 0; c < BS; ++c) acc[r] = __builtin_fma(v[r * BS + c], xv[c], acc[r]);
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < BS; ++r) acc[r] += __shfl_xor(acc[r], o);
  if (row < nbrows && l == 0) {
#pragma unroll
    for (int r = 0; r < BS; ++r) {
      const int64_t i = row * BS + r;
      y[i] = mode == 0 ? acc[r] : b[i] - alpha * acc[r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// nnz-balanced BSR SpMV on the lane-major layout (DevBSR::flat): a wave takes SPMV_CHUNK consecutive blocks whatever rows
// they belong to (block rows of the P2+FB operator hold 14 .. 125 blocks, so one-wave-per-row leaves most lanes idle), lane
// l of iteration u owns block base + 64 u + l: one coalesced 256-B colidx request and (bs*bs)/2 coalesced 1-KiB value
// requests (16 B per lane) per 64 blocks, x gathered through L2.  Per-row sums by a segmented inclusive scan over the wave (row starts = sign bit
// of colidx); a row running on into the next iteration travels in a wave-uniform carry, a row running on into the next
// chunk leaves its partial sum in carry[chunk] and bsr_spmv_fixup_kernel adds it to the row in chunk order
// (deterministic, no atomics).  The wave holding a row's LAST block stores y, everything earlier is a carry.
// ---------------------------------------------------------------------------------------------------------------------
typedef double spmv_d2 __attribute__((ext_vector_type(2)));
// ALIGNED (small matrices, where a cycle is bound by the number of dependent launches rather than by bandwidth): the chunks
// hold whole block rows -- chunk c = blocks [chunk_start[c], chunk_start[c + 1]), at most SPMV_CHUNK of them -- so no row
// continues into the next chunk and the fix-up launch disappears.
template <int BS, bool NT, bool ALIGNED>
__global__ __launch_bounds__(256) void bsr_spmv_flat_kernel(int64_t kbase, int64_t nnzb, int64_t nchunks,
                                                             const int64_t* __restrict__ chunk_start,
                                                             const int32_t* __restrict__ colflag,
                                                             const double* __restrict__ vals,
                                                             const int32_t* __restrict__ chunk_row,
                                                             const double* __restrict__ x, double* __restrict__ y,
                                                             const double* __restrict__ b, double alpha, int mode,
                                                             double* __restrict__ carry_out,
                                                             int32_t* __restrict__ carry_row, int xcd_map) {
  constexpr int BB = BS * BS;
  const int lane = threadIdx.x & 63;
  // Optional XCD-aware block -> chunk map (ALFI_XCD_MAP=1): workgroups go round-robin over the 8 XCDs (block b lands on
  // XCD b % 8), each with its own L2; the map gives every XCD one contiguous eighth of the chunk range so that x entries
  // shared by neighbouring rows are re-read from the same L2.  MEASURED (config 4): 1 % SLOWER than the plain
  // interleaved order (5.44 vs 5.50 TB/s, same box) -- x already lives in the Infinity Cache and the interleaved order
  // spreads the value stream over the HBM channels more evenly -- so it is off by default.
  int64_t blk = blockIdx.x;
  if (xcd_map == 1) {
    const int64_t nb = gridDim.x, per = nb >> 3, rem = nb & 7, xcd = blk & 7, idx = blk >> 3;
    blk = xcd * per + (xcd < rem ? xcd : rem) + idx;
  }
  const int64_t chunk = blk * 4 + (threadIdx.x >> 6);
  if (chunk >= nchunks) return;
  // blocks [kbase, kbase + nnzb) of the upload: a block-row range (the whole matrix, the owned rows, or the owned rows
  // with / without ghost columns of a partitioned level); the value / index layout is addressed by the absolute block
  const int64_t kend_all = ALIGNED ? chunk_start[chunk + 1] : kbase + nnzb;
  const int64_t base = ALIGNED ? chunk_start[chunk] : kbase + chunk * SPMV_CHUNK;
  int R = chunk_row[chunk];  // block row of lane 0's block
  int Rlast = R;
  double carry[BS];
#pragma unroll
  for (int r = 0; r < BS; ++r) carry[r] = 0.0;
  auto store_row = [&](int row, const double* s) {
#pragma unroll
    for (int r = 0; r < BS; ++r) {
      const int64_t i = (int64_t)row * BS + r;
      y[i] = mode == 0 ? s[r] : b[i] - alpha * s[r];
    }
  };
#pragma unroll
  for (int u = 0; u < SPMV_U; ++u) {
    if (ALIGNED && base + u * 64 >= kend_all) break;     // wave-uniform: the chunk is shorter than SPMV_CHUNK
    const int64_t k = base + u * 64 + lane;
    const bool valid = k < kend_all;
    const int32_t cf = valid ? (NT ? __builtin_nontemporal_load(colflag + k) : colflag[k]) : 0;
    const bool head = cf < 0;
    const int64_t col = cf & 0x7fffffff;
    double p[BS];
#pragma unroll
    for (int r = 0; r < BS; ++r) p[r] = 0.0;
    if (valid) {
      const double* v = vals + (k >> 6) * (64 * BB);
      const int kl = (int)(k & 63);        // position inside the 64-block group (== lane unless the view starts mid-group)
      double a[BB], xv[BS];
#pragma unroll
      for (int q = 0; q < BB / 2; ++q) {   // pair planes: one aligned 16-byte load per lane
        const spmv_d2* pq = reinterpret_cast<const spmv_d2*>(v + q * 128) + kl;
        const spmv_d2 t = NT ? __builtin_nontemporal_load(pq) : *pq;
        a[2 * q] = t.x;
        a[2 * q + 1] = t.y;
      }
      if (BB & 1) {
        const double* pl = v + (BB / 2) * 128 + kl;
        a[BB - 1] = NT ? __builtin_nontemporal_load(pl) : *pl;
      }
#pragma unroll
      for (int c = 0; c < BS; ++c) xv[c] = x[col * BS + c];
#pragma unroll
      for (int r = 0; r < BS; ++r)
#pragma unroll
        for (int c = 0; c < BS; ++c) p[r] = __builtin_fma(a[r * BS + c], xv[c], p[r]);
    }
    const unsigned long long hm = __ballot(head);
    if (u > 0) {
      // lane 0 starts a new row: the row carried over from the previous iteration is complete
      if ((hm & 1ull) && lane == 0) store_row(Rlast, carry);
      R = Rlast + (int)(hm & 1ull);
    }
    const int row = R + __popcll(hm & ((2ull << lane) - 2ull));  // row starts at lanes 1 .. lane
    // segmented inclusive scan; f = a row start lies at or before this lane (in this iteration)
    int f = head ? 1 : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      double tp[BS];
#pragma unroll
      for (int r = 0; r < BS; ++r) tp[r] = __shfl_up(p[r], d);
      const int tf = __shfl_up(f, d);
      if (lane >= d && !f) {
#pragma unroll
        for (int r = 0; r < BS; ++r) p[r] += tp[r];
        f = tf;
      }
    }
    if (!f) {
#pragma unroll
      for (int r = 0; r < BS; ++r) p[r] += carry[r];
    }
    // rows ending inside this iteration (the next lane starts a new one); lane 63's row is carried on
    if (lane < 63 && ((hm >> (lane + 1)) & 1ull)) store_row(row, p);
#pragma unroll
    for (int r = 0; r < BS; ++r) carry[r] = __shfl(p[r], 63);
    Rlast = __shfl(row, 63);
  }
  if (ALIGNED) {
    if (lane == 0 && kend_all > base) store_row(Rlast, carry);
    return;
  }
  const int64_t kend = base + SPMV_CHUNK;
  if (lane == 0) {
    const bool closed = kend >= kend_all || colflag[kend] < 0;
    if (closed) {
      store_row(Rlast, carry);
      carry_row[chunk] = -1;
    } else {
#pragma unroll
      for (int r = 0; r < BS; ++r) carry_out[chunk * BS + r] = carry[r];
      carry_row[chunk] = Rlast;
    }
  }
}

template <int BS>
__global__ __launch_bounds__(256) void bsr_spmv_fixup_kernel(int64_t nchunks, const double* __restrict__ carry,
                                                              const int32_t* __restrict__ carry_row,
                                                              double* __restrict__ y, double alpha, int mode) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= nchunks) return;
  const int32_t R = carry_row[c];
  if (R < 0 || (c > 0 && carry_row[c - 1] == R)) return;  // only the first chunk of a run adds, in chunk order
  double s[BS];
#pragma unroll
  for (int r = 0; r < BS; ++r) s[r] = 0.0;
  for (int64_t cc = c; cc < nchunks && carry_row[cc] == R; ++cc)
#pragma unroll
    for (int r = 0; r < BS; ++r) s[r] += carry[cc * BS + r];
#pragma unroll
  for (int r = 0; r < BS; ++r) y[(int64_t)R * BS + r] += mode == 0 ? s[r] : -alpha * s[r];
}

// ---------------------------------------------------------------------------------------------------------------------
// The same product with the x gathers de-duplicated per workgroup (DevBSR::dedup).  A workgroup owns SPMV_WG = 1024
// consecutive blocks (4 waves x 256); neighbouring block rows of the Morton-numbered operators share most of their columns
// (config 4: 247 distinct columns among 1024 blocks), so the workgroup first stages the distinct x entries of its group in
// LDS -- one 8 bs-byte gather per DISTINCT column instead of one per block: a quarter of the scattered requests -- and the
// waves then take x from LDS through a 2-byte local index (bit 15 = first block of a block row; 2 + ~1 bytes of index
// stream per block instead of 4).  Everything else (lane-major value planes, segmented scan, carries, fix-up) as above.
// ---------------------------------------------------------------------------------------------------------------------
template <int BS>
__global__ __launch_bounds__(256) void bsr_spmv_dedup_kernel(int64_t nnzb, int64_t nchunks,
                                                              const uint16_t* __restrict__ lidx,
                                                              const int32_t* __restrict__ ucol,
                                                              const int32_t* __restrict__ uptr,
                                                              const int32_t* __restrict__ colflag,
                                                              const double* __restrict__ vals,
                                                              const int32_t* __restrict__ chunk_row,
                                                              const double* __restrict__ x, double* __restrict__ y,
                                                              const double* __restrict__ b, double alpha, int mode,
                                                              double* __restrict__ carry_out,
                                                              int32_t* __restrict__ carry_row, int xcd_map) {
  constexpr int BB = BS * BS;
  __shared__ double xs[SPMV_DEDUP_MAX * BS];
  const int lane = threadIdx.x & 63;
  int64_t g = blockIdx.x;
  if (xcd_map == 1) {      // workgroup b runs on XCD b % 8: give every XCD (= every L2) one contiguous eighth of the groups
    const int64_t nb = gridDim.x, per = nb >> 3, rem = nb & 7, xcd = g & 7, idx = g >> 3;
    g = xcd * per + (xcd < rem ? xcd : rem) + idx;
  } else if (xcd_map > 1) {
    // strips: XCD x takes the groups of strips x, x + 8, x + 16, ... (xcd_map groups each): rows that share x entries meet
    // in ONE L2, while at any moment the eight XCDs stream eight neighbouring strips -- the value stream stays spread over
    // the HBM channels, which the contiguous-eighth map above loses.  The tail (last incomplete round of strips) keeps the
    // plain order.
    const int64_t S = xcd_map, nb = gridDim.x, full = nb / (8 * S) * (8 * S);
    if (g < full) {
      const int64_t xcd = g & 7, i = g >> 3, strip = i / S, within = i - strip * S;
      g = (strip * 8 + xcd) * S + within;
    }
  }
  const int64_t chunk = g * 4 + (threadIdx.x >> 6);
  const bool active = chunk < nchunks;
  const int64_t base = chunk * SPMV_CHUNK;
  // value planes of the first iteration are requested before the staging barrier (independent of x)
  const int32_t u0 = uptr[g], nu = uptr[g + 1] - u0;
  for (int i = threadIdx.x; i < nu; i += 256) {
    const int64_t c = __builtin_nontemporal_load(ucol + u0 + i);
#pragma unroll
    for (int q = 0; q < BS; ++q) xs[i * BS + q] = x[c * BS + q];
  }
  __syncthreads();
  if (!active) return;
  int R = chunk_row[chunk];
  int Rlast = R;
  double carry[BS];
#pragma unroll
  for (int r = 0; r < BS; ++r) carry[r] = 0.0;
  auto store_row = [&](int row, const double* s) {
#pragma unroll
    for (int r = 0; r < BS; ++r) {
      const int64_t i = (int64_t)row * BS + r;
      y[i] = mode == 0 ? s[r] : b[i] - alpha * s[r];
    }
  };
#pragma unroll
  for (int u = 0; u < SPMV_U; ++u) {
    const int64_t k = base + u * 64 + lane;
    const bool valid = k < nnzb;
    const uint16_t li = valid ? __builtin_nontemporal_load(lidx + k) : (uint16_t)0;
    const bool head = (li & 0x8000u) != 0;
    const int lc = li & 0x7fff;
    double p[BS];
#pragma unroll
    for (int r = 0; r < BS; ++r) p[r] = 0.0;
    if (valid) {
      const double* v = vals + (k >> 6) * (64 * BB);
      double a[BB], xv[BS];
#pragma unroll
      for (int q = 0; q < BB / 2; ++q) {
        const spmv_d2 t = __builtin_nontemporal_load(reinterpret_cast<const spmv_d2*>(v + q * 128) + lane);
        a[2 * q] = t.x;
        a[2 * q + 1] = t.y;
      }
      if (BB & 1) a[BB - 1] = __builtin_nontemporal_load(v + (BB / 2) * 128 + lane);
#pragma unroll
      for (int c = 0; c < BS; ++c) xv[c] = xs[lc * BS + c];
#pragma unroll
      for (int r = 0; r < BS; ++r)
#pragma unroll
        for (int c = 0; c < BS; ++c) p[r] = __builtin_fma(a[r * BS + c], xv[c], p[r]);
    }
    const unsigned long long hm = __ballot(head);
    if (u > 0) {
      if ((hm & 1ull) && lane == 0) store_row(Rlast, carry);
      R = Rlast + (int)(hm & 1ull);
    }
    const int row = R + __popcll(hm & ((2ull << lane) - 2ull));
    int f = head ? 1 : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      double tp[BS];
#pragma unroll
      for (int r = 0; r < BS; ++r) tp[r] = __shfl_up(p[r], d);
      const int tf = __shfl_up(f, d);
      if (lane >= d && !f) {
#pragma unroll
        for (int r = 0; r < BS; ++r) p[r] += tp[r];
        f = tf;
      }
    }
    if (!f) {
#pragma unroll
      for (int r = 0; r < BS; ++r) p[r] += carry[r];
    }
    if (lane < 63 && ((hm >> (lane + 1)) & 1ull)) store_row(row, p);
#pragma unroll
    for (int r = 0; r < BS; ++r) carry[r] = __shfl(p[r], 63);
    Rlast = __shfl(row, 63);
  }
  const int64_t kend = base + SPMV_CHUNK;
  if (lane == 0) {
    const bool closed = kend >= nnzb || colflag[kend] < 0;
    if (closed) {
      store_row(Rlast, carry);
      carry_row[chunk] = -1;
    } else {
#pragma unroll
      for (int r = 0; r < BS; ++r) carry_out[chunk * BS + r] = carry[r];
      carry_row[chunk] = Rlast;
    }
  }
}

// Setup of the de-duplication tables, one workgroup per group of SPMV_WG blocks: bitonic sort of the group's columns in LDS,
// distinct values flagged and counted, every block's column located by binary search.  Pass 1 leaves the distinct columns
// of group g at tmp[g * SPMV_WG ..) and their number in nuniq[g]; the host scans nuniq; pass 2 compacts.
__global__ __launch_bounds__(256) void spmv_dedup_sort_kernel(int64_t nnzb, const int32_t* __restrict__ colflag,
                                                               uint16_t* __restrict__ lidx, int32_t* __restrict__ tmp,
                                                               int32_t* __restrict__ nuniq) {
  static_assert(SPMV_WG == 1024, "the LDS sort below is written for 1024 keys");
  __shared__ int32_t key[SPMV_WG];
  __shared__ int32_t uniq[SPMV_WG];
  __shared__ int32_t cnt;
  const int64_t k0 = (int64_t)blockIdx.x * SPMV_WG;
  const int t = threadIdx.x;
  if (t == 0) cnt = 0;
  for (int i = t; i < SPMV_WG; i += 256) key[i] = k0 + i < nnzb ? (colflag[k0 + i] & 0x7fffffff) : 0x7fffffff;
  __syncthreads();
  for (int size = 2; size <= SPMV_WG; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = t; i < SPMV_WG / 2; i += 256) {
        const int lo = 2 * i - (i & (stride - 1));       // element with bit `stride` clear
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const int32_t a = key[lo], c = key[hi];
        if ((a > c) == up) {
          key[lo] = c;
          key[hi] = a;
        }
      }
      __syncthreads();
    }
  // distinct values, in order: position = number of distinct values in front (serial prefix per thread chunk of 4 + scan)
  int flag[4], local = 0;
  for (int q = 0; q < 4; ++q) {
    const int i = t * 4 + q;
    flag[q] = key[i] != 0x7fffffff && (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
    local += flag[q];
  }
  __shared__ int32_t part[256];
  part[t] = local;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int pos = part[t] - local;
  for (int q = 0; q < 4; ++q)
    if (flag[q]) uniq[pos++] = key[t * 4 + q];
  if (t == 255) cnt = part[255];
  __syncthreads();
  const int nu = cnt;
  for (int i = t; i < nu; i += 256) tmp[k0 + i] = uniq[i];
  if (t == 0) nuniq[blockIdx.x] = nu;
  for (int i = t; i < SPMV_WG; i += 256) {
    const int64_t k = k0 + i;
    if (k >= nnzb) break;
    const int32_t cf = colflag[k];
    const int32_t c = cf & 0x7fffffff;
    int lo = 0, hi = nu - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (uniq[mid] < c) lo = mid + 1;
      else hi = mid;
    }
    lidx[k] = (uint16_t)(lo | (cf < 0 ? 0x8000 : 0));
  }
}

__global__ __launch_bounds__(256) void spmv_dedup_compact_kernel(const int32_t* __restrict__ tmp,
                                                                  const int32_t* __restrict__ uptr,
                                                                  int32_t* __restrict__ ucol) {
  const int64_t g = blockIdx.x;
  const int32_t u0 = uptr[g], nu = uptr[g + 1] - u0;
  for (int i = threadIdx.x; i < nu; i += 256) ucol[u0 + i] = tmp[g * SPMV_WG + i];
}

int build_spmv_dedup(alfi_ctx* ctx, DevBSR* d) {
  d->dedup = false;
  if (!d->flat || d->aligned || d->nnzb < SPMV_WG) return 0;
  const int64_t nnzb = d->nnzb, ng = (nnzb + SPMV_WG - 1) / SPMV_WG;
  int32_t *tmp = nullptr, *nuniq = nullptr;
  ALFI_HIP_CHECK(ctx, hipMalloc((void**)&d->lidx, (size_t)nnzb * sizeof(uint16_t)));
  hipError_t e = hipMalloc((void**)&tmp, (size_t)ng * SPMV_WG * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&nuniq, (size_t)ng * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&d->uptr, (size_t)(ng + 1) * sizeof(int32_t));
  std::vector<int32_t> uptr((size_t)ng + 1, 0);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(spmv_dedup_sort_kernel, dim3((unsigned)ng), dim3(256), 0, ctx->stream, nnzb, d->colidx, d->lidx, tmp,
                       nuniq);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(uptr.data() + 1, nuniq, (size_t)ng * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  bool fits = e == hipSuccess;
  if (fits) {
    int64_t tot = 0;
    for (int64_t g = 0; g < ng; ++g) {
      tot += uptr[g + 1];
      fits = fits && tot <= INT32_MAX;
      uptr[g + 1] = (int32_t)tot;
    }
  }
  if (fits) {
    e = hipMalloc((void**)&d->ucol, (size_t)std::max<int64_t>(uptr[ng], 1) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpyAsync(d->uptr, uptr.data(), (size_t)(ng + 1) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(spmv_dedup_compact_kernel, dim3((unsigned)ng), dim3(256), 0, ctx->stream, tmp, d->uptr, d->ucol);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  (void)hipFree(tmp);
  (void)hipFree(nuniq);
  if (e != hipSuccess || !fits) {
    (void)hipFree(d->lidx);
    (void)hipFree(d->ucol);
    (void)hipFree(d->uptr);
    d->lidx = nullptr;
    d->ucol = nullptr;
    d->uptr = nullptr;
    if (e != hipSuccess) return alfi_set_error(ctx, ALFI_E_HIP, "SpMV de-duplication tables: %s", hipGetErrorString(e));
    return 0;
  }
  d->dedup = true;
  return 0;
}

// host-layout values of blocks [k0, k0 + nblk) -> lane-major
__global__ void bsr_vals_to_lanes_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t k0,
                                         int64_t total, int bb) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = k0 + e / bb;
    const int rc = (int)(e % bb);
    dst[bsr_val_index(1, k, rc, bb)] = src[e];
  }
}

int upload_bsr_values(alfi_ctx* ctx, DevBSR* d, const double* host_vals) {
  const int bb = d->bs * d->bs;
  if (d->nnzb == 0) return 0;
  if (!d->flat) {
    ALFI_HIP_CHECK(ctx, hipMemcpy(d->vals, host_vals, (size_t)d->nnzb * bb * sizeof(double), hipMemcpyHostToDevice));
    return 0;
  }
  const int64_t slab = (int64_t)1 << 22;  // blocks per staging slab (<= 302 MB at bs = 3)
  const int64_t nslab = d->nnzb < slab ? d->nnzb : slab;
  double* stage = nullptr;
  ALFI_HIP_CHECK(ctx, hipMalloc((void**)&stage, (size_t)nslab * bb * sizeof(double)));
  for (int64_t k0 = 0; k0 < d->nnzb; k0 += slab) {
    const int64_t nb = d->nnzb - k0 < slab ? d->nnzb - k0 : slab;
    hipError_t e = hipMemcpyAsync(stage, host_vals + k0 * bb, (size_t)nb * bb * sizeof(double), hipMemcpyHostToDevice,
                                  ctx->stream);
    if (e == hipSuccess) {
      int64_t blocks = (nb * bb + 255) / 256;
      if (blocks > 65536) blocks = 65536;
      hipLaunchKernelGGL(bsr_vals_to_lanes_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, stage, d->vals, k0,
                         nb * bb, bb);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      (void)hipFree(stage);
      return alfi_set_error(ctx, ALFI_E_HIP, "value upload failed: %s", hipGetErrorString(e));
    }
  }
  ALFI_HIP_CHECK(ctx, hipFree(stage));
  return 0;
}

template <int BS>
static int launch_bsr_spmv_bs(alfi_ctx* ctx, const DevBSR& A, const double* x, double* y, const double* b, double alpha,
                              int mode) {
  if (A.flat) {
    if (A.nnzb == 0) return 0;
    // operator values and indices are used once per product: they are streamed past the caches (nontemporal loads) so that x
    // and y keep the L2 / Infinity Cache.
    // Workgroups are dealt to the 8 XCDs in strips of 64 (de-duplicated kernel): config 4 finest, same box, 6.29 -> 6.45 TB/s
    // for any strip size 16 .. 1024; plain interleaved order and contiguous eighths measured slower
    constexpr int xcd = 64;
    if (A.aligned) {          // whole rows per chunk: one launch, no fix-up
      const int64_t nchunks = A.nchunks;
      hipLaunchKernelGGL((bsr_spmv_flat_kernel<BS, true, true>), dim3((unsigned)((nchunks + 3) / 4)), dim3(256), 0,
                         ctx->stream, A.kbase, A.nnzb, nchunks, A.chunk_start, A.colidx, A.vals, A.chunk_row, x, y, b,
                         alpha, mode, A.carry, A.carry_row, 0);
      ALFI_HIP_CHECK(ctx, hipGetLastError());
      return 0;
    }
    const int64_t nchunks = (A.nnzb + SPMV_CHUNK - 1) / SPMV_CHUNK;   // A may be a block-row-range view of the upload
    if (A.dedup && !A.view && A.kbase == 0)
      hipLaunchKernelGGL((bsr_spmv_dedup_kernel<BS>), dim3((unsigned)((nchunks + 3) / 4)), dim3(256), 0, ctx->stream,
                         A.nnzb, nchunks, A.lidx, A.ucol, A.uptr, A.colidx, A.vals, A.chunk_row, x, y, b, alpha, mode,
                         A.carry, A.carry_row, xcd);
    else
      hipLaunchKernelGGL((bsr_spmv_flat_kernel<BS, true, false>), dim3((unsigned)((nchunks + 3) / 4)), dim3(256), 0,
                         ctx->stream, A.kbase, A.nnzb, nchunks, (const int64_t*)nullptr, A.colidx, A.vals, A.chunk_row, x,
                         y, b, alpha, mode, A.carry, A.carry_row, xcd);
    ALFI_HIP_CHECK(ctx, hipGetLastError());
    hipLaunchKernelGGL((bsr_spmv_fixup_kernel<BS>), dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, ctx->stream,
                       nchunks, A.carry, A.carry_row, y, alpha, mode);
    ALFI_HIP_CHECK(ctx, hipGetLastError());
    return 0;
  }
  const double avg = A.nbrows > 0 ? (double)A.nnzb / (double)A.nbrows : 0.0;
  int lpr = 4;
  while (lpr < 64 && lpr < avg) lpr <<= 1;
  const int64_t threads = A.nbrows * lpr;
  dim3 grid((unsigned)((threads + 255) / 256)), block(256);
#define ALFI_SPMV_CASE(L)                                                                                       \
  case L:                                                                                                       \
    hipLaunchKernelGGL((bsr_spmv_kernel<BS, L>), grid, block, 0, ctx->stream, A.nbrows, A.rowptr, A.colidx,     \
                       A.vals, x, y, b, alpha, mode);                                                           \
    break;
  switch (lpr) {
    ALFI_SPMV_CASE(4)
    ALFI_SPMV_CASE(8)
    ALFI_SPMV_CASE(16)
    ALFI_SPMV_CASE(32)
    ALFI_SPMV_CASE(64)
  }
#undef ALFI_SPMV_CASE
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

int launch_bsr_spmv(alfi_ctx* ctx, const DevBSR& A, const double* x, double* y, const double* b, double alpha,
                    int mode) {
  if (A.view && A.nnzb == 0) return 0;   // an empty row range of a partitioned level: nothing to write
  if (A.nbrows == 0 || A.nnzb == 0) {
    // no entries: y = 0 or y = b on the rows of A
    if (A.nbrows > 0) {
      if (mode == 0) ALFI_HIP_CHECK(ctx, hipMemsetAsync(y, 0, sizeof(double) * A.nbrows * A.bs, ctx->stream));
      else ALFI_HIP_CHECK(ctx, hipMemcpyAsync(y, b, sizeof(double) * A.nbrows * A.bs, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return 0;
  }
  if (A.bs == 2) return launch_bsr_spmv_bs<2>(ctx, A, x, y, b, alpha, mode);
  if (A.bs == 3) return launch_bsr_spmv_bs<3>(ctx, A, x, y, b, alpha, mode);
  return alfi_set_error(ctx, ALFI_E_ARG, "unsupported block size %d", A.bs);
}

// ---------------------------------------------------------------------------------------------------------------------
// elementwise helpers
// ---------------------------------------------------------------------------------------------------------------------
static inline dim3 ew_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

__global__ void copy_kernel(double* __restrict__ y, const double* __restrict__ x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = x[i];
}
__global__ void axpy_kernel(double* __restrict__ y, const double* __restrict__ x, double a, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = __builtin_fma(a, x[i], y[i]);
}
__global__ void scale_by_inv_kernel(double* __restrict__ v, const double* __restrict__ w,
                                    const double* __restrict__ scal, int64_t n) {
  const double s = *scal;
  const double f = s != 0.0 ? 1.0 / s : 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    v[i] = w[i] * f;
}
__global__ void scatter_sub_kernel(double* __restrict__ x, const int32_t* __restrict__ idx,
                                   const double* __restrict__ t, double alpha, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[idx[i]] -= alpha * t[i];
}
__global__ void zero_dofs_kernel(double* __restrict__ x, const int32_t* __restrict__ idx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[idx[i]] = 0.0;
}
__global__ void copy_dofs_kernel(double* __restrict__ y, const double* __restrict__ x, const int32_t* __restrict__ idx,
                                 int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[idx[i]] = x[idx[i]];
}

#define ALFI_LAUNCH_EW(kern, n, ...)                                                       \
  do {                                                                                     \
    if ((n) > 0) {                                                                         \
      hipLaunchKernelGGL(kern, ew_grid(n), dim3(256), 0, ctx->stream, __VA_ARGS__);        \
      ALFI_HIP_CHECK(ctx, hipGetLastError());                                              \
    }                                                                                      \
  } while (0)

int launch_copy(alfi_ctx* ctx, double* y, const double* x, int64_t n) {
  ALFI_LAUNCH_EW(copy_kernel, n, y, x, n);
  return 0;
}
int launch_axpy(alfi_ctx* ctx, double* y, const double* x, double a, int64_t n) {
  ALFI_LAUNCH_EW(axpy_kernel, n, y, x, a, n);
  return 0;
}
int launch_scale_by_inv(alfi_ctx* ctx, double* v, const double* w, const double* scal, int64_t n) {
  ALFI_LAUNCH_EW(scale_by_inv_kernel, n, v, w, scal, n);
  return 0;
}
int launch_scatter_sub(alfi_ctx* ctx, double* x, const int32_t* idx, const double* t, double alpha, int64_t n) {
  ALFI_LAUNCH_EW(scatter_sub_kernel, n, x, idx, t, alpha, n);
  return 0;
}
int launch_zero_dofs(alfi_ctx* ctx, double* x, const int32_t* idx, int64_t n) {
  ALFI_LAUNCH_EW(zero_dofs_kernel, n, x, idx, n);
  return 0;
}
int launch_copy_dofs(alfi_ctx* ctx, double* y, const double* x, const int32_t* idx, int64_t n) {
  ALFI_LAUNCH_EW(copy_dofs_kernel, n, y, x, idx, n);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// reductions: two-stage (RED_BLOCKS partials, then one block sums them in a fixed order) -> deterministic
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int NV>
__device__ __forceinline__ void block_store_partials(double (&acc)[NV], double* __restrict__ partial) {
  __shared__ double red[4][NV];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const double s = wave_sum(acc[v]);
    if (lane == 0) red[wave][v] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV)
    partial[(int64_t)blockIdx.x * RED_MAXV + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// every thread of the block gets out[v] = sum_{b < nblocks <= RED_BLOCKS} partial[b][v], summed in one fixed order: all
// blocks of a launch (and the kernels that follow) see the identical value, and no separate reduction launch is needed
template <int NV>
__device__ __forceinline__ void block_reduce_partials(const double* __restrict__ partial, int nblocks, double (&out)[NV]) {
  __shared__ double red2[4][NV];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(int64_t)b * RED_MAXV + v];
    s = wave_sum(s);
    if (lane == 0) red2[wave][v] = s;
  }
  __syncthreads();
#pragma unroll
  for (int v = 0; v < NV; ++v) out[v] = (red2[0][v] + red2[1][v]) + (red2[2][v] + red2[3][v]);
}

// partial[b][v] = sum over the block's slice of V_v[i] * w[i]   (gridDim.x blocks: RED_BLOCKS, or fewer on small levels)
// VEC: every vector starts on a 16-byte boundary (even stride): a lane reads entry PAIRS, one 16-byte load per vector and
// iteration -- the scalar form (8 bytes per lane and load) ran at 1.7-2 TB/s on a 1.2 M-dof level with ~1 workgroup per CU
typedef double vec_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ vec_d2 ld2(const double* p, int64_t i2) { return reinterpret_cast<const vec_d2*>(p)[i2]; }
__device__ __forceinline__ void st2(double* p, int64_t i2, vec_d2 v) { reinterpret_cast<vec_d2*>(p)[i2] = v; }
__host__ inline bool vec2_ok(const void* a, const void* b, int64_t stride) {
  return (stride & 1) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}

template <int NV, bool VEC>
__global__ __launch_bounds__(256) void multi_dot_kernel(const double* __restrict__ V, int64_t stride,
                                                         const double* __restrict__ w, double* __restrict__ partial,
                                                         int64_t n) {
  double acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.0;
  if (VEC) {
    const int64_t n2 = n >> 1;
#pragma unroll 2
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
      const vec_d2 wi = ld2(w, i);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const vec_d2 a = ld2(V + v * stride, i);
        acc[v] = __builtin_fma(a.y, wi.y, __builtin_fma(a.x, wi.x, acc[v]));
      }
    }
    if ((n & 1) && blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {     // the odd last entry
      const double wi = w[n - 1];
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = __builtin_fma(V[v * stride + n - 1], wi, acc[v]);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      const double wi = w[i];
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = __builtin_fma(V[v * stride + i], wi, acc[v]);
    }
  }
  block_store_partials<NV>(acc, partial);
}

// out[v] = sum_b partial[b][v]   (one block, fixed order)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ partial, int nblocks, int nv,
                                                               double* __restrict__ out) {
  __shared__ double red[256];
  for (int v = 0; v < nv; ++v) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(int64_t)b * RED_MAXV + v];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[v] = red[0];
    __syncthreads();
  }
}

// w -= sum_v h[v] V_v ; partial[b][0] = sum of w^2 over the block's slice.  hblocks > 0: h is not given but reduced here
// from the hblocks (<= 256) dot partials hpart[b][v] of the preceding multi_dot_kernel; block 0 publishes it in h.
template <int NV, bool VEC>
__global__ __launch_bounds__(256) void multi_axpy_norm_kernel(const double* __restrict__ V, int64_t stride,
                                                               double* __restrict__ h, double* __restrict__ w,
                                                               double* __restrict__ partial, int64_t n,
                                                               const double* __restrict__ hpart, int hblocks) {
  double hv[NV];
  if (hblocks > 0) {
    block_reduce_partials<NV>(hpart, hblocks, hv);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
      for (int v = 0; v < NV; ++v) h[v] = hv[v];
    }
  } else {
#pragma unroll
    for (int v = 0; v < NV; ++v) hv[v] = h[v];
  }
  double acc[1] = {0.0};
  if (VEC) {
    const int64_t n2 = n >> 1;
#pragma unroll 2
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
      vec_d2 wi = ld2(w, i);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const vec_d2 a = ld2(V + v * stride, i);
        wi.x = __builtin_fma(-hv[v], a.x, wi.x);
        wi.y = __builtin_fma(-hv[v], a.y, wi.y);
      }
      st2(w, i, wi);
      acc[0] = __builtin_fma(wi.y, wi.y, __builtin_fma(wi.x, wi.x, acc[0]));
    }
    if ((n & 1) && blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {
      double wi = w[n - 1];
#pragma unroll
      for (int v = 0; v < NV; ++v) wi = __builtin_fma(-hv[v], V[v * stride + n - 1], wi);
      w[n - 1] = wi;
      acc[0] = __builtin_fma(wi, wi, acc[0]);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      double wi = w[i];
#pragma unroll
      for (int v = 0; v < NV; ++v) wi = __builtin_fma(-hv[v], V[v * stride + i], wi);
      w[i] = wi;
      acc[0] = __builtin_fma(wi, wi, acc[0]);
    }
  }
  block_store_partials<1>(acc, partial);
}

// x += sum_v y[v] Z_v
template <int NV, bool VEC>
__global__ __launch_bounds__(256) void multi_axpy_add_kernel(const double* __restrict__ Z, int64_t stride,
                                                              const double* __restrict__ y, double* __restrict__ x,
                                                              int64_t n) {
  double yv[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) yv[v] = y[v];
  if (VEC) {
    const int64_t n2 = n >> 1;
#pragma unroll 2
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
      vec_d2 xi = ld2(x, i);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const vec_d2 a = ld2(Z + v * stride, i);
        xi.x = __builtin_fma(yv[v], a.x, xi.x);
        xi.y = __builtin_fma(yv[v], a.y, xi.y);
      }
      st2(x, i, xi);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      double xi = x[n - 1];
#pragma unroll
      for (int v = 0; v < NV; ++v) xi = __builtin_fma(yv[v], Z[v * stride + n - 1], xi);
      x[n - 1] = xi;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      double xi = x[i];
#pragma unroll
      for (int v = 0; v < NV; ++v) xi = __builtin_fma(yv[v], Z[v * stride + i], xi);
      x[i] = xi;
    }
  }
}

// beta = sqrt(sum partial); hs[beta] = beta; grs[0] = beta
__global__ __launch_bounds__(256) void norm_init_finish_kernel(const double* __restrict__ partial, int nblocks,
                                                                double* __restrict__ hs, int K) {
  __shared__ double red[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(int64_t)b * RED_MAXV];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    HsLayout L(K);
    const double beta = sqrt(red[0]);
    hs[L.beta] = beta;
    hs[L.grs] = beta;
    for (int i = 1; i <= K; ++i) hs[L.grs + i] = 0.0;
  }
}

// (pythagoras_norm2, hessenberg_column, fgmres_back_substitution: hs_layout.h)

// ---------------------------------------------------------------------------------------------------------------------
// Fused smoother iteration for small levels (a smoother iteration there is a chain of launches of a few microseconds
// each: fewer, fatter launches).  Per FGMRES iteration j, with w = the unnormalised new Krylov direction:
//     patch apply on w (existing kernels; the smoother is linear, so the normalisation can follow the solves),
//     patch_sum_scale_kernel :  f = 1 / |w| from the norm partials;  z_j = f * (sum of the staged patch results),
//                               v_j = f * w;  one thread finishes column j-1 of the Hessenberg (or starts the rotated rhs),
//     bsr_spmv_dot_kernel    :  w = A z_j and the partials of V_i . w, i <= j, in the same pass over the rows,
//     multi_axpy_norm_kernel :  h from the partials, w -= V h, norm partials
// -- four launches instead of eight.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patch_sum_scale_kernel(int64_t n, const int32_t* __restrict__ dof_ptr,
                                                               const int32_t* __restrict__ dof_pos,
                                                               const double* __restrict__ stage,
                                                               const uint8_t* __restrict__ bc_mask,
                                                               const double* __restrict__ w, double* __restrict__ z,
                                                               double* __restrict__ v,
                                                               const double* __restrict__ normpart, int nblocks,
                                                               const double* __restrict__ h, double* __restrict__ hs, int j,
                                                               int K, int pou) {
  double s2[1];
  block_reduce_partials<1>(normpart, nblocks, s2);
  const double tt = sqrt(s2[0]);
  const double f = tt != 0.0 ? 1.0 / tt : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (j == 0) {                       // beta = |r0|, rotated rhs = beta e_1
      HsLayout L(K);
      hs[L.beta] = tt;
      hs[L.grs] = tt;
      for (int i = 1; i <= K; ++i) hs[L.grs + i] = 0.0;
    } else {
      hessenberg_column(hs, K, j - 1, h, tt);
    }
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double s = 0.0;
    const int32_t q0 = dof_ptr[i], q1 = dof_ptr[i + 1];
    for (int32_t q = q0; q < q1; ++q) s += stage[dof_pos[q]];
    if (pou && q1 - q0 > 1) s /= (double)(q1 - q0);
    const double wi = w[i];
    z[i] = f * (bc_mask[i] ? wi : s);   // Dirichlet dofs: the smoother copies its argument there
    v[i] = f * wi;
  }
}

// w = A z on the block rows [0, nbrows) of a flat-layout BSR matrix, LPR lanes per block row, and in the same pass the
// partials of V_v . w for v < NV (gridDim.x <= RED_BLOCKS blocks, grid-stride over the rows)
template <int BS, int LPR, int NV>
__global__ __launch_bounds__(256) void bsr_spmv_dot_kernel(int64_t nbrows, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ colflag,
                                                            const double* __restrict__ vals,
                                                            const double* __restrict__ z, double* __restrict__ w,
                                                            const double* __restrict__ V, int64_t stride,
                                                            double* __restrict__ partial) {
  constexpr int BB = BS * BS;
  const int l = threadIdx.x % LPR;
  double acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.0;
  for (int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPR; row < nbrows;
       row += (int64_t)gridDim.x * (256 / LPR)) {
    double s[BS];
#pragma unroll
    for (int r = 0; r < BS; ++r) s[r] = 0.0;
    const int32_t lo = rowptr[row], hi = rowptr[row + 1];
    for (int32_t k = lo + l; k < hi; k += LPR) {
      const int64_t col = colflag[k] & 0x7fffffff;
      const double* vb = vals + ((int64_t)k >> 6) * (64 * BB);
      const int kl = k & 63;
      double a[BB], xv[BS];
#pragma unroll
      for (int q = 0; q < BB / 2; ++q) {
        const spmv_d2 t = *(reinterpret_cast<const spmv_d2*>(vb + q * 128) + kl);
        a[2 * q] = t.x;
        a[2 * q + 1] = t.y;
      }
      if (BB & 1) a[BB - 1] = vb[(BB / 2) * 128 + kl];
#pragma unroll
      for (int c = 0; c < BS; ++c) xv[c] = z[col * BS + c];
#pragma unroll
      for (int r = 0; r < BS; ++r)
#pragma unroll
        for (int c = 0; c < BS; ++c) s[r] = __builtin_fma(a[r * BS + c], xv[c], s[r]);
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < BS; ++r) s[r] += __shfl_xor(s[r], o);
    if (l == 0) {
#pragma unroll
      for (int r = 0; r < BS; ++r) {
        const int64_t i = row * BS + r;
        w[i] = s[r];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = __builtin_fma(V[v * stride + i], s[r], acc[v]);
      }
    }
  }
  block_store_partials<NV>(acc, partial);
}

__global__ __launch_bounds__(256) void hessenberg_update_kernel(const double* __restrict__ partial, int nblocks,
                                                                 const double* __restrict__ h,
                                                                 double* __restrict__ hs, int j, int K,
                                                                 const double* __restrict__ ww) {
  __shared__ double red[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(int64_t)b * RED_MAXV];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) hessenberg_column(hs, K, j, h, sqrt(ww ? pythagoras_norm2(ww, h, j) : red[0]));
}

// back substitution on the triangularised Hessenberg (KSPFGMRESBuildSoln [3P]) -> y
__global__ void fgmres_finish_kernel(double* __restrict__ hs, int k, int K) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  fgmres_back_substitution(hs, k, K);
}

#define ALFI_NV_SWITCH(NVAL, MACRO) \
  switch (NVAL) {                   \
    MACRO(1) MACRO(2) MACRO(3) MACRO(4) MACRO(5) MACRO(6) MACRO(7) MACRO(8) MACRO(9) MACRO(10) MACRO(11) MACRO(12) \
    MACRO(13) MACRO(14) MACRO(15) MACRO(16)                                                                         \
  }

// blocks of the two-stage reductions over a vector of n entries: RED_BLOCKS for the large levels, one per 4096 entries on
// the small ones (there every kernel of a smoother iteration is a few microseconds of latency, and 1024 blocks with
// 1024 x nv partials to sum would dominate it)
int red_blocks_for(int64_t n) {
  const int64_t g = (n + 4095) / 4096;
  return (int)(g < 1 ? 1 : (g > RED_BLOCKS ? RED_BLOCKS : g));
}

int launch_multi_dot(alfi_ctx* ctx, const double* V, int64_t stride, int nv, const double* w, double* out, int64_t n) {
  // out[0..nv) = V_v . w ; more than 16 vectors: passes of 16
  const int G = red_blocks_for(n);
  for (int v0 = 0; v0 < nv; v0 += 16) {
    const int cnt = nv - v0 < 16 ? nv - v0 : 16;
    const bool vec = vec2_ok(V + (int64_t)v0 * stride, w, stride);
#define ALFI_CASE(N)                                                                                              \
  case N:                                                                                                         \
    if (vec)                                                                                                      \
      hipLaunchKernelGGL((multi_dot_kernel<N, true>), dim3(G), dim3(256), 0, ctx->stream, V + (int64_t)v0 * stride, \
                         stride, w, ctx->red_partial, n);                                                         \
    else                                                                                                          \
      hipLaunchKernelGGL((multi_dot_kernel<N, false>), dim3(G), dim3(256), 0, ctx->stream, V + (int64_t)v0 * stride, \
                         stride, w, ctx->red_partial, n);                                                         \
    break;
    ALFI_NV_SWITCH(cnt, ALFI_CASE)
#undef ALFI_CASE
    ALFI_HIP_CHECK(ctx, hipGetLastError());
    if (out) {
      hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->red_partial, G, cnt, out + v0);
      ALFI_HIP_CHECK(ctx, hipGetLastError());
    }
  }
  return 0;
}

// w -= sum h_v V_v (passes of 16; the last pass also produces |w|^2 partials, red_blocks_for(n) of them, in norm_partial).
// hblocks > 0 (nv <= 16, hblocks <= 256): h is reduced inside the kernel from the dot partials in ctx->red_partial.
int launch_multi_axpy_norm(alfi_ctx* ctx, const double* V, int64_t stride, int nv, double* h, double* w, int64_t n,
                           double* norm_partial, int hblocks) {
  const int G = red_blocks_for(n);
  for (int v0 = 0; v0 < nv; v0 += 16) {
    const int cnt = nv - v0 < 16 ? nv - v0 : 16;
    const bool vec = vec2_ok(V + (int64_t)v0 * stride, w, stride);
#define ALFI_CASE(N)                                                                                               \
  case N:                                                                                                          \
    if (vec)                                                                                                       \
      hipLaunchKernelGGL((multi_axpy_norm_kernel<N, true>), dim3(G), dim3(256), 0, ctx->stream,                    \
                         V + (int64_t)v0 * stride, stride, h + v0, w, norm_partial, n, ctx->red_partial, hblocks); \
    else                                                                                                           \
      hipLaunchKernelGGL((multi_axpy_norm_kernel<N, false>), dim3(G), dim3(256), 0, ctx->stream,                   \
                         V + (int64_t)v0 * stride, stride, h + v0, w, norm_partial, n, ctx->red_partial, hblocks); \
    break;
    ALFI_NV_SWITCH(cnt, ALFI_CASE)
#undef ALFI_CASE
    ALFI_HIP_CHECK(ctx, hipGetLastError());
  }
  return 0;
}

// the same, fused with v_next = w / tt (one launch less per smoother iteration; matters on the small levels, which are
// launch-bound).  Every block reduces the partials itself in the same fixed order, so all use the identical tt; block 0
// also performs the Hessenberg / Givens update.  Nobody reads what block 0 writes before the kernel ends.
__global__ __launch_bounds__(256) void hessenberg_scale_kernel(const double* __restrict__ partial, int nblocks,
                                                                const double* __restrict__ h, double* __restrict__ hs,
                                                                int j, int K, double* __restrict__ vnext,
                                                                const double* __restrict__ w, int64_t n,
                                                                const double* __restrict__ ww) {
  __shared__ double red[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(int64_t)b * RED_MAXV];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const double tt = sqrt(ww ? pythagoras_norm2(ww, h, j) : red[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0) hessenberg_column(hs, K, j, h, tt);
  const double f = tt != 0.0 ? 1.0 / tt : 0.0;
  if (((reinterpret_cast<uintptr_t>(vnext) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {     // 16 bytes per lane
    const int64_t n2 = n >> 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
      vec_d2 t = ld2(w, i);
      t.x *= f;
      t.y *= f;
      st2(vnext, i, t);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) vnext[n - 1] = w[n - 1] * f;
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) vnext[i] = w[i] * f;
}

int launch_hessenberg_scale(alfi_ctx* ctx, const double* partial, int nblocks, const double* h, double* hs, int j, int K,
                            double* vnext, const double* w, int64_t n, const double* ww) {
  hipLaunchKernelGGL(hessenberg_scale_kernel, ew_grid(n), dim3(256), 0, ctx->stream, partial, nblocks, h, hs, j, K, vnext,
                     w, n, ww);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

int launch_hessenberg_update(alfi_ctx* ctx, const double* partial, int nblocks, const double* h, double* hs, int j,
                             int K, const double* ww) {
  hipLaunchKernelGGL(hessenberg_update_kernel, dim3(1), dim3(256), 0, ctx->stream, partial, nblocks, h, hs, j, K, ww);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

int launch_norm_partials(alfi_ctx* ctx, const double* r, int64_t n) {
  // |r|^2 partials (red_blocks_for(n) of them) via the dot kernel with V = w = r
  if (vec2_ok(r, r, 0))
    hipLaunchKernelGGL((multi_dot_kernel<1, true>), dim3(red_blocks_for(n)), dim3(256), 0, ctx->stream, r, (int64_t)0, r,
                       ctx->red_partial, n);
  else
    hipLaunchKernelGGL((multi_dot_kernel<1, false>), dim3(red_blocks_for(n)), dim3(256), 0, ctx->stream, r, (int64_t)0, r,
                       ctx->red_partial, n);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

int launch_norm_init_finish(alfi_ctx* ctx, const double* partial, int nblocks, double* hs, int K) {
  hipLaunchKernelGGL(norm_init_finish_kernel, dim3(1), dim3(256), 0, ctx->stream, partial, nblocks, hs, K);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

int launch_reduce_partials(alfi_ctx* ctx, const double* partial, int nblocks, int nv, double* out) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, partial, nblocks, nv, out);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// halo exchange helpers (partitioned levels): pack owned nodes into the send buffer; add received ghost contributions
// onto their owners node by node in a fixed order (deterministic)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void halo_pack_kernel(double* __restrict__ buf, const double* __restrict__ v,
                                 const int32_t* __restrict__ nodes, int64_t total, int bs) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / bs;
    const int c = (int)(e - i * bs);
    buf[e] = v[(int64_t)nodes[i] * bs + c];
  }
}
__global__ void halo_add_kernel(double* __restrict__ v, const double* __restrict__ buf,
                                const int32_t* __restrict__ rev_nodes, const int32_t* __restrict__ rev_ptr,
                                const int32_t* __restrict__ rev_pos, int64_t total, int bs) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = e / bs;
    const int c = (int)(e - u * bs);
    const int64_t at = (int64_t)rev_nodes[u] * bs + c;
    double s = v[at];
    for (int32_t q = rev_ptr[u]; q < rev_ptr[u + 1]; ++q) s += buf[(int64_t)rev_pos[q] * bs + c];
    v[at] = s;
  }
}
__global__ void halo_sum_kernel(double* __restrict__ v, const double* __restrict__ buf, const int32_t* __restrict__ nodes,
                                const int32_t* __restrict__ ptr, const int32_t* __restrict__ src, int64_t total, int bs) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = e / bs;
    const int c = (int)(e - u * bs);
    const int64_t at = (int64_t)nodes[u] * bs + c;
    const double own = v[at];
    double s = 0.0;
    for (int32_t q = ptr[u]; q < ptr[u + 1]; ++q) s += src[q] < 0 ? own : buf[(int64_t)src[q] * bs + c];
    v[at] = s;
  }
}
int launch_halo_sum(alfi_ctx* ctx, double* v, const double* buf, const int32_t* nodes, const int32_t* ptr, const int32_t* src,
                    int64_t nshared, int bs) {
  ALFI_LAUNCH_EW(halo_sum_kernel, nshared * bs, v, buf, nodes, ptr, src, nshared * bs, bs);
  return 0;
}
int launch_halo_pack(alfi_ctx* ctx, double* buf, const double* v, const int32_t* nodes, int64_t nnodes, int bs) {
  ALFI_LAUNCH_EW(halo_pack_kernel, nnodes * bs, buf, v, nodes, nnodes * bs, bs);
  return 0;
}
int launch_halo_add(alfi_ctx* ctx, double* v, const double* buf, const int32_t* rev_nodes, const int32_t* rev_ptr,
                    const int32_t* rev_pos, int64_t nuniq, int bs) {
  ALFI_LAUNCH_EW(halo_add_kernel, nuniq * bs, v, buf, rev_nodes, rev_ptr, rev_pos, nuniq * bs, bs);
  return 0;
}

int launch_patch_sum_scale(alfi_level* L, const double* w, double* z, double* v, const double* normpart, int nblocks,
                           const double* h, double* hs, int j, int K) {
  alfi_ctx* ctx = L->ctx;
  const int64_t n = L->n;
  hipLaunchKernelGGL(patch_sum_scale_kernel, ew_grid(n), dim3(256), 0, ctx->stream, n, L->dof_ptr, L->dof_pos, L->stage,
                     L->bc_mask, w, z, v, normpart, nblocks, h, hs, j, K, L->pou ? 1 : 0);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

template <int BS, int LPR>
static int launch_spmv_dot_lpr(alfi_ctx* ctx, const DevBSR& A, const double* z, double* w, const double* V, int64_t stride,
                               int nv, double* partial, int* nblocks) {
  const int64_t rows_per_block = 256 / LPR;
  int64_t g = (A.nbrows + rows_per_block - 1) / rows_per_block;
  if (g > RED_BLOCKS) g = RED_BLOCKS;      // one partial per block: at most RED_BLOCKS of them
  if (g < 1) g = 1;
  *nblocks = (int)g;
#define ALFI_CASE(N)                                                                                                 \
  case N:                                                                                                            \
    hipLaunchKernelGGL((bsr_spmv_dot_kernel<BS, LPR, N>), dim3((unsigned)g), dim3(256), 0, ctx->stream, A.nbrows,      \
                       A.rowptr, A.colidx, A.vals, z, w, V, stride, partial);                                        \
    break;
  ALFI_NV_SWITCH(nv, ALFI_CASE)
#undef ALFI_CASE
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

// w = A z on the rows of a flat-layout matrix with the dot partials of nv <= 16 vectors; *nblocks = partials written
int launch_bsr_spmv_dot(alfi_ctx* ctx, const DevBSR& A, const double* z, double* w, const double* V, int64_t stride, int nv,
                        double* partial, int* nblocks) {
  if (!A.flat || nv < 1 || nv > 16) return alfi_set_error(ctx, ALFI_E_STATE, "fused SpMV needs the flat layout and <= 16 vectors");
  const double avg = A.nbrows > 0 ? (double)A.nnzb / (double)A.nbrows : 0.0;
  if (A.bs == 2) {
    if (avg <= 12) return launch_spmv_dot_lpr<2, 4>(ctx, A, z, w, V, stride, nv, partial, nblocks);
    if (avg <= 40) return launch_spmv_dot_lpr<2, 8>(ctx, A, z, w, V, stride, nv, partial, nblocks);
    return launch_spmv_dot_lpr<2, 32>(ctx, A, z, w, V, stride, nv, partial, nblocks);
  }
  if (A.bs == 3) {
    if (avg <= 24) return launch_spmv_dot_lpr<3, 8>(ctx, A, z, w, V, stride, nv, partial, nblocks);
    if (avg <= 48) return launch_spmv_dot_lpr<3, 16>(ctx, A, z, w, V, stride, nv, partial, nblocks);
    return launch_spmv_dot_lpr<3, 32>(ctx, A, z, w, V, stride, nv, partial, nblocks);
  }
  return alfi_set_error(ctx, ALFI_E_ARG, "unsupported block size %d", A.bs);
}

// column `col` of the Hessenberg from (h, partials of |w|^2), then the back substitution (last step of the fused smoother)
__global__ __launch_bounds__(256) void fgmres_finish_fused_kernel(const double* __restrict__ normpart, int nblocks,
                                                                  const double* __restrict__ h, double* __restrict__ hs,
                                                                  int k, int K) {
  double s2[1];
  block_reduce_partials<1>(normpart, nblocks, s2);
  if (threadIdx.x != 0) return;
  hessenberg_column(hs, K, k - 1, h, sqrt(s2[0]));
  HsLayout L(K);
  double* y = hs + L.y;
  const double* grs = hs + L.grs;
  for (int i = k - 1; i >= 0; --i) {
    double s = grs[i];
    for (int q = i + 1; q < k; ++q) s -= hs[L.H(q) + i] * y[q];
    const double d = hs[L.H(i) + i];
    y[i] = d != 0.0 ? s / d : 0.0;
  }
}
int launch_fgmres_finish_fused(alfi_ctx* ctx, const double* normpart, int nblocks, const double* h, double* hs, int k, int K) {
  hipLaunchKernelGGL(fgmres_finish_fused_kernel, dim3(1), dim3(256), 0, ctx->stream, normpart, nblocks, h, hs, k, K);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

int launch_fgmres_finish(alfi_ctx* ctx, double* hs, int k, int K) {
  hipLaunchKernelGGL(fgmres_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, hs, k, K);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

int launch_update_solution(alfi_ctx* ctx, double* x, const double* Z, int64_t stride, int k, const double* y,
                           int64_t n) {
  for (int v0 = 0; v0 < k; v0 += 16) {
    const int cnt = k - v0 < 16 ? k - v0 : 16;
    const bool vec = vec2_ok(Z + (int64_t)v0 * stride, x, stride);
    const dim3 grid = ew_grid(vec ? (n + 1) / 2 : n);
#define ALFI_CASE(N)                                                                                                \
  case N:                                                                                                           \
    if (vec)                                                                                                        \
      hipLaunchKernelGGL((multi_axpy_add_kernel<N, true>), grid, dim3(256), 0, ctx->stream, Z + (int64_t)v0 * stride, \
                         stride, y + v0, x, n);                                                                     \
    else                                                                                                            \
      hipLaunchKernelGGL((multi_axpy_add_kernel<N, false>), grid, dim3(256), 0, ctx->stream, Z + (int64_t)v0 * stride, \
                         stride, y + v0, x, n);                                                                     \
    break;
    ALFI_NV_SWITCH(cnt, ALFI_CASE)
#undef ALFI_CASE
    ALFI_HIP_CHECK(ctx, hipGetLastError());
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Schoeberl transfer: interior blocks.  binv holds nblk inverses, column-major, leading dimension ld (m rounded up to
// even), stride m_cols * ld per block -- the same convention as the patch inverses.
// ---------------------------------------------------------------------------------------------------------------------
// S = nu K_II + gamma D_II in row-major padded form (input of the inversion kernels); block b at S + b * stride
__global__ void block_build_kernel(int64_t nblk, int m, int ld, int64_t stride, const double* __restrict__ K,
                                   const double* __restrict__ D, double nu, double gamma, double* __restrict__ S) {
  const int64_t total = nblk * m * ld;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t blk = e / (m * ld);
    const int64_t in = e - blk * m * ld;
    const int r = (int)(in / ld), c = (int)(in % ld);
    S[blk * stride + in] = c < m ? nu * K[(blk * m + r) * m + c] + gamma * D[(blk * m + r) * m + c] : 0.0;
  }
}

// A_II = nu K_II + gamma D_II, column-major with leading dimension ld (even), built once per alfi_transfer_update: the operator of
// the refinement residual (block_residual_kernel below)
__global__ void block_operator_t_kernel(int64_t nblk, int m, int ld, const double* __restrict__ K, const double* __restrict__ D,
                                        double nu, double gamma, double* __restrict__ At) {
  const int64_t total = nblk * m * ld;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t blk = e / ((int64_t)m * ld);
    const int64_t in = e - blk * (int64_t)m * ld;
    const int j = (int)(in / ld), i = (int)(in % ld);
    At[e] = i < m ? nu * K[(blk * m + i) * m + j] + gamma * D[(blk * m + i) * m + j] : 0.0;
  }
}

int launch_block_build_invert(alfi_transfer* tr) {
  alfi_ctx* ctx = tr->ctx;
  const int64_t total = tr->nblk * tr->m * tr->ld;
  if (tr->patch_mode && tr->AIIt)   // the operator of the refinement step's residual
    ALFI_LAUNCH_EW(block_operator_t_kernel, total, tr->nblk, tr->m, tr->ld, tr->KII, tr->DII, tr->nu, tr->gamma, tr->AIIt);
  if (tr->m > SMALL_PATCH_MAX) return launch_big_factor_transfer(tr);   // blocked MFMA inversion (kernels_bigpatch.hip)
  const int64_t stride = tr->patch_mode ? tr->bstride : (int64_t)tr->m * tr->ld;
  ALFI_LAUNCH_EW(block_build_kernel, total, tr->nblk, tr->m, tr->ld, stride, tr->KII, tr->DII, tr->nu, tr->gamma,
                 tr->binv);
  if (tr->patch_mode)   // m > 32: register Gauss-Jordan of the patch smoother, result in the row-piece layout
    return launch_patch_invert_arrays(ctx, tr->nblk, tr->m, tr->pm_ptr, tr->pm_inv_ptr, tr->binv, tr->status);
  return launch_invert_small_any(ctx, tr->m, tr->nblk, nullptr, nullptr, tr->m, (int64_t)tr->m * tr->ld, tr->binv,
                                 tr->status);
}

// out[blk*m + i] = sum_j binv_blk[i][j] * in_j ; in_j = in[blk_dofs[blk*m + j]] (gather) or in[blk*m + j].
// G lanes per block (G >= m, power of two), lane i owns row i, x_j broadcast by shuffle.
template <int G>
__global__ __launch_bounds__(256) void block_gemv_kernel(int64_t nblk, int m, int ld, const double* __restrict__ binv,
                                                          const int32_t* __restrict__ blk_dofs,
                                                          const double* __restrict__ in, double* __restrict__ out,
                                                          int gather) {
  const int64_t blk = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
  const int i = threadIdx.x % G;
  const bool valid = blk < nblk && i < m;
  double xi = 0.0;
  if (valid) xi = gather ? in[blk_dofs[blk * m + i]] : in[blk * m + i];
  const double* T = binv + (valid ? blk * (int64_t)m * ld : 0);
  double acc = 0.0;
  for (int j = 0; j < m; ++j) {
    const double xj = __shfl(xi, j, G);
    if (valid) acc = __builtin_fma(T[(int64_t)j * ld + i], xj, acc);
  }
  if (valid) out[blk * m + i] = acc;
}

__global__ void compact_rows_kernel(double* __restrict__ out, const double* __restrict__ in, int64_t total, int m,
                                    int ld) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = in[(e / m) * ld + e % m];
}

// One step of iterative refinement around the explicit inverse of an interior block: the reference solves these blocks by LU
// (patch_sub_pc_type lu, transfer.py:100-113), and nu K + gamma D of a macro cell has condition number ~gamma / nu, which an
// explicit inverse alone turns into a relative error of cond * eps in the transferred vector.
//
// r = b - A_II t per interior block (A_II column-major: block_operator_t_kernel): one workgroup per block, t in LDS, a lane per PAIR of rows (one 16-byte load per
// column, the lanes of a wave along the column: whole lines), BR_U columns in flight.  (Until round 5: K_II and D_II read
// row-major with a wave per row and a shuffle reduction per row -- 2.6 TB/s for twice the bytes; 63 % of config 5's PROLONG.)
constexpr int BR_U = 8;
__global__ __launch_bounds__(256) void block_residual_kernel(int m, int ld, const double* __restrict__ At,
                                                              const int32_t* __restrict__ idx, const double* __restrict__ in,
                                                              const double* __restrict__ t, double* __restrict__ r) {
  __shared__ double ts[PATCH_MAX];
  const int64_t blk = blockIdx.x;
  for (int i = threadIdx.x; i < m; i += 256) ts[i] = t[blk * m + i];
  __syncthreads();
  const double* A = At + blk * (int64_t)m * ld;
  for (int p = threadIdx.x; 2 * p < m; p += 256) {
    const double* col = A + 2 * p;
    double a0 = 0.0, a1 = 0.0;
    int j = 0;
    for (; j + BR_U <= m; j += BR_U) {
      vec_d2 a[BR_U];
#pragma unroll
      for (int u = 0; u < BR_U; ++u) a[u] = __builtin_nontemporal_load(reinterpret_cast<const vec_d2*>(col + (int64_t)(j + u) * ld));
#pragma unroll
      for (int u = 0; u < BR_U; ++u) {
        a0 = __builtin_fma(a[u].x, ts[j + u], a0);
        a1 = __builtin_fma(a[u].y, ts[j + u], a1);
      }
    }
    for (; j < m; ++j) {
      const vec_d2 a = __builtin_nontemporal_load(reinterpret_cast<const vec_d2*>(col + (int64_t)j * ld));
      a0 = __builtin_fma(a.x, ts[j], a0);
      a1 = __builtin_fma(a.y, ts[j], a1);
    }
    const int i0 = 2 * p;
    r[blk * m + i0] = in[idx[blk * m + i0]] - a0;
    if (i0 + 1 < m) r[blk * m + i0 + 1] = in[idx[blk * m + i0 + 1]] - a1;
  }
}

// out (compact, stride m) = X in[idx] with the row-piece inverses of a patch-mode transfer
static int pm_apply(alfi_transfer* tr, const int32_t* idx, const double* in, double* out) {
  alfi_ctx* ctx = tr->ctx;
  double* dst = tr->pm_tmp ? tr->pm_tmp : out;
  if (tr->m > SMALL_PATCH_MAX)
    ALFI_CHECK(launch_big_apply_arrays(ctx, tr->nblk, tr->m, tr->pm_ptr, idx, tr->pm_inv_ptr, tr->pm_stage_ptr, tr->binv, in, dst));
  else
    ALFI_CHECK(launch_patch_apply_arrays(ctx, tr->nblk, tr->pm_ptr, idx, tr->pm_inv_ptr, tr->pm_stage_ptr, tr->binv, in,
                                         dst));
  if (tr->pm_tmp) {   // odd m: rows were written with stride ld = m + 1
    const int64_t total = tr->nblk * tr->m;
    ALFI_LAUNCH_EW(compact_rows_kernel, total, out, tr->pm_tmp, total, tr->m, tr->ld);
  }
  return 0;
}

int launch_block_gemv(alfi_transfer* tr, const double* in, double* out, bool gather_in) {
  alfi_ctx* ctx = tr->ctx;
  if (tr->nblk == 0) return 0;
  if (tr->patch_mode) {
    const int32_t* idx = gather_in ? tr->blk_dofs : tr->pm_iota;
    ALFI_CHECK(pm_apply(tr, idx, in, out));
    if (tr->pm_res) {   // t += X (b - A t)
      hipLaunchKernelGGL(block_residual_kernel, dim3((unsigned)tr->nblk), dim3(256), 0, ctx->stream, tr->m, tr->ld, tr->AIIt,
                         idx, in, out, tr->pm_res);
      ALFI_HIP_CHECK(ctx, hipGetLastError());
      ALFI_CHECK(pm_apply(tr, tr->pm_iota, tr->pm_res, tr->pm_cor));
      ALFI_CHECK(launch_axpy(ctx, out, tr->pm_cor, 1.0, tr->nblk * tr->m));
    }
    return 0;
  }
  const int m = tr->m;
  const int G = m <= 8 ? 8 : (m <= 16 ? 16 : 32);
  const int64_t threads = tr->nblk * G;
  dim3 grid((unsigned)((threads + 255) / 256)), block(256);
  if (G == 8)
    hipLaunchKernelGGL(block_gemv_kernel<8>, grid, block, 0, ctx->stream, tr->nblk, m, tr->ld, tr->binv, tr->blk_dofs,
                       in, out, gather_in ? 1 : 0);
  else if (G == 16)
    hipLaunchKernelGGL(block_gemv_kernel<16>, grid, block, 0, ctx->stream, tr->nblk, m, tr->ld, tr->binv, tr->blk_dofs,
                       in, out, gather_in ? 1 : 0);
  else
    hipLaunchKernelGGL(block_gemv_kernel<32>, grid, block, 0, ctx->stream, tr->nblk, m, tr->ld, tr->binv, tr->blk_dofs,
                       in, out, gather_in ? 1 : 0);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// coarse solve: y = Ainv x, Ainv dense row-major n x n; one workgroup per row
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_gemv_kernel(const double* __restrict__ A, const double* __restrict__ x,
                                                          double* __restrict__ y, int64_t n) {
  __shared__ double red[4];
  const int64_t row = blockIdx.x;
  const double* a = A + row * n;
  double acc = 0.0;
  for (int64_t j = threadIdx.x; j < n; j += 256) acc = __builtin_fma(a[j], x[j], acc);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) y[row] = red[0] + red[1] + red[2] + red[3];
}

int launch_dense_gemv(alfi_ctx* ctx, const double* A, const double* x, double* y, int64_t n) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(dense_gemv_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, A, x, y, n);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// outer saddle-point solve: scalar CSR products with the discrete divergence B / its transpose, diagonal mass scaling,
// removal of the constant pressure mode.  Off the roofline path (B holds 3 % of the operator's entries).
// ---------------------------------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void csr_spmv_kernel(int64_t nrows, const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ colidx,
                                                        const double* __restrict__ vals, const double* __restrict__ x,
                                                        double* __restrict__ y, const double* __restrict__ b,
                                                        double alpha, int mode) {
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPR;
  const int l = threadIdx.x % LPR;
  double acc = 0.0;
  if (row < nrows)
    for (int32_t k = rowptr[row] + l; k < rowptr[row + 1]; k += LPR) acc = __builtin_fma(vals[k], x[colidx[k]], acc);
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (row < nrows && l == 0)
    y[row] = mode == 0 ? acc : (mode == 1 ? b[row] - alpha * acc : (mode == 3 ? alpha * acc : y[row] + acc));
}

int launch_csr_spmv(alfi_ctx* ctx, const DevCSR& A, const double* x, double* y, const double* b, double alpha,
                    int mode) {
  if (A.nrows == 0) return 0;
  const double avg = (double)A.nnz / (double)A.nrows;
  dim3 block(256);
  if (avg > 24) {
    hipLaunchKernelGGL(csr_spmv_kernel<32>, dim3((unsigned)((A.nrows * 32 + 255) / 256)), block, 0, ctx->stream, A.nrows,
                       A.rowptr, A.colidx, A.vals, x, y, b, alpha, mode);
  } else if (avg > 6) {
    hipLaunchKernelGGL(csr_spmv_kernel<8>, dim3((unsigned)((A.nrows * 8 + 255) / 256)), block, 0, ctx->stream, A.nrows,
                       A.rowptr, A.colidx, A.vals, x, y, b, alpha, mode);
  } else {
    hipLaunchKernelGGL(csr_spmv_kernel<2>, dim3((unsigned)((A.nrows * 2 + 255) / 256)), block, 0, ctx->stream, A.nrows,
                       A.rowptr, A.colidx, A.vals, x, y, b, alpha, mode);
  }
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

__global__ void scale_rows_kernel(double* __restrict__ y, const double* __restrict__ x, const double* __restrict__ d,
                                  double a, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = a * d[i] * x[i];
}
int launch_scale_rows(alfi_ctx* ctx, double* y, const double* x, const double* d, double a, int64_t n) {
  ALFI_LAUNCH_EW(scale_rows_kernel, n, y, x, d, a, n);
  return 0;
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ x, double* __restrict__ partial,
                                                            int64_t n) {
  double acc[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)RED_BLOCKS * 256) acc[0] += x[i];
  block_store_partials<1>(acc, partial);
}
__global__ __launch_bounds__(256) void subtract_mean_kernel(double* __restrict__ x, const double* __restrict__ partial,
                                                             int64_t n) {
  __shared__ double red[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < RED_BLOCKS; b += 256) s += partial[(int64_t)b * RED_MAXV];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const double mean = red[0] / (double)n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] -= mean;
}
int launch_remove_mean(alfi_ctx* ctx, double* x, int64_t n) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(RED_BLOCKS), dim3(256), 0, ctx->stream, x, ctx->red_partial, n);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  hipLaunchKernelGGL(subtract_mean_kernel, ew_grid(n), dim3(256), 0, ctx->stream, x, ctx->red_partial, n);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

__global__ void xmy_kernel(double* __restrict__ w, const double* __restrict__ b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    w[i] = b[i] - w[i];
}
int launch_xmy(alfi_ctx* ctx, double* w, const double* b, int64_t n) {
  ALFI_LAUNCH_EW(xmy_kernel, n, w, b, n);
  return 0;
}

// ---- pieces of the outer solve on partitioned levels (alfi_saddle_*, api_saddle.hip) ------------------------------------------
// *out = sum(x) in one fixed order (the rank's part of a global sum; all-reduced by the caller).  n == 0: *out = 0.
int launch_sum_to(alfi_ctx* ctx, const double* x, int64_t n, double* out) {
  hipLaunchKernelGGL(sum_partials_kernel, dim3(RED_BLOCKS), dim3(256), 0, ctx->stream, x, ctx->red_partial, n);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return launch_reduce_partials(ctx, ctx->red_partial, RED_BLOCKS, 1, out);
}
__global__ void sub_scaled_kernel(double* __restrict__ x, const double* __restrict__ s, double f, int64_t n) {
  const double d = f * *s;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] -= d;
}
int launch_sub_scaled(alfi_ctx* ctx, double* x, int64_t n, const double* s, double f) {
  ALFI_LAUNCH_EW(sub_scaled_kernel, n, x, s, f, n);
  return 0;
}
__global__ void add_kernel(double* __restrict__ y, const double* __restrict__ a, const double* __restrict__ b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = a[i] + b[i];
}
int launch_add(alfi_ctx* ctx, double* y, const double* a, const double* b, int64_t n) {
  ALFI_LAUNCH_EW(add_kernel, n, y, a, b, n);
  return 0;
}

// inject on a non-nested (barycentric) hierarchy: coarse node i = sum_k w_k fine node c_k (the fine function evaluated at the
// coarse node: firedrake.inject [3P], alfi/solver.py:645-652), every component alike; one thread per coarse dof, fixed order
__global__ void inject_csr_kernel(int64_t total, int bs, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                  const double* __restrict__ vals, const double* __restrict__ xf, double* __restrict__ xc) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / bs;
    const int c = (int)(e - i * bs);
    double s = 0.0;
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s = __builtin_fma(vals[k], xf[(int64_t)colidx[k] * bs + c], s);
    xc[e] = s;
  }
}
int launch_inject_csr(alfi_ctx* ctx, const DevCSR& J, int bs, const double* xf, double* xc) {
  ALFI_LAUNCH_EW(inject_csr_kernel, J.nrows * bs, J.nrows * bs, bs, J.rowptr, J.colidx, J.vals, xf, xc);
  return 0;
}

// ---- residual probe of the coarse solvers (alfi_coarse_factor / _sparse): e = a +-1 vector that is a function of the index, made
// on the device, and || r - e ||_inf reduced there -- a refactorisation per Newton step moves 8 bytes, not two coarse vectors
__device__ inline double probe_entry(int64_t i) {
  uint32_t h = (uint32_t)i * 2654435761u + 12345u;
  h ^= h >> 15;
  h *= 2246822519u;
  h ^= h >> 13;
  return (h & 1u) ? 1.0 : -1.0;
}
__global__ void probe_fill_kernel(double* __restrict__ e, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) e[i] = probe_entry(i);
}
__global__ void probe_residual_kernel(const double* __restrict__ r, int64_t n, unsigned long long* __restrict__ worst) {
  double w = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double d = fabs(r[i] - probe_entry(i));
    if (!(d == d)) d = INFINITY;
    w = d > w ? d : w;
  }
  // (non-negative doubles order like their bit patterns)
  if (w > 0.0) atomicMax(worst, (unsigned long long)__double_as_longlong(w));
}

int launch_probe_fill(alfi_ctx* ctx, double* e, int64_t n) {
  if (n == 0) return 0;
  int64_t blocks = std::min<int64_t>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(probe_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, e, n);
  ALFI_HIP_CHECK(ctx, hipGetLastError());
  return 0;
}

// *worst_host = max_i | r_i - e_i | (infinity if any entry is not a number); synchronises
int launch_probe_residual(alfi_ctx* ctx, const double* r, int64_t n, double* worst_host) {
  unsigned long long* d = nullptr;
  ALFI_HIP_CHECK(ctx, hipMalloc((void**)&d, sizeof(unsigned long long)));
  hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream);
  if (e == hipSuccess && n > 0) {
    int64_t blocks = std::min<int64_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(probe_residual_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, r, n, d);
    e = hipGetLastError();
  }
  unsigned long long bits = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&bits, d, sizeof(bits), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return alfi_set_error(ctx, ALFI_E_HIP, "coarse probe: %s", hipGetErrorString(e));
  double w;
  std::memcpy(&w, &bits, sizeof(w));
  *worst_host = w;
  return 0;
}
