// Internal structures of libalfi_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>
#include "alfi_hip.h"

// Every host <-> device copy the library makes is counted (alfi_transfer_stats): that a Newton step moves nothing but scalars
// across PCIe is tested (tests/test_gpu_newton_state.py), not asserted.  Process-wide counters.
inline std::atomic<int64_t> g_alfi_h2d_bytes{0}, g_alfi_d2h_bytes{0};
inline hipError_t alfi_counted_memcpy(void* dst, const void* src, size_t n, hipMemcpyKind kind) {
  if (kind == hipMemcpyHostToDevice) g_alfi_h2d_bytes += (int64_t)n;
  else if (kind == hipMemcpyDeviceToHost) g_alfi_d2h_bytes += (int64_t)n;
  return hipMemcpy(dst, src, n, kind);
}
inline hipError_t alfi_counted_memcpy_async(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s) {
  if (kind == hipMemcpyHostToDevice) g_alfi_h2d_bytes += (int64_t)n;
  else if (kind == hipMemcpyDeviceToHost) g_alfi_d2h_bytes += (int64_t)n;
  return hipMemcpyAsync(dst, src, n, kind, s);
}
#define hipMemcpy(dst, src, n, kind) alfi_counted_memcpy(dst, src, n, kind)
#define hipMemcpyAsync(dst, src, n, kind, stream) alfi_counted_memcpy_async(dst, src, n, kind, stream)

// HOST helper for the set-up passes over large index arrays (validation of contributor lists, layout conversions): fn(begin,
// end) on disjoint ranges of [0, n) over up to 16 host threads (a GPU box gives one process about that many cores); the
// ranges are contiguous and fixed by n and the thread count only.  Small n: the calling thread.
template <class F>
inline void host_parallel_ranges(int64_t n, F&& fn) {
  if (n <= 0) return;
  unsigned hw = std::thread::hardware_concurrency();
  int64_t nt = hw ? (hw < 16 ? hw : 16) : 4;
  if (n < ((int64_t)1 << 16)) nt = 1;
  if (nt <= 1) {
    fn((int64_t)0, n);
    return;
  }
  const int64_t step = (n + nt - 1) / nt;
  std::vector<std::thread> th;
  for (int64_t t = 1; t < nt; ++t) {
    const int64_t a = t * step, b = a + step < n ? a + step : n;
    if (a < b) th.emplace_back([&fn, a, b] { fn(a, b); });
  }
  fn((int64_t)0, step < n ? step : n);
  for (auto& t : th) t.join();
}

struct alfi_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  // exchange points since alfi_ctx_comm_stats(reset): halo exchanges (forward + reverse), all-reduces, doubles this rank sent
  int64_t comm_nhalo = 0, comm_nred = 0, comm_sent = 0;
  int32_t* dev_err = nullptr;       // sticky device-side error word (a bounded wait of a persistent kernel ran out); read by
                                    // every call that synchronises (alfi_ctx_sync, alfi_memcpy_d2h)
  bool use_graph = false;           // alfi_ctx_set_graph: replay whole cycles as hipGraphs (not while profiling / partitioned)
  void* big_arena = nullptr;        // scratch of the large-block factorisation (kernels_bigpatch.hip), kept between calls
  size_t big_arena_bytes = 0;
  void* asm_scratch = nullptr;      // element blocks / element vectors of the operator refresh (kernels_assemble.hip), grow-only
  size_t asm_scratch_bytes = 0;
  int64_t asm_scratch_limit = (int64_t)24 << 30;   // bytes of element blocks per batch of cells (alfi_ctx_set_assembly_scratch)
  bool own_stream = false;
  std::string err;
  // profiling
  int prof = 0;   // 0 off, 1 every class, 2 PATCH_APPLY and COMM only (alfi_prof_enable)
  struct EvPair {
    hipEvent_t a, b;
    int kind;
    int tag;
  };
  std::vector<EvPair> ev_pool;   // all created pairs
  size_t ev_used = 0;            // pairs in use since last reset
  int cur_tag = -1;      // level id attached to profiling records
  int next_level_id = 0;
  // scratch for reductions: partial sums [RED_BLOCKS][RED_MAXV]
  double* red_partial = nullptr;
  double* red_partial2 = nullptr;   // second set (|w|^2 partials written while the dot partials are still being read)
  // mesh-partition parallelism (alfi_ctx_set_comm)
  alfi_comm_fn comm = nullptr;
  void* comm_user = nullptr;
  int big_mult_ncu = 0, big_mult_per_cu[2] = {0, 0};   // the same for the large-patch sweep (kernels_bigpatch.hip)
  int mult_ncu = 0, mult_per_cu[2] = {0, 0};   // resident-grid size of the persistent multiplicative sweep on THIS ctx's device
  bool comm_allow_self = false;   // test hook (alfi_ctx_comm_allow_self): a rank may be its own neighbour
  bool exact_norm = true;   // partitioned FGMRES: second all-reduce for |w - V h| (PETSc's VecNorm); false: Pythagorean identity
  double* dred = nullptr;  // device buffer that is all-reduced: [0, RED_MAXV) dots, [RED_MAXV] norm^2 (caller-owned with
                           // alfi_ctx_set_comm, owned by the ctx with alfi_ctx_comm_init)
  // native transport (alfi_ctx_comm_init, comm.hip): the ctx owns an RCCL communicator and exchanges by itself
  struct NativeComm* nat = nullptr;
};

// comm.hip: the library's own exchanges over RCCL (no host callback)
struct alfi_level;
int native_allreduce(alfi_ctx* ctx, int64_t offset, int64_t count);
// dir 0: forward (send buffer segments -> the neighbours' receive buffers), 1: reverse (receive buffer segments back to
// the owners' send buffers); async: on the transport's own stream between an event pair (..._BEGIN), finished by native_wait
int native_exchange(alfi_level* L, int dir, bool async);
int native_wait(alfi_ctx* ctx);
void native_destroy(alfi_ctx* ctx);

constexpr int RED_BLOCKS = 1024;  // blocks used by the two-stage dot/norm reductions
constexpr int RED_MAXV = 32;      // max simultaneous dot products

int alfi_set_error(alfi_ctx* ctx, int code, const char* fmt, ...);

// ALFI_TEST_LARGE_PATHS=1 -- a TEST HOOK, not a tuning switch: the size thresholds that select the kernels of the large levels
// (nnz-balanced SpMV with the fix-up launch and the de-duplicated x gathers instead of whole-row chunks, the general smoother
// chain instead of the fused iteration of small levels) are lowered to zero, so that the small hierarchies of the test suite
// run the code the 10 M-dof levels run.  Read once per process.
bool alfi_test_large_paths();

#define ALFI_HIP_CHECK(ctx, call)                                                                      \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return alfi_set_error(ctx, ALFI_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),    \
                            __FILE__, __LINE__);                                                       \
  } while (0)

#define ALFI_CHECK(expr)     \
  do {                       \
    int rc_ = (expr);        \
    if (rc_ != 0) return rc_; \
  } while (0)

// RAII-less profiling scope: begin/end record events on the ctx stream when profiling is on.
int alfi_prof_begin(alfi_ctx* ctx, int kind);
int alfi_prof_end(alfi_ctx* ctx, int token);

constexpr int SMALL_PATCH_MAX = 160;   // register-resident inversion / one wave per patch up to here (kernels_patch.hip)
constexpr int PATCH_MAX = 4096;        // blocked MFMA inversion / one workgroup per patch beyond (kernels_bigpatch.hip)

// ---- storage of one dense patch inverse (n x n, rows padded to ld = n rounded up to even) --------------------------------
// Row pieces: as many 128-row pieces as fit, then the binary digits of the remainder (64, 32, ..., 2).  A piece of R rows
// is stored [column][R] contiguously, pieces follow each other, so the whole inverse is ld * n contiguous doubles that
// the apply kernel streams front to back exactly once, every wave instruction reading 1 KiB of consecutive bytes
// (64 / (R/2) columns at a time).  Offset of entry (r, c):
__host__ __device__ inline int64_t patch_inv_index(int r, int c, int n, int ld) {
  int row0 = r & ~127, rows = 128;
  if (row0 + 128 > ld) {
    const int rem = ld - row0;
    int rr = r - row0;
    for (int bit = 64; bit >= 2; bit >>= 1) {
      if (rem & bit) {
        if (rr < bit) {
          rows = bit;
          break;
        }
        rr -= bit;
        row0 += bit;
      }
    }
  }
  return (int64_t)row0 * n + (int64_t)c * rows + (r - row0);
}

// Interleaved storage of SMALL patch inverses (every n_p <= 2 G <= 32): G lanes share a patch (lane l owns the row pair
// 2l, 2l + 1), PW = 64 / G patches share a wave, and the wave's PW patches are stored column by column,
//     il[((w * nc + c) * PW * G + q * G + l) * 2 + (r & 1)],   p = w * PW + q,  l = r / 2,  c < nc = max_np of the level
// (rows / columns a patch does not have are zero), so that one wave instruction -- column c of all PW patches -- reads
// PW * G * 16 consecutive bytes and consecutive instructions continue where the previous one stopped.  The row-piece layout
// above gives such a wave 3 pieces x 8 patches of 16 .. 64-byte segments per instruction.
__host__ __device__ inline int64_t patch_il_index(int64_t p, int r, int c, int G, int nc) {
  const int PW = 64 / G;
  const int64_t w = p / PW;
  const int q = (int)(p - w * PW);
  return ((w * nc + c) * (int64_t)(PW * G) + q * G + (r >> 1)) * 2 + (r & 1);
}

// Block-CSR on the device.  Two value layouts:
//   flat == 0: vals[k][bs*bs] (the host layout); one group of lanes per block row (bsr_spmv_kernel);
//   flat == 1: "lane-major": blocks in groups of 64; within a group the bs*bs entries of a block are taken in pairs
//              (e0,e1), (e2,e3), ... and stored vals[k / 64][pair][k % 64][2] (a trailing unpaired entry as [k % 64]), so
//              that a wave handling 64 consecutive blocks reads each pair plane as one contiguous 1-KiB request of
//              16 B per lane (bs = 3: four of those plus one 512-B request); the sign bit of colidx[k]
//              marks the first block of a block row; chunk_row[c] = block row of block c * SPMV_CHUNK.  Used by the
//              nnz-balanced segmented SpMV (bsr_spmv_flat_kernel) whenever every block row is non-empty.
#ifndef ALFI_SPMV_U
#define ALFI_SPMV_U 4                     // tuning builds: -DALFI_SPMV_U=2 / 8 (4 measured best)
#endif
constexpr int SPMV_U = ALFI_SPMV_U;       // wave iterations of 64 blocks
constexpr int SPMV_CHUNK = 64 * SPMV_U;   // blocks per wave
struct DevBSR {
  int64_t nbrows = 0, nbcols = 0, nnzb = 0;
  int bs = 0;
  int32_t* rowptr = nullptr;
  int32_t* colidx = nullptr;
  double* vals = nullptr;
  int flat = 0;
  int64_t kbase = 0;            // first block of a block-row-range view (flat layout; 0 for the upload itself)
  bool view = false;            // a row-range view: rows outside the range are not touched
  int64_t nchunks = 0;
  int32_t* chunk_row = nullptr;
  double* carry = nullptr;      // (nchunks, bs): partial sums of block rows that continue into the next chunk
  int32_t* carry_row = nullptr;  // (nchunks): that block row, or -1
  // small matrices (nnzb <= SPMV_ALIGNED_MAX): chunks of whole block rows, chunk c = blocks [chunk_start[c], chunk_start[c+1])
  bool aligned = false;
  int64_t* chunk_start = nullptr;   // (nchunks + 1)
  // large matrices: x columns de-duplicated per workgroup (4 chunks = SPMV_WG blocks; neighbouring block rows share most of
  // their columns -- config 4: 247 distinct columns among 1024 blocks).  ucol[uptr[g] .. uptr[g+1]) = distinct columns of
  // group g ascending, lidx[k] = position of block k's column in its group's list (bit 15 = first block of a block row)
  bool dedup = false;
  uint16_t* lidx = nullptr;
  int32_t* ucol = nullptr;
  int32_t* uptr = nullptr;
};
constexpr int SPMV_WG = 4 * SPMV_CHUNK;      // blocks per workgroup of the flat product
constexpr int SPMV_DEDUP_MAX = 1024;         // distinct columns per group the product stages in LDS (24 KiB at bs = 3)
constexpr int64_t SPMV_ALIGNED_MAX = (int64_t)1 << 21;   // blocks; beyond, the product is bandwidth-bound and equal-sized
                                                          // chunks (+ the fix-up launch) keep every lane busy
__host__ __device__ inline int64_t bsr_val_index(int flat, int64_t k, int rc, int bb) {
  if (!flat) return k * bb + rc;
  const int64_t g = (k >> 6) * 64 * bb;
  const int l = (int)(k & 63), npair = bb >> 1;
  if (rc < 2 * npair) return g + (int64_t)(rc >> 1) * 128 + l * 2 + (rc & 1);
  return g + (int64_t)npair * 128 + l;
}

// Condensed ("statically condensed") patch factors, alfi_patches_set_groups: the dofs of patch p are split into groups
// I_1 .. I_m that are coupled to the rest of the patch only through the remaining (skeleton) dofs S -- for the macro-star
// patches of the Scott-Vogelius discretisation the dofs interior to one macro cell each (alfi/relaxation.py:163-177 on the
// Alfeld-split meshes of alfi/bary.py) -- and inv(A_p) is stored EXACTLY as the block factorisation
//     t_g = X_g x_g,   y_S = inv(Sigma) (x_S - sum_g B_g t_g),   y_g = t_g - W_g y_S[S_g]
// with X_g = inv(A_gg), B_g = A[S_g, g], W_g = X_g A[g, S_g], Sigma = A_SS - sum_g B_g W_g (S_g: the skeleton dofs coupled
// to group g).  Storage sum_g (m_g^2 + 2 m_g s_g) + s^2 doubles instead of n_p^2: 7 x less for the P3 macro stars of
// config 5 (m_g <= 48, s_g <= 51, s <= 585 of n_p <= 1941), and the apply streams exactly that.
struct CondDev {
  const int32_t* dofs = nullptr;   // (sum_n) global dof of every patch entry, condensed order [groups | skeleton]
  const int32_t* slot = nullptr;   // (sum_n) position of the entry in the patch's ascending order (its staging slot)
  const int64_t* gptr = nullptr;   // (npatch+1) groups of patch p
  const int32_t* g_off = nullptr;  // per group: first entry inside the patch (condensed order)
  const int32_t* g_m = nullptr;    //            entries
  const int32_t* g_sc = nullptr;   //            coupled skeleton entries
  const int32_t* g_uoff = nullptr; //            offset of its B_g t_g segment in the patch's u buffer
  const int64_t* g_mat = nullptr;  //            offset of [X (m x m) | B (sc x m) | W (m x sc)] (column-major each) in mat
  const int64_t* g_sidx = nullptr; //            offset of its sc skeleton positions in sidx
  const int32_t* sidx = nullptr;
  const int32_t* p_nI = nullptr;   // per patch: interior entries = where the skeleton starts
  const int64_t* sptr = nullptr;   // (npatch+1) prefix sums of the skeleton sizes
  const int64_t* sinv_ptr = nullptr;  // (npatch+1) offsets of inv(Sigma) (row-piece layout, patch_inv_index) in sinv
  const int32_t* s_uptr = nullptr; // (sum_s + 1) skeleton row -> contributions in the u buffer
  const int32_t* s_uidx = nullptr;
  double* mat = nullptr;
  double* sinv = nullptr;
  const int32_t* order = nullptr;  // (npatch) patches by descending factor bytes: dispatch order of a full-range apply
  // three-launch apply (cond_front / cond_sigma / cond_back): the rows of inv(Sigma) in chunks of COND_SIGMA_ROWS
  const int32_t* ch_patch = nullptr;  // per chunk: its patch (natural order; the chunks of patch p: h_cond_chptr[p .. p + 1])
  const int32_t* ch_row = nullptr;    //            its first row
  double* tmp = nullptr;              // (sum_n) per patch [t_g ... | y_S] between the launches
  const int64_t* uptr = nullptr;      // (npatch+1) prefix sums of the patches' u buffer lengths
  const int32_t* u_dst = nullptr;     // (sum of those) u buffer entry -> its place in the row-sorted order (inverse of s_uidx)
  // row PAIRS (rows 2 i, 2 i + 1 of a group's X / W, or of its B): a lane of cond_front / cond_back owns one
  const int64_t* xp_ptr = nullptr;    // (npatch+1) prefix sums of the patches' X / W row pairs
  const int32_t* xp_grp = nullptr;    // per pair: its group
  const int32_t* g_xp = nullptr;      // per group: its first pair in the patch's list
  const int64_t* bp_ptr = nullptr;    // the same for the rows of B (the u buffer)
  const int32_t* bp_grp = nullptr;
  const int32_t* g_bp = nullptr;
  // chunks of consecutive groups of one patch (<= 256 row pairs of X / W and of B): the work units of cond_gfront / cond_gback.
  // One descriptor per workgroup (CondChunk); a lane decodes its row pair from the per-pair group index and the group arrays
  // (cond_xpair / cond_bpair, kernels_bigpatch.hip).  Round 3 stored a 48 / 32-byte descriptor per LANE: 8 % more bytes than
  // the factors themselves on config 5's finest level.
  const struct CondChunk* gc = nullptr;   // (nchunk)
  double* ubuf = nullptr;             // (sum of the u buffer lengths) u_g = B_g t_g in the row-sorted order of u_dst
};
struct CondChunk {
  int64_t off;        // patch_ptr[p]
  int64_t ubase;      // uptr[p]
  int64_t sidx0;      // first entry of the chunk's S_g lists in CondDev::sidx
  int64_t stage_off;  // stage_ptr[p]
  int32_t xq0, xq1;   // the chunk's X / W row pairs in xpd
  int32_t bq0, bq1;   // the chunk's B row pairs in bpd
  int32_t e0, ne;     // its interior entries (adjacent in the condensed order)
  int32_t u0, nu;     // its entries of the patch's u layout
  int32_t nI;
  int32_t xp0, bp0;   // first row pair of the PATCH in the X / W and in the B numbering (a pair's place inside its group)
  int32_t pad;
};
// what a lane needs of its row pair (decoded in the kernel, never stored)
struct CondXPair {
  int64_t xoff, woff; // the pair's rows in X and in W (offsets into CondDev::mat)
  int32_t ld, m, sc;  // leading dimension of X / W, columns of X (= group size), columns of W
  int32_t o, uo, i;   // group's first interior entry, its offset in the u layout, the pair's first row inside the group
  int32_t sl0, sl1;   // staging slots of the two rows (-1: the row does not exist)
};
struct CondBPair {
  int64_t boff;       // the pair's rows in B
  int32_t ld, m, o;   // leading dimension, columns (= group size), group's first interior entry
  int32_t d0, d1;     // places of the two results in the row-sorted u buffer (-1: the row does not exist)
  int32_t pad;
};

// rows of inv(Sigma) per workgroup of the sigma kernels of the condensed apply (4 waves; kernels_bigpatch.hip, chunk table in
// alfi_patches_set_groups)
#ifndef ALFI_COND_SIGMA_ROWS
#define ALFI_COND_SIGMA_ROWS 64
#endif
constexpr int COND_SIGMA_ROWS = ALFI_COND_SIGMA_ROWS;

// storage of one group's matrices in CondDev::mat: [X (m x m) | B (sc x m) | W (m x sc)], column-major each, the leading
// dimensions rounded up to EVEN (a lane streams two rows of a column with one 16-byte load; the pad row is never stored)
// (Measured and dropped, round 5: leading dimensions of > 8 rows rounded up to whole 128-byte lines and every group on a line
// boundary -- no column shares a line with its neighbour, +3.8 % bytes: config 5 24.13 against 23.60-23.68 ms per cycle, same box.)
__host__ __device__ inline int cond_ldim(int rows) { return (rows + 1) & ~1; }
__host__ __device__ inline int cond_pairs(int rows) { return (rows + 1) / 2; }     // row pairs a lane each
__host__ __device__ inline int64_t cond_group_doubles(int m, int sc) {
  return (int64_t)cond_ldim(m) * m + (int64_t)cond_ldim(sc) * m + (int64_t)cond_ldim(m) * sc;
}

// device-side data of the operator refresh (alfi_level_set_assembly, kernels_assemble.hip)
struct AssemblyDev {
  bool ready = false;
  int nloc = 0;
  int64_t ncell = 0, npairs = 0;
  int64_t nstate = 0;            // nodes of the state vector the cells index (= the level's nodes; on a partitioned level the
                                 // local nodes followed by the other nodes of the cells that touch them)
  bool full_div = false;         // grad-div term gamma (div u, div v) (Scott-Vogelius) instead of the cell-averaged one
  uint8_t* bc_code = nullptr;    // (nnzb) Dirichlet flags of every block's row / column dofs (built on first use)
  uint8_t* bc_all = nullptr;     // (n) partitioned levels: Dirichlet dofs among ALL local dofs, ghosts included
                                 // (alfi_level_set_assembly_bc; the level's own mask marks owned dofs only)
  int64_t* cptr = nullptr;       // (nnzb + 1) contributor lists per block
  int32_t* ccell = nullptr;      // (npairs) cell
  uint16_t* cba = nullptr;       // (npairs) b * nloc + a
  int32_t* cell_nodes = nullptr; // (ncell, nloc)
  double* grad = nullptr;        // (ncell, d + 1, d) gradients of the barycentric coordinates
  double* vol = nullptr;         // (ncell)
  double* etab = nullptr;        // (nloc * nloc, 2 (d+1) nloc + (d+1)^2): per (a, b) the slices of T1 and S the cell kernel
                                 // reads as wave-uniform operands (layout: kernels_assemble.hip)
  double* bItab = nullptr;       // (nloc, d + 1) avg d_i phi_a
  int32_t* diag = nullptr;       // (nodes) index of the diagonal block of every block row
  // SUPG (alfi_level_set_supg): quadrature tables of the element and the cell sizes
  bool supg_ready = false;
  int nq = 0;
  double* wq = nullptr;          // (nq) weights summing to 1
  double* phi = nullptr;         // (nq, nloc)
  double* dphi = nullptr;        // (nq, nloc, d + 1) derivatives w.r.t. the barycentric coordinates
  double* d2phi = nullptr;       // (nq, nloc, d + 1, d + 1)
  double* hcell = nullptr;       // (ncell) cell size (2 x circumradius)
  int nq8 = 0;                   // nq rounded up to a multiple of eight (zero-weight copies of point 0): two k-steps of the MFMA per chunk
  double* wq8 = nullptr;         // (nq8)
  double* qtab = nullptr;        // (nq8, nloc, 1 + (d+1) + (d+1)(d+2)/2): phi | dphi | upper triangle of d2phi per (point, node)
};

struct alfi_level {
  alfi_ctx* ctx = nullptr;
  AssemblyDev asmb;
  int id = 0;
  int64_t n = 0;  // scalar dofs
  int bs = 0;
  DevBSR A;
  // partition: owned prefix [0, n_own) of the local numbering, then ghosts (serial: n_own == n)
  int64_t n_own = 0;
  DevBSR A_own;             // view of A restricted to the owned block rows
  int max_row_blocks = 0;   // longest block row of A
  // overlap of the forward halo with work that needs no ghost value (alfi_level_set_overlap)
  bool overlap = false;
  DevBSR A_int, A_bnd;       // owned rows without / with ghost columns (own chunk tables)
  int64_t npatch_int = 0;    // leading patches without ghost dofs
  bool distributed = false;  // smoother / SpMV exchange halos and all-reduce
  bool has_halo = false;
  int64_t halo_nsend = 0, halo_nghost = 0;  // nodes
  int32_t* halo_send_nodes = nullptr;       // (nsend) owned nodes, grouped by destination
  double *halo_sendbuf = nullptr, *halo_recvbuf = nullptr;  // caller-owned unless own_halo_bufs
  bool own_halo_bufs = false;
  // alfi_level_set_neighbours: the ranks this level exchanges with and the segment of the send / receive buffer that
  // belongs to each (offsets and counts in doubles)
  std::vector<int> nbr_rank;
  std::vector<int64_t> nbr_send_off, nbr_send_cnt, nbr_recv_off, nbr_recv_cnt;
  int64_t rev_nuniq = 0;                    // reverse-add: unique owned nodes receiving contributions
  // merged reverse-add + forward exchange (alfi_level_set_sum_exchange; native transport only): every holder of a shared
  // node sends its partial value to every other holder, all add in ascending rank order
  bool sum_ready = false;
  std::vector<int> sum_rank;
  std::vector<int64_t> sum_off, sum_cnt;    // doubles, per neighbour (same layout for send and receive)
  int32_t *sum_send_nodes = nullptr, *sum_nodes = nullptr, *sum_ptr = nullptr, *sum_src = nullptr;
  int64_t sum_nsend = 0, sum_nshared = 0;   // nodes
  double *sum_sendbuf = nullptr, *sum_recvbuf = nullptr;
  int32_t *rev_nodes = nullptr, *rev_ptr = nullptr, *rev_pos = nullptr;
  int32_t* bc_dofs = nullptr;
  int64_t nbc = 0;
  uint8_t* bc_mask = nullptr;  // (n) 1 on Dirichlet dofs: the dof-wise sum of the patch results copies x there
  // patches
  int64_t npatch = 0, sum_n = 0, sum_n2 = 0, inv_doubles = 0;
  int max_np = 0;
  int64_t* patch_ptr = nullptr;   // (npatch+1) offsets into patch_dofs / staging buffer
  int32_t* patch_dofs = nullptr;  // (sum_n)
  int64_t* inv_ptr = nullptr;     // (npatch+1) offsets (doubles) into inv
  int64_t* stage_ptr = nullptr;   // (npatch+1) offsets into stage (ld_p slots per patch, so 16-byte aligned)
  double* inv = nullptr;          // column-major padded inverses
  // small-patch levels (all n_p <= 32: the 2-D stars): a wave-contiguous copy of the inverses the additive apply streams
  // (patch_il_index below); derived from `inv` at the end of every factorisation / repair (kernels_patch.hip)
  double* inv_il = nullptr;
  int64_t il_doubles = 0;
  int il_G = 0, il_nc = 0;        // lanes per patch (row pairs), columns stored per patch (= max_np)
  bool il_valid = false;
  double* stage = nullptr;        // (sum ld_p) staged patch results
  int64_t stage_len = 0;
  int32_t* dof_ptr = nullptr;     // (n+1) CSR dof -> positions in stage
  int32_t* dof_pos = nullptr;     // (sum_n)
  bool factored = false;
  bool pou = false;               // patch_pc_patch_partition_of_unity: additive results weighted by 1 / multiplicity
  int* status = nullptr;          // device flag: nonzero if a zero pivot was met
  // residual probe of the stored inverses + pivoted repair (kernels_check.hip)
  double* chk = nullptr;          // device: [0] worst residual (as ordered bits), [1] number of flagged patches (int)
  int32_t* chk_list = nullptr;    // device: flagged patch ids
  int chk_cap = 0;
  double chk_worst = -1.0, chk_worst_after = -1.0;   // of the last factorisation: before / after the repair (-1: not run)
  int64_t chk_flagged = 0, chk_repaired = 0;
  // condensed patch factors (alfi_patches_set_groups)
  bool cond = false;
  bool inv_shrunk = false;               // the dense inverse storage was released in favour of the condensed factors
  CondDev cd;
  std::vector<void*> cond_allocs;        // every device array cd points to
  std::vector<int64_t> h_sptr;           // host copy of cd.sptr (sizes of the Schur complements)
  int64_t cond_ngroups = 0, cond_mat_doubles = 0, cond_sinv_doubles = 0;
  int cond_lds_bytes = 0, cond_max_s = 0, cond_umax = 0, cond_lds_front = 0, cond_lds_back = 0;
  std::vector<int64_t> h_cond_chptr;      // (npatch+1) chunks of the three-launch condensed apply
  std::vector<int64_t> h_cond_gcptr;      // (npatch+1) group chunks of the patches
  int cond_lds_gfront = 0, cond_lds_gback = 0;
  std::vector<int64_t> h_cond_gptr;      // host copy of cd.gptr
  // multiplicative sweeps: positions of the iteration sequence grouped into dependency wavefronts
  bool mult = false, mult_symmetrise = false;
  bool mult_big = false;                // a patch holds more than 64 nodes: workgroup-per-patch sweep kernel
  int32_t* mult_seq = nullptr;          // (nit) patch ids, wavefront-major
  std::vector<int64_t> mult_wave_ptr;   // (nwave+1) offsets into mult_seq
  // persistent schedule of the whole apply (forward sweep, then -- symmetrised -- the wavefronts in reverse): item -> patch,
  // predecessor counts (the last writers of the nodes an item reads), successor lists
  int32_t mult_nitems = 0;
  int32_t *mult_items = nullptr, *mult_pred0 = nullptr, *mult_pred = nullptr, *mult_succ_ptr = nullptr, *mult_succ = nullptr;
  int32_t* mult_ctl = nullptr;          // [0] ticket counter (zeroed before every launch)
  int32_t* mult_rowtab = nullptr;       // (npatch, 64, 3) first block, block count, node of every patch node (zero-padded)
  std::vector<int32_t> h_patch_dofs;    // host copy of the patch dofs (needed to build the wavefronts)
  // FGMRES workspace
  int kmax = 0;
  int64_t ldv = 0;       // stride of V and Z: n rounded up to even (16-byte aligned basis vectors)
  double* V = nullptr;   // (kmax+1) x ldv
  double* Z = nullptr;   // kmax x n
  double* w = nullptr;   // n
  double* hs = nullptr;  // small device arrays: Hessenberg etc.
  // coarse dense inverse
  double* cinv = nullptr;
  bool cinv_owned = false;
  struct MfDev* mf = nullptr;     // sparse (multifrontal) coarse factorisation, alfi_coarse_factor_sparse
  double cinv_residual = -1.0;    // || A X e - e ||_inf of the inverse built by alfi_coarse_factor (-1: supplied by the caller)
  // multigrid work vectors (owned by alfi_mg but stored per level)
  double *mg_b = nullptr, *mg_x = nullptr, *mg_r = nullptr;
  std::vector<int64_t> h_patch_ptr;  // host copy (for get_inverse)
  std::vector<int64_t> h_inv_ptr;
};

struct alfi_transfer {
  alfi_ctx* ctx = nullptr;
  alfi_level *coarse = nullptr, *fine = nullptr;
  int bs = 0;
  DevBSR P, PT, PTp, DI, DIT;
  bool ptp_alias = false;
  int64_t nblk = 0;
  int m = 0, ld = 0;
  int32_t* blk_dofs = nullptr;  // (nblk*m)
  double *KII = nullptr, *DII = nullptr;
  double* binv = nullptr;  // (nblk, m cols, ld) column-major padded inverses (m <= 32) or row-piece layout (m > 32)
  // m > 32 (macro-cell blocks of the Scott-Vogelius transfer): the blocks go through the patch kernels
  bool patch_mode = false;
  int64_t bstride = 0;                  // doubles between consecutive block inverses
  int64_t *pm_ptr = nullptr, *pm_inv_ptr = nullptr, *pm_stage_ptr = nullptr;   // (nblk+1) each
  int32_t* pm_iota = nullptr;           // (nblk*m) identity index list: compact input vectors
  double* pm_tmp = nullptr;             // (nblk*ld) only for odd m: output of the patch kernel before compaction
  double* pm_res = nullptr;             // (nblk*m) residual b - A t of the refinement step (NULL: ALFI_TRANSFER_REFINE=0)
  double* pm_cor = nullptr;             // (nblk*m) its correction X r
  double* AIIt = nullptr;               // (nblk, m, ld) nu K_II + gamma D_II, COLUMN-major per block (entry (i, j) at j * ld + i): what
                                        // the refinement's residual reads -- one matrix instead of K_II and D_II, lanes along a column
  double *tI = nullptr, *bI = nullptr;  // compact interior vectors (nblk*m)
  double* tmp_f = nullptr;              // fine work vector
  int32_t* inj = nullptr;               // (coarse nodes) fine node coinciding with each coarse node
  struct DevCSR* injm = nullptr;        // non-nested hierarchies: (coarse nodes x fine nodes) point-evaluation weights
  double gamma = 0, nu = 0;
  bool ready = false;
  int* status = nullptr;
};

// one captured cycle (alfi_ctx_set_graph): the whole V- or full cycle for a given (b, x) pair replayed as a hipGraph
struct CycleGraph {
  int kind = 0;                 // 0 V-cycle, 1 full cycle
  const double* b = nullptr;
  double* x = nullptr;
  std::vector<uint64_t> sig;    // everything a captured kernel argument depends on (pointers, nu, gamma, k, ...)
  hipGraphExec_t exec = nullptr;
  bool failed = false;          // capture was refused once: run this cycle eagerly from then on
};

struct alfi_mg {
  alfi_ctx* ctx = nullptr;
  std::vector<alfi_level*> levels;
  std::vector<alfi_transfer*> transfers;
  int k = 0;
  int robust = 0;
  std::vector<CycleGraph> graphs;
};

struct DevCSR {
  int64_t nrows = 0, ncols = 0, nnz = 0;
  int32_t* rowptr = nullptr;
  int32_t* colidx = nullptr;
  double* vals = nullptr;
};

struct alfi_saddle {
  alfi_ctx* ctx = nullptr;
  alfi_mg* mg = nullptr;
  alfi_level* fine = nullptr;
  int64_t nu_dofs = 0, np_dofs = 0;
  DevCSR B, BT;
  double* minv = nullptr;  // 1 / diag(M_p)
  DevCSR Minv;             // block-diagonal inverse mass matrix of a discontinuous P_k pressure (alfi_saddle_set_mass_inverse)
  bool has_Minv = false;
  double nu = 0, gamma = 0;
  bool remove_nullspace = false;
  // partitioned finest level: vectors hold (owned velocity dofs | owned pressure dofs); B has the rank's pressure rows over
  // ALL local velocity dofs (owned + ghost), BT the local velocity rows (ghost rows = contributions for their owners)
  bool par = false;
  int64_t n_loc = 0;          // local velocity dofs incl. ghost slots (the length of a level vector)
  double np_global = 0.0;     // pressure dofs of all ranks (mean removal)
  double *wa = nullptr, *wb = nullptr, *wc = nullptr;   // level vectors (n_loc)
  // outer FGMRES workspace
  int restart = 0;
  int64_t ldv = 0;            // stride of V, Z: n rounded up to even
  double *V = nullptr, *Z = nullptr, *w = nullptr, *hs = nullptr, *tmp_u = nullptr, *tmp_p = nullptr;
  double* dotbuf = nullptr;   // result of alfi_saddle_dot (serial levels)
};

// ---- kernel launch wrappers (defined in the .hip files) --------------------------------------------------------------
// y = A x (mode 0) or y = b - alpha * A x (mode 1)
int launch_bsr_spmv(alfi_ctx* ctx, const DevBSR& A, const double* x, double* y, const double* b, double alpha, int mode);
// host (nnzb, bs, bs) values -> d->vals in d's layout (staged through a bounded device buffer)
int upload_bsr_values(alfi_ctx* ctx, DevBSR* d, const double* host_vals);
int build_spmv_dedup(alfi_ctx* ctx, DevBSR* d);     // column de-duplication tables of the flat product (kernels_vec.hip)
int launch_patch_gather_dense(alfi_level* lvl);
int launch_patch_invert(alfi_level* lvl);
// kernels_check.hip: probe || A_p X_p e - e || of every stored inverse, pivoted re-inversion of the patches that fail
int patch_verify_and_repair(alfi_level* lvl, int unpivoted_status);
int launch_patch_invert_arrays(alfi_ctx* ctx, int64_t npatch, int max_np, const int64_t* patch_ptr,
                               const int64_t* inv_ptr, double* inv, int* status);
int launch_patch_apply_arrays(alfi_ctx* ctx, int64_t npatch, const int64_t* patch_ptr, const int32_t* patch_dofs,
                              const int64_t* inv_ptr, const int64_t* stage_ptr, const double* inv, const double* x,
                              double* stage);
int launch_patch_apply(alfi_level* lvl, const double* x, double* y);          // both stages, all patches
int launch_patch_apply_range(alfi_level* lvl, int64_t p0, int64_t p1, const double* x);   // stage 1, patches [p0, p1)
int launch_big_apply_range(alfi_level* lvl, int64_t p0, int64_t p1, const double* x);     // the same for levels with n_p > 160
int launch_big_factor_transfer(alfi_transfer* tr);                                            // dense nu K + gamma D blocks, m > 160
int launch_big_apply_arrays(alfi_ctx* ctx, int64_t npatch, int max_np, const int64_t* patch_ptr,
                            const int32_t* patch_dofs, const int64_t* inv_ptr, const int64_t* stage_ptr, const double* inv,
                            const double* x, double* stage);
int mf_factor(alfi_level* lvl, const double* coords, int dim, int leaf_nodes);   // multifrontal L D U of the level operator
int mf_solve(alfi_level* lvl, const double* b, double* x);
void mf_free(struct MfDev* m);
int64_t mf_bytes(const struct MfDev* m);
int launch_coarse_factor(alfi_level* lvl, double* out);                                   // dense inverse of the whole level operator
int launch_big_factor(alfi_level* lvl);
// condensed patches (kernels_bigpatch.hip): block factorisation / its apply for the patches [p0, p1)
int launch_cond_factor(alfi_level* lvl);
int launch_cond_apply_range(alfi_level* lvl, int64_t p0, int64_t p1, const double* x);
int launch_cond_schur_one(alfi_level* lvl, int64_t p, const int64_t* d_zero, double* scr);   // repair path (kernels_check.hip)                                                    // gather + blocked MFMA inversion
int build_patch_il(alfi_level* lvl);   // small-patch levels: (re)build the interleaved copy of the inverses from lvl->inv
int launch_patch_sum(alfi_level* lvl, const double* x, double* y);             // stage 2
int launch_patch_sum_range(alfi_level* lvl, int64_t i0, int64_t i1, const double* x, double* y);   // stage 2, dofs [i0, i1)
// one dependency wavefront of a multiplicative sweep: patches seq[0..count): y_p += inv(A_p) (x - A y)_p
int launch_patch_mult_wave(alfi_level* lvl, const int32_t* seq, int64_t count, const double* x, double* y);
// the whole (symmetrised) sweep as one launch of a resident grid walking the schedule with per-item dependency counters
int launch_patch_mult_persistent(alfi_level* lvl, const double* x, double* y);
// the same with a workgroup per patch: patches of more than 64 nodes (macro stars)
int launch_big_mult_wave(alfi_level* lvl, const int32_t* seq, int64_t count, const double* x, double* y);
int launch_big_mult_persistent(alfi_level* lvl, const double* x, double* y);   // ... as one launch over the ticketed schedule
int launch_invert_small_any(alfi_ctx* ctx, int nmax, int64_t nmat, const int64_t* ptr, const int64_t* inv_ptr,
                            int fixed_n, int64_t fixed_stride, double* inv, int* status);
int launch_block_build_invert(alfi_transfer* tr);
// out_compact = binv * in;  gather != nullptr: in is a level vector indexed through blk_dofs
int launch_block_gemv(alfi_transfer* tr, const double* in, double* out, bool gather_in);
int launch_scatter_sub(alfi_ctx* ctx, double* x, const int32_t* idx, const double* t, double alpha, int64_t n);
int launch_zero_dofs(alfi_ctx* ctx, double* x, const int32_t* idx, int64_t n);
int launch_copy_dofs(alfi_ctx* ctx, double* y, const double* x, const int32_t* idx, int64_t n);
int launch_dense_gemv(alfi_ctx* ctx, const double* A, const double* x, double* y, int64_t n);
// blas1
int launch_copy(alfi_ctx* ctx, double* y, const double* x, int64_t n);
int launch_axpy(alfi_ctx* ctx, double* y, const double* x, double a, int64_t n);                 // y += a x
// |r|^2 partials (red_blocks_for(n) of them) into ctx->red_partial; then beta = sqrt(sum of nblocks partials), grs = beta e_1
int launch_norm_partials(alfi_ctx* ctx, const double* r, int64_t n);
int launch_norm_init_finish(alfi_ctx* ctx, const double* partial, int nblocks, double* hs, int K);
int launch_reduce_partials(alfi_ctx* ctx, const double* partial, int nblocks, int nv, double* out);  // out[v] = sum_b partial[b][v]
int red_blocks_for(int64_t n);   // number of partials the two-stage reductions over n entries produce
// ww: nullptr, or (partitioned levels) the all-reduced |w|^2 before the projection -> |w_new|^2 by Pythagoras
int launch_hessenberg_update(alfi_ctx* ctx, const double* partial, int nblocks, const double* h, double* hs, int j,
                             int K, const double* ww);
int launch_hessenberg_scale(alfi_ctx* ctx, const double* partial, int nblocks, const double* h, double* hs, int j, int K,
                            double* vnext, const double* w, int64_t n, const double* ww);
// scalar CSR: mode 0: y = A x; 1: y = b - alpha A x; 2: y += A x
int launch_csr_spmv(alfi_ctx* ctx, const DevCSR& A, const double* x, double* y, const double* b, double alpha, int mode);
int launch_scale_rows(alfi_ctx* ctx, double* y, const double* x, const double* d, double a, int64_t n);  // y = a d x
int launch_remove_mean(alfi_ctx* ctx, double* x, int64_t n);
int launch_inject_csr(alfi_ctx* ctx, const struct DevCSR& J, int bs, const double* xf, double* xc);
int launch_sum_to(alfi_ctx* ctx, const double* x, int64_t n, double* out);             // *out = sum(x), fixed order
int launch_sub_scaled(alfi_ctx* ctx, double* x, int64_t n, const double* s, double f); // x -= f * *s
int launch_add(alfi_ctx* ctx, double* y, const double* a, const double* b, int64_t n); // y = a + b
int launch_xmy(alfi_ctx* ctx, double* w, const double* b, int64_t n);                                   // w = b - w                                            // x -= mean(x)
// halo helpers
int launch_halo_pack(alfi_ctx* ctx, double* buf, const double* v, const int32_t* nodes, int64_t nnodes, int bs);
int launch_halo_sum(alfi_ctx* ctx, double* v, const double* buf, const int32_t* nodes, const int32_t* ptr, const int32_t* src,
                    int64_t nshared, int bs);   // v[node] = sum over its sources (src < 0: v[node] itself), fixed order
int launch_halo_add(alfi_ctx* ctx, double* v, const double* buf, const int32_t* rev_nodes, const int32_t* rev_ptr,
                    const int32_t* rev_pos, int64_t nuniq, int bs);
int launch_scale_by_inv(alfi_ctx* ctx, double* v, const double* w, const double* scal, int64_t n);  // v = w / *scal
int launch_multi_dot(alfi_ctx* ctx, const double* V, int64_t stride, int nv, const double* w, double* out, int64_t n);
// w -= sum_v h[v] V_v; |w|^2 partials (red_blocks_for(n)) into norm_partial.  hblocks > 0: h is first reduced inside the
// kernel from the hblocks (<= 256) dot partials in ctx->red_partial (launch_multi_dot with out == nullptr) and published in h
int launch_multi_axpy_norm(alfi_ctx* ctx, const double* V, int64_t stride, int nv, double* h, double* w, int64_t n,
                           double* norm_partial, int hblocks);
int launch_fgmres_finish(alfi_ctx* ctx, double* hs, int k, int K);  // back substitution -> y
// fused smoother iteration of the small levels (kernels_vec.hip): z = f * (dof-wise sum of the staged patch results),
// v = f * w with f = 1 / |w| from the norm partials, and column j - 1 of the Hessenberg (j == 0: the rotated rhs)
int launch_patch_sum_scale(alfi_level* lvl, const double* w, double* z, double* v, const double* normpart, int nblocks,
                           const double* h, double* hs, int j, int K);
// w = A z with the partials of V_v . w (v < nv <= 16) in the same pass; *nblocks = number of partials per vector
int launch_bsr_spmv_dot(alfi_ctx* ctx, const DevBSR& A, const double* z, double* w, const double* V, int64_t stride, int nv,
                        double* partial, int* nblocks);
// kernels_assemble.hip: the level operator from its cells (element blocks -> one fixed-order gather); out_vals: the operator's layout
int launch_operator_refresh(alfi_level* lvl, double nu, double gamma, double adv, const double* d_state, bool with_elements,
                            bool with_supg, double weight, double magic, bool accumulate, bool apply_bc, double* out_vals);
int launch_element_mult(alfi_level* lvl, double nu, double gamma, double adv, const double* d_state, const double* dx, double* dy);
int launch_supg_residual(alfi_level* lvl, double nu, double weight, double magic, const double* d_state, double* d_F);
bool element_kernel_exists(int d, int nloc);
int launch_vals_from_lanes(alfi_ctx* ctx, const DevBSR& A, double* d_out);
int launch_probe_fill(alfi_ctx* ctx, double* e, int64_t n);                                    // the +-1 probe vector of the coarse solvers
int launch_probe_residual(alfi_ctx* ctx, const double* r, int64_t n, double* worst_host);   // max | r - e |, reduced on the device
int launch_apply_bc(alfi_level* lvl);
int launch_patch_invert_mfma(alfi_ctx* ctx, int64_t npatch, int max_np, const int64_t* patch_ptr, const int64_t* inv_ptr,
                             double* inv, int* status, int* handled);   // kernels_invert.hip
int launch_fgmres_finish_fused(alfi_ctx* ctx, const double* normpart, int nblocks, const double* h, double* hs, int k, int K);
int launch_update_solution(alfi_ctx* ctx, double* x, const double* Z, int64_t stride, int k, const double* y, int64_t n);
