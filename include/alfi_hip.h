/* alfi_hip.h -- C ABI of libalfi_hip.so: the MI355X (gfx950) implementation of alfi's multigrid hot path.
 *
 * Every entry point replaces one piece of third-party machinery that florianwechsung/alfi configures but does not
 * contain (file:line = where the reference selects / calls it; [3P] = PETSc / Firedrake code that is not in the
 * reference tree).  The reference-side binding a maintainer would add is shown in
 * INTEGRATION.md (a PCPython class and a transfer object calling these through ctypes).
 *
 * Conventions
 *   - every function returns 0 on success, a negative ALFI_E_* code otherwise; alfi_last_error() gives the message.
 *     No exception or abort crosses this boundary; HIP errors are captured and reported.
 *   - `const T* host` arguments are borrowed for the duration of the call and copied to the device.
 *   - `double* dvec` / `const double* dvec` arguments are DEVICE pointers (from alfi_malloc or any hipMalloc'ed
 *     buffer on the ctx's device) holding a level vector: n = nbrows * bs doubles, node-major / component-minor
 *     (alfi/bubble.py:86; PETSc block size = tdim, alfi/solver.py:512).
 *   - one ctx = one device = one HIP stream; calls are stream-ordered and asynchronous unless they return host data.
 *     A ctx is not thread-safe; use one ctx per thread / process (1 process per GPU, as 1 MPI rank in the reference).
 *   - all floating point is FP64, all indices int32 (PETSC_ARCH ...-int32, examples/submission_template.pbs:28)
 *     except offsets into per-patch storage (int64).
 */
#ifndef ALFI_HIP_H
#define ALFI_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALFI_OK 0
#define ALFI_E_HIP -1      /* a HIP runtime call failed */
#define ALFI_E_ARG -2      /* invalid argument / unsupported size */
#define ALFI_E_STATE -3    /* call sequence error (e.g. apply before factor) */
#define ALFI_E_SINGULAR -4 /* zero pivot met while inverting a patch or transfer block */
#define ALFI_E_COMM -5     /* an RCCL call failed / librccl could not be loaded */

typedef struct alfi_ctx alfi_ctx;
typedef struct alfi_level alfi_level;
typedef struct alfi_transfer alfi_transfer;
typedef struct alfi_mg alfi_mg;

/* ---- context, memory --------------------------------------------------------------------------------------------- */
/* stream: a hipStream_t to launch on (e.g. torch.cuda.current_stream().cuda_stream), or NULL to create one. */
int alfi_ctx_create(int device, void* stream, alfi_ctx** out);
int alfi_ctx_destroy(alfi_ctx* ctx);
int alfi_ctx_sync(alfi_ctx* ctx);
const char* alfi_last_error(alfi_ctx* ctx); /* ctx may be NULL: last error of a failed alfi_ctx_create */
int alfi_malloc(alfi_ctx* ctx, int64_t bytes, void** dptr);
int alfi_free(alfi_ctx* ctx, void* dptr);
int alfi_memcpy_h2d(alfi_ctx* ctx, void* dst, const void* src, int64_t bytes); /* synchronous */
int alfi_memcpy_d2h(alfi_ctx* ctx, void* dst, const void* src, int64_t bytes); /* synchronous */
int alfi_memset0(alfi_ctx* ctx, void* dst, int64_t bytes);                     /* stream-ordered */

/* ---- per-kernel-class timers (the reference's PETSc event report, alfi/driver.py:77-92) -------------------------- */
enum {
  ALFI_EV_PATCH_APPLY = 0,   /* PCPATCHApply: per-patch gather + dense GEMV into the staging buffer */
  ALFI_EV_PATCH_SCATTER = 1, /* PCPATCHScatter: dof-wise sum of the staged patch results */
  ALFI_EV_PATCH_FACTOR = 2,  /* PCPatchComputeOp + inversion */
  ALFI_EV_MATMULT = 3,       /* MatMult */
  ALFI_EV_BLAS1 = 4,         /* FGMRES orthogonalisation / updates */
  ALFI_EV_PROLONG = 5,       /* SchoeberlProlong */
  ALFI_EV_RESTRICT = 6,      /* SchoeberlRestrict */
  ALFI_EV_COARSE = 7,        /* coarse solve */
  ALFI_EV_COMM = 8,          /* halo pack/exchange/unpack and all-reduces (VecScatter / MPI_Allreduce in the reference) */
  ALFI_EV_COUNT = 9
};
/* on = 1: a hipEvent pair around every launch of the classes above; on = 2: PATCH_APPLY and COMM only (fewer event
 * records on launch-bound levels); on = 3: PATCH_APPLY only; 0: off */
int alfi_prof_enable(alfi_ctx* ctx, int on);
int alfi_prof_reset(alfi_ctx* ctx);
/* synchronises, then returns summed device time (ms) and launch count of one class since the last reset */
int alfi_prof_get(alfi_ctx* ctx, int ev, double* total_ms, int64_t* count);
/* same, restricted to launches issued on behalf of one level (level_id from alfi_level_id; -1 = all levels) */
int alfi_prof_get_level(alfi_ctx* ctx, int ev, int level_id, double* total_ms, int64_t* count);

/* ---- mesh-partition parallelism: one ctx per GPU / process --------------------------------------------------------- */
/* The reference runs one MPI rank per mesh partition (alfi/solver.py:604-605, alfi/relaxation.py:120-121) and PETSc
 * performs the ghost exchanges (VecScatter forward / reverse-add around PCApply_PATCH and MatMult) and the reductions of
 * the Gram-Schmidt step (MPI_Allreduce) [3P].  Here every exchange point of a smoother, transfer or cycle is served by
 * one of two transports:
 *
 * (1) NATIVE (the product path): the ctx owns an RCCL communicator and exchanges by itself, stream-ordered on its own
 *     stream: one group of ncclSend / ncclRecv pairs with the level's actual neighbours (alfi_level_set_neighbours; a box
 *     partition has <= 7 of them on a node = one xGMI link each) per halo exchange, one ncclAllReduce of <= 33 doubles
 *     per reduction.  The host program only distributes the 128-byte id once (any channel: MPI_Bcast in the reference's
 *     world, torch.distributed / a file here): alfi_comm_unique_id on one rank, alfi_ctx_comm_init on every rank
 *     (collective).  librccl is resolved at run time (dlopen). */
#define ALFI_COMM_ID_BYTES 128
int alfi_comm_unique_id(void* id_out, int64_t len);                                  /* ncclGetUniqueId */
int alfi_ctx_comm_init(alfi_ctx* ctx, const void* id, int rank, int nranks);        /* ncclCommInitRank on the ctx's device */
/* TEST HOOK: allow a rank to name itself as a neighbour (a grouped ncclSend + ncclRecv to one's own rank), so that a
 * one-GPU box exercises the real point-to-point path with non-empty halos */
int alfi_ctx_comm_allow_self(alfi_ctx* ctx, int on);
int alfi_ctx_comm_size(alfi_ctx* ctx, int* rank, int* nranks);
/* exchange points issued since the last reset: halo exchanges (forward and reverse-add), all-reduces, and the doubles this
 * rank sent in them (either transport).  What the reference spends in PetscSF scatters and MPI_Allreduce [3P]. */
int alfi_ctx_comm_stats(alfi_ctx* ctx, int64_t* halo_exchanges, int64_t* allreduces, int64_t* doubles_sent, int reset);
int alfi_ctx_comm_destroy(alfi_ctx* ctx);                                            /* also done by alfi_ctx_destroy */
/* (2) CALLBACK (test transport: CPU-staged exchanges between ranks sharing one GPU, gloo): the library packs / unpacks
 *     halo buffers and calls `fn` at every exchange point; the host program performs the exchange stream-ordered on the
 *     ctx's stream and returns 0.  ops:
 *   ALFI_COMM_ALLREDUCE: sum dred[offset .. offset+count) over all ranks, in place;
 *   ALFI_COMM_HALO_FWD : level `level_id`: every owner's send buffer -> the ghosts' receive buffers;
 *   ALFI_COMM_HALO_REV : the reverse route: receive buffers (ghost contributions) -> owners' send buffers;
 *   ALFI_COMM_HALO_FWD_BEGIN / _END: the forward exchange split in two, so that the library can launch work that needs
 *                        no ghost value in between (only used on levels prepared with alfi_level_set_overlap): BEGIN
 *                        starts the exchange without making the ctx stream wait for it, END makes the stream wait;
 *   ALFI_COMM_HALO_REV_BEGIN / _END: the same for the reverse route.
 * Every rank of the group reaches every call (the exchanges are collective). */
enum { ALFI_COMM_ALLREDUCE = 0, ALFI_COMM_HALO_FWD = 1, ALFI_COMM_HALO_REV = 2, ALFI_COMM_HALO_FWD_BEGIN = 3,
       ALFI_COMM_HALO_FWD_END = 4, ALFI_COMM_HALO_REV_BEGIN = 5, ALFI_COMM_HALO_REV_END = 6,
       ALFI_COMM_HALO_SUM = 7 /* native transport only: never handed to a callback */ };
typedef int (*alfi_comm_fn)(void* user, int op, int level_id, int64_t offset, int64_t count);
/* dred: device buffer of dred_len >= 64 doubles owned by the caller (reduction scratch the callback all-reduces). */
int alfi_ctx_set_comm(alfi_ctx* ctx, alfi_comm_fn fn, void* user, double* dred, int64_t dred_len);

/* ---- level operator: PETSc MatMult on the BAIJ level matrix [3P], alfi/solver.py:512 ----------------------------- */
/* Block-CSR, bs x bs row-major blocks (bs = 2 or 3), nbrows block rows; bc_dofs: Dirichlet dofs of the level
 * (their rows/columns of the operator are expected to be the identity, as firedrake.assemble(a, bcs) gives [3P]).
 * bvals_host == NULL: the sparsity only; the values start as zeros and are formed on the device by alfi_level_set_assembly +
 * alfi_level_assemble (the patch factors and the coarse inverse are then asked for when the first cycle runs). */
int alfi_level_create(alfi_ctx* ctx, int64_t nbrows, int bs, const int32_t* browptr_host, const int32_t* bcolidx_host,
                      const double* bvals_host, const int32_t* bc_dofs_host, int64_t nbc, alfi_level** out);
int alfi_level_destroy(alfi_level* lvl);
/* Partitioned level: the first nb_owned block rows (nodes) are owned by this rank, the remaining nb_ghost are ghost
 * copies of nodes owned elsewhere (local numbering: owned first).  Vector kernels, reductions and the SpMV then run on
 * the owned prefix; ghost slots of any vector handed to the library are scratch.  distributed != 0: the level's
 * smoother and SpMV exchange halos and all-reduce (needs alfi_ctx_set_comm); distributed == 0: a level owned by one
 * rank whose halo is only used by the transfer to a distributed finer level.  send_nodes_host: owned nodes other ranks
 * hold as ghosts, grouped by destination rank (the layout of d_sendbuf: nsend * bs doubles); d_recvbuf: nb_ghost * bs
 * doubles in ghost order (ghosts grouped by owner).  Both buffers are device memory owned by the caller (callback
 * transport, which exchanges them), or both NULL: the library allocates them (native transport). */
int alfi_level_set_partition(alfi_level* lvl, int64_t nb_owned, int distributed, int64_t nsend,
                             const int32_t* send_nodes_host, double* d_sendbuf, double* d_recvbuf, int64_t nb_ghost);
/* Native transport: the ranks this level exchanges with (ascending, without this rank) and how many nodes of the send
 * list go to / how many ghost nodes come from each; the counts must add up to nsend and nb_ghost.  A rank with no
 * neighbour on a level passes nnbr = 0. */
int alfi_level_set_neighbours(alfi_level* lvl, int nnbr, const int32_t* ranks_host, const int64_t* send_nodes_host,
                              const int64_t* recv_nodes_host);
/* Native transport: plan of the merged reverse-add + forward exchange of the smoother ("sum exchange").  A node held by
 * several ranks (its owner and the ranks keeping a ghost copy) gets the SUM of the holders' values on every holder in ONE
 * exchange: each rank sends its value of every shared node to every other holder; all add the contributions in the same
 * (ascending rank) order, so owner and copies agree bitwise.  The smoother then applies z = M^-1 v with one exchange
 * after the local patch solves and the following product A z needs no forward exchange: 2 halo exchanges per FGMRES
 * iteration instead of 3.  ranks_host: ascending neighbour ranks; counts_host[i]: shared nodes with neighbour i (the same
 * number goes out and comes in); send_nodes_host: local node of every outgoing entry, neighbour by neighbour; the nshared
 * shared local nodes sum_nodes_host with CSR sum_ptr_host / sum_src_host: position of each contribution in the receive
 * buffer (neighbour segments in the order of ranks_host), -1 = this rank's own value. */
int alfi_level_set_sum_exchange(alfi_level* lvl, int nnbr, const int32_t* ranks_host, const int64_t* counts_host,
                                const int32_t* send_nodes_host, int64_t nshared, const int32_t* sum_nodes_host,
                                const int32_t* sum_ptr_host, const int32_t* sum_src_host);
/* Communication / computation overlap on a distributed level (after alfi_level_set_partition and alfi_patches_set): the
 * caller numbered the owned nodes so that the first nb_interior have operator rows without ghost columns, and ordered the
 * patches so that the first npatch_interior hold no ghost dof.  SpMV and patch apply then run those parts between
 * ALFI_COMM_HALO_FWD_BEGIN and _END.  Both claims are verified. */
int alfi_level_set_overlap(alfi_level* lvl, int64_t nb_interior, int64_t npatch_interior);
/* new Newton step / new Reynolds number: same sparsity, new values (PatchPC.update -> PCSetUp_PATCH [3P]). */
int alfi_level_update_values(alfi_level* lvl, const double* bvals_host);
/* Operator refresh on the device (PatchPC.update: `precompute_element_tensors` + `save_operators`, alfi/solver.py:320, 325;
 * PETSc event PCPatchComputeOp, alfi/driver.py:80).  alfi_level_set_assembly hands over, once, what does not change
 * between Newton steps: the cells (nodes, gradients of the barycentric coordinates (ncell, d+1, d), volumes), the
 * reference-cell tensors of the nodal element -- S (nloc, nloc, d+1, d+1) = avg d_i phi_a d_j phi_b, bI (nloc, d+1) =
 * avg d_i phi_a, T1 (nloc, d+1, nloc, nloc) = avg phi_k d_i phi_b phi_a (index order k, i, b, a) --, whether the grad-div
 * term is the full one gamma (div u, div v) of the Scott-Vogelius forms (alfi/solver.py:609-619; full_div != 0) or the
 * cell-averaged one of the P0-pressure forms (solver.py:565-568), and the contributor lists (cptr (nnzb+1), ccell,
 * cba = b*nloc+a per pair; fixed order).
 * alfi_level_assemble then writes  A = nu K + gamma D + adv N(state)  into the level's operator (the linearisation of
 * alfi/solver.py:565-568 about the DEVICE-resident nodal field `d_state`), Dirichlet rows / columns as identity when
 * apply_bc != 0: every cell forms its element matrix, every block of the operator sums its contributors in list order
 * (bitwise reproducible); the patches must be factored again afterwards (alfi_patches_factor).
 * PARTITIONED levels (alfi_level_set_partition first; one rank per mesh partition re-assembling its own patch operators,
 * alfi/solver.py:604-605): pass the cells that touch a LOCAL node (owned or ghost); cell_nodes index the level's local nodes
 * and, beyond them (indices >= nbrows), the cells' remaining nodes in any fixed order -- the state vector then holds
 * alfi_level_assembly_state_size entries, local nodes first --; the contributor lists hold only the pairs whose two nodes
 * are local (at most ncell * nloc^2).  No exchange is involved: every rank reads its own copy of the state. */
int alfi_level_set_assembly(alfi_level* lvl, int64_t ncell, int nloc, const int32_t* cell_nodes, const double* grad,
                            const double* vol, const double* S, const double* bI, const double* T1, int full_div,
                            const int64_t* cptr, const int32_t* ccell, const uint16_t* cba);
int alfi_level_assemble(alfi_level* lvl, double nu, double gamma, double adv, const double* d_state, int apply_bc);
/* bytes of element blocks the refresh may hold at once (default 24 GiB; beyond, the cells are taken in batches) */
int alfi_ctx_set_assembly_scratch(alfi_ctx* ctx, int64_t max_bytes);
/* partitioned levels: Dirichlet dofs among ALL local dofs (owned and ghost) for the refresh's identity rows / columns */
int alfi_level_set_assembly_bc(alfi_level* lvl, const int32_t* bc_dofs_host, int64_t nbc);
int alfi_level_assembly_state_size(alfi_level* lvl, int64_t* n);   /* doubles d_state must hold (= n of the level, unpartitioned) */
/* y = (nu K + gamma D + adv N(state)) x, no boundary conditions, MATRIX-FREE (every cell multiplies its element matrix with
 * its entries of x; no value array is formed): the level's operator and its patch factors stay valid.  The nonlinear
 * residual F_u of alfi/solver.py:565-568 is this product with x = state and HALF the advection weight
 * (N(u) u = 2 (u . grad) u).  dx: alfi_level_assembly_state_size entries (like d_state); dy: the level's rows. */
int alfi_level_assemble_mult(alfi_level* lvl, double nu, double gamma, double adv, const double* d_state, const double* dx,
                             double* dy);
/* SUPG stabilisation on the device (alfi/stabilisation.py:47-97 with the Shakib coefficient, alfi/solver.py:204-234; the
 * reference's production option `--stabilisation-type supg`).  alfi_level_set_supg (after alfi_level_set_assembly) hands over the
 * quadrature tables of the element -- weights wq (nq, summing to 1), phi (nq, nloc), dphi (nq, nloc, d+1), d2phi (nq, nloc, d+1,
 * d+1): basis and derivatives w.r.t. the barycentric coordinates -- and the cell sizes hcell (ncell; Firedrake's CellSize).
 * alfi_level_assemble_supg is the refresh of a stabilised run in one pass: A = nu K + gamma D + adv N(state) + the Newton
 * linearisation of `weight * beta * (Lu, (u . grad) v)` about the state, then the boundary conditions.  alfi_level_supg adds that
 * linearisation to the operator as it stands (add_to_operator != 0; between alfi_level_assemble(..., apply_bc = 0) and
 * alfi_level_apply_bc) and / or the residual contribution of the term to d_F (the level's n doubles, may be NULL).
 * alfi_level_apply_bc turns the Dirichlet rows / columns of the operator into identity. */
int alfi_level_set_supg(alfi_level* lvl, int nq, const double* wq, const double* phi, const double* dphi, const double* d2phi,
                        const double* hcell);
int alfi_level_assemble_supg(alfi_level* lvl, double nu, double gamma, double adv, const double* d_state, double weight,
                             double magic, int apply_bc);
int alfi_level_supg(alfi_level* lvl, double nu, double weight, double magic, const double* d_state, int add_to_operator,
                    double* d_F);
int alfi_level_apply_bc(alfi_level* lvl);
/* the operator values in the host layout (nnzb, bs, bs) -- diagnostics / tests */
int alfi_level_get_values(alfi_level* lvl, double* bvals_host);
int alfi_level_size(alfi_level* lvl, int64_t* n);
int alfi_level_id(alfi_level* lvl, int* id); /* creation-order id within the ctx, used by alfi_prof_get_level */
int alfi_spmv(alfi_level* lvl, const double* dx, double* dy);                        /* y = A x      */
int alfi_residual(alfi_level* lvl, const double* db, const double* dx, double* dr); /* r = b - A x  */

/* ---- patch smoother: firedrake.PatchPC -> PETSc PCPATCH [3P], alfi/solver.py:318-328, 599-602 -------------------- */
/* Patches as produced by a patch-construction callable (alfi/relaxation.py:110-150) after PCPATCH's dof mapping:
 * patch p owns dofs patch_dofs[patch_ptr[p] .. patch_ptr[p+1]), ascending, Dirichlet dofs excluded, sizes <= 4096
 * (levels whose largest patch exceeds 160 dofs -- macro stars -- use the blocked matrix-core inversion and one workgroup
 * per patch in the apply). */
int alfi_patches_set(alfi_level* lvl, int64_t npatch, const int64_t* patch_ptr_host, const int32_t* patch_dofs_host);
/* PCSetUp_PATCH with save_operators + dense_inverse (solver.py:320, 602): A_p = A[dofs_p, dofs_p], store inv(A_p). */
int alfi_patches_factor(alfi_level* lvl);
/* Condensed patch factors (optional, between alfi_patches_set and alfi_patches_factor).  group[q] (one label per entry of
 * patch_dofs) >= 0: the entry belongs to that group of its patch; -1: it belongs to the patch's skeleton.  A group must be
 * coupled to the rest of its patch only through skeleton entries (verified against the operator's sparsity; violated:
 * ALFI_E_ARG) and hold <= 64 entries coupled to <= 64 skeleton entries; patches must consist of whole nodes.  The library
 * then stores inv(A_p) EXACTLY as a block factorisation -- X_g = inv(A_gg), B_g = A[S_g, g], W_g = X_g A[g, S_g] per group
 * and inv(A_SS - sum_g B_g W_g) -- instead of the dense inverse: the same result in exact arithmetic and to rounding,
 * sum_g (m_g^2 + 2 m_g s_g) + s^2 doubles instead of n_p^2, both in memory and in the bytes an apply streams.  Made for
 * the macro-star patches of the Scott-Vogelius discretisation (alfi/relaxation.py:163-177 on the Alfeld-split meshes of
 * alfi/bary.py, where the reference itself keeps SPARSE patch factors, alfi/solver.py:655-659): the dofs interior to one
 * macro cell form a group; 7 x less for [P3]^3.  group == NULL: back to dense inverses.  alfi_patches_factor_bytes: the bytes
 * the factors of the level occupy (= the bytes one additive apply reads). */
int alfi_patches_set_groups(alfi_level* lvl, const int32_t* group_host);
int alfi_patches_factor_bytes(alfi_level* lvl, int64_t* bytes);
/* Every alfi_patches_factor ends with a residual probe of every stored inverse, rho_p = || A_p (X_p e) - e ||_inf with a
 * fixed +-1 vector e, and re-inverts the patches with rho_p > 1e-6 (ALFI_PATCH_CHECK_TOL) -- or with a zero pivot -- by
 * LU with partial pivoting and triangular sweeps (the reference factors with pivoted LAPACK / UMFPACK LU, solver.py:599-602,
 * 655-659; the fast kernels do not pivot); for condensed factors the Schur complement of the flagged patch is formed again
 * and re-inverted the same way, in place.  ALFI_E_SINGULAR if a patch still fails beyond ALFI_PATCH_CHECK_FAIL (1000 x the
 * tolerance) afterwards.  The group matrices of a condensed factor are stored with even leading dimensions
 * (alfi_patches_factor_bytes counts the padding).  This reports, for the last
 * factorisation: the worst residual of the fast inversion, how many patches were flagged and repaired, and the worst
 * residual after the repair (any pointer may be NULL; -1 = probe disabled with ALFI_PATCH_CHECK=0). */
int alfi_patches_check(alfi_level* lvl, double* worst_residual, int64_t* flagged, int64_t* repaired, double* worst_after);
/* PCApply_PATCH, additive, no partition of unity (solver.py:321-322): y = sum_p R_p^T inv(A_p) R_p x; y[bc] = x[bc].
 * x is not modified.  Deterministic (no atomics): patch results are staged and summed dof-wise in a fixed order. */
int alfi_patch_apply(alfi_level* lvl, const double* dx, double* dy);
/* patch_pc_patch_partition_of_unity (solver.py:321; the reference always passes False): on != 0 weights the additive sum
 * of every dof by 1 / (number of patches holding it), as PCPATCH does with its dof_weights [3P]. */
int alfi_patches_set_partition_of_unity(alfi_level* lvl, int on);
/* PCPATCH local_type multiplicative (solver.py:322-324, 332-335): visit the patches in the order `iterset_host` (nit
 * positions; the concatenation of one coordinate-sorted permutation per '|' sweep of the constructor's sort_order,
 * relaxation.py:139-150, so a patch may appear more than once), each solve seeing the residual left by all earlier
 * ones; symmetrise != 0 appends the same sequence reversed (patch_pc_patch_symmetrise_sweep).  The library orders
 * the positions into dependency wavefronts (two patches are independent when no operator entry couples them), which
 * reproduces the sequential sweep exactly.  Afterwards alfi_patch_apply and alfi_smooth_fgmres on this level use the
 * multiplicative sweep; nit == 0 switches back to additive.  Patches must consist of whole nodes (all bs components); a
 * wave sweeps a patch of up to 64 nodes / 160 dofs, a workgroup the larger ones (macro stars, solver.py:339-342).  On a partitioned level every rank sweeps over its own patches (local Gauss-Seidel,
 * additive between ranks, ghost contributions reverse-added at the end), as PCPATCH does under MPI. */
int alfi_patches_set_multiplicative(alfi_level* lvl, int64_t nit, const int64_t* iterset_host, int symmetrise);
/* number of dependency wavefronts of one sweep (0 = additive) */
int alfi_patches_multiplicative_levels(alfi_level* lvl, int64_t* nwave);
/* sum_p n_p^2 (doubles held as inverses) and sum_p n_p, for roofline accounting */
int alfi_patches_stats(alfi_level* lvl, int64_t* npatch, int64_t* sum_n, int64_t* sum_n2);
/* debugging / parity tests: copy the dense inverse of patch p (row-major n_p x n_p) to the host */
int alfi_patch_get_inverse(alfi_level* lvl, int64_t p, double* out_host);

/* ---- level smoother: PETSc KSPFGMRES, k iterations, convergence_test skip [3P], alfi/solver.py:314-317 ----------- */
/* Right-preconditioned FGMRES(k) with classical Gram-Schmidt, preconditioned by alfi_patch_apply; x is updated in
 * place; nonzero_guess = 0 treats the incoming x as zero.  Entirely device-resident (no host synchronisation). */
int alfi_smooth_fgmres(alfi_level* lvl, int k, const double* db, double* dx, int nonzero_guess);

/* ---- coarse solve: firedrake.AssembledPC + LU [3P], alfi/solver.py:369-378 ---------------------------------------- */
/* alfi_coarse_factor: the level operator's dense inverse (n x n doubles on the device, n <= 131072) by the library's own
 * blocked Gauss-Jordan on the FP64 matrix cores (64-wide pivot blocks, rank-64 trailing updates as v_mfma_f64_16x16x4,
 * two Newton-Schulz polish steps) -- no library GEMM, no host LAPACK -- followed by the residual probe || A X e - e ||
 * (ALFI_E_SINGULAR above 1e-5; alfi_coarse_residual reports it).  The level must be owned by one rank.
 * alfi_coarse_set_inverse: an inverse computed by the caller instead (row-major n x n, host or device memory).
 * The solve is a device GEMV either way. */
int alfi_coarse_factor(alfi_level* lvl);
/* alfi_coarse_factor_sparse: the same exact coarse solve for operators beyond the dense inverse -- a multifrontal block
 * L D U on a nested-dissection ordering (what the reference obtains from MUMPS / SuperLU_DIST, alfi/solver.py:369-378),
 * dense fronts on the FP64 matrix cores, plus one step of iterative refinement per solve.  coords: nbrows x dim node
 * coordinates in host memory for the geometric bisection, or NULL (level sets of the operator's graph); leaf_nodes: nodes
 * per leaf subdomain (<= 0: default 64).  Same residual probe as alfi_coarse_factor.
 * alfi_coarse_factor_bytes: device bytes of the stored factors (dense: 8 n^2). */
int alfi_coarse_factor_sparse(alfi_level* lvl, const double* node_coords, int dim, int leaf_nodes);
int alfi_coarse_factor_bytes(alfi_level* lvl, int64_t* bytes);
int alfi_coarse_residual(alfi_level* lvl, double* worst);
int alfi_coarse_set_inverse(alfi_level* lvl, const double* inv, int inv_is_device);
int alfi_coarse_solve(alfi_level* lvl, const double* db, double* dx);

/* ---- grid transfer: alfi/transfer.py:91-356 (PkP0SchoeberlTransfer), alfi/bubble.py (standard transfer) ----------- */
typedef struct {
  int64_t nbrows, nbcols;
  const int32_t* rowptr; /* host */
  const int32_t* colidx; /* host */
  const double* vals;    /* host, (nnzb, bs, bs) */
} alfi_bsr_host;
/* P: standard prolongation (fine x coarse; bubble-corrected for 3-D P1+FB, transfer.py:334-356), PT its transpose,
 * PT_plain: transpose of the plain nodal interpolation used when restriction is not robust (solver.py:595; may be
 * NULL = same as PT).  D_I: rows of the cell-averaged grad-div matrix (gamma = 1) for the coarse-cell interior dofs
 * in block order; D_IT its transpose.  blk_dofs: (nblk, m) interior dofs per coarse cell (transfer.py:13-46; m <= 4096: beyond 32 -- the coarse
 * MACRO cells of the Scott-Vogelius transfer, transfer.py:49-88 -- the blocks are solved with the patch kernels, beyond
 * 160 with the large-patch ones);
 * K_II, D_II: (nblk, m, m) dense interior blocks of (2 sym grad u, grad v) and (cell_avg div u, div v). */
int alfi_transfer_create(alfi_ctx* ctx, alfi_level* coarse, alfi_level* fine, const alfi_bsr_host* P,
                         const alfi_bsr_host* PT, const alfi_bsr_host* PT_plain, const alfi_bsr_host* D_I,
                         const alfi_bsr_host* D_IT, int64_t nblk, int m, const int32_t* blk_dofs_host,
                         const double* K_II_host, const double* D_II_host, alfi_transfer** out);
int alfi_transfer_destroy(alfi_transfer* tr);
/* (re)build the interior solves for new (nu, gamma): AutoSchoeberlTransfer.rebuild, transfer.py:173-184, 238-244 */
int alfi_transfer_update(alfi_transfer* tr, double nu, double gamma);
/* debugging / parity tests: the stored inverse of interior block blk (row-major m x m) copied to the host.  (Blocks with
 * m > 32 are applied with one step of iterative refinement on top of this inverse.) */
int alfi_transfer_get_block_inverse(alfi_transfer* tr, int64_t blk, double* out_host);
/* prolong: fine = (I - E_I inv(A_II) E_I^T gamma D) P coarse   (transfer.py:246-259); fine Dirichlet dofs zeroed */
int alfi_prolong(alfi_transfer* tr, const double* dxc, double* dxf);
/* restrict: robust = 1: coarse = P^T (I - gamma D E_I inv(A_II) E_I^T) fine (transfer.py:261-275);
 *           robust = 0: coarse = P_plain^T fine (firedrake.restrict).  Coarse Dirichlet dofs zeroed; fine untouched. */
int alfi_restrict(alfi_transfer* tr, const double* drf, double* drc, int robust);

/* inject: coarse = nodal values of fine at the coarse nodes (firedrake.inject, the third entry of the transfer tuple,
 * alfi/solver.py:595; moves the Newton state to the coarse levels, alfi/stabilisation.py:41).  fine_node_host[i] = fine
 * node coinciding with coarse node i (all velocity dofs are point evaluations).  Serial levels only. */
int alfi_transfer_set_injection(alfi_transfer* tr, const int32_t* fine_node_host);
int alfi_inject(alfi_transfer* tr, const double* dxf, double* dxc);

/* ---- multigrid cycle: PETSc PCMG [3P], alfi/solver.py:359-379 ------------------------------------------------------ */
/* levels[0] = coarsest (needs a coarse inverse / factorisation), levels[l] l >= 1 need factored patches -- both by the time a
 * cycle runs (alfi_mg_vcycle / _fcycle return ALFI_E_STATE otherwise), not at creation; transfers[l-1] links
 * level l-1 and l.  k = smoother iterations (solver.py:309-310), robust_restriction = the --restriction flag. */
int alfi_mg_create(alfi_ctx* ctx, int nlevels, alfi_level** levels, alfi_transfer** transfers, int k,
                   int robust_restriction, alfi_mg** out);
int alfi_mg_destroy(alfi_mg* mg);
/* one multiplicative V-cycle on the finest level: x <- V(b, x) (PCMGMCycle_Private) */
int alfi_mg_vcycle(alfi_mg* mg, const double* db, double* dx);
/* pc_mg_type full (solver.py:366): x <- F(b), x need not be initialised (PCMGFCycle_Private) */
int alfi_mg_fcycle(alfi_mg* mg, const double* db, double* dx);
/* on != 0: alfi_mg_vcycle / alfi_mg_fcycle replay their launch sequence as a hipGraph from the second call with the same
 * (db, dx) pair on (the small levels of a hierarchy are launch-bound).  New operator values in place need no new capture;
 * new (nu, gamma), patches, coarse inverse or smoother length re-capture.  Runs with profiling on or on partitioned
 * levels stay eager. */
int alfi_ctx_set_graph(alfi_ctx* ctx, int on);

/* ---- outer solve around the path: alfi/solver.py:386-422, 463-499 ------------------------------------------------------- */
/* The linear system of one Newton step, [A B^T; B 0] [u; p] = [f; g], solved as the reference configures PETSc:
 * KSPFGMRES(restart), right-preconditioned by PCFIELDSPLIT Schur with the FULL factorisation (solver.py:407-411):
 *   y_u = MG(b_u);  y_p = S~^-1 (b_p - B y_u);  y_u = MG(b_u - B^T y_p),
 * fieldsplit_0 = one PCMG full cycle (alfi_mg_fcycle; solver.py:359-379), fieldsplit_1 = DGMassInv:
 * S~^-1 q = -(nu + gamma) M_p^-1 q (solver.py:15-38) with the diagonal P0 mass matrix M_p.  A is the finest level's
 * operator (with the augmented-Lagrangian term), B the discrete divergence (rows = pressure dofs, Dirichlet velocity
 * columns zero), BT its transpose; scalar CSR, host arrays.  remove_constant_nullspace: orthogonalise the pressure part
 * against constants after every preconditioner application (the nullspace alfi attaches for enclosed flows,
 * alfi/solver.py:267-272 / problem has_nullspace). */
typedef struct alfi_saddle alfi_saddle;
typedef struct {
  int64_t nrows, ncols;
  const int32_t* rowptr; /* host */
  const int32_t* colidx; /* host */
  const double* vals;    /* host */
} alfi_csr_host;
int alfi_saddle_create(alfi_mg* mg, const alfi_csr_host* B, const alfi_csr_host* BT, const double* mass_diag_host,
                       double nu, double gamma, int remove_constant_nullspace, alfi_saddle** out);
int alfi_saddle_destroy(alfi_saddle* s);
/* Discontinuous P_k pressure (Scott-Vogelius, solver.py:624-629): DGMassInv's ``Tensor(inner(u, v)*dx).inv`` (solver.py:24) is
 * block diagonal, one (k+1)(k+2)(k+3)/6 block per cell; pass it as scalar CSR (n_p x n_p).  Replaces mass_diag. */
int alfi_saddle_set_mass_inverse(alfi_saddle* s, const alfi_csr_host* Minv);
int alfi_saddle_update(alfi_saddle* s, double nu, double gamma);
/* db, dx: device vectors of n_u + n_p doubles (velocity dofs first).  Zero initial guess.  Convergence as KSP's default
 * test on the (recurrence) residual norm: ||r|| <= max(rtol ||b||, atol); max_it iterations at most (solver.py:404,
 * 463-499).  Returns the iteration count and the final true residual norm. */
int alfi_saddle_solve(alfi_saddle* s, const double* db, double* dx, double rtol, double atol, int max_it, int restart,
                      int* iterations, double* residual_norm);
/* Building blocks of the same solve on PARTITIONED levels (the host drives the outer Krylov loop: 2 full cycles per
 * iteration dwarf its few vector operations; alfi_amd/dist.py: DistSaddle).  Halo routes of a level set up with
 * alfi_level_set_partition: forward = owner -> ghost copies, reverse_add = ghost contributions summed onto their owners
 * (PETSc's VecScatter forward / reverse-add on the velocity DM).  No-ops on levels that are not distributed. */
int alfi_level_halo_forward(alfi_level* lvl, double* dv);
int alfi_level_halo_reverse_add(alfi_level* lvl, double* dv);
int alfi_level_halo_sum(alfi_level* lvl, double* dv);   /* the merged exchange of alfi_level_set_sum_exchange */
/* a scalar CSR matrix on the device (the rank's rows of the discrete divergence and its transpose) */
typedef struct alfi_csr alfi_csr;
int alfi_csr_create(alfi_ctx* ctx, const alfi_csr_host* M, alfi_csr** out);
/* inject on a NON-NESTED hierarchy (the barycentric refinements of the Scott-Vogelius pair, alfi/solver.py:641-652): the coarse
 * nodes are not fine nodes, the fine function is evaluated at them.  J: (coarse nodes x fine nodes) scalar CSR of basis values,
 * applied to every component; replaces the index map of alfi_transfer_set_injection for alfi_inject.  Serial levels only. */
int alfi_transfer_set_injection_matrix(alfi_transfer* tr, const alfi_csr_host* J);
int alfi_csr_destroy(alfi_csr* m);
/* mode 0: y = M x;  1: y = b - alpha M x;  2: y += M x;  3: y = alpha M x */
int alfi_csr_mult(alfi_csr* m, const double* dx, double* dy, const double* db, double alpha, int mode);
/* The Newton loop around the path keeps its state, update and residual in HBM (the reference's NonlinearVariationalSolver
 * keeps z as a distributed Function, alfi/solver.py:245-273): small vector operations on device vectors, stream-ordered;
 * only scalars come back.  alfi_saddle_dot: x . y of two (velocity | pressure) vectors of the outer solve, summed over the
 * ranks on a partitioned finest level (every rank gets the same value), fixed summation order.  alfi_level_zero_bc: dv[Dirichlet
 * dofs of the level] = 0 (bc.zero(F), solver.py:282-286).  alfi_vec_gather: dst[i] = src[idx[i]], bs doubles per index (idx a
 * DEVICE array).  alfi_transfer_stats: bytes every host <-> device copy of the library has moved (process-wide). */
int alfi_vec_axpy(alfi_ctx* ctx, double* dy, const double* dx, double a, int64_t n);   /* y += a x */
int alfi_vec_copy(alfi_ctx* ctx, double* dy, const double* dx, int64_t n);
int alfi_vec_gather(alfi_ctx* ctx, double* dst, const double* dsrc, const int32_t* d_idx, int64_t nidx, int bs);
int alfi_level_zero_bc(alfi_level* lvl, double* dv);
int alfi_saddle_dot(alfi_saddle* s, const double* dx, const double* dy, double* out_host);
int alfi_transfer_stats(int64_t* h2d_bytes, int64_t* d2h_bytes, int reset);
/* y = [A B^T; B 0] x and y = P^-1 x on device vectors (tests, monitors) */
int alfi_saddle_mult(alfi_saddle* s, const double* dx, double* dy);
int alfi_saddle_precond(alfi_saddle* s, const double* dx, double* dy);

#ifdef __cplusplus
}
#endif
#endif
