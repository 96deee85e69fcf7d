"""Mesh-partition data parallelism of the multigrid hot path: one process per GPU, halos over torch.distributed.

The reference distributes every level's mesh over MPI ranks with a vertex-star overlap (``distribution_parameters``,
alfi/solver.py:604-605) and builds one patch per *owned* vertex (``ownership_ranges``, alfi/relaxation.py:120-121);
PETSc then scatters ghost values before and adds ghost contributions after every patch apply, and MatMult scatters
ghost columns [3P].  The same decomposition here, sized for 8 MI355X on xGMI:

* nodes of a level are numbered along a Morton curve (fespace.py), so a contiguous index range is a box-like sub-domain:
  rank r **owns** the node range ``[splits[r], splits[r+1])`` (balanced by patch-inverse + operator bytes) and the
  patches seeded in it.  Its **ghosts** are every other node its owned rows, owned patches, coarse-cell transfer blocks
  or (as the coarse side of a transfer) its fine rows' prolongation stencils touch.  Local numbering = owned nodes
  first, then ghosts ascending (= grouped by owner), so all vector kernels run on the owned prefix.
* per level ONE halo plan, two exchanges: *forward* (owner -> ghost copies) and *reverse-add* (ghost contributions ->
  owner), both one ``all_to_all_single`` of surface-sized buffers (RCCL grouped send/recv over the xGMI links), plus an
  all-reduce of <= k+1 doubles per Gram-Schmidt step.  There is no other collective on the data path.
* levels below ``min_dofs`` live on rank 0 alone (the reference telescopes the coarse grid the same way,
  solver.py:354-358); the other ranks only hold the ghost copies the transfer to the first distributed level needs.

The cycle itself stays inside libalfi_hip.so: the library packs/unpacks halo buffers and calls back
(``alfi_ctx_set_comm``) at the exchange points; this module answers the callback with torch.distributed.  With the NCCL
(= RCCL) backend the buffers are exchanged device to device on the library's stream; with gloo (CPU tests, or several
ranks sharing one GPU) they are staged through the host.
"""
import ctypes

import numpy as np

from .problem import BSR

COMM_ALLREDUCE, COMM_HALO_FWD, COMM_HALO_REV, COMM_HALO_FWD_BEGIN, COMM_HALO_FWD_END = 0, 1, 2, 3, 4
COMM_HALO_REV_BEGIN, COMM_HALO_REV_END = 5, 6
RED_LEN = 64


# ---------------------------------------------------------------------------------------------------------------------
# partition (host, NumPy; identical on every rank because every rank generates the same global hierarchy)
# ---------------------------------------------------------------------------------------------------------------------
def level_weights(L):
    """HBM bytes a node causes per smoother iteration: its operator row and, for patch seeds, the patch inverse."""
    A = L.A
    w = np.diff(A.rowptr).astype(np.float64) * (8.0 * A.bs * A.bs + 4.0) + 16.0 * A.bs
    if getattr(L, "patch_ptr", None) is not None and L.level > 0:
        npd = np.diff(L.patch_ptr).astype(np.float64)
        w[L.V.vertex_nodes[L.patch_seeds]] += 8.0 * npd * npd
    return w


# Time of a node of a SINGLE-OWNER level relative to its bytes: those levels are small (launch latency, not bandwidth) and their
# owner also solves the coarse grid.  Measured with every rank of config 4's 8-way partition alone on the GPU (scripts/
# solo_rank_time.py, profiles/r05_solo_rank_time_cfg4_8ranks.txt): rank 0 spends 4.2 ms per V-cycle on levels 0-1 next to 21.0 ms
# of partitioned work per rank, 1.8 x what the bytes of those levels would cost at the rate of the partitioned ones.
SINGLE_OWNER_TIME_FACTOR = 1.8


def choose_splits(levels, world, min_dofs=400000, balance_single_owner=True, full_cycle=False):
    """splits[l]: int64 (world+1) node split points of level l.  Levels with fewer than ``min_dofs`` dofs -- and, so that
    the coarse solve needs no exchange, always level 0 -- belong to rank 0 entirely.  balance_single_owner: rank 0 gets a
    smaller share of the partitioned levels, by the work it does alone on the single-owner ones (round 5: the equal split left
    rank 0 at 25.3 ms of kernels per config-4 V-cycle against 21.0 ms on the other seven ranks).  full_cycle: the work is that of
    a full cycle (``pc_mg_type full``, what the outer solves apply, alfi/solver.py:366): level l of nl is visited nl - l times
    instead of once, which weighs the single-owner levels more."""
    single = [world == 1 or L.level == 0 or L.n < min_dofs for L in levels]
    visits = [(len(levels) - L.level) if full_cycle else 1 for L in levels]
    totals = [v * float(level_weights(L).sum()) for L, v in zip(levels, visits)]
    w_single = SINGLE_OWNER_TIME_FACTOR * sum(t for t, s_ in zip(totals, single) if s_)
    w_dist = sum(t for t, s_ in zip(totals, single) if not s_)
    # rank 0's share f0 of every partitioned level: f0 w_dist + w_single = (w_dist + w_single) / world, at least half a share
    f0 = 1.0 / world
    if balance_single_owner and world > 1 and w_dist > 0.0:
        f0 = max(0.5 / world, (w_dist + w_single) / (world * w_dist) - w_single / w_dist)
    fractions = np.concatenate([[f0], f0 + (1.0 - f0) * np.arange(1, world - 1) / (world - 1)]) if world > 1 else np.zeros(0)
    out = []
    for L, one in zip(levels, single):
        nb = L.A.nbrows
        if one:
            out.append(np.array([0] + [nb] * world, dtype=np.int64))
            continue
        c = np.cumsum(level_weights(L))
        s = np.searchsorted(c, c[-1] * fractions, side="left") + 1
        s = np.maximum.accumulate(np.concatenate([[0], np.minimum(s, nb), [nb]]).astype(np.int64))
        if (np.diff(s) <= 0).any():
            raise ValueError("level %d (%d nodes) is too small to split over %d ranks" % (L.level, nb, world))
        out.append(s)
    # a single-owner level above a distributed one would serialise the fine work: not produced by the rule above
    # because dofs grow with the level, but guard against odd hierarchies
    for l in range(1, len(out)):
        if out[l][1] == levels[l].A.nbrows and out[l - 1][1] != levels[l - 1].A.nbrows:
            raise ValueError("level %d is single-owner above a distributed level" % l)
    return out


def overlap_decision(splits, bs, distributed, overlap, overlap_min_dofs):
    """Whether a level hides its halo exchanges behind interior work (alfi_level_set_overlap).  The decision is COLLECTIVE:
    the overlapped smoother iteration exchanges forward / reverse / forward, the plain one forward / sum, so two neighbours
    that decide differently pair send/recv groups of different counts and directions (an RCCL hang or wrong halos).  It is
    therefore taken from the split points -- identical on every rank -- by the SMALLEST owned share of the level, not from
    the calling rank's own share."""
    if not (distributed and overlap):
        return False
    shares = np.diff(np.asarray(splits, dtype=np.int64))
    shares = shares[shares > 0]
    return bool(shares.size > 0 and int(shares.min()) * bs >= overlap_min_dofs)


# The overlap rule (scripts/overlap_rule.py measures its two constants on one GPU: profiles/r05_overlap_rule.txt).
# The overlapped smoother iteration splits every patch apply / product into interior | boundary launches around asynchronous
# begin / end pairs on RCCL's own stream (two cross-stream waits each) and needs three exchanges where the plain one needs two:
# a FIXED cost per iteration, OVERLAP_FIXED_US.  What it can hide is the time one halo exchange takes on the critical path,
#     t_exchange = EXCHANGE_LATENCY_US + bytes to the busiest neighbour / link bandwidth
# (xGMI is point to point: the messages to the <= 7 neighbours of a box partition travel over separate links at the same time).
# An exchange hides only behind work that needs no ghost value: the patch solves / rows of the rank's interior, t_interior.
# The plain iteration exposes two exchanges, so overlap pays when 2 min(t_exchange, t_interior) exceeds the fixed cost.
OVERLAP_FIXED_US = 65.7           # per smoother iteration, measured: (overlapped - plain) iteration with a 1-rank RCCL group
EXCHANGE_LATENCY_US = 17.5        # one grouped ncclSend / ncclRecv exchange of a few KB incl. pack / unpack kernels, measured
XGMI_LINK_GBPS = 0.8 * 153.0      # MI355X: 153 GB/s per link and direction, 80 % reachable (MI355X_MICROARCH.md)
HBM_STREAM_GBPS = 6000.0          # what the patch apply streams its inverses at


def overlap_rule(max_neighbour_bytes, min_interior_bytes):
    """True when hiding the exchanges of a smoother iteration behind interior work is expected to pay:
    2 min(t_exchange, t_interior) > OVERLAP_FIXED_US.  ``max_neighbour_bytes``: the largest message of the level's halo
    exchange; ``min_interior_bytes``: the patch-inverse bytes of the interior patches (what the interior launch streams);
    maximum / minimum over ALL ranks -- the decision must be collective, see overlap_decision."""
    t_exchange = EXCHANGE_LATENCY_US + max_neighbour_bytes / (XGMI_LINK_GBPS * 1e3)
    t_interior = min_interior_bytes / (HBM_STREAM_GBPS * 1e3)
    return 2.0 * min(t_exchange, t_interior) > OVERLAP_FIXED_US


class LevelPart(object):
    """One rank's view of one level: owned range, ghosts, local numbering, halo plan."""

    def __init__(self, level, bs, splits, rank, ghosts, force_distributed=False, boundary_mask=None):
        """boundary_mask: (nb_own,) bool over the owned global range, True for owned nodes whose operator row reaches a
        ghost.  Local numbering = owned interior nodes, owned boundary nodes, ghosts: the rows / patches that need no
        ghost value come first, so the library can work on them while the forward halo is in flight."""
        self.level, self.bs, self.splits, self.rank = level, bs, np.asarray(splits, dtype=np.int64), rank
        self.lo, self.hi = int(splits[rank]), int(splits[rank + 1])
        self.nb_own = self.hi - self.lo
        if boundary_mask is None:
            boundary_mask = np.zeros(self.nb_own, dtype=bool)
        own = np.arange(self.lo, self.hi, dtype=np.int64)
        self.own_nodes = np.concatenate([own[~boundary_mask], own[boundary_mask]])     # global ids in local order
        self.nb_int = int(np.count_nonzero(~boundary_mask))
        self.own_perm = np.empty(self.nb_own, dtype=np.int64)                           # global - lo -> local index
        self.own_perm[self.own_nodes - self.lo] = np.arange(self.nb_own)
        self.ghosts = np.asarray(ghosts, dtype=np.int64)              # global ids, ascending => grouped by owner
        self.nb_ghost = self.ghosts.shape[0]
        self.nb_loc = self.nb_own + self.nb_ghost
        # force_distributed: run the exchange points even when one rank owns everything (a 1-rank NCCL group on the
        # single-GPU test box exercises the RCCL code path: empty halos, 1-rank all-reduces)
        self.distributed = bool(np.count_nonzero(np.diff(self.splits)) > 1) or bool(force_distributed)
        owner = np.searchsorted(self.splits, self.ghosts, side="right") - 1
        world = len(splits) - 1
        self.recv_counts = np.bincount(owner, minlength=world).astype(np.int64)   # ghost nodes per owner
        self.send_nodes = [np.zeros(0, dtype=np.int64)] * world                   # filled by set_send_lists
        self.send_counts = np.zeros(world, dtype=np.int64)

    @property
    def nodes(self):
        return np.concatenate([self.own_nodes, self.ghosts])

    def own_dofs(self):
        """Global dof numbers of the owned entries of a local vector, in local order."""
        return (self.own_nodes[:, None] * self.bs + np.arange(self.bs)).ravel()

    def ghosts_by_owner(self):
        off = np.concatenate([[0], np.cumsum(self.recv_counts)])
        return [self.ghosts[off[q]:off[q + 1]] for q in range(len(self.recv_counts))]

    def set_send_lists(self, wanted):
        """wanted[q]: global node ids rank q holds as ghosts of mine (ascending, q's ghost order)."""
        for q, w in enumerate(wanted):
            w = np.asarray(w, dtype=np.int64)
            assert w.size == 0 or (w.min() >= self.lo and w.max() < self.hi), "rank %d asked for nodes I do not own" % q
        self.send_nodes = [self.own_perm[np.asarray(w, dtype=np.int64) - self.lo] for w in wanted]
        self.send_counts = np.array([s.shape[0] for s in self.send_nodes], dtype=np.int64)

    def set_sum_plan(self, all_ghosts):
        """Plan of the merged reverse-add + forward exchange ("sum exchange"): every rank holding a node -- its owner and
        the ranks that keep a ghost copy -- sends its partial value to every other holder, and all of them add the
        contributions in ascending rank order, so that owner and copies end with bitwise the same total after ONE exchange
        (instead of ghost -> owner, then owner -> ghosts).  all_ghosts[q]: ascending global ghost ids of rank q."""
        world, me = len(self.splits) - 1, self.rank
        held = []                                        # S_q: nodes rank q holds and somebody else holds too
        for q in range(world):
            lo, hi = int(self.splits[q]), int(self.splits[q + 1])
            mine = [g[(g >= lo) & (g < hi)] for r, g in enumerate(all_ghosts) if r != q]
            iface = np.unique(np.concatenate(mine)) if mine else np.zeros(0, dtype=np.int64)
            held.append(np.union1d(np.asarray(all_ghosts[q], dtype=np.int64), iface))
        ranks, counts, send, rows, srcs, rnk = [], [], [], [], [], []
        off = 0
        for q in range(world):
            if q == me:
                continue
            common = np.intersect1d(held[me], held[q], assume_unique=True)
            if common.size == 0:
                continue
            loc = self.g2l(common)
            assert (loc >= 0).all()
            ranks.append(q)
            counts.append(common.size)
            send.append(loc)
            rows.append(loc)
            srcs.append(off + np.arange(common.size, dtype=np.int64))
            rnk.append(np.full(common.size, q, dtype=np.int64))
            off += common.size
        mine_loc = self.g2l(held[me])
        rows.append(mine_loc)
        srcs.append(np.full(mine_loc.size, -1, dtype=np.int64))
        rnk.append(np.full(mine_loc.size, me, dtype=np.int64))
        rows, srcs, rnk = np.concatenate(rows), np.concatenate(srcs), np.concatenate(rnk)
        order = np.lexsort((rnk, rows))                  # by node, inside a node by contributing rank
        rows, srcs = rows[order], srcs[order]
        nodes, start = np.unique(rows, return_index=True)
        self.sum_ranks = np.asarray(ranks, dtype=np.int32)
        self.sum_counts = np.asarray(counts, dtype=np.int64)
        self.sum_send_nodes = np.concatenate(send).astype(np.int32) if send else np.zeros(0, dtype=np.int32)
        self.sum_nodes = nodes.astype(np.int32)
        self.sum_ptr = np.concatenate([start, [rows.size]]).astype(np.int32)
        self.sum_src = srcs.astype(np.int32)

    @property
    def has_halo(self):
        return bool(self.recv_counts.sum() + self.send_counts.sum() > 0)

    def g2l(self, g):
        """Local index of global nodes (-1 where absent)."""
        g = np.asarray(g, dtype=np.int64)
        if g.size > 4096 and g.size * 8 > self.splits[-1]:
            # many lookups: one dense global -> local table (int32 per global node) beats the binary searches
            if getattr(self, "_g2l_table", None) is None:
                t = np.full(int(self.splits[-1]), -1, dtype=np.int32)
                t[self.own_nodes] = np.arange(self.nb_own, dtype=np.int32)
                t[self.ghosts] = self.nb_own + np.arange(self.nb_ghost, dtype=np.int32)
                self._g2l_table = t
            return self._g2l_table[g].astype(np.int64)
        out = np.full(g.shape, -1, dtype=np.int64)
        own = (g >= self.lo) & (g < self.hi)
        out[own] = self.own_perm[g[own] - self.lo]
        if self.nb_ghost:
            pos = np.searchsorted(self.ghosts, g)
            pos[pos >= self.nb_ghost] = self.nb_ghost - 1
            hit = (~own) & (self.ghosts[pos] == g)
            out[hit] = self.nb_own + pos[hit]
        return out


def _rows_with_cols_in(B, lo, hi):
    """Mask over the block rows of BSR B: rows holding at least one column in [lo, hi)."""
    m = (B.colidx >= lo) & (B.colidx < hi)
    rows = np.repeat(np.arange(B.nbrows, dtype=np.int64), np.diff(B.rowptr))[m]
    out = np.zeros(B.nbrows, dtype=bool)
    out[rows] = True
    return out


def _ragged_take(ptr, sel):
    """Index array gathering the ragged segments ``sel`` of a CSR-like ``ptr``; also the new ptr."""
    ptr = np.asarray(ptr, dtype=np.int64)
    cnt = ptr[sel + 1] - ptr[sel]
    nptr = np.concatenate([[0], np.cumsum(cnt)])
    idx = np.repeat(ptr[sel] - nptr[:-1], cnt) + np.arange(nptr[-1])
    return idx, nptr


def owned_patches(L, lo, hi):
    seed_nodes = L.V.vertex_nodes[L.patch_seeds].astype(np.int64)
    return np.flatnonzero((seed_nodes >= lo) & (seed_nodes < hi))


def transfer_blocks(T, bs, lo, hi):
    """Coarse-cell blocks (transfer.py:13-46) whose closure holds a fine node in [lo, hi): the rows of D_I of a block
    reach exactly the fine nodes of the coarse cell's closure."""
    if hasattr(T.D_I, "blocks_with_cols_in"):                      # rank-local generation (alfi_amd.lazy)
        return np.flatnonzero(T.D_I.blocks_with_cols_in(lo, hi))
    mb = T.blk_dofs.shape[1] // bs
    return np.flatnonzero(_rows_with_cols_in(T.D_I, lo, hi).reshape(-1, mb).any(axis=1))


def _cols_of_rows(B, rows):
    """Concatenated column indices of the block rows ``rows`` of a BSR (or lazy) matrix."""
    if hasattr(B, "cols_of_rows"):
        return B.cols_of_rows(rows)
    idx, _ = _ragged_take(B.rowptr, rows)
    return B.colidx[idx]


def _cols_of_row_range(B, lo, hi):
    if hasattr(B, "cols_of_row_range"):
        return B.cols_of_row_range(lo, hi)
    return B.colidx[B.rowptr[lo]:B.rowptr[hi]]


def owned_boundary_mask(A, lo, hi):
    """Over the owned rows [lo, hi) of BSR A: True where the row holds a column outside the owned range."""
    if hi <= lo:
        return np.zeros(0, dtype=bool)
    k0, k1 = int(A.rowptr[lo]), int(A.rowptr[hi])
    col = A.colidx[k0:k1]
    out_of_range = (col < lo) | (col >= hi)
    rows = np.repeat(np.arange(hi - lo, dtype=np.int64), np.diff(A.rowptr[lo:hi + 1]))[out_of_range]
    mask = np.zeros(hi - lo, dtype=bool)
    mask[rows] = True
    return mask


def compute_ghosts(levels, transfers, splits, l, rank):
    """Global ids (ascending) of the nodes of level l that rank needs but does not own."""
    L = levels[l]
    lo, hi = int(splits[l][rank]), int(splits[l][rank + 1])
    need = []
    if hi > lo:
        A = L.A
        need.append(A.colidx[A.rowptr[lo]:A.rowptr[hi]])                          # SpMV columns of owned rows
        if l > 0 and getattr(L, "patch_ptr", None) is not None:
            idx, _ = _ragged_take(L.patch_ptr, owned_patches(L, lo, hi))
            need.append(L.patch_dofs[idx][::L.bs] // L.bs)                        # star patches of owned vertices
        if l > 0:
            T = transfers[l - 1]
            mb = T.blk_dofs.shape[1] // L.bs
            blocks = transfer_blocks(T, L.bs, lo, hi)
            rows = (blocks[:, None] * mb + np.arange(mb)).ravel()
            need.append(_cols_of_rows(T.D_I, rows))                               # closures of those coarse cells
    if l + 1 < len(levels):
        T = transfers[l]
        flo, fhi = int(splits[l + 1][rank]), int(splits[l + 1][rank + 1])
        if fhi > flo:
            need.append(_cols_of_row_range(T.P, flo, fhi))                        # prolongation stencils
            if T.PT_plain is not T.PT:
                if getattr(T.PT_plain, "is_lazy", False):
                    need.append(_cols_of_row_range(T.PT_plain.transpose(), flo, fhi))
                else:
                    need.append(np.flatnonzero(_rows_with_cols_in(T.PT_plain, flo, fhi)))
    if not need:
        return np.zeros(0, dtype=np.int64)
    # a mark per node of the level instead of a sort of the tens of millions of column indices
    mark = np.zeros(int(L.A.nbrows), dtype=bool)
    for x in need:
        mark[np.asarray(x)] = True
    mark[lo:hi] = False
    return np.flatnonzero(mark).astype(np.int64)


def build_parts(levels, transfers, splits, rank, exchange_lists=None, force_distributed_above=None):
    """LevelPart for every level on ``rank``.  ``exchange_lists(obj) -> list over ranks`` is an all-gather of Python
    objects (torch.distributed.all_gather_object); None = compute every rank's ghost lists locally (tests)."""
    world = len(splits[0]) - 1
    parts = []
    mine = []
    for l, L in enumerate(levels):
        force = force_distributed_above is not None and l > 0 and L.n >= force_distributed_above
        lo, hi = int(splits[l][rank]), int(splits[l][rank + 1])
        p = LevelPart(l, L.bs, splits[l], rank, compute_ghosts(levels, transfers, splits, l, rank), force,
                      boundary_mask=owned_boundary_mask(L.A, lo, hi))
        parts.append(p)
        mine.append(p.ghosts_by_owner())
    if exchange_lists is not None:
        everyone = exchange_lists(mine)                  # everyone[q][l][owner] = ghosts of rank q owned by owner
        for l, p in enumerate(parts):
            p.set_send_lists([everyone[q][l][rank] for q in range(world)])
            if p.distributed:
                p.set_sum_plan([np.concatenate([np.asarray(g, dtype=np.int64) for g in everyone[q][l]]) if everyone[q][l]
                                else np.zeros(0, dtype=np.int64) for q in range(world)])
    else:
        for l, p in enumerate(parts):
            wanted = []
            for q in range(world):
                if q == rank:
                    wanted.append(np.zeros(0, dtype=np.int64))
                    continue
                g = compute_ghosts(levels, transfers, splits, l, q)
                wanted.append(g[(g >= p.lo) & (g < p.hi)])
            p.set_send_lists(wanted)
            if p.distributed:
                p.set_sum_plan([p.ghosts if q == rank else compute_ghosts(levels, transfers, splits, l, q)
                                for q in range(world)])
    return parts


# ---------------------------------------------------------------------------------------------------------------------
# localisation: the rank's pieces of the operators, patches and transfers in local numbering
# ---------------------------------------------------------------------------------------------------------------------
class LocalLevel(object):
    pass


class LocalTransfer(object):
    pass


def _map_cols(B, part, nbcols, allow_drop=False):
    """BSR with columns renumbered to part's local numbering; entries with absent columns are dropped (only legal for
    ghost rows, whose remote couplings no owned patch needs)."""
    lc = part.g2l(B.colidx)
    if (lc < 0).any():
        if not allow_drop:
            raise AssertionError("a needed column is not in the local node set")
        keep = lc >= 0
        csum = np.concatenate([[0], np.cumsum(keep)])
        rowptr = csum[B.rowptr.astype(np.int64)]
        return BSR(B.nbrows, nbcols, B.bs, rowptr, lc[keep], B.vals[keep])
    return BSR(B.nbrows, nbcols, B.bs, B.rowptr, lc, B.vals)


def localize_operator(A, part):
    """The rank's rows of the (global or lazy) level operator ``A`` in local numbering: owned rows complete, ghost rows
    restricted to local columns."""
    bs = A.bs
    rows = A.select_rows(part.nodes)                 # fresh arrays (assembled or cut for this call): edited in place below
    nnz_own = int(rows.rowptr[part.nb_own])
    lc = part.g2l(rows.colidx)
    if (lc[:nnz_own] < 0).any():
        raise AssertionError("a needed column is not in the local node set")
    keep = lc[nnz_own:] >= 0
    if keep.all():
        return BSR(part.nb_loc, part.nb_loc, bs, rows.rowptr, lc, rows.vals)      # (vals None: the sparsity only, values on the device)
    # the owned rows are a prefix and stay where they are (2 GB of values per rank at config 4 over 4 ranks: no copy of them);
    # the ghost rows' kept blocks are packed behind them in the same arrays
    kept = np.flatnonzero(keep)
    nk = kept.shape[0]
    vals = rows.vals
    if vals is not None:
        vals[nnz_own:nnz_own + nk] = vals[nnz_own:][kept]
    lc[nnz_own:nnz_own + nk] = lc[nnz_own:][kept]
    csum = np.concatenate([[0], np.cumsum(keep)])
    gh_ptr = rows.rowptr[part.nb_own:].astype(np.int64) - nnz_own
    rowptr = np.concatenate([rows.rowptr[:part.nb_own].astype(np.int64), nnz_own + csum[gh_ptr]])
    return BSR(part.nb_loc, part.nb_loc, bs, rowptr, lc[:nnz_own + nk], None if vals is None else vals[:nnz_own + nk])


def assembly_cells(V, part):
    """What a rank needs to re-assemble ITS operator rows (owned and ghost) on the device: the mesh cells that touch a
    local node, their nodes in the numbering of the rank's state vector -- local nodes first (the level's local numbering),
    then the cells' remaining nodes, ascending global id -- and the global node of every state entry."""
    gcn = np.asarray(V.cell_nodes, dtype=np.int64)
    loc = part.g2l(gcn.ravel()).reshape(gcn.shape)
    cells = np.flatnonzero((loc >= 0).any(axis=1))
    loc, gcn = loc[cells], gcn[cells]
    extra = np.unique(gcn[loc < 0])
    cn = np.where(loc >= 0, loc, part.nb_loc + np.searchsorted(extra, gcn))
    return cells, cn.astype(np.int32), np.concatenate([part.nodes, extra])


def localize_level(L, part):
    """Operator rows of all local nodes (owned rows complete; ghost rows restricted to local columns -- they only feed the
    patch sub-matrix gather), owned Dirichlet dofs, owned patches; everything in local numbering."""
    out = LocalLevel()
    bs = L.bs
    out.level, out.bs, out.part = L.level, bs, part
    out.n, out.n_own = part.nb_loc * bs, part.nb_own * bs
    out.A = localize_operator(L.A, part)
    bcn = np.asarray(L.bc_dofs, dtype=np.int64)[::bs] // bs
    bcn = np.sort(part.g2l(bcn[(bcn >= part.lo) & (bcn < part.hi)]))
    out.bc_dofs = (bcn[:, None] * bs + np.arange(bs)).ravel().astype(np.int32)
    if L.level > 0 and part.nb_own > 0:
        sel = owned_patches(L, part.lo, part.hi)
        idx, nptr = _ragged_take(L.patch_ptr, sel)
        gd = L.patch_dofs[idx].astype(np.int64)
        ld = part.g2l(gd // bs) * bs + gd % bs
        assert (ld >= 0).all()
        pid = np.repeat(np.arange(len(sel)), np.diff(nptr))
        # patches without a ghost dof first (the library applies them while the forward halo is in flight); inside a
        # patch ascending local dofs
        has_ghost = np.zeros(len(sel), dtype=bool)
        has_ghost[pid[ld >= part.nb_own * bs]] = True
        porder = np.concatenate([np.flatnonzero(~has_ghost), np.flatnonzero(has_ghost)])
        rank_of = np.empty(len(sel), dtype=np.int64)
        rank_of[porder] = np.arange(len(sel))
        order = np.lexsort((ld, rank_of[pid]))
        cnt = np.diff(nptr)[porder]
        out.patch_ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        out.patch_dofs = ld[order].astype(np.int32)
        if getattr(L, "patch_groups", None) is not None:        # labels of condensed patch factors follow their dofs
            out.patch_groups = np.asarray(L.patch_groups)[idx][order].astype(np.int32)
        out.patch_ids = sel[porder]
        out.npatch_int = int(np.count_nonzero(~has_ghost))
    else:
        out.patch_ptr, out.patch_dofs = np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32)
        out.patch_ids = np.zeros(0, dtype=np.int64)
        out.npatch_int = 0
    return out


def localize_transfer(T, Lf, pc, pf):
    """The rank's share of the transfer between levels pc.level and pf.level (pf.nb_own > 0)."""
    out = LocalTransfer()
    bs = Lf.bs
    mb = T.blk_dofs.shape[1] // bs
    blocks = transfer_blocks(T, bs, pf.lo, pf.hi)
    gd = T.blk_dofs[blocks].astype(np.int64)
    ld = pf.g2l(gd // bs) * bs + gd % bs
    assert (ld >= 0).all()
    out.blocks = blocks
    out.blk_dofs = ld.astype(np.int32)
    if hasattr(T, "interior_mats"):                                  # rank-local generation: assembled for these blocks only
        out.K_II, out.D_II = T.interior_mats(blocks)
    else:
        out.K_II, out.D_II = np.ascontiguousarray(T.K_II[blocks]), np.ascontiguousarray(T.D_II[blocks])
    rows = (blocks[:, None] * mb + np.arange(mb)).ravel()
    out.D_I = _map_cols(T.D_I.select_rows(rows), pf, pf.nb_loc)
    # rows of D_I^T for the owned fine nodes only: s = r - gamma D_I^T t is formed on owned rows
    out.D_IT = out.D_I.transpose().row_range(0, pf.nb_own)
    P = _map_cols(T.P.select_rows(pf.own_nodes), pc, pc.nb_loc)
    out.P = P
    out.PT = P.transpose()                                  # (coarse local) x (fine owned): partial sums, reverse-added
    if T.PT_plain is not T.PT:
        Pp = T.PT_plain.transpose().select_rows(pf.own_nodes)
        out.PT_plain = _map_cols(Pp, pc, pc.nb_loc).transpose()
    else:
        out.PT_plain = out.PT
    out.nu, out.gamma = T.nu, T.gamma
    return out


def localize(levels, transfers, parts):
    """(local levels, local transfers, first level present) for the rank the parts belong to."""
    present = [p.nb_loc > 0 for p in parts]
    if True not in present:
        raise ValueError("rank %d holds no part of any level: no level reaches min_dofs, so nothing is partitioned over the "
                         "%d ranks (lower min_dofs / ALFI_DIST_MIN_DOFS or use fewer ranks)" % (parts[0].rank, len(parts[0].splits) - 1))
    lmin = present.index(True)
    assert all(present[lmin:]), "levels present on a rank must be contiguous"
    llev = [localize_level(levels[l], parts[l]) for l in range(lmin, len(levels))]
    ltr = []
    for l in range(lmin + 1, len(levels)):
        if parts[l].nb_own > 0:
            ltr.append(localize_transfer(transfers[l - 1], levels[l], parts[l - 1], parts[l]))
        else:
            raise AssertionError("a level above a present one must be owned in part")
    return llev, ltr, lmin


def localize_pressure(B, mass_diag, cell_nodes, part, bs, mass_inv=None):
    """The rank's share of the discontinuous pressure space for the outer solve: a cell -- with its one (P0) or several
    (P_{k-1}^dg of the Scott-Vogelius pair: rows c * npc .. c * npc + npc - 1 of B, sv.build_sv_pressure_coupling) pressure
    dofs -- belongs to the rank that owns its lowest-numbered node (all its nodes are then local: they are columns of that
    node's operator row).  Returns (owned pressure rows ascending, B_loc = those rows of B with columns in the local dof
    numbering (scipy CSR, n_p_own x n_loc), mass_diag_loc or None, the rows' diagonal blocks of ``mass_inv`` or None)."""
    import scipy.sparse as sp
    cell_nodes = np.asarray(cell_nodes)
    npc = int(B.shape[0]) // cell_nodes.shape[0]
    assert npc * cell_nodes.shape[0] == B.shape[0], "pressure dofs must come cell by cell"
    low = cell_nodes.min(axis=1).astype(np.int64)
    cells = np.flatnonzero((low >= part.lo) & (low < part.hi))
    prows = (cells[:, None] * npc + np.arange(npc)).ravel()
    Bl = sp.csr_matrix(B)[prows].tocoo()
    lcol = part.g2l(Bl.col // bs) * bs + Bl.col % bs
    assert (lcol >= 0).all(), "a velocity dof of an owned cell is not in the local node set"
    Bloc = sp.csr_matrix((Bl.data, (Bl.row, lcol)), shape=(len(prows), part.nb_loc * bs))
    Bloc.sort_indices()
    md = np.asarray(mass_diag)[prows] if mass_diag is not None else None
    Mi = None
    if mass_inv is not None:
        Mi = sp.csr_matrix(mass_inv)[prows][:, prows].tocsr()     # block diagonal by cell: closed under the selection
        Mi.sort_indices()
    return prows, Bloc, md, Mi


# ---------------------------------------------------------------------------------------------------------------------
# communication (torch.distributed)
# ---------------------------------------------------------------------------------------------------------------------
class Comm(object):
    """Halo exchange and all-reduce on torch tensors.  CPU tensors go straight to the backend; device tensors go
    straight to NCCL/RCCL, or are staged through the host when the backend is gloo."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)

    def all_gather_object(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj, group=self.group)
        return out

    def _staged(self, t):
        return t.is_cuda and self.backend != "nccl"

    def allreduce(self, t):
        if self._staged(t):
            import torch
            torch.cuda.current_stream().synchronize()
            h = t.cpu()
            self.dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, group=self.group)

    def exchange_begin(self, send, recv, send_counts, recv_counts):
        """Start the exchange without making the current stream wait for it; returns a handle for exchange_end.  The
        collective runs on the backend's own stream after everything enqueued so far on the current stream (RCCL); with
        a host-staged backend the whole exchange happens here."""
        if self._staged(send):
            self.exchange(send, recv, send_counts, recv_counts)
            return None
        return self.dist.all_to_all_single(recv, send, recv_counts, send_counts, group=self.group, async_op=True)

    @staticmethod
    def exchange_end(work):
        if work is not None:
            work.wait()          # stream-level wait (no host block) for the RCCL backend

    def exchange(self, send, recv, send_counts, recv_counts):
        """recv[segment q] <- rank q's send[segment me]; counts in elements of the tensors."""
        sc, rc = send_counts, recv_counts
        if not isinstance(sc, list):
            sc, rc = [int(c) for c in sc], [int(c) for c in rc]
        if self._staged(send):
            import torch
            torch.cuda.current_stream().synchronize()
            hs, hr = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
            self.dist.all_to_all_single(hr, hs, rc, sc, group=self.group)
            recv.copy_(hr)
        else:
            self.dist.all_to_all_single(recv, send, rc, sc, group=self.group)


class SoloComm(object):
    """ONE rank of a ``world``-rank job alone in its process (``DistMultigrid(solo=(rank, world))``): the partition, the
    localisation and every kernel launch are those of that rank, the exchange points do nothing (ghost values stay as they are,
    reductions keep the local value).  The numbers such a run produces are meaningless; the DEVICE TIME of its kernels is that of
    the rank's share -- a measurement of the per-rank compute of an N-GPU run on a box with one GPU (scripts/solo_rank_time.py)."""

    backend = "solo"
    dist = None
    group = None

    def __init__(self, rank, world):
        self.rank, self.world = int(rank), int(world)

    def all_gather_object(self, obj):
        return [obj] * self.world

    def allreduce(self, t):
        pass

    def exchange_begin(self, send, recv, send_counts, recv_counts):
        return None

    @staticmethod
    def exchange_end(work):
        pass

    def exchange(self, send, recv, send_counts, recv_counts):
        pass


class HaloBuffers(object):
    """Per level: the torch-owned buffers the library packs into / unpacks from, and the exchange split sizes."""

    def __init__(self, part, device):
        import torch
        bs = part.bs
        self.part = part
        self.send_counts = part.send_counts * bs
        self.recv_counts = part.recv_counts * bs
        self.sendbuf = torch.zeros(max(int(self.send_counts.sum()), 1), dtype=torch.float64, device=device)
        self.recvbuf = torch.zeros(max(int(self.recv_counts.sum()), 1), dtype=torch.float64, device=device)
        self.nsend, self.nrecv = int(self.send_counts.sum()), int(self.recv_counts.sum())
        # the callbacks run ~200 times per cycle: keep the per-call host work to the collective itself
        self._send, self._recv = self.sendbuf[:self.nsend], self.recvbuf[:self.nrecv]
        self._sc, self._rc = [int(c) for c in self.send_counts], [int(c) for c in self.recv_counts]
        self._work = None

    def forward(self, comm):
        comm.exchange(self._send, self._recv, self._sc, self._rc)

    def forward_begin(self, comm):
        self._work = comm.exchange_begin(self._send, self._recv, self._sc, self._rc)

    def forward_end(self, comm):
        comm.exchange_end(self._work)
        self._work = None

    def reverse_begin(self, comm):
        self._work = comm.exchange_begin(self._recv, self._send, self._rc, self._sc)

    reverse_end = forward_end

    def reverse(self, comm):
        comm.exchange(self._recv, self._send, self._rc, self._sc)


CommFn = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64)


class DistMultigrid(object):
    """The rank's share of PCMG (alfi/solver.py:359-379) on its GPU; collective over the process group.

    ``levels, transfers``: the *global* hierarchy of alfi_amd.problem.build_hierarchy (every rank generates the same one;
    only the rank's rows are uploaded)."""

    def __init__(self, levels, transfers, k, robust_restriction=False, group=None, device=None, min_dofs=400000,
                 coarse_inverse=None, verbose=False, force_distributed=False, overlap=None, overlap_min_dofs=None,
                 transport=None, on_stage=None, use_overlap_rule=True, solo=None, full_cycle=False):
        """transport: "rccl" -- the library's own RCCL communicator serves every exchange point of a cycle (no Python
        between the kernels; the default whenever the process group's backend is nccl) -- or "callback": the library
        calls back into this module, which exchanges through torch.distributed (the test transport: gloo, ranks sharing
        a GPU).  ALFI_DIST_TRANSPORT overrides.  on_stage(name): called at "partition", "localize", "comm_init",
        "upload_factor" (bench.py's per-rank stage markers); ``setup_s`` holds the seconds each took."""
        import os
        import time
        import torch
        from . import hip
        stage = on_stage or (lambda name: None)
        self.setup_s = {}
        # solo = (rank, world): that rank of a world-rank job alone in this process, exchange points stubbed (SoloComm: timing only)
        self.comm = SoloComm(*solo) if solo is not None else Comm(group)
        rank = self.comm.rank
        if solo is not None:
            transport = "callback"
        if transport is None:
            transport = os.environ.get("ALFI_DIST_TRANSPORT") or ("rccl" if self.comm.backend == "nccl" else "callback")
        if transport not in ("rccl", "callback"):
            raise ValueError("transport must be 'rccl' or 'callback'")
        self.transport = transport
        if overlap is None:
            import os
            overlap = os.environ.get("ALFI_DIST_OVERLAP", "1") != "0"
        self.overlap = overlap
        self._in_cycle = False
        self._red_views = {}
        # which levels overlap: ALFI_DIST_OVERLAP_MIN_DOFS (smallest per-rank share that overlaps; tests, measurements) or,
        # without it, the rule above from the halo sizes of the partition (use_overlap_rule=False: never)
        use_rule = False
        if overlap_min_dofs is None:
            import os
            if "ALFI_DIST_OVERLAP_MIN_DOFS" in os.environ:
                overlap_min_dofs = int(os.environ["ALFI_DIST_OVERLAP_MIN_DOFS"])
            else:
                overlap_min_dofs = 1 << 62
                use_rule = overlap and use_overlap_rule
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        stage("partition")
        t0 = time.time()
        # full_cycle: the caller applies full cycles (the outer solves of a Newton step): the partition balances THEIR work
        self.splits = choose_splits(levels, self.comm.world, min_dofs, full_cycle=full_cycle)
        self.parts = build_parts(levels, transfers, self.splits, rank, None if solo is not None else self.comm.all_gather_object,
                                 force_distributed_above=min_dofs if force_distributed else None)
        self.setup_s["partition"] = time.time() - t0
        stage("localize")
        t0 = time.time()
        llev, ltr, self.lmin = localize(levels, transfers, self.parts)
        self.setup_s["localize"] = time.time() - t0
        self.local_levels, self.local_transfers = llev, ltr
        self.k = k
        with torch.cuda.stream(self.stream):
            self.ctx = ctx = hip.Context(device.index or 0, stream=self.stream.cuda_stream)
            if transport == "rccl":
                from . import _lib
                stage("comm_init")
                t0 = time.time()
                box = [_lib.comm_unique_id() if rank == 0 else None]
                self.comm.dist.broadcast_object_list(box, src=self.comm.dist.get_global_rank(group, 0) if group else 0,
                                                     group=group)
                ctx.comm_init(box[0], rank, self.comm.world)          # collective: ncclCommInitRank
                lib_rank, lib_world = ctx.comm_size()
                if (lib_rank, lib_world) != (rank, self.comm.world):
                    raise RuntimeError("the library's communicator reports rank %d of %d, the process group rank %d of %d"
                                       % (lib_rank, lib_world, rank, self.comm.world))
                self.setup_s["comm_init"] = time.time() - t0
            else:
                self.red = torch.zeros(RED_LEN, dtype=torch.float64, device=device)
                self._cb = CommFn(self._callback)
                ctx.set_comm(self._cb, self.red.data_ptr(), RED_LEN)
            self.halos = {}
            self.levels = []
            self.overlap_levels = []
            rule = {}
            if use_rule:
                # collective, once, over the global list of levels: the largest message of every level's exchange on any rank
                # (and the fewest inverse bytes of interior patches)
                mine = []
                for i, p in enumerate(self.parts):
                    msg = 8 * p.bs * int(max(p.send_counts.max(initial=0), p.recv_counts.max(initial=0))) if p.distributed else 0
                    LL = llev[i - self.lmin] if i >= self.lmin else None
                    inner = float("inf")
                    if LL is not None and LL.level > 0 and p.nb_own > 0:
                        sizes = np.diff(np.asarray(LL.patch_ptr[:LL.npatch_int + 1], dtype=np.int64)).astype(np.float64)
                        inner = 8.0 * float((sizes * sizes).sum())
                    mine.append((msg, inner))
                every = self.comm.all_gather_object(mine)
                rule = {p.level: bool(p.distributed and overlap_rule(max(e[i][0] for e in every), min(e[i][1] for e in every)))
                        for i, p in enumerate(self.parts)}
            stage("upload_factor")
            t0 = time.time()
            for LL in llev:
                p = LL.part
                dl = hip.Level(ctx, LL.A, LL.bc_dofs)
                send_nodes = np.concatenate(p.send_nodes).astype(np.int32) if p.send_counts.sum() else \
                    np.zeros(0, dtype=np.int32)
                if transport == "rccl":
                    dl.set_partition(p.nb_own, p.distributed, send_nodes, None, None, p.nb_ghost)
                    nbr = np.flatnonzero((p.send_counts > 0) | (p.recv_counts > 0))
                    dl.set_neighbours(nbr, p.send_counts[nbr], p.recv_counts[nbr])
                    if p.distributed and getattr(p, "sum_ranks", None) is not None:
                        # merged reverse-add + forward exchange of the smoother: 2 instead of 3 halo exchanges per iteration
                        dl.set_sum_exchange(p.sum_ranks, p.sum_counts, p.sum_send_nodes, p.sum_nodes, p.sum_ptr, p.sum_src)
                else:
                    hb = HaloBuffers(p, device)
                    dl.set_partition(p.nb_own, p.distributed, send_nodes, hb.sendbuf.data_ptr(), hb.recvbuf.data_ptr(),
                                     p.nb_ghost)
                    self.halos[dl.id] = hb
                if LL.level > 0:
                    dl.set_patches(LL.patch_ptr, LL.patch_dofs)
                    if hip.condense_patches(LL):
                        dl.set_patch_groups(LL.patch_groups)
                    if LL.A.vals is not None:          # (None: the operators are formed on the device first, the caller factors)
                        dl.factor_with_fallback()
                    decided = rule[LL.level] if use_rule else overlap_decision(p.splits, p.bs, p.distributed, overlap,
                                                                               overlap_min_dofs)
                    if decided:
                        # interior rows / patches are worked on while the forward halo is in flight (overlap_rule above)
                        self.overlap_levels.append(LL.level)
                        dl.set_overlap(p.nb_int, LL.npatch_int)
                elif p.nb_own > 0 and LL.A.vals is not None:
                    self._coarse(dl, levels[0].A, coarse_inverse)
                self.levels.append(dl)
            self.mg = hip.Multigrid.__new__(hip.Multigrid)
            self.mg._from_device_levels(ctx, self.levels, ltr, k, robust_restriction)
            self.setup_s["upload_factor"] = time.time() - t0
        self.fine = llev[-1]
        self.n_own = self.fine.n_own
        self.n_loc = self.fine.n
        if verbose:
            print("[alfi_amd.dist] rank %d: levels %d..%d, finest owns %d of %d dofs (+%d ghost), %d patches"
                  % (rank, self.lmin, len(levels) - 1, self.n_own, levels[-1].n, self.n_loc - self.n_own,
                     len(self.fine.patch_ptr) - 1), flush=True)

    def _coarse(self, dl, A0, coarse_inverse):
        """Coarse solve of the rank that owns level 0: the library's own dense factorisation of the (complete) local
        operator, or an inverse the caller computes from the global one."""
        if coarse_inverse is None:
            dl.coarse_factor_auto()      # (local numbering: no coordinates -- the sparse path bisects by graph level sets)
            return
        inv = coarse_inverse(A0)
        if isinstance(inv, tuple):
            inv, self._keep_inv = inv
        dl.set_coarse_inverse(inv)

    # the library calls this at every exchange point of the cycle (alfi_ctx_set_comm): ~200 times per V-cycle, so the host
    # work per call is kept to the collective itself (on small levels the cycle is bound by exactly this host time)
    def _callback(self, user, op, level_id, offset, count):
        try:
            if self._in_cycle:            # vcycle()/fcycle() made the library's stream current: nothing to switch
                self._dispatch(op, level_id, offset, count)
            else:
                import torch
                # the collectives must be ordered against the library's stream whatever stream the caller had current
                with torch.cuda.stream(self.stream):
                    self._dispatch(op, level_id, offset, count)
            return 0
        except Exception:                                   # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return -1

    def _dispatch(self, op, level_id, offset, count):
        if op == COMM_ALLREDUCE:
            key = (offset, count)
            view = self._red_views.get(key)
            if view is None:
                view = self._red_views[key] = self.red[offset:offset + count]
            self.comm.allreduce(view)
        elif op == COMM_HALO_FWD:
            self.halos[level_id].forward(self.comm)
        elif op == COMM_HALO_REV:
            self.halos[level_id].reverse(self.comm)
        elif op == COMM_HALO_FWD_BEGIN:
            self.halos[level_id].forward_begin(self.comm)
        elif op == COMM_HALO_REV_BEGIN:
            self.halos[level_id].reverse_begin(self.comm)
        elif op == COMM_HALO_FWD_END or op == COMM_HALO_REV_END:
            self.halos[level_id].forward_end(self.comm)
        else:
            raise ValueError("unknown exchange op %d" % op)

    def update(self, levels, coarse_inverse=None):
        """New operator values on every level (a Newton step): the rank's rows are cut out of the global operators again,
        patches re-gathered and re-inverted, the coarse inverse rebuilt by its owner."""
        import torch
        from . import hip
        with torch.cuda.stream(self.stream):
            for dl, LL in zip(self.levels, self.local_levels):
                new = localize_level(levels[LL.level], LL.part)
                LL.A = new.A
                dl.update_values(new.A.vals)
                if LL.level > 0:
                    dl.factor_with_fallback()
                elif LL.part.nb_own > 0:
                    self._coarse(dl, levels[0].A, coarse_inverse)

    def refactor(self, levels=None, coarse_inverse=None):
        """The operator values changed ON THE DEVICE (alfi_level_assemble): patches re-gathered and re-inverted, the coarse
        factorisation rebuilt by its owner."""
        import torch
        with torch.cuda.stream(self.stream):
            for dl, LL in zip(self.levels, self.local_levels):
                if LL.level > 0:
                    if LL.part.nb_own > 0:
                        dl.factor_with_fallback()
                elif LL.part.nb_own > 0:
                    self._coarse(dl, levels[0].A if levels is not None else None, coarse_inverse)

    def local_vec(self, global_array=None):
        """Device vector of the finest level in local numbering (owned + ghost slots), filled from a global array."""
        v = self.ctx.vec(self.n_loc)
        if global_array is not None:
            p, bs = self.fine.part, self.fine.bs
            loc = np.zeros(self.n_loc)
            loc[:self.n_own] = np.asarray(global_array)[p.own_dofs()]
            v.set(loc)
        return v

    def owned(self, v):
        return v.get()[:self.n_own]

    def vcycle(self, b, x):
        import torch
        with torch.cuda.stream(self.stream):
            self._in_cycle = True
            try:
                self.mg.vcycle(b, x)
            finally:
                self._in_cycle = False

    def fcycle(self, b, x):
        import torch
        with torch.cuda.stream(self.stream):
            self._in_cycle = True
            try:
                self.mg.fcycle(b, x)
            finally:
                self._in_cycle = False

    def sync(self):
        self.ctx.sync()

    def close(self):
        self.mg.close()
        self.ctx.close()


class DistSaddle(object):
    """The outer solve of one Newton step (alfi/solver.py:386-422: FGMRES around PCFIELDSPLIT-Schur-full, fieldsplit_0 = one
    PCMG full cycle, fieldsplit_1 = DGMassInv) on partitioned levels.  The whole Krylov loop runs inside the library
    (``alfi_saddle_solve`` on a partitioned finest level: the same code as on one GPU, every dot product and norm one
    all-reduce, the products with the rank's rows of the discrete divergence and the halo routes between its kernels); this
    class cuts the rank's pieces of B and of the pressure mass matrix out of the global ones and hands them over.
    Velocity dofs are owned with their nodes, pressure dofs with their cells (``localize_pressure``); vectors hold
    (owned velocity dofs | owned pressure dofs)."""

    def __init__(self, dmg, B, mass_diag, cell_nodes, nu, gamma, remove_constant_nullspace=True, mass_inv=None):
        """mass_inv: scipy sparse inverse of the block-diagonal pressure mass matrix (discontinuous P_{k-1} pressure of the
        Scott-Vogelius pair, DGMassInv solver.py:15-38) -- replaces ``mass_diag`` (P0); ``self.cells`` then holds the owned
        pressure ROWS (npc per owned cell)."""
        from . import hip
        self.dmg = dmg
        F = dmg.fine
        self.part, self.bs = F.part, F.bs
        self.cells, Bloc, md, Mi = localize_pressure(B, None if mass_inv is not None else mass_diag, cell_nodes, self.part,
                                                     self.bs, mass_inv)
        self.n_own, self.n_loc, self.np_own = dmg.n_own, dmg.n_loc, len(self.cells)
        self.n = self.n_own + self.np_own
        self.np_global = int(B.shape[0])
        with self._on_stream():
            # collective: the library all-reduces the number of pressure dofs once
            self.sad = hip.Saddle(dmg.mg, Bloc, md, nu, gamma, remove_constant_nullspace, mass_inv=Mi, n_u=self.n_own)

    def _on_stream(self):
        """The library's stream current and the callback transport told so (its collectives are torch.distributed calls)."""
        import contextlib
        import torch
        dmg = self.dmg

        @contextlib.contextmanager
        def cm():
            with torch.cuda.stream(dmg.stream):
                prev, dmg._in_cycle = dmg._in_cycle, True
                try:
                    yield
                finally:
                    dmg._in_cycle = prev
        return cm()

    def update(self, nu, gamma):
        self.sad.update(nu, gamma)

    def mult(self, x, y):
        """y = [A B^T; B 0] x on (owned velocity dofs | owned pressure dofs); x, y: device vectors (``.ptr``)."""
        with self._on_stream():
            self.sad.mult(x, y)

    def precond(self, v, z):
        """z = P^-1 v: y_u = MG(b_u); y_p = -(nu + gamma) M^-1 (b_p - B y_u); y_u = MG(b_u - B^T y_p)."""
        with self._on_stream():
            self.sad.precond(v, z)

    def solve(self, b, rtol=1e-8, atol=1e-8, max_it=500, restart=30):
        """b: the rank's entries of the right-hand side (n_own + np_own) as a NumPy array or a torch tensor on the device.
        Zero initial guess, KSP's default convergence test on the recurrence residual.  Returns (x, iterations, true residual
        norm), x of the kind b was."""
        from .hip import RawVec
        ctx = self.dmg.ctx
        if hasattr(b, "data_ptr"):            # a torch tensor on the library's device
            import torch
            x = torch.empty_like(b)
            with self._on_stream():
                its, rn = self.sad.solve(RawVec(b.data_ptr(), self.n), RawVec(x.data_ptr(), self.n), rtol, atol, max_it, restart)
            return x, its, rn
        db, dx = ctx.vec(np.ascontiguousarray(b, dtype=np.float64)), ctx.vec(max(self.n, 1))
        with self._on_stream():
            its, rn = self.sad.solve(db, dx, rtol, atol, max_it, restart)
        return dx.get()[:self.n], its, rn

    def close(self):
        self.sad.close()


class _StatePart(object):
    """What HaloBuffers reads of a LevelPart."""

    def __init__(self, bs, send_counts, recv_counts):
        self.bs, self.send_counts, self.recv_counts = bs, send_counts, recv_counts


class StateExchange(object):
    """The distributed Newton state as input of the rank's operator refresh: every level's state vector (the level's local
    nodes and the ring of nodes of the cells around them, alfi_level_set_assembly) filled ON THE DEVICE from the velocity the
    ranks own on the finest level -- one halo exchange per refresh, then an index gather per level (on the nested hierarchies
    a node of level l IS a node of the finest level: the composition of the ``inject`` maps, alfi/solver.py:595).
    The exchange runs through a level that exists only for its halo plan (identity operator): owned block = the rank's owned
    finest nodes in the finest level's local order, ghosts = every other finest node some level's refresh reads here; it
    therefore takes whichever transport the multigrid levels take (the library's RCCL communicator, or the callback)."""

    def __init__(self, dmg, levels, transfers, asm_nodes, device):
        """asm_nodes[i]: global node ids (numbering of level dmg.lmin + i) of the state entries of local level i, or None."""
        from . import hip
        from .problem import BSR
        p = dmg.fine.part
        bs, rank, world = p.bs, dmg.comm.rank, dmg.comm.world
        nlev = len(levels)
        to_fine = [None] * nlev                       # node of level l -> finest node at the same position
        to_fine[-1] = np.arange(levels[-1].A.nbrows, dtype=np.int64)
        for l in range(nlev - 2, -1, -1):
            T = transfers[l]
            if T.inject_map is None:
                raise ValueError("non-nested hierarchy: no node-to-node inject")
            to_fine[l] = to_fine[l + 1][np.asarray(T.inject_map, dtype=np.int64)]
        need = [None if a is None else to_fine[dmg.lmin + i][np.asarray(a, dtype=np.int64)] for i, a in enumerate(asm_nodes)]
        allneed = np.unique(np.concatenate([n for n in need if n is not None] + [np.zeros(0, dtype=np.int64)]))
        ghosts = allneed[(allneed < p.lo) | (allneed >= p.hi)]               # ascending => grouped by owner
        owner = np.searchsorted(p.splits, ghosts, side="right") - 1
        recv_counts = np.bincount(owner, minlength=world).astype(np.int64)
        off = np.concatenate([[0], np.cumsum(recv_counts)])
        wanted = dmg.comm.all_gather_object([ghosts[off[q]:off[q + 1]] for q in range(world)])
        send_lists = [p.own_perm[np.asarray(wanted[q][rank], dtype=np.int64) - p.lo] for q in range(world)]
        send_counts = np.array([len(x) for x in send_lists], dtype=np.int64)
        send_nodes = np.concatenate(send_lists).astype(np.int32) if send_counts.sum() else np.zeros(0, dtype=np.int32)
        nb = p.nb_own + len(ghosts)
        ctx = dmg.ctx
        A = BSR(nb, nb, bs, np.arange(nb + 1, dtype=np.int32), np.arange(nb, dtype=np.int32), np.tile(np.eye(bs), (nb, 1, 1)))
        self.level = hip.Level(ctx, A, np.zeros(0, dtype=np.int32))
        if dmg.transport == "rccl":
            self.level.set_partition(p.nb_own, True, send_nodes, None, None, len(ghosts))
            nbr = np.flatnonzero((send_counts > 0) | (recv_counts > 0))
            self.level.set_neighbours(nbr, send_counts[nbr], recv_counts[nbr])
        else:
            hb = HaloBuffers(_StatePart(bs, send_counts, recv_counts), device)
            self.level.set_partition(p.nb_own, True, send_nodes, hb.sendbuf.data_ptr(), hb.recvbuf.data_ptr(), len(ghosts))
            dmg.halos[self.level.id] = hb
        self.ctx, self.bs, self.n_own = ctx, bs, p.nb_own * bs
        self.vec = ctx.vec(max(nb * bs, 1))
        # position of a finest node in that vector: owned -> its local index, ghost -> behind the owned block
        def pos(g):
            own = (g >= p.lo) & (g < p.hi)
            out = np.empty(g.shape[0], dtype=np.int64)
            out[own] = p.own_perm[g[own] - p.lo]
            out[~own] = p.nb_own + np.searchsorted(ghosts, g[~own])
            return out
        self.idx = [None if n is None else ctx.ivec(pos(n)) for n in need]
        self.bytes_received = int(len(ghosts)) * bs * 8

    def refresh(self, du_owned, states):
        """du_owned: device vector that starts with the rank's owned finest velocity; states[i]: device state of local level i."""
        self.ctx.copy(self.vec, du_owned, n=self.n_own)
        self.level.halo_forward(self.vec)
        for ix, st in zip(self.idx, states):
            if ix is not None:
                self.ctx.gather(st, self.vec, ix, self.bs)

    def close(self):
        self.level.close()


def _dist_ns_solver_class():
    from .nssolver import HipNavierStokesSolver

    class DistNavierStokesSolver(HipNavierStokesSolver):
        """HipNavierStokesSolver with the device side on partitioned levels (one process per GPU): DistMultigrid + DistSaddle.
        Every rank rediscretises ITS OWN rows of the level operators -- on its device from the cells that touch its nodes
        (``_rediscretise_device``; on its host cores only with ALFI_DEVICE_ASSEMBLY=0).  The Newton state is DISTRIBUTED on the
        devices, every rank its owned velocity and pressure dofs (``StateExchange`` feeds the levels' refresh states from it;
        ``u`` / ``p`` gather it, collectively, when somebody asks); the barycentric hierarchy of the Scott-Vogelius pair and
        the host-assembly path keep a replicated host state."""

        def __init__(self, *args, min_dofs=400000, group=None, device_state=True, **kwargs):
            """device_state False: the Newton state replicated on the hosts and gathered after every linear solve (the loop of
            round 4; kept for comparisons)."""
            self._min_dofs, self._group, self._want_device_state = min_dofs, group, bool(device_state)
            super().__init__(*args, **kwargs)

        def _device_state_resident(self):
            # the state lives distributed on the devices -- every rank its owned velocity and pressure dofs -- whenever the
            # operators are refreshed there and the hierarchy is nested (the barycentric one of the Scott-Vogelius pair injects
            # by point evaluation: it keeps the replicated host state)
            return self.device_assembly and not self.sv and getattr(self, "_exch", None) is not None

        def _any_rank(self, flag):
            return any(self.dmg.comm.all_gather_object(bool(flag)))

        def _lazy_generation(self):
            # rank-local generation: every rank assembles the operator / transfer rows of its partition only (config 4 on 8
            # ranks: 4.5 GB of host memory per rank instead of 25).  The HOST refresh of SUPG terms works on global values: with
            # SUPG the generation is rank-local only while the operators are refreshed on the device (the default).
            import os
            return (not self.supg or self.device_assembly) and os.environ.get("ALFI_DIST_GLOBAL_GENERATION") != "1"

        def _create_device(self, restriction):
            self.dmg = DistMultigrid(self.levels, self.transfers, self.params["fieldsplit_0"]["mg_levels"]["ksp_max_it"],
                                     robust_restriction=restriction, group=self._group, min_dofs=self._min_dofs,
                                     full_cycle=self.params["fieldsplit_0"].get("pc_mg_type") == "full")
            self.ctx = self.dmg.ctx
            L = self.levels[-1]
            self.saddle = DistSaddle(self.dmg, self.B, self.vol, L.V.cell_nodes, self.nu, self.gamma,
                                     remove_constant_nullspace=self.nullspace, mass_inv=self.Minv if self.sv else None)

        def _push_operators(self):
            self.dmg.update(self.levels)

        def _supg_host_needs_global_values(self):
            from .lazy import LazyOperator
            if isinstance(self.levels[-1].A, LazyOperator):
                raise RuntimeError("SUPG on partitioned levels without the device-side operator refresh needs the global "
                                   "operator values: start with ALFI_DIST_GLOBAL_GENERATION=1 (the hierarchy was generated "
                                   "rank-locally because the device refresh was expected to be available)")

        # -- operator refresh on the device, every rank its own rows (alfi/solver.py:320, 325 under solver.py:604-605) ----------
        def _device_assembly_possible(self):
            # every discretisation the single-GPU solver refreshes on the device: the P0-pressure pairs, the Scott-Vogelius pair
            # (its per-level states come from the replicated state by the sparse bary injection on the host, _winds; the full
            # grad-div term is state-independent) and the SUPG terms (element matrices of the rank's cells -- all cells that
            # touch a local node -- gathered into the rank's rows)
            return True

        def _setup_device_assembly(self):
            """Once per solver: every local level with owned rows gets the cells that touch its local nodes and the contributor
            lists of its local sparsity (alfi_level_set_assembly on a partitioned level: every term of the operator is formed
            on the device, cell by cell); the state of a Newton step is then uploaded per level -- local
            nodes and the ring of nodes around them, a few megabytes -- and the operators are rebuilt from it on the device.
            Ranks that hold only ghost copies of a level (the coarse side of the first distributed transfer) skip it: no patch
            and no product reads those rows."""
            from .lazy import LazyOperator
            dmg = self.dmg
            self._asm = []
            with self._on_stream():
                for dl, LL in zip(dmg.levels, dmg.local_levels):
                    p = LL.part
                    if p.nb_own == 0:
                        self._asm.append(None)
                        continue
                    L = self.levels[LL.level]
                    V = L.V
                    cells, cn, nodes = assembly_cells(V, p)
                    dl.set_assembly(V, LL.A.rowptr, LL.A.colidx, full_div=self.sv, cells=cells, cell_nodes=cn)
                    bcn = np.flatnonzero(V.bc_node_mask[p.nodes])              # Dirichlet nodes among ALL local nodes
                    dl.set_assembly_bc((bcn[:, None] * L.bs + np.arange(L.bs)).ravel())
                    if self.supg:
                        dl.set_supg(V, cells=cells)
                    self._asm.append((nodes, self.ctx.vec(dl.assembly_state_size())))
                L = self.levels[-1]
                self._dres = self.ctx.vec(dmg.n_loc)
                # the residual's divergence products with ALL columns of B (the Jacobian's B, Dirichlet columns zeroed, lives
                # in the saddle solver): the rank's cells over its local velocity dofs
                from . import hip
                rows, Bloc, _, _ = localize_pressure(self.B_raw, None, L.V.cell_nodes, dmg.fine.part, L.bs)
                self._res_rows = rows
                self._dB = hip.Csr(self.ctx, Bloc)
                self._dBT = hip.Csr(self.ctx, Bloc.T.tocsr())
                self._dp, self._dFp = self.ctx.vec(max(len(rows), 1)), self.ctx.vec(max(len(rows), 1))
                self._dwc = self.ctx.vec(dmg.n_loc)
                self._exch = None
                if not self.sv and self._want_device_state:
                    # the distributed device-resident state: (owned velocity | owned pressure) per rank, and the exchange that
                    # feeds every level's refresh from it
                    assert np.array_equal(self._res_rows, self.saddle.cells)
                    n = self.saddle.n
                    self._dz, self._dF, self._dd = self.ctx.vec(n + 1), self.ctx.vec(n + 1), self.ctx.vec(n + 1)
                    self._exch = StateExchange(dmg, self.levels, self.transfers, [None if a is None else a[0] for a in self._asm],
                                               dmg.device)
            self._asm_ready = True

        # -- the device-resident loop on partitions (nssolver.HipNavierStokesSolver._solve_on_device) ------------------------
        def _refresh_states(self):
            self._exch.refresh(self._dz, [None if a is None else a[1] for a in self._asm])

        def _push_state(self):
            part, sad = self.dmg.fine.part, self.saddle
            with self._on_stream():
                self._dz.set(np.concatenate([self._host_u[part.own_dofs()], self._host_p[sad.cells],
                                             np.zeros(self._dz.n - sad.n)]))

        def _fetch_state(self):
            """COLLECTIVE: every rank contributes its owned entries (``u`` / ``p`` after a solve must be read on all ranks)."""
            if self._device_newer:
                part, sad = self.dmg.fine.part, self.saddle
                with self._on_stream():
                    z = self._dz.get()
                u, p = np.zeros(self.n_u), np.zeros(self.n_p)
                for dofs, cells, zu, zp in self.dmg.comm.all_gather_object((part.own_dofs(), sad.cells, z[:sad.n_own],
                                                                            z[sad.n_own:sad.n])):
                    u[dofs] = zu
                    p[np.asarray(cells)] = zp
                self._host_u, self._host_p = u, p
                self._device_newer, self._device_current = False, True

        def _push_load(self):
            part = self.dmg.fine.part
            if getattr(self, "_dload", None) is None:
                self._dload = self.ctx.vec(max(self.dmg.n_own, 1))
            with self._on_stream():
                self._dload.set(np.ascontiguousarray(self._load[part.own_dofs()]) if self.dmg.n_own else np.zeros(1))

        def _zdot(self, x, y):
            with self._on_stream():
                return self.saddle.sad.dot(x, y)

        def _zsolve(self, b, x):
            with self._on_stream():
                return self.saddle.sad.solve(b, x, self.rtol, self.atol, self.params["ksp_max_it"], 30)

        def _zaxpy(self, y, x, a):
            with self._on_stream():
                self.ctx.axpy(y, x, a, n=self.saddle.n)

        def _shift_pressure(self):
            from . import hip
            sad = self.saddle
            if getattr(self, "_dvolz", None) is None:
                with self._on_stream():
                    self._dvolz = self.ctx.vec(np.concatenate([np.zeros(sad.n_own), self.vol[sad.cells], np.zeros(self._dz.n - sad.n)]))
                    self._dones = self.ctx.vec(np.ones(max(sad.np_own, 1)))
            c = self._zdot(self._dvolz, self._dz) / self.area
            with self._on_stream():
                self.ctx.axpy(self._dz, self._dones, -c, n=sad.np_own, y_off=sad.n_own)

        def _residual_on_device(self, adv):
            """The rank's rows of F(z) for the distributed state in ``_dz`` into ``_dF`` = (F_u owned | F_p owned): the state of
            the finest level's refresh comes through the exchange, the matrix-free product and the divergence products run over
            the rank's cells, the rank's share of B^T p is reverse-added onto the owners; nothing crosses to the host."""
            from . import hip
            dmg, fin, sad = self.dmg, self.dmg.levels[-1], self.saddle
            n_own, np_own = sad.n_own, sad.np_own
            st = self._asm[-1][1]
            with self._on_stream():
                self._refresh_states()
                fin.assemble_mult(self.nu, self.gamma, 0.5 * adv, st if adv else None, st, self._dres)
                if adv and self.supg:
                    fin.supg(self.nu, self.supg_weight, self.supg_magic, st, False, self._dres)
                dp = hip.view(self._dz, n_own, max(np_own, 1))
                self._dBT.mult(dp, self._dwc)
                fin.halo_reverse_add(self._dwc)
                self.ctx.copy(self._dF, self._dres, n=n_own)
                self.ctx.axpy(self._dF, self._dwc, 1.0, n=n_own)
                if self._load is not None:
                    self.ctx.axpy(self._dF, self._dload, -1.0, n=n_own)
                fin.zero_bc(self._dF)
                self._dB.mult(st, hip.view(self._dF, n_own, max(np_own, 1)))

        def _upload_states(self, u):
            for asm, w in zip(self._asm, self._winds(u)[self.dmg.lmin:]):
                if asm is not None:
                    asm[1].set(np.ascontiguousarray(w[asm[0]]).ravel())

        def _first_operators_on_device(self):
            """Every rank forms the Stokes operators of ITS rows on its device (the hierarchy was generated without operator
            values); patches and coarse grid factored."""
            with self._on_stream():
                for asm, dl in zip(self._asm, self.dmg.levels):
                    if asm is not None:
                        dl.assemble(self.nu, self.gamma, 0.0, None, True)
                self.dmg.sync()
            self.dmg.refactor(self.levels)
            self.dmg.sync()

        def _rediscretise_device(self, u, adv):
            import time
            t0 = time.time()
            with self._on_stream():
                if u is None:                 # the state lives on the devices: one exchange feeds every level's refresh
                    self._refresh_states()
                else:
                    self._upload_states(u)
                for asm, dl in zip(self._asm, self.dmg.levels):
                    if asm is None:
                        continue
                    if adv and self.supg:     # A = nu K + gamma D + N(w) + the linearised SUPG term, THEN the boundary conditions
                        dl.assemble_supg(self.nu, self.gamma, adv, asm[1], self.supg_weight, self.supg_magic, True)
                    else:
                        dl.assemble(self.nu, self.gamma, adv, asm[1] if adv else None, True)
                self.dmg.sync()
            t1 = time.time()
            for L in self.levels:
                L.nu = self.nu
            self.dmg.refactor(self.levels)
            self.dmg.sync()
            self.timings["assemble_s"] += t1 - t0
            self.timings["factor_s"] += time.time() - t1

        def _residual_device(self, u, p, adv):
            """The rank's rows of F_u = (nu K + gamma D + 1/2 N(u)) u + B^T p on the device -- one matrix-free product over the
            rank's cells (alfi_level_assemble_mult: the state carries the ring of nodes around the local ones, no exchange),
            the rank's cells' share of B^T p reverse-added onto the owners -- and its rows of F_p = B u; the pieces are then
            gathered (the Newton state is replicated)."""
            L = self.levels[-1]
            dmg, fin = self.dmg, self.dmg.levels[-1]
            nodes, st = self._asm[-1]
            n_own = dmg.n_own
            with self._on_stream():
                st.set(np.ascontiguousarray(u.reshape(-1, L.bs)[nodes]).ravel())
                fin.assemble_mult(self.nu, self.gamma, 0.5 * adv, st if adv else None, st, self._dres)
                if adv and self.supg:         # + the SUPG residual of the rank's rows, gathered on the device into the same vector
                    fin.supg(self.nu, self.supg_weight, self.supg_magic, st, False, self._dres)
                self._dp.set(np.ascontiguousarray(p[self._res_rows]) if len(self._res_rows) else np.zeros(1))
                self._dBT.mult(self._dp, self._dwc)
                fin.halo_reverse_add(self._dwc)
                self._dB.mult(st, self._dFp)
                f_own = self._dres.get()[:n_own] + self._dwc.get()[:n_own]
                fp_own = self._dFp.get()[:len(self._res_rows)]
            Fu, Fp = np.zeros(self.n_u), np.zeros(self.n_p)
            for dofs, f, rows, fp in dmg.comm.all_gather_object((dmg.fine.part.own_dofs(), f_own, self._res_rows, fp_own)):
                Fu[dofs] = f
                Fp[rows] = fp
            if self._load is not None:
                Fu -= self._load
            Fu[L.bc_dofs] = 0.0
            return Fu, Fp

        def _rediscretise(self, u, adv):
            """Every rank assembles ITS rows only: the level operators become lazy (alfi_amd.lazy.LazyOperator: sparsity now,
            values of a row subset on demand) and DistMultigrid.update cuts the rank's rows out of them -- one rank per mesh
            partition assembling its own cells, as in the reference (alfi/solver.py:604-605).  SUPG terms are assembled by
            the global host pass and keep the replicated path."""
            if self.device_assembly:
                return self._rediscretise_device(u, adv)
            if self.supg:
                self._supg_host_needs_global_values()
                return super()._rediscretise(u, adv)
            from .lazy import LazyOperator
            for L, w in zip(self.levels, self._winds(u)):
                V = L.V
                L.A = LazyOperator(V, L.A.rowptr, L.A.colidx, V.mesh.cell_geometry(), V.element.reference_tensors(), self.nu,
                                   self.gamma, adv, np.ascontiguousarray(w), full_div=self.sv)
                L.nu = self.nu
            self._push_operators()

        def residual(self, u, p, adv):
            """F(u, p) with every rank assembling ITS rows of (nu K + gamma D + 1/2 N(u)) u only -- one pass over its own
            cells instead of two global assemblies on every rank (the replicated host path of the base class, whose cost per
            rank GROWS with the number of ranks sharing the host's cores) -- and the pieces gathered.  SUPG keeps the
            replicated path."""
            if self.device_assembly:
                return self._residual_device(u, p, adv)
            if self.supg:
                self._supg_host_needs_global_values()
                return super().residual(u, p, adv)
            from . import _hostlib
            from .lazy import _take_rows, _row_map
            from .problem import BSR
            L = self.levels[-1]
            V, d = L.V, L.V.dim
            part = self.dmg.fine.part
            rows = np.asarray(part.own_nodes, dtype=np.int64)
            ptr, cols = _take_rows(L.A.rowptr, L.A.colidx, rows)
            ptr32 = ptr.astype(np.int32)
            g, vol = V.mesh.cell_geometry()
            wind = np.ascontiguousarray(u.reshape(-1, d))
            vals = _hostlib.assemble_bsr(V.cell_nodes, g, vol, V.element.reference_tensors(), d, ptr32, cols, nu=self.nu,
                                         gamma=0.0 if self.sv else self.gamma, gamma_full=self.gamma if self.sv else 0.0,
                                         adv=0.5 * adv, wind=wind if adv else None, row_map=_row_map(V.num_nodes, rows))
            f_own = BSR(len(rows), V.num_nodes, d, ptr32, cols, vals).to_scipy() @ u
            Fu = np.zeros(self.n_u)
            for dofs, f in self.dmg.comm.all_gather_object((part.own_dofs(), f_own)):
                Fu[dofs] = f
            Fu += self.B_raw.T @ p
            if self._load is not None:
                Fu -= self._load
            Fu[L.bc_dofs] = 0.0
            return Fu, self.B_raw @ u

        def _set_parameters(self):
            # transfers present on this rank link local levels lmin.. ; their (nu, gamma) follow the solver's
            ltr = self.dmg.local_transfers
            for T, dt in zip(ltr, self.dmg.mg.transfers):
                if T.nu != self.nu:
                    T.nu = self.nu
                    with self._on_stream():
                        dt.update(self.nu, self.gamma)
            for T in self.transfers:
                T.nu = self.nu
            self.saddle.update(self.nu, self.gamma)

        def _on_stream(self):
            import torch
            return torch.cuda.stream(self.dmg.stream)

        def _linear_solve(self, rhs):
            sad, part = self.saddle, self.dmg.fine.part
            loc = np.concatenate([rhs[:self.n_u][part.own_dofs()], rhs[self.n_u:][sad.cells]])
            x, its, rn = sad.solve(loc, self.rtol, self.atol, self.params["ksp_max_it"], 30)
            pieces = self.dmg.comm.all_gather_object((part.own_dofs(), sad.cells, x[:sad.n_own], x[sad.n_own:]))
            delta = np.zeros(self.n_u + self.n_p)
            for dofs, cells, xu, xp in pieces:
                delta[dofs] = xu
                delta[self.n_u + np.asarray(cells)] = xp
            return delta, its, rn

        def close(self):
            if getattr(self, "_asm_ready", False):
                self._dB.close()
                self._dBT.close()
                if self._exch is not None:
                    self._exch.close()
            self.saddle.close()
            self.dmg.close()

    return DistNavierStokesSolver


def DistNavierStokesSolver(*args, **kwargs):
    """Factory (the class derives from alfi_amd.nssolver.HipNavierStokesSolver, imported lazily)."""
    return _dist_ns_solver_class()(*args, **kwargs)
