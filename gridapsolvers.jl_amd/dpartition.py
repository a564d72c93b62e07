"""Row partition of ARBITRARY global operators by dof -> owner maps (multi-GPU).

`partition.py` builds the local operators of the structured Poisson hierarchy analytically (nothing of global size is ever
formed: 576^3 nodes on 8 GPUs).  This module is its general counterpart for operators that exist as global sparse matrices at
driver scale -- the vector-valued Q2 / discontinuous-P1 Stokes blocks of test/Applications/mpi/StokesGMG.jl, their velocity
hierarchy, transfer operators and patch tables: what GridapDistributed + PartitionedArrays give the reference, in the same
conventions (JacobiLinearSolvers.jl:29-56, PatchSolvers.jl:227-258):

  * every vector SPACE has an owner per dof; rank r's local numbering is [own (ascending global id) | ghost (by owner, then id)];
  * an operator's local part has the rows its row space owns on r and the columns of its column space in local numbering;
  * the ghosts of a space on r are the columns any of r's operator rows reference plus the dofs its owned patches touch;
  * exchange plans list, per neighbour, the owned entries to send (ascending global id -- the order the neighbour stores them in).

Host-side numpy / scipy only; nothing here is on the timed path."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from . import poisson as po

__all__ = ["Space", "partition_spaces", "local_operator", "local_patches", "OverlapSpace", "FoldedSpace", "stack_rows"]


class Space:
    """One vector space on every rank: owner per global dof, then (after `partition_spaces`) per rank the own / ghost global ids,
    the global -> local map and the exchange plan."""

    def __init__(self, name, owner, nranks):
        self.name = name
        self.owner = np.asarray(owner, dtype=np.int64)
        self.n = self.owner.size
        self.nranks = int(nranks)
        assert self.owner.min(initial=0) >= 0 and self.owner.max(initial=0) < nranks
        self.own = [np.nonzero(self.owner == r)[0].astype(np.int64) for r in range(nranks)]
        self.need = [set() for _ in range(nranks)]          # ghost candidates collected from operators / patches
        self.ghost = None

    def require(self, rank, gids):
        g = np.asarray(gids, dtype=np.int64)
        self.need[rank].update(g[self.owner[g] != rank].tolist())

    def finalize(self):
        self.ghost, self.g2l, self.plan = [], [], []
        for r in range(self.nranks):
            g = np.array(sorted(self.need[r]), dtype=np.int64)
            order = np.lexsort((g, self.owner[g])) if g.size else np.zeros(0, dtype=np.int64)
            g = g[order]
            self.ghost.append(g)
            m = -np.ones(self.n, dtype=np.int64)
            m[self.own[r]] = np.arange(self.own[r].size)
            m[g] = self.own[r].size + np.arange(g.size)
            self.g2l.append(m)
        for r in range(self.nranks):
            nbr, snd_ptr, rcv_ptr, snd = [], [0], [0], []
            for q in range(self.nranks):
                if q == r:
                    continue
                mine_at_q = self.ghost[q][self.owner[self.ghost[q]] == r]       # ascending id: q stores them in this order
                n_rcv = int(np.count_nonzero(self.owner[self.ghost[r]] == q))
                if mine_at_q.size == 0 and n_rcv == 0:
                    continue
                nbr.append(q)
                snd.append(self.g2l[r][mine_at_q])
                snd_ptr.append(snd_ptr[-1] + mine_at_q.size)
                rcv_ptr.append(rcv_ptr[-1] + n_rcv)
            assert rcv_ptr[-1] == self.ghost[r].size
            self.plan.append(dict(nbr_rank=np.asarray(nbr, dtype=np.int32), snd_ptr=np.asarray(snd_ptr, dtype=np.int64),
                                  snd_idx=np.concatenate(snd).astype(np.int64) if snd else np.zeros(0, np.int64),
                                  rcv_ptr=np.asarray(rcv_ptr, dtype=np.int64)))

    # views a rank works with
    def n_own(self, r):
        return int(self.own[r].size)

    def n_ghost(self, r):
        return int(self.ghost[r].size)

    def local_gid(self, r):
        return np.concatenate([self.own[r], self.ghost[r]])


def partition_spaces(spaces, operators, patches=()):
    """Collect the ghosts every rank needs and freeze the local numberings.

    spaces    : list of Space
    operators : list of (A_global (scipy sparse), row_space, col_space)
    patches   : list of (patch_ptr, patch_dofs (global ids), space, patch_owner[npatch])"""
    for A, rs, cs in operators:
        A = A.tocsr()
        for r in range(rs.nranks):
            rows = rs.own[r]
            if rows.size:
                cs.require(r, A[rows].indices)
    for pp, pd, spc, powner in patches:
        for r in range(spc.nranks):
            mine = np.nonzero(powner == r)[0]
            if mine.size:
                cs_ = np.concatenate([pd[pp[p]:pp[p + 1]] for p in mine]) if mine.size else np.zeros(0, np.int64)
                spc.require(r, cs_)
    for s in spaces:
        s.finalize()


def local_operator(A, rs, cs, rank):
    """Rows rs owns on `rank`, columns in cs's [own | ghost] numbering (ascending column order per row, the order a sequential
    mat-vec sums in -- ghosts come after the owned columns)."""
    A = A.tocsr()
    L = A[rs.own[rank]].tocoo()
    cols = cs.g2l[rank][L.col]
    assert (cols >= 0).all(), "operator references a dof that is neither owned nor ghost"
    M = sp.csr_matrix((L.data, (L.row, cols)), shape=(rs.n_own(rank), cs.n_own(rank) + cs.n_ghost(rank)))
    M.sort_indices()
    return po.CSR(M.shape, M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data)


def local_patches(pp, pd, spc, powner, rank, A_global=None):
    """Patches owned by `rank` in local numbering, in the serial order; with A_global also their dense matrices (column-major,
    concatenated: gmg_set_smoother_patch_matrices -- a rank's local rows cannot supply the blocks of ghost-reaching patches)."""
    mine = np.nonzero(powner == rank)[0]
    ptr = np.zeros(mine.size + 1, dtype=np.int64)
    loc, glob, blocks = [], [], []
    Ag = A_global.tocsr() if A_global is not None else None
    for k, p in enumerate(mine):
        g = pd[pp[p]:pp[p + 1]].astype(np.int64)
        l = spc.g2l[rank][g]
        assert (l >= 0).all()
        loc.append(l); glob.append(g)
        ptr[k + 1] = ptr[k] + g.size
        if Ag is not None:
            blocks.append(Ag[g][:, g].toarray().reshape(-1, order="F"))
    cat = (lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt))
    return ptr, cat(loc, np.int64), cat(glob, np.int64), (cat(blocks, np.float64) if Ag is not None else None)


class OverlapSpace:
    """A vector space in the OVERLAPPING layout (gmg_set_partition_overlap) for operators that exist as global sparse matrices --
    the general counterpart of partition._OverlapGeom, e.g. the vector-valued Q2 velocity levels of the Stokes hierarchy
    (test/Applications/mpi/StokesGMG.jl:5-12; both components of a node share its coordinates).

    coords[i] = integer node coordinates of global dof i; a rank owns the dofs of a node box.  Its local entries are ALL dofs whose
    node lies in that box extended by `layers` node layers, numbered by ascending global id -- owned and ghost entries interleaved,
    so an owned row of a local operator is summed in the order of the single-GPU run.  layers = depth * reach with
    reach = order for Richardson(Jacobi), 3 order - 2 for Richardson(PatchSolver) on vertex stars (partition._OverlapGeom)."""

    def __init__(self, name, owner, coords, nranks, layers):
        self.name, self.nranks, self.layers = name, int(nranks), int(layers)
        self.owner = np.asarray(owner, dtype=np.int64)
        self.n = self.owner.size
        X = np.asarray(coords, dtype=np.int64).reshape(self.n, -1)
        self.own = [np.nonzero(self.owner == r)[0].astype(np.int64) for r in range(nranks)]
        self.ext, self.g2l, self.is_own = [], [], []
        for r in range(nranks):
            lo, hi = X[self.own[r]].min(axis=0) - self.layers, X[self.own[r]].max(axis=0) + self.layers
            e = np.nonzero(((X >= lo) & (X <= hi)).all(axis=1))[0].astype(np.int64)      # ascending global id
            assert np.isin(self.own[r], e).all()
            m = -np.ones(self.n, dtype=np.int64)
            m[e] = np.arange(e.size)
            self.ext.append(e); self.g2l.append(m); self.is_own.append(self.owner[e] == r)
        self.plan = []
        for r in range(nranks):
            nbr, snd_ptr, rcv_ptr, snd, rcv = [], [0], [0], [], []
            for q in range(nranks):
                if q == r:
                    continue
                mine_at_q = self.ext[q][self.owner[self.ext[q]] == r]         # what q holds of my owned dofs, ascending id
                theirs = self.ext[r][self.owner[self.ext[r]] == q]
                if mine_at_q.size == 0 and theirs.size == 0:
                    continue
                nbr.append(q); snd.append(self.g2l[r][mine_at_q]); rcv.append(self.g2l[r][theirs])
                snd_ptr.append(snd_ptr[-1] + mine_at_q.size); rcv_ptr.append(rcv_ptr[-1] + theirs.size)
            cat = (lambda xs: np.concatenate(xs).astype(np.int64) if xs else np.zeros(0, np.int64))
            assert rcv_ptr[-1] == self.ext[r].size - self.own[r].size
            self.plan.append(dict(nbr_rank=np.asarray(nbr, dtype=np.int32), snd_ptr=np.asarray(snd_ptr, dtype=np.int64), snd_idx=cat(snd),
                                  rcv_ptr=np.asarray(rcv_ptr, dtype=np.int64), rcv_idx=cat(rcv)))

    def n_local(self, r):
        return int(self.ext[r].size)

    def n_own(self, r):
        return int(self.own[r].size)

    def own_idx(self, r):
        return self.g2l[r][self.own[r]]

    def square(self, A, rank):
        """A[ext, ext]: rows of ghost entries are the true global rows restricted to the local columns"""
        e = self.ext[rank]
        M = A.tocsr()[e][:, e].tocsr()
        M.sort_indices()
        return po.CSR(M.shape, M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data)

    def patches(self, pp, pd, rank):
        """every patch whose dofs ALL lie among the local entries, in the serial order, local numbering (blocks A[p,p] are then exact
        in the local matrix: gmg_set_smoother_patch / gmg_set_prolongation_patch_correction, no caller-assembled matrices)"""
        m = self.g2l[rank]
        ptr, loc = [0], []
        for p in range(pp.size - 1):
            l = m[pd[pp[p]:pp[p + 1]].astype(np.int64)]
            if l.size and (l >= 0).all():
                loc.append(l); ptr.append(ptr[-1] + l.size)
        return np.asarray(ptr, dtype=np.int64), (np.concatenate(loc).astype(np.int64) if loc else np.zeros(0, np.int64))


def sliced_operator(M, rows, colmap=None, keep_row=None, ncols=None, strict_rows=None):
    """Rows `rows` (global ids, in that order) of the global operator M with the columns renumbered through `colmap` (global ->
    local id, -1 = not held by this rank: such entries are DROPPED -- in the overlapping layout they belong to rows of ghost layers the
    next exchange refreshes; colmap None = global columns as they are).  keep_row[i] False empties row i (R in the overlapping layout:
    the rows of non-owned coarse entries, whose residual arrives with the exchange that opens the smoothing block).  strict_rows: a
    mask of rows that must not lose an entry (the owned ones)."""
    S = M.tocsr()[np.asarray(rows, dtype=np.int64)].tocoo()
    r, c, v = S.row, S.col.astype(np.int64), S.data
    if colmap is not None:
        c = colmap[c]
    keep = c >= 0
    if strict_rows is not None:
        assert keep[np.asarray(strict_rows)[r]].all(), "an owned row references a dof outside the local entries: more ghost layers needed"
    if keep_row is not None:
        keep &= np.asarray(keep_row)[r]
    nc = int(ncols if ncols is not None else M.shape[1])
    L = sp.csr_matrix((v[keep], (r[keep], c[keep])), shape=(len(rows), nc))
    L.sort_indices()
    return po.CSR(L.shape, L.indptr.astype(np.int64), L.indices.astype(np.int32), L.data)


# ----------------------------------------------------------------------------------------------------------------------
# folding a partition onto ONE rank (gmg_comm_set_loopback; partition.fold_ranks is the structured-grid twin)
# ----------------------------------------------------------------------------------------------------------------------
def stack_rows(mats, maps, ncols):
    """rows of the ranks' matrices stacked in rank order, columns through the rank's local -> folded map (None: unchanged)"""
    ptr, idx, val, base = [np.zeros(1, dtype=np.int64)], [], [], 0
    for M, m in zip(mats, maps):
        ptr.append(M.ptr[1:] + base)
        base += M.nnz
        idx.append(M.idx.astype(np.int64) if m is None else m[M.idx])
        val.append(M.val)
    return po.CSR((sum(M.shape[0] for M in mats), ncols), np.concatenate(ptr), np.concatenate(idx), np.concatenate(val))


class FoldedSpace:
    """A partitioned Space seen from ONE rank whose every neighbour is itself: the ranks' owned dofs stacked in rank order, their ghosts
    stacked behind them; one message per (receiving rank, neighbour) pair in the order the folded ghosts are stored, its send list = what
    the neighbour sends, with the neighbour's offset -- so message k's send segment lands in receive segment k (how a self send / receive
    pairs up).  cmap[r] maps rank r's [own | ghost] numbering into the folded one; every neighbour is named rank 1."""

    def __init__(self, S):
        W = S.nranks
        no = np.array([S.n_own(r) for r in range(W)], dtype=np.int64)
        ng = np.array([S.n_ghost(r) for r in range(W)], dtype=np.int64)
        off, goff = np.concatenate([[0], np.cumsum(no)]), np.concatenate([[0], np.cumsum(ng)])
        self.n_own, self.n_ghost = int(off[-1]), int(goff[-1])
        self.cmap = [np.concatenate([off[r] + np.arange(no[r], dtype=np.int64), off[-1] + goff[r] + np.arange(ng[r], dtype=np.int64)]) for r in range(W)]
        self.own_gid = np.concatenate([S.own[r] for r in range(W)])
        snd, sp, rp = [], [0], [0]
        for a in range(W):
            pa_ = S.plan[a]
            for k, q in enumerate(pa_["nbr_rank"]):
                pq = S.plan[int(q)]
                kq = int(np.nonzero(pq["nbr_rank"] == a)[0][0])
                sseg = off[int(q)] + pq["snd_idx"][pq["snd_ptr"][kq]:pq["snd_ptr"][kq + 1]]
                nr = int(pa_["rcv_ptr"][k + 1] - pa_["rcv_ptr"][k])
                assert sseg.size == nr
                snd.append(sseg); sp.append(sp[-1] + sseg.size); rp.append(rp[-1] + nr)
        self.nbr_rank = np.ones(len(snd), dtype=np.int32)
        self.snd_ptr, self.rcv_ptr = np.asarray(sp, dtype=np.int64), np.asarray(rp, dtype=np.int64)
        self.snd_idx = np.concatenate(snd).astype(np.int64) if snd else np.zeros(0, np.int64)
        assert self.rcv_ptr[-1] == self.n_ghost

    def plan_into(self, obj):
        obj.n_own, obj.n_ghost = self.n_own, self.n_ghost
        obj.nbr_rank, obj.snd_ptr, obj.snd_idx, obj.rcv_ptr = self.nbr_rank, self.snd_ptr, self.snd_idx, self.rcv_ptr
        return obj
