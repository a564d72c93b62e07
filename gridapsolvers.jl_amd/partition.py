"""Box row-partition of the structured Poisson hierarchy (multi-GPU).

Mirrors what PartitionedArrays / GridapDistributed give the reference: per rank and
per level a local matrix whose rows are the OWNED dofs and whose columns are numbered
own-first-then-ghost (`own_values`/`partition` convention, JacobiLinearSolvers.jl:29-56),
plus the neighbour lists a `consistent!` (owner -> ghost copy, PatchSolvers.jl:231) needs.

Ranks form a px x py x pz grid; rank r owns the nodes  c0 < i <= c1  of its cell box in
every direction, on every level (coarse node I coincides with fine node 2I, so ownership
nests and restriction / prolongation need only the one-layer halo of the 27-point
operator).  Ghosts are ordered by (owner rank, global lexicographic id); send lists are
sorted by global id, so both sides of an exchange agree without negotiation.

Host-side numpy only; nothing here is on the timed path.
"""
from __future__ import annotations

import numpy as np

from . import poisson as po

__all__ = ["rank_grid", "LocalLevel", "build_local_hierarchy", "global_cells", "local_vertex_star_patches", "overlap_hints"]


def rank_grid(nranks, dim=3):
    """1x1x1, 2x1x1, 2x2x1, 2x2x2 for 1/2/4/8 GPUs; generic power-of-two otherwise."""
    g = [1, 1, 1]
    k = 0
    n = int(nranks)
    if n < 1 or (n & (n - 1)):
        raise ValueError("number of ranks must be a power of two")
    while n > 1:
        g[k % dim] *= 2
        n //= 2
        k += 1
    return tuple(g)


def global_cells(cells_per_rank, grid):
    return tuple(int(c) * int(g) for c, g in zip(cells_per_rank, grid))


def _coords(rank, grid):
    px, py, pz = grid
    return rank % px, (rank // px) % py, rank // (px * py)


def _rank_of(cx, cy, cz, grid):
    return cx + grid[0] * (cy + grid[1] * cz)


class LocalLevel:
    """One level on one rank: local operators in [own | ghost] column numbering + exchange plan."""

    def __init__(self):
        self.A = self.P = self.R = None
        self.n_own = self.n_ghost = 0
        self.nbr_rank = np.zeros(0, np.int32)
        self.snd_ptr = np.zeros(1, np.int64)
        self.snd_idx = np.zeros(0, np.int64)
        self.rcv_ptr = np.zeros(1, np.int64)
        self.own_gid = None      # global lexicographic free-dof id of every owned dof
        self.ghost_gid = None
        self.replicated = False
        # overlapping layout (gmg_set_partition_overlap): one local numbering = the box extended by `depth` node layers, lexicographic
        self.overlap = False
        self.depth = 0
        self.n_local = 0
        self.rcv_idx = np.zeros(0, np.int64)
        self.own_idx = None      # local ids of the owned entries (ascending global id)
        self.local_gid = None    # global id of every local entry


def _axis_ranges(ncell_global, order, nparts, coord):
    """Owned node range (lo, hi] -> free nodes lo+1..hi (clipped), and the extended range."""
    per = ncell_global // nparts
    c0, c1 = coord * per, (coord + 1) * per
    nlast = order * ncell_global - 1                  # last free node
    lo, hi = order * c0 + 1, min(order * c1, nlast)   # owned free nodes lo..hi inclusive
    # halo width in nodes: the operator couples nodes up to `order` apart; the restriction R = P^T of the quadratic
    # element reaches fine nodes up to 3 away from a coarse node (quarter points of both adjacent coarse cells)
    w = 1 if order == 1 else 3
    elo, ehi = max(lo - w, 1), min(hi + w, nlast)
    return lo, hi, elo, ehi


def _axis_tables_local(ncell_global, order, lo, hi, elo, ehi, active, length=1.0):
    """1-D padded row tables for owned rows lo..hi with columns in extended-box numbering."""
    if not active:
        return 1, np.zeros((1, 1), dtype=np.int64), np.zeros((1, 1)), np.ones((1, 1))
    K, M, S = po._assemble_1d(ncell_global, order, length)
    nn = order * ncell_global + 1
    col_map = -np.ones(nn, dtype=np.int64)
    col_map[elo:ehi + 1] = np.arange(ehi - elo + 1)
    rows = np.arange(lo, hi + 1)
    cols, (kv, mv) = po._padded_rows(S, [K, M], rows, col_map)
    return ehi - elo + 1, cols, kv, mv


def _interp_tables_local(nc_coarse_global, order, flo, fhi, celo, cehi, active):
    """1-D prolongation rows for fine owned nodes flo..fhi, columns = coarse extended numbering."""
    if not active:
        return 1, np.zeros((1, 1), dtype=np.int64), np.ones((1, 1))
    P, S = po._interp_1d(nc_coarse_global, order)
    nH = P.shape[1]
    col_map = -np.ones(nH, dtype=np.int64)
    col_map[celo:cehi + 1] = np.arange(cehi - celo + 1)
    col_map[0] = -1
    col_map[nH - 1] = -1
    c, (v,) = po._padded_rows(S, [P], np.arange(flo, fhi + 1), col_map)
    return cehi - celo + 1, c, v


def _restr_tables_local(nc_coarse_global, order, clo, chi, felo, fehi, active):
    """1-D restriction (= interpolation transposed) rows for coarse owned nodes, fine extended columns."""
    if not active:
        return 1, np.zeros((1, 1), dtype=np.int64), np.ones((1, 1))
    P, S = po._interp_1d(nc_coarse_global, order)
    R, SR = P.T.copy(), S.T.copy()
    nh = P.shape[0]
    col_map = -np.ones(nh, dtype=np.int64)
    col_map[felo:fehi + 1] = np.arange(fehi - felo + 1)
    col_map[0] = -1
    col_map[nh - 1] = -1
    c, (v,) = po._padded_rows(SR, [R], np.arange(clo, chi + 1), col_map)
    return fehi - felo + 1, c, v


class _LevelGeom:
    """Index bookkeeping of one level on one rank."""

    def __init__(self, cells_global, order, grid, rank, d):
        self.cells, self.order, self.grid, self.rank, self.d = cells_global, order, grid, rank, d
        co = _coords(rank, grid)
        self.rng = []
        for k in range(3):
            if k < d:
                self.rng.append(_axis_ranges(cells_global[k], order, grid[k], co[k]))
            else:
                self.rng.append((0, 0, 0, 0))
        self.nfree = [order * cells_global[k] - 1 if k < d else 1 for k in range(3)]
        own = [r[1] - r[0] + 1 for r in self.rng]
        ext = [r[3] - r[2] + 1 for r in self.rng]
        self.own_shape, self.ext_shape = own, ext
        # global node coordinates of the extended box (z,y,x order arrays)
        ex = [np.arange(r[2], r[3] + 1) for r in self.rng]
        Z, Y, X = np.meshgrid(ex[2], ex[1], ex[0], indexing="ij")
        is_own = ((X >= self.rng[0][0]) & (X <= self.rng[0][1]) & (Y >= self.rng[1][0]) & (Y <= self.rng[1][1])
                  & (Z >= self.rng[2][0]) & (Z <= self.rng[2][1]))
        # owner rank of every extended node
        def owner_1d(i, k):
            if k >= d:
                return np.zeros_like(i)
            per = order * (cells_global[k] // grid[k])
            return np.minimum((i - 1) // per, grid[k] - 1)
        owner = _rank_of(owner_1d(X, 0), owner_1d(Y, 1), owner_1d(Z, 2), grid)
        off = [1 if k < d else 0 for k in range(3)]
        gid = (X - off[0]) + self.nfree[0] * ((Y - off[1]) + self.nfree[1] * (Z - off[2]))
        self.is_own = is_own.reshape(-1)
        self.owner = owner.reshape(-1).astype(np.int64)
        self.gid = gid.reshape(-1).astype(np.int64)
        n_ext = self.is_own.size
        own_ext = np.nonzero(self.is_own)[0]                       # lexicographic inside the box
        gh_ext = np.nonzero(~self.is_own)[0]
        order_g = np.lexsort((self.gid[gh_ext], self.owner[gh_ext]))   # by owner, then global id
        gh_ext = gh_ext[order_g]
        self.n_own, self.n_ghost = own_ext.size, gh_ext.size
        self.ext2loc = np.empty(n_ext, dtype=np.int64)
        self.ext2loc[own_ext] = np.arange(self.n_own)
        self.ext2loc[gh_ext] = self.n_own + np.arange(self.n_ghost)
        self.own_gid = self.gid[own_ext]
        self.ghost_gid = self.gid[gh_ext]
        self.ghost_owner = self.owner[gh_ext]

    def remap(self, M):
        """Columns from extended-box numbering to [own | ghost]."""
        e2l = self.ext2loc.astype(np.int32) if self.ext2loc.size < 2 ** 31 - 1 else self.ext2loc     # int32 gather: half the traffic, no second conversion
        return po.CSR((M.shape[0], self.n_own + self.n_ghost), M.ptr, np.take(e2l, M.idx), M.val)


class _OverlapGeom:
    """One level on one rank in the overlapping layout: the owned node box extended by `layers` node layers per direction (clipped at
    the Dirichlet boundary), numbered lexicographically (x fastest) -- owned and ghost entries interleaved.  Exact rows exist for
    every local entry except the outermost `order` layers, so `depth` smoothing sweeps can run between two exchanges when
    layers = depth * reach, reach = how far one sweep carries an error inwards: `order` nodes for Richardson(Jacobi) (the operator couples
    nodes up to `order` apart), 3 * order - 2 for Richardson(PatchSolver) on vertex stars (a star reaches order - 1 nodes from its
    vertex, so dx_i needs r within 2 (order - 1) of i, and r_i -= (A dx)_i another `order`)."""

    def __init__(self, cells_global, grid, rank, d, depth, order=1, reach=None):
        self.cells, self.grid, self.rank, self.d, self.depth, self.order = cells_global, grid, rank, d, int(depth), int(order)
        self.reach = int(order if reach is None else reach)
        self.layers = self.depth * self.reach
        co = _coords(rank, grid)
        self.rng = []
        for k in range(3):
            if k < d:
                lo, hi, _, _ = _axis_ranges(cells_global[k], order, grid[k], co[k])
                nlast = order * cells_global[k] - 1
                self.rng.append((lo, hi, max(lo - self.layers, 1), min(hi + self.layers, nlast)))
            else:
                self.rng.append((0, 0, 0, 0))
        self.nfree = [order * cells_global[k] - 1 if k < d else 1 for k in range(3)]
        self.ext_shape = [r[3] - r[2] + 1 for r in self.rng]
        self.own_shape = [r[1] - r[0] + 1 for r in self.rng]
        self.n_local = int(np.prod(self.ext_shape))
        ex = [np.arange(r[2], r[3] + 1) for r in self.rng]
        Z, Y, X = np.meshgrid(ex[2], ex[1], ex[0], indexing="ij")
        own = ((X >= self.rng[0][0]) & (X <= self.rng[0][1]) & (Y >= self.rng[1][0]) & (Y <= self.rng[1][1])
               & (Z >= self.rng[2][0]) & (Z <= self.rng[2][1])).reshape(-1)
        off = [1 if k < d else 0 for k in range(3)]
        self.gid = ((X - off[0]) + self.nfree[0] * ((Y - off[1]) + self.nfree[1] * (Z - off[2]))).reshape(-1).astype(np.int64)
        self.is_own = own
        self.own_idx = np.nonzero(own)[0].astype(np.int64)
        self.n_own = self.own_idx.size
        self.own_gid = self.gid[self.own_idx]

    def vertex_star_patches(self):
        """(patch_ptr, patch_dofs): every vertex star whose dofs ALL lie in this rank's extended box, in lexicographic vertex order,
        dofs ascending, in the LOCAL numbering of the overlapping layout -- the serial `poisson.vertex_star_patches` restricted to
        the box.  Blocks A[p,p] of such patches are exact in the local matrix (its rows only drop columns outside the box), so the
        patch smoother of an overlapping level needs neither caller-assembled matrices nor assemble!: stars cut by the box edge are
        left out, which only touches the layers the exchange refreshes."""
        order, d = self.order, self.d
        per_axis = []
        for k in range(3):
            if k >= d:
                per_axis.append([np.zeros(1, dtype=np.int64)]); continue
            nlast = order * self.cells[k] - 1
            lst = []
            for v in range(self.cells[k] + 1):               # boundary vertices too: their stars hold the free nodes next to them
                node = order * v
                a, b = max(node - (order - 1), 1), min(node + (order - 1), nlast)
                if b < a:
                    continue                                 # (order 1: a boundary vertex has no free dof)
                if a >= self.rng[k][2] and b <= self.rng[k][3]:
                    lst.append(np.arange(a, b + 1, dtype=np.int64) - self.rng[k][2])
            per_axis.append(lst)
        ex, ey = self.ext_shape[0], self.ext_shape[1]
        ptr, dofs = [0], []
        for lz in per_axis[2]:
            for ly in per_axis[1]:
                for lx in per_axis[0]:
                    idx = (lz[:, None, None] * (ey * ex) + ly[None, :, None] * ex + lx[None, None, :]).reshape(-1)
                    dofs.append(idx)
                    ptr.append(ptr[-1] + idx.size)
        return np.asarray(ptr, dtype=np.int64), (np.concatenate(dofs).astype(np.int64) if dofs else np.zeros(0, np.int64))

    def _sub_box(self, box):
        """local ids (lexicographic) of the global node box [(lo,hi)]*3 intersected with this rank's extended box"""
        sub = []
        for k in range(3):
            if k >= self.d:
                sub.append(np.zeros(1, dtype=np.int64)); continue
            a, b = max(box[k][0], self.rng[k][2]), min(box[k][1], self.rng[k][3])
            sub.append(np.arange(a, b + 1, dtype=np.int64) - self.rng[k][2] if b >= a else np.zeros(0, dtype=np.int64))
        ex, ey = self.ext_shape[0], self.ext_shape[1]
        return (sub[2][:, None, None] * (ey * ex) + sub[1][None, :, None] * ex + sub[0][None, None, :]).reshape(-1)

    def plan(self):
        """(nbr, snd_ptr, snd_idx, rcv_ptr, rcv_idx): with rank q I exchange  my owned box ∩ q's extended box  (sent) and
        q's owned box ∩ my extended box  (received), both enumerated lexicographically = ascending global id on either side."""
        grid, d = self.grid, self.d
        nranks = int(np.prod(grid))
        nbr, snd_ptr, rcv_ptr, snd, rcv = [], [0], [0], [], []
        for q in range(nranks):
            if q == self.rank:
                continue
            qc = _coords(q, grid)
            q_own, q_ext = [], []
            for k in range(3):
                if k >= d:
                    q_own.append((0, 0)); q_ext.append((0, 0)); continue
                lo, hi, _, _ = _axis_ranges(self.cells[k], self.order, grid[k], qc[k])
                q_own.append((lo, hi)); q_ext.append((max(lo - self.layers, 1), min(hi + self.layers, self.order * self.cells[k] - 1)))
            my_own = [(self.rng[k][0], self.rng[k][1]) for k in range(3)]
            s_box = [(max(my_own[k][0], q_ext[k][0]), min(my_own[k][1], q_ext[k][1])) for k in range(3)]
            sidx = self._sub_box(s_box)
            ridx = self._sub_box(q_own)
            if sidx.size == 0 and ridx.size == 0:
                continue
            nbr.append(q); snd.append(sidx); rcv.append(ridx)
            snd_ptr.append(snd_ptr[-1] + sidx.size); rcv_ptr.append(rcv_ptr[-1] + ridx.size)
        rcv_idx = np.concatenate(rcv).astype(np.int64) if rcv else np.zeros(0, np.int64)
        assert rcv_idx.size == self.n_local - self.n_own and not self.is_own[rcv_idx].any(), "every ghost must be received from exactly one neighbour"
        return (np.asarray(nbr, dtype=np.int32), np.asarray(snd_ptr, dtype=np.int64),
                np.concatenate(snd).astype(np.int64) if snd else np.zeros(0, np.int64), np.asarray(rcv_ptr, dtype=np.int64), rcv_idx)


def local_vertex_star_patches(cells_global, order, grid, rank):
    """Vertex-star patches OWNED by `rank` on one level, for the distributed patch smoother (PatchSolvers.jl:227-258).

    Every mesh vertex -- and with it its patch -- belongs to exactly one rank: the owner of the vertex node (a Dirichlet
    boundary vertex goes with the adjacent free node).  A patch's dofs are the free dofs strictly inside the star of
    the vertex; those of an interface vertex reach into the neighbour's dofs, which are GHOST dofs here -- hence
    consistent!(b) before the local solves and assemble!(x) after.  Returns (patch_ptr, patch_dofs in this rank's
    [own | ghost] numbering, the same dofs as global free-dof ids), patches in lexicographic vertex order, dofs ascending
    by global id inside a patch (the serial `poisson.vertex_star_patches` order restricted to the owned vertices)."""
    nc = tuple(int(c) for c in cells_global)
    d = len(nc)
    nc3 = nc + (1,) * (3 - d)
    grid = tuple(grid) + (1,) * (3 - len(grid))
    g = _LevelGeom(nc3, order, grid, rank, d)
    ext_shape = g.ext_shape                                   # extended box, x fastest

    def axis_lists(k):
        """per owned vertex of axis k: its free nodes (global node indices)"""
        if k >= d:
            return [np.zeros(1, dtype=np.int64)]
        lo, hi = g.rng[k][0], g.rng[k][1]
        nlast = order * nc3[k] - 1
        out = []
        for v in range(nc3[k] + 1):
            node = order * v
            key = min(max(node, 1), nlast)                    # boundary vertices go with the adjacent free node
            if not (lo <= key <= hi):
                continue
            a, b = max(node - (order - 1), 1), min(node + (order - 1), nlast)
            out.append(np.arange(a, b + 1, dtype=np.int64))
        return out

    ax, ay, az = axis_lists(0), axis_lists(1), axis_lists(2)
    off = [1 if k < d else 0 for k in range(3)]
    ptr, loc, glob = [0], [], []
    for lz in az:
        for ly in ay:
            for lx in ax:
                if len(lx) == 0 or len(ly) == 0 or len(lz) == 0:
                    ptr.append(ptr[-1]); continue
                ex = ((lz[:, None, None] - g.rng[2][2]) * (ext_shape[1] * ext_shape[0]) + (ly[None, :, None] - g.rng[1][2]) * ext_shape[0]
                      + (lx[None, None, :] - g.rng[0][2])).reshape(-1)
                gid = ((lx[None, None, :] - off[0]) + g.nfree[0] * ((ly[None, :, None] - off[1]) + g.nfree[1] * (lz[:, None, None] - off[2]))).reshape(-1)
                loc.append(g.ext2loc[ex]); glob.append(gid)
                ptr.append(ptr[-1] + ex.size)
    loc = np.concatenate(loc) if loc else np.zeros(0, dtype=np.int64)
    glob = np.concatenate(glob) if glob else np.zeros(0, dtype=np.int64)
    return np.asarray(ptr, dtype=np.int64), loc.astype(np.int32), glob.astype(np.int64)


def _exchange_plan(me):
    """snd/rcv lists of one rank on one level, computed analytically: what I send to rank q is
    the part of my owned box inside q's extended box, enumerated lexicographically (= ascending
    global id, the order in which q stores its ghosts owned by me)."""
    grid, d = me.grid, me.d
    co = _coords(me.rank, grid)
    nb = []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx == dy == dz == 0:
                    continue
                q = (co[0] + dx, co[1] + dy, co[2] + dz)
                if all(0 <= q[k] < grid[k] for k in range(3)):
                    nb.append(_rank_of(*q, grid))
    nb = sorted(set(nb))
    snd_ptr, rcv_ptr, snd_idx, nbr = [0], [0], [], []
    for q in nb:
        qc = _coords(q, grid)
        sub = []
        for k in range(3):
            if k >= d:
                sub.append(np.zeros(1, dtype=np.int64)); continue
            _, _, qelo, qehi = _axis_ranges(me.cells[k], me.order, grid[k], qc[k])
            lo, hi = me.rng[k][0], me.rng[k][1]
            a, b = max(lo, qelo), min(hi, qehi)
            sub.append(np.arange(a, b + 1, dtype=np.int64) - lo if b >= a else np.zeros(0, dtype=np.int64))
        ox, oy = me.own_shape[0], me.own_shape[1]
        s = (sub[2][:, None, None] * (oy * ox) + sub[1][None, :, None] * ox + sub[0][None, None, :]).reshape(-1)
        nrecv = int(np.count_nonzero(me.ghost_owner == q))
        if s.size == 0 and nrecv == 0:
            continue
        nbr.append(q)
        snd_idx.append(s)
        snd_ptr.append(snd_ptr[-1] + s.size)
        rcv_ptr.append(rcv_ptr[-1] + nrecv)
    assert rcv_ptr[-1] == me.n_ghost, "every ghost must be received from exactly one neighbour"
    return (np.asarray(nbr, dtype=np.int32), np.asarray(snd_ptr, dtype=np.int64),
            np.concatenate(snd_idx).astype(np.int64) if snd_idx else np.zeros(0, np.int64),
            np.asarray(rcv_ptr, dtype=np.int64))


def _restr_tables_ext(nc_coarse_global, order, c_elo, c_ehi, clo, chi, felo, fehi, active):
    """1-D restriction rows for the coarse EXTENDED range c_elo..c_ehi (overlapping layout): the rows of owned coarse nodes
    clo..chi are complete, the rows of ghost coarse nodes are empty (their residual arrives with the next exchange)."""
    if not active:
        return 1, np.zeros((1, 1), dtype=np.int64), np.ones((1, 1))
    n, c, v = _restr_tables_local(nc_coarse_global, order, c_elo, c_ehi, felo, fehi, active)
    rows = np.arange(c_elo, c_ehi + 1)
    ghost = (rows < clo) | (rows > chi)
    c = c.copy()
    c[ghost, :] = -1
    return n, c, v


def build_local_hierarchy(cells_global_fine, nlevels, grid, rank, order=1, lengths=None, rep_from=None, depth=None, smoother="jacobi",
                          finest_depth=0, sub_from=None, sub_ranks=None):
    """Local operators of `rank` for every level.

    Levels >= rep_from are REPLICATED (global operators on every rank, no halo); by default only the
    coarsest level is.  `depth`: None / 0 = every partitioned level in the own | ghost layout (one exchange per mat-vec); an int or a
    per-level list = ghost layers of the OVERLAPPING layout on the partitioned levels >= 1 (`LocalLevel.overlap`: one local numbering
    over the extended box, square local matrix, one exchange per `depth` sweeps; any order; `smoother` = "jacobi" or "patch" sets how
    many node layers a sweep consumes, see _OverlapGeom).  finest_depth > 0: the FINEST level in the overlapping layout too -- the
    preconditioner's level 0 then lives in the extended-box numbering while the Krylov solver keeps the caller's own | ghost vectors:
    the dict gains "krylov" = the finest operator in the own | ghost layout (a LocalLevel with A and the exchange plan) whose
    `own_idx` maps every owned entry to its local id on level 0 (gmg_set_krylov_map).
    sub_from / sub_ranks: the partitioned levels sub_from .. rep_from-1 live on the FIRST `sub_ranks` ranks only (their own rank grid
    over the same domain; the other ranks hold nothing of them -- entries None) -- the reference's np_per_level / redistribute!
    (ModelHierarchies.jl:80-148, GridTransferOperators.jl:447-532).  Level sub_from then exists in two partitions: the GLUED one on all
    ranks (coarse nodes with the owners of the fine nodes they coincide with: P / R of level sub_from-1 are built against it, own |
    ghost numbering) and the subset's; the dict gains "sub" = dict(sub_from, members, member, n_glue_own, n_glue_ghost, to_sub,
    from_sub): the two redistribution plans (restricted residual: glued owners -> subset owners; correction: subset owners -> glued
    own AND ghost entries, so no separate consistent! is needed), each dict(nbr_rank, snd_ptr, snd_idx, rcv_ptr, rcv_idx, self_src,
    self_dst) with source / destination LOCAL ids, both sides enumerating by ascending global id.  Returns dict(levels=[LocalLevel...],
    rep_from, rep_gid (global ids, in level rep_from numbering, of the rows this rank's boundary restriction produces), cells, grid)."""
    nc = tuple(int(c) for c in cells_global_fine)
    d = len(nc)
    nc3 = nc + (1,) * (3 - d)
    grid = tuple(grid) + (1,) * (3 - len(grid))
    nranks = int(np.prod(grid))
    if rep_from is None:
        rep_from = nlevels - 1
    if nranks == 1:
        rep_from = nlevels          # nothing to replicate
    if not (1 <= rep_from <= nlevels):
        raise ValueError("rep_from must be in 1..nlevels")
    if depth is None:
        depth = 0
    depths = [int(depth)] * nlevels if np.isscalar(depth) else [int(v) for v in depth] + [0] * (nlevels - len(depth))
    depths[0] = int(finest_depth or 0)   # the finest level's vectors are the caller's (own | ghost) unless a separate Krylov operator is built
    for l in range(nlevels):
        if l >= rep_from or nranks == 1:
            depths[l] = 0
    reach = order if smoother == "jacobi" else 3 * order - 2
    cells = [tuple(c // (2 ** l) for c in nc3[:d]) + (1,) * (3 - d) for l in range(nlevels)]
    if sub_from is not None and nranks > 1:
        if not (1 <= sub_from < rep_from and 1 <= int(sub_ranks) < nranks):
            raise ValueError("sub_from must be in 1..rep_from-1 and sub_ranks in 1..nranks-1")
        subgrid = tuple(rank_grid(int(sub_ranks), d))[:3]
    else:
        sub_from, subgrid = None, None

    def grid_of(l):
        return subgrid if (sub_from is not None and l >= sub_from) else grid
    member = sub_from is None or rank < int(sub_ranks)

    def active(l):
        """does this rank hold a part of level l?"""
        return l >= rep_from or sub_from is None or l < sub_from or member
    for l in range(nlevels):
        for k in range(d):
            if cells[l][k] * 2 ** l != nc3[k] or cells[l][k] < 2:
                raise ValueError("cells must be divisible by 2^(nlevels-1) with >= 2 coarsest cells")
            if l < min(rep_from + 1, nlevels) and (cells[l][k] % grid_of(l)[k] or cells[l][k] // grid_of(l)[k] < 2):
                raise ValueError("partitioned levels need >= 2 cells per rank and direction")
    Ls = po._lengths(lengths, d)
    ngeom = min(rep_from + 1, nlevels)
    geoms = [_LevelGeom(cells[l], order, grid_of(l), rank, d) if (active(l) and (member or l < rep_from)) else None for l in range(ngeom)]
    ogeoms = [_OverlapGeom(cells[l], grid_of(l), rank, d, depths[l], order, reach) if (depths[l] > 0 and active(l)) else None for l in range(nlevels)]
    glue = _LevelGeom(cells[sub_from], order, grid, rank, d) if sub_from is not None else None   # level sub_from as the fine partition sees it

    def fine_cols(l):
        """(elo, ehi) per axis of the column numbering of level l's vectors + the remap to apply afterwards (None: lexicographic as is)"""
        if ogeoms[l] is not None:
            return [(ogeoms[l].rng[k][2], ogeoms[l].rng[k][3]) for k in range(3)], None
        return [(geoms[l].rng[k][2], geoms[l].rng[k][3]) for k in range(3)], geoms[l].remap

    def own_ghost_level(l):
        L = LocalLevel()
        g = geoms[l]
        tabs = [_axis_tables_local(cells[l][k], order, *g.rng[k], k < d, Ls[k]) for k in range(3)]
        ncols = [t[0] for t in tabs]
        cols = [t[1] for t in tabs]
        K = [t[2] for t in tabs]
        M = [t[3] for t in tabs]
        terms = [(K[0], M[1], M[2]), (M[0], K[1], M[2])]
        if d == 3:
            terms.append((M[0], M[1], K[2]))
        L.A = g.remap(po._tensor_csr(cols, terms, ncols))
        L.n_own, L.n_ghost = g.n_own, g.n_ghost
        L.own_gid, L.ghost_gid = g.own_gid, g.ghost_gid
        L.nbr_rank, L.snd_ptr, L.snd_idx, L.rcv_ptr = _exchange_plan(g)
        return L

    levels = []
    krylov = None
    for l in range(nlevels):
        L = LocalLevel()
        if l == 0 and l < rep_from and ogeoms[0] is not None:
            krylov = own_ghost_level(0)
            krylov.own_idx = ogeoms[0].own_idx        # owned entries (ascending global id on both sides) -> local ids on level 0
            assert np.array_equal(krylov.own_gid, ogeoms[0].own_gid)
        if not active(l):
            levels.append(None)
            continue
        if l < rep_from and ogeoms[l] is not None:
            og = ogeoms[l]
            tabs = [_axis_tables_local(cells[l][k], order, og.rng[k][2], og.rng[k][3], og.rng[k][2], og.rng[k][3], k < d, Ls[k]) for k in range(3)]
            ncols = [t[0] for t in tabs]
            cols = [t[1] for t in tabs]
            K = [t[2] for t in tabs]
            M = [t[3] for t in tabs]
            terms = [(K[0], M[1], M[2]), (M[0], K[1], M[2])]
            if d == 3:
                terms.append((M[0], M[1], K[2]))
            L.A = po._tensor_csr(cols, terms, ncols)
            L.overlap, L.depth, L.n_local, L.layers, L.ogeom = True, og.depth, og.n_local, og.layers, og
            L.n_own, L.n_ghost = og.n_own, og.n_local - og.n_own
            L.own_idx, L.local_gid, L.own_gid = og.own_idx, og.gid, og.own_gid
            L.nbr_rank, L.snd_ptr, L.snd_idx, L.rcv_ptr, L.rcv_idx = og.plan()
            L.ghost_gid = og.gid[L.rcv_idx]
        elif l < rep_from:
            L = own_ghost_level(l)
        else:
            L.A = po.poisson_matrix(cells[l][:d], order, lengths)
            L.n_own, L.n_ghost = L.A.shape[0], 0
            L.own_gid, L.ghost_gid = np.arange(L.n_own, dtype=np.int64), np.zeros(0, dtype=np.int64)
        L.replicated = l >= rep_from
        levels.append(L)
    rep_gid = np.zeros(0, dtype=np.int64)
    for l in range(nlevels - 1):
        if not active(l):
            continue
        if l + 1 <= rep_from and l < rep_from:    # fine level partitioned
            gf, gc = geoms[l], geoms[l + 1]
            of, oc = ogeoms[l], ogeoms[l + 1] if l + 1 < rep_from else None
            to_glue = sub_from is not None and l + 1 == sub_from     # the coarse level as the FINE partition sees it (own | ghost)
            if to_glue:
                gc, oc = glue, None
            # rows of P: the fine level's matrix rows (owned rows, or every local entry in the overlapping layout)
            frows = [(of.rng[k][2], of.rng[k][3]) if of is not None else (gf.rng[k][0], gf.rng[k][1]) for k in range(3)]
            fcols, fremap = fine_cols(l)
            if l + 1 < rep_from:                  # coarse level partitioned too
                if to_glue:
                    ccols, cremap = [(glue.rng[k][2], glue.rng[k][3]) for k in range(3)], glue.remap
                else:
                    ccols, cremap = fine_cols(l + 1)
                pt = [_interp_tables_local(cells[l + 1][k], order, frows[k][0], frows[k][1], ccols[k][0], ccols[k][1], k < d) for k in range(3)]
                Pm = po._tensor_csr([t[1] for t in pt], [tuple(t[2] for t in pt)], [t[0] for t in pt])
                levels[l].P = cremap(Pm) if cremap is not None else Pm
                if oc is not None:
                    rt = [_restr_tables_ext(cells[l + 1][k], order, oc.rng[k][2], oc.rng[k][3], oc.rng[k][0], oc.rng[k][1],
                                            fcols[k][0], fcols[k][1], k < d) for k in range(3)]
                else:
                    rt = [_restr_tables_local(cells[l + 1][k], order, gc.rng[k][0], gc.rng[k][1], fcols[k][0], fcols[k][1], k < d) for k in range(3)]
                Rm = po._tensor_csr([t[1] for t in rt], [tuple(t[2] for t in rt)], [t[0] for t in rt])
                levels[l].R = fremap(Rm) if fremap is not None else Rm
            else:                                 # boundary: fine partitioned, coarse replicated
                last = [order * cells[l + 1][k] - 1 for k in range(3)]
                pt = [_interp_tables_local(cells[l + 1][k], order, frows[k][0], frows[k][1], 1, last[k], k < d) for k in range(3)]
                levels[l].P = po._tensor_csr([t[1] for t in pt], [tuple(t[2] for t in pt)], [t[0] for t in pt])   # global coarse columns
                rt = [_restr_tables_local(cells[l + 1][k], order, gc.rng[k][0], gc.rng[k][1], fcols[k][0], fcols[k][1], k < d) for k in range(3)]
                Rm = po._tensor_csr([t[1] for t in rt], [tuple(t[2] for t in rt)], [t[0] for t in rt])
                levels[l].R = fremap(Rm) if fremap is not None else Rm
                rep_gid = gc.own_gid
        else:                                     # both replicated: global transfer operators
            levels[l].P = po.prolongation(cells[l + 1][:d], order)
            levels[l].R = levels[l].P.transpose()
    sub = None
    if sub_from is not None:
        M = int(sub_ranks)
        glues = [glue if q == rank else _LevelGeom(cells[sub_from], order, grid, q, d) for q in range(nranks)]

        def subset_own(m):
            """(global ids, local ids on the subset's level sub_from) of the entries member m owns, ascending global id"""
            if depths[sub_from] > 0:
                og = ogeoms[sub_from] if m == rank else _OverlapGeom(cells[sub_from], subgrid, m, d, depths[sub_from], order, reach)
                return og.own_gid, og.own_idx
            g = geoms[sub_from] if m == rank else _LevelGeom(cells[sub_from], order, subgrid, m, d)
            return g.own_gid, np.arange(g.n_own, dtype=np.int64)
        subs = [subset_own(m) for m in range(M)]

        def need(q):
            """(global ids, local ids) of glued own + ghost entries of rank q, sorted by global id"""
            gid = np.concatenate([glues[q].own_gid, glues[q].ghost_gid])
            o = np.argsort(gid, kind="stable")
            return gid[o], o.astype(np.int64)

        def plan(src_of, dst_of, senders, receivers):
            """rank -> (gids ascending, local ids) of what it holds (src) / wants (dst); entries travel from the sender that holds them
            to every receiver that wants them"""
            nbr, sp, rp, si, ri = [], [0], [0], [], []
            self_src = self_dst = np.zeros(0, np.int64)
            mine_s = src_of(rank) if rank in senders else None
            mine_d = dst_of(rank) if rank in receivers else None
            for q in range(nranks):
                snd = rcv = np.zeros(0, np.int64)
                if mine_s is not None and q in receivers:
                    gq, _ = dst_of(q)
                    common = np.intersect1d(mine_s[0], gq, assume_unique=True)
                    snd = mine_s[1][np.searchsorted(mine_s[0], common)]
                if mine_d is not None and q in senders:
                    gq, _ = src_of(q)
                    common = np.intersect1d(gq, mine_d[0], assume_unique=True)
                    rcv = mine_d[1][np.searchsorted(mine_d[0], common)]
                if q == rank:
                    self_src, self_dst = snd, rcv
                    continue
                if snd.size or rcv.size:
                    nbr.append(q); si.append(snd); ri.append(rcv)
                    sp.append(sp[-1] + snd.size); rp.append(rp[-1] + rcv.size)
            cat = (lambda xs: np.concatenate(xs).astype(np.int64) if xs else np.zeros(0, np.int64))
            return dict(nbr_rank=np.asarray(nbr, dtype=np.int32), snd_ptr=np.asarray(sp, dtype=np.int64), snd_idx=cat(si),
                        rcv_ptr=np.asarray(rp, dtype=np.int64), rcv_idx=cat(ri), self_src=self_src.astype(np.int64), self_dst=self_dst.astype(np.int64))
        allr, mem = set(range(nranks)), set(range(M))
        glue_own = lambda q: (glues[q].own_gid, np.arange(glues[q].n_own, dtype=np.int64))
        sub_own = lambda m: subs[m]
        sub = dict(sub_from=int(sub_from), members=M, member=bool(member), n_glue_own=int(glue.n_own), n_glue_ghost=int(glue.n_ghost),
                   to_sub=plan(glue_own, sub_own, allr, mem), from_sub=plan(sub_own, need, mem, allr),
                   n_sub_local=(int(levels[sub_from].n_own + levels[sub_from].n_ghost) if member else 0))
    return dict(levels=levels, rep_from=rep_from, rep_gid=np.ascontiguousarray(rep_gid, dtype=np.int64),
                cells=[c[:d] for c in cells], grid=grid[:d], order=order, rank=rank, nranks=nranks, depths=depths, krylov=krylov, sub=sub)


def overlap_hints(local, niter, smoother="jacobi"):
    """Per level (skip_r, skip_dx) for gmg_set_partition_overlap_hints: which of the two transfer exchanges of an overlapping level the
    geometry makes unnecessary.

    skip_r : after the LAST block of a smoothing pass (blocks of `depth` sweeps; the last one has niter - depth * ((niter - 1) // depth))
             the residual is still exact on `layers - last * reach` node layers around the owned box; the restriction of an owned coarse
             row reads fine nodes up to 1 (Q1) / 3 (Q2) away -- no consistent!(r) before it when that many layers are left.
    skip_dx: `r_own -= (A dxh)_own` reads dxh = P dxH up to `order` nodes outside the owned box; P's rows are complete there when the
             coarse level is replicated (global columns) or overlapping with >= `order` node layers -- and not the glued side of a
             redistribution (own | ghost numbering, filled by the redistribution itself)."""
    order = int(local["order"])
    reach = order if smoother == "jacobi" else 3 * order - 2
    rnodes = 1 if order == 1 else 3
    levels = local["levels"]
    sub = local.get("sub")
    out = []
    for l, L in enumerate(levels):
        if L is None or not getattr(L, "overlap", False) or l >= len(levels) - 1:
            out.append((False, False)); continue
        k = max(1, min(int(L.depth), int(niter)))
        last = int(niter) - k * ((int(niter) - 1) // k)
        skip_r = int(L.layers) - last * reach >= rnodes
        nxt = levels[l + 1]
        glued = sub is not None and sub["sub_from"] == l + 1
        skip_dx = (not glued) and nxt is not None and (nxt.replicated or (getattr(nxt, "overlap", False) and int(nxt.layers) >= order))
        out.append((bool(skip_r), bool(skip_dx)))
    return out


def overlap_geometry(local, smoother="jacobi"):
    """Per level the geometric facts gmg_set_partition_overlap_hints takes -- (exact node layers, layers one sweep consumes, reach of R,
    P's rows complete near the owned box) -- from which the library derives what overlap_hints() predicts for a given niter."""
    order = int(local["order"])
    reach = order if smoother == "jacobi" else 3 * order - 2
    rnodes = 1 if order == 1 else 3
    levels = local["levels"]
    hints = overlap_hints(local, 1, smoother)
    return [(int(L.layers), reach, rnodes, bool(hints[l][1])) if (L is not None and getattr(L, "overlap", False) and l < len(levels) - 1)
            else (0, 0, 0, False) for l, L in enumerate(levels)]


def fold_ranks(locals_):
    """Fold the local hierarchies of ALL ranks of a partition (build_local_hierarchy for rank 0 .. W-1) onto ONE rank whose every
    neighbour is itself -- the partition gmg_comm_set_loopback is made for.

    Per partitioned level: the ranks' owned rows stacked in rank order, the ranks' ghosts stacked behind them (own | ghost layout) or the
    ranks' extended boxes stacked (overlapping layout: block-diagonal local matrix); the exchange plan has one message per
    (receiving rank a, neighbour q of a) pair, in the order the folded ghosts are stored, its send list = what q sends to a with q's
    offset -- so message k's send segment lands exactly in receive segment k, which is how a self send / receive pairs up.  Every
    neighbour is named rank 1 (any rank other than 0 means "this rank" under loopback).  Replicated levels are taken as they are;
    rep_gid = the ranks' lists stacked (together: every row of level rep_from once).
    Mathematically the folded operators ARE the global ones (a ghost column is a copy of an owned entry the exchange keeps current),
    so the serial oracle on the global hierarchy is the reference; the sums are split own | ghost exactly as on W GPUs.
    No rank-subset levels, no separate Krylov operator (finest_depth = 0)."""
    W = len(locals_)
    assert W >= 2 and all(h["nranks"] == W for h in locals_) and all(h.get("sub") is None and h.get("krylov") is None for h in locals_)
    nlev = len(locals_[0]["levels"])
    rep_from = locals_[0]["rep_from"]
    lv = [[h["levels"][l] for h in locals_] for l in range(nlev)]

    def offsets(sizes):
        o = np.zeros(len(sizes) + 1, dtype=np.int64)
        np.cumsum(sizes, out=o[1:])
        return o

    kind, cmap, ncols = [], [], []        # per level: "rep" / "og" / "ovl"; per rank the local-column -> folded-column map
    for l in range(nlev):
        Ls = lv[l]
        if Ls[0].replicated:
            kind.append("rep"); cmap.append([None] * W); ncols.append(Ls[0].A.shape[0])
        elif Ls[0].overlap:
            off = offsets([L.n_local for L in Ls])
            kind.append("ovl"); cmap.append([off[a] + np.arange(Ls[a].n_local, dtype=np.int64) for a in range(W)]); ncols.append(int(off[-1]))
        else:
            off, goff = offsets([L.n_own for L in Ls]), offsets([L.n_ghost for L in Ls])
            kind.append("og")
            cmap.append([np.concatenate([off[a] + np.arange(Ls[a].n_own, dtype=np.int64),
                                         off[-1] + goff[a] + np.arange(Ls[a].n_ghost, dtype=np.int64)]) for a in range(W)])
            ncols.append(int(off[-1] + goff[-1]))

    def stack(mats, maps, nc):
        """rows of the ranks' matrices stacked, columns through the rank's map (None: unchanged)"""
        ptr = [np.zeros(1, dtype=np.int64)]
        idx, val, base = [], [], 0
        for M, m in zip(mats, maps):
            ptr.append(M.ptr[1:] + base)
            base += M.nnz
            idx.append(M.idx.astype(np.int64) if m is None else m[M.idx])
            val.append(M.val)
        return po.CSR((sum(M.shape[0] for M in mats), nc), np.concatenate(ptr), np.concatenate(idx), np.concatenate(val))

    levels = []
    for l in range(nlev):
        Ls = lv[l]
        F = LocalLevel()
        if kind[l] == "rep":
            F.A, F.n_own, F.n_ghost, F.replicated = Ls[0].A, Ls[0].n_own, 0, True
            F.own_gid, F.ghost_gid = Ls[0].own_gid, Ls[0].ghost_gid
        else:
            F.A = stack([L.A for L in Ls], cmap[l], ncols[l])
            F.n_own, F.n_ghost = sum(L.n_own for L in Ls), sum(L.n_ghost for L in Ls)
            F.own_gid, F.ghost_gid = np.concatenate([L.own_gid for L in Ls]), np.concatenate([L.ghost_gid for L in Ls])
            ovl = kind[l] == "ovl"
            loc_off = offsets([L.n_local for L in Ls]) if ovl else offsets([L.n_own for L in Ls])
            snd, rcv, sp, rp = [], [], [0], [0]
            for a in range(W):
                La = Ls[a]
                for k, q in enumerate(La.nbr_rank):
                    Lq = Ls[int(q)]
                    kq = int(np.nonzero(Lq.nbr_rank == a)[0][0])
                    s = loc_off[int(q)] + Lq.snd_idx[Lq.snd_ptr[kq]:Lq.snd_ptr[kq + 1]]
                    nr = int(La.rcv_ptr[k + 1] - La.rcv_ptr[k])
                    assert s.size == nr, "what q sends to a is what a receives from q"
                    snd.append(s); sp.append(sp[-1] + s.size); rp.append(rp[-1] + nr)
                    if ovl:
                        rcv.append(loc_off[a] + La.rcv_idx[La.rcv_ptr[k]:La.rcv_ptr[k + 1]])
            F.nbr_rank = np.ones(len(snd), dtype=np.int32)
            F.snd_ptr, F.rcv_ptr = np.asarray(sp, dtype=np.int64), np.asarray(rp, dtype=np.int64)
            F.snd_idx = np.concatenate(snd).astype(np.int64) if snd else np.zeros(0, np.int64)
            if ovl:
                F.overlap, F.depth, F.layers = True, Ls[0].depth, Ls[0].layers
                F.n_local = int(loc_off[-1])
                F.rcv_idx = np.concatenate(rcv).astype(np.int64) if rcv else np.zeros(0, np.int64)
                F.own_idx = np.concatenate([loc_off[a] + Ls[a].own_idx for a in range(W)])
                F.local_gid = np.concatenate([L.local_gid for L in Ls])
        if l < nlev - 1:
            if kind[l] == "rep":
                F.P, F.R = Ls[0].P, Ls[0].R
            else:
                F.P = stack([L.P for L in Ls], cmap[l + 1], ncols[l + 1])      # fine rows x coarse columns
                F.R = stack([L.R for L in Ls], cmap[l], ncols[l])              # coarse rows x fine columns
        levels.append(F)
    h0 = locals_[0]
    return dict(levels=levels, rep_from=rep_from, rep_gid=np.ascontiguousarray(np.concatenate([h["rep_gid"] for h in locals_]), dtype=np.int64),
                cells=h0["cells"], grid=h0["grid"], order=h0["order"], rank=0, nranks=2, depths=h0["depths"], krylov=None, sub=None,
                folded_from=W, structured=True, cmap=cmap)


def fold_patch_tables(folded, lengths=None):
    """Vertex-star patch tables of a folded hierarchy (fold_ranks) for DistributedGMG(patch_tables=...): per partitioned own | ghost level the
    patches every rank OWNS (local_vertex_star_patches) stacked in rank order, their dofs taken through the rank's column map into the folded
    [own | ghost] numbering, and their matrices assembled by the driver from the global operator (a rank's local matrix lacks the rows of
    ghost dofs: PatchSolvers.jl:137-150) -- exactly what DistributedGMG builds for one rank; replicated levels get the global table.
    The patch smoother of such a level needs consistent!(r) before the local solves AND assemble!(dx) after them (PatchSolvers.jl:227-258):
    over the loopback communicator both directions of the halo run as self-messages."""
    W, order, grid = folded["folded_from"], folded["order"], folded["grid"]
    levels = folded["levels"]
    out = []
    for l, L in enumerate(levels[:-1]):
        cells_l = folded["cells"][l]
        if L.replicated:
            pp, pd = po.vertex_star_patches(cells_l, order)
            out.append((pp, pd.astype(np.int64), None))
            continue
        assert not getattr(L, "overlap", False), "overlapping levels take their patches from the extended box (DistributedGMG does that itself)"
        Ag = po.poisson_matrix(cells_l, order, lengths).to_scipy().tocsr()
        ptr, dofs, blocks = [np.zeros(1, dtype=np.int64)], [], []
        for a in range(W):
            pp, pl, pg = local_vertex_star_patches(cells_l, order, grid, a)
            ptr.append(pp[1:] + ptr[-1][-1])
            dofs.append(folded["cmap"][l][a][pl.astype(np.int64)])
            for p_ in range(pp.size - 1):
                g_ = pg[pp[p_]:pp[p_ + 1]]
                blocks.append(Ag[g_][:, g_].toarray().reshape(-1, order="F"))
        out.append((np.concatenate(ptr), np.concatenate(dofs).astype(np.int64), np.ascontiguousarray(np.concatenate(blocks)) if blocks else np.zeros(0)))
    return out
