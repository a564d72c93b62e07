"""Lid-driven-cavity Stokes inputs on a structured 2-D mesh: Q2 velocity / discontinuous P1 pressure, grad-div
(augmented Lagrangian) velocity form -- driver-side input synthesis for the block-preconditioner path.

Stands in for the Gridap setup of the reference application test/Applications/StokesGMG.jl:79-166:

  * Omega = (0,1)^2, n x n cells; velocity Q2 (vector valued, Dirichlet on the whole boundary: u = (1,0) on the open top
    edge, 0 on the walls incl. the corners, :85-108); pressure P1 discontinuous (3 dofs per cell), the constant mode removed
    by fixing ONE dof (what Gridap's `constraint=:zeromean` does at the algebra level, :105);
  * forms (:113-118): a_u(u,v) = int grad v : grad u + alpha * int (div v) Pi_Qh(div u),  Pi_Qh = cell-wise L2 projection
    onto P1 -> alpha * B^T M_p^-1 B;   a((u,p),(v,q)) = a_u(u,v) - int (div v) p - int (div u) q;   l(v) = int v . f, f = (1,1);
  * velocity hierarchy re-discretised per level (:130), transfers = Q2 interpolation per component, vertex-star patches of the
    vector-valued space (:40-58), coarse-cell-interior patches + grad-div rhs form for the patch prolongation (:133-135);
  * preconditioner blocks (:151-157): [ A_uu (GMG)  -B^T ; 0  -1/alpha M_p (CG-Jacobi) ], upper triangular.

Velocity dofs are numbered node-major with the two components interleaved (dof = 2*node + component), nodes
lexicographic (x fastest) over the free (interior) nodes.  numpy/scipy only; nothing here is on the timed path."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from . import poisson as po

__all__ = ["stokes_system", "velocity_hierarchy"]


def _gauss(npts=4):
    x, w = np.polynomial.legendre.leggauss(npts)
    return 0.5 * (x + 1.0), 0.5 * w                      # on (0,1)


def _q2_1d(t):
    """values and derivatives (w.r.t. t in (0,1)) of the three 1-D quadratic Lagrange functions at nodes 0, 1/2, 1"""
    v = np.stack([2.0 * (t - 0.5) * (t - 1.0), 4.0 * t * (1.0 - t), 2.0 * t * (t - 0.5)])
    d = np.stack([4.0 * t - 3.0, 4.0 - 8.0 * t, 4.0 * t - 1.0])
    return v, d


def _element_matrices(h):
    """Q2 x P1disc element integrals on a square cell of side h (tensor Gauss rule, exact for these polynomials)."""
    t, w = _gauss(4)
    v, d = _q2_1d(t)
    M1 = h * np.einsum("aq,bq,q->ab", v, v, w)
    K1 = (1.0 / h) * np.einsum("aq,bq,q->ab", d, d, w)
    xi = 2.0 * t - 1.0
    C0 = h * (v @ w)                  # int phi_a
    C1 = h * (v @ (w * xi))           # int phi_a * xi
    E0 = d @ w                        # int phi_a'          (dx = h dt cancels the 1/h of the derivative)
    E1 = d @ (w * xi)
    # scalar Q2 stiffness on the cell, local node index = 3*b + a (x fastest)
    Ks = np.kron(M1, K1) + np.kron(K1, M1)
    # divergence form: Bx[k, 3*b+a] = int p_k d/dx(phi_a(x) phi_b(y)),  By likewise; p = (1, xi, eta)
    Bx = np.stack([np.kron(C0, E0), np.kron(C0, E1), np.kron(C1, E0)])
    By = np.stack([np.kron(E0, C0), np.kron(E0, C1), np.kron(E1, C0)])
    Mp = (h * h) * np.diag([1.0, 1.0 / 3.0, 1.0 / 3.0])
    F = np.kron(C0, C0)               # int phi_i
    return Ks, Bx, By, Mp, F


def _assemble(n, alpha):
    """Full (all nodes) operators: A_uu (vector Laplacian + grad-div), B (div), M_p, load; velocity dof = 2*node + comp."""
    h = 1.0 / n
    nn = 2 * n + 1
    Ks, Bx, By, Mp, F = _element_matrices(h)
    cy, cx = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    cells = (cy * n + cx).reshape(-1)
    # node ids of the 9 local nodes of every cell, local index 3*b + a
    a = np.arange(3)
    loc = ((2 * cy.reshape(-1, 1, 1) + a[None, :, None]) * nn + (2 * cx.reshape(-1, 1, 1) + a[None, None, :])).reshape(-1, 9)
    nu = 2 * nn * nn
    rows = np.repeat(loc, 9, axis=1).reshape(-1)
    cols = np.tile(loc, (1, 9)).reshape(-1)
    Kvals = np.tile(Ks.reshape(-1), cells.size)
    Kscal = sp.coo_matrix((Kvals, (rows, cols)), shape=(nn * nn, nn * nn)).tocsr()
    Auu = sp.kron(Kscal, sp.identity(2, format="csr"), format="csr")
    # B: 3 pressure dofs per cell x velocity dofs
    prow = np.repeat(3 * cells[:, None] + np.arange(3)[None, :], 9, axis=1).reshape(-1)          # [cell, k, i]
    pcol = np.tile(loc[:, None, :], (1, 3, 1)).reshape(-1)
    B = (sp.coo_matrix((np.tile(Bx.reshape(-1), cells.size), (prow, 2 * pcol)), shape=(3 * n * n, nu))
         + sp.coo_matrix((np.tile(By.reshape(-1), cells.size), (prow, 2 * pcol + 1)), shape=(3 * n * n, nu))).tocsr()
    Mpg = sp.kron(sp.identity(n * n, format="csr"), sp.csr_matrix(Mp), format="csr")
    Mpinv = sp.kron(sp.identity(n * n, format="csr"), sp.csr_matrix(np.linalg.inv(Mp)), format="csr")
    G = (alpha * (B.T @ Mpinv @ B)).tocsr()
    load = np.zeros(nu)
    np.add.at(load, 2 * loc.reshape(-1), np.tile(F, cells.size))
    np.add.at(load, 2 * loc.reshape(-1) + 1, np.tile(F, cells.size))
    return Auu, G, B, Mpg, load, nn


def _free_dirichlet(n):
    nn = 2 * n + 1
    j, i = np.meshgrid(np.arange(nn), np.arange(nn), indexing="ij")
    interior = ((i > 0) & (i < nn - 1) & (j > 0) & (j < nn - 1)).reshape(-1)
    free_nodes = np.nonzero(interior)[0]
    free = np.stack([2 * free_nodes, 2 * free_nodes + 1], axis=1).reshape(-1)
    ud = np.zeros(2 * nn * nn)
    top = ((j == nn - 1) & (i > 0) & (i < nn - 1)).reshape(-1)               # open top edge: u = (1,0); corners belong to the walls
    ud[2 * np.nonzero(top)[0]] = 1.0
    return free, ud


def _csr(M):
    M = M.tocsr(); M.sort_indices()
    return po.CSR(M.shape, M.indptr, M.indices, M.data)


def stokes_system(n, alpha=1.0e3):
    """Block system of the application on the finest level.

    Returns dict(A=[[Auu, Aup],[Apu, None]] (poisson.CSR blocks), Mp_scaled = -1/alpha M_p (the pressure block of the
    preconditioner), b (block vector, velocity block first), sizes, K (scipy, whole system for checks))."""
    Auu, G, B, Mp, load, nn = _assemble(n, alpha)
    free, ud = _free_dirichlet(n)
    A_full = (Auu + G).tocsr()
    np_all = 3 * n * n
    pkeep = np.arange(np_all - 1)                                              # fix the last pressure dof (zero-mean constraint)
    Auu_f = A_full[free][:, free]
    Bf = B[pkeep][:, free]
    bu = load[free] - (A_full[free] @ ud)
    bp = (B[pkeep] @ ud)                                                       # -(-B u_d)
    Aup, Apu = (-Bf.T).tocsr(), (-Bf).tocsr()
    Mps = ((-1.0 / alpha) * Mp[pkeep][:, pkeep]).tocsr()
    K = sp.bmat([[Auu_f, Aup], [Apu, None]]).tocsr()
    return dict(A=[[_csr(Auu_f), _csr(Aup)], [_csr(Apu), None]], Mp_scaled=_csr(Mps), b=np.concatenate([bu, bp]),
                sizes=(free.size, pkeep.size), K=K, n=n, alpha=alpha)


def _vector_table(pp, pd):
    """scalar-node patch table -> vector-dof table (both components of every node, ascending)"""
    dofs = np.stack([2 * pd.astype(np.int64), 2 * pd.astype(np.int64) + 1], axis=1).reshape(-1)
    return 2 * pp, dofs.astype(np.int32)


def velocity_hierarchy(n, nlevels, alpha=1.0e3):
    """GMG inputs for the velocity block: level matrices (re-discretised), P = P_Q2 (x) I_2, R = P^T, vertex-star patch tables
    (smoother), coarse-cell-interior patch tables + the grad-div rhs form (patch prolongation) -- all on free velocity dofs."""
    cells = [n // 2 ** l for l in range(nlevels)]
    if any(c * 2 ** l != n or c < 2 for l, c in enumerate(cells)):
        raise ValueError("n must be divisible by 2^(nlevels-1) with >= 2 coarsest cells")
    mats, Gs = [], []
    for c in cells:
        Auu, G, _B, _Mp, _load, _nn = _assemble(c, alpha)
        free, _ = _free_dirichlet(c)
        mats.append(_csr((Auu + G)[free][:, free]))
        Gs.append(_csr(G[free][:, free]))
    I2 = sp.identity(2, format="csr")
    Ps = [_csr(sp.kron(po.prolongation((cells[l + 1],) * 2, 2).to_scipy(), I2, format="csr")) for l in range(nlevels - 1)]
    Rs = [P.transpose() for P in Ps]
    star = [_vector_table(*po.vertex_star_patches((cells[l],) * 2, 2)) for l in range(nlevels - 1)]
    interior = [_vector_table(*po.coarse_cell_interior_patches((cells[l + 1],) * 2, 2)) for l in range(nlevels - 1)]
    return dict(mats=mats, prolongations=Ps, restrictions=Rs, star_patches=star, interior_patches=interior, graddiv=Gs[:-1],
                ncells=cells)


# ----------------------------------------------------------------------------------------------------------------------
# distributed runs (test/Applications/mpi/StokesGMG.jl:5-12): who owns which dof
# ----------------------------------------------------------------------------------------------------------------------
def velocity_owner(ncells, grid):
    """Owner rank of every FREE velocity dof of the n x n-cell Q2 space (dof = 2 * free node + component, free nodes
    lexicographic): a node goes with the cell box of the px x py rank grid it lies in, interface nodes with the upper box's lower
    neighbour -- i.e. node i in 1..2n-1 belongs to box min((i-1) // (2n/px), px-1), the rule of partition._LevelGeom."""
    n, (px, py) = int(ncells), tuple(grid)[:2]
    nn = 2 * n + 1
    i = np.arange(1, nn - 1)
    ox = np.minimum((i - 1) // (2 * n // px), px - 1)
    oy = np.minimum((i - 1) // (2 * n // py), py - 1)
    node_owner = (oy[:, None] * px + ox[None, :]).reshape(-1)
    return np.repeat(node_owner, 2)


def pressure_owner(ncells, grid):
    """Owner of every kept pressure dof (3 per cell, the last dof of the last cell removed by the zero-mean constraint): its cell's box"""
    n, (px, py) = int(ncells), tuple(grid)[:2]
    c = np.arange(n)
    cell_owner = (np.minimum(c // (n // py), py - 1)[:, None] * px + np.minimum(c // (n // px), px - 1)[None, :]).reshape(-1)
    return np.repeat(cell_owner, 3)[: 3 * n * n - 1]


def patch_owner(pp, pd, dof_owner):
    """Every patch belongs to exactly one rank: the owner of its first (smallest) dof; empty patches to rank 0"""
    first = np.where(pp[1:] > pp[:-1], pd[np.minimum(pp[:-1], max(pd.size - 1, 0))], 0) if pd.size else np.zeros(pp.size - 1, dtype=np.int64)
    own = dof_owner[first.astype(np.int64)] if pd.size else np.zeros(pp.size - 1, dtype=np.int64)
    return np.where(pp[1:] > pp[:-1], own, 0).astype(np.int64)


# ----------------------------------------------------------------------------------------------------------------------
# the same operators by STENCIL REPLICATION (round 6): on the uniform mesh every free velocity dof of one node class
# (vertex / edge midpoint in x / in y / cell centre) x component has the same row, up to the columns that fall on the Dirichlet
# boundary -- all cells around a free node exist.  The rows are read off a 4 x 4-cell reference assembly (the forms are scale
# invariant in 2-D: stiffness and grad-div ~ h^0) and written straight into CSR: seconds where the scipy assembly + triple
# product + fancy indexing above needs minutes (1024^2 cells: 8.4e6 velocity dofs, 2.7e8 nonzeros).  n a power of two keeps the
# scaling exact; the result equals the assembly above to rounding (tests/test_host_logic.py).
# ----------------------------------------------------------------------------------------------------------------------
_REF_N = 4


def _ref_stencils(alpha):
    Auu, G, _B, _Mp, load, nn = _assemble(_REF_N, alpha)
    A = (Auu + G).tocsr(); G = G.tocsr()
    A.sort_indices(); G.sort_indices()
    out = {}
    for name, M in (("A", A), ("G", G)):
        st = {}
        for pj in (0, 1):
            for pi in (0, 1):
                j0, i0 = 4 + pj, 4 + pi
                node = j0 * nn + i0
                for c in (0, 1):
                    r = 2 * node + c
                    cols, vals = M.indices[M.indptr[r]:M.indptr[r + 1]], M.data[M.indptr[r]:M.indptr[r + 1]]
                    nd, cc = cols // 2, cols % 2
                    dj, di = nd // nn - j0, nd % nn - i0
                    o = np.lexsort((cc, di, dj))                                  # ascending column in any mesh
                    keep = vals[o] != 0.0
                    st[(pj, pi, c)] = (dj[o][keep].astype(np.int64), di[o][keep].astype(np.int64), cc[o][keep].astype(np.int64), vals[o][keep])
        out[name] = st
    h2 = (1.0 / _REF_N) ** 2
    out["load"] = {(pj, pi): load[2 * ((4 + pj) * nn + 4 + pi)] / h2 for pj in (0, 1) for pi in (0, 1)}
    return out


def _gen_velocity(n, st, ud_top=None):
    """CSR (po.CSR) of the free-free block of the operator with row stencils `st` on the n x n-cell mesh; ud_top = 1.0: also the
    vector  -A[free, boundary] u_d  for u_d = (1, 0) on the open top edge."""
    nn = 2 * n + 1
    m = nn - 2
    nnz = np.zeros((m, m, 2), dtype=np.int64)
    cls = {}
    for (pj, pi, c), (dj, di, cc, v) in st.items():
        J = np.arange(1, nn - 1)[(np.arange(1, nn - 1) % 2) == pj]
        I = np.arange(1, nn - 1)[(np.arange(1, nn - 1) % 2) == pi]
        kj = ((J[:, None] + dj[None, :]) >= 1) & ((J[:, None] + dj[None, :]) <= nn - 2)      # [nJ, ne]
        ki = ((I[:, None] + di[None, :]) >= 1) & ((I[:, None] + di[None, :]) <= nn - 2)
        nnz[J[0] - 1::2, I[0] - 1::2, c] = kj.astype(np.int64) @ ki.astype(np.int64).T
        cls[(pj, pi, c)] = (J, I, kj, ki)
    ptr = np.zeros(m * m * 2 + 1, dtype=np.int64)
    np.cumsum(nnz.reshape(-1), out=ptr[1:])
    idx = np.empty(int(ptr[-1]), dtype=np.int32)
    val = np.empty(int(ptr[-1]))
    rhs = np.zeros(m * m * 2) if ud_top is not None else None
    for (pj, pi, c), (dj, di, cc, v) in st.items():
        J, I, kj, ki = cls[(pj, pi, c)]
        for j0 in range(0, J.size, 96):                                                       # (blocks of lines: bounds the temporaries)
            Jb, kjb = J[j0:j0 + 96], kj[j0:j0 + 96]
            row = (((Jb[:, None] - 1) * m + (I[None, :] - 1)) * 2 + c)                        # [nJ, nI]
            keep = kjb[:, None, :] & ki[None, :, :]                                          # [nJ, nI, ne]: row-major = CSR order inside the rows
            pos = (ptr[row][:, :, None] + (np.cumsum(keep, axis=2) - keep))[keep]
            col = (((Jb[:, None, None] + dj[None, None, :] - 1) * m + (I[None, :, None] + di[None, None, :] - 1)) * 2 + cc[None, None, :])
            idx[pos] = col[keep]
            val[pos] = np.broadcast_to(v[None, None, :], keep.shape)[keep]
            if rhs is not None:
                top = (~keep) & (cc[None, None, :] == 0) & ((Jb[:, None, None] + dj[None, None, :]) == nn - 1) \
                    & ((I[None, :, None] + di[None, None, :]) >= 1) & ((I[None, :, None] + di[None, None, :]) <= nn - 2)
                if top.any():
                    np.subtract.at(rhs, np.broadcast_to(row[:, :, None], keep.shape)[top], np.broadcast_to(v[None, None, :], keep.shape)[top] * ud_top)
    return po.CSR((m * m * 2, m * m * 2), ptr, idx, val), rhs


def _gen_pressure(n):
    """(B restricted to kept pressure rows x free velocity dofs as po.CSR, B[kept, boundary] u_d, diagonal of M_p on the kept dofs)."""
    h = 1.0 / n
    nn = 2 * n + 1
    m = nn - 2
    _Ks, Bx, By, Mp, _F = _element_matrices(h)
    # entries of a cell's three rows in ascending column order: local node (b, a), component
    ent = [(b, a, c) for b in range(3) for a in range(3) for c in range(2)]
    cy, cx = np.arange(n), np.arange(n)
    kb = np.stack([((2 * cy + b) >= 1) & ((2 * cy + b) <= nn - 2) for b in range(3)], axis=1)      # [n, 3]
    ka = np.stack([((2 * cx + a) >= 1) & ((2 * cx + a) <= nn - 2) for a in range(3)], axis=1)
    cnt = 2 * (kb.sum(1)[:, None] * ka.sum(1)[None, :])                                          # kept entries per row of the cell
    nrow = 3 * n * n
    nnz = np.repeat(cnt.reshape(-1), 3)
    ptr = np.zeros(nrow + 1, dtype=np.int64)
    np.cumsum(nnz, out=ptr[1:])
    idx = np.empty(int(ptr[-1]), dtype=np.int32)
    val = np.empty(int(ptr[-1]))
    bp = np.zeros(nrow)
    cell = cy[:, None] * n + cx[None, :]
    for k in range(3):
        row = 3 * cell + k
        base = ptr[row]
        rank = np.zeros(row.shape, dtype=np.int64)
        for (b, a, c) in ent:
            v = (Bx if c == 0 else By)[k, 3 * b + a]
            keep = kb[:, b][:, None] & ka[:, a][None, :]
            pos = (base + rank)[keep]
            col = (((2 * cy[:, None] + b - 1) * m + (2 * cx[None, :] + a - 1)) * 2 + c)
            idx[pos] = col[keep]
            val[pos] = v
            rank += keep
            if c == 0:
                top = ((2 * cy[:, None] + b) == nn - 1) & ((2 * cx[None, :] + a) >= 1) & ((2 * cx[None, :] + a) <= nn - 2)
                if top.any():
                    bp[row[top]] += v
    Bf = sp.csr_matrix((val, idx, ptr), shape=(nrow, 2 * m * m))[: nrow - 1]
    return Bf, bp[: nrow - 1], np.tile(np.diag(Mp), n * n)[: nrow - 1]


_GEN_CACHE = {}


def _gen_cached(n, alpha, name, with_rhs=False):
    key = (int(n), float(alpha), name, bool(with_rhs))
    if key not in _GEN_CACHE:
        if len(_GEN_CACHE) >= 4:
            _GEN_CACHE.clear()
        _GEN_CACHE[key] = _gen_velocity(n, _ref_stencils(alpha)[name], 1.0 if with_rhs else None)
    return _GEN_CACHE[key]


def stokes_system_fast(n, alpha=1.0e3, with_K=False):
    """stokes_system by stencil replication (n a power of two >= 8).  K (the whole system as one scipy matrix) only on request."""
    if n < 8 or n & (n - 1):
        raise ValueError("n must be a power of two >= 8")
    ref = _ref_stencils(alpha)
    Auu, ru = _gen_cached(n, alpha, "A", True)
    nn = 2 * n + 1
    m = nn - 2
    h2 = (1.0 / n) ** 2
    ld = np.zeros((m, m, 2))
    for (pj, pi), v in ref["load"].items():
        ld[(1 - pj) % 2::2, (1 - pi) % 2::2, :] = v * h2            # free node j = 1 + row index: parity of j = 1 - parity of the index
    bu = ld.reshape(-1) + ru
    Bf, bp, mpd = _gen_pressure(n)
    Aup, Apu = _csr(-Bf.T), _csr(-Bf)
    Mps = sp.diags((-1.0 / alpha) * mpd, format="csr")
    out = dict(A=[[Auu, Aup], [Apu, None]], Mp_scaled=_csr(Mps), b=np.concatenate([bu, bp]), sizes=(Auu.shape[0], Bf.shape[0]), K=None, n=n, alpha=alpha)
    if with_K:
        out["K"] = sp.bmat([[Auu.to_scipy(), Aup.to_scipy()], [Apu.to_scipy(), None]]).tocsr()
    return out


def velocity_hierarchy_fast(n, nlevels, alpha=1.0e3):
    """velocity_hierarchy by stencil replication; levels of fewer than 8 cells per direction are assembled the slow way."""
    cells = [n // 2 ** l for l in range(nlevels)]
    if any(c * 2 ** l != n or c < 2 for l, c in enumerate(cells)):
        raise ValueError("n must be divisible by 2^(nlevels-1) with >= 2 coarsest cells")
    ref = _ref_stencils(alpha)
    mats, Gs = [], []
    for c in cells:
        if c >= 8 and not (c & (c - 1)):
            mats.append(_gen_cached(c, alpha, "A", c == n)[0]); Gs.append(_gen_velocity(c, ref["G"])[0])
        else:
            Auu, G, _B, _Mp, _load, _nn = _assemble(c, alpha)
            free, _ = _free_dirichlet(c)
            mats.append(_csr((Auu + G)[free][:, free])); Gs.append(_csr(G[free][:, free]))
    I2 = sp.identity(2, format="csr")
    Ps = [_csr(sp.kron(po.prolongation((cells[l + 1],) * 2, 2).to_scipy(), I2, format="csr")) for l in range(nlevels - 1)]
    Rs = [P.transpose() for P in Ps]
    star = [_vector_table(*po.vertex_star_patches((cells[l],) * 2, 2)) for l in range(nlevels - 1)]
    interior = [_vector_table(*po.coarse_cell_interior_patches((cells[l + 1],) * 2, 2)) for l in range(nlevels - 1)]
    return dict(mats=mats, prolongations=Ps, restrictions=Rs, star_patches=star, interior_patches=interior, graddiv=Gs[:-1], ncells=cells)
