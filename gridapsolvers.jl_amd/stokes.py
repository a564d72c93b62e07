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
