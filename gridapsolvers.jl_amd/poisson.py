"""Structured-grid Poisson hierarchy generator (driver-side input synthesis).

Stands in for the Gridap setup the reference's tests run before they reach the
hot path (test/LinearSolvers/GMGTests.jl:204-224 `gmg_poisson_driver`,
src/MultilevelTools/FESpaceHierarchies.jl:151-174 `compute_hierarchy_matrices`,
src/MultilevelTools/GridTransferOperators.jl:391-401 prolongation semantics):

  * domain (0,1)^d, uniform Cartesian cells, Lagrangian Q1 / Q2,
    Dirichlet on the whole boundary ("boundary" tag, GMGTests.jl:213);
  * bilinear form  a(u,v) = int grad(v).grad(u)   (GMGTests.jl:208);
  * per-level re-discretised matrices (not Galerkin RAP; they coincide here);
  * prolongation P_ij = Phi_j^H(x_i^h) on free dofs (dv_H = 0 in :residual
    mode, GridTransferOperators.jl:226,230), restriction R = P^T
    (GridTransferOperators.jl:202-209,536-547 on nested meshes);
  * vertex-star patches (PatchTopology(ReferenceFE{0},model), assembly=:star,
    GMGTests.jl:18-36).

Everything is tensor-product: A = K(x)M(x)M + M(x)K(x)M + M(x)M(x)K with 1-D
stiffness K and mass M.  Free dofs are numbered lexicographically, x fastest.
The *structural* pattern is kept (3-D Q1: 27 entries/row including the six
exact-zero face couplings), which is what a finite-element assembler stores.

This module is host-side numpy only; nothing in it is on the timed path.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "CSR", "poisson_matrix", "prolongation", "dirichlet_lift_rhs", "nodal_values",
    "l2_error_sq", "vertex_star_patches", "coarse_cell_interior_patches", "build_hierarchy", "random_rhs", "level_sizes",
    "poisson_matrix_varcoef", "smooth_kappa", "StreamedCSR", "poisson_matrix_stream", "prolongation_stream",
    "restriction_stream",
]


class CSR:
    """Plain CSR container: int64 row pointers, int32 columns, fp64 values (0-based)."""

    def __init__(self, shape, ptr, idx, val):
        self.shape = (int(shape[0]), int(shape[1]))
        self.ptr = np.ascontiguousarray(ptr, dtype=np.int64)
        self.idx = np.ascontiguousarray(idx, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float64)

    @property
    def nnz(self):
        return int(self.ptr[-1])

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val, self.idx, self.ptr), shape=self.shape)

    def transpose(self):
        t = self.to_scipy().T.tocsr()
        t.sort_indices()
        return CSR(t.shape, t.indptr, t.indices, t.data)

    def matvec(self, x):
        return self.to_scipy() @ x


class StreamedCSR:
    """An operator delivered as consecutive row blocks instead of one CSR (shape known up front).

    `row_blocks()` yields (row0, CSR block) with global column indices; the solver hands every block to
    gmg_set_operator_rows and drops it, so the host never holds more than one block -- the driver-side counterpart of an
    assembler that emits the matrix slab by slab."""

    def __init__(self, shape, blocks_fn, plan_fn=None):
        self.shape = (int(shape[0]), int(shape[1]))
        self._fn = blocks_fn
        self._plan = plan_fn

    def row_blocks(self):
        return self._fn()

    def row_plan(self):
        """("block", row0, CSR) items, and ("repeat", nrows_block, count, col_shift) where the last nrows_block rows recur `count`
        more times with their columns shifted (gmg_set_operator_rows_repeat): what a structured-mesh assembler knows for free."""
        if self._plan is not None:
            return self._plan()
        return (("block", row0, B) for row0, B in self._fn())

    def materialize(self):
        """the whole operator as one CSR (tests / small sizes)"""
        ptrs, idxs, vals, base = [np.zeros(1, dtype=np.int64)], [], [], 0
        for row0, B in self.row_blocks():
            ptrs.append(B.ptr[1:] + base)
            idxs.append(B.idx); vals.append(B.val)
            base += B.nnz
        return CSR(self.shape, np.concatenate(ptrs), np.concatenate(idxs) if idxs else np.zeros(0, np.int32),
                   np.concatenate(vals) if vals else np.zeros(0))


# --------------------------------------------------------------------------
# 1-D building blocks
# --------------------------------------------------------------------------
def _elem_1d(order, h):
    if order == 1:
        K = np.array([[1.0, -1.0], [-1.0, 1.0]]) / h
        M = np.array([[2.0, 1.0], [1.0, 2.0]]) * (h / 6.0)
    elif order == 2:  # local nodes: left, mid, right
        K = np.array([[7.0, -8.0, 1.0], [-8.0, 16.0, -8.0], [1.0, -8.0, 7.0]]) / (3.0 * h)
        M = np.array([[4.0, 2.0, -1.0], [2.0, 16.0, 2.0], [-1.0, 2.0, 4.0]]) * (h / 30.0)
    else:
        raise ValueError("order must be 1 or 2")
    return K, M


def _assemble_1d(n, order, length=1.0):
    """Dense 1-D stiffness / mass on all order*n+1 nodes of (0,length), and the element pattern."""
    h = float(length) / n
    nn = order * n + 1
    Ke, Me = _elem_1d(order, h)
    K = np.zeros((nn, nn))
    M = np.zeros((nn, nn))
    S = np.zeros((nn, nn), dtype=bool)
    for e in range(n):
        sl = slice(order * e, order * e + order + 1)
        K[sl, sl] += Ke
        M[sl, sl] += Me
        S[sl, sl] = True
    return K, M, S


def _padded_rows(S, mats, rows, col_map):
    """Rows `rows` of the 1-D operators as fixed-width padded tables.

    Returns cols[nr,W] (mapped through col_map, -1 = absent / non-free) and a
    list of vals[nr,W] (one per matrix in mats)."""
    nr = len(rows)
    W = int(S[rows].sum(axis=1).max()) if nr else 1
    cols = -np.ones((nr, W), dtype=np.int64)
    vals = [np.zeros((nr, W)) for _ in mats]
    for a, r in enumerate(rows):
        cc = np.nonzero(S[r])[0]
        k = 0
        for c in cc:
            if col_map[c] < 0:
                continue
            cols[a, k] = col_map[c]
            for t, Mx in enumerate(mats):
                vals[t][a, k] = Mx[r, c]
            k += 1
    return cols, vals


def _lengths(lengths, d):
    """Domain is (0,L1)x..x(0,Ld); default the unit box of the reference tests."""
    if lengths is None:
        return (1.0,) * 3
    L = tuple(float(v) for v in lengths)
    return L + (1.0,) * (3 - len(L))


def _axis_tables(n, order, dim_active, length=1.0):
    """(ncols_free, cols[nr,W], Kvals, Mvals) for one axis; a collapsed axis
    (2-D problems) is the 1x1 identity for M and zero for K."""
    if not dim_active:
        return 1, np.zeros((1, 1), dtype=np.int64), np.zeros((1, 1)), np.ones((1, 1))
    K, M, S = _assemble_1d(n, order, length)
    nn = order * n + 1
    free = np.arange(1, nn - 1)
    col_map = -np.ones(nn, dtype=np.int64)
    col_map[free] = np.arange(len(free))
    cols, (kv, mv) = _padded_rows(S, [K, M], free, col_map)
    return len(free), cols, kv, mv


def _tensor_csr(axes, terms, ncols_axes):
    """CSR of sum_t  Z_t (x) Y_t (x) X_t  (x fastest) from padded 1-D row tables.

    axes  = [(cols_x, ...), (cols_y, ...), (cols_z, ...)] padded col tables
    terms = list of (vx, vy, vz) padded value tables
    Rows are emitted in lexicographic order with sorted columns, so no global
    sort is needed.  Processed in slabs over z rows to bound memory."""
    cx, cy, cz = axes
    nx, ny, nz = cx.shape[0], cy.shape[0], cz.shape[0]
    ncx, ncy, ncz = ncols_axes
    Wx, Wy, Wz = cx.shape[1], cy.shape[1], cz.shape[1]
    mx, my, mz = cx >= 0, cy >= 0, cz >= 0
    cntx, cnty, cntz = mx.sum(1), my.sum(1), mz.sum(1)
    row_nnz = (cntz[:, None, None] * cnty[None, :, None] * cntx[None, None, :]).reshape(-1)
    ptr = np.zeros(nx * ny * nz + 1, dtype=np.int64)
    np.cumsum(row_nnz, out=ptr[1:])
    nnz = int(ptr[-1])
    idx = np.empty(nnz, dtype=np.int32)
    val = np.empty(nnz, dtype=np.float64)
    # one z-plane of rows at a time; planes that agree up to a column shift (all interior planes of a uniform mesh) reuse the
    # first such plane's arrays (`_tensor_csr_blocks`: same products and sums, bit-identical) -- 10x faster than masking the
    # dense [z,y,x,wz,wy,wx] table of the whole slab
    for row0, B in _tensor_csr_blocks(axes, terms, ncols_axes, raw=True):
        lo = ptr[row0]
        idx[lo:lo + B[1].size] = B[1]
        val[lo:lo + B[2].size] = B[2]
    return CSR((nx * ny * nz, ncx * ncy * ncz), ptr, idx, val)


def _plane_maker(axes, terms, ncols_axes):
    """plane(z) -> (ptr, idx, val) of one z-plane of rows of  sum_t Z_t (x) Y_t (x) X_t  (random access).  Values are formed by the
    same products and sums as one big tensor product would.  Planes whose 1-D z-rows agree up to a shift of the column index
    (all interior planes of a uniform mesh, period `order`) reuse the first such plane's arrays: the columns get a constant
    added, the values are the very same array."""
    cx, cy, cz = axes
    nx, ny, nz = cx.shape[0], cy.shape[0], cz.shape[0]
    ncx, ncy, ncz = ncols_axes
    mx, my, mz = cx >= 0, cy >= 0, cz >= 0
    cache = {}

    def plane(z):
        valid = np.nonzero(mz[z])[0]
        first = int(cz[z, valid[0]]) if valid.size else 0
        key = (tuple(np.where(mz[z], cz[z] - first, -1).tolist()),) + tuple(tuple(t[2][z].tolist()) for t in terms)
        hit = cache.get(key)
        if hit is None:
            mask = (mz[z:z + 1, None, None, :, None, None] & my[None, :, None, None, :, None] & mx[None, None, :, None, None, :])
            col = (cz[z:z + 1, None, None, :, None, None] * (ncy * ncx) + cy[None, :, None, None, :, None] * ncx
                   + cx[None, None, :, None, None, :])
            v = None
            for (vx, vy, vz) in terms:
                t = (vz[z:z + 1, None, None, :, None, None] * vy[None, :, None, None, :, None] * vx[None, None, :, None, None, :])
                v = t if v is None else v + t
            ptr = np.zeros(ny * nx + 1, dtype=np.int64)
            np.cumsum(mask.reshape(ny * nx, -1).sum(axis=1), out=ptr[1:])
            hit = (first, ptr, col[mask].astype(np.int64), np.ascontiguousarray(v[mask]))
            del mask, col, v
            if len(cache) < 12:
                cache[key] = hit
        f0, ptr, idx0, val = hit
        idx = idx0 + (first - f0) * (ncy * ncx) if first != f0 else idx0
        return ptr, idx, val
    return plane


def _tensor_csr_blocks(axes, terms, ncols_axes, raw=False):
    """The same operator as `_tensor_csr`, one z-plane of rows at a time: yields (row0, CSR block) (bit-identical values)."""
    cx, cy, cz = axes
    nx, ny, nz = cx.shape[0], cy.shape[0], cz.shape[0]
    ncx, ncy, ncz = ncols_axes
    plane = _plane_maker(axes, terms, ncols_axes)
    for z in range(nz):
        ptr, idx, val = plane(z)
        if raw:
            yield z * ny * nx, (ptr, idx, val)
        else:
            yield z * ny * nx, CSR((ny * nx, ncx * ncy * ncz), ptr, idx, val)


def _axis_keys(c, vals):
    """Per 1-D row: (column pattern relative to the row's first column, value rows) and that first column -- rows with equal keys
    differ by a shift of their columns only."""
    m = c >= 0
    keys, firsts = [], []
    for r in range(c.shape[0]):
        valid = np.nonzero(m[r])[0]
        first = int(c[r, valid[0]]) if valid.size else 0
        keys.append((tuple(np.where(m[r], c[r] - first, -1).tolist()),) + tuple(tuple(v[r].tolist()) for v in vals))
        firsts.append(first)
    return keys, firsts


def _repeat_plan(keys, firsts):
    """[("item", r) | ("repeat", period, count, shift)]: runs of 1-D rows that recur with period 1, 2 or 4 up to a constant
    column shift are declared instead of listed."""
    n, r, out = len(keys), 0, []
    while r < n:
        done = False
        for p in (1, 2, 4):
            if r < p or r + p > n:
                continue
            delta = firsts[r] - firsts[r - p]
            count = 0
            while r + (count + 1) * p <= n and all(keys[r + count * p + t] == keys[r - p + t] and
                                                   firsts[r + count * p + t] - firsts[r - p + t] == delta * (count + 1) for t in range(p)):
                count += 1
            if count >= 1 and delta >= 0:
                out.append(("repeat", p, count, delta))
                r += count * p
                done = True
                break
        if not done:
            out.append(("item", r))
            r += 1
    return out


def _line_maker(axes, terms, ncols_axes):
    """line(z, y) -> (ptr, idx, val) of the nx rows of one grid line: the same products ((vz * vy) * vx), sums over the terms and
    column order as `_plane_maker`, hence the same bits."""
    cx, cy, cz = axes
    nx = cx.shape[0]
    ncx, ncy, ncz = ncols_axes
    mx, my, mz = cx >= 0, cy >= 0, cz >= 0

    def line(z, y):
        mask = mz[z][None, :, None, None] & my[y][None, None, :, None] & mx[:, None, None, :]
        col = cz[z][None, :, None, None] * (ncy * ncx) + cy[y][None, None, :, None] * ncx + cx[:, None, None, :]
        v = None
        for (vx, vy, vz) in terms:
            t = vz[z][None, :, None, None] * vy[y][None, None, :, None] * vx[:, None, None, :]
            v = t if v is None else v + t
        ptr = np.zeros(nx + 1, dtype=np.int64)
        np.cumsum(mask.reshape(nx, -1).sum(axis=1), out=ptr[1:])
        return ptr, col[mask].astype(np.int64), np.ascontiguousarray(v[mask])
    return line


def _tensor_csr_plan(axes, terms, ncols_axes, lines=True):
    """`_tensor_csr_blocks` with the repetition made explicit: yields ("block", row0, CSR) for the rows that have to be handed
    over as arrays and ("repeat", nrows_block, count, col_shift) for runs that recur up to a constant column shift -- whole
    node planes (all interior planes of a uniform mesh), and inside the planes that are sent the grid lines (lines=True): the
    driver generates 7 x 7 lines of 511 rows instead of 7 planes of 2.6 x 10^5 for the finest Q2 operator at 256^3."""
    cx, cy, cz = axes
    nx, ny, nz = cx.shape[0], cy.shape[0], cz.shape[0]
    ncx, ncy, ncz = ncols_axes
    zplan = _repeat_plan(*_axis_keys(cz, [t[2] for t in terms]))
    yplan = _repeat_plan(*_axis_keys(cy, [t[1] for t in terms])) if lines else None
    plane = _plane_maker(axes, terms, ncols_axes) if not lines else None   # random access: repeated planes are never generated
    line = _line_maker(axes, terms, ncols_axes) if lines else None
    z = 0
    for item in zplan:
        if item[0] == "repeat":
            _, p, count, delta = item
            yield ("repeat", p * ny * nx, count, delta * ncy * ncx)
            z += p * count
            continue
        z = item[1]
        if not lines:
            ptr, idx, val = plane(z)
            yield ("block", z * ny * nx, CSR((ny * nx, ncx * ncy * ncz), ptr, idx, val))
        else:
            for it in yplan:
                if it[0] == "repeat":
                    _, p, count, delta = it
                    yield ("repeat", p * nx, count, delta * ncx)
                else:
                    ptr, idx, val = line(z, it[1])
                    yield ("block", (z * ny + it[1]) * nx, CSR((nx, ncx * ncy * ncz), ptr, idx, val))
        z += 1


def _dims(ncells):
    nc = tuple(int(c) for c in ncells)
    if len(nc) not in (2, 3):
        raise ValueError("2-D or 3-D only")
    return nc + (1,) * (3 - len(nc)), len(nc)


def level_sizes(ncells, order):
    """Number of free dofs (order*n-1 per active direction)."""
    nc, d = _dims(ncells)
    return int(np.prod([order * nc[k] - 1 for k in range(d)]))


def poisson_matrix(ncells, order=1, lengths=None) -> CSR:
    """Free-free block of the Q`order` stiffness matrix of -Laplace on (0,1)^d (or (0,L_k) per axis)."""
    nc, d = _dims(ncells)
    Ls = _lengths(lengths, d)
    tabs = [_axis_tables(nc[k], order, k < d, Ls[k]) for k in range(3)]
    ncols = [t[0] for t in tabs]
    cols = [t[1] for t in tabs]
    K = [t[2] for t in tabs]
    M = [t[3] for t in tabs]
    terms = [(K[0], M[1], M[2]), (M[0], K[1], M[2])]
    if d == 3:
        terms.append((M[0], M[1], K[2]))
    return _tensor_csr(cols, terms, ncols)


def smooth_kappa(X, Y, Z):
    """The smooth, strictly positive diffusion coefficient of the variable-coefficient bench / test leg."""
    return 1.0 + 0.5 * np.sin(2.0 * np.pi * X + 0.3) * np.cos(3.0 * Y + 0.1) * np.sin(1.7 * np.pi * Z + 0.7) + 0.25 * X * Y


def poisson_matrix_varcoef(ncells, kappa=None, lengths=None) -> CSR:
    """Free-free block of the Q1 stiffness matrix of  a(u,v) = int kappa(x) grad(v).grad(u)  with kappa evaluated at the
    cell centres (one-point coefficient quadrature), Dirichlet on the whole boundary.  Same structural pattern as
    `poisson_matrix` (3^d entries per interior row, exact zeros included), but -- kappa being smooth and non-constant --
    (almost) every row is distinct, which is what a variable-coefficient or mapped-mesh problem hands to the solver:
    no row-pattern dictionary or value dictionary applies and the operator is streamed as 12 B per stored nonzero."""
    nc, d = _dims(ncells)
    if kappa is None:
        kappa = smooth_kappa
    Ls = _lengths(lengths, d)
    h = [Ls[k] / nc[k] for k in range(3)]
    # element matrices of the d-linear element, kappa = 1: tensor products of the 1-D blocks
    K1 = [_elem_1d(1, h[k])[0] if k < d else np.ones((1, 1)) for k in range(3)]
    M1 = [_elem_1d(1, h[k])[1] if k < d else np.ones((1, 1)) for k in range(3)]
    loc = [2 if k < d else 1 for k in range(3)]

    def ke(ax, bx, ay, by, az, bz):
        v = K1[0][ax, bx] * M1[1][ay, by] * M1[2][az, bz] + M1[0][ax, bx] * K1[1][ay, by] * M1[2][az, bz]
        if d == 3:
            v = v + M1[0][ax, bx] * M1[1][ay, by] * K1[2][az, bz]
        return v

    # cell-centre coefficient, padded by one layer of zeros so that "cells outside the mesh" contribute nothing
    xs = [(np.arange(nc[k]) + 0.5) * h[k] if k < d else np.zeros(1) for k in range(3)]
    Zc, Yc, Xc = np.meshgrid(xs[2], xs[1], xs[0], indexing="ij")
    kap = np.asarray(kappa(Xc, Yc, Zc), dtype=np.float64)
    nf = [nc[k] - 1 if k < d else 1 for k in range(3)]                  # free nodes per axis
    N = nf[0] * nf[1] * nf[2]
    offs = [(-1, 0, 1) if k < d else (0,) for k in range(3)]
    W = len(offs[0]) * len(offs[1]) * len(offs[2])
    vals = np.zeros((nf[2], nf[1], nf[0], W))
    # node (i,j,k) (free numbering, mesh node = i+1) touches cells (i+a, j+b, k+c), a,b,c in {0,1}; inside such a cell it is
    # local node (1-a,1-b,1-c) and the neighbour at node offset (ox,oy,oz) is local node (1-a+ox, ...) when that is in {0,1}
    sl = []
    for a in range(loc[0]):
        for b in range(loc[1]):
            for c in range(loc[2]):
                kc = kap[(slice(c, c + nf[2]) if d == 3 else slice(0, 1)), b:b + nf[1], a:a + nf[0]]
                la, lb, lc = (1 - a if loc[0] == 2 else 0), (1 - b if loc[1] == 2 else 0), (1 - c if loc[2] == 2 else 0)
                w = 0
                for oz in offs[2]:
                    for oy in offs[1]:
                        for ox in offs[0]:
                            ma, mb, mc = la + ox, lb + oy, lc + oz
                            if 0 <= ma < loc[0] and 0 <= mb < loc[1] and 0 <= mc < loc[2]:
                                vals[..., w] += kc * ke(la, ma, lb, mb, lc, mc)
                            w += 1
    # columns + Dirichlet pruning
    I = np.arange(nf[0])[None, None, :]
    J = np.arange(nf[1])[None, :, None]
    Kk = np.arange(nf[2])[:, None, None]
    cols = np.empty((nf[2], nf[1], nf[0], W), dtype=np.int64)
    keep = np.empty((nf[2], nf[1], nf[0], W), dtype=bool)
    w = 0
    for oz in offs[2]:
        for oy in offs[1]:
            for ox in offs[0]:
                ii, jj, kk = I + ox, J + oy, Kk + oz
                ok = (ii >= 0) & (ii < nf[0]) & (jj >= 0) & (jj < nf[1]) & (kk >= 0) & (kk < nf[2])
                keep[..., w] = ok
                cols[..., w] = (kk * nf[1] + jj) * nf[0] + ii
                w += 1
    keep = keep.reshape(N, W)
    ptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(keep.sum(axis=1), out=ptr[1:])
    return CSR((N, N), ptr, cols.reshape(N, W)[keep].astype(np.int32), vals.reshape(N, W)[keep])


def poisson_matrix_stream(ncells, order=1, lengths=None) -> StreamedCSR:
    """`poisson_matrix` as a row-block stream (one z-plane of nodes per block)."""
    nc, d = _dims(ncells)
    Ls = _lengths(lengths, d)

    def blocks():
        tabs = [_axis_tables(nc[k], order, k < d, Ls[k]) for k in range(3)]
        K = [t[2] for t in tabs]
        M = [t[3] for t in tabs]
        terms = [(K[0], M[1], M[2]), (M[0], K[1], M[2])]
        if d == 3:
            terms.append((M[0], M[1], K[2]))
        return [t[1] for t in tabs], terms, [t[0] for t in tabs]
    n = level_sizes(ncells, order)
    return StreamedCSR((n, n), lambda: _tensor_csr_blocks(*blocks()), lambda: _tensor_csr_plan(*blocks()))


def _interp_1d(nc_coarse, order):
    """Dense 1-D interpolation, fine nodes (2*nc cells) x coarse nodes."""
    nH = order * nc_coarse + 1
    nh = order * 2 * nc_coarse + 1
    P = np.zeros((nh, nH))
    S = np.zeros((nh, nH), dtype=bool)
    for E in range(nc_coarse):
        for a in range(2 * order + 1):           # fine nodes inside coarse cell E
            xi = a / (2.0 * order)
            if order == 1:
                phi = [1.0 - xi, xi]
            else:
                phi = [2.0 * (xi - 0.5) * (xi - 1.0), 4.0 * xi * (1.0 - xi), 2.0 * xi * (xi - 0.5)]
            i = 2 * order * E + a
            for b, w in enumerate(phi):
                if w != 0.0:
                    P[i, order * E + b] = w
                    S[i, order * E + b] = True
    return P, S


def prolongation(ncells_coarse, order=1) -> CSR:
    """P : coarse free dofs -> fine free dofs (fine mesh = coarse refined x2).

    y = P x  <=>  interpolate!(uH, fv_h, Uh) with zero Dirichlet values
    (GridTransferOperators.jl:391-401, :226,230)."""
    nc, d = _dims(ncells_coarse)
    cols, vals, ncols = [], [], []
    for k in range(3):
        if k >= d:
            cols.append(np.zeros((1, 1), dtype=np.int64)); vals.append(np.ones((1, 1))); ncols.append(1)
            continue
        P, S = _interp_1d(nc[k], order)
        nh, nH = P.shape
        free_h = np.arange(1, nh - 1)
        col_map = -np.ones(nH, dtype=np.int64)
        col_map[1:nH - 1] = np.arange(nH - 2)
        c, (v,) = _padded_rows(S, [P], free_h, col_map)
        cols.append(c); vals.append(v); ncols.append(nH - 2)
    return _tensor_csr(cols, [tuple(vals)], ncols)


def _transfer_tables(ncells_coarse, order, transpose):
    nc, d = _dims(ncells_coarse)
    cols, vals, ncols = [], [], []
    for k in range(3):
        if k >= d:
            cols.append(np.zeros((1, 1), dtype=np.int64)); vals.append(np.ones((1, 1))); ncols.append(1)
            continue
        P, S = _interp_1d(nc[k], order)
        nh, nH = P.shape
        if transpose:                                       # rows = coarse free nodes, columns = fine free nodes
            col_map = -np.ones(nh, dtype=np.int64)
            col_map[1:nh - 1] = np.arange(nh - 2)
            c, (v,) = _padded_rows(S.T.copy(), [P.T.copy()], np.arange(1, nH - 1), col_map)
            ncols.append(nh - 2)
        else:
            col_map = -np.ones(nH, dtype=np.int64)
            col_map[1:nH - 1] = np.arange(nH - 2)
            c, (v,) = _padded_rows(S, [P], np.arange(1, nh - 1), col_map)
            ncols.append(nH - 2)
        cols.append(c); vals.append(v)
    return cols, vals, ncols


def prolongation_stream(ncells_coarse, order=1) -> StreamedCSR:
    """`prolongation` as a row-block stream."""
    nc = tuple(int(c) for c in ncells_coarse)

    def args():
        cols, vals, ncols = _transfer_tables(nc, order, False)
        return cols, [tuple(vals)], ncols
    return StreamedCSR((level_sizes(tuple(2 * c for c in nc), order), level_sizes(nc, order)), lambda: _tensor_csr_blocks(*args()),
                       lambda: _tensor_csr_plan(*args()))


def restriction_stream(ncells_coarse, order=1) -> StreamedCSR:
    """R = P^T as a row-block stream (rows = coarse dofs): the tensor product of the transposed 1-D interpolations, so the
    entries are the very numbers of `prolongation(...).transpose()`."""
    nc = tuple(int(c) for c in ncells_coarse)

    def args():
        cols, vals, ncols = _transfer_tables(nc, order, True)
        return cols, [tuple(vals)], ncols
    return StreamedCSR((level_sizes(nc, order), level_sizes(tuple(2 * c for c in nc), order)), lambda: _tensor_csr_blocks(*args()),
                       lambda: _tensor_csr_plan(*args()))


def _apply_axes(G, mats):
    """(Mz (x) My (x) Mx) applied to G[z,y,x]."""
    out = G
    out = np.tensordot(out, mats[0].T, axes=([2], [0]))                # x
    out = np.moveaxis(np.tensordot(out, mats[1].T, axes=([1], [0])), 2, 1)  # y
    out = np.moveaxis(np.tensordot(out, mats[2].T, axes=([0], [0])), 2, 0)  # z
    return out


def _full_1d(nc, order, d, lengths=None):
    Ks, Ms, nn = [], [], []
    Ls = _lengths(lengths, d)
    for k in range(3):
        if k < d:
            K, M, _ = _assemble_1d(nc[k], order, Ls[k])
        else:
            K, M = np.zeros((1, 1)), np.ones((1, 1))
        Ks.append(K); Ms.append(M); nn.append(K.shape[0])
    return Ks, Ms, nn


def _node_coords(nc, order, d, lengths=None):
    Ls = _lengths(lengths, d)
    xs = [np.linspace(0.0, Ls[k], order * nc[k] + 1) if k < d else np.zeros(1) for k in range(3)]
    Z, Y, X = np.meshgrid(xs[2], xs[1], xs[0], indexing="ij")
    return X, Y, Z


def _interior(nn, d):
    return tuple(slice(1, nn[k] - 1) if k < d else slice(0, 1) for k in (2, 1, 0))


def dirichlet_lift_rhs(ncells, order=1, u=None, lengths=None):
    """b = -A_fd u_d for f = 0: the rhs of the reference's test problem
    u = x1 + x2 (GMGTests.jl:204-206: f = -Laplace(u) = 0)."""
    nc, d = _dims(ncells)
    if u is None:
        u = lambda X, Y, Z: X + Y
    Ks, Ms, nn = _full_1d(nc, order, d, lengths)
    X, Y, Z = _node_coords(nc, order, d, lengths)
    G = u(X, Y, Z).astype(np.float64)
    G[_interior(nn, d)] = 0.0                                  # keep only Dirichlet values
    AG = _apply_axes(G, [Ks[0], Ms[1], Ms[2]]) + _apply_axes(G, [Ms[0], Ks[1], Ms[2]])
    if d == 3:
        AG = AG + _apply_axes(G, [Ms[0], Ms[1], Ks[2]])
    return np.ascontiguousarray(-AG[_interior(nn, d)].reshape(-1))


def nodal_values(ncells, order=1, u=None, lengths=None):
    """Nodal interpolant of u on the free dofs (exact discrete solution for u in the FE space)."""
    nc, d = _dims(ncells)
    if u is None:
        u = lambda X, Y, Z: X + Y
    nn = [order * nc[k] + 1 if k < d else 1 for k in range(3)]
    X, Y, Z = _node_coords(nc, order, d, lengths)
    return np.ascontiguousarray(u(X, Y, Z)[_interior(nn, d)].reshape(-1).astype(np.float64))


def l2_error_sq(ncells, order, x, u=None, lengths=None):
    """E = int (u_h - u)^2 = e^T M e for u in the FE space -- the quantity the
    reference tests print / assert (SmoothersTests.jl:36-43, GMGTests.jl:134-142)."""
    nc, d = _dims(ncells)
    Ks, Ms, nn = _full_1d(nc, order, d, lengths)
    e = np.zeros((nn[2], nn[1], nn[0]))
    e[_interior(nn, d)] = (np.asarray(x) - nodal_values(ncells, order, u, lengths)).reshape(e[_interior(nn, d)].shape)
    return float(np.sum(e * _apply_axes(e, Ms)))


def random_rhs(n, seed=20240601):
    """Problem P-rand: b ~ U(-1,1) (deterministic)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    return rng.uniform(-1.0, 1.0, size=n)


def vertex_star_patches(ncells, order=1):
    """One patch per mesh vertex: the free dofs strictly inside the union of
    the cells touching that vertex (Q1: the vertex dof; Q2: up to 3^d dofs).
    Returns (patch_ptr int64, patch_dofs int32), patches in lexicographic
    vertex order, dofs sorted ascending inside each patch."""
    nc, d = _dims(ncells)
    nf = [order * nc[k] - 1 if k < d else 1 for k in range(3)]

    def axis_lists(k):
        if k >= d:
            return [np.zeros(1, dtype=np.int64)]
        out = []
        for v in range(nc[k] + 1):
            node = order * v
            lo, hi = node - (order - 1), node + (order - 1)
            nodes = np.arange(max(lo, 1), min(hi, order * nc[k] - 1) + 1)
            out.append(nodes - 1)                        # free numbering
        return out

    ax, ay, az = axis_lists(0), axis_lists(1), axis_lists(2)
    # vectorised over one z-layer of vertices at a time (16.6 M patches at 256^3): per axis the dof lists are runs
    # [start, start+cnt); a patch is their tensor product, dofs ascending = z-major enumeration of the product
    sx = np.array([l[0] if len(l) else 0 for l in ax], dtype=np.int64); cxn = np.array([len(l) for l in ax], dtype=np.int64)
    sy = np.array([l[0] if len(l) else 0 for l in ay], dtype=np.int64); cyn = np.array([len(l) for l in ay], dtype=np.int64)
    sizes_xy = (cyn[:, None] * cxn[None, :]).reshape(-1)
    ptr_parts, dof_parts, base = [np.zeros(1, dtype=np.int64)], [], 0
    for lz in az:
        cz = len(lz)
        sizes = sizes_xy * cz
        ptr_parts.append(base + np.cumsum(sizes))
        tot = int(sizes.sum())
        base += tot
        if tot == 0:
            continue
        pid = np.repeat(np.arange(sizes.size, dtype=np.int64), sizes)
        t = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(sizes) - sizes, sizes)      # index inside the patch
        py, px = pid // cxn.size, pid % cxn.size
        nxp, nyp = cxn[px], cyn[py]
        iz = t // (nyp * nxp)
        iy = (t // nxp) % nyp
        ix = t % nxp
        g = (lz[0] + iz) * (nf[1] * nf[0]) + (sy[py] + iy) * nf[0] + (sx[px] + ix)
        dof_parts.append(g.astype(np.int32))
    dofs = np.concatenate(dof_parts) if dof_parts else np.zeros(0, dtype=np.int32)
    return np.concatenate(ptr_parts), dofs


def coarse_cell_interior_patches(ncells_coarse, order=1):
    """One patch per COARSE cell: the fine free dofs strictly inside it (fine mesh = coarse refined x2) --
    the patches of CoarsePatchTopology / assembly=:interior used by the reference's patch prolongation
    (PatchBasedSmoothers/CoarsePatchTopologies.jl, PatchTransferOperators.jl).  Q1: 1 dof, Q2: 3^d dofs."""
    nc, d = _dims(ncells_coarse)
    nf = [order * 2 * nc[k] - 1 if k < d else 1 for k in range(3)]

    def axis_lists(k):
        if k >= d:
            return [np.zeros(1, dtype=np.int64)]
        return [np.arange(2 * order * I + 1, 2 * order * (I + 1)) - 1 for I in range(nc[k])]

    ax, ay, az = axis_lists(0), axis_lists(1), axis_lists(2)
    ptr, dofs = [0], []
    for lz in az:
        for ly in ay:
            for lx in ax:
                g = (lz[:, None, None] * (nf[1] * nf[0]) + ly[None, :, None] * nf[0] + lx[None, None, :]).reshape(-1)
                dofs.append(g)
                ptr.append(ptr[-1] + g.size)
    return np.asarray(ptr, dtype=np.int64), np.concatenate(dofs).astype(np.int32)


def build_hierarchy(ncells_fine, nlevels, order=1, lengths=None, kappa=None, stream_min_rows=None):
    """Level 1 = finest (reference convention, ModelHierarchies.jl:80-111).
    kappa (Q1 only): callable kappa(X,Y,Z) -> every level is the re-discretised variable-coefficient operator
    (GMGLinearSolvers.jl:342-353 assembles the level matrices from the weak form per level).
    stream_min_rows: operators with at least that many rows are returned as `StreamedCSR` row-block streams (never
    materialised here); restrictions of streamed prolongations are streamed too.

    Returns dict(mats=[A_1..A_L], prolongations=[P_1..P_{L-1}] (P_l: level l+1 -> l),
    restrictions=[R_l = P_l^T], ncells=[...], order=order)."""
    nc = tuple(int(c) for c in ncells_fine)
    cells = [tuple(c // (2 ** l) for c in nc) for l in range(nlevels)]
    for l in range(nlevels):
        if any(cells[l][k] * 2 ** l != nc[k] or cells[l][k] < 2 for k in range(len(nc))):
            raise ValueError("ncells must be divisible by 2^(nlevels-1) with >=2 coarsest cells")
    if kappa is not None:
        if order != 1:
            raise ValueError("variable-coefficient generator is Q1 only")
        mats = [poisson_matrix_varcoef(c, kappa, lengths) for c in cells]
    else:
        big = lambda c: stream_min_rows is not None and level_sizes(c, order) >= stream_min_rows
        mats = [poisson_matrix_stream(c, order, lengths) if big(c) else poisson_matrix(c, order, lengths) for c in cells]
        Ps = [prolongation_stream(cells[l + 1], order) if big(cells[l]) else prolongation(cells[l + 1], order) for l in range(nlevels - 1)]
        Rs = [restriction_stream(cells[l + 1], order) if big(cells[l]) else Ps[l].transpose() for l in range(nlevels - 1)]
        return dict(mats=mats, prolongations=Ps, restrictions=Rs, ncells=cells, order=order)
    Ps = [prolongation(cells[l + 1], order) for l in range(nlevels - 1)]
    Rs = [P.transpose() for P in Ps]
    return dict(mats=mats, prolongations=Ps, restrictions=Rs, ncells=cells, order=order)
