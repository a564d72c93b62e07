"""Independent numpy/scipy restatement of the reference algorithms -- a SECOND implementation the C oracle is checked
against (tests only).  Written from the reference sources, not from oracle/gmg_oracle.c; uses scipy.sparse for the
operators, numpy.linalg / scipy.linalg (LAPACK getrf/getrs, dlartg) for the dense pieces.

  cycle()         gmg_v_cycle! / gmg_w_cycle! / gmg_f_cycle!     GMGLinearSolvers.jl:468-610
  richardson()    solve!(x,::RichardsonSmootherNS,r)             RichardsonSmoothers.jl:84-98
  patch_solve()   solve_patch_overlapping! / solve_block_jacobi! PatchSolvers.jl:279-300, BlockJacobiSolvers.jl:141-170
  cg(), fgmres()  Krylov/CGSolvers.jl:73-120, FGMRESSolvers.jl:130-199 (restart, basis growth)
  block_apply()   BlockTriangularSolvers.jl:186-242 / BlockDiagonalSolvers.jl:165-177
"""
import numpy as np
import scipy.linalg as sla
import scipy.sparse.linalg as spla


class Jacobi:
    def __init__(self, A):
        self.dinv = 1.0 / A.diagonal()                        # JacobiLinearSolvers.jl:20-23

    def solve(self, r):
        return self.dinv * r                                  # :43-47


class Patch:
    """PatchSolver (pivoting LU, factorised once) / BlockJacobiSolver (NoPivot, re-extracted every apply)."""

    def __init__(self, A, pp, rows, cols=None, mats=None, pivot=True):
        self.A, self.pp, self.rows, self.cols, self.pivot = A.tocsr(), pp, rows, rows if cols is None else cols, pivot
        self.lu = []
        off = 0
        for p in range(pp.size - 1):
            r, c = self.rows[pp[p]:pp[p + 1]], self.cols[pp[p]:pp[p + 1]]
            if r.size == 0:
                self.lu.append(None); continue
            if mats is not None:
                B = mats[off:off + r.size * r.size].reshape(r.size, r.size, order="F"); off += r.size * r.size
            else:
                B = self.A[r][:, c].toarray()
            self.lu.append(sla.lu_factor(B) if pivot else B)

    def solve(self, b):
        x = np.zeros_like(b)                                  # PatchSolvers.jl:287
        for p in range(self.pp.size - 1):                     # :288
            r, c = self.rows[self.pp[p]:self.pp[p + 1]], self.cols[self.pp[p]:self.pp[p + 1]]
            if r.size == 0:
                continue
            if self.pivot:
                xp = sla.lu_solve(self.lu[p], b[r])           # :295 ldiv!
            else:                                             # lu!(A_k, NoPivot()) BlockJacobiSolvers.jl:162
                xp = np.linalg.solve(self.lu[p], b[r])
            x[c] += xp                                        # :296
        return x


def richardson(A, M, niter, omega, x, r):
    for _ in range(niter):                                    # RichardsonSmoothers.jl:90
        dx = omega * M.solve(r)                               # :91-92
        x += dx                                               # :93
        r -= A @ dx                                           # :94-95


class GMG:
    def __init__(self, mats, Ps, Rs, pre, post=None, cycle="v"):
        self.A = [m.to_scipy().tocsr() for m in mats]
        self.P = [p.to_scipy().tocsr() for p in Ps]
        self.R = [r.to_scipy().tocsr() for r in Rs]
        self.pre, self.post = pre, (pre if post is None else post)
        self.cyc = cycle
        self.coarse = spla.splu(self.A[-1].tocsc())           # LUSolver(), GMGLinearSolvers.jl:54

    def cycle(self, l, x, r, ctype=None):
        ctype = self.cyc if ctype is None else ctype
        if l == len(self.A) - 1:
            x[:] = self.coarse.solve(r)                       # :474
            return
        A = self.A[l]
        M, nit, om = self.pre[l]
        richardson(A, M, nit, om, x, r)                       # :481
        passes = 1 if ctype == "v" else 2
        for k in range(passes):
            if k == 1:
                Mp, np_, op = self.post[l]
                richardson(A, Mp, np_, op, x, r)              # W :531 / F :584
            rH = self.R[l] @ r                                # :484
            dxH = np.zeros(self.A[l + 1].shape[0])            # :487
            child = ctype if k == 0 else ("w" if ctype == "w" else "v")
            self.cycle(l + 1, dxH, rH, child)                 # :488
            dx = self.P[l] @ dxH                              # :491
            x += dx                                           # :494
            r -= A @ dx                                       # :495-496
        Mp, np_, op = self.post[l]
        richardson(A, Mp, np_, op, x, r)                      # :499

    def solve(self, b):
        """mode = :preconditioner, maxiter = 1"""
        x = np.zeros_like(b)
        r = b.copy()
        self.cycle(0, x, r)
        return x


def cg(A, b, Pl=None, maxiter=1000, atol=1e-12, rtol=1e-6, flexible=False):
    x = np.zeros_like(b)
    r = b - A @ x
    p = np.zeros_like(b); z = np.zeros_like(b)
    gamma = 1.0
    res = np.linalg.norm(r); hist = [res]
    it = 0
    done = it >= maxiter or 1.0 < rtol or res < atol
    while not done:
        if Pl is None:
            z = r.copy(); beta = gamma; gamma = r @ r; beta = gamma / beta
        elif not flexible:
            z = Pl(r); beta = gamma; gamma = z @ r; beta = gamma / beta
        else:
            delta = z @ r; z = Pl(r); beta = gamma; gamma = z @ r; beta = (gamma - delta) / beta
        p = z + beta * p
        w = A @ p
        alpha = gamma / (p @ w)
        x += alpha * p
        r -= alpha * w
        res = np.linalg.norm(r); hist.append(res); it += 1
        done = it >= maxiter or res / hist[0] < rtol or res < atol
    return x, it, np.array(hist)


def fgmres(A, b, Pr=None, m=5, restart=False, m_add=1, maxiter=100, atol=1e-12, rtol=1e-6):
    from scipy.linalg.lapack import dlartg
    n = b.size
    x = np.zeros(n)
    V = [np.zeros(n) for _ in range(m + 1)]; Z = [np.zeros(n) for _ in range(m)]
    r = b - A @ x
    beta = np.linalg.norm(r); hist = [beta]
    it = 0
    done = it >= maxiter or 1.0 < rtol or beta < atol
    mcur = m
    while not done:
        V[0] = r / beta
        H = np.zeros((mcur + 2, mcur + 1)); g = np.zeros(mcur + 2); c = np.zeros(mcur + 1); s = np.zeros(mcur + 1)
        g[0] = beta
        j = 0
        while not done and not (restart and j >= m):
            if j >= mcur:                                      # expand_krylov_caches!, FGMRESSolvers.jl:77-94,151-154
                for _ in range(m_add):
                    V.append(np.zeros(n)); Z.append(np.zeros(n))
                mcur += m_add
                H = np.pad(H, ((0, m_add), (0, m_add))); g = np.pad(g, (0, m_add)); c = np.pad(c, (0, m_add)); s = np.pad(s, (0, m_add))
            Z[j] = Pr(V[j]) if Pr is not None else V[j].copy()
            w = A @ Z[j]
            for i in range(j + 1):                             # modified Gram-Schmidt :160-163
                H[i, j] = w @ V[i]
                w = w - H[i, j] * V[i]
            H[j + 1, j] = np.linalg.norm(w)
            V[j + 1] = w / H[j + 1, j]
            for i in range(j):                                 # :168-172
                gm = c[i] * H[i, j] + s[i] * H[i + 1, j]
                H[i + 1, j] = -s[i] * H[i, j] + c[i] * H[i + 1, j]
                H[i, j] = gm
            c[j], s[j], _ = dlartg(H[j, j], H[j + 1, j])       # LinearAlgebra.givensAlgorithm :175
            H[j, j] = c[j] * H[j, j] + s[j] * H[j + 1, j]; H[j + 1, j] = 0.0
            g[j + 1] = -s[j] * g[j]; g[j] = c[j] * g[j]
            beta = abs(g[j + 1]); hist.append(beta); it += 1
            j += 1
            done = it >= maxiter or beta / hist[0] < rtol or beta < atol
        y = sla.solve_triangular(H[:j, :j], g[:j])             # :186-188
        for i in range(j):
            x += y[i] * Z[i]                                   # :191-193
        r = b - A @ x                                          # :194
    return x, it, np.array(hist)


def block_apply(kind, diag_solves, offd, coeffs, sizes, b):
    """solve!(x, BlockTriangularSolverNS|BlockDiagonalSolverNS, b): diag_solves[i](w) -> y ; offd[(i,j)] scipy matrices."""
    nb = len(sizes)
    off = np.concatenate([[0], np.cumsum(sizes)])
    x = np.zeros_like(b)
    order = range(nb - 1, -1, -1) if kind == "upper" else range(nb)
    for i in order:
        w = b[off[i]:off[i + 1]].copy()
        if kind != "diagonal":
            js = range(i + 1, nb) if kind == "upper" else range(i)
            for j in js:
                if (i, j) in offd and coeffs[i][j] != 0.0:
                    w -= coeffs[i][j] * (offd[(i, j)] @ x[off[j]:off[j + 1]])
        x[off[i]:off[i + 1]] = diag_solves[i](w)
    return x
