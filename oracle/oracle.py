"""ctypes front-end of the CPU oracle (oracle/gmg_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; never from the product package.
"parity unpinned" for V-cycle vectors (see gmg_oracle.c header).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

JACOBI, PATCH, BLOCKJACOBI = 0, 1, 2
PRECONDITIONER, SOLVER = 0, 1
V_CYCLE, W_CYCLE, F_CYCLE = 0, 1, 2

_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_ip = C.POINTER(C.c_int)


_VARIANT = "seq"      # "seq": liboracle.so (the checker) ; "omp": liboracle_omp.so (all host cores; bench.py cpu_baseline only)


def build(force=False, variant=None):
    name = {"omp": "liboracle_omp.so", "blas": "liboracle_blas.so"}.get(variant or _VARIANT, "liboracle.so")
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "gmg_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", name])
    return so


def set_variant(name):
    """Switch between the sequential checker and the OpenMP build (objects created before the switch stay on theirs)."""
    global _VARIANT, _LIB
    assert name in ("seq", "omp", "blas")
    if name != _VARIANT:
        _VARIANT, _LIB = name, None


def threads():
    return int(lib().orc_threads())


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_dot.restype = C.c_double
        L.orc_norm.restype = C.c_double
        L.orc_gmg_create.restype = C.c_void_p
        L.orc_block_create.restype = C.c_void_p
        for name in ("orc_gmg_solve", "orc_cg_solve", "orc_fgmres_solve", "orc_cg_smoother_solve", "orc_richardson_solve"):
            getattr(L, name).restype = C.c_int
    return _LIB


def _d(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_dp)


def _p64(a):
    assert a.dtype == np.int64 and a.flags.c_contiguous
    return a.ctypes.data_as(_i64p)


def _p32(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_i32p)


def spmv(A, x):
    y = np.empty(A.shape[0])
    x = np.ascontiguousarray(x, dtype=np.float64)
    lib().orc_spmv(C.c_int64(A.shape[0]), _p64(A.ptr), _p32(A.idx), _d(A.val), _d(x), _d(y))
    return y


def dot(a, b):
    return lib().orc_dot(C.c_int64(a.size), _d(a), _d(b))


def norm(a):
    return lib().orc_norm(C.c_int64(a.size), _d(a))


def jacobi_inv_diag(A):
    out = np.empty(A.shape[0])
    lib().orc_jacobi_setup(C.c_int64(A.shape[0]), _p64(A.ptr), _p32(A.idx), _d(A.val), _d(out))
    return out


def direct_solve(A, b):
    x = np.empty(A.shape[0])
    b = np.ascontiguousarray(b, dtype=np.float64)
    lib().orc_direct_solve(C.c_int64(A.shape[0]), _p64(A.ptr), _p32(A.idx), _d(A.val), _d(b), _d(x))
    return x


def givens(f, g):
    out = np.zeros(3)
    lib().orc_givens(C.c_double(f), C.c_double(g), _d(out))
    return tuple(out)


class Smoother:
    """RichardsonSmoother(M, niter, omega) with M in {Jacobi, PatchSolver, BlockJacobiSolver}."""

    def __init__(self, kind=JACOBI, niter=10, omega=2.0 / 3.0, patch_ptr=None, patch_dofs=None, patch_cols=None, patch_mats=None):
        """patch_dofs = patch_rows; patch_cols (optional) = separate column table; patch_mats (optional) = the caller's own
        patch matrices, column-major and concatenated (PatchSolvers.jl:137-150)."""
        self.kind, self.niter, self.omega = kind, int(niter), float(omega)
        self.patch_ptr = None if patch_ptr is None else np.ascontiguousarray(patch_ptr, dtype=np.int64)
        self.patch_dofs = None if patch_dofs is None else np.ascontiguousarray(patch_dofs, dtype=np.int32)
        self.patch_cols = None if patch_cols is None else np.ascontiguousarray(patch_cols, dtype=np.int32)
        self.patch_mats = None if patch_mats is None else np.ascontiguousarray(patch_mats, dtype=np.float64)


class GMG:
    """Mirror of GMGLinearSolver(smatrices, interp, restrict; ...) -> numerical setup."""

    def __init__(self, mats, prolongations, restrictions=None, pre_smoothers=None, post_smoothers=None,
                 mode=PRECONDITIONER, cycle=V_CYCLE, maxiter=100, atol=1e-14, rtol=1e-8, prolongation_patches=None,
                 coarse_cg=None):
        L = lib()
        nlev = len(mats)
        assert len(prolongations) == nlev - 1
        self._keep = [mats, prolongations, restrictions, pre_smoothers, post_smoothers]
        self.mats = mats
        self.h = C.c_void_p(L.orc_gmg_create(C.c_int(nlev)))
        for l, A in enumerate(mats):
            L.orc_gmg_set_matrix(self.h, C.c_int(l), C.c_int64(A.shape[0]), _p64(A.ptr), _p32(A.idx), _d(A.val))
        for l, P in enumerate(prolongations):
            L.orc_gmg_set_prolongation(self.h, C.c_int(l), C.c_int64(P.shape[0]), C.c_int64(P.shape[1]),
                                       _p64(P.ptr), _p32(P.idx), _d(P.val))
        if restrictions is not None:
            for l, R in enumerate(restrictions):
                L.orc_gmg_set_restriction(self.h, C.c_int(l), C.c_int64(R.shape[0]), C.c_int64(R.shape[1]),
                                          _p64(R.ptr), _p32(R.idx), _d(R.val))
        if prolongation_patches is not None:      # [(kind, patch_ptr, patch_dofs) or None per level]
            self._keep.append(prolongation_patches)
            for l, pc in enumerate(prolongation_patches):
                if pc is not None:
                    G = pc[3] if len(pc) > 3 else None
                    kind, pp, pd = pc[:3]
                    if G is not None:       # rhs form of the correction differs from the level operator
                        L.orc_gmg_set_prolongation_correction_rhs(self.h, C.c_int(l), C.c_int64(G.shape[0]), _p64(G.ptr), _p32(G.idx), _d(G.val))
                    pp = np.ascontiguousarray(pp, dtype=np.int64); pd = np.ascontiguousarray(pd, dtype=np.int32)
                    self._keep += [pp, pd]
                    L.orc_gmg_set_prolongation_correction(self.h, C.c_int(l), C.c_int(kind), C.c_int64(len(pp) - 1),
                                                          _p64(pp), _p32(pd))
        if pre_smoothers is None:
            pre_smoothers = [Smoother() for _ in range(nlev - 1)]
        for l in range(nlev - 1):
            pre = pre_smoothers[l]
            post = pre if post_smoothers is None else post_smoothers[l]
            if post is pre:
                self._set_sm(l, 2, pre)
            else:
                self._set_sm(l, 0, pre)
                self._set_sm(l, 1, post)
        if coarse_cg is not None:      # (maxiter, atol, rtol) of coarsest_solver = CGSolver(JacobiLinearSolver(); ...)
            L.orc_gmg_set_coarse_cg(self.h, C.c_int(coarse_cg[0]), C.c_double(coarse_cg[1]), C.c_double(coarse_cg[2]))
        L.orc_gmg_setup(self.h, C.c_int(mode), C.c_int(cycle), C.c_int(maxiter), C.c_double(atol), C.c_double(rtol))
        self.maxiter = maxiter

    def _set_sm(self, l, which, s):
        pp = _p64(s.patch_ptr) if s.patch_ptr is not None else None
        pd = _p32(s.patch_dofs) if s.patch_dofs is not None else None
        npatch = 0 if s.patch_ptr is None else len(s.patch_ptr) - 1
        pc = _p32(s.patch_cols) if getattr(s, "patch_cols", None) is not None else None
        pm = _d(s.patch_mats) if getattr(s, "patch_mats", None) is not None else None
        lib().orc_gmg_set_smoother_ex(self.h, C.c_int(l), C.c_int(which), C.c_int(s.kind), C.c_int(s.niter),
                                      C.c_double(s.omega), C.c_int64(npatch), pp, pd, pc, pm)

    def solve(self, b, x=None):
        """solve!(x, ns, b); returns (x, niters, flag, residual_history)."""
        n = self.mats[0].shape[0]
        x = np.zeros(n) if x is None else np.ascontiguousarray(x, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        hist = np.zeros(self.maxiter + 1)
        nit = C.c_int(0)
        flag = lib().orc_gmg_solve(self.h, _d(x), _d(b), C.byref(nit), _d(hist))
        return x, nit.value, flag, hist[: nit.value + 1].copy()

    def smooth(self, l, x, r, post=False):
        """solve!(x, RichardsonSmootherNumericalSetup, r): updates and returns (x, r)."""
        x = np.array(x, dtype=np.float64); r = np.array(r, dtype=np.float64)
        lib().orc_gmg_smooth(self.h, C.c_int(l), C.c_int(int(post)), _d(x), _d(r))
        return x, r

    def precond(self, l, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        dx = np.zeros_like(r)
        lib().orc_gmg_precond(self.h, C.c_int(l), _d(dx), _d(r))
        return dx

    def restrict(self, l, rh):
        rh = np.ascontiguousarray(rh, dtype=np.float64)
        out = np.zeros(self.mats[l + 1].shape[0])
        lib().orc_gmg_restrict(self.h, C.c_int(l), _d(rh), _d(out))
        return out

    def coarse_solve(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        out = np.zeros_like(r)
        lib().orc_gmg_coarse_solve(self.h, _d(r), _d(out))
        return out

    def __del__(self):
        try:
            if self.h:
                lib().orc_gmg_destroy(self.h)
                self.h = None
        except Exception:
            pass


DIAGONAL, LOWER, UPPER = 0, 1, 2
BD_GMG, BD_CG_JACOBI, BD_LU, BD_JACOBI = 1, 2, 3, 4


class BlockPreconditioner:
    """BlockDiagonalSolver / BlockTriangularSolver numerical setup on contiguous block vectors.

    diag[i] is a GMG (oracle) object or a tuple (kind, M[, maxiter, atol, rtol]) with kind in
    {BD_CG_JACOBI, BD_LU, BD_JACOBI}; offdiag = {(i,j): (CSR, coeff)}.
    """

    def __init__(self, sizes, diag, offdiag=None, kind=DIAGONAL):
        L = lib()
        sizes = np.ascontiguousarray(sizes, dtype=np.int64)
        self.n = int(sizes.sum())
        self._keep = [sizes, diag, offdiag]
        self.h = C.c_void_p(L.orc_block_create(C.c_int(len(sizes)), _p64(sizes), C.c_int(kind)))
        for i, d in enumerate(diag):
            if isinstance(d, GMG):
                L.orc_block_set_diag_gmg(self.h, C.c_int(i), d.h)
            else:
                k, M = d[0], d[1]
                maxiter, atol, rtol = (list(d[2:]) + [1000, 1e-12, 1e-6][len(d) - 2:])[:3]
                L.orc_block_set_diag_matrix(self.h, C.c_int(i), C.c_int(k), C.c_int64(M.shape[0]), _p64(M.ptr), _p32(M.idx),
                                            _d(M.val), C.c_int(maxiter), C.c_double(atol), C.c_double(rtol))
        for (i, j), (M, c) in (offdiag or {}).items():
            L.orc_block_set_offdiag(self.h, C.c_int(i), C.c_int(j), C.c_int64(M.shape[0]), C.c_int64(M.shape[1]),
                                    _p64(M.ptr), _p32(M.idx), _d(M.val), C.c_double(c))

    def apply(self, b):
        """solve!(x, ns, b) (stateful: the block work caches persist between calls, as in the reference)."""
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros(self.n)
        lib().orc_block_apply(self.h, _d(x), _d(b))
        return x

    def __del__(self):
        try:
            if self.h:
                lib().orc_block_destroy(self.h)
                self.h = None
        except Exception:
            pass


def _pc(A, P):
    """Preconditioner dispatch: None | GMG | "jacobi" (JacobiLinearSolver())."""
    if P is None:
        return 0, None, None
    if isinstance(P, GMG):
        return 1, P.h, P
    if isinstance(P, BlockPreconditioner):
        return 3, P.h, P
    if P == "jacobi":
        d = jacobi_inv_diag(A)
        return 2, C.cast(_d(d), C.c_void_p), d
    raise TypeError("unknown preconditioner")


def cg_solve(A, b, Pl=None, x0=None, maxiter=1000, atol=1e-12, rtol=1e-6, flexible=False):
    """solve!(x, CGNumericalSetup, b) -- returns (x, niters, flag, hist)."""
    n = A.shape[0]
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    hist = np.zeros(maxiter + 1)
    nit = C.c_int(0)
    kind, pc, _keep = _pc(A, Pl)
    flag = lib().orc_cg_solve(C.c_int64(n), _p64(A.ptr), _p32(A.idx), _d(A.val), C.c_int(kind), pc,
                              _d(x), _d(b), C.c_int(maxiter), C.c_double(atol), C.c_double(rtol),
                              C.c_int(int(flexible)), C.byref(nit), _d(hist))
    return x, nit.value, flag, hist[: nit.value + 1].copy()


def fgmres_solve(A, b, Pr=None, x0=None, m=5, restart=False, m_add=1, maxiter=100, atol=1e-12, rtol=1e-6, Pl=None):
    """solve!(x, FGMRESNumericalSetup, b) -- returns (x, niters, flag, hist).  Pl: optional left preconditioner (KrylovUtils.jl:14-18,46-50)."""
    n = A.shape[0]
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    hist = np.zeros(maxiter + 1)
    nit = C.c_int(0)
    kind, pc, _keep = _pc(A, Pr)
    lkind, lpc, _lkeep = _pc(A, Pl)
    L = lib()
    L.orc_fgmres_solve_pl.restype = C.c_int
    flag = L.orc_fgmres_solve_pl(C.c_int64(n), _p64(A.ptr), _p32(A.idx), _d(A.val),
                                 C.c_int(kind), pc, C.c_int(lkind), lpc, _d(x), _d(b), C.c_int(m),
                                 C.c_int(int(restart)), C.c_int(m_add), C.c_int(maxiter), C.c_double(atol),
                                 C.c_double(rtol), C.byref(nit), _d(hist))
    return x, nit.value, flag, hist[: nit.value + 1].copy()


def cg_smoother_solve(A, b, sm_niter=5, sm_omega=2.0 / 3.0, maxiter=1000, atol=1e-12, rtol=1e-8):
    """CGSolver(LinearSolverFromSmoother(RichardsonSmoother(Jacobi,niter,omega))) -- SmoothersTests.jl."""
    n = A.shape[0]
    x = np.zeros(n)
    b = np.ascontiguousarray(b, dtype=np.float64)
    hist = np.zeros(maxiter + 1)
    nit = C.c_int(0)
    flag = lib().orc_cg_smoother_solve(C.c_int64(n), _p64(A.ptr), _p32(A.idx), _d(A.val), C.c_int(sm_niter),
                                       C.c_double(sm_omega), _d(x), _d(b), C.c_int(maxiter), C.c_double(atol),
                                       C.c_double(rtol), C.byref(nit), _d(hist))
    return x, nit.value, flag, hist[: nit.value + 1].copy()


def richardson_solve(A, b, omega, Pl=None, x0=None, maxiter=1000, atol=1e-6, rtol=1e-10):
    """solve!(x, RichardsonLinearNumericalSetup, b) -- returns (x, niters, flag, hist).  Defaults: RichardsonLinearSolvers.jl:19."""
    n = A.shape[0]
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    hist = np.zeros(maxiter + 1)
    nit = C.c_int(0)
    kind, pc, _keep = _pc(A, Pl)
    flag = lib().orc_richardson_solve(C.c_int64(n), _p64(A.ptr), _p32(A.idx), _d(A.val), C.c_int(kind), pc, C.c_double(omega),
                                      _d(x), _d(b), C.c_int(maxiter), C.c_double(atol), C.c_double(rtol), C.byref(nit), _d(hist))
    return x, nit.value, flag, hist[: nit.value + 1].copy()
