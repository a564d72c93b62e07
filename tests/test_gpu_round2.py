"""GPU tests of the round-2 additions, same bar as tests/test_gpu_parity.py: HIP path through the C ABI against the
CPU oracle on the same inputs (iteration counts identical, histories <= 1e-8, solutions <= 1e-10)."""
import ctypes as C

import numpy as np
import pytest

from conftest import max_rel, rel_err

pytestmark = pytest.mark.gpu
TOL_HIST = 1e-8


def jac(S, nlev, niter=10, omega=2.0 / 3.0):
    return [S.RichardsonSmoother(S.JacobiLinearSolver(), niter, omega)] * (nlev - 1)


def make_gmg(S, H, **kw):
    nlev = len(H["mats"])
    kw.setdefault("pre_smoothers", jac(S, nlev))
    kw.setdefault("post_smoothers", kw["pre_smoothers"])
    kw.setdefault("maxiter", 1)
    return S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], **kw)


def setup(S, solver, A):
    return S.numerical_setup(S.symbolic_setup(solver, A), A)


# ---------------------------------------------------------------- variable coefficient: the generic 12 B/nnz path end to end
@pytest.mark.parametrize("nc,nlev", [((32, 32, 32), 3), ((24, 16, 8), 2), ((48, 48), 3)])
def test_variable_coefficient_cg_gmg_matches_oracle(S, po, orc, nc, nlev):
    """a(u,v) = int kappa(x) grad u . grad v, kappa smooth: every row distinct, so gmg_setup must fall back to the plain
    (col,val) stream by itself -- and CG+GMG must still match the oracle (iterations, history, solution)."""
    H = po.build_hierarchy(nc, nlev, 1, kappa=po.smooth_kappa)
    A = H["mats"][0]
    uex = po.nodal_values(nc, 1)
    b = A.matvec(uex)
    solver = S.CGSolver(make_gmg(S, H), maxiter=30, atol=1e-14, rtol=1e-8)
    ns = setup(S, solver, A)
    fmt = ns.P_ns.level_format(0)
    assert not fmt["row_patterns"] and not fmt["value_dictionary"], fmt     # nothing to compress
    assert fmt["stream_bytes_per_nnz"] == 12.0
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    xo, nit, flag, hist = orc.cg_solve(A, b, Pl=g, maxiter=30, atol=1e-14, rtol=1e-8)
    assert solver.log.num_iters == nit and solver.log.flag == flag
    np.testing.assert_allclose(solver.log.residuals[: nit + 1], hist, rtol=TOL_HIST)
    assert rel_err(x, xo) <= 1e-10
    assert np.max(np.abs(x - uex)) < 1e-6
    # per-kernel parity of the generic operator kernels on every level
    from gridapsolvers_jl_amd import abi
    for l in range(nlev):
        v = np.random.default_rng(l).uniform(-1, 1, H["mats"][l].shape[0])
        y = np.zeros_like(v)
        ns.P_ns.op_apply(l, abi.OP_A, v, y)
        assert max_rel(y, orc.spmv(H["mats"][l], v)) <= 1e-13
    st = ns.P_ns.kernel_stats()
    assert abs(st["layout_bytes"] / st["alg_bytes"] - 1.0) < 0.35           # generic layout: layout bytes ~ 12 B/nnz model (+ padding)


def test_layout_bytes_of_pattern_layout_is_small(S, po, hierarchy):
    H = hierarchy((32, 32, 32), 3)
    ns = setup(S, S.CGSolver(make_gmg(S, H)), H["mats"][0])
    st = ns.P_ns.kernel_stats()
    assert ns.P_ns.level_format(0)["row_patterns"]
    assert st["layout_bytes"] < 0.2 * st["alg_bytes"]                        # 2 B/row of matrix stream instead of 324
    assert ns.P_ns.stream_probe(1 << 26, 3) > 100.0                          # GB/s, sanity only


# ---------------------------------------------------------------- FGMRES workspace sized by the basis, not by maxiter
def test_fgmres_restart_with_huge_maxiter(S, po, orc, hierarchy):
    """maxiter=100000 with restart=true used to be rejected (Hessenberg sized by maxiter)."""
    nc, nlev = (32, 32), 3
    H = hierarchy(nc, nlev)
    b = po.dirichlet_lift_rhs(nc, 1)
    solver = S.FGMRESSolver(3, make_gmg(S, H), restart=True, maxiter=100000, atol=1e-14, rtol=1e-10)
    ns = setup(S, solver, H["mats"][0])
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    xo, nit, flag, hist = orc.fgmres_solve(H["mats"][0], b, Pr=g, m=3, restart=True, maxiter=100000, atol=1e-14, rtol=1e-10)
    assert solver.log.num_iters == nit and solver.log.flag == flag
    np.testing.assert_allclose(solver.log.residuals[: nit + 1], hist, rtol=1e-6)
    assert rel_err(x, xo) <= 1e-9


def test_fgmres_growth_past_initial_basis(S, po, orc, hierarchy):
    """no restart, m=2, many steps with a weak preconditioner: the basis AND the small arrays grow together (m_add=2)."""
    nc, nlev = (16, 16), 2
    H = hierarchy(nc, nlev)
    b = po.dirichlet_lift_rhs(nc, 1)
    sm = jac(S, nlev, 1, 0.3)
    solver = S.FGMRESSolver(2, make_gmg(S, H, pre_smoothers=sm), restart=False, m_add=2, maxiter=40, atol=1e-14, rtol=1e-12)
    ns = setup(S, solver, H["mats"][0])
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.JACOBI, 1, 0.3)] * (nlev - 1), maxiter=1)
    xo, nit, flag, hist = orc.fgmres_solve(H["mats"][0], b, Pr=g, m=2, restart=False, m_add=2, maxiter=40, atol=1e-14, rtol=1e-12)
    assert nit > 4                                                           # really outgrew m = 2
    assert solver.log.num_iters == nit and solver.log.flag == flag
    np.testing.assert_allclose(solver.log.residuals[: nit + 1], hist, rtol=1e-5)
    assert rel_err(x, xo) <= 1e-8


# ---------------------------------------------------------------- GMG's own log under a Krylov solver
@pytest.mark.parametrize("verbose", [0, 1])
def test_gmg_log_inside_cg(S, po, orc, hierarchy, verbose):
    """GMGLinearSolvers.jl:627-640: the GMG logs norm(rh) before and after its cycle.  verbose=1: both entries are filled
    and equal the oracle's; verbose=0 (default): the post-cycle norm is not evaluated on the CG fast path -> NaN."""
    nc, nlev = (16, 16, 16), 3
    H = hierarchy(nc, nlev)
    b = po.dirichlet_lift_rhs(nc, 1)
    gmg = make_gmg(S, H, verbose=verbose)
    solver = S.CGSolver(gmg, maxiter=20, atol=1e-14, rtol=1e-6)
    ns = setup(S, solver, H["mats"][0])
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    assert gmg.log.num_iters == 1
    # the last preconditioner application saw r of the last-but-one CG iterate: its norm is the CG history entry
    assert np.isclose(gmg.log.residuals[0], solver.log.residuals[solver.log.num_iters - 1], rtol=1e-12)
    if verbose:
        assert np.isfinite(gmg.log.residuals[1]) and 0 < gmg.log.residuals[1] < gmg.log.residuals[0]
        go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
        xo, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=go, maxiter=20, atol=1e-14, rtol=1e-6)
        assert nit == solver.log.num_iters
    else:
        assert np.isnan(gmg.log.residuals[1])


# ---------------------------------------------------------------- destruction order of a block solver and its GMG handles
@pytest.mark.parametrize("gmg_first", [True, False])
def test_block_and_gmg_handles_destroy_in_any_order(S, po, pkg, gmg_first):
    """gmg_block_destroy used to dereference borrowed GMG handles that a finalizer may already have freed."""
    lib = pkg.abi.load()
    abi = pkg.abi
    nc, nlev = (8, 8), 2
    H = po.build_hierarchy(nc, nlev, 1)
    A = H["mats"][0]; n = A.shape[0]
    gmg = make_gmg(S, H)
    solver = S.BlockDiagonalSolver([gmg, S.LUSolver()])
    ns = setup(S, solver, [[A, None], [None, A]])
    b = np.random.default_rng(0).uniform(-1, 1, 2 * n); x = np.zeros(2 * n)
    S.solve_(x, ns, b)
    g = ns.block_ns[0]
    if gmg_first:
        assert lib.gmg_destroy(g.h) == abi.OK                               # the block handle still borrows it
        g.h = None
        # the block solver noticed: it refuses to run until it is given a solver for that block again
        st = lib.gmg_block_precond_apply(ns.h, C.c_void_p(b.ctypes.data), C.c_void_p(x.ctypes.data), abi.MEM_HOST)
        assert st == abi.ERR_STATE
        assert lib.gmg_block_destroy(ns.h) == abi.OK
        ns.h = None
    else:
        assert lib.gmg_block_destroy(ns.h) == abi.OK
        ns.h = None
        # the GMG handle got its own stream back and still works
        z = np.zeros(n)
        S.solve_(z, g, b[:n])
        assert np.all(np.isfinite(z)) and np.linalg.norm(z) > 0
        g.close()
