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


# ---------------------------------------------------------------- variable coefficient: the generic paths end to end
@pytest.mark.parametrize("layout", ["SELL-O", "SELL-64"])
@pytest.mark.parametrize("nc,nlev", [((32, 32, 32), 3), ((24, 16, 8), 2), ((48, 48), 3)])
def test_variable_coefficient_cg_gmg_matches_oracle(S, po, orc, nc, nlev, layout, monkeypatch):
    """a(u,v) = int kappa(x) grad u . grad v, kappa smooth: every row's values distinct, so gmg_setup must give up the value
    dictionary / row-pattern layouts by itself -- it keeps the 8 B/nnz value stream and either finds the column structure
    repeating (SELL-O, the default) or streams the columns too (SELL-64, 12 B/nnz: GMG_OPATTERN=0, what an unstructured
    operator gets) -- and CG+GMG must still match the oracle (iterations, history, solution)."""
    monkeypatch.setenv("GMG_OPATTERN", "1" if layout == "SELL-O" else "0")
    H = po.build_hierarchy(nc, nlev, 1, kappa=po.smooth_kappa)
    A = H["mats"][0]
    uex = po.nodal_values(nc, 1)
    b = A.matvec(uex)
    solver = S.CGSolver(make_gmg(S, H), maxiter=30, atol=1e-14, rtol=1e-8)
    ns = setup(S, solver, A)
    fmt = ns.P_ns.level_format(0)
    assert not fmt["row_patterns"] and not fmt["value_dictionary"], fmt     # no value can be shared
    assert fmt["layout"] == layout, fmt
    if layout == "SELL-64":
        assert fmt["stream_bytes_per_nnz"] == 12.0
    else:
        assert 8.0 < fmt["stream_bytes_per_nnz"] < 8.5
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
    if layout == "SELL-64":
        assert abs(st["layout_bytes"] / st["alg_bytes"] - 1.0) < 0.35       # generic layout: layout bytes ~ 12 B/nnz model (+ padding)
    else:
        assert 0.55 < st["layout_bytes"] / st["alg_bytes"] < 0.95           # the column stream is gone


def test_offset_pattern_layout_is_bitwise_the_plain_stream(S, po, orc, monkeypatch):
    """SELL-O and SELL-64 multiply the same (col,val) pairs in the same order: operator products, V-cycle and CG history must
    agree to the last bit -- on a square operator (offsets relative to the row) and, through a level whose operator has
    more columns than a diagonal-relative table can describe, on the base-column form."""
    from gridapsolvers_jl_amd import abi
    nc, nlev = (20, 16, 12), 3
    H = po.build_hierarchy(nc, nlev, 1, kappa=po.smooth_kappa)
    # rectangular operators with row-dependent values on a repeating structure: P / R scaled row by row
    rng = np.random.default_rng(7)
    for key in ("prolongations", "restrictions"):
        for M in H[key]:
            scale = rng.uniform(0.5, 1.5, M.shape[0])
            M.val[:] = M.val * np.repeat(scale, np.diff(M.ptr))
    b = H["mats"][0].matvec(po.nodal_values(nc, 1))
    out = {}
    for name, flag in (("plain", "0"), ("offsets", "1")):
        monkeypatch.setenv("GMG_OPATTERN", flag)
        solver = S.FGMRESSolver(8, make_gmg(S, H), maxiter=30, atol=1e-14, rtol=1e-8)
        ns = setup(S, solver, H["mats"][0])
        assert ns.P_ns.level_format(0)["layout"] == ("SELL-O" if flag == "1" else "SELL-64")
        x = np.zeros_like(b)
        S.solve_(x, ns, b)
        prods = []
        for l in range(nlev - 1):
            for op, M in ((abi.OP_A, H["mats"][l]), (abi.OP_P, H["prolongations"][l]), (abi.OP_R, H["restrictions"][l])):
                v = np.random.default_rng(l).uniform(-1, 1, M.shape[1])
                y = np.zeros(M.shape[0])
                ns.P_ns.op_apply(l, op, v, y)
                assert max_rel(y, orc.spmv(M, v)) <= 1e-13
                prods.append(y)
        out[name] = (x, solver.log.residuals[: solver.log.num_iters + 1].copy(), prods)
    np.testing.assert_array_equal(out["plain"][0], out["offsets"][0])
    np.testing.assert_array_equal(out["plain"][1], out["offsets"][1])
    for a, c in zip(out["plain"][2], out["offsets"][2]):
        np.testing.assert_array_equal(a, c)


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


# ---------------------------------------------------------------- coarsest_solver hook (GMGLinearSolvers.jl:54,423-434)
@pytest.mark.parametrize("cg_maxiter", [1000, 3])
def test_coarse_solver_cg_jacobi_matches_oracle(S, po, orc, hierarchy, cg_maxiter):
    """coarsest_solver = CGSolver(JacobiLinearSolver(); maxiter, atol, rtol) on the device, also truncated (maxiter=3):
    outer CG iteration count, history and solution equal the oracle's with the same coarse solver."""
    nc, nlev = (16, 16, 16), 3
    H = hierarchy(nc, nlev)
    b = po.dirichlet_lift_rhs(nc, 1)
    cs = S.CGSolver(S.JacobiLinearSolver(), maxiter=cg_maxiter, atol=1e-14, rtol=1e-10)
    solver = S.CGSolver(make_gmg(S, H, coarsest_solver=cs), maxiter=20, atol=1e-14, rtol=1e-6, flexible=(cg_maxiter == 3))
    ns = setup(S, solver, H["mats"][0])
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1, coarse_cg=(cg_maxiter, 1e-14, 1e-10))
    xo, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=g, maxiter=20, atol=1e-14, rtol=1e-6, flexible=(cg_maxiter == 3))
    assert solver.log.num_iters == nit and solver.log.flag == flag
    np.testing.assert_allclose(solver.log.residuals[: nit + 1], hist, rtol=1e-7)
    assert rel_err(x, xo) <= 1e-9
    cl = ns.P_ns.coarse_log()
    assert cl["niters"] == orc.lib().orc_gmg_coarse_niters(g.h) and cl["niters"] <= cg_maxiter
    # standalone coarse solve through the ABI
    nL = H["mats"][-1].shape[0]
    rc = np.random.default_rng(1).uniform(-1, 1, nL); xc = np.full(nL, 7.0)
    ns.P_ns.coarse_solve(rc, xc)
    assert rel_err(xc, g.coarse_solve(rc)) <= 1e-10


def test_coarse_solver_host_callback(S, po, orc, hierarchy):
    """GMG_COARSE_HOST_CALLBACK: the host language's own exact solver (here scipy's sparse LU, the analogue of LUSolver() /
    PETSc in the Julia host) gives the same CG iteration as the device's dense inverse."""
    import scipy.sparse.linalg as spla
    nc, nlev = (16, 16, 16), 3
    H = hierarchy(nc, nlev)
    b = po.dirichlet_lift_rhs(nc, 1)
    lu = spla.splu(H["mats"][-1].to_scipy().tocsc())
    calls = []

    def coarse(r):
        calls.append(r.size)
        return lu.solve(r)
    solver = S.CGSolver(make_gmg(S, H, coarsest_solver=S.HostCallbackSolver(coarse)), maxiter=20, atol=1e-14, rtol=1e-6)
    ns = setup(S, solver, H["mats"][0])
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    xo, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=g, maxiter=20, atol=1e-14, rtol=1e-6)
    assert solver.log.num_iters == nit and len(calls) == nit
    np.testing.assert_allclose(solver.log.residuals[: nit + 1], hist, rtol=TOL_HIST)
    assert rel_err(x, xo) <= 1e-10
    # a failing callback surfaces as an error status, not as a crash or a silent wrong answer
    bad = S.CGSolver(make_gmg(S, H, coarsest_solver=S.HostCallbackSolver(lambda r: 1 / 0)), maxiter=20, atol=1e-14, rtol=1e-6)
    nsb = setup(S, bad, H["mats"][0])
    from gridapsolvers_jl_amd import abi
    with pytest.raises(abi.GmgError):
        S.solve_(np.zeros_like(b), nsb, b)


def test_nested_cg_inside_cg_keeps_scalars_apart(S, po, orc):
    """outer CG on a block system whose diagonal blocks are CG-Jacobi solves: both cg_core instances live on one engine and
    must not share their device scalar slots (gamma / dot(p,w))."""
    M = po.poisson_matrix((12, 12), 1); n = M.shape[0]
    inner = S.CGSolver(S.JacobiLinearSolver(), maxiter=200, atol=1e-14, rtol=1e-12)
    solver = S.CGSolver(S.BlockDiagonalSolver([inner, inner]), maxiter=30, atol=1e-14, rtol=1e-10, flexible=True)
    ns = setup(S, solver, [[M, None], [None, M]])
    b = np.random.default_rng(3).uniform(-1, 1, 2 * n); x = np.zeros(2 * n)
    S.solve_(x, ns, b)
    import scipy.sparse.linalg as spla
    lu = spla.splu(M.to_scipy().tocsc())
    xe = np.concatenate([lu.solve(b[:n]), lu.solve(b[n:])])
    assert solver.log.num_iters <= 3                         # an (almost) exact preconditioner
    assert rel_err(x, xe) <= 1e-8


# ---------------------------------------------------------------- caller-supplied patch matrices / factors, patch_rows != patch_cols
def _patch_blocks(A, pp, rows, cols=None, shift=0.0, jitter=None):
    """column-major, concatenated blocks A[rows_p, cols_p] (+ shift*I, + a per-patch relative perturbation)"""
    Sd = A.to_scipy().tocsr()
    cols = rows if cols is None else cols
    out = []
    for p in range(pp.size - 1):
        r, c = rows[pp[p]:pp[p + 1]], cols[pp[p]:pp[p + 1]]
        B = Sd[r][:, c].toarray() + shift * np.eye(r.size)
        if jitter is not None:
            B = B * (1.0 + 1e-3 * jitter.uniform(-1, 1, B.shape))
        out.append(B.reshape(-1, order="F"))
    return np.concatenate(out)


@pytest.mark.parametrize("variant", ["mats", "factors", "distinct"])
def test_patch_smoother_with_caller_matrices(S, po, orc, hierarchy, variant):
    """PatchSolvers.jl:137-150,175-188: the patch matrices come from the SOLVER's weak form (here A[p,p] + 0.5 I, i.e. a
    reaction term the system matrix does not have), optionally already factorised by lu! (LAPACK getrf via scipy)."""
    import scipy.linalg as sla
    nc, nlev, order = (8, 8, 8), 2, 2
    H = hierarchy(nc, nlev, order)
    A = H["mats"][0]
    pp, pd = po.vertex_star_patches(nc, order)
    rng = np.random.default_rng(5) if variant == "distinct" else None      # every block distinct: de-duplication must give up
    blocks = _patch_blocks(A, pp, pd, shift=0.5, jitter=rng)
    if variant == "factors":
        fac, piv = [], []
        off = 0
        for p in range(pp.size - 1):
            n_p = int(pp[p + 1] - pp[p])
            lu, ip = sla.lu_factor(blocks[off:off + n_p * n_p].reshape(n_p, n_p, order="F"))
            fac.append(lu.reshape(-1, order="F")); piv.append(ip.astype(np.int32) + 1)   # LAPACK ipiv is 1-based
            off += n_p * n_p
        M = S.PatchSolver(pp, pd, factors=np.concatenate(fac), pivots=np.concatenate(piv))
    else:
        M = S.PatchSolver(pp, pd, patch_mats=blocks)
    gmg = make_gmg(S, H, pre_smoothers=[S.RichardsonSmoother(M, 5, 0.2)])
    ns = setup(S, gmg, A)
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"],
                 pre_smoothers=[orc.Smoother(orc.PATCH, 5, 0.2, pp, pd, patch_mats=blocks)], maxiter=1)
    r = np.random.default_rng(17).uniform(-1, 1, A.shape[0])
    dx = np.zeros_like(r)
    ns.precond(0, r, dx)
    assert max_rel(dx, go.precond(0, r)) <= 1e-12
    z = np.zeros_like(r)
    S.solve_(z, ns, r)
    zo, _, _, ho = go.solve(r)
    assert rel_err(z, zo) <= 1e-11
    np.testing.assert_allclose(gmg.log.residuals[:2], ho, rtol=TOL_HIST)
    # the caller's matrices really are used: the A[p,p] smoother gives something else
    ns0 = setup(S, make_gmg(S, H, pre_smoothers=[S.RichardsonSmoother(S.PatchSolver(pp, pd), 5, 0.2)]), A)
    dx0 = np.zeros_like(r); ns0.precond(0, r, dx0)
    assert max_rel(dx0, dx) > 1e-3


@pytest.mark.parametrize("kind", ["patch", "block"])
def test_patch_rows_differ_from_cols(S, po, orc, hierarchy, kind):
    """separate patch_rows / patch_cols tables (PatchSolvers.jl:237-240,296; BlockJacobiSolvers.jl:141-170): x[cols_p] += A[rows_p,cols_p]^-1 b[rows_p]
    with the columns of every patch listed in another order than its rows."""
    nc, nlev, order = (16, 16), 2, 2
    H = hierarchy(nc, nlev, order)
    A = H["mats"][0]
    pp, pd = po.vertex_star_patches(nc, order)
    pc = pd.copy()
    if kind == "patch":                                                     # (a column permutation needs pivoting: NoPivot keeps cols == rows,
        for p in range(pp.size - 1):                                        #  passed as a separate table)
            pc[pp[p]:pp[p + 1]] = pd[pp[p]:pp[p + 1]][::-1]                # same dofs, reversed
    M = (S.PatchSolver if kind == "patch" else S.BlockJacobiSolver)(pp, pd, patch_cols=pc)
    gmg = make_gmg(S, H, pre_smoothers=[S.RichardsonSmoother(M, 4, 0.2)])
    ns = setup(S, gmg, A)
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"],
                 pre_smoothers=[orc.Smoother(orc.PATCH if kind == "patch" else orc.BLOCKJACOBI, 4, 0.2, pp, pd, patch_cols=pc)], maxiter=1)
    r = np.random.default_rng(3).uniform(-1, 1, A.shape[0])
    dx = np.zeros_like(r)
    ns.precond(0, r, dx)
    assert max_rel(dx, go.precond(0, r)) <= 1e-12
    z = np.zeros_like(r)
    S.solve_(z, ns, r)
    zo, _, _, ho = go.solve(r)
    assert rel_err(z, zo) <= 1e-11
    # and (mathematically) the same smoother as rows == cols
    ns1 = setup(S, make_gmg(S, H, pre_smoothers=[S.RichardsonSmoother(type(M)(pp, pd), 4, 0.2)]), A)
    z1 = np.zeros_like(r); S.solve_(z1, ns1, r)
    assert rel_err(z, z1) <= 1e-10


# ---------------------------------------------------------------- streamed operators (gmg_set_operator_rows)
@pytest.mark.parametrize("nc,nlev,order", [((16, 16, 16), 3, 1), ((8, 8, 8), 2, 2), ((24, 16), 3, 2)])
def test_streamed_operators_bitwise_equal_whole(S, po, orc, nc, nlev, order):
    """the hierarchy handed over as row-block streams (A, P, R of every level but the coarsest) gives bit-identical
    mat-vecs and V-cycles to the same hierarchy passed whole; the library never sees a CSR of the streamed operators."""
    from gridapsolvers_jl_amd import abi
    Hw = po.build_hierarchy(nc, nlev, order)
    nL = Hw["mats"][-1].shape[0]
    Hs = po.build_hierarchy(nc, nlev, order, stream_min_rows=nL + 1)        # everything above the coarsest level is streamed
    assert all(hasattr(M, "row_blocks") for M in Hs["mats"][:-1]) and not hasattr(Hs["mats"][-1], "row_blocks")
    nw = setup(S, make_gmg(S, Hw), Hw["mats"][0])
    ns = setup(S, make_gmg(S, Hs), Hs["mats"][0])
    assert ns.level_format(0)["row_patterns"]
    for l in range(nlev - 1):
        A, P = Hw["mats"][l], Hw["prolongations"][l]
        x = np.random.default_rng(l).uniform(-1, 1, A.shape[0])
        xc = np.random.default_rng(10 + l).uniform(-1, 1, P.shape[1])
        for op, v, m in ((abi.OP_A, x, A.shape[0]), (abi.OP_P, xc, P.shape[0]), (abi.OP_R, x, P.shape[1])):
            y1, y2 = np.zeros(m), np.zeros(m)
            nw.op_apply(l, op, v, y1); ns.op_apply(l, op, v, y2)
            if order == 1 or op == abi.OP_A:
                np.testing.assert_array_equal(y1, y2)
            else:       # Q2 transfers passed whole may take the CSR-stream kernel (G lanes per row, tree sum): same numbers to rounding
                assert max_rel(y2, y1) <= 1e-14
        assert max_rel(y2, orc.spmv(Hw["restrictions"][l], x)) <= 1e-13
    r = np.random.default_rng(99).uniform(-1, 1, Hw["mats"][0].shape[0])
    z1, z2 = np.zeros_like(r), np.zeros_like(r)
    S.solve_(z1, nw, r); S.solve_(z2, ns, r)
    if order == 1:
        np.testing.assert_array_equal(z1, z2)
    else:
        assert rel_err(z2, z1) <= 1e-13


def test_streamed_q2_patch_smoother_blocks_from_pattern(S, po, orc):
    """patch blocks A[p,p] gathered from the ROW-PATTERN form (no CSR exists for a streamed operator): same smoother, same
    FGMRES iteration as with the whole matrix and as the oracle."""
    nc, nlev, order = (8, 8, 8), 2, 2
    Hw = po.build_hierarchy(nc, nlev, order)
    Hs = po.build_hierarchy(nc, nlev, order, stream_min_rows=Hw["mats"][-1].shape[0] + 1)
    pp, pd = po.vertex_star_patches(nc, order)
    sm = [S.RichardsonSmoother(S.PatchSolver(pp, pd), 10, 0.2)]
    b = po.dirichlet_lift_rhs(nc, order)
    xs = []
    for H in (Hw, Hs):
        solver = S.FGMRESSolver(5, make_gmg(S, H, pre_smoothers=sm), maxiter=20, atol=1e-14, rtol=1e-8)
        ns = setup(S, solver, H["mats"][0])
        x = np.zeros_like(b)
        S.solve_(x, ns, b)
        xs.append((x, solver.log.num_iters, solver.log.residuals[: solver.log.num_iters + 1].copy()))
    assert xs[0][1] == xs[1][1]
    np.testing.assert_allclose(xs[1][2], xs[0][2], rtol=1e-12)
    assert rel_err(xs[1][0], xs[0][0]) <= 1e-13
    go = orc.GMG(Hw["mats"], Hw["prolongations"], Hw["restrictions"], pre_smoothers=[orc.Smoother(orc.PATCH, 10, 0.2, pp, pd)], maxiter=1)
    xo, nit, flag, hist = orc.fgmres_solve(Hw["mats"][0], b, Pr=go, m=5, maxiter=20, atol=1e-14, rtol=1e-8)
    assert nit == xs[1][1]
    np.testing.assert_allclose(xs[1][2], hist, rtol=1e-6)
    assert rel_err(xs[1][0], xo) <= 1e-9


def test_streamed_operator_errors(S, po, pkg):
    lib, abi = pkg.abi.load(), pkg.abi
    nc = (8, 8, 8)
    A = po.poisson_matrix(nc, 1)
    n = A.shape[0]
    h = C.c_void_p()
    assert lib.gmg_create(C.byref(h), 2, 0) == abi.OK
    blocks = list(po.poisson_matrix_stream(nc, 1).row_blocks())

    def send(row0, B, op=abi.OP_A, nrows=n, ncols=n):
        ptr = B.ptr.astype(np.int64); idx = B.idx.astype(np.int64)
        return lib.gmg_set_operator_rows(h, 0, op, nrows, ncols, row0, B.shape[0], C.c_void_p(ptr.ctypes.data), C.c_void_p(idx.ctypes.data),
                                         C.c_void_p(B.val.ctypes.data), 0, 8)
    assert send(*blocks[0]) == abi.OK
    assert send(*blocks[2]) == abi.ERR_STATE                                # out of order
    assert send(*blocks[1]) == abi.OK
    assert lib.gmg_setup(h) == abi.ERR_STATE                                # incomplete stream: no matrix on level 0 yet
    # a variable-coefficient operator is not pattern-compressible: rejected, not silently mangled
    V = po.poisson_matrix_varcoef((24, 24, 24))
    ptr = V.ptr.astype(np.int64); idx = V.idx.astype(np.int64)
    st = lib.gmg_set_operator_rows(h, 0, abi.OP_A, V.shape[0], V.shape[1], 0, V.shape[0], C.c_void_p(ptr.ctypes.data), C.c_void_p(idx.ctypes.data),
                                   C.c_void_p(V.val.ctypes.data), 0, 8)
    assert st == abi.ERR_UNSUPPORTED and b"4096" in lib.gmg_last_error(h)
    # unsorted columns
    B = blocks[0][1]
    bad = B.idx.astype(np.int64).copy(); bad[[0, 1]] = bad[[1, 0]]
    ptr = B.ptr.astype(np.int64)
    st = lib.gmg_set_operator_rows(h, 0, abi.OP_A, n, n, 0, B.shape[0], C.c_void_p(ptr.ctypes.data), C.c_void_p(bad.ctypes.data),
                                   C.c_void_p(B.val.ctypes.data), 0, 8)
    assert st == abi.ERR_UNSUPPORTED
    lib.gmg_destroy(h)


# ---------------------------------------------------------------- BASELINE config 3 shape at the largest size the oracle affords
def test_config3_shape_q2_32cubed_vs_oracle(S, po, orc):
    """3-D Poisson Q2 32^3, 4 levels, Richardson(PatchSolver,10,0.2) pre = post on every level, FGMRES(5), rtol 1e-6
    (test/LinearSolvers/GMGTests.jl:18-47,119-123): iteration count identical to the oracle, residual history <= 1e-6,
    solution <= 1e-9, L2 error below the reference's 1e-8 criterion.  The two finest levels are streamed."""
    nc, nlev, order = (32, 32, 32), 4, 2
    Hw = po.build_hierarchy(nc, nlev, order)
    Hs = po.build_hierarchy(nc, nlev, order, stream_min_rows=20000)
    assert hasattr(Hs["mats"][0], "row_blocks") and hasattr(Hs["mats"][1], "row_blocks") and not hasattr(Hs["mats"][2], "row_blocks")
    tabs = [po.vertex_star_patches(c, order) for c in Hw["ncells"][:-1]]
    sm = [S.RichardsonSmoother(S.PatchSolver(pp, pd), 10, 0.2) for pp, pd in tabs]
    solver = S.FGMRESSolver(5, make_gmg(S, Hs, pre_smoothers=sm), maxiter=20, atol=1e-14, rtol=1e-6)
    ns = setup(S, solver, Hs["mats"][0])
    b = po.dirichlet_lift_rhs(nc, order)
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    go = orc.GMG(Hw["mats"], Hw["prolongations"], Hw["restrictions"],
                 pre_smoothers=[orc.Smoother(orc.PATCH, 10, 0.2, pp, pd) for pp, pd in tabs], maxiter=1)
    xo, nit, flag, hist = orc.fgmres_solve(Hw["mats"][0], b, Pr=go, m=5, maxiter=20, atol=1e-14, rtol=1e-6)
    assert solver.log.num_iters == nit and solver.log.flag == flag
    np.testing.assert_allclose(solver.log.residuals[: nit + 1], hist, rtol=1e-6)
    assert rel_err(x, xo) <= 1e-9
    assert po.l2_error_sq(nc, order, x) < 1e-8
    # true residual through the device operator
    from gridapsolvers_jl_amd import abi
    y = np.zeros_like(b)
    ns.P_ns.op_apply(0, abi.OP_A, x, y)
    assert np.linalg.norm(b - y) <= 1.01e-6 * np.linalg.norm(b)


# ---------------------------------------------------------------- numerical_setup! value refresh
@pytest.mark.parametrize("smoother", ["jacobi", "patch"])
def test_value_refresh_is_bitwise_a_fresh_setup(S, po, orc, smoother, monkeypatch):
    """numerical_setup!(ns, A) with new values on the same sparsity (GMGLinearSolvers.jl:260-297): the refresh path keeps
    layouts / tables / work vectors and must give bit-identical results to a fresh numerical_setup on the new matrices."""
    import time
    monkeypatch.setenv("GMG_COARSE_HOST_MAX", "100000")      # same (host) coarse factorisation in both runs
    nc, nlev = (24, 24, 24), 3
    k1 = po.smooth_kappa
    k2 = lambda X, Y, Z: 2.0 + np.cos(3.0 * X) * np.sin(2.0 * Y + 0.2) + 0.5 * Z * Z
    H1 = po.build_hierarchy(nc, nlev, 1, kappa=k1)
    H2 = po.build_hierarchy(nc, nlev, 1, kappa=k2)
    assert all((a.ptr == b.ptr).all() and (a.idx == b.idx).all() for a, b in zip(H1["mats"], H2["mats"]))
    if smoother == "patch":
        # Q1 vertex-star patches are single dofs; use 2x2x2-node blocks to get real dense patch blocks
        n1 = nc[0] - 1
        def blocks(n):
            ids = np.arange(n ** 3).reshape(n, n, n)
            pp, pd = [0], []
            for k in range(0, n, 2):                                         # covers every dof (n odd: thinner last blocks)
                for j in range(0, n, 2):
                    for i in range(0, n, 2):
                        blk = ids[k:k + 2, j:j + 2, i:i + 2].reshape(-1)
                        pd.append(blk); pp.append(pp[-1] + blk.size)
            return np.asarray(pp, dtype=np.int64), np.concatenate(pd).astype(np.int32)
        tabs = [blocks(c[0] - 1) for c in H1["ncells"][:-1]]
        mk = lambda: [S.RichardsonSmoother(S.BlockJacobiSolver(pp, pd), 4, 0.5) for pp, pd in tabs]
    else:
        mk = lambda: jac(S, nlev)
    uex = po.nodal_values(nc, 1)
    b2 = H2["mats"][0].matvec(uex)
    solver = S.CGSolver(make_gmg(S, H1, pre_smoothers=mk()), maxiter=40, atol=1e-14, rtol=1e-8, flexible=(smoother == "patch"))
    ns = setup(S, solver, H1["mats"][0])
    x0 = np.zeros_like(b2); S.solve_(x0, ns, H1["mats"][0].matvec(uex))       # use the first operator once
    t0 = time.perf_counter()
    S.numerical_setup_(ns, H2["mats"][0], H2["mats"])                          # refresh every level
    t_refresh = time.perf_counter() - t0
    x = np.zeros_like(b2)
    S.solve_(x, ns, b2)
    it_refresh, hist_refresh = solver.log.num_iters, solver.log.residuals[: solver.log.num_iters + 1].copy()
    if it_refresh >= 40:
        pytest.fail("refreshed solver did not converge")
    solver2 = S.CGSolver(make_gmg(S, H2, pre_smoothers=mk()), maxiter=40, atol=1e-14, rtol=1e-8, flexible=(smoother == "patch"))
    t0 = time.perf_counter()
    ns2 = setup(S, solver2, H2["mats"][0])
    t_fresh = time.perf_counter() - t0
    xf = np.zeros_like(b2)
    S.solve_(xf, ns2, b2)
    from gridapsolvers_jl_amd import abi
    for l in range(nlev - 1):                                                  # level by level: operator, smoother's inner solve
        v = np.random.default_rng(l).uniform(-1, 1, H2["mats"][l].shape[0])
        y1, y2 = np.zeros_like(v), np.zeros_like(v)
        ns.P_ns.op_apply(l, abi.OP_A, v, y1); ns2.P_ns.op_apply(l, abi.OP_A, v, y2)
        np.testing.assert_array_equal(y1, y2, err_msg=f"operator of level {l} after refresh")
        ns.P_ns.precond(l, v, y1); ns2.P_ns.precond(l, v, y2)
        np.testing.assert_array_equal(y1, y2, err_msg=f"smoother blocks / D^-1 of level {l} after refresh")
    assert it_refresh == solver2.log.num_iters
    np.testing.assert_array_equal(hist_refresh, solver2.log.residuals[: solver2.log.num_iters + 1])
    np.testing.assert_array_equal(x, xf)
    assert np.max(np.abs(x - uex)) < 1e-6
    # (which path is cheaper is a measurement, not a parity property: first use of the refill kernels pays their code-object load;
    #  DESIGN.md quotes the setup timings)
    print(f"refresh {t_refresh * 1e3:.1f} ms, fresh setup {t_fresh * 1e3:.1f} ms")
    # a compressed (pattern) layout depends on the values: update falls back to a full setup and is still right
    Hc = po.build_hierarchy(nc, nlev, 1)
    sc = S.CGSolver(make_gmg(S, Hc), maxiter=20, atol=1e-14, rtol=1e-8)
    nsc = setup(S, sc, Hc["mats"][0])
    scaled = po.CSR(Hc["mats"][0].shape, Hc["mats"][0].ptr, Hc["mats"][0].idx, 2.0 * Hc["mats"][0].val)
    S.numerical_setup_(nsc, scaled)
    bc = scaled.matvec(uex); xc = np.zeros_like(bc)
    S.solve_(xc, nsc, bc)
    assert np.max(np.abs(xc - uex)) < 1e-6


# ---------------------------------------------------------------- FGMRES with a left preconditioner (KrylovUtils.jl:14-18,46-50)
@pytest.mark.parametrize("restart", [False, True])
def test_fgmres_left_preconditioner(S, po, orc, hierarchy, restart):
    """FGMRESSolver(m, Pr = GMG; Pl = JacobiLinearSolver()): krylov_residual! / krylov_mul! with Pl -- the history is that of
    the LEFT-preconditioned residual; iteration count, history and solution equal the oracle's."""
    nc, nlev = (24, 24), 3
    H = hierarchy(nc, nlev)
    b = po.dirichlet_lift_rhs(nc, 1)
    solver = S.FGMRESSolver(3, make_gmg(S, H), Pl=S.JacobiLinearSolver(), restart=restart, maxiter=30, atol=1e-14, rtol=1e-9)
    ns = setup(S, solver, H["mats"][0])
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    xo, nit, flag, hist = orc.fgmres_solve(H["mats"][0], b, Pr=g, Pl="jacobi", m=3, restart=restart, maxiter=30, atol=1e-14, rtol=1e-9)
    assert solver.log.num_iters == nit and solver.log.flag == flag
    np.testing.assert_allclose(solver.log.residuals[: nit + 1], hist, rtol=1e-6)
    assert rel_err(x, xo) <= 1e-9
    # and it differs from the un-left-preconditioned history (the norm is another one)
    _, _, _, h0 = orc.fgmres_solve(H["mats"][0], b, Pr=g, m=3, restart=restart, maxiter=30, atol=1e-14, rtol=1e-9)
    assert abs(h0[0] - hist[0]) > 1e-6 * h0[0]


# ---------------------------------------------------------------- eager row-pattern form of big structured CSR inputs
def test_eager_pattern_form_of_whole_matrices(S, po, orc, monkeypatch):
    """gmg_set_matrix / gmg_set_restriction on a big structured CSR keep only the row-pattern form (no host copy, no CSR upload):
    bit-identical solves to the general path, less device memory during setup, numerical_setup! still works."""
    nc, nlev = (48, 48, 48), 3                                               # 103 823 rows on the finest level
    H = po.build_hierarchy(nc, nlev, 1)
    b = po.dirichlet_lift_rhs(nc, 1)
    out = {}
    for eager in ("1", "0"):
        monkeypatch.setenv("GMG_EAGER", eager)
        solver = S.CGSolver(make_gmg(S, H), maxiter=20, atol=1e-14, rtol=1e-8)
        ns = setup(S, solver, H["mats"][0])
        x = np.zeros_like(b)
        S.solve_(x, ns, b)
        out[eager] = (x, solver.log.num_iters, solver.log.residuals[: solver.log.num_iters + 1].copy(), ns)
    assert out["1"][1] == out["0"][1]
    np.testing.assert_array_equal(out["1"][2], out["0"][2])
    np.testing.assert_array_equal(out["1"][0], out["0"][0])
    # numerical_setup! on a level that has no CSR copy: the mirror hands the matrix over again
    monkeypatch.setenv("GMG_EAGER", "1")
    ns = out["1"][3]
    A2 = po.CSR(H["mats"][0].shape, H["mats"][0].ptr, H["mats"][0].idx, 3.0 * H["mats"][0].val)
    S.numerical_setup_(ns, A2)
    x2 = np.zeros_like(b)
    S.solve_(x2, ns, 3.0 * b)
    assert rel_err(x2, out["1"][0]) <= 1e-6                                  # another system (3A x = 3b, coarse levels unscaled), solved to rtol 1e-8


# ---------------------------------------------------------------- whole smoothing pass in one launch (small levels)
@pytest.mark.parametrize("nc,nlev,niter", [((64, 64, 64), 4, 10), ((80, 80, 80), 3, 5), ((96, 96, 96), 3, 10), ((256, 256), 3, 7),
                                           ((16, 16, 16), 3, 2)])
def test_persistent_smoothing_pass_is_bitwise_the_sweep_loop(S, po, hierarchy, monkeypatch, nc, nlev, niter):
    """sells_smooth_kernel keeps r, x and the row's own s in registers for the whole pass and hands s between workgroups with
    agent-scope stores/loads + per-workgroup progress words; the arithmetic is that of the per-sweep kernel.  Smoothing
    passes (x given and x = 0 entry), repeated V-cycles (stale-data hazards would show as run-to-run differences) and a
    CG solve must agree to the last bit with GMG_PERSIST=0."""
    H = hierarchy(nc, nlev)
    n = H["mats"][0].shape[0]
    b = po.dirichlet_lift_rhs(nc, 1)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("GMG_PERSIST", flag)
        solver = S.CGSolver(make_gmg(S, H, pre_smoothers=jac(S, nlev, niter)), maxiter=30, atol=1e-14, rtol=1e-8)
        ns = setup(S, solver, H["mats"][0])
        out = []
        for l in range(nlev - 1):
            nl = H["mats"][l].shape[0]
            x, r = np.random.default_rng(3 + l).uniform(-1, 1, nl), np.random.default_rng(50 + l).uniform(-1, 1, nl)
            for _ in range(3):                                       # passes chained: flags / epochs advance between launches
                ns.P_ns.smooth(l, x, r)
            out += [x, r]
        z = np.zeros(n)
        for rep in range(25):
            rr = np.random.default_rng(100 + rep % 3).uniform(-1, 1, n)
            S.solve_(z, ns.P_ns, rr)
            if rep < 3:
                out.append(z.copy())
            else:
                np.testing.assert_array_equal(z, out[2 * (nlev - 1) + rep % 3], err_msg=f"V-cycle {rep} differs from its first run (persist={flag})")
        x = np.zeros(n)
        S.solve_(x, ns, b)
        out += [x, solver.log.residuals[: solver.log.num_iters + 1].copy()]
        res[flag] = out
    for a, c in zip(res["0"], res["1"]):
        np.testing.assert_array_equal(a, c)


def test_restriction_from_eager_prolongation(S, po, hierarchy, monkeypatch):
    """A big structured P is kept in row-pattern form only (no host copy, no CSR upload).  When the caller gives no R, R = P^T
    (GridTransferOperators.jl:536-547) is formed from that form: must equal the run that hands P^T over explicitly, and the
    run with the eager path switched off, to the last bit."""
    nc, nlev = (64, 64, 64), 3                                     # P_0: 250 047 rows >= the eager threshold
    H = hierarchy(nc, nlev)
    b = po.dirichlet_lift_rhs(nc, 1)
    res = []
    for eager, restr in (("1", None), ("1", H["restrictions"]), ("0", None)):
        monkeypatch.setenv("GMG_EAGER", eager)
        solver = S.CGSolver(S.GMGLinearSolver(H["mats"], H["prolongations"], restr, pre_smoothers=jac(S, nlev), post_smoothers=jac(S, nlev),
                                              maxiter=1), maxiter=30, atol=1e-14, rtol=1e-8)
        ns = setup(S, solver, H["mats"][0])
        x = np.zeros_like(b)
        S.solve_(x, ns, b)
        from gridapsolvers_jl_amd import abi
        v = np.random.default_rng(4).uniform(-1, 1, H["restrictions"][0].shape[1])
        y = np.zeros(H["restrictions"][0].shape[0])
        ns.P_ns.op_apply(0, abi.OP_R, v, y)
        res.append((x, y, solver.log.num_iters))
        ns.P_ns.close()
    for x, y, it in res[1:]:
        np.testing.assert_array_equal(x, res[0][0]); np.testing.assert_array_equal(y, res[0][1]); assert it == res[0][2]


def test_offset_patterns_and_eager_form_with_empty_and_ragged_rows(S, po, orc, monkeypatch):
    """Edge cases of the two structure-detecting layouts: (a) SELL-O on an operator with random values on a banded structure that has
    empty rows, short boundary rows and a ragged last slice; (b) the eager row-pattern scan (whole CSR handed to
    gmg_set_restriction / gmg_set_prolongation, >= 20 000 rows) on operators with empty rows.  Products must equal the oracle's
    left-to-right row sums bit for bit."""
    import scipy.sparse as sp
    from gridapsolvers_jl_amd import abi
    rng = np.random.default_rng(5)
    n = 64 * 400 + 37                                              # ragged last slice
    offs = np.array([-301, -300, -299, -1, 0, 1, 299, 300, 301])
    rows, cols = [], []
    for o in offs:
        i = np.arange(max(0, -o), min(n, n - o))
        rows.append(i); cols.append(i + o)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    keep = np.ones(rows.size, dtype=bool)
    dead = np.array([5, 6, 70, 1000, n - 1])                       # rows that keep only their diagonal ...
    keep &= ~(np.isin(rows, dead) & (rows != cols))
    A = sp.csr_matrix((rng.uniform(-1.0, 1.0, keep.sum()), (rows[keep], cols[keep])), shape=(n, n)).tocsr()
    A = (A + sp.diags(np.full(n, 20.0))).tocsr(); A.sort_indices()
    nH = n // 8
    # P: two entries per row, constant values, every 97th row EMPTY; R = P^T has ~16 entries per row and some empty rows as well
    pr = np.repeat(np.arange(n), 2); pc = np.minimum(np.stack([np.arange(n) // 8, np.arange(n) // 8 + 1], axis=1).reshape(-1), nH - 1)
    pk = (pr % 97) != 0
    pk &= ~((pc == 0) | (pc == 7))                                  # coarse columns 0 and 7 unused -> empty rows of R
    Pm = sp.csr_matrix((np.where(np.arange(pr.size) % 2 == 0, 0.75, 0.25)[pk], (pr[pk], pc[pk])), shape=(n, nH)).tocsr()
    Pm.sum_duplicates(); Pm.sort_indices()
    AH = (Pm.T @ A @ Pm + sp.diags(np.full(nH, 1.0))).tocsr(); AH.sort_indices()
    mk = lambda M: po.CSR(M.shape, M.indptr, M.indices, M.data)
    H = dict(mats=[mk(A), mk(AH)], prolongations=[mk(Pm)], restrictions=[mk(Pm.T.tocsr())])
    out = {}
    for name, env in {"default": {}, "plain": dict(GMG_OPATTERN="0", GMG_EAGER="0")}.items():
        for k in ("GMG_OPATTERN", "GMG_EAGER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ns = setup(S, make_gmg(S, H), H["mats"][0])
        if name == "default":
            assert ns.level_format(0)["layout"] == "SELL-O"
        ys = []
        for op, M in ((abi.OP_A, H["mats"][0]), (abi.OP_P, H["prolongations"][0]), (abi.OP_R, H["restrictions"][0])):
            v = np.random.default_rng(11).uniform(-1, 1, M.shape[1])
            y = np.full(M.shape[0], np.nan)
            ns.op_apply(0, op, v, y)
            assert np.array_equal(y, orc.spmv(M, v)), (name, op)
            ys.append(y)
        x, r = np.zeros(n), np.random.default_rng(12).uniform(-1, 1, n)
        ns.smooth(0, x, r)
        out[name] = ys + [x, r]
        ns.close()
    for a, b in zip(out["default"], out["plain"]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("coded", ["default", "coded"])
@pytest.mark.parametrize("nc,nlev", [((12, 12, 12), 2), ((24, 24), 3)])
def test_patch_operator_form_matches_patch_solves(S, po, orc, hierarchy, monkeypatch, nc, nlev, coded):
    """The additive-Schwarz operator sum_p R_p^T inv(A_pp) R_p in row-pattern form (default) against the patch-by-patch
    kernels (GMG_PATCH_OPERATOR=0) and the oracle: one application of the patch preconditioner (<= 1e-12), a V-cycle and the
    FGMRES solve (same iteration count, histories <= 1e-8).  Pre-summing the coefficients of a column over the patches changes
    the rounding order, not the operator."""
    order = 2
    if coded == "coded":
        monkeypatch.setenv("GMG_PAT_CODED_MIN_ROWS", "0")          # the coded shared-offset table (what the 10^8-dof levels use) for A and M
    H = hierarchy(nc, nlev, order)
    tabs = [po.vertex_star_patches(tuple(c // 2 ** l for c in nc), order) for l in range(nlev - 1)]
    b = po.dirichlet_lift_rhs(nc, order)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("GMG_PATCH_OPERATOR", flag)
        sm = [S.RichardsonSmoother(S.PatchSolver(pp, pd), 5, 0.2) for pp, pd in tabs]
        solver = S.FGMRESSolver(5, make_gmg(S, H, pre_smoothers=sm), maxiter=30, atol=1e-14, rtol=1e-8)
        ns = setup(S, solver, H["mats"][0])
        r = np.random.default_rng(8).uniform(-1, 1, H["mats"][0].shape[0])
        dx = np.zeros_like(r)
        ns.P_ns.precond(0, r, dx)
        z = np.zeros_like(r)
        S.solve_(z, ns.P_ns, r)
        x = np.zeros_like(b)
        S.solve_(x, ns, b)
        res[flag] = (dx, z, x, solver.log.num_iters, solver.log.residuals[: solver.log.num_iters + 1].copy())
        ns.P_ns.close()
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"],
                 pre_smoothers=[orc.Smoother(orc.PATCH, 5, 0.2, pp, pd) for pp, pd in tabs], maxiter=1)
    r = np.random.default_rng(8).uniform(-1, 1, H["mats"][0].shape[0])
    dxo = go.precond(0, r)
    for flag in ("1", "0"):
        assert max_rel(res[flag][0], dxo) <= 1e-12, flag
    assert max_rel(res["1"][0], res["0"][0]) <= 1e-12
    assert rel_err(res["1"][1], res["0"][1]) <= 1e-11
    assert res["1"][3] == res["0"][3]
    np.testing.assert_allclose(res["1"][4], res["0"][4], rtol=1e-8)
    assert rel_err(res["1"][2], res["0"][2]) <= 1e-9


def test_value_refresh_after_the_patch_operator_released_its_buffers(S, po, hierarchy, monkeypatch):
    """The row-pattern form of the patch operator releases the contribution buffer and the incidence lists.  A value refresh
    that cannot form it again (here: switched off between the two setups) must bring the patch-by-patch kernels back and equal a
    fresh setup made under the same switch, to the last bit."""
    nc, nlev, order = (12, 12, 12), 2, 2
    H = hierarchy(nc, nlev, order)
    pp, pd = po.vertex_star_patches(nc, order)
    mk = lambda: S.FGMRESSolver(5, make_gmg(S, H, pre_smoothers=[S.RichardsonSmoother(S.PatchSolver(pp, pd), 5, 0.2)]), maxiter=30, atol=1e-14, rtol=1e-8)
    b = po.dirichlet_lift_rhs(nc, order)
    monkeypatch.setenv("GMG_PATCH_OPERATOR", "1")
    monkeypatch.setenv("GMG_VDICT", "0"); monkeypatch.setenv("GMG_PATTERN", "0")     # explicit-value layout: the refresh path applies
    solver = mk()
    ns = setup(S, solver, H["mats"][0])
    x0 = np.zeros_like(b); S.solve_(x0, ns, b)
    monkeypatch.setenv("GMG_PATCH_OPERATOR", "0")
    A2 = po.CSR(H["mats"][0].shape, H["mats"][0].ptr, H["mats"][0].idx, 1.25 * H["mats"][0].val)
    mats2 = [A2] + list(H["mats"][1:])
    S.numerical_setup_(ns, A2, mats2)
    x = np.zeros_like(b); S.solve_(x, ns, b)
    H2 = dict(H, mats=mats2)
    s2 = S.FGMRESSolver(5, make_gmg(S, H2, pre_smoothers=[S.RichardsonSmoother(S.PatchSolver(pp, pd), 5, 0.2)]), maxiter=30, atol=1e-14, rtol=1e-8)
    ns2 = setup(S, s2, A2)
    xf = np.zeros_like(b); S.solve_(xf, ns2, b)
    np.testing.assert_array_equal(x, xf)
    assert solver.log.num_iters == s2.log.num_iters
    assert rel_err(x, x0 / 1.25) < 1e-6


def test_persistent_smoothing_pass_under_uneven_load(S, po, hierarchy, monkeypatch):
    """Inter-workgroup hand-offs must be tested with other work on the chip (stale-data and residency hazards hide on an idle GPU):
    V-cycles with the one-launch smoothing passes run while a second stream keeps the CUs busy with GEMMs of varying size;
    every result must equal the per-sweep reference bit for bit."""
    import torch
    nc, nlev = (64, 64, 64), 4
    H = hierarchy(nc, nlev)
    n = H["mats"][0].shape[0]
    rs = [np.random.default_rng(300 + k).uniform(-1, 1, n) for k in range(4)]
    monkeypatch.setenv("GMG_PERSIST", "0")
    ns0 = setup(S, make_gmg(S, H), H["mats"][0])
    ref = []
    for r in rs:
        z = np.zeros(n); S.solve_(z, ns0, r); ref.append(z)
    ns0.close()
    monkeypatch.setenv("GMG_PERSIST", "1")
    ns = setup(S, make_gmg(S, H), H["mats"][0])
    side = torch.cuda.Stream()
    a = torch.randn(3072, 3072, device="cuda"); b = torch.randn(3072, 3072, device="cuda")
    small = torch.randn(256, 256, device="cuda")
    rd = [torch.from_numpy(r).cuda() for r in rs]
    zd = torch.zeros(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for rep in range(60):
        with torch.cuda.stream(side):                               # uneven background load: big and tiny kernels alternate
            for k in range(3):
                (a @ b) if (rep + k) % 2 == 0 else (small @ small)
        S.solve_(zd, ns, rd[rep % 4])
        got = zd.cpu().numpy()
        np.testing.assert_array_equal(got, ref[rep % 4], err_msg=f"V-cycle {rep} under load")
    torch.cuda.synchronize()
    ns.close()


def test_offset_pattern_layout_large_level_paths(S, po, monkeypatch):
    """Levels above ~10^5 rows take the 4-waves-per-workgroup geometry and, above 128 MB of values, non-temporal loads; the
    small parity cases never reach those instantiations.  Variable-coefficient 96^3: SELL-O against SELL-64, bit for bit, on the
    operator, a smoothing pass and the CG solve; linearity and symmetry of the V-cycle."""
    import torch
    from gridapsolvers_jl_amd import abi
    nc, nlev = (96, 96, 96), 3
    H = po.build_hierarchy(nc, nlev, 1, kappa=po.smooth_kappa)
    A = H["mats"][0]
    n = A.shape[0]
    uex = po.nodal_values(nc, 1)
    b = A.matvec(uex)
    res = {}
    for layout, flag in (("SELL-64", "0"), ("SELL-O", "1")):
        monkeypatch.setenv("GMG_OPATTERN", flag)
        solver = S.CGSolver(make_gmg(S, H), maxiter=30, atol=1e-14, rtol=1e-8)
        ns = setup(S, solver, A)
        assert ns.P_ns.level_format(0)["layout"] == layout
        v = np.random.default_rng(1).uniform(-1, 1, n)
        y = np.zeros(n); ns.P_ns.op_apply(0, abi.OP_A, v, y)
        xs, rs = np.zeros(n), v.copy(); ns.P_ns.smooth(0, xs, rs)
        x = np.zeros(n); S.solve_(x, ns, b)
        res[layout] = (y, xs, rs, x, solver.log.num_iters)
        if layout == "SELL-O":
            assert np.max(np.abs(x - uex)) < 1e-6
            g = ns.P_ns
            r1, r2 = torch.from_numpy(np.random.default_rng(2).uniform(-1, 1, n)).cuda(), torch.from_numpy(np.random.default_rng(3).uniform(-1, 1, n)).cuda()
            z1, z2, z3 = torch.zeros_like(r1), torch.zeros_like(r1), torch.zeros_like(r1)
            torch.cuda.synchronize()
            S.solve_(z1, g, r1); S.solve_(z2, g, r2)
            r3 = (0.5 * r1 - r2).contiguous(); torch.cuda.synchronize()
            S.solve_(z3, g, r3)
            assert (torch.linalg.norm(z3 - (0.5 * z1 - z2)) / torch.linalg.norm(z3)).item() < 1e-12
            a12, a21 = torch.dot(z1, r2).item(), torch.dot(r1, z2).item()
            assert abs(a12 - a21) <= 1e-10 * max(abs(a12), abs(a21))
        ns.P_ns.close()
    for a, c in zip(res["SELL-64"][:4], res["SELL-O"][:4]):
        np.testing.assert_array_equal(a, c)
    assert res["SELL-64"][4] == res["SELL-O"][4]
    assert max_rel(res["SELL-O"][0], A.matvec(np.random.default_rng(1).uniform(-1, 1, n))) <= 1e-13
