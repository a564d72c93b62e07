"""CPU tests that pin the oracle (oracle/gmg_oracle.c).

The reference (GridapSolvers.jl, Julia) cannot run here and holds no golden
vectors for this path ("parity unpinned" for V-cycle vectors).  What it does hold
are analytic known-answer tests; the oracle is checked against every one of them
that touches the hot path, against an independent numpy/scipy restatement, and
against the committed fixtures."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from conftest import max_rel, rel_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- reference known-answer tests ------------------------------------------------
@pytest.mark.parametrize("nc", [(8, 8), (8, 8, 8)])
def test_reference_smoothers_test(po, orc, nc):
    """test/LinearSolvers/SmoothersTests.jl:13-43,58-74: CG(rtol=1e-8) preconditioned by
    LinearSolverFromSmoother(RichardsonSmoother(Jacobi,5,2/3)), u = x1+x2, @test E < 1e-8."""
    A = po.poisson_matrix(nc, 1)
    b = po.dirichlet_lift_rhs(nc, 1)
    x, nit, flag, hist = orc.cg_smoother_solve(A, b, 5, 2.0 / 3.0, maxiter=1000, atol=1e-12, rtol=1e-8)
    assert flag in (0, 1)
    assert po.l2_error_sq(nc, 1, x) < 1.0e-8


@pytest.mark.parametrize("nc", [(8, 8), (8, 8, 8)])
@pytest.mark.parametrize("Pl", ["jacobi", None])
def test_reference_richardson_linear_test(po, orc, nc, Pl):
    """test/LinearSolvers/RichardsonLinearTests.jl:14-26,64-73: RichardsonLinearSolver(0.5,1000;Pl=JacobiLinearSolver(),rtol=1e-8)
    and without preconditioner, u = x1+x2, @test E < 1e-6."""
    A = po.poisson_matrix(nc, 1)
    b = po.dirichlet_lift_rhs(nc, 1)
    x, nit, flag, hist = orc.richardson_solve(A, b, 0.5, Pl=Pl, maxiter=1000, rtol=1e-8)
    assert po.l2_error_sq(nc, 1, x) < 1.0e-6


@pytest.mark.parametrize("nc", [(8, 8), (8, 8, 8)])
@pytest.mark.parametrize("which", ["fgmres", "fgmres_restart", "cg", "pcg", "fpcg"])
def test_reference_krylov_tests(po, orc, nc, which):
    """test/LinearSolvers/KrylovTests.jl:12-25,77-90: P = JacobiLinearSolver(), rtol 1e-8, @test E < 1e-6."""
    A = po.poisson_matrix(nc, 1)
    b = po.dirichlet_lift_rhs(nc, 1)
    if which == "fgmres":
        x, *_ = orc.fgmres_solve(A, b, Pr="jacobi", m=10, rtol=1e-8)
    elif which == "fgmres_restart":
        x, *_ = orc.fgmres_solve(A, b, Pr="jacobi", m=10, restart=True, rtol=1e-8)
    elif which == "cg":
        x, *_ = orc.cg_solve(A, b, rtol=1e-8)
    elif which == "pcg":
        x, *_ = orc.cg_solve(A, b, Pl="jacobi", rtol=1e-8)
    else:
        x, *_ = orc.cg_solve(A, b, Pl="jacobi", flexible=True, rtol=1e-8)
    assert po.l2_error_sq(nc, 1, x) < 1.0e-6


@pytest.mark.parametrize("nc,nlev", [((16, 16), 3), ((8, 8, 4), 2), ((64, 64), 3), ((32, 32, 32), 3)])
def test_reference_gmg_tests_configuration(po, orc, hierarchy, nc, nlev):
    """test/LinearSolvers/GMGTests.jl:52,109-124,204-213: CG(maxiter=20,atol=1e-14,rtol=1e-6) +
    GMG(maxiter=1,:preconditioner,:v_cycle, Richardson(Jacobi,10,2/3)); the test only prints the
    L2 error -- here it must be tiny and CG must converge by rtol."""
    H = hierarchy(nc, nlev)
    b = po.dirichlet_lift_rhs(nc, 1)
    g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    x, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=g, maxiter=20, atol=1e-14, rtol=1e-6)
    assert flag == 1 and nit <= 5
    assert po.l2_error_sq(nc, 1, x) < 1.0e-10
    # survey-time expectation: 3 iterations on the larger meshes
    if nc in [(64, 64), (32, 32, 32)]:
        assert nit == 3


def test_baseline_md_histories(po, orc, hierarchy):
    """BASELINE.md section 2 relative residual histories of the independent survey-time restatement."""
    expect = {((64, 64), 3): [1, 6.8e-3, 2.8e-5, 1.6e-7], ((32, 32, 32), 3): [1, 5.1e-3, 1.7e-5, 5.5e-8]}
    for (nc, nlev), h in expect.items():
        H = hierarchy(nc, nlev)
        g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
        _, nit, _, hist = orc.cg_solve(H["mats"][0], po.dirichlet_lift_rhs(nc, 1), Pl=g, maxiter=20, atol=1e-14, rtol=1e-6)
        assert nit == 3
        np.testing.assert_allclose(hist / hist[0], h, rtol=0.05)


# ---- independent numpy/scipy twin --------------------------------------------------
def _np_vcycle(mats, Ps, lev, x, r, niter=10, omega=2.0 / 3.0):
    A = mats[lev].to_scipy()
    if lev == len(mats) - 1:
        x[:] = spla.spsolve(A.tocsc(), r)
        return
    dinv = 1.0 / A.diagonal()

    def smooth():
        for _ in range(niter):
            dx = omega * (dinv * r)
            x[:] = x + dx
            r[:] = r - A @ dx
    P = Ps[lev].to_scipy()
    smooth()
    rH = P.T @ r
    dxH = np.zeros(P.shape[1])
    _np_vcycle(mats, Ps, lev + 1, dxH, rH, niter, omega)
    dx = P @ dxH
    x[:] = x + dx
    r[:] = r - A @ dx
    smooth()


@pytest.mark.parametrize("nc,nlev", [((32, 32), 3), ((8, 8, 8), 2)])
def test_oracle_vs_numpy_twin(po, orc, hierarchy, nc, nlev):
    H = hierarchy(nc, nlev)
    n = H["mats"][0].shape[0]
    r0 = np.random.default_rng(3).uniform(-1, 1, n)
    g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    z, nit, _, hist = g.solve(r0)
    z2 = np.zeros(n); r2 = r0.copy()
    _np_vcycle(H["mats"], H["prolongations"], 0, z2, r2)
    assert nit == 1
    assert rel_err(z, z2) < 1e-12
    assert abs(hist[1] - np.linalg.norm(r2)) / hist[1] < 1e-10


# ---- operator identities (SURVEY 8c fixture 3) ---------------------------------------
@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("nc", [(4, 4), (4, 2, 2)])
def test_operator_identities(po, nc, order):
    fine = tuple(2 * c for c in nc)
    P = po.prolongation(nc, order)
    Ah, AH = po.poisson_matrix(fine, order), po.poisson_matrix(nc, order)
    Ps, As, AHs = P.to_scipy(), Ah.to_scipy(), AH.to_scipy()
    assert abs((Ps.T @ As @ Ps - AHs).toarray()).max() < 1e-13          # A_H == P^T A_h P
    R = P.transpose().to_scipy()
    assert abs((R - Ps.T).toarray()).max() == 0.0                        # R == P^T
    assert abs(As - As.T).max() < 1e-14                                  # symmetric
    # constants are reproduced by P away from the Dirichlet boundary: interior rows sum to 1
    rows = np.asarray(Ps.sum(axis=1)).ravel()
    assert np.isclose(rows, 1.0).any()
    if order == 1:
        assert np.isclose(rows.max(), 1.0)
    # u = x1+x2 is in the FE space: A u_free = b (rhs is the Dirichlet lift)
    assert max_rel(Ah.matvec(po.nodal_values(fine, order)), po.dirichlet_lift_rhs(fine, order)) < 1e-12


def test_structural_pattern_sizes(po):
    """SURVEY 8 size table: N=(n-1)^d, Z=(3(n-1)-2)^d, Z_P=(3 n_c-3)^d for Q1; Q2: N=(2n-1)^d, Z=(8n-9)^d."""
    A = po.poisson_matrix((16, 16, 16), 1)
    assert A.shape[0] == 15 ** 3 and A.nnz == (3 * 15 - 2) ** 3
    P = po.prolongation((8, 8, 8), 1)
    assert P.nnz == (3 * 8 - 3) ** 3
    A2 = po.poisson_matrix((4, 4, 4), 2)
    assert A2.shape[0] == 7 ** 3 and A2.nnz == (8 * 4 - 9) ** 3
    A1 = po.poisson_matrix((64, 64), 1)
    assert (A1.shape[0], A1.nnz) == (3969, 34969)


# ---- smoothers / patches ----------------------------------------------------------------
def test_q1_patch_smoother_is_jacobi(po, orc, hierarchy):
    """SURVEY 8c fixture 4: for Q1 a vertex-star patch holds the vertex dof only => PatchSolver == Jacobi."""
    nc = (8, 8, 8)
    H = hierarchy(nc, 2)
    pp, pd = po.vertex_star_patches(nc, 1)
    r = np.random.default_rng(5).uniform(-1, 1, H["mats"][0].shape[0])
    gj = orc.GMG(H["mats"], H["prolongations"], maxiter=1)
    for kind in (orc.PATCH, orc.BLOCKJACOBI):
        gp = orc.GMG(H["mats"], H["prolongations"], pre_smoothers=[orc.Smoother(kind, 10, 2.0 / 3.0, pp, pd)], maxiter=1)
        assert max_rel(gp.precond(0, r), gj.precond(0, r)) < 1e-15
        assert rel_err(gp.solve(r)[0], gj.solve(r)[0]) < 1e-13


def test_q2_patch_solve_vs_numpy(po, orc, hierarchy):
    nc, order = (8, 8), 2
    H = hierarchy(nc, 2, order)
    A = H["mats"][0].to_scipy().toarray()
    pp, pd = po.vertex_star_patches(nc, order)
    r = np.random.default_rng(6).uniform(-1, 1, A.shape[0])
    ref = np.zeros_like(r)
    for p in range(len(pp) - 1):
        d = pd[pp[p]:pp[p + 1]]
        if d.size:
            ref[d] += np.linalg.solve(A[np.ix_(d, d)], r[d])
    assert max((pp[1:] - pp[:-1])) == 9                     # interior vertex: 3^2 dofs
    for kind in (orc.PATCH, orc.BLOCKJACOBI):
        g = orc.GMG(H["mats"], H["prolongations"], pre_smoothers=[orc.Smoother(kind, 10, 0.2, pp, pd)], maxiter=1)
        assert max_rel(g.precond(0, r), ref) < 1e-13


def test_direct_solver_and_givens(po, orc):
    A = po.poisson_matrix((8, 8, 8), 1)
    b = np.random.default_rng(1).uniform(-1, 1, A.shape[0])
    x = orc.direct_solve(A, b)
    assert max_rel(A.matvec(x), b) < 1e-12
    for f, g in [(3.0, 4.0), (-3.0, 4.0), (5.0, -1.0), (0.0, 2.0), (2.0, 0.0), (1e-300, 1e-300), (1e200, -1e200)]:
        c, s, r = orc.givens(f, g)
        assert abs(c * c + s * s - 1.0) < 1e-14
        assert abs(-s * f + c * g) <= 1e-14 * max(abs(f), abs(g), 1e-300) * 4
        assert np.isclose(c * f + s * g, r, rtol=1e-14)


def test_stopping_rule_semantics(po, orc, hierarchy):
    """SolverTolerances.jl:117-128: strict '<'; maxiter caps; init! evaluates with e_r = 1."""
    H = hierarchy((16, 16), 3)
    A = H["mats"][0]
    b = po.dirichlet_lift_rhs((16, 16), 1)
    x, nit, flag, hist = orc.cg_solve(A, b, maxiter=2, rtol=1e-30, atol=0.0)
    assert nit == 2 and flag == 2 and hist.size == 3                # SOLVER_DIVERGED_MAXITER
    x, nit, flag, hist = orc.cg_solve(A, b, maxiter=50, rtol=2.0, atol=0.0)
    assert nit == 0 and flag == 1                                    # e_r = 1 < rtol at init
    x, nit, flag, hist = orc.cg_solve(A, np.zeros_like(b), maxiter=50, rtol=1e-6, atol=1e-12)
    assert nit == 0 and flag == 0 and np.all(x == 0)                 # zero rhs: atol at init


# ---- committed fixtures ------------------------------------------------------------------
def test_golden_config1(po, orc, hierarchy):
    gold = np.load(os.path.join(GOLD, "config1_q1_64x64.npz"))
    nc, nlev = (64, 64), 3
    H = hierarchy(nc, nlev)
    b = po.dirichlet_lift_rhs(nc, 1)
    g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    x, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=g, maxiter=20, atol=1e-14, rtol=1e-6)
    assert nit == int(gold["niters"]) == 3
    np.testing.assert_allclose(hist, gold["hist"], rtol=1e-12)
    assert rel_err(x, gold["x"]) < 1e-13
    assert po.l2_error_sq(nc, 1, gold["x"]) < 1e-8                   # reference criterion
    xf, nitf, _, histf = orc.fgmres_solve(H["mats"][0], b, Pr=g, m=5, maxiter=20, atol=1e-14, rtol=1e-6)
    assert nitf == int(gold["fgmres_niters"])
    np.testing.assert_allclose(histf, gold["fgmres_hist"], rtol=1e-10)


def test_w_and_f_cycles_converge_faster_than_v(po, orc, hierarchy):
    H = hierarchy((16, 16, 16), 3)
    r = np.random.default_rng(2).uniform(-1, 1, H["mats"][0].shape[0])
    out = {}
    for name, cyc in (("v", orc.V_CYCLE), ("w", orc.W_CYCLE), ("f", orc.F_CYCLE)):
        g = orc.GMG(H["mats"], H["prolongations"], cycle=cyc, maxiter=1)
        out[name] = g.solve(r)[3][1]
    assert out["w"] <= out["f"] <= out["v"]


@pytest.mark.parametrize("nc,order", [((4, 4), 2), ((2, 2, 2), 2), ((4, 4), 1)])
def test_patch_corrected_prolongation(po, orc, nc, order):
    """PatchProlongationOperator (PatchTransferOperators.jl:153-172): y = P x - sum_p A_pp^-1 (A P x)_p over the
    interiors of the coarse cells.  Checked against the formula in numpy, and through its defining property:
    the corrected prolongation is discretely harmonic inside every coarse cell, (A y)_p = 0."""
    H = po.build_hierarchy(tuple(2 * c for c in nc), 2, order)
    A, P = H["mats"][0].to_scipy(), H["prolongations"][0].to_scipy()
    pp, pd = po.coarse_cell_interior_patches(nc, order)
    assert len(pp) - 1 == int(np.prod(nc)) and (pp[1:] - pp[:-1]).max() == (2 * order - 1) ** len(nc)
    xH = np.random.default_rng(4).uniform(-1, 1, P.shape[1])
    y = P @ xH
    t = A @ y
    Ad = A.toarray()
    for p in range(len(pp) - 1):
        d = pd[pp[p]:pp[p + 1]]
        y[d] -= np.linalg.solve(Ad[np.ix_(d, d)], t[d])
    assert np.abs((A @ y)[pd]).max() < 1e-12 * np.abs(t).max()            # harmonic inside the coarse cells
    # the oracle applies it inside the cycle: compare one two-level V-cycle with / without niter (smoothers off)
    for kind in (orc.PATCH, orc.BLOCKJACOBI):
        g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.JACOBI, 0, 1.0)],
                    maxiter=1, prolongation_patches=[(kind, pp, pd)])
        r = np.random.default_rng(5).uniform(-1, 1, A.shape[0])
        z = g.solve(r)[0]
        xc = spla.spsolve(H["mats"][1].to_scipy().tocsc(), P.T @ r)       # no smoothing: z = Ptilde A_H^-1 R r
        yy = P @ xc
        tt = A @ yy
        for p in range(len(pp) - 1):
            d = pd[pp[p]:pp[p + 1]]
            yy[d] -= np.linalg.solve(Ad[np.ix_(d, d)], tt[d])
        assert rel_err(z, yy) < 1e-12


# ---------------------------------------------------------------- block preconditioners
def _csr(po, M):
    M = M.tocsr(); M.sort_indices()
    return po.CSR(M.shape, M.indptr, M.indices, M.data)


def test_block_triangular_lu_blocks_reference_test_problem(po, orc):
    """test/BlockSolvers/BlockTriangularSolversTests.jl:22-45: a = u1 v1 + u2 v2 + u1 v2 - u2 v1, i.e. the block
    matrix [[M, -M], [M, M]] with LU block solvers, :upper and :lower.  The reference asserts nothing; here the
    block back/forward substitution identities are checked (BlockTriangularSolvers.jl:186-242), and
    BlockDiagonalSolversTests.jl:38-45 (block-diagonal system, LU blocks => exact solve, norm(x1-x) < 1e-8)."""
    M = po.poisson_matrix((8, 8), 1)
    Ms = M.to_scipy()
    n = M.shape[0]
    negM = _csr(po, -Ms)
    b = np.random.default_rng(0).uniform(-1, 1, 2 * n)
    lu = spla.splu(Ms.tocsc())
    up = orc.BlockPreconditioner([n, n], [(orc.BD_LU, M), (orc.BD_LU, M)], {(0, 1): (negM, 1.0), (1, 0): (M, 1.0)}, orc.UPPER)
    x = up.apply(b)
    x2 = lu.solve(b[n:]); x1 = lu.solve(b[:n] + Ms @ x2)
    assert rel_err(x, np.concatenate([x1, x2])) < 1e-12
    lo = orc.BlockPreconditioner([n, n], [(orc.BD_LU, M), (orc.BD_LU, M)], {(0, 1): (negM, 1.0), (1, 0): (M, 1.0)}, orc.LOWER)
    x = lo.apply(b)
    y1 = lu.solve(b[:n]); y2 = lu.solve(b[n:] - Ms @ y1)
    assert rel_err(x, np.concatenate([y1, y2])) < 1e-12
    dg = orc.BlockPreconditioner([n, n], [(orc.BD_LU, M), (orc.BD_LU, M)], None, orc.DIAGONAL)
    x = dg.apply(b)
    assert np.linalg.norm(x - np.concatenate([lu.solve(b[:n]), lu.solve(b[n:])])) < 1e-8


def test_block_triangular_coefficients_and_stateful_cg_block(po, orc):
    """coeffs[i,j] scale the off-diagonal contribution and a zero coefficient drops it (:194-197); a CG block solver
    starts from the previous application's result (the work cache y persists, :202-205)."""
    A = po.poisson_matrix((8, 8), 1); n = A.shape[0]
    As = A.to_scipy()
    rng = np.random.default_rng(1)
    C12 = _csr(po, sp.random(n, n, density=0.05, random_state=3, format="csr"))
    b = rng.uniform(-1, 1, 2 * n)
    lu = spla.splu(As.tocsc())
    for c in (0.0, 1.0, -2.5):
        P = orc.BlockPreconditioner([n, n], [(orc.BD_LU, A), (orc.BD_JACOBI, A)], {(0, 1): (C12, c)}, orc.UPPER)
        x = P.apply(b)
        x2 = b[n:] / As.diagonal()
        x1 = lu.solve(b[:n] - c * (C12.to_scipy() @ x2))
        assert rel_err(x, np.concatenate([x1, x2])) < 1e-12
    P = orc.BlockPreconditioner([n], [(orc.BD_CG_JACOBI, A, 3, 1e-30, 1e-30)], None, orc.DIAGONAL)
    xa = P.apply(b[:n]); xb = P.apply(b[:n])          # second call continues from xa: 3 + 3 CG iterations (restarted)
    ref3 = orc.cg_solve(A, b[:n], Pl="jacobi", maxiter=3, atol=1e-30, rtol=1e-30)[0]
    ref33 = orc.cg_solve(A, b[:n], Pl="jacobi", x0=ref3, maxiter=3, atol=1e-30, rtol=1e-30)[0]
    assert np.array_equal(xa, ref3) and np.array_equal(xb, ref33)


def test_fgmres_block_triangular_gmg_stokes_like(po, orc, hierarchy):
    """The StokesGMG.jl:142-153 solver shape on a synthetic saddle-point system: FGMRES(20) right-preconditioned by
    BlockTriangularSolver([GMG(maxiter=4), CG-Jacobi(maxiter=20, rtol=1e-6)], coeffs=[1 1;0 1], :upper) with the
    pressure block solver set up on -1/alpha * Mp.  Converges and solves the system."""
    H = hierarchy((8, 8, 8), 2)
    A = H["mats"][0]; n1 = A.shape[0]
    R = H["restrictions"][0]; n2 = R.shape[0]
    alpha = 10.0
    B = _csr(po, 0.5 * R.to_scipy())
    Bt = _csr(po, 0.5 * R.to_scipy().T)
    Mp = _csr(po, (-1.0 / alpha) * (sp.identity(n2) + 0.1 * H["mats"][1].to_scipy()))
    K = _csr(po, sp.bmat([[A.to_scipy(), Bt.to_scipy()], [B.to_scipy(), None]]))
    g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=4, rtol=1e-8)
    P = orc.BlockPreconditioner([n1, n2], [g, (orc.BD_CG_JACOBI, Mp, 20, 1e-14, 1e-6)], {(0, 1): (Bt, 1.0), (1, 0): (B, 0.0)}, orc.UPPER)
    b = np.random.default_rng(5).uniform(-1, 1, n1 + n2)
    x, nit, flag, hist = orc.fgmres_solve(K, b, Pr=P, m=20, maxiter=100, atol=1e-10, rtol=1e-12)
    assert flag in (0, 1) and nit < 100
    assert np.linalg.norm(K.to_scipy() @ x - b) < 1e-7      # StokesGMG.jl:166 @test norm(r) < 1.e-7


# ---------------------------------------------------------------- round 2: variable-coefficient inputs, cpu_baseline child
def test_varcoef_generator_reduces_to_constant_and_is_spd(po):
    import scipy.sparse.linalg as sl
    for nc in [(8, 8, 8), (6, 10, 4), (12, 8)]:
        A = po.poisson_matrix(nc, 1)
        B = po.poisson_matrix_varcoef(nc, lambda X, Y, Z: np.ones_like(X))
        assert (A.ptr == B.ptr).all() and (A.idx == B.idx).all()
        assert np.abs(A.val - B.val).max() <= 4e-16 * np.abs(A.val).max()
    V = po.poisson_matrix_varcoef((12, 12, 12))
    Sv = V.to_scipy()
    assert abs(Sv - Sv.T).max() == 0.0
    assert sl.eigsh(Sv, k=1, which="SA")[0][0] > 0
    # (almost) every row distinct: what forces the generic storage layout
    assert len(np.unique(V.val)) > 0.1 * V.nnz


def test_varcoef_cg_gmg_oracle_converges_like_constant(po, orc):
    nc, nlev = (16, 16, 16), 3
    H = po.build_hierarchy(nc, nlev, 1, kappa=po.smooth_kappa)
    uex = po.nodal_values(nc, 1)
    b = H["mats"][0].matvec(uex)
    g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    x, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=g, maxiter=30, atol=1e-14, rtol=1e-8)
    assert flag == 1 and nit <= 8
    assert np.max(np.abs(x - uex)) < 1e-6


def test_cpu_baseline_child_runs_both_variants():
    """bench.py's cpu_baseline leg: the child process times the sequential checker and the OpenMP build and agrees on iterations."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for variant, thr in (("seq", 1), ("omp", 2)):
        env = dict(os.environ, OMP_NUM_THREADS=str(thr), OMP_PROC_BIND="close")
        p = subprocess.run([sys.executable, os.path.join(root, "oracle", "cpu_baseline.py"), "--cells", "16", "--levels", "3",
                            "--variant", variant, "--limit-s", "0", "--max-reps", "1"], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-500:]
        outs[variant] = json.loads(p.stdout.strip().splitlines()[-1])
    assert outs["seq"]["threads"] == 1 and outs["omp"]["threads"] == 2
    assert outs["seq"]["iters"] == outs["omp"]["iters"] == 3
    np.testing.assert_allclose(outs["seq"]["hist"], outs["omp"]["hist"], rtol=1e-9)


# ---------------------------------------------------------------- round 2: independent twins of every oracle component
def _twin():
    import numpy_twin
    return numpy_twin


def _seed(n, s):
    return np.random.default_rng(s).uniform(-1, 1, n)


@pytest.mark.parametrize("cyc", ["v", "w", "f"])
@pytest.mark.parametrize("nc,nlev", [((16, 16), 3), ((8, 8, 8), 2)])
def test_twin_cycles_jacobi(po, orc, cyc, nc, nlev):
    """V/W/F cycles (GMGLinearSolvers.jl:468-610): oracle == independent numpy twin."""
    T = _twin()
    H = po.build_hierarchy(nc, nlev, 1)
    A = [m.to_scipy().tocsr() for m in H["mats"]]
    tw = T.GMG(H["mats"], H["prolongations"], H["restrictions"], [(T.Jacobi(A[l]), 10, 2.0 / 3.0) for l in range(nlev - 1)], cycle=cyc)
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], cycle={"v": orc.V_CYCLE, "w": orc.W_CYCLE, "f": orc.F_CYCLE}[cyc], maxiter=1)
    r = _seed(A[0].shape[0], 3)
    zo, _, _, _ = go.solve(r)
    zt = tw.solve(r)
    assert np.linalg.norm(zo - zt) <= 1e-12 * np.linalg.norm(zt)


@pytest.mark.parametrize("kind", ["patch", "block"])
def test_twin_patch_and_block_jacobi(po, orc, kind):
    """PatchSolvers.jl:279-300 / BlockJacobiSolvers.jl:141-170 incl. caller matrices and rows != cols (LAPACK getrf/getrs twin)."""
    T = _twin()
    nc, order = (8, 8), 2
    A = po.poisson_matrix(nc, order)
    pp, pd = po.vertex_star_patches(nc, order)
    r = _seed(A.shape[0], 5)
    H = po.build_hierarchy(nc, 2, order)
    okind = orc.PATCH if kind == "patch" else orc.BLOCKJACOBI
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(okind, 3, 0.2, pp, pd)], maxiter=1)
    tw = T.Patch(A.to_scipy(), pp, pd, pivot=(kind == "patch"))
    assert np.max(np.abs(go.precond(0, r) - tw.solve(r))) <= 1e-12 * np.max(np.abs(r)) * 10
    # whole smoothing pass
    x, rr = np.zeros_like(r), r.copy()
    T.richardson(A.to_scipy().tocsr(), tw, 3, 0.2, x, rr)
    xo, ro = go.smooth(0, np.zeros_like(r), r)
    assert np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(x) and np.linalg.norm(rr - ro) <= 1e-12 * np.linalg.norm(r)
    if kind == "patch":
        pc = pd.copy()
        for p in range(pp.size - 1):
            pc[pp[p]:pp[p + 1]] = pd[pp[p]:pp[p + 1]][::-1]
        go2 = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(okind, 3, 0.2, pp, pd, patch_cols=pc)], maxiter=1)
        tw2 = T.Patch(A.to_scipy(), pp, pd, cols=pc)
        assert np.max(np.abs(go2.precond(0, r) - tw2.solve(r))) <= 1e-11


@pytest.mark.parametrize("m,restart,m_add", [(5, False, 1), (2, True, 1), (2, False, 2)])
def test_twin_fgmres_restart_and_growth(po, orc, m, restart, m_add):
    """FGMRESSolvers.jl:130-199 with a GMG right preconditioner: iteration count and history of the oracle == twin
    (Givens rotations from LAPACK dlartg, triangular solve from LAPACK)."""
    T = _twin()
    nc, nlev = (16, 16), 2
    H = po.build_hierarchy(nc, nlev, 1)
    A = [mm.to_scipy().tocsr() for mm in H["mats"]]
    b = po.dirichlet_lift_rhs(nc, 1)
    tw = T.GMG(H["mats"], H["prolongations"], H["restrictions"], [(T.Jacobi(A[0]), 1, 0.5)])
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.JACOBI, 1, 0.5)], maxiter=1)
    xo, nit, flag, hist = orc.fgmres_solve(H["mats"][0], b, Pr=go, m=m, restart=restart, m_add=m_add, maxiter=40, atol=1e-14, rtol=1e-10)
    xt, nt, ht = T.fgmres(A[0], b, Pr=tw.solve, m=m, restart=restart, m_add=m_add, maxiter=40, atol=1e-14, rtol=1e-10)
    assert nit == nt and nit > m
    np.testing.assert_allclose(hist, ht, rtol=1e-7, atol=1e-15 * hist[0])      # the Givens estimate |g_j| carries eps*|r_0| absolute noise
    assert np.linalg.norm(xo - xt) <= 1e-9 * np.linalg.norm(xt)


@pytest.mark.parametrize("kind", ["upper", "lower", "diagonal"])
def test_twin_block_triangular_apply(po, orc, kind):
    """BlockTriangularSolvers.jl:186-242 / BlockDiagonalSolvers.jl:165-177 with LU diagonal blocks and a coefficient."""
    import scipy.sparse.linalg as spla
    T = _twin()
    M = po.poisson_matrix((8, 8), 1); n = M.shape[0]
    Ms = M.to_scipy().tocsr()
    negM = po.CSR(M.shape, M.ptr, M.idx, -M.val)
    lu = spla.splu(Ms.tocsc())
    okind = {"upper": orc.UPPER, "lower": orc.LOWER, "diagonal": orc.DIAGONAL}[kind]
    offd = None if kind == "diagonal" else {(0, 1): (negM, 0.5), (1, 0): (M, 2.0)}
    P = orc.BlockPreconditioner([n, n], [(orc.BD_LU, M), (orc.BD_LU, M)], offd, okind)
    b = _seed(2 * n, 9)
    xt = T.block_apply(kind, [lu.solve, lu.solve], {(0, 1): -Ms, (1, 0): Ms}, [[1.0, 0.5], [2.0, 1.0]], [n, n], b)
    assert np.linalg.norm(P.apply(b) - xt) <= 1e-12 * np.linalg.norm(xt)


# ---------------------------------------------------------------- order of summation: left-to-right checker vs BLAS-style dot / nrm2
@pytest.mark.parametrize("case", ["config1", "config2_32", "q2_patch"])
def test_iteration_counts_do_not_depend_on_summation_order(po, orc, case):
    """Julia's dot / norm on Vector{Float64} run through BLAS (interleaved partial sums, scaled nrm2), its CSC mul! scatters by
    column.  (1) CSC column scatter adds a row's terms in ascending column order -- the very order of the CSR gather, so
    mul! is bit-identical (checked below).  (2) With BLAS-style dot / nrm2 (oracle/liboracle_blas.so) the Krylov iteration
    counts are identical and the residual histories agree to 1e-8: the parity tolerances do not hinge on summation order."""
    if case == "config1":
        nc, nlev, order, kry = (64, 64), 3, 1, "cg"
    elif case == "config2_32":
        nc, nlev, order, kry = (32, 32, 32), 3, 1, "cg"
    else:
        nc, nlev, order, kry = (16, 16), 2, 2, "fgmres"
    H = po.build_hierarchy(nc, nlev, order)
    b = po.dirichlet_lift_rhs(nc, order)
    # (1) CSC scatter == CSR gather, bitwise
    A = H["mats"][0]
    S = A.to_scipy().tocsc(); S.sort_indices()
    x = _seed(A.shape[0], 1)
    y = np.zeros(A.shape[0])
    for j in range(min(S.shape[1], 400)):                   # mul!(y, ::SparseMatrixCSC, x): y[rowval[k]] += nzval[k]*x[j]
        for k in range(S.indptr[j], S.indptr[j + 1]):
            y[S.indices[k]] += S.data[k] * x[j]
    xx = np.zeros_like(x); xx[:min(S.shape[1], 400)] = x[:min(S.shape[1], 400)]
    np.testing.assert_array_equal(y, orc.spmv(A, xx))
    out = {}
    try:
        for variant in ("seq", "blas"):
            orc.set_variant(variant)
            if order == 2:
                pp, pd = po.vertex_star_patches(nc, order)
                g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.PATCH, 10, 0.2, pp, pd)], maxiter=1)
            else:
                g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
            if kry == "cg":
                out[variant] = orc.cg_solve(A, b, Pl=g, maxiter=20, atol=1e-14, rtol=1e-6)
            else:
                out[variant] = orc.fgmres_solve(A, b, Pr=g, m=5, maxiter=20, atol=1e-14, rtol=1e-6)
    finally:
        orc.set_variant("seq")
    (x1, n1, f1, h1), (x2, n2, f2, h2) = out["seq"], out["blas"]
    assert n1 == n2 and f1 == f2
    np.testing.assert_allclose(h2, h1, rtol=1e-8)
    assert np.linalg.norm(x1 - x2) <= 1e-10 * np.linalg.norm(x1)
    assert not np.array_equal(h1, h2) or n1 == 0             # the two builds really sum differently


def test_twin_fgmres_left_preconditioner(po, orc):
    """Pl != nothing (KrylovUtils.jl:14-18,46-50): oracle vs a scipy-free restatement through the numpy twin on the
    explicitly left-preconditioned system  (D^-1 A) x = D^-1 b."""
    import scipy.sparse as sp
    T = _twin()
    nc = (12, 12)
    A = po.poisson_matrix(nc, 1); As = A.to_scipy().tocsr()
    b = _seed(A.shape[0], 2)
    dinv = 1.0 / As.diagonal()
    xo, nit, flag, hist = orc.fgmres_solve(A, b, Pr=None, Pl="jacobi", m=4, restart=True, maxiter=60, atol=1e-14, rtol=1e-10)
    xt, nt, ht = T.fgmres(sp.diags(dinv) @ As, dinv * b, Pr=None, m=4, restart=True, maxiter=60, atol=1e-14, rtol=1e-10)
    assert nit == nt
    np.testing.assert_allclose(hist, ht, rtol=1e-6, atol=1e-15 * hist[0])
    assert np.linalg.norm(xo - xt) <= 1e-8 * np.linalg.norm(xt)


# ---------------------------------------------------------------- the reference's restriction is P^T only to ~1e-6 (CG mass solve)
@pytest.mark.parametrize("nc,nlev,order,krylov", [((64, 64), 3, 1, "cg"), ((16, 16, 16), 3, 1, "cg"), ((8, 8, 8), 2, 2, "fgmres")])
def test_inexact_restriction_does_not_change_iteration_counts(po, orc, nc, nlev, order, krylov):
    """test/LinearSolvers/GMGTests.jl:66-74 builds `restrict` with CGSolver(JacobiLinearSolver()) (rtol 1e-6) as the mass solver of
    the dual projection (GridTransferOperators.jl:536-547): the reference's R is P^T (1 + O(1e-6)).  With R perturbed by that
    much the oracle's iteration counts are unchanged and its residual histories move by <= 1e-5 relative (3 eps, linear)."""
    H = po.build_hierarchy(nc, nlev, order)
    b = po.dirichlet_lift_rhs(nc, order)
    out = []
    for eps in (0.0, 1e-6):
        rng = np.random.default_rng(11)
        Rs = [po.CSR(R.shape, R.ptr, R.idx, R.val * (1.0 + eps * rng.uniform(-1, 1, R.val.size))) for R in H["restrictions"]]
        if krylov == "cg":
            go = orc.GMG(H["mats"], H["prolongations"], Rs, maxiter=1)
            x, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=go, maxiter=20, atol=1e-14, rtol=1e-6)
        else:
            tabs = [po.vertex_star_patches(c, order) for c in H["ncells"][:-1]]
            go = orc.GMG(H["mats"], H["prolongations"], Rs, pre_smoothers=[orc.Smoother(orc.PATCH, 10, 0.2, pp, pd) for pp, pd in tabs], maxiter=1)
            x, nit, flag, hist = orc.fgmres_solve(H["mats"][0], b, Pr=go, m=5, maxiter=20, atol=1e-14, rtol=1e-6)
        out.append((nit, hist, x))
    assert out[0][0] == out[1][0]
    assert np.max(np.abs(out[1][1] - out[0][1]) / out[0][1]) <= 1e-5
    assert np.linalg.norm(out[1][2] - out[0][2]) <= 1e-10 * np.linalg.norm(out[0][2])
