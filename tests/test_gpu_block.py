"""GPU parity tests of the block preconditioner glue: BlockDiagonalSolver /
BlockTriangularSolver and the outer Krylov solve on a block system, device path through the C ABI against
the CPU oracle on the same seeded inputs.

Tolerances (fp64): one preconditioner application <= 1e-11 relative 2-norm; outer residual histories
<= 1e-6 relative per entry (inner iterative block solves amplify rounding); iteration counts identical."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL_APPLY, TOL_HIST = 1e-11, 1e-6


def _csr(po, M):
    M = M.tocsr(); M.sort_indices()
    return po.CSR(M.shape, M.indptr, M.indices, M.data)


def jac(S, nlev, niter=10, omega=2.0 / 3.0):
    return [S.RichardsonSmoother(S.JacobiLinearSolver(), niter, omega)] * (nlev - 1)


def setup(S, solver, A):
    return S.numerical_setup(S.symbolic_setup(solver, A), A)


@pytest.mark.parametrize("half", ["upper", "lower", "diagonal"])
def test_block_lu_blocks_reference_test_problem(S, po, orc, half):
    """BlockTriangularSolversTests.jl:22-45 / BlockDiagonalSolversTests.jl:38-45: [[M,-M],[M,M]] with LU blocks."""
    M = po.poisson_matrix((8, 8), 1); n = M.shape[0]
    negM = _csr(po, -M.to_scipy())
    mat = [[M, negM], [M, M]] if half != "diagonal" else [[M, None], [None, M]]
    if half == "diagonal":
        solver = S.BlockDiagonalSolver([S.LUSolver(), S.LUSolver()])
        P = orc.BlockPreconditioner([n, n], [(orc.BD_LU, M), (orc.BD_LU, M)], None, orc.DIAGONAL)
    else:
        solver = S.BlockTriangularSolver([S.LUSolver(), S.LUSolver()], half=half)
        P = orc.BlockPreconditioner([n, n], [(orc.BD_LU, M), (orc.BD_LU, M)], {(0, 1): (negM, 1.0), (1, 0): (M, 1.0)},
                                    orc.UPPER if half == "upper" else orc.LOWER)
    ns = setup(S, solver, mat)
    b = np.random.default_rng(0).uniform(-1, 1, 2 * n)
    x = np.zeros(2 * n)
    S.solve_(x, ns, b)
    assert rel_err(x, P.apply(b)) <= TOL_APPLY
    if half == "diagonal":
        lu = spla.splu(M.to_scipy().tocsc())
        assert np.linalg.norm(x - np.concatenate([lu.solve(b[:n]), lu.solve(b[n:])])) < 1e-8   # BlockDiagonalSolversTests.jl:45
    # mul!(y,A,x) on the block system
    K = sp.bmat([[None if m is None else m.to_scipy() for m in row] for row in mat]).tocsr()
    y = np.zeros(2 * n)
    ns.mul(y, b)
    assert rel_err(y, K @ b) <= 1e-13
    ns.close()


def test_block_coefficients_matrix_blocks_and_stateful_cg(S, po, orc):
    """coeffs scale / drop the off-diagonal blocks; MatrixBlock replaces the system's block in the preconditioner;
    the CG block solver starts from its previous result (work cache y)."""
    A = po.poisson_matrix((8, 8), 1); n = A.shape[0]
    C12 = _csr(po, sp.random(n, n, density=0.05, random_state=3, format="csr"))
    other = _csr(po, sp.random(n, n, density=0.05, random_state=7, format="csr"))
    b = np.random.default_rng(1).uniform(-1, 1, 2 * n)
    for c in (0.0, 1.0, -2.5):
        solver = S.BlockTriangularSolver([[S.LinearSystemBlock(), S.MatrixBlock(C12)], [S.LinearSystemBlock(), S.LinearSystemBlock()]],
                                         [S.LUSolver(), S.JacobiLinearSolver()], coeffs=[[1.0, c], [1.0, 1.0]], half="upper")
        ns = setup(S, solver, [[A, other], [None, A]])        # the system's (0,1) block differs from the preconditioner's
        x = np.zeros(2 * n)
        S.solve_(x, ns, b)
        P = orc.BlockPreconditioner([n, n], [(orc.BD_LU, A), (orc.BD_JACOBI, A)], {(0, 1): (C12, c)}, orc.UPPER)
        assert rel_err(x, P.apply(b)) <= TOL_APPLY
        ns.close()
    cg = S.CGSolver(S.JacobiLinearSolver(), maxiter=3, atol=1e-30, rtol=1e-30)
    ns = setup(S, S.BlockDiagonalSolver([cg]), [[A]])
    P = orc.BlockPreconditioner([n], [(orc.BD_CG_JACOBI, A, 3, 1e-30, 1e-30)], None, orc.DIAGONAL)
    for _ in range(2):
        x = np.zeros(n)
        S.solve_(x, ns, b[:n])
        assert rel_err(x, P.apply(b[:n])) <= 1e-10
        assert cg.log.num_iters == 3
    ns.close()


def _stokes_like(S, po, orc, hierarchy, nc, nlev, stabilised):
    """[[A, B^T], [B, C]]: C = 0 (saddle point, the Stokes shape) or C = the pressure block the preconditioner uses
    with a weak coupling (stabilised: the triangular preconditioner is then close to exact)."""
    H = hierarchy(nc, nlev)
    A = H["mats"][0]; n1 = A.shape[0]
    R = H["restrictions"][0]; n2 = R.shape[0]
    alpha = 10.0
    cb = 0.05 if stabilised else 0.5
    B = _csr(po, cb * R.to_scipy())
    Bt = _csr(po, cb * R.to_scipy().T)
    Mp = _csr(po, (-1.0 / alpha) * (sp.identity(n2) + 0.1 * H["mats"][1].to_scipy()))
    K = _csr(po, sp.bmat([[A.to_scipy(), Bt.to_scipy()], [B.to_scipy(), Mp.to_scipy() if stabilised else None]]))
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=4, rtol=1e-8)
    Po = orc.BlockPreconditioner([n1, n2], [go, (orc.BD_CG_JACOBI, Mp, 20, 1e-14, 1e-6)], {(0, 1): (Bt, 1.0), (1, 0): (B, 0.0)}, orc.UPPER)
    gmg = S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=jac(S, nlev), post_smoothers=jac(S, nlev),
                            maxiter=4, rtol=1e-8, mode="preconditioner")
    solver_p = S.CGSolver(S.JacobiLinearSolver(), maxiter=20, atol=1e-14, rtol=1e-6)
    blocks = [[S.LinearSystemBlock(), S.LinearSystemBlock()], [S.LinearSystemBlock(), S.MatrixBlock(Mp)]]
    Pd = S.BlockTriangularSolver(blocks, [gmg, solver_p], coeffs=[[1.0, 1.0], [0.0, 1.0]], half="upper")
    return dict(K=K, mat=[[A, Bt], [B, Mp if stabilised else None]], Po=Po, Pd=Pd, n=n1 + n2, n1=n1, gmg=gmg, solver_p=solver_p, keep=(go, Mp, B, Bt))


@pytest.mark.parametrize("nc,nlev,stabilised", [((8, 8, 8), 2, True), ((16, 16, 16), 3, True), ((16, 16, 16), 3, False)])
def test_fgmres_block_triangular_gmg_matches_oracle(S, po, orc, hierarchy, nc, nlev, stabilised):
    """StokesGMG.jl:142-153 solver shape: FGMRES(20, BlockTriangularSolver([GMG(maxiter=4), CG-Jacobi], [1 1;0 1], :upper)).
    The saddle-point variant needs ~60 non-restarted iterations with stagnation steps, at which the Arnoldi residual
    estimate amplifies rounding: its history is held to 1e-5 of the initial residual instead of 1e-6."""
    T = _stokes_like(S, po, orc, hierarchy, nc, nlev, stabilised)
    b = np.random.default_rng(5).uniform(-1, 1, T["n"])
    # one application of the preconditioner
    nsP = setup(S, T["Pd"], T["mat"])
    z = np.zeros(T["n"])
    S.solve_(z, nsP, b)
    zo = T["Po"].apply(b)
    assert rel_err(z, zo) <= 1e-9
    assert 0 < T["gmg"].log.num_iters <= 4 and 0 < T["solver_p"].log.num_iters <= 20
    nsP.close()
    # the outer solve (fresh setups: the block work caches are stateful)
    T = _stokes_like(S, po, orc, hierarchy, nc, nlev, stabilised)
    solver = S.FGMRESSolver(20, T["Pd"], atol=1e-10, rtol=1e-12, maxiter=100)
    ns = setup(S, solver, T["mat"])
    x = np.zeros(T["n"])
    S.solve_(x, ns, b)
    xo, nit, flag, hist = orc.fgmres_solve(T["K"], b, Pr=T["Po"], m=20, maxiter=100, atol=1e-10, rtol=1e-12)
    assert solver.log.num_iters == nit and solver.log.flag == flag
    assert np.all(np.abs(solver.log.residuals[: nit + 1] - hist) <= (TOL_HIST if stabilised else 1e-5) * hist[0])
    assert rel_err(x, xo) <= 1e-7
    assert np.linalg.norm(T["K"].to_scipy() @ x - b) < 1e-7      # StokesGMG.jl:166
    ns.P_ns.close()


def test_cg_block_diagonal_spd_system_and_device_tensors(S, po, orc, hierarchy):
    """CG on an SPD 2x2 block system preconditioned by BlockDiagonalSolver([GMG, LU]); device-resident vectors."""
    import torch
    nlev = 3
    H = hierarchy((16, 16, 16), nlev)
    A = H["mats"][0]; n1 = A.shape[0]
    n2 = H["mats"][1].shape[0]
    D = _csr(po, H["mats"][1].to_scipy() + sp.identity(n2))                  # weak coupling keeps the block system SPD
    Cc = _csr(po, 0.01 * H["prolongations"][0].to_scipy())
    Ct = _csr(po, 0.01 * H["prolongations"][0].to_scipy().T)
    K = _csr(po, sp.bmat([[A.to_scipy(), Cc.to_scipy()], [Ct.to_scipy(), D.to_scipy()]]))
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    Po = orc.BlockPreconditioner([n1, n2], [go, (orc.BD_LU, D)], None, orc.DIAGONAL)
    gmg = S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=jac(S, nlev), post_smoothers=jac(S, nlev), maxiter=1)
    solver = S.CGSolver(S.BlockDiagonalSolver([gmg, S.LUSolver()]), maxiter=50, atol=1e-14, rtol=1e-8)
    ns = setup(S, solver, [[A, Cc], [Ct, D]])
    b = np.random.default_rng(9).uniform(-1, 1, n1 + n2)
    xo, nit, flag, hist = orc.cg_solve(K, b, Pl=Po, maxiter=50, atol=1e-14, rtol=1e-8)
    x = np.zeros(n1 + n2)
    S.solve_(x, ns, b)
    assert solver.log.num_iters == nit and solver.log.flag == flag
    assert np.all(np.abs(solver.log.residuals[: nit + 1] - hist) <= 1e-8 * hist[0])
    assert rel_err(x, xo) <= 1e-9
    xd = torch.zeros(n1 + n2, dtype=torch.float64, device="cuda")
    bd = torch.from_numpy(b).cuda()
    S.solve_(xd, ns, bd)
    torch.cuda.synchronize()
    assert np.array_equal(xd.cpu().numpy(), x)                  # same kernels, same order: bit-identical
    ns.P_ns.close()


def test_block_error_behaviour(S, po, hierarchy, pkg):
    from gridapsolvers_jl_amd import abi
    lib = abi.load()
    A = po.poisson_matrix((8, 8), 1); n = A.shape[0]
    with pytest.raises(ValueError):
        S.BlockTriangularSolver([S.LUSolver(), S.LUSolver()], half="middle")
    with pytest.raises(NotImplementedError):
        setup(S, S.BlockDiagonalSolver([S.CGSolver(None)]), [[A]])           # unpreconditioned CG block: not on the device path
    h = C.c_void_p()
    sizes = np.array([n, n], dtype=np.int64)
    assert lib.gmg_block_create(C.byref(h), 2, sizes.ctypes.data, 7, 0) == abi.ERR_INVALID
    assert lib.gmg_block_create(C.byref(h), 2, sizes.ctypes.data, abi.BLOCK_UPPER, 0) == abi.OK
    v = np.zeros(2 * n)
    assert lib.gmg_block_precond_apply(h, v.ctypes.data, v.ctypes.data, abi.MEM_HOST) == abi.ERR_STATE      # before setup
    assert b"gmg_block_setup" in lib.gmg_block_last_error(h)
    assert lib.gmg_block_setup(h) == abi.ERR_STATE                                                          # no solvers
    assert lib.gmg_block_set_system_block(h, 0, 2, n, n, 0, None, None, None, 0, 0, 8) == abi.ERR_INVALID   # block index
    assert lib.gmg_block_set_system_block(h, 0, 1, n, n + 1, A.nnz, A.ptr.ctypes.data, A.idx.astype(np.int64).ctypes.data,
                                          A.val.ctypes.data, 0, 0, 8) == abi.ERR_INVALID                    # shape
    assert lib.gmg_block_set_diag_solver(h, 0, abi.BLOCK_LU, 0, 0.0, 0.0) == abi.OK
    assert lib.gmg_block_set_diag_solver(h, 1, abi.BLOCK_LU, 0, 0.0, 0.0) == abi.OK
    assert lib.gmg_block_setup(h) == abi.ERR_STATE                                                          # no matrix for the LU blocks
    # a GMG handle of the wrong size
    H = hierarchy((8, 8), 2)
    gns = setup(S, S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=jac(S, 2), post_smoothers=jac(S, 2)), H["mats"][0])
    sizes2 = np.array([n + 5], dtype=np.int64)
    h2 = C.c_void_p()
    assert lib.gmg_block_create(C.byref(h2), 1, sizes2.ctypes.data, abi.BLOCK_DIAGONAL, 0) == abi.OK
    assert lib.gmg_block_set_diag_gmg(h2, 0, gns.h) == abi.OK
    assert lib.gmg_block_setup(h2) == abi.ERR_INVALID
    lib.gmg_block_destroy(h2)
    lib.gmg_block_destroy(h)
    # the GMG handle is still usable on its own stream after the block handle is gone
    x = np.zeros(H["mats"][0].shape[0])
    S.solve_(x, gns, np.ones_like(x))
    assert np.isfinite(x).all() and np.abs(x).max() > 0
    gns.close()


# ---------------------------------------------------------------- real Stokes inputs: Q2 / discontinuous P1, grad-div augmented
def _real_stokes(S, po, orc, n, nlev, alpha=1.0e3):
    """test/Applications/StokesGMG.jl:79-166 on synthesised inputs (gridapsolvers.jl_amd/stokes.py): lid-driven cavity,
    Q2 x P1disc, a_u = grad:grad + alpha (div v) Pi_Qh(div u); velocity GMG with vertex-star patch smoothers
    Richardson(PatchSolver,10,0.2), patch prolongation with rhs = graddiv, LU coarsest, maxiter=4; pressure block
    -1/alpha M_p by CG-Jacobi; upper block-triangular preconditioner, FGMRES(20; atol=1e-10, rtol=1e-12)."""
    from gridapsolvers_jl_amd import stokes as st
    sysd = st.stokes_system(n, alpha)
    Hv = st.velocity_hierarchy(n, nlev, alpha)
    sm = [S.RichardsonSmoother(S.PatchSolver(pp, pd), 10, 0.2) for pp, pd in Hv["star_patches"]]
    interp = [S.PatchProlongationOperator(Hv["prolongations"][l], *Hv["interior_patches"][l], pivoting=True, rhs=Hv["graddiv"][l])
              for l in range(nlev - 1)]
    gmg = S.GMGLinearSolver(Hv["mats"], interp, Hv["restrictions"], pre_smoothers=sm, post_smoothers=sm,
                            coarsest_solver=S.LUSolver(), maxiter=4, mode="preconditioner")
    solver_p = S.CGSolver(S.JacobiLinearSolver(), maxiter=20, atol=1e-14, rtol=1e-6)
    blocks = [[S.LinearSystemBlock(), S.LinearSystemBlock()], [S.LinearSystemBlock(), S.MatrixBlock(sysd["Mp_scaled"])]]
    Pd = S.BlockTriangularSolver(blocks, [gmg, solver_p], coeffs=[[1.0, 1.0], [0.0, 1.0]], half="upper")
    osm = [orc.Smoother(orc.PATCH, 10, 0.2, pp, pd) for pp, pd in Hv["star_patches"]]
    go = orc.GMG(Hv["mats"], Hv["prolongations"], Hv["restrictions"], pre_smoothers=osm, maxiter=4, rtol=1e-8,
                 prolongation_patches=[(orc.PATCH, *Hv["interior_patches"][l], Hv["graddiv"][l]) for l in range(nlev - 1)])
    nu, npp = sysd["sizes"]
    Po = orc.BlockPreconditioner([nu, npp], [go, (orc.BD_CG_JACOBI, sysd["Mp_scaled"], 20, 1e-14, 1e-6)],
                                 {(0, 1): (sysd["A"][0][1], 1.0), (1, 0): (sysd["A"][1][0], 0.0)}, orc.UPPER)
    return sysd, Hv, gmg, solver_p, Pd, Po, go


@pytest.mark.parametrize("n,nlev", [(8, 2), (16, 3)])
def test_real_stokes_q2_p1disc_block_triangular_fgmres(S, po, orc, n, nlev):
    sysd, Hv, gmg, solver_p, Pd, Po, go = _real_stokes(S, po, orc, n, nlev)
    b = sysd["b"]
    N = b.size
    # (1) the velocity GMG alone (patch smoothers + patch prolongation with the grad-div rhs) vs the oracle
    nsg = setup(S, gmg, Hv["mats"][0])
    r = np.random.default_rng(2).uniform(-1, 1, sysd["sizes"][0])
    z = np.zeros_like(r)
    S.solve_(z, nsg, r)
    zo, nit_g, _, hist_g = go.solve(r)
    assert gmg.log.num_iters == nit_g
    assert rel_err(z, zo) <= 1e-8
    nsg.close()
    # (2) the whole solve: FGMRES(20) + upper block-triangular preconditioner
    sysd, Hv, gmg, solver_p, Pd, Po, go = _real_stokes(S, po, orc, n, nlev)    # fresh (stateful block caches)
    solver = S.FGMRESSolver(20, Pd, atol=1e-10, rtol=1e-12, maxiter=100)
    ns = setup(S, solver, sysd["A"])
    x = np.zeros(N)
    S.solve_(x, ns, b)
    Kc = po.CSR(sysd["K"].shape, sysd["K"].indptr, sysd["K"].indices, sysd["K"].data)
    xo, nit, flag, hist = orc.fgmres_solve(Kc, b, Pr=Po, m=20, maxiter=100, atol=1e-10, rtol=1e-12)
    assert solver.log.num_iters == nit and solver.log.flag == flag
    assert np.all(np.abs(solver.log.residuals[: nit + 1] - hist) <= 1e-6 * hist[0])
    assert np.linalg.norm(sysd["K"] @ x - b) < 1e-7                             # StokesGMG.jl:166
    assert rel_err(x, xo) <= 1e-6
    # discretely divergence-free velocity, lid velocity visible in the solution
    nu = sysd["sizes"][0]
    assert np.linalg.norm(sysd["A"][1][0].to_scipy() @ x[:nu] - b[nu:]) < 1e-7
    assert x[:nu].max() > 0.3
    ns.P_ns.close()


@pytest.mark.parametrize("n,T", [(32, 12), (24, 3), (40, 1)])
def test_vector_q2_velocity_operator_in_the_line_walk_form(S, po, orc, monkeypatch, capfd, n, T):
    """Round 6: the Stokes velocity operator (2-D, vector Q2 + grad-div: 5 grid lines x offsets -5 .. +5 on interleaved dofs) laid out as the
    5 x 5 grid of runs the walk form of the wide-row kernels takes (sellw_zwalk_kernel with "planes" = grid lines; option pat_wide_grid,
    default on) -- what the config-5 leg's dominant kernel (r -= A dx of the patch sweep) runs at 1024^2 cells.  Forced onto small levels
    (pat_zwalk = 2; chains of 12 / 3 / 1 lines): mul!(y, A, x) equals the oracle's row sums bit for bit, also against the per-slice kernel
    on the greedy run cover (pat_wide_grid = 0), with Inf / NaN in x; the patch-smoothed velocity GMG gives the same bits in both forms."""
    import importlib
    from gridapsolvers_jl_amd import abi
    st = importlib.import_module(po.__name__.rsplit(".", 1)[0] + ".stokes")
    monkeypatch.setenv("GMG_PAT_CODED_MIN_ROWS", "0")               # the coded shared-offset table on these small levels
    monkeypatch.setenv("GMG_SETUP_TIMING", "1")
    nlev = 2
    Hv = st.velocity_hierarchy(n, nlev, 1.0e3)
    A = Hv["mats"][0]
    nu = A.shape[0]
    x = np.random.default_rng(21).uniform(-1, 1, nu)
    xi = x.copy(); xi[nu // 2 + 1] = np.inf; xi[5] = np.nan
    r = np.random.default_rng(22).uniform(-1, 1, nu)
    res = {}
    for mode, opts in (("walk", {"pat_zwalk": 2, "pat_zwalk_T": T}), ("slice_grid", {"pat_zwalk": 0}), ("slice_greedy", {"pat_zwalk": 0, "pat_wide_grid": 0})):
        sm = [S.RichardsonSmoother(S.PatchSolver(pp, pd), 3, 0.2) for pp, pd in Hv["star_patches"]]
        gmg = S.GMGLinearSolver(Hv["mats"], Hv["prolongations"], Hv["restrictions"], pre_smoothers=sm, post_smoothers=sm,
                                coarsest_solver=S.LUSolver(), maxiter=2, mode="solver", options=opts)
        capfd.readouterr()
        ns = setup(S, gmg, A)
        y, yi, z = np.zeros(nu), np.zeros(nu), np.zeros(nu)
        ns.op_apply(0, abi.OP_A, x, y)
        ns.op_apply(0, abi.OP_A, xi, yi)
        S.solve_(z, ns, r)
        err = capfd.readouterr().err
        assert ("wide-row z-walk" in err) == (mode == "walk"), (mode, err[-600:])
        if mode == "walk":
            # both operators of the patch sweep walk: A (8-bit codes) and the additive-Schwarz operator (too many distinct values for
            # those: the auxiliary 16-bit coded table next to its plain one)
            assert err.count("wide-row z-walk") >= 2, err[-1200:]
        res[mode] = (y, np.isfinite(yi), np.where(np.isfinite(yi), yi, 0.0), z, gmg.log.num_iters)
        ns.close()
    yo = orc.spmv(A, x)
    for mode in res:
        assert np.array_equal(res[mode][0], yo), mode
        for a_, b_ in zip(res[mode], res["slice_greedy"]):
            assert np.array_equal(a_, b_), mode
    As = A.to_scipy().tocsc()
    touched = np.zeros(nu, dtype=bool)
    for k in (nu // 2 + 1, 5):
        touched[As.indices[As.indptr[k]:As.indptr[k + 1]]] = True
    assert np.array_equal(~res["walk"][1], touched)
